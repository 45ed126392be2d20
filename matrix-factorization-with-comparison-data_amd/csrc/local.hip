// local.hip — single-workgroup, LDS-resident form of the fused optimiser step for TINY problems (gfx950).
//
// Regime: (n+m)*d <= 8192 elements (BASELINE configs[0] n=m=256 d=8; the notebooks' default n=m=1000 d=2).
// There the register-resident multi-CU form is bound by its cross-CU hand-off (~8 us/step): the whole problem
// is smaller than one CU's LDS, so ONE workgroup of 16 waves runs every step of a call with the parameters in
// LDS, the Adam moments in the registers of the thread that owns the element, and
// workgroup barriers as the only synchronisation.  Any d >= 1 (no power-of-two restriction), any B <= 4096.
//   phase A  scores: a sample takes the next power of two >= d lanes (at most a wave), so one wave scores 64/LPS
//            samples at a time from the LDS rows (DPP reduction inside the lane group), sigmoid, backward
//            coefficient g_t -> LDS; the sigmoid output goes to the loss buffer.
//   phase B  row gradients by CLAIM ROUNDS: every pending (sample, role) does an LDS atomic-min of its batch
//            position into claim[row]; after a barrier the one winner of each row adds its contribution to the
//            row's accumulator in LDS (winners sit on distinct rows) and retires.  Round r therefore applies the
//            r-th hit of every row, in batch order (role i before role j inside a sample): same per-row
//            summation order as the other two forms, no float atomics, deterministic.  Rounds = the largest
//            number of hits on one row (3-5 at B = 64 on a few hundred rows); ONE barrier per round (three
//            claim arrays in rotation: read round r, write round r+1, retire round r-1).
//   phase C  dense Adam on the owned elements (flavour as mfcd_set_resident_math selects); the records of the
//            next batch, fetched from global memory during phase A, are staged into LDS.
// Same arithmetic as the other two forms (structure.py:847-851 per step).
#include "common.h"
#include "train_common.h"

namespace {

#ifndef MFCD_LOCAL_PACKED
#define MFCD_LOCAL_PACKED 1   // Adam on packed fp32 pairs (v_pk_*): bit-identical to the scalar form
#endif
constexpr unsigned kNoClaim = 0xFFFFFFFFu;
constexpr int kSlots = 2;       // hits a lane group owns per step
constexpr int kMaxLpsShift = 3; // at most 8 lanes per sample / per hit, i.e. at least 128 lane groups
constexpr int kLocalThreads = 1024;
constexpr int kRecSlots = 4;   // largest batch = 4 * 1024

struct LocalArgs {
    float *U, *V, *mU, *vU, *mV, *vV;
    const mfcd_sample *samples;
    const StepScalars *sc;   // [K]
    float *loss_terms;       // [N] sigmoid outputs (the finalize kernel forms the BCE terms)
    int64_t N;
    int B, n, m, d, K;
    int Tpad, Rpad, Bpad;    // LDS carve-up (elements / rows / batch, each padded to a multiple of 4)
    int lps_shift;           // lanes per sample in phase A = 1 << lps_shift
    AdamStatic ac;
};

// sum over aligned groups of (1 << shift) lanes; every lane of a group gets the group's sum
__device__ __forceinline__ float group_sum(float x, int shift)
{
    if (shift >= 6) return wave_sum64(x);
    if (shift >= 1) x += dpp_move<0xB1>(x);   // xor 1
    if (shift >= 2) x += dpp_move<0x4E>(x);   // xor 2
    if (shift >= 3) x += dpp_move<0x141>(x);  // row_half_mirror (== xor 4 once quads are uniform)
    if (shift >= 4) x += dpp_move<0x140>(x);  // row_mirror      (== xor 8)
    if (shift >= 5) x += __shfl_xor(x, 16, MFCD_WAVE);
    return x;
}

template <int QL, bool FAST, int RS>   // RS = records a thread stages per step (batch <= RS * 1024)
__global__ __launch_bounds__(kLocalThreads) void local_train_kernel(LocalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *pL = lds;                                              // [QL*1024] parameters, U rows then V rows, zero pad
    float *gacc = lds + QL * kLocalThreads;                       // [QL*1024] row gradients of the current step
    unsigned *claim = reinterpret_cast<unsigned *>(gacc + QL * kLocalThreads);  // [3][Rpad] per global row, by round % 3
    float *gco = reinterpret_cast<float *>(claim + 3 * a.Rpad);   // [Bpad] backward coefficients of the batch
    mfcd_sample *recs = reinterpret_cast<mfcd_sample *>(gco + a.Bpad);  // [B] records of the batch
    int *flags = reinterpret_cast<int *>(recs + a.B);             // [3] "a claim was placed in round r", by r % 3
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, n = a.n;
    const int TU = n * d, T = (n + a.m) * d;

    float m1[QL], m2[QL];
#pragma unroll
    for (int q = 0; q < QL; ++q) {
        const int e = tid + kLocalThreads * q;
        m1[q] = m2[q] = 0.0f;
        float pe = 0.0f;   // pad elements (e >= T): parameter 0, gradient 0, never written back
        if (e < T) {
            if (e < TU) { pe = a.U[e]; m1[q] = a.mU[e]; m2[q] = a.vU[e]; }
            else { pe = a.V[e - TU]; m1[q] = a.mV[e - TU]; m2[q] = a.vV[e - TU]; }
        }
        pL[e] = pe;
        gacc[e] = 0.0f;
    }
    for (int r = tid; r < 3 * a.Rpad; r += kLocalThreads) claim[r] = kNoClaim;

    auto fetch = [&](int step, mfcd_sample (&buf)[RS]) {
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            const int t = tid + kLocalThreads * s;
            const int64_t pos = (int64_t)step * a.B + t;
            buf[s].u = buf[s].i = buf[s].j = 0;
            buf[s].z = 0.0f;
            if (step < a.K && t < a.B && pos < a.N) buf[s] = a.samples[pos];
        }
    };
    auto stage = [&](const mfcd_sample (&buf)[RS]) {
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            const int t = tid + kLocalThreads * s;
            if (t < a.B) recs[t] = buf[s];
        }
    };
    mfcd_sample nxt[RS];
    fetch(0, nxt);
    stage(nxt);
    StepScalars sc = a.sc[0];
    __syncthreads();

    const int lps_shift = a.lps_shift, LPS = 1 << lps_shift, G = MFCD_WAVE >> lps_shift;
    const int sub = lane & (LPS - 1), grp = lane >> lps_shift;
    const int NG = kLocalThreads >> lps_shift, gid = tid >> lps_shift;   // lane groups of the workgroup

    for (int k = 0; k < a.K; ++k) {
        const int64_t pos0 = (int64_t)k * a.B;
        const int Bk = (int)((a.N - pos0) < a.B ? (a.N - pos0) : a.B);
        const float inv_batch = 1.0f / (float)Bk;
        fetch(k + 1, nxt);   // consumed in phase C
        const StepScalars sc_next = a.sc[k + 1 < a.K ? k + 1 : k];

        // hit h of the batch: h in [0,Bk) = role u of sample h, [Bk,2Bk) = role i, [2Bk,3Bk) = role j.  Lane group
        // gid (the LPS lanes that also score one sample) owns the hits h = gid and gid + NG (kSlots = 2 slots, so
        // 3B <= 2 NG) and keeps everything it needs about them in registers for the whole step.
        // Claim key = batch position: t on U rows, 2t + (role j) on V rows.
        const int H = 3 * Bk;
        int h_row[kSlots], h_t[kSlots], h_role[kSlots], h_a[kSlots], h_b[kSlots];
        unsigned h_key[kSlots];
        int h_state[kSlots];   // 0 pending, 1 applied in the last round (claim still to retire), 2 retired / no hit
#pragma unroll
        for (int sl = 0; sl < kSlots; ++sl) {
            const int h = gid + sl * NG;
            const int role = (h >= Bk) + (h >= 2 * Bk), t = h - role * Bk;
            const mfcd_sample rc = recs[h < H ? t : 0];
            h_state[sl] = h < H ? 0 : 2;
            h_role[sl] = role;
            h_t[sl] = h < H ? t : 0;
            h_row[sl] = role == 0 ? rc.u : n + (role == 1 ? rc.i : rc.j);
            h_key[sl] = role == 0 ? (unsigned)t : 2u * (unsigned)t + (unsigned)(role - 1);
            h_a[sl] = role == 0 ? TU + rc.i * d : rc.u * d;     // first operand row in pL
            h_b[sl] = role == 0 ? TU + rc.j * d : rc.u * d;     // second operand row (role u only)
        }
        unsigned *c_cur = claim, *c_nxt = claim + a.Rpad, *c_old = claim + 2 * a.Rpad;
        int f_cur = 0, f_nxt = 1, f_old = 2;

        // ---- phase A: round-0 claims; scores and backward coefficients ----
        if (sub == 0) {
#pragma unroll
            for (int sl = 0; sl < kSlots; ++sl)
                if (h_state[sl] == 0) atomicMin(&c_cur[h_row[sl]], h_key[sl]);
        }
        if (tid == 0) flags[1] = 0;
        for (int t0 = wave * G; t0 < Bk; t0 += 16 * G) {   // wave-uniform trip count: every lane runs the DPP steps
            const int t = t0 + grp;
            const bool valid = t < Bk;
            const mfcd_sample s = recs[valid ? t : 0];
            const float *ur = pL + s.u * d, *vi = pL + TU + s.i * d, *vj = pL + TU + s.j * d;
            float acc = 0.0f;
            for (int c = sub; c < d; c += LPS) acc += ur[c] * (vi[c] - vj[c]);
            const float pr = sigmoid_f32(group_sum(acc, lps_shift));
            if (valid && sub == 0) {
                gco[t] = bce_sigmoid_backward_f32(pr, s.z, inv_batch);
                a.loss_terms[pos0 + t] = pr;
            }
        }
        __syncthreads();

        // ---- phase B: claim rounds, one barrier each.  Between barrier r and barrier r+1 the claims of round r
        // (c_cur) are only read, those of round r+1 (c_nxt) only written and those of round r-1 (c_old) retired.
        // The winner of a row adds its contribution to the row's gradient accumulator in LDS: winners of one round
        // sit on distinct rows, so the read-modify-write needs no atomics.  Everything a slot might need is loaded
        // up front (one LDS round trip per round), then the outcome is decided. ----
        for (int r = 0;; ++r) {
            unsigned seen[kSlots];
            float hg[kSlots], va[kSlots], vb[kSlots], ga[kSlots];
#pragma unroll
            for (int sl = 0; sl < kSlots; ++sl) {
                seen[sl] = kNoClaim;
                hg[sl] = va[sl] = vb[sl] = ga[sl] = 0.0f;
                if (h_state[sl] == 0) {   // only pending hits pay for the loads (after round 0 most waves skip them)
                    seen[sl] = c_cur[h_row[sl]];
                    hg[sl] = gco[h_t[sl]];
                    const int c = sub < d ? sub : 0;
                    va[sl] = pL[h_a[sl] + c];
                    vb[sl] = pL[h_b[sl] + c];
                    ga[sl] = gacc[h_row[sl] * d + c];
                }
            }
            const bool more = r == 0 || flags[f_cur] != 0;   // workgroup-uniform: was anything claimed in round r?
            bool pending = false;
#pragma unroll
            for (int sl = 0; sl < kSlots; ++sl) {
                if (h_state[sl] == 1) {          // applied in round r-1: retire that claim
                    if (sub == 0) c_old[h_row[sl]] = kNoClaim;
                    h_state[sl] = 2;
                } else if (h_state[sl] == 0) {
                    if (seen[sl] == h_key[sl]) {  // this hit is the earliest pending one on its row: apply it
                        h_state[sl] = 1;
                        const float g = hg[sl];
                        float *dst = gacc + h_row[sl] * d;
                        const float *pa = pL + h_a[sl], *pb = pL + h_b[sl];
                        const int role = h_role[sl];
                        if (sub < d) {
                            const float v = role == 0 ? g * (va[sl] - vb[sl]) : g * va[sl];   // g (V[i]-V[j])  |  g U[u]
                            dst[sub] = ga[sl] + (role == 2 ? -v : v);
                        }
                        for (int c = sub + LPS; c < d; c += LPS) {
                            const float v = role == 0 ? g * (pa[c] - pb[c]) : g * pa[c];
                            dst[c] += role == 2 ? -v : v;
                        }
                    } else {
                        if (sub == 0) atomicMin(&c_nxt[h_row[sl]], h_key[sl]);
                        pending = true;
                    }
                }
            }
            if (!more) break;   // nothing was claimed in round r, so nothing was applied above: all claims retired
            if (pending && sub == 0) flags[f_nxt] = 1;
            if (tid == 0) flags[f_old] = 0;
            __syncthreads();
            unsigned *tc = c_old; c_old = c_cur; c_cur = c_nxt; c_nxt = tc;
            const int tf = f_old; f_old = f_cur; f_cur = f_nxt; f_nxt = tf;
        }
        // (no barrier here: the last contributions were written before the barrier that ended the last productive round)

        // ---- phase C: dense Adam on the owned elements (pad elements stay 0); stage the next batch ----
        {
            float pe[QL], gr[QL];
#pragma unroll
            for (int q = 0; q < QL; ++q) {
                pe[q] = pL[tid + kLocalThreads * q];
                gr[q] = gacc[tid + kLocalThreads * q];
                gacc[tid + kLocalThreads * q] = 0.0f;
            }
            adam_update_q<FAST, QL, MFCD_LOCAL_PACKED != 0>(pe, m1, m2, gr, a.ac, sc);
#pragma unroll
            for (int q = 0; q < QL; ++q) pL[tid + kLocalThreads * q] = pe[q];
        }
        stage(nxt);
        sc = sc_next;
        __syncthreads();
    }

#pragma unroll
    for (int q = 0; q < QL; ++q) {
        const int e = tid + kLocalThreads * q;
        if (e < T) {
            if (e < TU) { a.U[e] = pL[e]; a.mU[e] = m1[q]; a.vU[e] = m2[q]; }
            else { a.V[e - TU] = pL[e]; a.mV[e - TU] = m1[q]; a.vV[e - TU] = m2[q]; }
        }
    }
}

int local_ql(int n, int m, int d)
{
    const int ql = ((n + m) * d + kLocalThreads - 1) / kLocalThreads;
    return ql <= 1 ? 1 : ql <= 2 ? 2 : ql <= 4 ? 4 : 8;
}

size_t local_lds_bytes(int B, int n, int m, int d, LocalArgs *a)
{
    const int R = n + m;
    const int Tpad = local_ql(n, m, d) * kLocalThreads, Rpad = (R + 3) & ~3, Bpad = (B + 3) & ~3;
    if (a) { a->Tpad = Tpad; a->Rpad = Rpad; a->Bpad = Bpad; }
    return sizeof(float) * (2 * (size_t)Tpad + 3 * (size_t)Rpad + (size_t)Bpad) + sizeof(mfcd_sample) * (size_t)B + 16;
}

template <int QL, bool FAST, int RS>
int launch_local_inst(const LocalArgs &a, size_t lds_bytes, hipStream_t st)
{
    static size_t allowed = 0;   // per instantiation: raise the dynamic-LDS limit only when a launch needs more
    if (lds_bytes > allowed) {
        MFCD_HIP_TRY(hipFuncSetAttribute((const void *)local_train_kernel<QL, FAST, RS>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        allowed = lds_bytes;
    }
    hipLaunchKernelGGL((local_train_kernel<QL, FAST, RS>), dim3(1), dim3(kLocalThreads), lds_bytes, st, a);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

template <int QL>
int launch_local(const LocalArgs &a, size_t lds_bytes, hipStream_t st)
{
    // one CU does all the Adam arithmetic here, so the flavour matters even more than in the resident form
    const bool fast = mfcd_detail::g_resident_math != 0, small_batch = a.B <= kLocalThreads;
    if (fast) return small_batch ? launch_local_inst<QL, true, 1>(a, lds_bytes, st)
                                 : launch_local_inst<QL, true, kRecSlots>(a, lds_bytes, st);
    return small_batch ? launch_local_inst<QL, false, 1>(a, lds_bytes, st)
                       : launch_local_inst<QL, false, kRecSlots>(a, lds_bytes, st);
}

}  // namespace

namespace mfcd_detail {

bool local_applies(int64_t N, int B, int n, int m, int d)
{
    if (!(N > 0 && B >= 1 && B <= kRecSlots * kLocalThreads && d >= 1 && (int64_t)(n + m) * d <= kLocalMaxElems))
        return false;
    int lps_shift = 0;
    while ((1 << lps_shift) < d && lps_shift < kMaxLpsShift) ++lps_shift;
    if ((int64_t)3 * B > (int64_t)kSlots * (kLocalThreads >> lps_shift)) return false;   // hits per lane group
    return local_lds_bytes(B, n, m, d, nullptr) <= (size_t)160 * 1024;
}

int launch_local_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV, const mfcd_sample *samples,
                       int64_t N, int B, int n, int m, int d, const StepScalars *sc_dev, const AdamStatic &ac,
                       float *loss_terms, int K, hipStream_t st)
{
    LocalArgs a;
    a.U = U; a.V = V; a.mU = mU; a.vU = vU; a.mV = mV; a.vV = vV;
    a.samples = samples; a.sc = sc_dev; a.loss_terms = loss_terms;
    a.N = N; a.B = B; a.n = n; a.m = m; a.d = d; a.K = K; a.ac = ac;
    const size_t lds = local_lds_bytes(B, n, m, d, &a);
    a.lps_shift = 0;
    while ((1 << a.lps_shift) < d && a.lps_shift < kMaxLpsShift) ++a.lps_shift;
    const int ql = local_ql(n, m, d);
    if (ql <= 1) return launch_local<1>(a, lds, st);
    if (ql <= 2) return launch_local<2>(a, lds, st);
    if (ql <= 4) return launch_local<4>(a, lds, st);
    return launch_local<8>(a, lds, st);
}

}  // namespace mfcd_detail
