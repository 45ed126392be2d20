// local.hip — single-workgroup, LDS-resident form of the fused optimiser step for TINY problems (gfx950).
//
// Regime: (n+m)*d <= 16384 elements (BASELINE configs[0] n=m=256 d=8; the notebooks' default n=m=1000 d=2).
// There the register-resident multi-CU form is bound by its cross-CU hand-off (~8 us/step): the whole problem
// is smaller than one CU's LDS, so ONE workgroup of 16 waves runs every step of a call with the parameters in
// LDS (64 KiB) and the Adam moments in registers, and the only synchronisation is three workgroup barriers per
// step.  Any d >= 1 (no power-of-two restriction), any B <= 4096.
//   phase A  16 waves take the samples round-robin: x_t from the LDS rows (DPP wave reduction), sigmoid,
//            backward coefficient g_t -> LDS; the sigmoid output goes to the loss buffer.
//   phase B  row gradients into an LDS accumulator: global row r belongs to wave r % 16, every wave walks the
//            batch in order and handles its rows -> batch-order summation per row, no atomics, deterministic.
//   phase C  dense Adam: thread t owns elements t, t+1024, ...; p in LDS, m/v in registers (flavour as
//            mfcd_set_resident_math selects).
// Same arithmetic and per-row summation order as the other two forms (structure.py:847-851 per step).
#include "common.h"
#include "train_common.h"

namespace {

struct LocalArgs {
    float *U, *V, *mU, *vU, *mV, *vV;
    const mfcd_sample *samples;
    const StepScalars *sc;   // [K]
    float *loss_terms;       // [N] sigmoid outputs (the finalize kernel forms the BCE terms)
    int64_t N;
    int B, n, m, d, K, Tpad;
    AdamStatic ac;
};

template <int QL, bool FAST>
__global__ __launch_bounds__(1024) void local_train_kernel(LocalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *pL = lds;                 // [T]   parameters, U rows then V rows
    float *gacc = lds + a.Tpad;      // [T]   sparse row gradients of the current step
    float *gco = gacc + a.Tpad;      // [B]   backward coefficients of the current batch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d;
    const int TU = a.n * d, T = (a.n + a.m) * d;

    float m1[QL], m2[QL];
#pragma unroll
    for (int q = 0; q < QL; ++q) {
        const int e = tid + 1024 * q;
        m1[q] = m2[q] = 0.0f;
        if (e < T) {
            if (e < TU) { pL[e] = a.U[e]; m1[q] = a.mU[e]; m2[q] = a.vU[e]; }
            else { pL[e] = a.V[e - TU]; m1[q] = a.mV[e - TU]; m2[q] = a.vV[e - TU]; }
        }
    }
    __syncthreads();

    // lane t of EVERY wave holds record t of the current batch's first 64 samples; the next batch is fetched
    // while the current one is processed, so no phase waits on global memory for its records
    auto load_chunk0 = [&](int step) {
        mfcd_sample s;
        s.u = s.i = s.j = 0;
        s.z = 0.0f;
        const int64_t pos = (int64_t)step * a.B + lane;
        if (step < a.K && lane < a.B && pos < a.N) s = a.samples[pos];
        return s;
    };
    auto lane_sample = [&](const mfcd_sample &s, int tl) {
        mfcd_sample r;
        r.u = __builtin_amdgcn_readlane(s.u, tl);
        r.i = __builtin_amdgcn_readlane(s.i, tl);
        r.j = __builtin_amdgcn_readlane(s.j, tl);
        r.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.z), tl));
        return r;
    };
    mfcd_sample rec = load_chunk0(0);
    StepScalars sc = a.sc[0];

    for (int k = 0; k < a.K; ++k) {
        const int64_t pos0 = (int64_t)k * a.B;
        const int Bk = (int)((a.N - pos0) < a.B ? (a.N - pos0) : a.B);
        const float inv_batch = 1.0f / (float)Bk;
        const mfcd_sample rec_next = load_chunk0(k + 1);
        const StepScalars sc_next = a.sc[k + 1 < a.K ? k + 1 : k];

        // ---- phase A: coefficients; clear the gradient accumulator ----
#pragma unroll
        for (int q = 0; q < QL; ++q) {
            const int e = tid + 1024 * q;
            if (e < T) gacc[e] = 0.0f;
        }
        for (int t = wave; t < Bk; t += 16) {
            const mfcd_sample s = t < MFCD_WAVE ? lane_sample(rec, t) : a.samples[pos0 + t];
            const float *ur = pL + s.u * d, *vi = pL + TU + s.i * d, *vj = pL + TU + s.j * d;
            float acc = 0.0f;
            for (int c = lane; c < d; c += MFCD_WAVE) acc += ur[c] * (vi[c] - vj[c]);
            const float pr = sigmoid_f32(wave_sum64(acc));
            if (lane == 0) {
                gco[t] = bce_sigmoid_backward_f32(pr, s.z, inv_batch);
                a.loss_terms[pos0 + t] = pr;
            }
        }
        __syncthreads();

        // ---- phase B: row gradients, rows partitioned over the 16 waves, batch order per row ----
        for (int base = 0; base < Bk; base += MFCD_WAVE) {
            const int t = base + lane;
            mfcd_sample s;
            s.u = s.i = s.j = 0;
            s.z = 0.0f;
            const bool valid = t < Bk;
            if (base == 0) s = rec;
            else if (valid) s = a.samples[pos0 + t];
            const unsigned long long mu = __ballot(valid && (s.u & 15) == wave);
            const unsigned long long mi = __ballot(valid && ((s.i + a.n) & 15) == wave);
            const unsigned long long mj = __ballot(valid && ((s.j + a.n) & 15) == wave);
            unsigned long long mask = mu | mi | mj;
            while (mask) {
                const int tl = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int uu = __builtin_amdgcn_readlane(s.u, tl), ii = __builtin_amdgcn_readlane(s.i, tl),
                          jj = __builtin_amdgcn_readlane(s.j, tl);
                const float g = gco[base + tl];
                const float *ur = pL + uu * d, *vi = pL + TU + ii * d, *vj = pL + TU + jj * d;
                if ((mu >> tl) & 1ull) {
                    float *dst = gacc + uu * d;
                    for (int c = lane; c < d; c += MFCD_WAVE) dst[c] += g * (vi[c] - vj[c]);
                }
                if ((mi >> tl) & 1ull) {
                    float *dst = gacc + TU + ii * d;
                    for (int c = lane; c < d; c += MFCD_WAVE) dst[c] += g * ur[c];
                }
                if ((mj >> tl) & 1ull) {
                    float *dst = gacc + TU + jj * d;
                    for (int c = lane; c < d; c += MFCD_WAVE) dst[c] += -(g * ur[c]);
                }
            }
        }
        __syncthreads();

        // ---- phase C: dense Adam ----
#pragma unroll
        for (int q = 0; q < QL; ++q) {
            const int e = tid + 1024 * q;
            if (e < T) {
                float pe = pL[e];
                adam_update_t<FAST>(pe, m1[q], m2[q], gacc[e], a.ac, sc);
                pL[e] = pe;
            }
        }
        __syncthreads();
        rec = rec_next;
        sc = sc_next;
    }

#pragma unroll
    for (int q = 0; q < QL; ++q) {
        const int e = tid + 1024 * q;
        if (e < T) {
            if (e < TU) { a.U[e] = pL[e]; a.mU[e] = m1[q]; a.vU[e] = m2[q]; }
            else { a.V[e - TU] = pL[e]; a.mV[e - TU] = m1[q]; a.vV[e - TU] = m2[q]; }
        }
    }
}

template <int QL>
int launch_local(const LocalArgs &a, size_t lds_bytes, hipStream_t st)
{
    // one CU does all the Adam arithmetic here, so the flavour matters even more than in the resident form
    if (mfcd_detail::g_resident_math) {
        MFCD_HIP_TRY(hipFuncSetAttribute((const void *)local_train_kernel<QL, true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((local_train_kernel<QL, true>), dim3(1), dim3(1024), lds_bytes, st, a);
    } else {
        MFCD_HIP_TRY(hipFuncSetAttribute((const void *)local_train_kernel<QL, false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((local_train_kernel<QL, false>), dim3(1), dim3(1024), lds_bytes, st, a);
    }
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

namespace mfcd_detail {

bool local_applies(int64_t N, int B, int n, int m, int d)
{
    return N > 0 && B >= 1 && B <= 4096 && d >= 1 && (int64_t)(n + m) * d <= kLocalMaxElems;
}

int launch_local_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV, const mfcd_sample *samples,
                       int64_t N, int B, int n, int m, int d, const StepScalars *sc_dev, const AdamStatic &ac,
                       float *loss_terms, int K, hipStream_t st)
{
    LocalArgs a;
    a.U = U; a.V = V; a.mU = mU; a.vU = vU; a.mV = mV; a.vV = vV;
    a.samples = samples; a.sc = sc_dev; a.loss_terms = loss_terms;
    a.N = N; a.B = B; a.n = n; a.m = m; a.d = d; a.K = K; a.ac = ac;
    const int T = (n + m) * d;
    a.Tpad = (T + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)2 * a.Tpad + (size_t)B);
    const int ql = (T + 1023) / 1024;
    if (ql <= 1) return launch_local<1>(a, lds, st);
    if (ql <= 2) return launch_local<2>(a, lds, st);
    if (ql <= 4) return launch_local<4>(a, lds, st);
    if (ql <= 8) return launch_local<8>(a, lds, st);
    return launch_local<16>(a, lds, st);
}

}  // namespace mfcd_detail
