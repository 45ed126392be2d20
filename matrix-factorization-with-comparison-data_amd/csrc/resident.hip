// resident.hip — persistent, register-resident form of the fused optimiser step (gfx950).
//
// Regime: the whole Adam state (p, m, v of U and V: 12*(n+m)*d bytes) fits in the chip's register
// files (256 CUs x 512 KiB).  Then one launch runs ALL optimiser steps of a call (an epoch), the
// state never leaves VGPRs between steps, and HBM sees it twice per launch instead of twice per step.
//
// Decomposition (wave = owner, no workgroup barrier, no grid barrier):
//   * the concatenated table [U;V] is cut into NW contiguous slices of EW = 64*Q elements (a multiple
//     of d, so rows are never split); wave w keeps p, m, v of slice w in 3*Q VGPRs per lane
//     (lane l, register q <-> element w*EW + q*64 + l);
//   * per step a wave scans the batch (B 16-byte records).  Samples that touch one of ITS rows are its
//     "hits"; for a hit it needs the two/three rows of that sample owned by other waves, as they are
//     after the previous step.  Owners publish exactly those rows right after their own update:
//     each value travels as ONE naturally aligned 8-byte {tag = step+1, fp32 value} granule written
//     with an agent-scope relaxed atomic store (write-through, sc1) into a per-sample mailbox slot;
//     consumers poll the granules with agent-scope relaxed loads until every tag matches
//     (cdna_hip_programming.md Guideline 16, form R2: the data is the flag; no fence, no flag word).
//     A wave's own row comes straight from its registers.
//   * a wave without a hit in a step waits for nobody: it applies the weight-decay gradient, runs Adam
//     on its registers and moves on, so producers run ahead of consumers.  Dependencies only point to
//     earlier steps, every wave is resident (grid <= what the chip holds, checked on the host), every
//     spin is bounded and observes a shared abort word: no deadlock, bounded run time.
//   * summation order per row is batch order, as in the streaming kernel; results are deterministic.
//
// The arithmetic is the same as train.hip's (common.h helpers): structure.py:847-851 per step.
#include <vector>

#include "common.h"
#include "train_common.h"

namespace {

typedef unsigned long long u64;

constexpr unsigned kSpinLimit = 1u << 22;  // polls before a wave gives up (~seconds); sets status = 1

__device__ __forceinline__ u64 pack_granule(unsigned tag, float v)
{
    return ((u64)tag << 32) | (u64)__float_as_uint(v);
}

__device__ __forceinline__ u64 load_granule(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_granule(u64 *p, unsigned tag, float v)
{
    __hip_atomic_store(p, pack_granule(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct ResidentArgs {
    float *U, *V, *mU, *vU, *mV, *vV;
    const mfcd_sample *samples;
    const StepScalars *sc;   // [K]
    u64 *mailbox;            // [N][3][D] granules, zero-filled before the launch
    float *loss_terms;       // [N]
    int *status;             // 0 = ok, 1 = a bounded spin expired
    int64_t N;
    int B, n, m, K, NW;
    AdamStatic ac;
};

// D: factor width (power of two <= 256).  Q: registers per array per lane (slice = 64*Q elements).
template <int D, int Q>
__global__ __launch_bounds__(256) void resident_train_kernel(ResidentArgs a)
{
    constexpr int S = D >= 64 ? D / 64 : 1;        // registers per gathered row
    constexpr int EW = 64 * Q;
    static_assert(EW % D == 0, "a wave's slice must hold whole rows");
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (gw >= a.NW) return;  // whole wave

    const int64_t TU = (int64_t)a.n * D, T = (int64_t)(a.n + a.m) * D;
    const int64_t ebase = (int64_t)gw * EW;
    const int Rlo = (int)(ebase / D);
    const int64_t eend = (ebase + EW < T) ? ebase + EW : T;
    const int Rhi = (int)(eend / D);  // my rows are global row ids [Rlo, Rhi); V rows are offset by n
    const int lcol = lane & (D - 1);  // column of my lane when D < 64

    // ---- load my slice of the state into registers ----
    float p[Q], m1[Q], m2[Q], gr[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t e = ebase + q * 64 + lane;
        p[q] = m1[q] = m2[q] = 0.0f;
        if (e < T) {
            if (e < TU) { p[q] = a.U[e]; m1[q] = a.mU[e]; m2[q] = a.vU[e]; }
            else { p[q] = a.V[e - TU]; m1[q] = a.mV[e - TU]; m2[q] = a.vV[e - TU]; }
        }
    }

    // publish the rows of mine that batch `step` needs (their values after step-1), tag = step+1
    auto publish = [&](int step) {
        const int64_t pos0 = (int64_t)step * a.B;
        const int Bk = (int)((a.N - pos0) < a.B ? (a.N - pos0) : a.B);
        const unsigned tag = (unsigned)step + 1u;
        for (int base = 0; base < Bk; base += MFCD_WAVE) {
            const int t = base + lane;
            mfcd_sample s;
            s.u = s.i = s.j = -0x40000000;
            s.z = 0.0f;
            if (t < Bk) s = a.samples[pos0 + t];
            const int ru = s.u, ri = s.i + a.n, rj = s.j + a.n;
            const u64 mu = __ballot(t < Bk && ru >= Rlo && ru < Rhi);
            const u64 mi = __ballot(t < Bk && ri >= Rlo && ri < Rhi);
            const u64 mj = __ballot(t < Bk && rj >= Rlo && rj < Rhi);
            u64 mask = mu | mi | mj;
            while (mask) {
                const int tl = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int rows[3] = {__shfl(ru, tl, MFCD_WAVE), __shfl(ri, tl, MFCD_WAVE), __shfl(rj, tl, MFCD_WAVE)};
                const bool fl[3] = {(bool)((mu >> tl) & 1ull), (bool)((mi >> tl) & 1ull), (bool)((mj >> tl) & 1ull)};
                const int64_t slot0 = (pos0 + base + tl) * 3;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (!fl[r]) continue;
                    u64 *dst = a.mailbox + (slot0 + r) * D;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const int o = q * 64 + lane;
                        if (Rlo + o / D == rows[r] && ebase + o < T) store_granule(dst + (o & (D - 1)), tag, p[q]);
                    }
                }
            }
        }
    };

    publish(0);

    for (int k = 0; k < a.K; ++k) {
        const int64_t pos0 = (int64_t)k * a.B;
        const int Bk = (int)((a.N - pos0) < a.B ? (a.N - pos0) : a.B);
        const float inv_batch = 1.0f / (float)Bk;
        const unsigned tag = (unsigned)k + 1u;
        const StepScalars sc = a.sc[k];
#pragma unroll
        for (int q = 0; q < Q; ++q) gr[q] = 0.0f;

        for (int base = 0; base < Bk; base += MFCD_WAVE) {
            const int t = base + lane;
            mfcd_sample s;
            s.u = s.i = s.j = -0x40000000;
            s.z = 0.0f;
            if (t < Bk) s = a.samples[pos0 + t];
            const int ru = s.u, ri = s.i + a.n, rj = s.j + a.n;
            const u64 mu = __ballot(t < Bk && ru >= Rlo && ru < Rhi);
            const u64 mi = __ballot(t < Bk && ri >= Rlo && ri < Rhi);
            const u64 mj = __ballot(t < Bk && rj >= Rlo && rj < Rhi);
            u64 mask = mu | mi | mj;
            while (mask) {
                const int tl = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int rows[3] = {__shfl(ru, tl, MFCD_WAVE), __shfl(ri, tl, MFCD_WAVE), __shfl(rj, tl, MFCD_WAVE)};
                const bool own[3] = {(bool)((mu >> tl) & 1ull), (bool)((mi >> tl) & 1ull), (bool)((mj >> tl) & 1ull)};
                const float zz = __shfl(s.z, tl, MFCD_WAVE);
                const int64_t pos = pos0 + base + tl;
                const u64 *slot = a.mailbox + pos * 3 * D;

                // the three rows in "gathered layout": lane holds column lane + 64*s (zero when >= D)
                float row[3][S];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2) row[r][s2] = 0.0f;
                    if (!own[r]) continue;
                    // my own row comes from my registers
                    if constexpr (D >= 64) {
#pragma unroll
                        for (int q = 0; q < Q; ++q)
                            if (Rlo + (q * 64) / D == rows[r]) row[r][(q * 64 % D) / 64] = p[q];
                    } else {
                        const int rl = rows[r] - Rlo;       // local row
                        const int q0 = rl / (64 / D);       // register that holds it
                        float sel = 0.0f;
#pragma unroll
                        for (int q = 0; q < Q; ++q) sel = (q == q0) ? p[q] : sel;
                        const float v = __shfl(sel, (rl % (64 / D)) * D + lcol, MFCD_WAVE);
                        row[r][0] = lane < D ? v : 0.0f;
                    }
                }
                // rows owned by other waves: poll their granules until every tag is this step's
                unsigned spins = 0;
                bool aborted = false;
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        if (own[r]) continue;
#pragma unroll
                        for (int s2 = 0; s2 < S; ++s2) {
                            const int c = lane + 64 * s2;
                            if (c < D) {
                                const u64 gq = load_granule(slot + (int64_t)r * D + c);
                                ok = ok && ((unsigned)(gq >> 32) == tag);
                                row[r][s2] = __uint_as_float((unsigned)gq);
                            }
                        }
                    }
                    if (__all(ok)) break;
                    ++spins;
                    if (spins > kSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED,
                                                                                      __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        aborted = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (aborted) {
                    if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;  // state is abandoned; the host reports the failure
                }

                float acc = 0.0f;
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) acc += row[0][s2] * (row[1][s2] - row[2][s2]);
                const float pr = sigmoid_f32(wave_sum64(acc));
                const float g = bce_sigmoid_backward_f32(pr, zz, inv_batch);
                if (own[0] && lane == 0) a.loss_terms[pos] = bce_term_f32(pr, zz);  // u's owner records the loss term

                // contribution vectors at MY lane's column for each register
                float du[S], dv[S];  // g*(vi - vj) and g*u, gathered layout
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) {
                    du[s2] = g * (row[1][s2] - row[2][s2]);
                    dv[s2] = g * row[0][s2];
                }
                if constexpr (D < 64) {
                    du[0] = __shfl(du[0], lcol, MFCD_WAVE);
                    dv[0] = __shfl(dv[0], lcol, MFCD_WAVE);
                }
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int rowq = Rlo + (q * 64 + lane) / D;
                    const int sq = D >= 64 ? (q * 64 % D) / 64 : 0;
                    if (own[0] && rowq == rows[0]) gr[q] += du[sq];
                    if (own[1] && rowq == rows[1]) gr[q] += dv[sq];
                    if (own[2] && rowq == rows[2]) gr[q] += -dv[sq];
                }
            }
        }

        // ---- dense Adam on my registers ----
#pragma unroll
        for (int q = 0; q < Q; ++q) adam_update(p[q], m1[q], m2[q], gr[q], a.ac, sc);

        if (k + 1 < a.K) publish(k + 1);
    }

    // ---- write my slice back ----
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t e = ebase + q * 64 + lane;
        if (e < T) {
            if (e < TU) { a.U[e] = p[q]; a.mU[e] = m1[q]; a.vU[e] = m2[q]; }
            else { a.V[e - TU] = p[q]; a.mV[e - TU] = m1[q]; a.vV[e - TU] = m2[q]; }
        }
    }
}

template <int D, int Q>
void launch_resident(const ResidentArgs &a, int blocks, hipStream_t st)
{
    if constexpr ((64 * Q) % D == 0) {
        hipLaunchKernelGGL((resident_train_kernel<D, Q>), dim3(blocks), dim3(256), 0, st, a);
    }
}

template <int D>
bool dispatch_q(int Q, const ResidentArgs &a, int blocks, hipStream_t st)
{
    switch (Q) {
        case 1: launch_resident<D, 1>(a, blocks, st); return true;
        case 2: launch_resident<D, 2>(a, blocks, st); return true;
        case 4: launch_resident<D, 4>(a, blocks, st); return true;
        case 8: launch_resident<D, 8>(a, blocks, st); return true;
        case 16: launch_resident<D, 16>(a, blocks, st); return true;
        case 32: launch_resident<D, 32>(a, blocks, st); return true;
        default: return false;
    }
}

}  // namespace

namespace mfcd_detail {

// Resident-path plan: Q registers per array, NW owner waves, or ok=false when the path does not apply.
// Every wave of the grid must be resident at once (waves wait on each other), so the wave count is
// bounded by what the register allocation of each instantiation admits per CU (kernel-resource-usage:
// Q<=8 -> <=128 VGPRs, Q=16 -> 176, Q=32 -> 256), kept at or below 2 workgroups of 4 waves per CU.
ResidentPlan plan_resident(int n, int m, int d, int num_cus)
{
    ResidentPlan pl{};
    pl.ok = false;
    if (d < 1 || d > 256 || (d & (d - 1)) != 0 || num_cus <= 0) return pl;
    const int64_t T = (int64_t)(n + m) * d;
    for (int Q = d > 64 ? d / 64 : 1; Q <= 32; Q *= 2) {
        const int waves_per_cu = Q <= 16 ? 8 : 4;
        const int64_t nw = (T + 64 * (int64_t)Q - 1) / (64 * Q);
        if (nw <= (int64_t)num_cus * waves_per_cu) {
            pl.ok = true;
            pl.Q = Q;
            pl.NW = (int)nw;
            pl.blocks = (pl.NW + 3) / 4;
            return pl;
        }
    }
    return pl;
}

int launch_resident_steps(const ResidentPlan &pl, float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                          const mfcd_sample *samples, int64_t N, int B, int n, int m, int d, const StepScalars *sc_dev,
                          const AdamStatic &ac, unsigned long long *mailbox, float *loss_terms, int *status, int K,
                          hipStream_t st)
{
    ResidentArgs a;
    a.U = U; a.V = V; a.mU = mU; a.vU = vU; a.mV = mV; a.vV = vV;
    a.samples = samples; a.sc = sc_dev; a.mailbox = mailbox; a.loss_terms = loss_terms; a.status = status;
    a.N = N; a.B = B; a.n = n; a.m = m; a.K = K; a.NW = pl.NW; a.ac = ac;
    bool ok = false;
    switch (d) {
        case 1: ok = dispatch_q<1>(pl.Q, a, pl.blocks, st); break;
        case 2: ok = dispatch_q<2>(pl.Q, a, pl.blocks, st); break;
        case 4: ok = dispatch_q<4>(pl.Q, a, pl.blocks, st); break;
        case 8: ok = dispatch_q<8>(pl.Q, a, pl.blocks, st); break;
        case 16: ok = dispatch_q<16>(pl.Q, a, pl.blocks, st); break;
        case 32: ok = dispatch_q<32>(pl.Q, a, pl.blocks, st); break;
        case 64: ok = dispatch_q<64>(pl.Q, a, pl.blocks, st); break;
        case 128: ok = dispatch_q<128>(pl.Q, a, pl.blocks, st); break;
        case 256: ok = dispatch_q<256>(pl.Q, a, pl.blocks, st); break;
        default: break;
    }
    if (!ok) return MFCD_EINVAL;
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mfcd_detail
