// big.hip — register-resident optimiser steps for states of up to 8 388 608 elements at d = 64 (BASELINE configs[3]:
// n = m = 65536, d = 64 is EXACTLY 1 024 SIMDs x 64 lanes x 128 rows), one MI355X.  Opt-in form (mfcd_train_steps_big).
//
// Same step as mfcd_train_steps (structure.py:845-852: forward, BCE, backward, dense coupled-L2 Adam) and the same
// arithmetic as its streaming form (adam_update, sigmoid_f32, bce_*: results are bit-identical to it), but the Adam state
// never leaves the chip during a call: ONE persistent launch of 2 048 waves, two per SIMD, each owning 64 rows of the
// virtual table [U; V] (MFCD_BIG_ROWS = 128: 1 024 waves, one per SIMD, 128 rows each — the first form built; a lone
// wave leaves its dependent chains' bubbles unfilled: 9.5 us per step at C4 against 5.9 for two waves of 64 rows):
//   * exp_avg and exp_avg_sq of the slice live in 2 x 64 registers per lane (2 x 128 at one wave per SIMD, where a lone
//     wave addresses 512 and the compiler places what does not fit the 256 architectural VGPRs in AccVGPRs), at
//     compile-time indices of the unrolled sweep;
//   * the parameters of the slice live in LDS (16 KB per wave, 64 KB per workgroup of four waves, two workgroups per
//     CU), where the sparse part of the step — hits of the batch on this wave's rows — can address a row by number;
//   * per step: every wave scans the batch's <= 64 records (one lane per sample), publishes the rows it owns that the
//     batch names (state after the previous step) as tagged 8-byte granules into the sample's mailbox slot, polls the
//     slots of the rows it does not own (bounded spins, sticky abort word), forms the coefficient and accumulates its
//     rows' gradients in batch order (a few LDS slots), then sweeps its 128 rows with the dense update.
// The sweep is bound by vector issue (128 registers per SIMD; the streaming form moves 24 bytes per element through
// HBM / the Infinity Cache instead), and C4's uniform streams carry short dependency chains (16-20 links per 1 049
// steps), so the publish-right-before-use protocol costs little here; a look-ahead window as in resident_kernel.h is
// not needed.
#include <vector>

#include "train_common.h"

namespace {

typedef unsigned long long u64;
#ifndef MFCD_BIG_ROWS
#define MFCD_BIG_ROWS 64      // measured at C4: 64-row slices at two waves per SIMD 5.9 us per step, 128-row slices at one 9.5
#endif
constexpr int kRows = MFCD_BIG_ROWS;                 // rows (= registers per state array) per wave: 128 or 64
constexpr int kRowsLog2 = kRows == 128 ? 7 : (kRows == 64 ? 6 : 5);
constexpr int kD = 64;
constexpr int kWavesPerSimd = 128 / kRows;           // 1 (512 registers per lane) or 2 (256 each)
constexpr int kWaves = 1024 * kWavesPerSimd;         // over the 1 024 SIMDs of the 256 CUs
constexpr int kSlots = kRows == 128 ? 16 : (kRows == 64 ? 8 : 6);   // distinct rows of a wave one batch may touch (more: the call is refused)
static_assert(kRows == 128 || kRows == 64 || kRows == 32, "slices of 128, 64 or 32 rows");
constexpr unsigned kSpinLimit = 1u << 22;

struct BigArgs {
    float *U, *V, *mU, *vU, *mV, *vV;
    const mfcd_sample *samples;
    long long N;
    int B, K, n, m;
    const StepScalars *sc;          // [K]
    AdamStatic ac;
    u64 *mailbox;                   // [N][3][64] granules {tag, fp32}, zeroed before the launch
    float *terms;                   // [N] BCE terms (the u-owner records)
    int *status;                    // sticky: 0 ok, 1 a bounded wait expired, 2 more than kSlots rows of a wave in a batch
};

__device__ __forceinline__ u64 pack(unsigned tag, float v) { return ((u64)tag << 32) | (u64)__float_as_uint(v); }

// FAST: the resident form's arithmetic (hardware sqrt / rcp / exp2 with one Newton correction per quotient, packed pairs:
// a few ulp from the IEEE sequence, ~3x fewer issue slots); !FAST: the streaming form's IEEE sequence, bit-identical to it.
template <bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWavesPerSimd, kWavesPerSimd))) void big_train_kernel(BigArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    float *prow = lds + (size_t)wave * kRows * kD;                          // [kRows][64] parameters of my rows
    float *grad = lds + (size_t)4 * kRows * kD + (size_t)wave * kSlots * kD; // [kSlots][64] sparse row gradients
    const int Rlo = gw * kRows, R = a.n + a.m;                              // my virtual rows: [Rlo, Rlo + kRows)

    float m1[kRows], m2[kRows];
#pragma unroll
    for (int q = 0; q < kRows; ++q) {
        const int v = Rlo + q;
        float p = 0.0f;
        m1[q] = m2[q] = 0.0f;
        if (v < R) {
            const bool item = v >= a.n;
            const long long o = (long long)(item ? v - a.n : v) * kD + lane;
            p = item ? a.V[o] : a.U[o];
            m1[q] = item ? a.mV[o] : a.mU[o];
            m2[q] = item ? a.vV[o] : a.vU[o];
        }
        prow[q * kD + lane] = p;
    }

    bool alive = true;
    // the batch records and Adam scalars of a step are requested one step ahead: their load latency sits under the
    // previous step's sweep instead of in front of every step
    mfcd_sample s_next;
    s_next.u = s_next.i = s_next.j = -1;
    s_next.z = 0.0f;
    if (lane < (a.N < a.B ? (int)a.N : a.B)) s_next = a.samples[lane];
    StepScalars sc_next = a.sc[0];
    for (int k = 0; k < a.K && alive; ++k) {
        const long long pos0 = (long long)k * a.B;
        const int Bk = (int)((a.N - pos0) < a.B ? (a.N - pos0) : a.B);
        const unsigned tag = (unsigned)k + 1u;
        const StepScalars sc = sc_next;
        // ---- which samples of the batch name my rows? (one lane per sample, virtual ids: users first, then items)
        const mfcd_sample s = s_next;
        if (k + 1 < a.K) {
            const long long pos1 = pos0 + a.B;
            const int B1 = (int)((a.N - pos1) < a.B ? (a.N - pos1) : a.B);
            s_next.u = s_next.i = s_next.j = -1;
            s_next.z = 0.0f;
            if (lane < B1) s_next = a.samples[pos1 + lane];
            sc_next = a.sc[k + 1];
        }
        const int vu = s.u - Rlo, vi = s.i + a.n - Rlo, vj = s.j + a.n - Rlo;
        const bool ou = lane < Bk && vu >= 0 && vu < kRows;
        const bool oi = lane < Bk && vi >= 0 && vi < kRows;
        const bool oj = lane < Bk && vj >= 0 && vj < kRows;
        const u64 mu = __ballot(ou), mi = __ballot(oi), mj = __ballot(oj);
        u64 mine = mu | mi | mj;
        u64 hit_lo = 0, hit_hi = 0;                 // rows of mine the batch names (bit = local row)
        int nslots = 0;
        int slot_row[kSlots];
#pragma unroll
        for (int x = 0; x < kSlots; ++x) slot_row[x] = -1;

        if (mine) {
            // ---- publish: my rows as they are after step k-1, into the slots of the samples that name them
            for (u64 w = mine; w; w &= w - 1) {
                const int t = __ffsll((long long)w) - 1;
                u64 *slot = a.mailbox + ((pos0 + t) * 3) * kD + lane;
                const int ru = __shfl(vu, t, MFCD_WAVE), ri = __shfl(vi, t, MFCD_WAVE), rj = __shfl(vj, t, MFCD_WAVE);
                if ((mu >> t) & 1ull)
                    __hip_atomic_store(slot, pack(tag, prow[ru * kD + lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((mi >> t) & 1ull)
                    __hip_atomic_store(slot + kD, pack(tag, prow[ri * kD + lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((mj >> t) & 1ull)
                    __hip_atomic_store(slot + 2 * kD, pack(tag, prow[rj * kD + lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ---- hits, in batch order: the other rows from their owners, the coefficient, my rows' gradients
            const float inv_batch = 1.0f / (float)Bk;
            for (u64 w = mine; w && alive; w &= w - 1) {
                const int t = __ffsll((long long)w) - 1;
                const bool own[3] = {(bool)((mu >> t) & 1ull), (bool)((mi >> t) & 1ull), (bool)((mj >> t) & 1ull)};
                const int lr[3] = {__shfl(vu, t, MFCD_WAVE), __shfl(vi, t, MFCD_WAVE), __shfl(vj, t, MFCD_WAVE)};
                const float zz = __shfl(s.z, t, MFCD_WAVE);
                const u64 *slot = a.mailbox + ((pos0 + t) * 3) * kD + lane;
                float row[3];
                unsigned spins = 0;
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        if (own[r]) {
                            row[r] = prow[lr[r] * kD + lane];
                        } else {
                            const u64 g = __hip_atomic_load(slot + r * kD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ok && (unsigned)(g >> 32) == tag;
                            row[r] = __uint_as_float((unsigned)g);
                        }
                    }
                    if (__all(ok)) break;
                    ++spins;
                    if (spins > kSpinLimit ||
                        ((spins & 255u) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        if (lane == 0) __hip_atomic_fetch_max(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        alive = false;
                        break;
                    }
                }
                if (!alive) break;
                const float x = wave_sum64(row[0] * (row[1] - row[2]));          // d = 64: one product per lane
                float pr, g;
                if constexpr (FAST) {      // as resident_kernel.h's fast hit path
                    const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896340736f);      // exp(-x)
                    const float dn = 1.0f + e;
                    pr = div_newton(1.0f, dn, __builtin_amdgcn_rcpf(dn));
                    pr = (e == __builtin_inff()) ? 0.0f : pr;
                    const float den = fmaxf((1.0f - pr) * pr, 1e-12f);
                    const float num = inv_batch * (pr - zz);
                    g = div_newton(num, den, __builtin_amdgcn_rcpf(den)) * (1.0f - pr) * pr;
                } else {
                    pr = sigmoid_f32(x);
                    g = bce_sigmoid_backward_f32(pr, zz, inv_batch);
                }
                if (own[0] && lane == 0) a.terms[pos0 + t] = bce_term_f32(pr, zz);
                const float contrib[3] = {g * (row[1] - row[2]), g * row[0], -(g * row[0])};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (!own[r]) continue;
                    int sl = -1;
#pragma unroll
                    for (int x2 = 0; x2 < kSlots; ++x2)
                        if (slot_row[x2] == lr[r]) sl = x2;
                    if (sl < 0) {
                        if (nslots == kSlots) {
                            if (lane == 0) __hip_atomic_fetch_max(a.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            alive = false;
                            break;
                        }
                        sl = nslots;
#pragma unroll
                        for (int x2 = 0; x2 < kSlots; ++x2)
                            if (x2 == nslots) slot_row[x2] = lr[r];
                        ++nslots;
                        grad[sl * kD + lane] = 0.0f;
                        if (lr[r] < 64) hit_lo |= 1ull << lr[r];
                        else hit_hi |= 1ull << (lr[r] & 63);
                    }
                    grad[sl * kD + lane] += contrib[r];
                }
            }
            if (!alive) break;
        }

        // ---- dense Adam over my 128 rows; the few rows the batch named take their sparse gradient from LDS
        // (rows in groups of kG, each group's LDS reads issued together and nothing moved across a group's end: the
        // scheduler would otherwise hoist dozens of reads and spill state registers to make room.  A lone wave per SIMD
        // has nobody to hide its dependent chains behind: kG / 2 independent packed chains per group do that.)
        constexpr int kG = 8;
        // two register sets: the next group's parameters are read from LDS while this group is updated (a lone wave has
        // nobody to hide the LDS latency behind: the counters showed 44 % of its cycles in s_waitcnt without this)
        float pa[kG], pb[kG];
#pragma unroll
        for (int x = 0; x < kG; ++x) pa[x] = prow[x * kD + lane];
#pragma unroll
        for (int g8 = 0; g8 < kRows / kG; ++g8) {
            const int q0 = g8 * kG;
            float (&pq)[kG] = (g8 & 1) ? pb : pa;
            float (&pn)[kG] = (g8 & 1) ? pa : pb;
            if (g8 + 1 < kRows / kG) {
#pragma unroll
                for (int x = 0; x < kG; ++x) pn[x] = prow[(q0 + kG + x) * kD + lane];
            }
            float gq[kG];
#pragma unroll
            for (int x = 0; x < kG; ++x) {
                const int q = q0 + x;
                gq[x] = 0.0f;
                const u64 hm = q < 64 ? hit_lo : hit_hi;
                if ((hm >> (q & 63)) & 1ull) {
                    int sl = 0;
#pragma unroll
                    for (int x2 = 0; x2 < kSlots; ++x2)
                        if (slot_row[x2] == q) sl = x2;
                    gq[x] = grad[sl * kD + lane];
                }
            }
            if constexpr (FAST) {
#pragma unroll
                for (int x = 0; x < kG; x += 2)
                    adam_update_fast2(pq[x], pq[x + 1], m1[q0 + x], m1[q0 + x + 1], m2[q0 + x], m2[q0 + x + 1], gq[x], gq[x + 1],
                                      a.ac, sc);
            } else {
#pragma unroll
                for (int x = 0; x < kG; ++x) adam_update(pq[x], m1[q0 + x], m2[q0 + x], gq[x], a.ac, sc);
            }
#pragma unroll
            for (int x = 0; x < kG; ++x) prow[(q0 + x) * kD + lane] = pq[x];
            asm volatile("" ::: "memory");
        }
    }

    // ---- the state goes back to the caller's tables (also after an abort: the status word says what it is worth)
#pragma unroll
    for (int q = 0; q < kRows; ++q) {
        const int v = Rlo + q;
        if (v < R) {
            const bool item = v >= a.n;
            const long long o = (long long)(item ? v - a.n : v) * kD + lane;
            const float p = prow[q * kD + lane];
            if (item) { a.V[o] = p; a.mV[o] = m1[q]; a.vV[o] = m2[q]; }
            else { a.U[o] = p; a.mU[o] = m1[q]; a.vU[o] = m2[q]; }
        }
    }
}

// mean of the batch's BCE terms in the fixed order of the streaming form's batch_mean_kernel (one wave per step)
__global__ __launch_bounds__(64) void big_batch_mean_kernel(const float *__restrict__ terms, long long N, int B,
                                                            float *__restrict__ out)
{
    const int lane = threadIdx.x;
    const long long off = (long long)blockIdx.x * B;
    const int b = (int)((N - off) < B ? (N - off) : B);
    float acc = 0.0f;
    for (int t = lane; t < b; t += MFCD_WAVE) acc += terms[off + t];
    acc = wave_sum64(acc);
    if (lane == 0) out[blockIdx.x] = acc / (float)b;
}

// per step: row references of the batch per wave slice (an upper bound of the distinct rows a wave must hold gradient
// slots for); the maximum over the call goes to *max_out
__global__ __launch_bounds__(256) void big_check_kernel(const mfcd_sample *__restrict__ samples, long long N, int B, int n,
                                                        int *__restrict__ max_out)
{
    __shared__ int bins[kWaves];
    for (int x = threadIdx.x; x < kWaves; x += 256) bins[x] = 0;
    __syncthreads();
    const long long off = (long long)blockIdx.x * B;
    const int b = (int)((N - off) < B ? (N - off) : B);
    for (int x = threadIdx.x; x < 3 * b; x += 256) {
        const mfcd_sample s = samples[off + x / 3];
        const int role = x % 3;
        const int v = role == 0 ? s.u : n + (role == 1 ? s.i : s.j);
        if (v >= 0 && (v >> kRowsLog2) < kWaves) atomicAdd(&bins[v >> kRowsLog2], 1);
    }
    __syncthreads();
    int mx = 0;
    for (int x = threadIdx.x; x < kWaves; x += 256) mx = bins[x] > mx ? bins[x] : mx;
    for (int o = 32; o > 0; o >>= 1) {
        const int y = __shfl_xor(mx, o, MFCD_WAVE);
        mx = y > mx ? y : mx;
    }
    if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(max_out, mx);
}

size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

constexpr size_t kLds = sizeof(float) * ((size_t)4 * kRows * kD + (size_t)4 * kSlots * kD);

}  // namespace

namespace mfcd_detail {
// train.hip: host-side Adam constants (f64 as Python computes them, rounded where ATen rounds)
AdamStatic big_adam_static(double beta1, double beta2, double eps, double wd);
StepScalars big_step_scalars(double lr, double beta1, double beta2, int64_t step);
}  // namespace mfcd_detail

extern "C" size_t mfcd_train_big_workspace_bytes(int64_t N, int B)
{
    if (N <= 0 || B <= 0 || B > 64) return 0;
    const size_t K = (size_t)((N + B - 1) / B);
    return 256 + up256(sizeof(StepScalars) * K) + up256(sizeof(float) * (size_t)N) + up256(sizeof(u64) * (size_t)N * 3 * kD);
}

extern "C" int mfcd_train_steps_big(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                                    const mfcd_sample *samples, int64_t N, int B, int64_t step0, int n, int m, int d,
                                    double lr, double beta1, double beta2, double eps, double weight_decay,
                                    float *loss_per_step, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!U || !V || !mU || !vU || !mV || !vV || N < 0 || step0 < 0 || n <= 0 || m <= 0) return MFCD_EINVAL;
    if (d != kD || B <= 0 || B > 64 || (int64_t)n + m > (int64_t)kWaves * kRows) return MFCD_EINVAL;
    if (N == 0) return 0;
    if (!samples || !workspace) return MFCD_EINVAL;
    if (workspace_bytes < mfcd_train_big_workspace_bytes(N, B)) return MFCD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int K = (int)((N + B - 1) / B);
    // one wave per SIMD, all of them resident at once: the polls rely on it
    const bool fast = mfcd_detail::g_resident_math != 0;      // mfcd_set_resident_math: 1 = fast (default), 0 = IEEE
    auto kernel = fast ? big_train_kernel<true> : big_train_kernel<false>;
    static int checked[2] = {0, 0};
    if (!checked[fast]) {
        int dev = 0, cus = 0, nb = 0;
        MFCD_HIP_TRY(hipGetDevice(&dev));
        MFCD_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        MFCD_HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        MFCD_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, kLds));
        if (cus * nb < kWaves / 4 || nb < kWavesPerSimd) return MFCD_EINVAL;
        checked[fast] = 1;
    }
    char *ws = (char *)workspace;
    int *status = (int *)ws;
    StepScalars *sc_dev = (StepScalars *)(ws + 256);
    float *terms = (float *)((char *)sc_dev + up256(sizeof(StepScalars) * (size_t)K));
    u64 *mailbox = (u64 *)((char *)terms + up256(sizeof(float) * (size_t)N));
    std::vector<StepScalars> sc((size_t)K);
    for (int k = 0; k < K; ++k) sc[(size_t)k] = mfcd_detail::big_step_scalars(lr, beta1, beta2, step0 + k + 1);
    MFCD_HIP_TRY(hipMemcpyAsync(sc_dev, sc.data(), sizeof(StepScalars) * (size_t)K, hipMemcpyHostToDevice, st));
    MFCD_HIP_TRY(hipStreamSynchronize(st));          // (the table is on this call's stack)
    MFCD_HIP_TRY(hipMemsetAsync(status, 0, 256, st));
    MFCD_HIP_TRY(hipMemsetAsync(mailbox, 0, sizeof(u64) * (size_t)N * 3 * kD, st));
    BigArgs a{U, V, mU, vU, mV, vV, samples, (long long)N, B, K, n, m, sc_dev,
              mfcd_detail::big_adam_static(beta1, beta2, eps, weight_decay), mailbox, terms, status};
    hipLaunchKernelGGL(kernel, dim3(kWaves / 4), dim3(256), kLds, st, a);
    MFCD_HIP_TRY(hipGetLastError());
    if (loss_per_step) {
        hipLaunchKernelGGL(big_batch_mean_kernel, dim3((unsigned)K), dim3(64), 0, st, terms, (long long)N, B, loss_per_step);
        MFCD_HIP_TRY(hipGetLastError());
    }
    return 0;
}

// status word of the last call on this workspace (device int32 at the workspace's start): 0 ok, 1 a bounded wait expired,
// 2 a batch named more than 16 distinct rows of one wave (the call is not valid for this sample stream)
extern "C" int mfcd_train_big_status(const void *workspace, int *status_out, void *stream)
{
    if (!workspace || !status_out) return MFCD_EINVAL;
    MFCD_HIP_TRY(hipMemcpyAsync(status_out, workspace, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    MFCD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// Largest number of row references one wave's slice receives from one batch of the stream (device int32, zeroed here):
// the form takes a call when this is <= mfcd_train_big_slots() (an upper bound of the distinct rows, so the test is
// conservative).  Enqueued on `stream`; the caller reads *max_out_dev.
extern "C" int mfcd_train_big_slots(void) { return kSlots; }

extern "C" int mfcd_train_big_check(const mfcd_sample *samples, int64_t N, int B, int n, int m, int *max_out_dev,
                                    void *stream)
{
    if (!samples || !max_out_dev || N <= 0 || B <= 0 || B > 64 || n <= 0 || m <= 0 || (int64_t)n + m > (int64_t)kWaves * kRows)
        return MFCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    MFCD_HIP_TRY(hipMemsetAsync(max_out_dev, 0, sizeof(int), st));
    hipLaunchKernelGGL(big_check_kernel, dim3((unsigned)((N + B - 1) / B)), dim3(256), 0, st, samples, (long long)N, B, n,
                       max_out_dev);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}
