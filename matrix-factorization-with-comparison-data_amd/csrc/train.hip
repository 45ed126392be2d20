// train.hip — fused optimiser step for triplet-comparison MF on gfx950 (streaming form).
//
// One launch = one optimiser step of the reference loop (structure.py:847-851):
//   gather U[u],V[i],V[j] -> x -> sigmoid -> BCE backward coefficient g_t
//   -> row gradients accumulated in LDS (no dense gradient in HBM)
//   -> dense Adam with coupled L2 over every element (torch/optim/adam.py _single_tensor_adam).
//
// Ownership decomposition (no inter-workgroup communication inside a launch):
//   workgroup b owns a fixed flat range of E elements of one table (U or V) and the Adam moments
//   of that range.  It scans the batch (B 16-byte records), and for every sample that touches one
//   of its rows recomputes that sample's x_t from the INPUT copy of the tables and accumulates the
//   row gradient in LDS, in batch order (deterministic; all contributions to one row are handled
//   by one wave).  Parameters are ping-ponged (read Uin/Vin, write Uout/Vout) so that a workgroup
//   may read rows other workgroups are updating in the same launch; m and v are updated in place.
//   HBM traffic per element is the 24-byte minimum: read p,m,v, write p,m,v.
//
// Roofline: HBM-bound streaming; algorithmic bytes per step = 24*(n+m)*d + 12*B*d + 16*B.
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <type_traits>
#include <vector>

#include "common.h"
#include "rccl_dyn.h"
#include "train_common.h"

namespace {

struct AdamConst {
    AdamStatic st;
    StepScalars sc;
};

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    using T = float4;
};
template <>
struct Vec<1> {
    using T = float;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float *p, float (&r)[VEC])
{
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else {
        r[0] = *p;
    }
}

template <int VEC>
__device__ __forceinline__ void load_vec(const mfcd_bf16 *p, float (&r)[VEC])
{
    if constexpr (VEC == 4) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);   // 4 bf16 = 8 bytes
        r[0] = __uint_as_float(t.x << 16); r[1] = __uint_as_float(t.x & 0xffff0000u);
        r[2] = __uint_as_float(t.y << 16); r[3] = __uint_as_float(t.y & 0xffff0000u);
    } else {
        r[0] = (float)*p;
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(mfcd_bf16 *p, const float (&r)[VEC])
{
    if constexpr (VEC == 4) {   // round to nearest even, once per step (the defined rounding point)
        const unsigned short b0 = __builtin_bit_cast(unsigned short, (mfcd_bf16)r[0]);
        const unsigned short b1 = __builtin_bit_cast(unsigned short, (mfcd_bf16)r[1]);
        const unsigned short b2 = __builtin_bit_cast(unsigned short, (mfcd_bf16)r[2]);
        const unsigned short b3 = __builtin_bit_cast(unsigned short, (mfcd_bf16)r[3]);
        *reinterpret_cast<uint2 *>(p) = make_uint2((unsigned)b0 | ((unsigned)b1 << 16), (unsigned)b2 | ((unsigned)b3 << 16));
    } else {
        *p = (mfcd_bf16)r[0];
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float *p, const float (&r)[VEC])
{
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
        *p = r[0];
    }
}

// E = 256 * VEC * CHUNKS elements per workgroup.
// MODE 0: the fused step.  MODE 1 (data-parallel, before the all-reduce): only this rank's dense gradient,
// Gu/Gv[e] = sum of the local samples' row gradients (no Adam, parameters untouched).  MODE 2 (after the
// all-reduce): Adam from the dense gradient Gu/Gv, no batch scan.  MODE 3 (row-sharded state, mfcd_shard_*): the
// tables are this rank's SHARDS (rows [u_off, u_off + n) of U, [v_off, v_off + m) of V), the batch names GLOBAL rows,
// and the three rows of every sample come from the exchange buffer g_in = xbuf[role][g_stride][d] (the rows as they
// were before this step, gathered from their owners), so the update is in place; workgroup 0 also forms every
// sample's BCE term, which makes the step's loss available on every rank without a collective.
template <int VEC, int CHUNKS, int MODE = 0, typename TP = float>
__global__ __launch_bounds__(256) void train_step_kernel(
    const TP *__restrict__ Uin, const TP *__restrict__ Vin, TP *__restrict__ Uout,
    TP *__restrict__ Vout, float *__restrict__ mU, float *__restrict__ vU, float *__restrict__ mV,
    float *__restrict__ vV, const mfcd_sample *__restrict__ batch, const float *__restrict__ g_in,
    int Bk, float inv_batch, int n, int m, int d, int blocksU, AdamConst ac,
    float *__restrict__ loss_terms, float *__restrict__ Gu, float *__restrict__ Gv, int g_stride, int u_off = 0,
    int v_off = 0)
{
    constexpr int E = 256 * VEC * CHUNKS;
    extern __shared__ __attribute__((aligned(16))) float sg[];  // [(row_hi-row_lo)*d] sparse row gradients

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool isV = (int)blockIdx.x >= blocksU;
    const int tb = isV ? (int)blockIdx.x - blocksU : (int)blockIdx.x;
    const int64_t cnt = (int64_t)(isV ? m : n) * d;
    const int64_t e0 = (int64_t)tb * E;
    const int64_t e1 = (e0 + E < cnt) ? e0 + E : cnt;
    const int row_lo = (int)(e0 / d);
    const int row_hi = (int)((e1 + d - 1) / d);
    const int sg_off = (int)(e0 - (int64_t)row_lo * d);  // position of element e0 inside sg

    const TP *__restrict__ Pin = isV ? Vin : Uin;
    TP *__restrict__ Pout = isV ? Vout : Uout;
    float *__restrict__ M1 = isV ? mV : mU;
    float *__restrict__ M2 = isV ? vV : vU;
    float *__restrict__ G = isV ? Gv : Gu;

    // ---- phase 0: put this workgroup's p, m, v loads in flight before touching the batch ----
    float pr[CHUNKS][VEC], mr[CHUNKS][VEC], vr[CHUNKS][VEC];
    if constexpr (MODE != 1) {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int64_t e = e0 + (int64_t)(c * 256 + tid) * VEC;
            if (e < e1) {
                load_vec<VEC>(Pin + e, pr[c]);
                load_vec<VEC>(M1 + e, mr[c]);
                load_vec<VEC>(M2 + e, vr[c]);
            }
        }
    }

    // ---- phase 1: which samples of the batch touch my rows? ----
    int any = 0;
    for (int base = 0; MODE != 2 && base < Bk; base += MFCD_WAVE) {
        const int t = base + lane;
        if (t < Bk) {
            mfcd_sample s = batch[t];
            if constexpr (MODE == 3) { s.u -= u_off; s.i -= v_off; s.j -= v_off; }   // global -> shard-local rows
            if (isV)
                any |= (s.i >= row_lo && s.i < row_hi) | (s.j >= row_lo && s.j < row_hi);
            else
                any |= (s.u >= row_lo && s.u < row_hi);
        }
    }
    if constexpr (MODE != 2) any = __syncthreads_or(any);

    if constexpr (MODE == 3) {
        // every rank holds all three rows of every sample: workgroup 0 records all BCE terms (same dot-product order as
        // the owner-recorded term of MODE 0)
        if (blockIdx.x == 0 && loss_terms) {
            const float *xb = reinterpret_cast<const float *>(g_in);
            for (int t = wave; t < Bk; t += 4) {
                const float *ur = xb + (int64_t)t * d, *vi = xb + ((int64_t)g_stride + t) * d,
                            *vj = xb + ((int64_t)2 * g_stride + t) * d;
                float acc = 0.0f;
                for (int k = lane; k < d; k += MFCD_WAVE) acc += ur[k] * (vi[k] - vj[k]);
                const float p = sigmoid_f32(wave_sum64(acc));
                if (lane == 0) loss_terms[t] = bce_term_f32(p, batch[t].z);
            }
        }
    }

    if (MODE != 2 && any) {
        const int nsg = (row_hi - row_lo) * d;
        for (int k = tid; k < nsg; k += 256) sg[k] = 0.0f;
        __syncthreads();
        // every wave walks the batch in order and takes the rows congruent to its id (mod 4)
        for (int base = 0; base < Bk; base += MFCD_WAVE) {
            const int t = base + lane;
            mfcd_sample s;
            s.u = s.i = s.j = -1;
            s.z = 0.0f;
            if (t < Bk) {
                s = batch[t];
                if constexpr (MODE == 3) { s.u -= u_off; s.i -= v_off; s.j -= v_off; }
            }
            const bool hu = !isV && s.u >= row_lo && s.u < row_hi && ((s.u - row_lo) & 3) == wave;
            const bool hi = isV && s.i >= row_lo && s.i < row_hi && ((s.i - row_lo) & 3) == wave;
            const bool hj = isV && s.j >= row_lo && s.j < row_hi && ((s.j - row_lo) & 3) == wave;
            const unsigned long long mu = __ballot(hu), mi = __ballot(hi), mj = __ballot(hj);
            unsigned long long mask = mu | mi | mj;
            while (mask) {
                const int tl = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int uu = __shfl(s.u, tl, MFCD_WAVE), ii = __shfl(s.i, tl, MFCD_WAVE),
                          jj = __shfl(s.j, tl, MFCD_WAVE);
                const float zz = __shfl(s.z, tl, MFCD_WAVE);
                // MODE 3: the rows of sample base + tl as gathered before this step — always fp32 in the exchange buffer
                // (bf16 tables are widened exactly by the pack kernels)
                using XT = typename std::conditional<MODE == 3, float, TP>::type;
                const XT *ur, *vi, *vj;
                if constexpr (MODE == 3) {
                    ur = g_in + (int64_t)(base + tl) * d;
                    vi = g_in + ((int64_t)g_stride + base + tl) * d;
                    vj = g_in + ((int64_t)2 * g_stride + base + tl) * d;
                } else {
                    ur = Uin + (int64_t)uu * d;
                    vi = Vin + (int64_t)ii * d;
                    vj = Vin + (int64_t)jj * d;
                }
                float g;
                if (MODE != 3 && g_in) {
                    g = g_in[(base + tl) * g_stride];   // stride 2: interleaved {g, term} pairs of the DP exchange
                } else {
                    float acc = 0.0f;
                    for (int k = lane; k < d; k += MFCD_WAVE) acc += ldf(ur, k) * (ldf(vi, k) - ldf(vj, k));
                    const float p = sigmoid_f32(wave_sum64(acc));
                    g = bce_sigmoid_backward_f32(p, zz, inv_batch);
                    // the workgroup that owns the first element of row u records the loss term
                    if (MODE != 3 && ((mu >> tl) & 1ull) && loss_terms && lane == 0) {
                        const int64_t first = (int64_t)uu * d;
                        if (first >= e0 && first < e1) loss_terms[base + tl] = bce_term_f32(p, zz);
                    }
                }
                if ((mu >> tl) & 1ull) {
                    float *dst = sg + (int64_t)(uu - row_lo) * d;
                    for (int k = lane; k < d; k += MFCD_WAVE) dst[k] += g * (ldf(vi, k) - ldf(vj, k));
                }
                if ((mi >> tl) & 1ull) {
                    float *dst = sg + (int64_t)(ii - row_lo) * d;
                    for (int k = lane; k < d; k += MFCD_WAVE) dst[k] += g * ldf(ur, k);
                }
                if ((mj >> tl) & 1ull) {
                    float *dst = sg + (int64_t)(jj - row_lo) * d;
                    for (int k = lane; k < d; k += MFCD_WAVE) dst[k] += -(g * ldf(ur, k));
                }
            }
        }
        __syncthreads();
    }

    // ---- phase 2: dense Adam over my range (MODE 1: write the dense gradient instead) ----
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int loc = (c * 256 + tid) * VEC;
        const int64_t e = e0 + loc;
        if (e < e1) {
            float gs[VEC];
            if constexpr (MODE == 2) {
                load_vec<VEC>(G + e, gs);
            } else if (any) {
                load_vec<VEC>(sg + sg_off + loc, gs);
            } else {
#pragma unroll
                for (int q = 0; q < VEC; ++q) gs[q] = 0.0f;
            }
            if constexpr (MODE == 1) {
                store_vec<VEC>(G + e, gs);
            } else {
                float po[VEC], mo[VEC], vo[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    po[q] = pr[c][q];
                    mo[q] = mr[c][q];
                    vo[q] = vr[c][q];
                    adam_update(po[q], mo[q], vo[q], gs[q], ac.st, ac.sc);
                }
                store_vec<VEC>(Pout + e, po);
                store_vec<VEC>(M1 + e, mo);
                store_vec<VEC>(M2 + e, vo);
            }
        }
    }
}

// One wave per sample: sigmoid output, BCE term and backward coefficient (split DP form).
__global__ __launch_bounds__(256) void coeff_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                    const mfcd_sample *__restrict__ batch, int B, int d,
                                                    float inv_batch, float *__restrict__ g_out,
                                                    float *__restrict__ term_out, float *__restrict__ p_out)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= B) return;
    const mfcd_sample s = batch[t];
    const float p = sigmoid_f32(wave_score(U, V, s.u, s.i, s.j, d, lane));
    if (lane == 0) {
        if (g_out) g_out[t] = bce_sigmoid_backward_f32(p, s.z, inv_batch);
        if (term_out) term_out[t] = bce_term_f32(p, s.z);
        if (p_out) p_out[t] = p;
    }
}

// Data-parallel exchange slot of one rank for one step: B interleaved pairs {g_t, BCE term_t}; one wave per slot entry.
// Entries past the rank's (possibly short or empty) shard are written as {0, 0} so that the gathered buffer of a
// step is fully defined.
template <typename TP>
__global__ __launch_bounds__(256) void dp_coeff_kernel(const TP *__restrict__ U, const TP *__restrict__ V,
                                                       const mfcd_sample *__restrict__ shard, int myB, int B, int d,
                                                       float inv_batch, float2 *__restrict__ slot)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= B) return;
    if (t >= myB) {
        if (lane == 0) slot[t] = make_float2(0.0f, 0.0f);
        return;
    }
    const mfcd_sample s = shard[t];
    const float p = sigmoid_f32(wave_score(U, V, s.u, s.i, s.j, d, lane));
    if (lane == 0) slot[t] = make_float2(bce_sigmoid_backward_f32(p, s.z, inv_batch), bce_term_f32(p, s.z));
}

// out[k] = mean of the BCE terms of global batch k from the gathered exchange buffer xbuf[k][world*B] (pairs);
// one wave per step, fixed summation order (identical on every rank).
__global__ __launch_bounds__(64) void dp_loss_kernel(const float2 *__restrict__ xbuf, int64_t N, int Bg,
                                                     float *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int64_t off = (int64_t)blockIdx.x * Bg;
    const int b = (int)((N - off) < Bg ? (N - off) : Bg);
    float acc = 0.0f;
    for (int t = lane; t < b; t += MFCD_WAVE) acc += xbuf[off + t].y;
    acc = wave_sum64(acc);
    if (lane == 0) out[blockIdx.x] = acc / (float)b;
}

// out[k] = mean(terms[k*B .. min((k+1)*B,N))) — one wave per batch, fixed summation order.
// With `samples` set, terms[] holds sigmoid outputs p and the BCE term is formed here from p and the label
// (the resident kernel keeps the logs off its critical path); otherwise terms[] holds ready BCE terms.
__global__ __launch_bounds__(64) void batch_mean_kernel(const float *__restrict__ terms,
                                                        const mfcd_sample *__restrict__ samples, int64_t N, int B,
                                                        float *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int64_t off = (int64_t)blockIdx.x * B;
    const int b = (int)((N - off) < B ? (N - off) : B);
    float acc = 0.0f;
    for (int t = lane; t < b; t += MFCD_WAVE)
        acc += samples ? bce_term_f32(terms[off + t], samples[off + t].z) : terms[off + t];
    acc = wave_sum64(acc);
    if (lane == 0) out[blockIdx.x] = acc / (float)b;
}

struct Plan {
    int vec, chunks, E, blocksU, blocksV;
    size_t lds;
};

Plan make_plan(const void *const *ptrs, int nptrs, int n, int m, int d)
{
    Plan pl;
    bool al16 = (d % 4) == 0;
    for (int k = 0; k < nptrs; ++k) al16 = al16 && ((reinterpret_cast<uintptr_t>(ptrs[k]) & 15u) == 0);
    pl.vec = al16 ? 4 : 1;
    const int64_t total = (int64_t)(n + m) * d;
    pl.chunks = 1;
    // at most 4 chunks (16 KiB of each array per workgroup): measured at C3 / C4 / C5 size (profiles/r02_stream_chunks.txt),
    // 8 chunks cost 4-10 % (fewer workgroups in flight per CU: the per-workgroup LDS tile doubles)
    while (pl.chunks < 4 && total / (256 * pl.vec * pl.chunks) > 2048) pl.chunks *= 2;
    if (mfcd_detail::g_tune.stream_chunks) pl.chunks = mfcd_detail::g_tune.stream_chunks;   // experiment knob
    pl.E = 256 * pl.vec * pl.chunks;
    pl.blocksU = (int)(((int64_t)n * d + pl.E - 1) / pl.E);
    pl.blocksV = (int)(((int64_t)m * d + pl.E - 1) / pl.E);
    pl.lds = sizeof(float) * (size_t)(pl.E + 2 * d);
    return pl;
}

AdamStatic adam_static(double beta1, double beta2, double eps, double wd)
{
    AdamStatic a;
    a.w1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.wd = (float)wd;
    return a;
}

StepScalars step_scalars(double lr, double beta1, double beta2, int64_t step)
{
    // bias corrections in f64 as Python does (adam.py: 1 - beta**step, lr / bc1, bc2 ** 0.5)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    StepScalars s;
    s.neg_step_size = (float)(-(lr / bc1));
    s.bc2_sqrt = (float)sqrt(bc2);
    s.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    s.pad = 0.0f;
    return s;
}

}  // namespace
namespace mfcd_detail {   // big.hip builds its own per-step table
AdamStatic big_adam_static(double beta1, double beta2, double eps, double wd) { return adam_static(beta1, beta2, eps, wd); }
StepScalars big_step_scalars(double lr, double beta1, double beta2, int64_t step) { return step_scalars(lr, beta1, beta2, step); }
}  // namespace mfcd_detail
namespace {

AdamConst adam_const(double lr, double beta1, double beta2, double eps, double wd, int64_t step)
{
    AdamConst ac;
    ac.st = adam_static(beta1, beta2, eps, wd);
    ac.sc = step_scalars(lr, beta1, beta2, step);
    return ac;
}

template <int VEC, int CHUNKS, int MODE, typename TP>
void launch_step(const Plan &pl, hipStream_t st, const TP *Uin, const TP *Vin, TP *Uout, TP *Vout,
                 float *mU, float *vU, float *mV, float *vV, const mfcd_sample *batch, const float *g_in, int Bk,
                 float inv_batch, int n, int m, int d, const AdamConst &ac, float *loss_terms, float *Gu, float *Gv,
                 int g_stride, int u_off, int v_off)
{
    hipLaunchKernelGGL((train_step_kernel<VEC, CHUNKS, MODE, TP>), dim3(pl.blocksU + pl.blocksV), dim3(256), pl.lds, st,
                       Uin, Vin, Uout, Vout, mU, vU, mV, vV, batch, g_in, Bk, inv_batch, n, m, d, pl.blocksU, ac,
                       loss_terms, Gu, Gv, g_stride, u_off, v_off);
}

template <int MODE = 0, typename TP = float>
void dispatch_step(const Plan &pl, hipStream_t st, const TP *Uin, const TP *Vin, TP *Uout, TP *Vout,
                   float *mU, float *vU, float *mV, float *vV, const mfcd_sample *batch, const float *g_in, int Bk,
                   float inv_batch, int n, int m, int d, const AdamConst &ac, float *loss_terms, float *Gu = nullptr,
                   float *Gv = nullptr, int g_stride = 1, int u_off = 0, int v_off = 0)
{
#define MFCD_CASE(V, C)                                                                                              \
    if (pl.vec == V && pl.chunks == C)                                                                               \
        return launch_step<V, C, MODE, TP>(pl, st, Uin, Vin, Uout, Vout, mU, vU, mV, vV, batch, g_in, Bk, inv_batch, \
                                           n, m, d, ac, loss_terms, Gu, Gv, g_stride, u_off, v_off);
    MFCD_CASE(4, 1) MFCD_CASE(4, 2) MFCD_CASE(4, 4) MFCD_CASE(4, 8)
    MFCD_CASE(1, 1) MFCD_CASE(1, 2) MFCD_CASE(1, 4) MFCD_CASE(1, 8)
#undef MFCD_CASE
}

int check_common(const void *U, const void *V, int n, int m, int d)
{
    if (!U || !V || n <= 0 || m <= 0 || d <= 0 || d > MFCD_MAX_D) return MFCD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(U) & 3u) || (reinterpret_cast<uintptr_t>(V) & 3u)) return MFCD_EALIGN;
    return 0;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

namespace {

constexpr size_t kStatusBytes = 256;  // workspace[0..3] = int32 status word (sticky: set by an aborting resident launch)

int g_train_path = 0;  // 0 auto, 1 streaming, 2 resident, 3 local (mfcd_set_train_path)

int device_cus()
{
    static int cached = 0;
    if (!cached) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            cached = cus;
        else {
            (void)hipGetLastError();
            cached = 256;  // MI355X
        }
    }
    return cached;
}

constexpr size_t kColdBytes = 128;  // ResidentCold (resident_kernel.h), directly in front of the scalar table
constexpr size_t kDbgBytes = 256 + 2 * 16 * 256 * 8 * 8;  // [8] who gave up first + 2 banks of [<=4096 waves][8] u64 of the diagnostic builds (tools/)
constexpr size_t kMaxMailboxBytes = (size_t)24 << 30;  // beyond this the resident form is not planned
constexpr int kTagStepBits = 21;    // granule tag = launch id << 21 | (step + 1)
constexpr unsigned kMaxLaunchId = (1u << (32 - kTagStepBits)) - 1;

// Fixed carve-up of a registered workspace (mfcd_train_workspace_init), a function of the CAPACITY it was planned for:
//   status | dbg | stage (ResidentCold + K_cap+1 step scalars) | terms [N_cap] (8-byte tagged granules) | event-list
//   counters (resident; kept all-zero between launches) | event-list entries |
//   { U_alt, V_alt (streaming) } overlapping { translated samples, mailbox (resident) }
// Every call with N <= N_cap and ceil(N/B) <= K_cap uses these offsets, so nothing has to be re-initialised per call.
struct TrainLayout {
    size_t dbg_off, stage_off, stage_bytes, terms_off, evcnt_off, evcnt_bytes, event_off, event_bytes;
    size_t ualt_off, valt_off, xs_off, mailbox_off, mailbox_bytes, total;
    size_t alt_end;  // end of the streaming members of the union (U_alt, V_alt)
    // second set of the regions the prologue kernel writes (stage table, list counters, list entries, translated samples):
    // a call's prologue may be STAGED on a side stream under the previous call's step kernel (mfcd_train_call_stage)
    size_t stage2_off, evcnt2_off, event2_off, xs2_off;
    bool two_sets;
    int64_t K_cap;
    int64_t nch_cap;                    // chunks per wave of the event lists
    mfcd_detail::ResidentEvents ev;     // geometry of the event lists (tshift 0: none)
    bool resident;   // the resident regions exist
};

// geometric feasibility of the resident form (tuning knobs ignored: the layout must not depend on them)
bool resident_feasible(int n, int m, int d, int cus)
{
    if (d < 2 || d > 256 || (d & (d - 1)) != 0) return false;
    const int64_t T = (int64_t)(n + m) * d;
    static const int kQ[5] = {1, 2, 4, 16, 32};
    for (int Q : kQ) {
        if ((64 * Q) % d != 0) continue;
        const int64_t nw = (T + 64 * (int64_t)Q - 1) / (64 * Q);
        if (nw <= (int64_t)cus * (Q <= 2 ? 16 : 8) && nw <= mfcd_detail::kResidentMaxWaves) return true;
    }
    return false;
}

TrainLayout train_layout(int64_t N_cap, int B, int n, int m, int d)
{
    TrainLayout L{};
    const int64_t Nc = N_cap > 0 ? N_cap : 1;
    L.K_cap = (Nc + B - 1) / B;
    size_t off = kStatusBytes;
    L.dbg_off = off;
    off += kDbgBytes;
    L.stage_off = off;
    L.stage_bytes = kColdBytes + sizeof(StepScalars) * (size_t)(L.K_cap + 1);   // one pad entry: the kernel reads step k+1
    off += align256(L.stage_bytes);
    L.terms_off = off;
    off += align256(sizeof(unsigned long long) * (size_t)Nc);
    L.mailbox_bytes = sizeof(unsigned long long) * (size_t)Nc * 3 * (size_t)d;
    L.resident = resident_feasible(n, m, d, device_cus()) && L.mailbox_bytes <= kMaxMailboxBytes &&
                 L.K_cap < ((int64_t)1 << 31) - 64;
    L.ev = L.resident ? mfcd_detail::resident_events(B, n, m, d, device_cus()) : mfcd_detail::ResidentEvents{0, 0, 0};
    L.nch_cap = L.ev.tshift ? mfcd_detail::resident_event_chunks(L.K_cap, L.ev.tshift) : 0;
    L.evcnt_off = off;
    L.evcnt_bytes = sizeof(unsigned) * (size_t)L.ev.waves * (size_t)L.nch_cap;
    off += align256(L.evcnt_bytes);
    L.event_off = off;
    L.event_bytes = (size_t)16 * mfcd_detail::kResidentEventCap * (size_t)L.ev.waves * (size_t)L.nch_cap;
    off += align256(L.event_bytes);
    // streaming members of the union
    size_t a_off = off;
    L.ualt_off = a_off;
    a_off += align256(sizeof(float) * (size_t)n * d);
    L.valt_off = a_off;
    a_off += align256(sizeof(float) * (size_t)m * d);
    // resident members of the union
    size_t b_off = off;
    L.xs_off = b_off;
    L.mailbox_off = b_off;
    if (L.resident) {
        b_off += align256(sizeof(mfcd_sample) * (size_t)Nc);
        L.mailbox_off = b_off;
        b_off += align256(L.mailbox_bytes);
    }
    L.alt_end = a_off;
    L.total = a_off > b_off ? a_off : b_off;
    L.two_sets = L.resident && L.ev.tshift != 0;
    L.stage2_off = L.evcnt2_off = L.event2_off = L.xs2_off = 0;
    if (L.two_sets) {
        size_t c = align256(L.total);
        L.stage2_off = c; c += align256(L.stage_bytes);
        L.evcnt2_off = c; c += align256(L.evcnt_bytes);
        L.event2_off = c; c += align256(L.event_bytes);
        L.xs2_off = c; c += align256(sizeof(mfcd_sample) * (size_t)Nc);
        L.total = c;
    }
    return L;
}

size_t streaming_bytes(int64_t N, int n, int m, int d)   // mfcd_apply_step's own (unregistered) workspace
{
    return kStatusBytes + align256(sizeof(float) * (size_t)n * d) + align256(sizeof(float) * (size_t)m * d) +
           align256(sizeof(float) * (size_t)(N > 0 ? N : 1));
}

// ---- registered workspaces: the host-side state that belongs to one caller-owned workspace ----
// (one per model / stream; replaces the process-wide staging buffer of round 1).  A workspace is driven by one host
// thread at a time; the registry itself is guarded.
constexpr int kStageSlots = 4;

struct StageSlot {
    void *host = nullptr;      // pinned
    void *devview = nullptr;   // the same memory as the device addresses it
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
};

struct WsState {
    size_t bytes = 0;
    int device = 0;
    int64_t N_cap = 0;
    int B = 0, n = 0, m = 0, d = 0;
    TrainLayout L{};
    unsigned launch_id = 0;    // resident launches so far (mod kMaxLaunchId): the tag base of the next one
    // prologue staged ahead of its call (mfcd_train_call_stage): which call it belongs to and which set of regions it wrote
    struct Staged {
        bool valid = false;
        const void *samples = nullptr, *U = nullptr;
        int64_t N = 0, step0 = 0;
        float *loss = nullptr;
        int set = 0;
    } staged;
    int last_set = 0;             // set of regions the most recently enqueued persistent launch reads
    bool lists_dirty[2] = {false, false};   // a prologue filled this set's event lists and no launch has consumed them (the
                                  // launch zeroes the counters it read): a staged prologue whose call never came.  The
                                  // next prologue into the set zeroes the counters first instead of appending to them
    bool terms_dirty = false;     // a call that keeps plain fp32 terms (streaming / local / generic resident) wrote the term
                                  // region: zeroed before the next launch that reads it as tagged granules
    bool mailbox_dirty = false;   // a streaming-form call wrote U_alt / V_alt over the head of the mailbox (same union):
                                  // fp32 bit patterns there could pass for tagged granules, so the next resident launch
                                  // zeroes that prefix first
    StageSlot slot[kStageSlots];
    unsigned next = 0;
    ~WsState()
    {
        for (auto &s : slot) {
            if (s.ev) (void)hipEventDestroy(s.ev);
            if (s.host) (void)hipHostFree(s.host);
        }
    }
};

std::mutex g_ws_mu;
std::unordered_map<void *, std::unique_ptr<WsState>> g_ws;

WsState *find_ws(void *workspace)
{
    std::lock_guard<std::mutex> lock(g_ws_mu);
    auto it = g_ws.find(workspace);
    return it == g_ws.end() ? nullptr : it->second.get();
}

// A staging slot nobody reads any more, at least `need` bytes.  Blocks only when kStageSlots calls of this workspace
// are still queued on the device (bounded run-ahead of the host), never behind the previous call.
int stage_reserve(StageSlot &s, size_t need);

int stage_acquire(WsState &S, size_t need, StageSlot **out)
{
    StageSlot &s = S.slot[S.next++ % kStageSlots];
    if (s.pending) {
        MFCD_HIP_TRY(hipEventSynchronize(s.ev));
        s.pending = false;
    }
    if (int rc = stage_reserve(s, need)) return rc;
    *out = &s;
    return 0;
}

// pinned memory and event of one slot (both are created when the workspace is initialised: a pinned allocation costs
// ~0.1-0.2 ms of host time, which must not land on a training call)
int stage_reserve(StageSlot &s, size_t need)
{
    if (!s.ev) MFCD_HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    if (s.cap < need) {
        if (s.host) (void)hipHostFree(s.host);
        s.host = nullptr;
        s.cap = need < 4096 ? 4096 : need;
        // device-visible and coherent: the prologue kernel reads the slot directly over the host link
        if (hipHostMalloc(&s.host, s.cap, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            MFCD_HIP_TRY(hipHostMalloc(&s.host, s.cap, hipHostMallocDefault));
        }
        MFCD_HIP_TRY(hipHostGetDevicePointer(&s.devview, s.host, 0));
    }
    return 0;
}

}  // namespace

extern "C" int mfcd_set_train_path(int mode)
{
    if (mode < 0 || mode > 3) return MFCD_EINVAL;
    g_train_path = mode;
    return 0;
}

extern "C" int mfcd_set_resident_math(int fast)
{
    if (fast != 0 && fast != 1) return MFCD_EINVAL;
    mfcd_detail::g_resident_math = fast;
    return 0;
}

extern "C" int mfcd_set_tuning(int key, int64_t value)
{
    mfcd_detail::Tuning &t = mfcd_detail::g_tune;
    switch (key) {
        case MFCD_TUNE_RESIDENT_Q:
            if (value != 0 && value != 1 && value != 2 && value != 4 && value != 16) return MFCD_EINVAL;
            t.resident_q = (int)value;
            return 0;
        case MFCD_TUNE_RESIDENT_WPC:
            if (value != 8 && value != 16) return MFCD_EINVAL;
            t.resident_wpc = (int)value;
            return 0;
        case MFCD_TUNE_RESIDENT_LOOKAHEAD:
            if (value < -1 || value == 1 || value > 16) return MFCD_EINVAL;
            t.lookahead = (int)value;
            return 0;
        case MFCD_TUNE_RESIDENT_LDS_PAD:
            if (value < 0 || value > 160 * 1024) return MFCD_EINVAL;
            t.lds_pad = (int)value;
            return 0;
        case MFCD_TUNE_RESIDENT_SPIN_LIMIT:
            if (value < 0 || value > 0x7fffffff) return MFCD_EINVAL;
            t.spin_limit = value == 0 ? mfcd_detail::kSpinLimitDefault : (unsigned)value;
            return 0;
        case MFCD_TUNE_UVT_WPE128: return mfcd_detail::set_uvt_wpe128((int)value);
        case MFCD_TUNE_UVT_SPLIT: return mfcd_detail::set_uvt_split((int)value);
        case MFCD_TUNE_RANK_SORT: return mfcd_detail::set_rank_sort((int)value);
        case MFCD_TUNE_UVT_TARGET_WGS: return mfcd_detail::set_uvt_target_wgs((int)value);
        case MFCD_TUNE_UVT_MIN_STAGES: return mfcd_detail::set_uvt_min_stages((int)value);
        case MFCD_TUNE_STREAM_CHUNKS:
            if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return MFCD_EINVAL;
            t.stream_chunks = (int)value;
            return 0;
        case MFCD_TUNE_SHORT_CALL_STEPS:
            if (value < 0 || value > 0x7fffffff) return MFCD_EINVAL;
            t.short_call_steps = (int)value;
            return 0;
        case MFCD_TUNE_SHARD_PIPELINE:
            if (value < 0 || value > 2) return MFCD_EINVAL;
            t.shard_pipeline = (int)value;
            return 0;
        default: return MFCD_EINVAL;
    }
}


extern "C" size_t mfcd_train_workspace_bytes(int64_t N, int B, int n, int m, int d)
{
    if (N < 0 || B <= 0 || n <= 0 || m <= 0 || d <= 0) return 0;
    return train_layout(N, B, n, m, d).total;
}

extern "C" int mfcd_train_workspace_init(void *workspace, size_t workspace_bytes, int64_t N_cap, int B, int n, int m,
                                         int d, void *stream)
{
    if (!workspace || N_cap < 0 || B <= 0 || n <= 0 || m <= 0 || d <= 0 || d > MFCD_MAX_D) return MFCD_EINVAL;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return MFCD_EALIGN;
    const TrainLayout L = train_layout(N_cap, B, n, m, d);
    if (workspace_bytes < L.total) return MFCD_EWORKSPACE;
    // status word, touch strings, mailbox tags: everything a launch polls or ORs into starts from zero, once
    MFCD_HIP_TRY(hipMemsetAsync(workspace, 0, L.total, (hipStream_t)stream));
    auto S = std::make_unique<WsState>();
    S->bytes = workspace_bytes;
    (void)hipGetDevice(&S->device);
    S->N_cap = N_cap > 0 ? N_cap : 1;
    S->B = B; S->n = n; S->m = m; S->d = d;
    S->L = L;
    for (auto &sl : S->slot)
        if (int rc = stage_reserve(sl, L.stage_bytes)) return rc;
    std::lock_guard<std::mutex> lock(g_ws_mu);
    g_ws[workspace] = std::move(S);
    return 0;
}

extern "C" int mfcd_train_workspace_release(void *workspace)
{
    std::unique_ptr<WsState> dead;
    {
        std::lock_guard<std::mutex> lock(g_ws_mu);
        auto it = g_ws.find(workspace);
        if (it == g_ws.end()) return 0;
        dead = std::move(it->second);
        g_ws.erase(it);
    }
    for (auto &s : dead->slot)   // a slot a queued prologue still reads must outlive it
        if (s.pending) (void)hipEventSynchronize(s.ev);
    return 0;
}

namespace {

// form of the fused step a call with these sizes takes under the current settings
struct FormChoice {
    int form;   // 1 streaming, 2 resident, 3 local, <0 error
    mfcd_detail::ResidentPlan rp;
};

FormChoice choose_form(bool f32, bool resident_planned, int ev_tshift, int64_t N, int B, int n, int m, int d)
{
    FormChoice c{};
    const int64_t nsteps = (N + B - 1) / B;
    const bool local_ok = f32 && nsteps <= 0x7fffffff && mfcd_detail::local_applies(N, B, n, m, d);
    if (g_train_path == 3) {
        c.form = local_ok ? 3 : MFCD_EINVAL;
        return c;
    }
    if (g_train_path == 0 && local_ok) {   // measured 1.2-5.7x faster than the resident form wherever it applies
        c.form = 3;
        return c;
    }
    bool resident_ok = false;
    if (resident_planned && g_train_path != 1 && N > 0 && nsteps <= 0x7fffffff &&
        nsteps + 1 < ((int64_t)1 << kTagStepBits)) {
        c.rp = mfcd_detail::plan_resident(N, B, n, m, d, device_cus(), !f32, ev_tshift);
        resident_ok = c.rp.ok;
    }
    if (g_train_path == 2) {
        c.form = resident_ok ? 2 : MFCD_EINVAL;
        return c;
    }
    // auto: the persistent launch has a fixed cost (prologue kernel, slice load / store: ~10 us at C2) that a call of
    // one or two steps does not earn back against one streaming launch per step
    c.form = (resident_ok && nsteps >= mfcd_detail::g_tune.short_call_steps) ? 2 : 1;
    return c;
}

// Shared body of mfcd_train_steps / mfcd_train_steps_timed.  With `timing_us` set, every step launch is
// bracketed by its own pair of HIP events on the launch stream and the host waits for them at the end.
template <typename TP>
int run_train_steps(TP *U, TP *V, float *mU, float *vU, float *mV, float *vV, const mfcd_sample *samples,
                    int64_t N, int B, int64_t step0, int n, int m, int d, double lr, double beta1, double beta2,
                    double eps, double weight_decay, float *loss_per_step, void *workspace, size_t workspace_bytes,
                    void *stream, float *timing_us, bool stage_only = false)
{
    if (int rc = check_common(U, V, n, m, d)) return rc;
    if (!mU || !vU || !mV || !vV || N < 0 || B <= 0 || step0 < 0) return MFCD_EINVAL;
    if (N == 0) return 0;
    if (!samples || !workspace) return MFCD_EINVAL;
    WsState *S = find_ws(workspace);
    if (!S) return MFCD_ESTATE;   // mfcd_train_workspace_init has not been called on this workspace
    const int64_t nsteps = (N + B - 1) / B;
    if (n != S->n || m != S->m || d != S->d) return MFCD_ESTATE;
    if (N > S->N_cap || nsteps > S->L.K_cap || workspace_bytes < S->L.total) return MFCD_EWORKSPACE;
    const TrainLayout &L = S->L;
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)workspace;
    int *status = (int *)workspace;

    constexpr bool kF32 = sizeof(TP) == 4;   // bf16 factor tables: streaming or resident form (the local form is fp32 only)
    const FormChoice fc = choose_form(kF32, L.resident, L.ev.tshift, N, B, n, m, d);
    if (fc.form < 0) return fc.form;

    if (stage_only && !(fc.form == 2 && fc.rp.lookahead > 0 && L.two_sets)) return 0;   // nothing to stage for this form
    if (fc.form == 2 || fc.form == 3) {
        // ---- persistent forms: ONE launch for all nsteps (resident.hip / local.hip) behind ONE prologue kernel ----
        const bool resident = fc.form == 2;
        // look-ahead form (B <= 64): the batch means are formed inside the launch; otherwise by batch_mean_kernel
        const bool means_inside = resident && fc.rp.lookahead > 0;
        // which set of prologue-written regions this call uses: the one its prologue was staged into (if this is the
        // staged call), the other one when staging now, else the set of the last launch (free again in stream order)
        const bool staged_hit = !stage_only && S->staged.valid && S->staged.samples == (const void *)samples &&
                                S->staged.U == (const void *)U &&
                                S->staged.N == N && S->staged.step0 == step0 && S->staged.loss == loss_per_step &&
                                means_inside && !timing_us;
        const int set = staged_hit ? S->staged.set : (stage_only ? 1 - S->last_set : S->last_set);
        if (!stage_only && S->staged.valid && S->staged.set == set && !staged_hit) S->staged.valid = false;   // overwritten below
        const size_t stage_off = set ? L.stage2_off : L.stage_off, evcnt_off = set ? L.evcnt2_off : L.evcnt_off;
        const size_t event_off = set ? L.event2_off : L.event_off, xs_off = set ? L.xs2_off : L.xs_off;
        StepScalars *sc_dev = (StepScalars *)(base + stage_off + kColdBytes);
        void *terms = base + L.terms_off;
        StageSlot *slot = nullptr;
        const size_t need = kColdBytes + sizeof(StepScalars) * (size_t)(nsteps + 1);
        // short calls: the table is built on this thread's stack and copied into the prologue's kernel arguments at launch
        alignas(16) unsigned char inline_stage[4096];
        const bool stage_inline = need <= mfcd_detail::train_inline_stage_bytes() && need <= sizeof(inline_stage);
        if (!stage_inline && !staged_hit)
            if (int rc = stage_acquire(*S, need, &slot)) return rc;
        void *const stage_host = (stage_inline || staged_hit) ? (void *)inline_stage : slot->host;
        void **cold = (void **)stage_host;   // ResidentCold (resident_kernel.h)
        if (!staged_hit) {
        cold[0] = U; cold[1] = V; cold[2] = mU; cold[3] = vU; cold[4] = mV; cold[5] = vV; cold[6] = status;
        cold[7] = (void *)(uintptr_t)mfcd_detail::g_tune.spin_limit;
        cold[8] = base + evcnt_off;
        cold[9] = base + event_off;
        cold[10] = (void *)(uintptr_t)L.nch_cap;
        cold[11] = (void *)(uintptr_t)fc.rp.tshift;
        cold[12] = means_inside ? (void *)loss_per_step : nullptr;
        for (int k = 13; k < 16; ++k) cold[k] = nullptr;
        StepScalars *sc_host = (StepScalars *)((char *)stage_host + kColdBytes);
        for (int64_t k = 0; k <= nsteps; ++k) sc_host[k] = step_scalars(lr, beta1, beta2, step0 + k + 1);
        }

        mfcd_sample *xs = resident ? (mfcd_sample *)(base + xs_off) : nullptr;
        const int rpw = resident ? 64 * fc.rp.Q / d : 0;
        const int look = means_inside ? fc.rp.lookahead : 0;
        if (!staged_hit && means_inside && S->lists_dirty[set]) {
            MFCD_HIP_TRY(hipMemsetAsync(base + evcnt_off, 0, L.evcnt_bytes, st));
            S->lists_dirty[set] = false;
        }
        if (stage_only) {
            // the prologue of a LATER call, on the caller's side stream: stage table, translated samples, event lists
            S->lists_dirty[set] = true;
            if (int rc = mfcd_detail::launch_train_prologue(stage_host, stage_inline ? nullptr : slot->devview, base + stage_off,
                                                            need, samples, N, B, n, m, rpw, fc.rp.tshift, look, L.nch_cap, xs,
                                                            (unsigned *)(base + evcnt_off), base + event_off, st))
                return rc;
            if (slot) {
                MFCD_HIP_TRY(hipEventRecord(slot->ev, st));
                slot->pending = true;
            }
            S->staged.valid = true;
            S->staged.samples = samples; S->staged.U = U; S->staged.N = N; S->staged.step0 = step0; S->staged.loss = loss_per_step;
            S->staged.set = set;
            return 0;
        }
        unsigned long long *mailbox = (unsigned long long *)(base + L.mailbox_off);
        unsigned tag_base = 0;
        if (resident) {
            if (S->launch_id >= kMaxLaunchId) {   // the launch ids wrap: forget every granule of the past, once
                MFCD_HIP_TRY(hipMemsetAsync(mailbox, 0, L.mailbox_bytes, st));
                MFCD_HIP_TRY(hipMemsetAsync(terms, 0, sizeof(unsigned long long) * (size_t)S->N_cap, st));
                S->launch_id = 0;
            }
            tag_base = ++S->launch_id << kTagStepBits;
            if (S->mailbox_dirty) {
                if (L.alt_end > L.mailbox_off) {
                    const size_t nb = L.alt_end - L.mailbox_off;
                    MFCD_HIP_TRY(hipMemsetAsync(mailbox, 0, nb < L.mailbox_bytes ? nb : L.mailbox_bytes, st));
                }
                S->mailbox_dirty = false;
            }
            if (S->terms_dirty) {   // a form that keeps plain fp32 terms ran on this workspace: no stale bit pattern may
                                    // pass for a tagged term
                MFCD_HIP_TRY(hipMemsetAsync(terms, 0, sizeof(unsigned long long) * (size_t)S->N_cap, st));
                S->terms_dirty = false;
            }
        }
        if (!means_inside) S->terms_dirty = true;
        if (staged_hit) {
            S->staged.valid = false;          // consumed: its prologue ran on the side stream (the caller ordered the streams)
        } else if (int rc = mfcd_detail::launch_train_prologue(stage_host, stage_inline ? nullptr : slot->devview,
                                                               base + stage_off, need, samples, N, B, n, m, rpw,
                                                               fc.rp.tshift, look, L.nch_cap, xs,
                                                               (unsigned *)(base + evcnt_off), base + event_off, st))
            return rc;
        if (resident) {
            S->last_set = set;
            S->lists_dirty[set] = false;      // the launch below reads the lists and leaves their counters at zero
        }

        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timing_us) {
            MFCD_HIP_TRY(hipEventCreate(&e0));
            MFCD_HIP_TRY(hipEventCreate(&e1));
            MFCD_HIP_TRY(hipEventRecord(e0, st));
        }
        int rc = 0;
        if (resident)
            rc = mfcd_detail::launch_resident_steps(fc.rp, base + stage_off, xs, N, B, n, m, d, sc_dev,
                                                    adam_static(beta1, beta2, eps, weight_decay), mailbox, tag_base,
                                                    terms, (unsigned long long *)(base + L.dbg_off), (int)nsteps, st);
        else
            rc = mfcd_detail::launch_local_steps((float *)U, (float *)V, mU, vU, mV, vV, samples, N, B, n, m, d, sc_dev,
                                                 adam_static(beta1, beta2, eps, weight_decay), (float *)terms,
                                                 (int)nsteps, st);
        if (rc) return rc;
        if (timing_us) MFCD_HIP_TRY(hipEventRecord(e1, st));
        if (loss_per_step && !means_inside) {
            hipLaunchKernelGGL(batch_mean_kernel, dim3((unsigned)nsteps), dim3(64), 0, st, (const float *)terms, samples, N,
                               B, loss_per_step);
            MFCD_HIP_TRY(hipGetLastError());
        }
        // the slot is free again once the prologue has read it; recorded behind the call's last launch so that the
        // record does not sit between two launches (any later point of the stream implies the prologue is done)
        if (slot) {
            MFCD_HIP_TRY(hipEventRecord(slot->ev, st));
            slot->pending = true;
        }
        if (timing_us) {
            MFCD_HIP_TRY(hipEventSynchronize(e1));
            float ms = 0.0f;
            MFCD_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            timing_us[0] = timing_us[1] = timing_us[2] = ms * 1e3f / (float)nsteps;  // whole launch / steps
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        return 0;
    }

    // ---- streaming form: one launch per optimiser step ----
    if (L.resident) S->mailbox_dirty = S->terms_dirty = true;
    TP *Ualt = (TP *)(base + L.ualt_off);
    TP *Valt = (TP *)(base + L.valt_off);
    float *terms = (float *)(base + L.terms_off);

    const void *ptrs[] = {U, V, mU, vU, mV, vV, Ualt, Valt};
    const Plan pl = make_plan(ptrs, 8, n, m, d);
    std::vector<hipEvent_t> ev;
    if (timing_us) {
        ev.resize(2 * (size_t)nsteps);
        for (auto &e : ev) MFCD_HIP_TRY(hipEventCreate(&e));
    }
    for (int64_t k = 0; k < nsteps; ++k) {
        const int64_t off = k * B;
        const int Bk = (int)((N - off) < B ? (N - off) : B);
        const AdamConst ac = adam_const(lr, beta1, beta2, eps, weight_decay, step0 + k + 1);
        const bool even = (k & 1) == 0;
        if (timing_us) MFCD_HIP_TRY(hipEventRecord(ev[2 * k], st));
        dispatch_step<0, TP>(pl, st, even ? U : Ualt, even ? V : Valt, even ? Ualt : U, even ? Valt : V, mU, vU, mV, vV,
                             samples + off, nullptr, Bk, 1.0f / (float)Bk, n, m, d, ac, terms + off);
        if (timing_us) MFCD_HIP_TRY(hipEventRecord(ev[2 * k + 1], st));
    }
    MFCD_HIP_TRY(hipGetLastError());
    if (nsteps & 1) {
        MFCD_HIP_TRY(hipMemcpyAsync(U, Ualt, sizeof(TP) * (size_t)n * d, hipMemcpyDeviceToDevice, st));
        MFCD_HIP_TRY(hipMemcpyAsync(V, Valt, sizeof(TP) * (size_t)m * d, hipMemcpyDeviceToDevice, st));
    }
    if (loss_per_step) {
        hipLaunchKernelGGL(batch_mean_kernel, dim3((unsigned)nsteps), dim3(64), 0, st, terms,
                           (const mfcd_sample *)nullptr, N, B, loss_per_step);
        MFCD_HIP_TRY(hipGetLastError());
    }
    if (timing_us) {
        MFCD_HIP_TRY(hipEventSynchronize(ev.back()));
        double sum = 0.0;
        float mn = 1e30f, mx = 0.0f;
        for (int64_t k = 0; k < nsteps; ++k) {
            float ms = 0.0f;
            MFCD_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
            sum += ms;
            mn = ms < mn ? ms : mn;
            mx = ms > mx ? ms : mx;
        }
        timing_us[0] = (float)(sum / (double)nsteps * 1e3);
        timing_us[1] = mn * 1e3f;
        timing_us[2] = mx * 1e3f;
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    return 0;
}

}  // namespace

extern "C" int mfcd_train_plan_query(int64_t N, int B, int n, int m, int d, int bf16_factors, mfcd_train_plan *out)
{
    if (!out || N < 0 || B <= 0 || n <= 0 || m <= 0 || d <= 0 || d > MFCD_MAX_D) return MFCD_EINVAL;
    std::memset(out, 0, sizeof(*out));
    const TrainLayout L = train_layout(N, B, n, m, d);
    const FormChoice fc = choose_form(!bf16_factors, L.resident, L.ev.tshift, N > 0 ? N : 1, B, n, m, d);
    if (fc.form < 0) return fc.form;
    out->form = fc.form;
    if (fc.form == 2) {
        out->resident_q = fc.rp.Q;
        out->resident_waves = fc.rp.NW;
        out->resident_blocks = fc.rp.blocks;
        out->resident_lookahead = fc.rp.lookahead;
        out->fast_math = fc.rp.fast_math ? 1 : 0;
    } else if (fc.form == 3) {
        out->fast_math = mfcd_detail::g_resident_math != 0;
    } else {
        const void *none[] = {nullptr};
        const Plan pl = make_plan(none, 0, n, m, d);   // alignment of real pointers can only lower vec to 1
        out->streaming_vec = pl.vec;
        out->streaming_chunks = pl.chunks;
        out->streaming_blocks = pl.blocksU + pl.blocksV;
    }
    out->device_cus = device_cus();
    return 0;
}

extern "C" int mfcd_train_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                                const mfcd_sample *samples, int64_t N, int B, int64_t step0, int n, int m, int d,
                                double lr, double beta1, double beta2, double eps, double weight_decay,
                                float *loss_per_step, void *workspace, size_t workspace_bytes, void *stream)
{
    return run_train_steps<float>(U, V, mU, vU, mV, vV, samples, N, B, step0, n, m, d, lr, beta1, beta2, eps,
                                  weight_decay, loss_per_step, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int mfcd_train_steps_bf16(uint16_t *U, uint16_t *V, float *mU, float *vU, float *mV, float *vV,
                                     const mfcd_sample *samples, int64_t N, int B, int64_t step0, int n, int m, int d,
                                     double lr, double beta1, double beta2, double eps, double weight_decay,
                                     float *loss_per_step, void *workspace, size_t workspace_bytes, void *stream)
{
    return run_train_steps<mfcd_bf16>((mfcd_bf16 *)U, (mfcd_bf16 *)V, mU, vU, mV, vV, samples, N, B, step0, n, m, d, lr,
                                      beta1, beta2, eps, weight_decay, loss_per_step, workspace, workspace_bytes, stream,
                                      nullptr);
}

extern "C" int mfcd_train_steps_timed(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                                      const mfcd_sample *samples, int64_t N, int B, int64_t step0, int n, int m,
                                      int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                                      float *loss_per_step, void *workspace, size_t workspace_bytes, void *stream,
                                      float *kernel_us_host)
{
    if (!kernel_us_host) return MFCD_EINVAL;
    return run_train_steps<float>(U, V, mU, vU, mV, vV, samples, N, B, step0, n, m, d, lr, beta1, beta2, eps,
                                  weight_decay, loss_per_step, workspace, workspace_bytes, stream, kernel_us_host);
}

namespace {
struct TrainCall {
    void *U, *V;
    float *mU, *vU, *mV, *vV;
    int bf16, B, n, m, d;
    double lr, beta1, beta2, eps, wd;
    void *workspace;
    size_t workspace_bytes;
};
}  // namespace

extern "C" int mfcd_train_call_prepare(void *U, void *V, float *mU, float *vU, float *mV, float *vV, int bf16_factors,
                                       int B, int n, int m, int d, double lr, double beta1, double beta2, double eps,
                                       double weight_decay, void *workspace, size_t workspace_bytes, void **handle_out)
{
    if (!handle_out) return MFCD_EINVAL;
    *handle_out = nullptr;
    if (int rc = check_common(U, V, n, m, d)) return rc;
    if (!mU || !vU || !mV || !vV || B <= 0 || !workspace) return MFCD_EINVAL;
    WsState *S = find_ws(workspace);
    if (!S) return MFCD_ESTATE;
    if (n != S->n || m != S->m || d != S->d) return MFCD_ESTATE;
    if (workspace_bytes < S->L.total) return MFCD_EWORKSPACE;
    *handle_out = new TrainCall{U, V, mU, vU, mV, vV, bf16_factors ? 1 : 0, B, n, m, d, lr, beta1, beta2, eps,
                                weight_decay, workspace, workspace_bytes};
    return 0;
}

extern "C" int mfcd_train_call_run(void *handle, const mfcd_sample *samples, int64_t N, int64_t step0,
                                   float *loss_per_step, void *stream)
{
    const TrainCall *c = (const TrainCall *)handle;
    if (!c) return MFCD_EINVAL;
    if (c->bf16)
        return run_train_steps<mfcd_bf16>((mfcd_bf16 *)c->U, (mfcd_bf16 *)c->V, c->mU, c->vU, c->mV, c->vV, samples, N,
                                          c->B, step0, c->n, c->m, c->d, c->lr, c->beta1, c->beta2, c->eps, c->wd,
                                          loss_per_step, c->workspace, c->workspace_bytes, stream, nullptr);
    return run_train_steps<float>((float *)c->U, (float *)c->V, c->mU, c->vU, c->mV, c->vV, samples, N, c->B, step0,
                                  c->n, c->m, c->d, c->lr, c->beta1, c->beta2, c->eps, c->wd, loss_per_step,
                                  c->workspace, c->workspace_bytes, stream, nullptr);
}

extern "C" int mfcd_train_call_stage(void *handle, const mfcd_sample *samples, int64_t N, int64_t step0,
                                     float *loss_per_step, void *side_stream)
{
    const TrainCall *c = (const TrainCall *)handle;
    if (!c) return MFCD_EINVAL;
    if (c->bf16)
        return run_train_steps<mfcd_bf16>((mfcd_bf16 *)c->U, (mfcd_bf16 *)c->V, c->mU, c->vU, c->mV, c->vV, samples, N,
                                          c->B, step0, c->n, c->m, c->d, c->lr, c->beta1, c->beta2, c->eps, c->wd,
                                          loss_per_step, c->workspace, c->workspace_bytes, side_stream, nullptr, true);
    return run_train_steps<float>((float *)c->U, (float *)c->V, c->mU, c->vU, c->mV, c->vV, samples, N, c->B, step0,
                                  c->n, c->m, c->d, c->lr, c->beta1, c->beta2, c->eps, c->wd, loss_per_step,
                                  c->workspace, c->workspace_bytes, side_stream, nullptr, true);
}

extern "C" int mfcd_train_call_release(void *handle)
{
    delete (TrainCall *)handle;
    return 0;
}

extern "C" int mfcd_batch_coefficients(const float *U, const float *V, const mfcd_sample *samples, int B, int n,
                                       int m, int d, int batch_divisor, float *g_out, float *term_out,
                                       float *p_out, void *stream)
{
    if (int rc = check_common(U, V, n, m, d)) return rc;
    if (B < 0 || batch_divisor <= 0) return MFCD_EINVAL;
    if (B == 0) return 0;
    if (!samples) return MFCD_EINVAL;
    hipLaunchKernelGGL(coeff_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, U, V, samples, B, d,
                       1.0f / (float)batch_divisor, g_out, term_out, p_out);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mfcd_apply_step(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                               const mfcd_sample *samples, const float *g, int B, int64_t step, int n, int m, int d,
                               double lr, double beta1, double beta2, double eps, double weight_decay,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_common(U, V, n, m, d)) return rc;
    if (!mU || !vU || !mV || !vV || B < 0 || step < 1 || !workspace) return MFCD_EINVAL;
    if (B > 0 && (!samples || !g)) return MFCD_EINVAL;
    if (workspace_bytes < streaming_bytes(B, n, m, d)) return MFCD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace + kStatusBytes;
    float *Ualt = (float *)ws;
    ws += align256(sizeof(float) * (size_t)n * d);
    float *Valt = (float *)ws;
    const void *ptrs[] = {U, V, mU, vU, mV, vV, Ualt, Valt};
    const Plan pl = make_plan(ptrs, 8, n, m, d);
    const AdamConst ac = adam_const(lr, beta1, beta2, eps, weight_decay, step);
    dispatch_step(pl, st, U, V, Ualt, Valt, mU, vU, mV, vV, samples, g, B, 0.0f, n, m, d, ac, nullptr);
    MFCD_HIP_TRY(hipGetLastError());
    MFCD_HIP_TRY(hipMemcpyAsync(U, Ualt, sizeof(float) * (size_t)n * d, hipMemcpyDeviceToDevice, st));
    MFCD_HIP_TRY(hipMemcpyAsync(V, Valt, sizeof(float) * (size_t)m * d, hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int mfcd_dense_grad(const float *U, const float *V, const mfcd_sample *samples, int B, int n, int m, int d,
                               int batch_divisor, float *gradU, float *gradV, float *term_out, void *stream)
{
    if (int rc = check_common(U, V, n, m, d)) return rc;
    if (!gradU || !gradV || B < 0 || batch_divisor <= 0) return MFCD_EINVAL;
    if (B > 0 && !samples) return MFCD_EINVAL;
    const void *ptrs[] = {U, V, gradU, gradV};
    const Plan pl = make_plan(ptrs, 4, n, m, d);
    AdamConst ac{};
    dispatch_step<1, float>(pl, (hipStream_t)stream, U, V, (float *)nullptr, (float *)nullptr, nullptr, nullptr, nullptr,
                            nullptr, samples, nullptr, B, 1.0f / (float)batch_divisor, n, m, d, ac, term_out, gradU,
                            gradV);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mfcd_dense_grad_from_coefficients(const float *U, const float *V, const mfcd_sample *samples,
                                                 const float *g, int B, int n, int m, int d, float *gradU,
                                                 float *gradV, void *stream)
{
    if (int rc = check_common(U, V, n, m, d)) return rc;
    if (!gradU || !gradV || B < 0) return MFCD_EINVAL;
    if (B > 0 && (!samples || !g)) return MFCD_EINVAL;
    const void *ptrs[] = {U, V, gradU, gradV};
    const Plan pl = make_plan(ptrs, 4, n, m, d);
    AdamConst ac{};
    dispatch_step<1, float>(pl, (hipStream_t)stream, U, V, (float *)nullptr, (float *)nullptr, nullptr, nullptr, nullptr,
                            nullptr, samples, g, B, 0.0f, n, m, d, ac, nullptr, gradU, gradV);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mfcd_adam_dense(float *U, float *V, float *mU, float *vU, float *mV, float *vV, const float *gradU,
                               const float *gradV, int64_t step, int n, int m, int d, double lr, double beta1,
                               double beta2, double eps, double weight_decay, void *stream)
{
    if (int rc = check_common(U, V, n, m, d)) return rc;
    if (!mU || !vU || !mV || !vV || !gradU || !gradV || step < 1) return MFCD_EINVAL;
    const void *ptrs[] = {U, V, mU, vU, mV, vV, gradU, gradV};
    const Plan pl = make_plan(ptrs, 8, n, m, d);
    const AdamConst ac = adam_const(lr, beta1, beta2, eps, weight_decay, step);
    // element-wise: reading and writing the same element in place is safe (no gather in this mode)
    dispatch_step<2>(pl, (hipStream_t)stream, U, V, U, V, mU, vU, mV, vV, nullptr, nullptr, 0, 0.0f, n, m, d, ac,
                     nullptr, const_cast<float *>(gradU), const_cast<float *>(gradV));
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Data-parallel training loop, native: per optimiser step one coefficient kernel over this rank's shard, ONE RCCL
// all-gather (in place) of B {g, term} pairs per rank, one fused step kernel over the whole global batch — all
// enqueued from here on one stream with no host synchronisation (replaces the per-step Python loop of mfcd/dist.py
// for the "allgather" exchange; same protocol, same results).
extern "C" int mfcd_dp_unique_id(void *id_out, size_t id_bytes)
{
    if (!id_out || id_bytes < sizeof(ncclUniqueId)) return MFCD_EINVAL;
    const mfcd_detail::RcclApi &R = mfcd_detail::rccl();
    if (!R.ok) return MFCD_ERCCL;
    ncclUniqueId id;
    if (R.GetUniqueId(&id) != ncclSuccess) return MFCD_ERCCL;
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int mfcd_dp_comm_create(const void *id, size_t id_bytes, int rank, int world, void **comm_out)
{
    if (!id || id_bytes < sizeof(ncclUniqueId) || !comm_out || world < 1 || rank < 0 || rank >= world) return MFCD_EINVAL;
    const mfcd_detail::RcclApi &R = mfcd_detail::rccl();
    if (!R.ok) return MFCD_ERCCL;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    if (R.CommInitRank(&comm, world, uid, rank) != ncclSuccess) return MFCD_ERCCL;
    *comm_out = (void *)comm;
    return 0;
}

extern "C" int mfcd_dp_comm_destroy(void *comm)
{
    if (!comm) return 0;
    const mfcd_detail::RcclApi &R = mfcd_detail::rccl();
    if (!R.ok) return MFCD_ERCCL;
    return R.CommDestroy((ncclComm_t)comm) == ncclSuccess ? 0 : MFCD_ERCCL;
}

extern "C" size_t mfcd_dp_workspace_bytes(int64_t N, int B, int world, int n, int m, int d)
{
    if (N < 0 || B <= 0 || world < 1 || n <= 0 || m <= 0 || d <= 0) return 0;
    const int64_t Bg = (int64_t)B * world, nsteps = (N + Bg - 1) / Bg;
    return kStatusBytes + align256(sizeof(float) * (size_t)n * d) + align256(sizeof(float) * (size_t)m * d) +
           align256(sizeof(float2) * (size_t)(nsteps > 0 ? nsteps : 1) * (size_t)Bg);
}

namespace {
// fp32 or bf16 factor tables (BASELINE configs[2]); the moments, the coefficients on the wire and the arithmetic are fp32
template <typename TP>
int run_dp_train_steps(TP *U, TP *V, float *mU, float *vU, float *mV, float *vV, const mfcd_sample *samples, int64_t N,
                       int B, int rank, int world, int64_t step0, int n, int m, int d, double lr, double beta1,
                       double beta2, double eps, double weight_decay, float *loss_per_step, void *workspace,
                       size_t workspace_bytes, void *comm, void *stream)
{
    if (int rc = check_common(U, V, n, m, d)) return rc;
    if (!mU || !vU || !mV || !vV || N < 0 || B <= 0 || world < 1 || rank < 0 || rank >= world || step0 < 0)
        return MFCD_EINVAL;
    if (N == 0) return 0;
    if (!samples || !workspace) return MFCD_EINVAL;
    if (workspace_bytes < mfcd_dp_workspace_bytes(N, B, world, n, m, d)) return MFCD_EWORKSPACE;
    const mfcd_detail::RcclApi *R = nullptr;
    if (comm) {
        R = &mfcd_detail::rccl();
        if (!R->ok) return MFCD_ERCCL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t Bg = (int64_t)B * world, nsteps = (N + Bg - 1) / Bg;
    char *ws = (char *)workspace + kStatusBytes;
    TP *Ualt = (TP *)ws;
    ws += align256(sizeof(float) * (size_t)n * d);
    TP *Valt = (TP *)ws;
    ws += align256(sizeof(float) * (size_t)m * d);
    float2 *xbuf = (float2 *)ws;   // [nsteps][world][B] pairs = global sample order inside a step
    const void *ptrs[] = {U, V, mU, vU, mV, vV, Ualt, Valt};
    const Plan pl = make_plan(ptrs, 8, n, m, d);
    for (int64_t k = 0; k < nsteps; ++k) {
        const int64_t lo = k * Bg, hi = (lo + Bg < N) ? lo + Bg : N;
        const int nglob = (int)(hi - lo);          // divisor of the mean (structure.py:849): the GLOBAL batch
        const bool even = (k & 1) == 0;
        const TP *Uc = even ? U : Ualt, *Vc = even ? V : Valt;
        float2 *xk = xbuf + (size_t)k * Bg;
        // ranks this process computes: its own; or, without a communicator, every rank in turn (replicas are
        // bit-identical, so this reproduces the gathered buffer exactly: single-process rehearsal of any world size)
        const int r0 = comm ? rank : 0, r1 = comm ? rank + 1 : world;
        for (int r = r0; r < r1; ++r) {
            const int64_t mylo = (lo + (int64_t)r * B < hi) ? lo + (int64_t)r * B : hi;
            const int64_t myhi = (mylo + B < hi) ? mylo + B : hi;
            hipLaunchKernelGGL(dp_coeff_kernel<TP>, dim3((B + 3) / 4), dim3(256), 0, st, Uc, Vc, samples + mylo,
                               (int)(myhi - mylo), B, d, 1.0f / (float)nglob, xk + (size_t)r * B);
        }
        if (comm) {   // also on a one-rank communicator: the call path is the same at every world size
            if (R->AllGather(xk + (size_t)rank * B, xk, (size_t)B * 2, ncclFloat, (ncclComm_t)comm, st) != ncclSuccess)
                return MFCD_ERCCL;
        }
        const AdamConst ac = adam_const(lr, beta1, beta2, eps, weight_decay, step0 + k + 1);
        dispatch_step<0, TP>(pl, st, Uc, Vc, even ? Ualt : U, even ? Valt : V, mU, vU, mV, vV, samples + lo,
                                (const float *)xk, nglob, 0.0f, n, m, d, ac, nullptr, nullptr, nullptr, 2);
    }
    MFCD_HIP_TRY(hipGetLastError());
    if (nsteps & 1) {
        MFCD_HIP_TRY(hipMemcpyAsync(U, Ualt, sizeof(TP) * (size_t)n * d, hipMemcpyDeviceToDevice, st));
        MFCD_HIP_TRY(hipMemcpyAsync(V, Valt, sizeof(TP) * (size_t)m * d, hipMemcpyDeviceToDevice, st));
    }
    if (loss_per_step) {
        hipLaunchKernelGGL(dp_loss_kernel, dim3((unsigned)nsteps), dim3(64), 0, st, xbuf, N, (int)Bg, loss_per_step);
        MFCD_HIP_TRY(hipGetLastError());
    }
    return 0;
}
}  // namespace

extern "C" int mfcd_dp_train_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                                   const mfcd_sample *samples, int64_t N, int B, int rank, int world, int64_t step0,
                                   int n, int m, int d, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, float *loss_per_step, void *workspace, size_t workspace_bytes,
                                   void *comm, void *stream)
{
    return run_dp_train_steps<float>(U, V, mU, vU, mV, vV, samples, N, B, rank, world, step0, n, m, d, lr, beta1, beta2,
                                     eps, weight_decay, loss_per_step, workspace, workspace_bytes, comm, stream);
}

extern "C" int mfcd_dp_train_steps_bf16(uint16_t *U, uint16_t *V, float *mU, float *vU, float *mV, float *vV,
                                        const mfcd_sample *samples, int64_t N, int B, int rank, int world, int64_t step0,
                                        int n, int m, int d, double lr, double beta1, double beta2, double eps,
                                        double weight_decay, float *loss_per_step, void *workspace,
                                        size_t workspace_bytes, void *comm, void *stream)
{
    return run_dp_train_steps<mfcd_bf16>((mfcd_bf16 *)U, (mfcd_bf16 *)V, mU, vU, mV, vV, samples, N, B, rank, world, step0,
                                         n, m, d, lr, beta1, beta2, eps, weight_decay, loss_per_step, workspace,
                                         workspace_bytes, comm, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Row-sharded training (strong scaling): rank r holds rows [lo_r, hi_r) of U, V and of their Adam
// moments; the batch stays the reference's B (structure.py:668) and every rank sees the same records, so the run equals
// the single-GPU run with the same B.  Per optimiser step: (1) every rank writes the rows of the batch it owns into an
// exchange buffer xbuf[role][B][d] (zeros elsewhere), (2) ONE all-reduce(sum) of the buffer taken as 32-bit integers —
// exactly one rank contributes non-zero bits per row, so the sum reproduces the rows bit for bit — (3) the fused step
// (MODE 3 of train_step_kernel) over the shard, in place, with the dense Adam sweep over 1/R of the state.
namespace {

template <typename TP>
__global__ __launch_bounds__(256) void shard_pack_kernel(const TP *__restrict__ Us, const TP *__restrict__ Vs,
                                                         const mfcd_sample *__restrict__ batch, int Bk, int Bcap, int d,
                                                         int u_off, int nu, int v_off, int nv, float *__restrict__ xbuf,
                                                         int merge)
{
    // merge = 0: every row of the buffer is written (zeros where this rank owns nothing): the input of the all-reduce;
    // merge = 1: only owned rows are written on top of what is there (single-process rehearsal of the all-reduce)
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);   // (role, t)
    if (w >= 3 * Bcap) return;
    const int role = w / Bcap, t = w - role * Bcap;
    const TP *src = nullptr;
    if (t < Bk) {
        const mfcd_sample s = batch[t];
        const int row = role == 0 ? s.u - u_off : (role == 1 ? s.i : s.j) - v_off;
        if (row >= 0 && row < (role == 0 ? nu : nv)) src = (role == 0 ? Us : Vs) + (int64_t)row * d;
    }
    if (merge && !src) return;
    float *dst = xbuf + (int64_t)w * d;
    for (int k = lane; k < d; k += MFCD_WAVE) dst[k] = src ? ldf(src, k) : 0.0f;   // bf16 rows widen exactly
}

// Exchange rows of the NEXT batch, rolled forward over the step that is about to run (the look-ahead rule of the
// resident form): a row that the current batch does not touch changes in that step by the dense update with
// a zero sparse gradient alone, a pure function of its (p, m, v) — so its value AFTER the step can be put on the wire
// BEFORE the step runs, and the collective of batch k+1 overlaps the step of batch k.  Same adam_update, same operation
// order as the step kernel: the rolled value equals the in-place one bit for bit (rehearsal tests).  The caller
// guarantees that no row of `next` is named by the current batch (mfcd_shard_collisions).
template <typename TP>
__global__ __launch_bounds__(256) void shard_pack_ahead_kernel(const TP *__restrict__ Us, const TP *__restrict__ Vs,
                                                               const float *__restrict__ mU, const float *__restrict__ vU,
                                                               const float *__restrict__ mV, const float *__restrict__ vV,
                                                               const mfcd_sample *__restrict__ next, int Bk, int Bcap,
                                                               int d, int u_off, int nu, int v_off, int nv, AdamConst ac,
                                                               float *__restrict__ xbuf, int merge)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);   // (role, t)
    if (w >= 3 * Bcap) return;
    const int role = w / Bcap, t = w - role * Bcap;
    int64_t src = -1;
    if (t < Bk) {
        const mfcd_sample s = next[t];
        const int row = role == 0 ? s.u - u_off : (role == 1 ? s.i : s.j) - v_off;
        if (row >= 0 && row < (role == 0 ? nu : nv)) src = (int64_t)row * d;
    }
    if (merge && src < 0) return;
    float *dst = xbuf + (int64_t)w * d;
    const TP *P = role == 0 ? Us : Vs;
    const float *M1 = role == 0 ? mU : mV, *M2 = role == 0 ? vU : vV;
    for (int k = lane; k < d; k += MFCD_WAVE) {
        float v = 0.0f;
        if (src >= 0) {
            float p = ldf(P, src + k), m1 = M1[src + k], m2 = M2[src + k];
            adam_update(p, m1, m2, 0.0f, ac.st, ac.sc);
            if constexpr (sizeof(TP) == 2) p = (float)(mfcd_bf16)p;   // the step's one rounding point (store_vec)
            v = p;
        }
        dst[k] = v;
    }
}

// flags[k] = 1 when batch k+1 names a row that batch k names too (same table), 0 otherwise and for the last batch.
__global__ __launch_bounds__(256) void shard_collisions_kernel(const mfcd_sample *__restrict__ samples, int64_t N, int B,
                                                               int64_t nsteps, uint8_t *__restrict__ flags)
{
    const int64_t k = blockIdx.x;
    __shared__ int hit;
    if (threadIdx.x == 0) hit = 0;
    __syncthreads();
    if (k + 1 < nsteps) {
        const int64_t o0 = k * B, o1 = (k + 1) * B;
        const int b0 = (int)((N - o0) < B ? (N - o0) : B), b1 = (int)((N - o1) < B ? (N - o1) : B);
        int mine = 0;
        for (int64_t pr = threadIdx.x; pr < (int64_t)b0 * b1 && !mine; pr += 256) {
            const mfcd_sample a = samples[o0 + pr / b1], b = samples[o1 + pr % b1];
            mine = (a.u == b.u) | (a.i == b.i) | (a.i == b.j) | (a.j == b.i) | (a.j == b.j);
        }
        if (mine) hit = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) flags[k] = (uint8_t)hit;
}

void shard_range(int rows, int rank, int world, int *lo, int *hi)
{
    *lo = (int)((int64_t)rows * rank / world);
    *hi = (int)((int64_t)rows * (rank + 1) / world);
}

size_t shard_xbuf_bytes(int B, int d) { return align256(sizeof(float) * 3 * (size_t)B * d); }

// BCE terms of a batch from the exchange buffer alone: what workgroup 0 of the MODE 3 step records, for a rank whose
// shard is EMPTY (more ranks than rows) and therefore launches no step kernel.  Same dot-product order.
__global__ __launch_bounds__(256) void shard_terms_kernel(const float *__restrict__ xb, const mfcd_sample *__restrict__ batch,
                                                          int Bk, int Bcap, int d, float *__restrict__ loss_terms)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= Bk) return;
    const float *ur = xb + (int64_t)t * d, *vi = xb + ((int64_t)Bcap + t) * d, *vj = xb + ((int64_t)2 * Bcap + t) * d;
    float acc = 0.0f;
    for (int k = lane; k < d; k += MFCD_WAVE) acc += ur[k] * (vi[k] - vj[k]);
    const float p = sigmoid_f32(wave_sum64(acc));
    if (lane == 0) loss_terms[t] = bce_term_f32(p, batch[t].z);
}

}  // namespace

extern "C" int mfcd_shard_rows(int rows, int rank, int world, int *lo, int *hi)
{
    if (rows <= 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return MFCD_EINVAL;
    shard_range(rows, rank, world, lo, hi);
    return 0;
}

extern "C" size_t mfcd_shard_workspace_bytes(int64_t N, int B, int d)
{
    if (N < 0 || B <= 0 || d <= 0) return 0;
    const size_t nsteps = (size_t)((N + B - 1) / B);
    return 2 * shard_xbuf_bytes(B, d) + align256(sizeof(float) * (size_t)(N > 0 ? N : 1)) + align256(nsteps + 1);
}

extern "C" int mfcd_shard_collisions(const mfcd_sample *samples, int64_t N, int B, uint8_t *flags_dev, void *stream)
{
    if (N < 0 || B <= 0) return MFCD_EINVAL;
    if (N == 0) return 0;
    if (!samples || !flags_dev) return MFCD_EINVAL;
    const int64_t nsteps = (N + B - 1) / B;
    hipLaunchKernelGGL(shard_collisions_kernel, dim3((unsigned)nsteps), dim3(256), 0, (hipStream_t)stream, samples, N, B,
                       nsteps, flags_dev);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mfcd_shard_pack_ahead(const float *U_shard, const float *V_shard, const float *mU, const float *vU,
                                     const float *mV, const float *vV, const mfcd_sample *next_batch, int Bk, int B,
                                     int64_t step, int d, int u_lo, int u_hi, int v_lo, int v_hi, double lr, double beta1,
                                     double beta2, double eps, double weight_decay, float *xbuf, void *stream)
{
    if (!next_batch || !xbuf || Bk < 0 || B <= 0 || Bk > B || d <= 0 || d > MFCD_MAX_D || u_hi < u_lo || v_hi < v_lo ||
        step < 1)
        return MFCD_EINVAL;
    if ((u_hi > u_lo && (!U_shard || !mU || !vU)) || (v_hi > v_lo && (!V_shard || !mV || !vV))) return MFCD_EINVAL;
    const AdamConst ac = adam_const(lr, beta1, beta2, eps, weight_decay, step);
    hipLaunchKernelGGL(shard_pack_ahead_kernel<float>, dim3((3 * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, U_shard, V_shard,
                       mU, vU, mV, vV, next_batch, Bk, B, d, u_lo, u_hi - u_lo, v_lo, v_hi - v_lo, ac, xbuf, 0);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mfcd_shard_pack(const float *U_shard, const float *V_shard, const mfcd_sample *batch, int Bk, int B,
                               int d, int u_lo, int u_hi, int v_lo, int v_hi, float *xbuf, void *stream)
{
    if (!batch || !xbuf || Bk < 0 || B <= 0 || Bk > B || d <= 0 || d > MFCD_MAX_D || u_hi < u_lo || v_hi < v_lo)
        return MFCD_EINVAL;
    if ((u_hi > u_lo && !U_shard) || (v_hi > v_lo && !V_shard)) return MFCD_EINVAL;
    hipLaunchKernelGGL(shard_pack_kernel<float>, dim3((3 * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, U_shard, V_shard,
                       batch, Bk, B, d, u_lo, u_hi - u_lo, v_lo, v_hi - v_lo, xbuf, 0);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

namespace {
template <typename TP>
int shard_apply_t(TP *U_shard, TP *V_shard, float *mU, float *vU, float *mV, float *vV, const mfcd_sample *batch, int Bk,
                  int B, const float *xbuf, int64_t step, int d, int u_lo, int u_hi, int v_lo, int v_hi, double lr,
                  double beta1, double beta2, double eps, double weight_decay, float *loss_terms, void *stream)
{
    if (!batch || !xbuf || Bk <= 0 || B <= 0 || Bk > B || d <= 0 || d > MFCD_MAX_D || step < 1) return MFCD_EINVAL;
    const int nu = u_hi - u_lo, nv = v_hi - v_lo;
    if (nu < 0 || nv < 0) return MFCD_EINVAL;
    if (nu + nv == 0) {   // this rank owns no row (world > rows): nothing to update, only the step's loss terms
        if (loss_terms) {
            hipLaunchKernelGGL(shard_terms_kernel, dim3((Bk + 3) / 4), dim3(256), 0, (hipStream_t)stream, xbuf, batch, Bk,
                               B, d, loss_terms);
            MFCD_HIP_TRY(hipGetLastError());
        }
        return 0;
    }
    if ((nu > 0 && (!U_shard || !mU || !vU)) || (nv > 0 && (!V_shard || !mV || !vV))) return MFCD_EINVAL;
    const void *ptrs[] = {U_shard, V_shard, mU, vU, mV, vV, xbuf};
    // an empty table side is legal (a rank may own rows of one table only when world > rows): one dummy row count
    const Plan pl = make_plan(ptrs, 7, nu > 0 ? nu : 0, nv > 0 ? nv : 0, d);
    const AdamConst ac = adam_const(lr, beta1, beta2, eps, weight_decay, step);
    dispatch_step<3, TP>(pl, (hipStream_t)stream, U_shard, V_shard, U_shard, V_shard, mU, vU, mV, vV, batch, xbuf, Bk,
                         1.0f / (float)Bk, nu, nv, d, ac, loss_terms, nullptr, nullptr, B, u_lo, v_lo);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int mfcd_shard_apply(float *U_shard, float *V_shard, float *mU, float *vU, float *mV, float *vV,
                                const mfcd_sample *batch, int Bk, int B, const float *xbuf, int64_t step, int d,
                                int u_lo, int u_hi, int v_lo, int v_hi, double lr, double beta1, double beta2,
                                double eps, double weight_decay, float *loss_terms, void *stream)
{
    return shard_apply_t<float>(U_shard, V_shard, mU, vU, mV, vV, batch, Bk, B, xbuf, step, d, u_lo, u_hi, v_lo, v_hi, lr,
                                beta1, beta2, eps, weight_decay, loss_terms, stream);
}

namespace {
// fp32 or bf16 factor shards (BASELINE configs[2]'s storage); the exchange buffer, the moments and the arithmetic are fp32
template <typename TP>
int run_shard_train_steps(TP *U, TP *V, float *mU, float *vU, float *mV, float *vV, const mfcd_sample *samples, int64_t N,
                          int B, int rank, int world, int64_t step0, int n, int m, int d, double lr, double beta1,
                          double beta2, double eps, double weight_decay, float *loss_per_step, void *workspace,
                          size_t workspace_bytes, void *comm, void *stream)
{
    if (n <= 0 || m <= 0 || d <= 0 || d > MFCD_MAX_D || N < 0 || B <= 0 || world < 1 || rank < 0 || rank >= world ||
        step0 < 0)
        return MFCD_EINVAL;
    if (N == 0) return 0;
    if (!samples || !workspace || !U || !V || !mU || !vU || !mV || !vV) return MFCD_EINVAL;
    if (workspace_bytes < mfcd_shard_workspace_bytes(N, B, d)) return MFCD_EWORKSPACE;
    const mfcd_detail::RcclApi *R = nullptr;
    if (comm) {
        R = &mfcd_detail::rccl();
        if (!R->ok || !R->AllReduce) return MFCD_ERCCL;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t xb = shard_xbuf_bytes(B, d);
    float *xbufs[2] = {(float *)workspace, (float *)((char *)workspace + xb)};
    float *terms = (float *)((char *)workspace + 2 * xb);
    uint8_t *flags_dev = (uint8_t *)((char *)terms + align256(sizeof(float) * (size_t)N));
    const int64_t nsteps = (N + B - 1) / B;
    // with a communicator the table pointers are this rank's SHARDS; without one they are the FULL tables and this
    // process plays every rank in turn (single-process rehearsal of any world size; exact, the protocol is the same)
    const int r0 = comm ? rank : 0, r1 = comm ? rank + 1 : world;
    auto range = [&](int r, int &ul, int &uh, int &vl, int &vh, int64_t &uo, int64_t &vo) {
        shard_range(n, r, world, &ul, &uh);
        shard_range(m, r, world, &vl, &vh);
        uo = comm ? 0 : (int64_t)ul * d;
        vo = comm ? 0 : (int64_t)vl * d;
    };
    // rows of batch k as they are NOW into xbuf (after step k-1 has run)
    auto pack_now = [&](int64_t k, float *xbuf) -> int {
        const int64_t off = k * B;
        const int Bk = (int)((N - off) < B ? (N - off) : B);
        for (int r = r0; r < r1; ++r) {
            int ul, uh, vl, vh;
            int64_t uo, vo;
            range(r, ul, uh, vl, vh, uo, vo);
            hipLaunchKernelGGL(shard_pack_kernel<TP>, dim3((3 * B + 3) / 4), dim3(256), 0, st, U + uo, V + vo, samples + off, Bk,
                               B, d, ul, uh - ul, vl, vh - vl, xbuf, (comm || r == r0) ? 0 : 1);
        }
        MFCD_HIP_TRY(hipGetLastError());
        return 0;
    };
    // rows of batch k+1 as they will be after step k, which has not run yet (no row shared with batch k)
    auto pack_ahead = [&](int64_t k, float *xbuf) -> int {
        const int64_t off = (k + 1) * B;
        const int Bk = (int)((N - off) < B ? (N - off) : B);
        const AdamConst ac = adam_const(lr, beta1, beta2, eps, weight_decay, step0 + k + 1);
        for (int r = r0; r < r1; ++r) {
            int ul, uh, vl, vh;
            int64_t uo, vo;
            range(r, ul, uh, vl, vh, uo, vo);
            hipLaunchKernelGGL(shard_pack_ahead_kernel<TP>, dim3((3 * B + 3) / 4), dim3(256), 0, st, U + uo, V + vo, mU + uo,
                               vU + uo, mV + vo, vV + vo, samples + off, Bk, B, d, ul, uh - ul, vl, vh - vl, ac, xbuf,
                               (comm || r == r0) ? 0 : 1);
        }
        MFCD_HIP_TRY(hipGetLastError());
        return 0;
    };
    auto apply = [&](int64_t k, const float *xbuf) -> int {
        const int64_t off = k * B;
        const int Bk = (int)((N - off) < B ? (N - off) : B);
        for (int r = r0; r < r1; ++r) {
            int ul, uh, vl, vh;
            int64_t uo, vo;
            range(r, ul, uh, vl, vh, uo, vo);
            if (int rc = shard_apply_t<TP>(U + uo, V + vo, mU + uo, vU + uo, mV + vo, vV + vo, samples + off, Bk, B, xbuf,
                                           step0 + k + 1, d, ul, uh, vl, vh, lr, beta1, beta2, eps, weight_decay,
                                           r == r0 ? terms + off : nullptr, stream))
                return rc;
        }
        return 0;
    };

    // 1 (default): pipelined where there is a collective to hide (world > 1) and in the comm-less rehearsal; a one-rank
    // communicator takes the strict chain (measured, C4 on a one-rank group: the two stream hops of a pipelined step cost
    // 10 us more than they hide when the collective is a local copy); 2 forces the pipelined chain (tests, timing)
    const int pipe = mfcd_detail::g_tune.shard_pipeline;
    if (pipe == 0 || nsteps < 2 || (pipe == 1 && comm && world == 1)) {
        // the strict chain: pack -> all-reduce -> step, one stream
        for (int64_t k = 0; k < nsteps; ++k) {
            if (int rc = pack_now(k, xbufs[0])) return rc;
            if (comm && R->AllReduce(xbufs[0], xbufs[0], (size_t)3 * B * d, ncclUint32, ncclSum, (ncclComm_t)comm, st) !=
                            ncclSuccess)
                return MFCD_ERCCL;
            if (int rc = apply(k, xbufs[0])) return rc;
        }
    } else {
        // Pipelined exchange: where batch k+1 shares no row with batch k, its rows are packed AHEAD of step k (rolled
        // forward over it) and their all-reduce runs on a side stream underneath step k; only the pairs of batches that
        // do share a row keep the strict chain.  Which pairs those are is a property of the sample stream alone (the
        // same on every rank): one kernel marks them, the host reads the marks once per call (the one host wait of this
        // entry point; the enqueue pattern — hence the collective sequence of every rank — depends on them).
        if (int rc = mfcd_shard_collisions(samples, N, B, flags_dev, stream)) return rc;
        std::vector<uint8_t> collide((size_t)nsteps);
        MFCD_HIP_TRY(hipMemcpyAsync(collide.data(), flags_dev, (size_t)nsteps, hipMemcpyDeviceToHost, st));
        MFCD_HIP_TRY(hipStreamSynchronize(st));
        hipStream_t cs = nullptr;
        hipEvent_t packed[2] = {nullptr, nullptr}, reduced[2] = {nullptr, nullptr};
        int rc = 0;
        auto fail = [&](int code) { if (!rc) rc = code; };
        if (comm) {
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return MFCD_EINVAL;
            for (int e = 0; e < 2; ++e) {
                if (hipEventCreateWithFlags(&packed[e], hipEventDisableTiming) != hipSuccess) fail(MFCD_EINVAL);
                if (hipEventCreateWithFlags(&reduced[e], hipEventDisableTiming) != hipSuccess) fail(MFCD_EINVAL);
            }
        }
        // the collective of the buffer just packed on `st`, on the side stream; `st` meets it again at wait_reduced
        auto reduce_async = [&](int b) {
            if (!comm || rc) return;
            if (hipEventRecord(packed[b], st) != hipSuccess || hipStreamWaitEvent(cs, packed[b], 0) != hipSuccess)
                return fail(MFCD_EINVAL);
            if (R->AllReduce(xbufs[b], xbufs[b], (size_t)3 * B * d, ncclUint32, ncclSum, (ncclComm_t)comm, cs) != ncclSuccess)
                return fail(MFCD_ERCCL);
            if (hipEventRecord(reduced[b], cs) != hipSuccess) fail(MFCD_EINVAL);
        };
        auto wait_reduced = [&](int b) {
            if (comm && !rc && hipStreamWaitEvent(st, reduced[b], 0) != hipSuccess) fail(MFCD_EINVAL);
        };
        // a collective on the main stream itself: no overlap wanted, so no stream hops (the first batch, and every
        // batch that shares a row with the one before it: the strict chain, at the strict chain's cost).  Never concurrent
        // with a side-stream collective on the same communicator: `st` has met every earlier one by then, and the next
        // one on the side stream waits for an event recorded on `st` after this call.
        bool on_side[2] = {false, false};
        auto reduce_inline = [&](int b) {
            on_side[b] = false;
            if (!comm || rc) return;
            if (R->AllReduce(xbufs[b], xbufs[b], (size_t)3 * B * d, ncclUint32, ncclSum, (ncclComm_t)comm, st) != ncclSuccess)
                fail(MFCD_ERCCL);
        };
        if (!rc) fail(pack_now(0, xbufs[0]));
        reduce_inline(0);
        for (int64_t k = 0; k < nsteps && !rc; ++k) {
            const int cur = (int)(k & 1), nxt = cur ^ 1;
            const bool more = k + 1 < nsteps, ahead = more && !collide[(size_t)k];
            if (ahead) {                      // batch k+1 goes on the wire before step k runs
                fail(pack_ahead(k, xbufs[nxt]));
                reduce_async(nxt);
                on_side[nxt] = true;
            }
            if (on_side[cur]) wait_reduced(cur);
            if (!rc) fail(apply(k, xbufs[cur]));
            if (more && !ahead) {             // a shared row: batch k+1 is packed from the updated state
                if (!rc) fail(pack_now(k + 1, xbufs[nxt]));
                reduce_inline(nxt);
            }
        }
        if (comm) {
            // every collective was met by `st` (wait_reduced precedes each step), so nothing is pending on the side stream
            for (int e = 0; e < 2; ++e) {
                if (packed[e]) (void)hipEventDestroy(packed[e]);
                if (reduced[e]) (void)hipEventDestroy(reduced[e]);
            }
            (void)hipStreamDestroy(cs);
        }
        if (rc) return rc;
    }
    if (loss_per_step) {
        hipLaunchKernelGGL(batch_mean_kernel, dim3((unsigned)nsteps), dim3(64), 0, st, terms,
                           (const mfcd_sample *)nullptr, N, B, loss_per_step);
        MFCD_HIP_TRY(hipGetLastError());
    }
    return 0;
}
}  // namespace

extern "C" int mfcd_shard_train_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                                      const mfcd_sample *samples, int64_t N, int B, int rank, int world, int64_t step0,
                                      int n, int m, int d, double lr, double beta1, double beta2, double eps,
                                      double weight_decay, float *loss_per_step, void *workspace, size_t workspace_bytes,
                                      void *comm, void *stream)
{
    return run_shard_train_steps<float>(U, V, mU, vU, mV, vV, samples, N, B, rank, world, step0, n, m, d, lr, beta1, beta2,
                                        eps, weight_decay, loss_per_step, workspace, workspace_bytes, comm, stream);
}

extern "C" int mfcd_shard_train_steps_bf16(uint16_t *U, uint16_t *V, float *mU, float *vU, float *mV, float *vV,
                                           const mfcd_sample *samples, int64_t N, int B, int rank, int world,
                                           int64_t step0, int n, int m, int d, double lr, double beta1, double beta2,
                                           double eps, double weight_decay, float *loss_per_step, void *workspace,
                                           size_t workspace_bytes, void *comm, void *stream)
{
    return run_shard_train_steps<mfcd_bf16>((mfcd_bf16 *)U, (mfcd_bf16 *)V, mU, vU, mV, vV, samples, N, B, rank, world,
                                            step0, n, m, d, lr, beta1, beta2, eps, weight_decay, loss_per_step, workspace,
                                            workspace_bytes, comm, stream);
}
