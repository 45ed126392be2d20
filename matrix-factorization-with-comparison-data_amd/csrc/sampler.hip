// sampler.hip — triplet sampling on the device.
//
// Replaces the rejection loops of generation_data.py:16-224 (one Python iteration, one or two generator calls and a
// set lookup per ATTEMPT; proximity / top_k additionally two O(m) torch.topk per attempt) with a block of attempts
// per launch: one thread per attempt draws (u, i, j) from the strategy's law, applies the strategy's filter, and the
// block is reduced to "the first `want` distinct, allowed triplets in attempt order" — exactly what the reference's
// loop keeps — by a stable device-wide radix sort of the 64-bit triplet keys (the caller's barred keys placed in
// front, so a barred triplet is never the first of its run), a first-of-run mark scattered back to attempt order,
// an exclusive scan and an emit pass.  Laws:
//   MFCD_LAW_UNIFORM   u ~ U[0,n), i, j ~ U[0,m) independent (generation_data.py:21-22 random, 66-67 margin)
//   MFCD_LAW_ITEM_CDF  i, j from an item law p over the catalogue (cdf given in f64):
//                      pair_rule 0 = numpy RandomState.choice(m, size=2, replace=False, p) (generation_data.py:124,
//                      popularity): a ~ p, b ~ p, and only if b == a is b redrawn from p with p[a] = 0;
//                      pair_rule 1 = sequential draw without replacement (torch.multinomial(p, 2), :95, variance)
//   MFCD_LAW_LISTS     i = list_i[row(u)][U[0,k)], j = list_j[row(u)][U[0,k)] with per-user rows (proximity :36-39:
//                      the user's k best / k worst items; top_k :208-213: the k best twice, pair_rule 1 = positions
//                      distinct, which is the law of the reference's redraw-while-equal loop) or one list shared by
//                      all users (row stride 0; svd :168-169 with `users` = the top users)
// Optional filter: |X[u][i] - X[u][j]| <= margin (generation_data.py:72-73), X dense fp32 or A B^T by its factors.
// Randomness: Philox4x32-10 keyed by the caller's seed, counter = (attempt index, draw group): a function of
// (seed, attempt index) alone.  It is NOT the reference's Mersenne-Twister stream: parity is distributional (same law
// per attempt, same keep-the-first-distinct rule); seeded bit-level replay of a reference run stays with the host
// forms in generation_data.py, which consume torch's and numpy's generators draw for draw.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

constexpr uint64_t KEY_NONE = ~0ull;

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1)
{
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ uint64_t pair64(unsigned lo, unsigned hi) { return ((uint64_t)hi << 32) | lo; }

// uniform integer in [0, range) from 64 random bits (multiply-high; bias <= range / 2^64)
__device__ __forceinline__ int64_t below(uint64_t bits, int64_t range) { return (int64_t)__umul64hi(bits, (uint64_t)range); }

// uniform double in [0, 1) with 53 bits, as numpy's random_sample forms it
__device__ __forceinline__ double unit53(uint64_t bits) { return (double)(bits >> 11) * (1.0 / 9007199254740992.0); }

// searchsorted(cdf, x, side='right'), clamped to the catalogue
__device__ __forceinline__ int cdf_pick(const double *__restrict__ cdf, int m, double x)
{
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo < m ? lo : m - 1;
}

__global__ __launch_bounds__(256) void sample_attempts_kernel(mfcd_sampler law, int64_t attempt0, int64_t A,
                                                              unsigned seed_lo, unsigned seed_hi,
                                                              uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                              int64_t E)
{
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const uint64_t t = (uint64_t)(attempt0 + a);
    unsigned c0[4] = {(unsigned)t, (unsigned)(t >> 32), 0u, 0x73616d70u};
    unsigned c1[4] = {(unsigned)t, (unsigned)(t >> 32), 1u, 0x73616d70u};
    philox4x32_10(c0, seed_lo, seed_hi);
    philox4x32_10(c1, seed_lo, seed_hi);
    const uint64_t bu = pair64(c0[0], c0[1]), b0 = pair64(c0[2], c0[3]), b1 = pair64(c1[0], c1[1]);

    int u = law.users ? law.users[below(bu, law.n_users)] : (int)below(bu, law.n);
    int i, j;
    bool ok = true;
    if (law.law == MFCD_LAW_UNIFORM) {
        i = (int)below(b0, law.m);
        j = (int)below(b1, law.m);
    } else if (law.law == MFCD_LAW_ITEM_CDF) {
        const double *cdf = law.cdf;
        i = cdf_pick(cdf, law.m, unit53(b0));
        j = law.pair_rule == 0 ? cdf_pick(cdf, law.m, unit53(b1)) : i;
        if (j == i) {
            // b from p with p[i] = 0, renormalised: the cdf with i's mass cut out, inverted without a loop
            const double start = i ? cdf[i - 1] : 0.0, mass = cdf[i] - start;
            const uint64_t b2 = law.pair_rule == 0 ? pair64(c1[2], c1[3]) : b1;
            double x = unit53(b2) * (1.0 - mass);
            if (x >= start) x += mass;
            j = cdf_pick(cdf, law.m, x);
            if (j == i) j = i + 1 < law.m ? i + 1 : i - 1;       // x landed on i's upper edge after rounding
            ok = (1.0 - mass) > 0.0 && j >= 0;
        }
    } else {
        const int k = law.k;
        const int64_t row = (int64_t)u * law.list_row_stride;
        const int pi = (int)below(b0, k);
        int pj;
        if (law.pair_rule == 0) {
            pj = (int)below(b1, k);
        } else {
            pj = (int)below(b1, k - 1);
            pj += pj >= pi;
        }
        i = law.list_i[row + pi];
        j = law.list_j[row + pj];
    }
    ok = ok && i != j;
    if (ok && law.use_margin) {
        float diff;
        if (law.X) {
            diff = law.X[(int64_t)u * law.m + i] - law.X[(int64_t)u * law.m + j];       // generation_data.py:72
        } else {
            const float *pa = law.A + (int64_t)u * law.dx, *pi = law.B + (int64_t)i * law.dx,
                        *pj = law.B + (int64_t)j * law.dx;
            float xi = 0.0f, xj = 0.0f;
            for (int q = 0; q < law.dx; ++q) {
                xi = fmaf(pa[q], pi[q], xi);
                xj = fmaf(pa[q], pj[q], xj);
            }
            diff = xi - xj;
        }
        ok = (double)fabsf(diff) <= law.margin;
    }
    keys[E + a] = ok ? ((uint64_t)u * (uint64_t)law.m + (uint64_t)i) * (uint64_t)law.m + (uint64_t)j : KEY_NONE;
    vals[E + a] = (uint32_t)(E + a);
}

__global__ __launch_bounds__(256) void place_barred_kernel(const int64_t *__restrict__ barred, int64_t E,
                                                           uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= E) return;
    keys[p] = (uint64_t)barred[p];
    vals[p] = (uint32_t)p;
}

__global__ __launch_bounds__(256) void mark_first_kernel(const uint64_t *__restrict__ skeys,
                                                         const uint32_t *__restrict__ svals, int64_t total, int64_t E,
                                                         int32_t *__restrict__ flags)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const uint64_t key = skeys[p];
    const uint32_t v = svals[p];
    if (v < E || key == KEY_NONE) return;
    if (p == 0 || skeys[p - 1] != key) flags[v - E] = 1;     // stable sort: an equal barred key or earlier attempt sits in front
}

__global__ __launch_bounds__(256) void emit_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ flags,
                                                   const int32_t *__restrict__ pos, int64_t A, int64_t E, int64_t want,
                                                   int m, int32_t *__restrict__ trip, int64_t *__restrict__ keys_out,
                                                   int64_t *__restrict__ counts)
{
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const int f = flags[a];
    const int64_t p = pos[a];
    if (f && p < want) {
        const uint64_t key = keys[E + a];
        const uint64_t ui = key / (uint64_t)m;
        trip[3 * p] = (int32_t)(ui / (uint64_t)m);
        trip[3 * p + 1] = (int32_t)(ui % (uint64_t)m);
        trip[3 * p + 2] = (int32_t)(key % (uint64_t)m);
        keys_out[p] = (int64_t)key;
        if (p == want - 1) counts[1] = a + 1;               // the attempt that completed the request
    }
    if (a == A - 1) {
        const int64_t got = p + f;
        counts[0] = got < want ? got : want;
        if (got < want) counts[1] = A;
    }
}

struct Carve {
    uint64_t *keys_in, *keys_out;
    uint32_t *vals_in, *vals_out;
    int32_t *flags, *pos;
    void *temp;
    size_t temp_bytes, total_bytes;
};

size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

int carve(int64_t A, int64_t E, void *base, Carve &c)
{
    const size_t T = (size_t)(A + E);
    size_t sort_bytes = 0, scan_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                  (uint32_t *)nullptr, T, 0, 64, (hipStream_t)0) != hipSuccess) return MFCD_EINVAL;
    if (rocprim::exclusive_scan(nullptr, scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, 0, (size_t)A,
                                rocprim::plus<int32_t>(), (hipStream_t)0) != hipSuccess) return MFCD_EINVAL;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += up256(bytes); return (void *)r; };
    c.keys_in = (uint64_t *)take(T * 8);
    c.keys_out = (uint64_t *)take(T * 8);
    c.vals_in = (uint32_t *)take(T * 4);
    c.vals_out = (uint32_t *)take(T * 4);
    c.flags = (int32_t *)take((size_t)A * 4);
    c.pos = (int32_t *)take((size_t)A * 4);
    c.temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    c.temp = take(c.temp_bytes);
    c.total_bytes = off;
    return 0;
}

}  // namespace

extern "C" size_t mfcd_sample_workspace_bytes(int64_t attempts, int64_t n_barred)
{
    if (attempts <= 0 || n_barred < 0 || attempts + n_barred >= (int64_t)0xFFFFFFFFll) return 0;
    Carve c;
    return carve(attempts, n_barred, nullptr, c) == 0 ? c.total_bytes : 0;
}

extern "C" int mfcd_sample_triplets(const mfcd_sampler *law, const int64_t *barred_keys, int64_t n_barred,
                                    int64_t attempt0, int64_t attempts, uint64_t seed, int64_t want,
                                    int32_t *triplets_out, int64_t *keys_out, int64_t *counts_out, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    if (!law || attempts <= 0 || n_barred < 0 || want <= 0 || attempt0 < 0 || !triplets_out || !keys_out ||
        !counts_out || !workspace)
        return MFCD_EINVAL;
    if (attempts + n_barred >= (int64_t)0xFFFFFFFFll || (n_barred && !barred_keys)) return MFCD_EINVAL;
    const mfcd_sampler L = *law;
    if (L.n <= 0 || L.m < 2 || (L.users && L.n_users <= 0)) return MFCD_EINVAL;
    if ((double)L.n * (double)L.m * (double)L.m >= 9.2e18) return MFCD_EINVAL;            // the key must fit 63 bits
    if (L.law == MFCD_LAW_ITEM_CDF) {
        if (!L.cdf || (L.pair_rule != 0 && L.pair_rule != 1)) return MFCD_EINVAL;
    } else if (L.law == MFCD_LAW_LISTS) {
        if (!L.list_i || !L.list_j || L.k < 1 || (L.pair_rule == 1 && L.k < 2) || L.list_row_stride < 0)
            return MFCD_EINVAL;
    } else if (L.law != MFCD_LAW_UNIFORM) {
        return MFCD_EINVAL;
    }
    if (L.use_margin && !L.X && (!L.A || !L.B || L.dx <= 0)) return MFCD_EINVAL;
    Carve c;
    if (carve(attempts, n_barred, workspace, c) != 0) return MFCD_EINVAL;
    if (workspace_bytes < c.total_bytes) return MFCD_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t A = attempts, E = n_barred, T = A + E;
    MFCD_HIP_TRY(hipMemsetAsync(c.flags, 0, (size_t)A * 4, s));
    if (E)
        hipLaunchKernelGGL(place_barred_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, barred_keys, E,
                           c.keys_in, c.vals_in);
    hipLaunchKernelGGL(sample_attempts_kernel, dim3((unsigned)((A + 255) / 256)), dim3(256), 0, s, L, attempt0, A,
                       (unsigned)seed, (unsigned)(seed >> 32), c.keys_in, c.vals_in, E);
    MFCD_HIP_TRY(hipGetLastError());
    size_t tb = c.temp_bytes;
    MFCD_HIP_TRY(rocprim::radix_sort_pairs(c.temp, tb, c.keys_in, c.keys_out, c.vals_in, c.vals_out, (size_t)T, 0, 64, s));
    hipLaunchKernelGGL(mark_first_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, c.keys_out, c.vals_out, T,
                       E, c.flags);
    tb = c.temp_bytes;
    MFCD_HIP_TRY(rocprim::exclusive_scan(c.temp, tb, c.flags, c.pos, 0, (size_t)A, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((A + 255) / 256)), dim3(256), 0, s, c.keys_in, c.flags, c.pos, A, E,
                       want, L.m, triplets_out, keys_out, counts_out);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}
