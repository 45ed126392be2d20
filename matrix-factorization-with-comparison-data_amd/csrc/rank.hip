// rank.hip — per-row Spearman rank correlation on gfx950.
//
// Replaces the reference's Python loop of scipy.stats.spearmanr over the rows of the centred U V^T and X
// (structure.py:1023-1031): rho[r] = Pearson correlation of the AVERAGE ranks (scipy.stats.rankdata, ties share the
// mean of their positions) of row r of A and row r of X.
//
// One workgroup of 1024 threads per row pair, everything in LDS:
//   * the row becomes (sortable 32-bit key, 16-bit column) pairs; -0.0 ranks as +0.0 (numeric equality, as numpy
//     compares);
//   * bitonic sort of the pairs in the normalised (all-ascending) form over the next power of two P, the positions past
//     m being virtual keys that sort last and never move (P/2 comparators per stage, log2(P)(log2(P)+1)/2 stages, one
//     barrier each);
//   * a NaN anywhere in either row makes rho NaN (scipy's nan_policy='propagate');
//   * every run of equal keys gets the doubled average rank 2*rank = first + last + 2 (0-based positions), written by
//     the thread that holds the run's first element;
//   * A's doubled ranks are scattered back to column order (u16); during X's pass each sorted element looks up its
//     column's A rank, and the three sums of the correlation are accumulated as EXACT 64-bit integers of the doubled,
//     centred ranks (|2 rank - (m+1)| <= m, sums <= 4 m^3 < 2^46), reduced in fixed order; rho is formed in f64.
// The result is therefore independent of thread scheduling and equals scipy's float64 computation to rounding.
// LDS: 8 m bytes (156 KiB at m = 20000).
// Rows longer than 20448 columns (BASELINE configs[3]: 65536 items) do not fit a workgroup's LDS: mfcd_spearman_rows_long
// sorts a block of rows at a time in global memory — one device-wide SEGMENTED radix sort of (key, column) pairs per
// matrix (a segment = a row) — and a workgroup per row then walks its sorted row exactly as above, with the sorted
// arrays and A's ranks by column in a caller-provided workspace (L2-resident: 512 KiB per row at m = 65536) instead of
// LDS.  Same keys, same run rule, same exact integer sums: the two paths agree bit for bit where both apply.
#include <cstring>

#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"

namespace {

constexpr int kRankThreads = 1024;
constexpr int kRankMaxCols = 20448;   // 8 bytes of LDS per column (+ the reduction scratch) within 160 KiB

__device__ __forceinline__ unsigned sortable_key(float f)
{
    if (f == 0.0f) f = 0.0f;                       // -0.0 -> +0.0
    const unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__device__ __forceinline__ long long block_sum_i64(long long v, long long *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, MFCD_WAVE);
    __syncthreads();                               // red[] may still be read from the previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    long long t = 0;
    for (int w = 0; w < kRankThreads / 64; ++w) t += red[w];   // fixed order, every thread
    return t;
}

__global__ __launch_bounds__(kRankThreads) void spearman_rows_kernel(const float *__restrict__ A, int64_t lda,
                                                                     const float *__restrict__ X, int64_t ldx, int m,
                                                                     int P, double *__restrict__ rho)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *key = reinterpret_cast<unsigned *>(smem);                       // [m]
    unsigned short *idx = reinterpret_cast<unsigned short *>(key + m);         // [m]
    unsigned short *ra2 = idx + m;                                             // [m] doubled rank of A by column
    __shared__ long long red[kRankThreads / 64];
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    const long long centre = (long long)m + 1;     // 2 * mean rank
    long long saa = 0, sxx = 0, sxy = 0;
    int has_nan = 0;   // scipy.stats.spearmanr propagates NaN: a NaN anywhere in either row makes rho NaN

    for (int pass = 0; pass < 2; ++pass) {
        const float *row = pass == 0 ? A + r * lda : X + r * ldx;
        for (int p = tid; p < m; p += kRankThreads) {
            const float f = row[p];
            has_nan |= (f != f);
            key[p] = sortable_key(f);
            idx[p] = (unsigned short)p;
        }
        __syncthreads();
        // Bitonic network over P = next power of two, in its NORMALISED form: every comparator is ascending (the first
        // step of a merge pairs an element with its mirror image in the block, the following steps are half-cleaners).
        // Positions m .. P-1 are VIRTUAL keys that sort last: with ascending comparators only they never move, so a
        // comparator that touches one is a no-op and is skipped, and LDS holds m elements, not P (m = 20000, BASELINE
        // configs[4], fits: 8 m bytes).
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += kRankThreads) {
                    const int i = 2 * j * (t / j) + (t % j);
                    const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : i + j;   // mirror in the k-block / half-cleaner
                    if (l >= m) continue;
                    const unsigned a = key[i], b = key[l];
                    if (a > b) {
                        key[i] = b;
                        key[l] = a;
                        const unsigned short ia = idx[i];
                        idx[i] = idx[l];
                        idx[l] = ia;
                    }
                }
                __syncthreads();
            }
        }
        // NaN keys (sortable value above +inf's) sort last among the real ones
        long long s_own = 0, s_xy = 0;
        for (int p = tid; p < m; p += kRankThreads) {
            const unsigned kp = key[p];
            if (p > 0 && key[p - 1] == kp) continue;            // not the first element of its run
            int e = p;
            while (e + 1 < m && key[e + 1] == kp) ++e;
            const long long c2 = (long long)(p + e + 2) - centre;   // doubled, centred average rank of the run
            s_own += c2 * c2 * (long long)(e - p + 1);
            if (pass == 0) {
                for (int q = p; q <= e; ++q) ra2[idx[q]] = (unsigned short)(p + e + 2);
            } else {
                for (int q = p; q <= e; ++q) s_xy += c2 * ((long long)ra2[idx[q]] - centre);
            }
        }
        if (pass == 0) saa = s_own;
        else { sxx = s_own; sxy = s_xy; }
        __syncthreads();                                         // ra2 complete / key, idx free for the next pass
    }
    const long long Saa = block_sum_i64(saa, red), Sxx = block_sum_i64(sxx, red), Sxy = block_sum_i64(sxy, red);
    const long long Snan = block_sum_i64((long long)has_nan, red);   // (not __syncthreads_or: it takes LDS of its own)
    if (tid == 0)   // 0/0 -> NaN, as scipy for constants
        rho[r] = Snan != 0 ? __longlong_as_double(0x7ff8000000000000ll) : (double)Sxy / (sqrt((double)Saa) * sqrt((double)Sxx));
}


// Round 3: the same kernel with the sort done by a block-wide LSD RADIX sort (rocPRIM's block_radix_sort: keys and
// column indices in registers, IPT per thread, digits ranked with wave-level match operations, exchanged through LDS)
// instead of the 120-stage bitonic network through LDS: ~5x fewer instructions per element at m = 20000.  Everything
// after the sort — average ranks per run of equal keys, exact 64-bit sums, NaN propagation — is the code above.
// LDS: [sort storage, then sorted keys (4 m) + columns (2 m)] + A's doubled ranks by column (2 m).
template <int IPT>
__global__ __launch_bounds__(kRankThreads) void spearman_rows_radix_kernel(const float *__restrict__ A, int64_t lda,
                                                                           const float *__restrict__ X, int64_t ldx,
                                                                           int m, int region0, double *__restrict__ rho)
{
    using sort_t = rocprim::block_radix_sort<unsigned, kRankThreads, IPT, unsigned short>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typename sort_t::storage_type &storage = *reinterpret_cast<typename sort_t::storage_type *>(smem);
    unsigned *key = reinterpret_cast<unsigned *>(smem);                        // [m]   (after the sort)
    unsigned short *idx = reinterpret_cast<unsigned short *>(key + m);          // [m]
    unsigned short *ra2 = reinterpret_cast<unsigned short *>(smem + region0);   // [m] doubled rank of A by column
    __shared__ long long red[kRankThreads / 64];
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    const long long centre = (long long)m + 1;     // 2 * mean rank
    long long saa = 0, sxx = 0, sxy = 0;
    int has_nan = 0;

    for (int pass = 0; pass < 2; ++pass) {
        const float *row = pass == 0 ? A + r * lda : X + r * ldx;
        unsigned k[IPT];
        unsigned short c[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {                  // striped (coalesced) load: any arrangement may enter a sort
            const int p = j * kRankThreads + tid;
            k[j] = 0xFFFFFFFFu;                          // padding sorts last
            c[j] = (unsigned short)0xFFFF;
            if (p < m) {
                const float f = row[p];
                has_nan |= (f != f);
                k[j] = sortable_key(f);
                c[j] = (unsigned short)p;
            }
        }
        __syncthreads();                                 // the previous pass's key / idx readers are done
        sort_t().sort(k, c, storage);                    // -> blocked arrangement: thread t holds positions t*IPT ..
        __syncthreads();                                 // the sort's storage becomes the sorted arrays
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int p = tid * IPT + j;
            if (p < m) {
                key[p] = k[j];
                idx[p] = c[j];
            }
        }
        __syncthreads();
        long long s_own = 0, s_xy = 0;
        for (int p = tid; p < m; p += kRankThreads) {
            const unsigned kp = key[p];
            if (p > 0 && key[p - 1] == kp) continue;            // not the first element of its run
            int e = p;
            while (e + 1 < m && key[e + 1] == kp) ++e;
            const long long c2 = (long long)(p + e + 2) - centre;   // doubled, centred average rank of the run
            s_own += c2 * c2 * (long long)(e - p + 1);
            if (pass == 0) {
                for (int q = p; q <= e; ++q) ra2[idx[q]] = (unsigned short)(p + e + 2);
            } else {
                for (int q = p; q <= e; ++q) s_xy += c2 * ((long long)ra2[idx[q]] - centre);
            }
        }
        if (pass == 0) saa = s_own;
        else { sxx = s_own; sxy = s_xy; }
    }
    const long long Saa = block_sum_i64(saa, red), Sxx = block_sum_i64(sxx, red), Sxy = block_sum_i64(sxy, red);
    const long long Snan = block_sum_i64((long long)has_nan, red);
    if (tid == 0)
        rho[r] = Snan != 0 ? __longlong_as_double(0x7ff8000000000000ll) : (double)Sxy / (sqrt((double)Saa) * sqrt((double)Sxx));
}

template <int IPT>
int launch_radix(const float *A, int64_t lda, const float *X, int64_t ldx, int rows, int m, double *rho, hipStream_t st)
{
    using sort_t = rocprim::block_radix_sort<unsigned, kRankThreads, IPT, unsigned short>;
    size_t region0 = sizeof(typename sort_t::storage_type);
    if (region0 < (size_t)m * 6) region0 = (size_t)m * 6;
    region0 = (region0 + 15) & ~(size_t)15;
    const size_t lds = region0 + (size_t)m * 2;
    static size_t allowed = 0;
    if (lds > allowed) {
        MFCD_HIP_TRY(hipFuncSetAttribute((const void *)spearman_rows_radix_kernel<IPT>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed = lds;
    }
    hipLaunchKernelGGL(spearman_rows_radix_kernel<IPT>, dim3((unsigned)rows), dim3(kRankThreads), lds, st, A, lda, X, ldx, m,
                       (int)region0, rho);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- rows of any length: sorted in global memory --------------------------------------------------------------
struct RowOffset {
    int m;
    __host__ __device__ unsigned operator()(unsigned r) const { return r * (unsigned)m; }
};

__global__ __launch_bounds__(256) void rank_keys_kernel(const float *__restrict__ M, int64_t ld, int rows, int m,
                                                        unsigned *__restrict__ keys, unsigned *__restrict__ cols,
                                                        int *__restrict__ nan_rows)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)rows * m) return;
    const int r = (int)(e / m), c = (int)(e - (int64_t)r * m);
    const float f = M[(int64_t)r * ld + c];
    if (f != f) nan_rows[r] = 1;
    keys[e] = sortable_key(f);
    cols[e] = (unsigned)c;
}

// one workgroup per row: pass 0 walks A's sorted row (ranks by column -> ra2, sum of squares), pass 1 walks X's
__global__ __launch_bounds__(kRankThreads) void spearman_sorted_rows_kernel(const unsigned *__restrict__ keyA,
                                                                            const unsigned *__restrict__ colA,
                                                                            const unsigned *__restrict__ keyX,
                                                                            const unsigned *__restrict__ colX, int m,
                                                                            unsigned *__restrict__ ra2_all,
                                                                            const int *__restrict__ nan_rows,
                                                                            double *__restrict__ rho)
{
    __shared__ long long red[kRankThreads / 64];
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    const long long centre = (long long)m + 1;
    unsigned *ra2 = ra2_all + r * m;
    long long saa = 0, sxx = 0, sxy = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const unsigned *key = (pass == 0 ? keyA : keyX) + r * m, *idx = (pass == 0 ? colA : colX) + r * m;
        long long s_own = 0, s_xy = 0;
        for (int p = tid; p < m; p += kRankThreads) {
            const unsigned kp = key[p];
            if (p > 0 && key[p - 1] == kp) continue;            // not the first element of its run
            int e = p;
            while (e + 1 < m && key[e + 1] == kp) ++e;
            const long long c2 = (long long)(p + e + 2) - centre;
            s_own += c2 * c2 * (long long)(e - p + 1);
            if (pass == 0) {
                for (int q = p; q <= e; ++q) ra2[idx[q]] = (unsigned)(p + e + 2);
            } else {
                for (int q = p; q <= e; ++q) s_xy += c2 * ((long long)ra2[idx[q]] - centre);
            }
        }
        if (pass == 0) saa = s_own;
        else { sxx = s_own; sxy = s_xy; }
        __threadfence_block();
        __syncthreads();                                         // ra2 of this row is complete before X's pass reads it
    }
    const long long Saa = block_sum_i64(saa, red), Sxx = block_sum_i64(sxx, red), Sxy = block_sum_i64(sxy, red);
    if (tid == 0)
        rho[r] = nan_rows[r] ? __longlong_as_double(0x7ff8000000000000ll) : (double)Sxy / (sqrt((double)Saa) * sqrt((double)Sxx));
}

struct LongCarve {
    unsigned *k_in, *c_in, *kA, *cA, *kX, *cX, *ra2;
    int *nan_rows;
    void *temp;
    size_t temp_bytes, total;
};

int long_carve(int block_rows, int m, void *base, LongCarve &c)
{
    const size_t E = (size_t)block_rows * m;
    size_t tb = 0;
    using off_it = rocprim::transform_iterator<rocprim::counting_iterator<unsigned>, RowOffset>;
    off_it begin(rocprim::counting_iterator<unsigned>(0), RowOffset{m}), end(rocprim::counting_iterator<unsigned>(1), RowOffset{m});
    if (rocprim::segmented_radix_sort_pairs(nullptr, tb, (unsigned *)nullptr, (unsigned *)nullptr, (unsigned *)nullptr,
                                            (unsigned *)nullptr, (unsigned)E, (unsigned)block_rows, begin, end, 0, 32,
                                            (hipStream_t)0) != hipSuccess)
        return MFCD_EINVAL;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *q = p ? p + off : nullptr; off += (bytes + 255) & ~(size_t)255; return (void *)q; };
    c.k_in = (unsigned *)take(E * 4); c.c_in = (unsigned *)take(E * 4);
    c.kA = (unsigned *)take(E * 4); c.cA = (unsigned *)take(E * 4);
    c.kX = (unsigned *)take(E * 4); c.cX = (unsigned *)take(E * 4);
    c.ra2 = (unsigned *)take(E * 4);
    c.nan_rows = (int *)take((size_t)block_rows * 4);
    c.temp_bytes = tb;
    c.temp = take(tb);
    c.total = off;
    return 0;
}

// rows sorted per call of the long path: bounds the workspace (28 bytes per element) near 256 MiB
int long_block_rows(int rows, int m)
{
    int64_t b = ((int64_t)256 << 20) / ((int64_t)28 * m);
    if (b < 1) b = 1;
    return (int)(b < rows ? b : rows);
}

int g_rank_sort = 1;   // 1 = radix (default), 0 = bitonic network (mfcd_set_tuning(MFCD_TUNE_RANK_SORT))

}  // namespace

namespace mfcd_detail {
int set_rank_sort(int v)
{
    if (v != 0 && v != 1) return MFCD_EINVAL;
    g_rank_sort = v;
    return 0;
}
}  // namespace mfcd_detail

extern "C" int mfcd_spearman_max_columns(void) { return kRankMaxCols; }

extern "C" int mfcd_spearman_rows(const float *A, int64_t lda, const float *X, int64_t ldx, int rows, int m,
                                  double *rho, void *stream)
{
    if (!A || !X || !rho || rows < 0 || m <= 0 || lda < m || ldx < m) return MFCD_EINVAL;
    if (m > kRankMaxCols) return MFCD_EINVAL;
    if (rows == 0) return 0;
    if (g_rank_sort == 1) {
        if (m <= 4 * kRankThreads) return launch_radix<4>(A, lda, X, ldx, rows, m, rho, (hipStream_t)stream);
        if (m <= 8 * kRankThreads) return launch_radix<8>(A, lda, X, ldx, rows, m, rho, (hipStream_t)stream);
        return launch_radix<20>(A, lda, X, ldx, rows, m, rho, (hipStream_t)stream);
    }
    int P = 2;
    while (P < m) P <<= 1;
    const size_t lds = (size_t)m * 8;
    static size_t allowed = 0;
    if (lds > allowed) {
        MFCD_HIP_TRY(hipFuncSetAttribute((const void *)spearman_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
        allowed = lds;
    }
    hipLaunchKernelGGL(spearman_rows_kernel, dim3((unsigned)rows), dim3(kRankThreads), lds, (hipStream_t)stream, A, lda, X,
                       ldx, m, P, rho);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" size_t mfcd_spearman_long_workspace_bytes(int rows, int m)
{
    if (rows <= 0 || m <= 0 || (int64_t)long_block_rows(rows, m) * m >= (int64_t)0x7fffffff) return 0;
    LongCarve c;
    return long_carve(long_block_rows(rows, m), m, nullptr, c) == 0 ? c.total : 0;
}

extern "C" int mfcd_spearman_rows_long(const float *A, int64_t lda, const float *X, int64_t ldx, int rows, int m,
                                       double *rho, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!A || !X || !rho || rows < 0 || m <= 0 || lda < m || ldx < m) return MFCD_EINVAL;
    if (rows == 0) return 0;
    if (!workspace) return MFCD_EINVAL;
    const int R = long_block_rows(rows, m);
    if ((int64_t)R * m >= (int64_t)0x7fffffff) return MFCD_EINVAL;
    LongCarve c;
    if (long_carve(R, m, workspace, c) != 0) return MFCD_EINVAL;
    if (workspace_bytes < c.total) return MFCD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    using off_it = rocprim::transform_iterator<rocprim::counting_iterator<unsigned>, RowOffset>;
    off_it begin(rocprim::counting_iterator<unsigned>(0), RowOffset{m}), end(rocprim::counting_iterator<unsigned>(1), RowOffset{m});
    for (int r0 = 0; r0 < rows; r0 += R) {
        const int nr = rows - r0 < R ? rows - r0 : R;
        const int64_t E = (int64_t)nr * m;
        MFCD_HIP_TRY(hipMemsetAsync(c.nan_rows, 0, (size_t)nr * 4, st));
        for (int pass = 0; pass < 2; ++pass) {
            const float *M = pass == 0 ? A + (int64_t)r0 * lda : X + (int64_t)r0 * ldx;
            hipLaunchKernelGGL(rank_keys_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, M, pass == 0 ? lda : ldx,
                               nr, m, c.k_in, c.c_in, c.nan_rows);
            size_t tb = c.temp_bytes;
            MFCD_HIP_TRY(rocprim::segmented_radix_sort_pairs(c.temp, tb, c.k_in, pass == 0 ? c.kA : c.kX, c.c_in,
                                                             pass == 0 ? c.cA : c.cX, (unsigned)E, (unsigned)nr, begin, end,
                                                             0, 32, st));
        }
        hipLaunchKernelGGL(spearman_sorted_rows_kernel, dim3((unsigned)nr), dim3(kRankThreads), 0, st, c.kA, c.cA, c.kX, c.cX,
                           m, c.ra2, c.nan_rows, rho + r0);
        MFCD_HIP_TRY(hipGetLastError());
    }
    return 0;
}
