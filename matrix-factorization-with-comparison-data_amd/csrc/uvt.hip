// uvt.hip — dense UV^T reconstruction / correlation pass on gfx950 without materialising UV^T.
//
// Replaces the GEMM + reductions of compute_reconstruction_error (structure.py:940-952) and of
// compute_alpha_and_norm_ratios (structure.py:982-996, 1003-1009, 1038-1064).
//
// Algebra that removes passes over the n x m product (G = U V^T):
//   column-centred  G - colmean(G) = G[r][c] - cm[c],  cm[c] = mean_r(U) . V[c]     (structure.py:943)
//   row-centred     G - rowmean(G) = G[r][c] - rm[r],  rm[r] = U[r] . mean_c(V)     (structure.py:985)
//   row-centred X   c = X[r][c] - xm[r]; xm and sum c^2, sum x^2 come from one f64 pre-pass over X
// so one MFMA pass with a fused epilogue yields, per row, sum a*c and sum a*a (a = G - rm) and,
// globally, sum (G - cm - sX)^2.  Every scalar the reference derives (alpha, norm ratio, scaled
// errors, per-row Pearson / slope / alpha_i) is a function of those sums (host side, f64).
//
// Main kernel: one wave owns a 32-row block of U (A operand resident in registers for the whole
// sweep) and walks 32-column tiles of a column split; v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain),
// accumulators in registers, epilogue reads the matching X tile straight from HBM (two 128-byte row
// segments per load instruction).  The k index of the MFMA is permuted (lane half h, step kk ->
// k = h*D/2 + kk) so that each lane's A and B fragments are contiguous floats of one row (16-byte loads).
// Roofline: MFMA-bound for d >= 64 (2*n*m*d flop vs 4*n*m bytes of X), HBM-bound (X read) below.
#include "common.h"

#ifndef MFCD_UVT_EXP
#define MFCD_UVT_EXP 0   // diagnostic builds only (tools/): 1 no X loads, 2 no f64 epilogue, 3 no MFMA
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSlices = 128;  // row slices for the deterministic column-sum of U and V

// partial[slice][k] = sum over rows of the slice of T[row][k]  (f64), grid = (kSlices, 2 tables)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                             int n, int m, int d, double *__restrict__ part)
{
    const bool isV = blockIdx.y == 1;
    const float *T = isV ? V : U;
    const int rows = isV ? m : n;
    const int per = (rows + kSlices - 1) / kSlices;
    const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    double *out = part + ((size_t)blockIdx.y * kSlices + blockIdx.x) * d;
    for (int k = threadIdx.x; k < d; k += 256) {
        double acc = 0.0;
        for (int r = r0; r < r1; ++r) acc += (double)T[(int64_t)r * d + k];
        out[k] = acc;
    }
}

// bar[table][k] = (sum over slices) / rows, stored fp32 (the reference works on fp32 tensors)
__global__ __launch_bounds__(256) void colsum_final_kernel(const double *__restrict__ part, int n, int m, int d,
                                                           float *__restrict__ bar)
{
    const int tab = blockIdx.y;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < d; k += gridDim.x * 256) {
        double acc = 0.0;
        for (int s = 0; s < kSlices; ++s) acc += part[((size_t)tab * kSlices + s) * d + k];
        bar[(size_t)tab * d + k] = (float)(acc / (double)(tab ? m : n));
    }
}

// One wave per row: rm[r] = U[r].vbar (rows 0..n-1), cm[c] = ubar.V[c] (rows n..n+m-1); f64 accumulate.
__global__ __launch_bounds__(256) void centre_vectors_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                             const float *__restrict__ bar, int n, int m, int d,
                                                             float *__restrict__ rm, float *__restrict__ cm)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)n + m) return;
    const bool isV = w >= n;
    const float *row = isV ? V + (w - n) * d : U + w * d;
    const float *other = isV ? bar : bar + d;  // V rows pair with ubar (bar[0]), U rows with vbar (bar[1])
    double acc = 0.0;
    for (int k = lane; k < d; k += MFCD_WAVE) acc += (double)row[k] * (double)other[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, MFCD_WAVE);
    if (lane == 0) {
        if (isV) cm[w - n] = (float)acc;
        else rm[w] = (float)acc;
    }
}

// One wave per row of X: xrow[r] = {mean (fp32), sum (x-mean)^2 (f64), sum x^2 (f64)}
__global__ __launch_bounds__(256) void x_rows_kernel(const float *__restrict__ X, int n, int m, float *__restrict__ xm,
                                                     double *__restrict__ scc, double *__restrict__ sxx)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float *row = X + r * m;
    double s1 = 0.0, s2 = 0.0;
    for (int c = lane; c < m; c += MFCD_WAVE) {
        const double x = (double)row[c];
        s1 += x;
        s2 += x * x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, MFCD_WAVE);
        s2 += __shfl_xor(s2, off, MFCD_WAVE);
    }
    // second sweep with the fp32 mean, as the reference centres in fp32 (structure.py:987)
    const float mean = (float)(s1 / (double)m);
    double cc = 0.0;
    for (int c = lane; c < m; c += MFCD_WAVE) {
        const double cx = (double)(row[c] - mean);
        cc += cx * cx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cc += __shfl_xor(cc, off, MFCD_WAVE);
    if (lane == 0) {
        xm[r] = mean;
        scc[r] = cc;
        sxx[r] = s2;
    }
}

// acc[reg] of a 32x32 tile:  row = (reg&3) + 8*(reg>>2) + 4*half,  col = lane&31   (gfx950 C/D map)
__device__ __forceinline__ int tile_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// MFMA tile: D is the compile-time factor width (multiple of 8).  a[] holds this lane's A fragment.
template <int D>
__device__ __forceinline__ void tile_mfma(const float (&a)[D / 2], const float *__restrict__ vrow, f32x16 &acc)
{
    float b[D / 2];
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(vrow + 4 * q);
        b[4 * q + 0] = t.x; b[4 * q + 1] = t.y; b[4 * q + 2] = t.z; b[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int kk = 0; kk < D / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
}

// part_rows[split][row][2] (sum a*c, sum a*a) and part_err[wave] for the final reduction.
template <int D>
__global__ __launch_bounds__(256) void uvt_main_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                       const float *__restrict__ X, const float *__restrict__ rm,
                                                       const float *__restrict__ cm, const float *__restrict__ xm,
                                                       int n, int m, int d_rt, float s, int cols_per_split,
                                                       double *__restrict__ part_rows, double *__restrict__ part_err)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int rt = blockIdx.x * 4 + (threadIdx.x >> 6);  // 32-row tile index
    const int split = blockIdx.y;
    const int row0 = rt * 32;
    if (row0 >= n) return;  // whole wave exits together
    const int c_begin = split * cols_per_split;
    const int c_end = min(m, c_begin + cols_per_split);

    // A fragment: U[row0 + l31][half*D/2 .. +D/2), rows past n clamp to n-1 (masked in the epilogue)
    float a[(D > 0 ? D : 8) / 2];
    float rmr[16], xmr[16];
    bool rok[16];
    double sac[16], saa[16];
    if constexpr (D > 0) {
        const float *urow = U + (int64_t)min(row0 + l31, n - 1) * D + half * (D / 2);
#pragma unroll
        for (int q = 0; q < D / 8; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(urow + 4 * q);
            a[4 * q + 0] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + tile_row(r, half);
        rok[r] = row < n;
        const int rc = min(row, n - 1);
        rmr[r] = rm[rc];
        xmr[r] = xm[rc];
        sac[r] = 0.0;
        saa[r] = 0.0;
    }
    double err2 = 0.0;

    for (int c0 = c_begin; c0 < c_end; c0 += 32) {
        const int col = c0 + l31;
        const bool cok = col < c_end;
        const int cc = min(col, m - 1);
        // X tile first: 16 loads in flight under the MFMA chain
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = X[(int64_t)min(row0 + tile_row(r, half), n - 1) * m + cc];
        const float cmc = cm[cc];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        if constexpr (D > 0) {
            tile_mfma<D>(a, V + (int64_t)cc * D + half * (D / 2), acc);
        } else {
            // generic factor width: plain FMA loop in the same accumulator layout
            const float *vrow = V + (int64_t)cc * d_rt;
            for (int k = 0; k < d_rt; ++k) {
                const float vk = vrow[k];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[r] += U[(int64_t)min(row0 + tile_row(r, half), n - 1) * d_rt + k] * vk;
            }
            if (half) { /* both halves computed their own rows; nothing to exchange */ }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (rok[r] && cok) {
                const float g = acc[r];
                const float av = g - rmr[r];     // structure.py:985
                const float cv = x[r] - xmr[r];  // structure.py:987
                sac[r] += (double)av * (double)cv;
                saa[r] += (double)av * (double)av;
                const float e = (g - cmc) - s * x[r];  // structure.py:943, 949
                err2 += (double)e * (double)e;
            }
        }
    }
    // reduce each row over the 32 lanes of its half (xor offsets < 32 stay inside the half)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            sac[r] += __shfl_xor(sac[r], off, MFCD_WAVE);
            saa[r] += __shfl_xor(saa[r], off, MFCD_WAVE);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off, MFCD_WAVE);
    if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + tile_row(r, half);
            if (row < n) {
                double *o = part_rows + ((size_t)split * n + row) * 2;
                o[0] = sac[r];
                o[1] = saa[r];
            }
        }
    }
    if (lane == 0) part_err[(size_t)split * ((n + 31) / 32) + rt] = err2;
}

// Tiled form of the main kernel for D in {32, 64, 128, 256}: a workgroup of NW waves owns NW*32 rows of U (each
// wave keeps its 32-row A fragment in registers for the whole sweep, as above) and the TC-column stages of V are
// staged through LDS once per WORKGROUP (double-buffered, one barrier per stage, coalesced 16-byte global loads
// issued a stage ahead) instead of once per wave from L2 — at D = 256 the per-wave V fetch (32 KiB per 32x32
// tile) was what bounded the kernel.  LDS rows are padded by 4 floats so the per-lane 16-byte fragment reads
// (lane = column, stride = one row) spread over the banks.  With NW = 8 two waves share a SIMD, so one wave's
// f64 epilogue overlaps the other's MFMA chain.
// Work mapping is XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs by the hardware; when there are
// >= 8 column splits, every row block of split s runs on XCD s % 8, whose L2 then serves that split's V rows.
template <int D, int NW, int TC>
__global__ __launch_bounds__(NW * 64) void uvt_tiled_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                            const float *__restrict__ X, const float *__restrict__ rm,
                                                            const float *__restrict__ cm, const float *__restrict__ xm,
                                                            int n, int m, float s, int cols_per_split, int splits,
                                                            int row_blocks, double *__restrict__ part_rows,
                                                            double *__restrict__ part_err)
{
    constexpr int LD = D + 4, NT = NW * 64, NLD = TC * D / 4 / NT;   // padded LDS row; float4 loads per thread per stage
    static_assert(TC * D / 4 % NT == 0 && TC % 32 == 0, "stage must split evenly over the workgroup");
    extern __shared__ __attribute__((aligned(16))) float vt[];   // [2][TC][LD]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar, so that row bases below stay in SGPRs
    int split, rb;
    if (splits >= 8) {
        const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
        split = (w / row_blocks) * 8 + xcd;
        rb = w % row_blocks;
    } else {
        split = blockIdx.x / row_blocks;
        rb = blockIdx.x % row_blocks;
    }
    if (split >= splits) return;   // whole workgroup
    const int rt = rb * NW + wave, row0 = rt * 32;
    const bool active = row0 < n;  // waves past the last row still help with the V stages and the barriers
    const bool full_rows = row0 + 32 <= n;   // wave-uniform: every row of this wave's tile exists
    const int c_begin = split * cols_per_split;
    const int c_end = min(m, c_begin + cols_per_split);

    float a[D / 2];
    float rmr[16], xmr[16];
    bool rok[16];
    double sac[16], saa[16];
    {
        const float *urow = U + (int64_t)min(row0 + l31, n - 1) * D + half * (D / 2);
#pragma unroll
        for (int q = 0; q < D / 8; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(urow + 4 * q);
            a[4 * q + 0] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + tile_row(r, half);
        rok[r] = row < n;
        const int rc = min(row, n - 1);
        rmr[r] = rm[rc];
        xmr[r] = xm[rc];
        sac[r] = 0.0;
        saa[r] = 0.0;
    }
    double err2 = 0.0;

    // this thread's share of a V stage: NLD 16-byte pieces, piece idx = tid + i*NT -> (row idx / (D/4), float4 idx % (D/4))
    float4 stg[NLD];
#define MFCD_LOAD_STAGE(C0)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                        \
        const int idx = tid + i * NT, r = idx / (D / 4), k4 = idx % (D / 4);                                 \
        stg[i] = *reinterpret_cast<const float4 *>(V + (int64_t)min((C0) + r, m - 1) * D + 4 * k4);         \
    }
#define MFCD_STORE_STAGE(BUF)                                                                                \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                        \
        const int idx = tid + i * NT, r = idx / (D / 4), k4 = idx % (D / 4);                                 \
        *reinterpret_cast<float4 *>(vt + (size_t)(BUF) * TC * LD + r * LD + 4 * k4) = stg[i];                \
    }
    MFCD_LOAD_STAGE(c_begin)
    MFCD_STORE_STAGE(0)
    __syncthreads();

    int buf = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += TC) {
        const bool has_next = c0 + TC < c_end;
        if (has_next) { MFCD_LOAD_STAGE(c0 + TC) }
        if (active) {
#pragma unroll 1
            for (int j = 0; j < TC / 32; ++j) {
                const int cb = c0 + 32 * j;
                if (cb >= c_end) break;
                const int col = cb + l31;
                const bool cok = col < c_end;
                const int cc = min(col, m - 1);
                // X tile: scalar row base (rows past n are masked below, so any in-range row will do: clamp so that
                // the +4 of the upper lane half stays inside) + one 32-bit lane offset
                float x[16];
                const unsigned xoff = (unsigned)(half * 4) * (unsigned)m + (unsigned)cc;
                if (full_rows) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float *xr = X + (int64_t)(row0 + tile_row(r, 0)) * m;
#if MFCD_UVT_EXP == 1
                        x[r] = (float)xoff;
#else
                        x[r] = xr[xoff];
#endif
                    }
                } else {   // last row tile of a ragged n: per-lane clamped rows (rows past n are masked below)
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = X[(int64_t)min(row0 + tile_row(r, half), n - 1) * m + cc];
                }
                const float cmc = cm[cc];
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                const float *brow = vt + (size_t)buf * TC * LD + (32 * j + l31) * LD + half * (D / 2);
                // B fragments from LDS in batches of QB 16-byte reads, one batch ahead of the MFMAs that consume them
                // (the scheduling barrier keeps the compiler from hoisting every read to the top: 64+ registers)
                constexpr int QB = 4, NB = D / 8 / QB;
                float4 t[2][QB];
#pragma unroll
                for (int q = 0; q < QB; ++q) t[0][q] = *reinterpret_cast<const float4 *>(brow + 4 * q);
#pragma unroll
                for (int bi = 0; bi < NB; ++bi) {
                    if (bi + 1 < NB) {
#pragma unroll
                        for (int q = 0; q < QB; ++q)
                            t[(bi + 1) & 1][q] = *reinterpret_cast<const float4 *>(brow + 4 * ((bi + 1) * QB + q));
                    }
#pragma unroll
                    for (int q = 0; q < QB; ++q) {
                        const float4 tq = t[bi & 1][q];
                        const int k0 = 4 * (bi * QB + q);
#if MFCD_UVT_EXP == 3
                        acc[0] += a[k0] * tq.x + a[k0 + 1] * tq.y + a[k0 + 2] * tq.z + a[k0 + 3] * tq.w;
                        continue;
#endif
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k0 + 0], tq.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k0 + 1], tq.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k0 + 2], tq.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k0 + 3], tq.w, acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {   // branch-free: out-of-range rows / columns contribute exact zeros
                    const bool ok = rok[r] && cok;
                    const float g = acc[r];
                    const float av = ok ? g - rmr[r] : 0.0f;             // structure.py:985
                    const float cv = x[r] - xmr[r];                      // structure.py:987
                    const float e = ok ? (g - cmc) - s * x[r] : 0.0f;    // structure.py:943, 949
#if MFCD_UVT_EXP == 2
                    err2 += (double)(av + cv + e);
                    continue;
#endif
                    sac[r] += (double)av * (double)cv;
                    saa[r] += (double)av * (double)av;
                    err2 += (double)e * (double)e;
                }
            }
        }
        if (has_next) { MFCD_STORE_STAGE(buf ^ 1) }
        __syncthreads();
        buf ^= 1;
    }
#undef MFCD_LOAD_STAGE
#undef MFCD_STORE_STAGE
    if (!active) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            sac[r] += __shfl_xor(sac[r], off, MFCD_WAVE);
            saa[r] += __shfl_xor(saa[r], off, MFCD_WAVE);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off, MFCD_WAVE);
    if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + tile_row(r, half);
            if (row < n) {
                double *o = part_rows + ((size_t)split * n + row) * 2;
                o[0] = sac[r];
                o[1] = saa[r];
            }
        }
    }
    if (lane == 0) part_err[(size_t)split * ((n + 31) / 32) + rt] = err2;
}

// row_stats[r][8] and scal[4] from the partials (fixed order → deterministic)
__global__ __launch_bounds__(256) void uvt_final_kernel(const double *__restrict__ part_rows,
                                                        const double *__restrict__ part_err,
                                                        const float *__restrict__ rm, const float *__restrict__ xm,
                                                        const double *__restrict__ scc, const double *__restrict__ sxx,
                                                        int n, int splits, int n_err, double s,
                                                        double *__restrict__ row_stats, double *__restrict__ scal)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) {
        double ac = 0.0, aa = 0.0;
        for (int sp = 0; sp < splits; ++sp) {
            ac += part_rows[((size_t)sp * n + r) * 2 + 0];
            aa += part_rows[((size_t)sp * n + r) * 2 + 1];
        }
        double *o = row_stats + (size_t)r * 8;
        o[0] = ac; o[1] = aa; o[2] = scc[r]; o[3] = (double)rm[r]; o[4] = (double)xm[r];
        o[5] = sxx[r]; o[6] = 0.0; o[7] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const int lane = threadIdx.x;
        double e = 0.0, q = 0.0;
        for (int k = lane; k < n_err; k += MFCD_WAVE) e += part_err[k];
        for (int k = lane; k < n; k += MFCD_WAVE) q += sxx[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            e += __shfl_xor(e, off, MFCD_WAVE);
            q += __shfl_xor(q, off, MFCD_WAVE);
        }
        if (lane == 0) {
            scal[0] = e;
            scal[1] = s * s * q;  // ||sX||_F^2  (structure.py:946)
            scal[2] = 0.0;
            scal[3] = 0.0;
        }
    }
}

// k rows of UV^T (structure.py:389-392): one wave per (row, 64-column chunk)
__global__ __launch_bounds__(256) void uvt_rows_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                       const int32_t *__restrict__ row_ids, int k, int m, int d,
                                                       float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    (void)lane;
    if (r >= k || c >= m) return;
    const float *ur = U + (int64_t)row_ids[r] * d, *vc = V + (int64_t)c * d;
    float acc = 0.0f;
    for (int q = 0; q < d; ++q) acc += ur[q] * vc[q];
    out[(int64_t)r * m + c] = acc;
}

struct UvtWs {
    double *colpart;  // [2][kSlices][d]
    float *bar;       // [2][d]
    float *rm, *cm, *xm;
    double *scc, *sxx, *part_rows, *part_err;
    int splits, cols_per_split, n_err;
    size_t bytes;
};

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct TiledCfg {
    int NW, TC;   // waves per workgroup (32 rows each), columns per LDS stage; NW == 0: no tiled form for this d
};

TiledCfg tiled_cfg(int d)
{
    switch (d) {
    case 256: return {4, 32};    // A fragment = 128 registers: one wave per SIMD
    case 128: return {8, 32};
    case 64: return {8, 64};
    case 32: return {8, 128};
    default: return {0, 0};
    }
}

UvtWs plan_ws(char *base, int n, int m, int d)
{
    UvtWs w;
    const int rtiles = (n + 31) / 32;
    const TiledCfg tc = n >= 32 ? tiled_cfg(d) : TiledCfg{0, 0};
    if (tc.NW) {
        // tiled form: >= ~3 workgroups per CU in total, a split's V rows small enough for one XCD's L2 (4 MiB),
        // >= 2 stages per split; 8 or more splits (a multiple of 8 where the column count allows) so that the
        // XCD-aware mapping applies
        const int row_blocks = (n + tc.NW * 32 - 1) / (tc.NW * 32);
        const int stages = (m + tc.TC - 1) / tc.TC;
        int64_t want = (768 + row_blocks - 1) / row_blocks;
        const int64_t by_l2 = ((int64_t)m * d * 4 + (2 << 20) - 1) / (2 << 20);
        if (by_l2 > want) want = by_l2;
        int splits = want <= 1 ? 1 : (int)((want + 7) / 8 * 8);
        const int max_splits = stages / 2 > 0 ? stages / 2 : 1;
        if (splits > max_splits) splits = max_splits >= 8 ? max_splits / 8 * 8 : max_splits;
        if (splits > 256) splits = 256;
        const int per = (stages + splits - 1) / splits;
        w.cols_per_split = per * tc.TC;
        w.splits = (m + w.cols_per_split - 1) / w.cols_per_split;
    } else {
        // enough (row tile, split) wave items to fill 256 CUs several times over, 32-column granularity
        int splits = 1;
        const int ctiles = (m + 31) / 32;
        while (splits < ctiles && (int64_t)rtiles * splits < 4096) splits *= 2;
        if (splits > ctiles) splits = ctiles;
        if (splits > 64) splits = 64;
        w.cols_per_split = ((ctiles + splits - 1) / splits) * 32;
        w.splits = (m + w.cols_per_split - 1) / w.cols_per_split;
    }
    w.n_err = w.splits * rtiles;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base ? base + off : nullptr; off += al(b); return p; };
    w.colpart = (double *)take(sizeof(double) * 2 * kSlices * (size_t)d);
    w.bar = (float *)take(sizeof(float) * 2 * (size_t)d);
    w.rm = (float *)take(sizeof(float) * (size_t)n);
    w.cm = (float *)take(sizeof(float) * (size_t)m);
    w.xm = (float *)take(sizeof(float) * (size_t)n);
    w.scc = (double *)take(sizeof(double) * (size_t)n);
    w.sxx = (double *)take(sizeof(double) * (size_t)n);
    w.part_rows = (double *)take(sizeof(double) * 2 * (size_t)n * w.splits);
    w.part_err = (double *)take(sizeof(double) * (size_t)w.n_err);
    w.bytes = off;
    return w;
}

}  // namespace

extern "C" size_t mfcd_uvt_workspace_bytes(int n, int m, int d)
{
    if (n <= 0 || m <= 0 || d <= 0) return 0;
    return plan_ws(nullptr, n, m, d).bytes;
}

extern "C" int mfcd_uvt_stats(const float *U, const float *V, const float *X, int n, int m, int d, double s,
                              double *row_stats, double *scal, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!U || !V || !X || !row_stats || !scal || !workspace || n <= 0 || m <= 0 || d <= 0 || d > MFCD_MAX_D)
        return MFCD_EINVAL;
    const UvtWs w = plan_ws((char *)workspace, n, m, d);
    if (workspace_bytes < w.bytes) return MFCD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(kSlices, 2), dim3(256), 0, st, U, V, n, m, d, w.colpart);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((d + 255) / 256, 2), dim3(256), 0, st, w.colpart, n, m, d, w.bar);
    hipLaunchKernelGGL(centre_vectors_kernel, dim3((unsigned)(((int64_t)n + m + 3) / 4)), dim3(256), 0, st, U, V,
                       w.bar, n, m, d, w.rm, w.cm);
    hipLaunchKernelGGL(x_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, X, n, m, w.xm, w.scc, w.sxx);
    const dim3 grid(((n + 31) / 32 + 3) / 4, w.splits);
    const bool al16 = ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(V)) & 15u) == 0;
    const TiledCfg tc = n >= 32 ? tiled_cfg(d) : TiledCfg{0, 0};
#define MFCD_UVT(DD)                                                                                            \
    hipLaunchKernelGGL((uvt_main_kernel<DD>), grid, dim3(256), 0, st, U, V, X, w.rm, w.cm, w.xm, n, m, d, (float)s, \
                       w.cols_per_split, w.part_rows, w.part_err)
#define MFCD_UVT_TILED(DD, NW, TC)                                                                               \
    do {                                                                                                         \
        const int row_blocks = (n + NW * 32 - 1) / (NW * 32);                                                    \
        const unsigned blocks = (unsigned)row_blocks * (w.splits >= 8 ? 8u * ((w.splits + 7) / 8) : (unsigned)w.splits); \
        const size_t lds = sizeof(float) * 2 * TC * (DD + 4);                                                    \
        MFCD_HIP_TRY(hipFuncSetAttribute((const void *)uvt_tiled_kernel<DD, NW, TC>,                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
        hipLaunchKernelGGL((uvt_tiled_kernel<DD, NW, TC>), dim3(blocks), dim3(NW * 64), lds, st, U, V, X, w.rm,   \
                           w.cm, w.xm, n, m, (float)s, w.cols_per_split, w.splits, row_blocks, w.part_rows,      \
                           w.part_err);                                                                          \
    } while (0)
    if (al16 && tc.NW && d == 256) MFCD_UVT_TILED(256, 4, 32);
    else if (al16 && tc.NW && d == 128) MFCD_UVT_TILED(128, 8, 32);
    else if (al16 && tc.NW && d == 64) MFCD_UVT_TILED(64, 8, 64);
    else if (al16 && tc.NW && d == 32) MFCD_UVT_TILED(32, 8, 128);
    else if (al16 && d == 8) MFCD_UVT(8);
    else if (al16 && d == 16) MFCD_UVT(16);
    else if (al16 && d == 32) MFCD_UVT(32);
    else if (al16 && d == 64) MFCD_UVT(64);
    else if (al16 && d == 128) MFCD_UVT(128);
    else if (al16 && d == 256) MFCD_UVT(256);
    else MFCD_UVT(0);
#undef MFCD_UVT
#undef MFCD_UVT_TILED
    hipLaunchKernelGGL(uvt_final_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w.part_rows, w.part_err, w.rm, w.xm,
                       w.scc, w.sxx, n, w.splits, w.n_err, s, row_stats, scal);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mfcd_uvt_rows(const float *U, const float *V, const int32_t *row_ids, int k, int n, int m, int d,
                             float *out, void *stream)
{
    if (!U || !V || !out || k < 0 || n <= 0 || m <= 0 || d <= 0) return MFCD_EINVAL;
    if (k == 0) return 0;
    if (!row_ids) return MFCD_EINVAL;
    hipLaunchKernelGGL(uvt_rows_kernel, dim3((m + 255) / 256, k), dim3(256), 0, (hipStream_t)stream, U, V, row_ids, k,
                       m, d, out);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}
