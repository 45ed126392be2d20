// uvt.hip — dense UV^T reconstruction / correlation pass on gfx950 without materialising UV^T.
//
// Replaces the GEMM + reductions of compute_reconstruction_error (structure.py:940-952) and of
// compute_alpha_and_norm_ratios (structure.py:982-996, 1003-1009, 1038-1064).
//
// Algebra that removes passes over the n x m product (G = U V^T):
//   column-centred  G - colmean(G) = G[r][c] - cm[c],  cm[c] = mean_r(U) . V[c]     (structure.py:943)
//   row-centred     G - rowmean(G) = G[r][c] - rm[r],  rm[r] = U[r] . mean_c(V)     (structure.py:985)
//   row-centred X   c = X[r][c] - xm[r]; xm, sum c^2 and sum x^2 are per-row sums of the SAME sweep (tiled form: taken
//                   relative to a per-split shift and re-centred in f64 afterwards; generic form: an f64 pre-pass)
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

#ifndef MFCD_UVT_STAMPS
#define MFCD_UVT_STAMPS 0   // diagnostic builds only (tools/diag_uvt_stamps.py): in-kernel cycle accounting per phase
#endif
#if MFCD_UVT_STAMPS
__device__ unsigned long long mfcd_uvt_dbg[8];   // chain, epilogue, sync, dma-issue cycles; tiles; waves
// MFCD_UVT_STAMPS=2: no per-phase stamp (the kernel keeps its timing); only the clock of the stage loop:
// [6] += shader cycles, [7] += 100 MHz real-time ticks, per wave
#if MFCD_UVT_STAMPS == 1
#define MFCD_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define MFCD_STAMP(var) const unsigned long long var = 0
#endif
#else
#define MFCD_STAMP(var)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSlices = 256;  // row slices for the deterministic column-sum of U and V
constexpr int kTiledRowSums = 6;   // doubles the tiled kernel leaves per (split, row) for the final kernel

// a table of `rows` rows is cut into slices of >= 128 rows (at most kSlices of them)
__host__ __device__ inline int slices_for(int rows) { return rows >= 128 * kSlices ? kSlices : (rows + 127) / 128; }

// partial[slice][k] = sum over rows of the slice of T[row][k]  (f64), grid = (kSlices, 2 tables).
// The 256 threads form G = 256/d row groups (1 when d >= 256): thread -> (group g, column k); a group walks every
// G-th row of the slice with four loads in flight, the groups are combined through LDS in fixed order.
// The first kernel of every pass: it also zeroes the pass's arrival counters (cnt[0 .. ncnt)), which the tiled main
// kernel's tail counts workgroups with (the workspace is the caller's memory: nothing in it survives between passes).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                             int n, int m, int d, double *__restrict__ part,
                                                             unsigned *__restrict__ cnt, int ncnt)
{
    __shared__ double red[256];
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < ncnt; i += 256) cnt[i] = 0u;
    const bool isV = blockIdx.y == 1;
    const float *T = isV ? V : U;
    const int rows = isV ? m : n;
    const int nsl = slices_for(rows);
    if ((int)blockIdx.x >= nsl) return;
    const int per = (rows + nsl - 1) / nsl;
    const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    double *out = part + ((size_t)blockIdx.y * kSlices + blockIdx.x) * d;
    const int G = d >= 256 ? 1 : 256 / d;
    for (int kb = 0; kb < d; kb += 256) {
        const int g = d >= 256 ? 0 : (int)threadIdx.x / d;
        const int k = d >= 256 ? kb + (int)threadIdx.x : (int)threadIdx.x % d;
        const bool live = g < G && k < d;
        // sixteen loads in flight per thread (a slice is a few such batches: the kernel is a chain of memory round trips)
        double a[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) a[x] = 0.0;
        if (live) {
            int r = r0 + g;
            for (; r + 15 * G < r1; r += 16 * G) {
                float t[16];
#pragma unroll
                for (int x = 0; x < 16; ++x) t[x] = T[(int64_t)(r + x * G) * d + k];
#pragma unroll
                for (int x = 0; x < 16; ++x) a[x] += (double)t[x];
            }
            for (; r + 3 * G < r1; r += 4 * G) {
                const float t0 = T[(int64_t)r * d + k], t1 = T[(int64_t)(r + G) * d + k];
                const float t2 = T[(int64_t)(r + 2 * G) * d + k], t3 = T[(int64_t)(r + 3 * G) * d + k];
                a[0] += (double)t0; a[1] += (double)t1; a[2] += (double)t2; a[3] += (double)t3;
            }
            for (; r < r1; r += G) a[0] += (double)T[(int64_t)r * d + k];
        }
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
            for (int x = 0; x < w; ++x) a[x] += a[x + w];
        red[threadIdx.x] = a[0];
        __syncthreads();
        if (live && g == 0) {
            double acc = 0.0;
            for (int gg = 0; gg < G; ++gg) acc += red[gg * d + k - kb];
            out[k] = acc;
        }
        __syncthreads();
    }
}

// bar[table][k] = (sum over slices) / rows, stored fp32 (large tables only; small ones do this inside
// centre_vectors_kernel).  Same summation order as there.
__global__ __launch_bounds__(256) void colsum_final_kernel(const double *__restrict__ part, int n, int m, int d,
                                                           float *__restrict__ bar)
{
    // 64 columns per workgroup, four threads per column: thread q sums the slices q, q+4, ... (plus the tail on q = 0),
    // then (a0 + a1) + (a2 + a3): the order centre_vectors_kernel uses when it does this itself
    __shared__ double red[4][64];
    const int tab = blockIdx.y, q = threadIdx.x >> 6, k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rows = tab ? m : n, nsl = slices_for(rows), full = nsl & ~3;
    double a = 0.0;
    if (k < d) {
        for (int sl = q; sl < full; sl += 4) a += part[((size_t)tab * kSlices + sl) * d + k];
        if (q == 0)
            for (int sl = full; sl < nsl; ++sl) a += part[((size_t)tab * kSlices + sl) * d + k];
    }
    red[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && k < d) {
        const int c = threadIdx.x & 63;
        bar[(size_t)tab * d + k] = (float)(((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) / (double)rows);
    }
}

// rm[r] = U[r].vbar (rows 0..n-1), cm[c] = ubar.V[c] (rows n..n+m-1); f64 accumulate, stored fp32 (the reference
// works on fp32 tensors).  A workgroup first forms ubar, vbar = (sum over slices) / rows from the partial column sums
// (fixed order; every workgroup repeats this small reduction instead of a launch of its own), then takes
// `rows_per_block` consecutive rows of [U; V], one wave per row.
// vsplit != nullptr (split-product form of the main kernel, d a multiple of 16): every V row is also written as its
// two-term bf16 expansion v = hi + mid (+ 2^-16 |v|), hi = bf16(v), mid = bf16(v - hi), in the FRAGMENT-MAJOR layout the
// main kernel's MFMA operand reads want: [32-row tile][16-wide k block][part hi / mid][lane = (k / 8 % 2) * 32 + row % 32]
// [8 bf16] — one wave-level 16-byte read per (tile, k block, part) is one contiguous KiB.
__global__ __launch_bounds__(256) void centre_vectors_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                             const double *__restrict__ part,
                                                             const float *__restrict__ bar_pre, int n, int m, int d,
                                                             int rows_per_block, float *__restrict__ rm,
                                                             float *__restrict__ cm, mfcd_bf16 *__restrict__ vsplit)
{
    extern __shared__ __attribute__((aligned(16))) float bar[];   // [2][d]: ubar, vbar (+ [4][2 d] doubles behind it, small tables)
    if (bar_pre) {   // large tables: colsum_final_kernel has reduced the partial sums once
        for (int idx = threadIdx.x; idx < 2 * d; idx += 256) bar[idx] = bar_pre[idx];
    } else {
        // same order in every workgroup and as colsum_final_kernel: ((s0+s4+..) + (s1+s5+..)) + ((s2+..) + (s3+..)), the tail
        // slices on the first.  One work item per (column, quarter): its loads are independent and issued eight at a
        // time — the prologue of every workgroup of this kernel used to be a chain of ~nsl/4 memory round trips
        double *quart = reinterpret_cast<double *>(bar + 2 * d + (2 * d & 1));   // [4][2 d]
        for (int item = threadIdx.x; item < 8 * d; item += 256) {
            const int q = item / (2 * d), idx = item - q * 2 * d;
            const int tab = idx / d, k = idx - tab * d;
            const int rows = tab ? m : n, nsl = slices_for(rows), full = nsl & ~3;
            const double *src = part + (size_t)tab * kSlices * d + k;
            double a = 0.0;
            int sl = q;
            for (; sl + 28 < full; sl += 32) {
                double t[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) t[x] = src[(size_t)(sl + 4 * x) * d];
#pragma unroll
                for (int x = 0; x < 8; ++x) a += t[x];
            }
            for (; sl < full; sl += 4) a += src[(size_t)sl * d];
            if (q == 0)
                for (sl = full; sl < nsl; ++sl) a += src[(size_t)sl * d];
            quart[q * 2 * d + idx] = a;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 2 * d; idx += 256) {
            const int rows = idx >= d ? m : n;
            bar[idx] = (float)(((quart[idx] + quart[2 * d + idx]) + (quart[4 * d + idx] + quart[6 * d + idx])) / (double)rows);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t w1 = w0 + rows_per_block < (int64_t)n + m ? w0 + rows_per_block : (int64_t)n + m;
    for (int64_t w = w0 + wave; w < w1; w += 4) {
        const bool isV = w >= n;
        const float *row = isV ? V + (w - n) * d : U + w * d;
        const float *other = isV ? bar : bar + d;  // V rows pair with ubar (bar[0]), U rows with vbar (bar[1])
        double acc = 0.0;
        for (int k = lane; k < d; k += MFCD_WAVE) acc += (double)row[k] * (double)other[k];
        if (isV && vsplit) {
            const int64_t c = w - n;
            const int64_t tile_base = (c >> 5) * (int64_t)(d >> 4);        // (tile, k block) index of k block 0
            const int l31 = (int)(c & 31);
            for (int k = lane; k < d; k += MFCD_WAVE) {
                const float x = row[k];
                const mfcd_bf16 hi = (mfcd_bf16)x;
                const mfcd_bf16 mid = (mfcd_bf16)(x - (float)hi);
                const int64_t chunk = ((tile_base + (k >> 4)) * 2) * 64 + ((k >> 3) & 1) * 32 + l31;   // part 0
                vsplit[chunk * 8 + (k & 7)] = hi;
                vsplit[(chunk + 64) * 8 + (k & 7)] = mid;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, MFCD_WAVE);
        if (lane == 0) {
            if (isV) cm[w - n] = (float)acc;
            else rm[w] = (float)acc;
        }
    }
}

// One wave per row of X, ONE sweep: xm[r] = fp32 mean, sxx[r] = sum x^2, scc[r] = sum (x - xm)^2 formed in f64 as
// sum x^2 - 2 xm sum x + m xm^2 with the fp32-rounded mean the reference centres with (structure.py:987).
// XV: rows are 16-byte aligned (m % 4 == 0): 16-byte loads, four per lane in flight.
template <bool XV>
__global__ __launch_bounds__(256) void x_rows_kernel(const float *__restrict__ X, int n, int m, float *__restrict__ xm,
                                                     double *__restrict__ scc, double *__restrict__ sxx)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float *row = X + r * m;
    double s1 = 0.0, s2 = 0.0;
    if constexpr (XV) {
        const float4 *row4 = reinterpret_cast<const float4 *>(row);
        const int n4 = m >> 2;
        int c = lane;
        for (; c + 192 < n4; c += 256) {
            const float4 t0 = row4[c], t1 = row4[c + 64], t2 = row4[c + 128], t3 = row4[c + 192];
            const float v[16] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w,
                                 t2.x, t2.y, t2.z, t2.w, t3.x, t3.y, t3.z, t3.w};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double x = (double)v[q];
                s1 += x;
                s2 += x * x;
            }
        }
        for (; c < n4; c += 64) {
            const float4 t = row4[c];
            const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double x = (double)v[q];
                s1 += x;
                s2 += x * x;
            }
        }
    } else {
        for (int c = lane; c < m; c += MFCD_WAVE) {
            const double x = (double)row[c];
            s1 += x;
            s2 += x * x;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, MFCD_WAVE);
        s2 += __shfl_xor(s2, off, MFCD_WAVE);
    }
    if (lane == 0) {
        const float mean = (float)(s1 / (double)m);
        const double mu = (double)mean;
        xm[r] = mean;
        scc[r] = fmax(0.0, s2 - 2.0 * mu * s1 + (double)m * mu * mu);
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

// Tiled form of the main kernel for D in {32, 64, 128, 256}.
//
// A workgroup of NW waves owns NW*32 rows of U; the columns of its split arrive as TC-column stages of V in LDS,
// once per WORKGROUP, written by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass),
// double-buffered, one barrier per stage.  A stage is the plain [TC][D] array cut into 1-KiB pieces (one wave-level
// DMA instruction each: the source of piece P is simply the P-th KiB of the stage's V rows) with 16 bytes of padding
// after every piece, which spreads the per-lane 16-byte fragment reads (lane = column, stride = one row) over the
// banks: conflict-free at D = 256 (one row per piece), 1024/(4D)-way below, where LDS time does not matter.  A lane's
// fragment chunks are consecutive, so all reads of a tile share one address register (immediate offsets).
//
// Each wave computes the TRANSPOSED tile: V rows are the MFMA's A operand, the wave's 32 U rows (registers, for
// the whole sweep) its B operand, so that in the accumulator layout a lane holds ONE row of U V^T and 16 of its
// columns (4 runs of 4 consecutive columns).  Per-row sums are then per-lane scalars (no 16-wide f64 accumulator
// arrays) and the matching X values are four 16-byte loads per lane.  Inside a tile the epilogue sums its 16
// terms in fp32 (packed two-wide), across tiles in f64.
//
// The f32 MFMA runs on the SIMD's vector lanes: other waves' VALU work does not hide under it (in-kernel stamps:
// an epilogue beside another wave's chain takes 10x its issue time), so what counts is the number of non-MFMA
// vector instructions per tile; hence the immediate-offset addressing and the packed epilogue.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_dma16(const float *src, float *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma4(const float *src, float *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ unsigned lds_addr(const float *p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
}
// 16-byte LDS read issued without the compiler's wait bookkeeping (immediate byte offset); pair with lds_wait4.
template <int OFF>
__device__ __forceinline__ void lds_read16_issue(f32x4 &dst, unsigned byte_addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(byte_addr), "n"(OFF) : "memory");
}
// the QB reads of batch `bi` (compile-time after unrolling): byte offsets 16*(QB*bi + q)
template <int QB>
__device__ __forceinline__ void lds_read16_next(f32x4 (&t)[QB], unsigned abase, int bi)
{
    static_assert(QB == 4, "batch of four reads");
    switch (bi) {   // folded: bi is a constant in the unrolled caller
#define MFCD_B(B) case B: lds_read16_issue<64 * B>(t[0], abase); lds_read16_issue<64 * B + 16>(t[1], abase); \
                  lds_read16_issue<64 * B + 32>(t[2], abase); lds_read16_issue<64 * B + 48>(t[3], abase); break;
        MFCD_B(1) MFCD_B(2) MFCD_B(3) MFCD_B(4) MFCD_B(5) MFCD_B(6) MFCD_B(7)
#undef MFCD_B
    default: break;
    }
}
// wait until at most N LDS operations are outstanding; the operands tie the consumers to the wait
template <int N>
__device__ __forceinline__ void lds_wait4(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

__device__ __forceinline__ void split_read_batch(f32x4 (&t)[4], unsigned abase, int bi)
{
    switch (bi) {   // folded: bi is a constant in the unrolled caller
#define MFCD_B(B) case B: lds_read16_issue<(4 * B + 0) * 1024>(t[0], abase); lds_read16_issue<(4 * B + 1) * 1024>(t[1], abase); \
                  lds_read16_issue<(4 * B + 2) * 1024>(t[2], abase); lds_read16_issue<(4 * B + 3) * 1024>(t[3], abase); break;
        MFCD_B(0) MFCD_B(1) MFCD_B(2) MFCD_B(3) MFCD_B(4) MFCD_B(5) MFCD_B(6) MFCD_B(7)
#undef MFCD_B
    default: break;
    }
}

// PFX: fetch a tile's X values one tile ahead (needs ~45 more registers: two register sets, two copies of the tile
// body).  On for d <= 128, where an HBM round trip outlasts a tile's MFMA chain; off for d = 256, whose 128-MFMA chain
// covers it and whose 128-register operand leaves no room.
// WHAT: 1 = per-row sums only (compute_alpha_and_norm_ratios never reads the global error), 2 = global error sum only
// (compute_reconstruction_error never reads the rows), 3 = both.  The epilogue's vector work runs on the lanes the fp32
// MFMA uses, so what the caller does not need is not computed: 7 / 3 / 10 packed operations per pair of outputs.
// SPLIT: the product runs on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the rate of the fp32 MFMA) as THREE
// bf16 products with fp32 accumulation: u = uh + um, v = vh + vm (two-term bf16 expansions, 16 significant bits each),
// u v ~ vm uh + vh um + vh uh; the dropped terms are <= 2^-16 |u v| each, the measured error of a row sum of U V^T is
// 3-4e-6 relative (tests hold 2e-5; the reference's own fp32 GEMM is at 1e-7).  `V` then points at the split table in
// the fragment-major layout centre_vectors_kernel writes (same bytes per row, same stage and piece arithmetic; a
// wave's fragment read is one contiguous KiB, so the stage needs no padding), and the wave splits its own U rows once.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// What the tiled kernel's TAIL needs (round 3: the final reduction is no launch of its own).  The workgroups of a row
// block (one per column split) count themselves in at cnt[1 + row block] once their partial sums are written; the one
// that arrives last re-centres the block's rows in f64 (the arithmetic of the former uvt_final_tiled_kernel) and writes
// the block's share of the two global sums; row blocks count themselves in at cnt[0], and the last one adds the shares
// in block order.  Every sum has a fixed order, whichever workgroup happens to form it: results do not depend on timing.
// Partial sums travel between workgroups (possibly of different XCDs, whose L2s are not coherent with each other) as
// agent-scope stores and loads: they go to the memory side themselves, so no release / acquire FENCE is needed — a fence
// at agent scope writes back and invalidates the XCD's whole L2 on this chip, and one per finishing workgroup threw the
// V rows the other workgroups were reading out of it (the pass took twice as long).
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct UvtTail {
    double *row_stats;   // [n][8] (WHAT & 1)
    double *blk;         // [row blocks][2]
    double *scal;        // [4] (WHAT & 2)
    unsigned *cnt;       // [1 + row blocks], zero when the kernel starts (colsum_partial_kernel)
    double s;
};

template <int D, int NW, int TC, bool XV, int WPE = 2, bool PFX = (D <= 128), int WHAT = 3, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE)))
void uvt_tiled_kernel(const float *__restrict__ U, const float *__restrict__ V, const float *__restrict__ X,
                      const float *__restrict__ rm, const float *__restrict__ cm, int n,
                      int m, float s, int cols_per_split, int splits, int row_blocks, double *part_rows,
                      double *part_err, double *part_xx, UvtTail tail)
{
    constexpr int CPR = D / 4, PIECES = TC * D / 256, PPW = PIECES / NW, CMW = (TC + 63) / 64, PF = SPLIT ? 256 : 260;
    static_assert(!SPLIT || (D % 32 == 0 && D <= 256 && PFX && XV), "split-product form: d in {32, 64, 128, 256}, prefetch form");
    static_assert(PIECES % NW == 0 && TC % 32 == 0 && CMW <= NW, "stage must split evenly over the waves");
    __shared__ __attribute__((aligned(16))) float vts[2][PIECES * PF];   // [buffer][piece][256 + 4 pad]
    __shared__ __attribute__((aligned(16))) float cmss[2][CMW * 64];      // column means of U V^T, stage's columns
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int split, rb;
    if (splits >= 8) {   // workgroup ids go round-robin to the 8 XCDs: keep a split's V rows in one XCD's L2
        const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
        split = (w / row_blocks) * 8 + xcd;
        rb = w % row_blocks;
    } else {
        split = blockIdx.x / row_blocks;
        rb = blockIdx.x % row_blocks;
    }
    if (split >= splits) return;   // whole workgroup
    const int rt = rb * NW + wave, row0 = rt * 32;
    const bool active = row0 < n;  // waves past the last row still issue their share of the stages
    const int c_begin = split * cols_per_split;
    const int c_end = min(m, c_begin + cols_per_split);
    const int myrow = min(row0 + l31, n - 1);
    const bool rowok = row0 + l31 < n;

    // issue one stage: PPW 1-KiB pieces per wave (piece P = floats [256 P, 256 P + 256) of the stage's V rows; the host
    // guarantees m*D < 2^31) + the stage's TC column means
    const int vlast = m * D - 4;
    const int dma0 = (wave * PPW * 64 + lane) * 4;
    auto issue_stage = [&](int c0, float *vt, float *cms) {
        const int base = c0 * D + dma0;
        if (SPLIT || c0 + TC <= m) {      // (the split table is padded to whole stages)
            const float *vb = V + base;   // one 64-bit address per stage, the pieces are 1 KiB apart
#pragma unroll
            for (int i = 0; i < PPW; ++i) lds_dma16(vb + i * 256, vt + (wave * PPW + i) * PF);
        } else {   // ragged last stage: rows past m are never used, read something in range instead
#pragma unroll
            for (int i = 0; i < PPW; ++i) lds_dma16(V + min(base + i * 256, vlast), vt + (wave * PPW + i) * PF);
        }
        if (wave < CMW) {
            const int c = wave * 64 + lane;
            if (c < TC) lds_dma4(cm + min(c0 + c, m - 1), cms + wave * 64);
        }
    };
    issue_stage(c_begin, vts[0], cmss[0]);

    float u[SPLIT ? 1 : D / 2];   // B operand: this lane's half of its U row (k index permuted: lane half h, step kk -> h*D/2+kk)
    bf16x8 uh[SPLIT ? D / 16 : 1], um[SPLIT ? D / 16 : 1];   // split form: k block kb, elements k = 16 kb + 8 half + 0..7
    if constexpr (SPLIT) {
#pragma unroll
        for (int kb = 0; kb < D / 16; ++kb) {
            const float *up = U + (int64_t)myrow * D + kb * 16 + half * 8;
            const float4 t0 = *reinterpret_cast<const float4 *>(up), t1 = *reinterpret_cast<const float4 *>(up + 4);
            const float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const mfcd_bf16 hi = (mfcd_bf16)x[e];
                uh[kb][e] = hi;
                um[kb][e] = (mfcd_bf16)(x[e] - (float)hi);
            }
        }
        u[0] = 0.0f;
    } else {
        const float *urow = U + (int64_t)myrow * D + half * (D / 2);
#pragma unroll
        for (int q = 0; q < D / 8; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(urow + 4 * q);
            u[4 * q + 0] = t.x; u[4 * q + 1] = t.y; u[4 * q + 2] = t.z; u[4 * q + 3] = t.w;
        }
    }
    // The row statistics of X (mean, sum x^2, centred sum of squares) come out of THIS sweep (round 1 read X a second
    // time for them): every term is taken relative to x0 = the row's first value in this split, so that the fp32
    // per-tile sums do not cancel, and the tail of this kernel re-centres in f64.
    float rmv = rm[myrow], x0v = X[(int64_t)myrow * m + c_begin];
    // every load so far is consumed HERE, before the loop: left pending, the wait for it would sit in front of the
    // first MFMA of every iteration (and, with an LDS-DMA in flight, be a full vmcnt(0))
    if constexpr (SPLIT) {
#pragma unroll
        for (int kb = 0; kb < D / 16; ++kb) asm volatile("" : "+v"(uh[kb]), "+v"(um[kb]));
    } else {
#pragma unroll
        for (int k = 0; k < D / 2; ++k) asm volatile("" : "+v"(u[k]));
    }
    asm volatile("" : "+v"(rmv), "+v"(x0v));
    const float *xrow0 = X + (int64_t)myrow * m;   // + cb + 8g + 4*half: this lane's 16-byte pieces of a tile
    // fragment address of tile j of a stage: row 32j + l31, chunks half*CPR/2 + q (one piece, consecutive)
    const int f0 = l31 * CPR + half * (CPR / 2);
    const unsigned frag0 = (unsigned)(((f0 >> 6) * PF + (f0 & 63) * 4) * 4);          // bytes, tile 0
    constexpr unsigned frag_step = (unsigned)((32 * CPR / 64) * PF * 4);              // bytes per tile (32 rows)
    double sac = 0.0, saa = 0.0, err2 = 0.0, ssa = 0.0, ssc = 0.0, sscc = 0.0;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#if MFCD_UVT_STAMPS
    unsigned long long cyc_chain = 0, cyc_epi = 0, cyc_sync = 0, cyc_dma = 0, n_tiles = 0;
#endif
#if MFCD_UVT_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    int buf = 0;
    // X values of this lane's row for the tile at column cb: columns cb + 8g + 4*half + {0,1,2,3}, g = 0..3.
    // They are fetched ONE TILE AHEAD (round 1 issued them at the top of their own tile): an HBM round trip is ~2 us,
    // i.e. longer than the 64-MFMA chain of a d = 128 tile, so the epilogue used to stall on them (C3 58 % vs C5 81 %
    // in-kernel).  The two register sets swap roles every tile (no copy: copying a pending load would wait for it).
    auto load_x = [&](int cb, f32x4 (&xq)[4]) {
        // branch-free (a branch here makes the values phi nodes, which the compiler resolves with register copies of
        // loads still in flight, i.e. with a wait): 16-byte pieces clamped to stay inside the row; columns at or past
        // the split's end are masked in the epilogue
        if constexpr (XV) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#if MFCD_UVT_EXP == 1
                xq[g] = f32x4{(float)cb, 1.f, 2.f, 3.f};
#else
                xq[g] = *reinterpret_cast<const f32x4 *>(xrow0 + min(cb + 8 * g + 4 * half, m - 4));
#endif
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) xq[r >> 2][r & 3] = xrow0[min(cb + tile_row(r, half), m - 1)];
        }
    };
    static_assert(!PFX || XV, "the prefetch form is built for 16-byte X pieces (four loads per request)");
    constexpr int kPend = 4;   // vector-memory operations one X prefetch puts in flight
    f32x4 xa[4], xb[4];
    if (PFX && active) load_x(c_begin, xa);

    auto tile = [&](int cb, int ncb, int j, const float *cms, f32x4 (&xq)[4], f32x4 (&xn)[4]) __attribute__((always_inline)) {
                const bool full = cb + 32 <= c_end;   // wave-uniform
                MFCD_STAMP(tt0);
                asm volatile("" ::: "memory");   // the stage's LDS-DMA (issued above) stays OLDER than this prefetch
                if constexpr (PFX) load_x(ncb, xn);   // the NEXT tile's values (a dummy re-read of this tile at the very end)
                else load_x(cb, xq);                 // this tile's own values, in flight under its MFMA chain
                asm volatile("" ::: "memory");
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                // A fragments (V) from LDS in batches of QB 16-byte reads, one batch ahead of the MFMAs that consume
                // them.  The reads are issued from inline asm with counted lgkmcnt waits: for a compiler-visible
                // ds_read hipcc (ROCm 7.2) first drains EVERY outstanding LDS-DMA (s_waitcnt vmcnt(0)), i.e. the next
                // stage issued just above, and the DMA would never overlap the MFMA chain.  LDS returns in order, so
                // "at most QB operations outstanding" means the older batch has landed (extra operations the compiler
                // may have in flight only make the wait stricter).
                const unsigned cmaddr = lds_addr(cms + 32 * j + 4 * half);
                f32x4 cq[4];
                if constexpr (SPLIT) {
                    // fragments of tile j: KiB number ((j * D/16 + kb) * 2 + part); this lane's 16 bytes at lane * 16
                    constexpr int NKB = D / 16, NBS = NKB / 2;
                    const unsigned abase = lds_addr(vts[buf]) + (unsigned)j * (unsigned)(NKB * 2 * 1024) + (unsigned)lane * 16u;
                    f32x4 t[2][4];
                    split_read_batch(t[0], abase, 0);
#pragma unroll
                    for (int bi = 0; bi < NBS; ++bi) {
                        if (bi + 1 < NBS) {
                            split_read_batch(t[(bi + 1) & 1], abase, bi + 1);
                            lds_wait4<4>(t[bi & 1][0], t[bi & 1][1], t[bi & 1][2], t[bi & 1][3]);
                        } else {
                            lds_wait4<0>(t[bi & 1][0], t[bi & 1][1], t[bi & 1][2], t[bi & 1][3]);
                            if constexpr ((WHAT & 2) != 0) {
                                lds_read16_issue<0>(cq[0], cmaddr);
                                lds_read16_issue<32>(cq[1], cmaddr);
                                lds_read16_issue<64>(cq[2], cmaddr);
                                lds_read16_issue<96>(cq[3], cmaddr);
                            }
                        }
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const int kb = 2 * bi + kk;
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, t[bi & 1][2 * kk]);
                            const bf16x8 am = __builtin_bit_cast(bf16x8, t[bi & 1][2 * kk + 1]);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, uh[kb], acc, 0, 0, 0);   // small terms first
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, um[kb], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, uh[kb], acc, 0, 0, 0);
                        }
                    }
                } else {
                constexpr int QB = 4, NB = D / 8 / QB;
                const unsigned abase = lds_addr(vts[buf]) + frag0 + (unsigned)j * frag_step;
                f32x4 t[2][QB];
                lds_read16_issue<0>(t[0][0], abase);
                lds_read16_issue<16>(t[0][1], abase);
                lds_read16_issue<32>(t[0][2], abase);
                lds_read16_issue<48>(t[0][3], abase);
#pragma unroll
                for (int bi = 0; bi < NB; ++bi) {
                    if (bi + 1 < NB) {
                        lds_read16_next<QB>(t[(bi + 1) & 1], abase, bi + 1);
                        lds_wait4<QB>(t[bi & 1][0], t[bi & 1][1], t[bi & 1][2], t[bi & 1][3]);
                    } else {
                        lds_wait4<0>(t[bi & 1][0], t[bi & 1][1], t[bi & 1][2], t[bi & 1][3]);
                        // the tile's 16 column means, under the last 16 MFMAs (inline asm like the fragment reads: a
                        // compiler-visible ds_read would drain every outstanding vector-memory operation first, i.e.
                        // the X values just requested for the NEXT tile)
                        if constexpr ((WHAT & 2) != 0) {
                            lds_read16_issue<0>(cq[0], cmaddr);
                            lds_read16_issue<32>(cq[1], cmaddr);
                            lds_read16_issue<64>(cq[2], cmaddr);
                            lds_read16_issue<96>(cq[3], cmaddr);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < QB; ++q) {
                        const f32x4 tq = t[bi & 1][q];
                        const int k0 = 4 * (bi * QB + q);
#if MFCD_UVT_EXP == 3
                        acc[0] += u[k0] * tq.x + u[k0 + 1] * tq.y + u[k0 + 2] * tq.z + u[k0 + 3] * tq.w;
                        continue;
#endif
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tq.x, u[k0 + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tq.y, u[k0 + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tq.z, u[k0 + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tq.w, u[k0 + 3], acc, 0, 0, 0);
                    }
                }
                }
                MFCD_STAMP(tt1);
                // the X values become visible to the epilogue arithmetic only here: otherwise the scheduler moves
                // x - xm, s*x up into the MFMA chain and with them the wait for the X loads (and, as an LDS-DMA is in
                // flight, for everything: vmcnt(0)) to the top of the chain
                // (tied to the accumulator so that it stays behind the last MFMA)
                // (with PFX the values were requested one tile ago and are compiler-visible loads: the wait the compiler
                // puts here covers them; hiding them in inline asm with hand-counted waits was tried and was NOT safe —
                // the register allocator may copy a register it believes ready while the load is still in flight)
                asm volatile("" : "+v"(acc), "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]));
                if constexpr ((WHAT & 2) != 0) lds_wait4<0>(cq[0], cq[1], cq[2], cq[3]);
                else cq[0] = cq[1] = cq[2] = cq[3] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                // acc[4g+e] = (U V^T)[myrow][cb + 8g + 4*half + e]; cq[g][e] = the matching column mean
                float pac, paa, pe, psa, psc, pscc;
                if (full) {   // two terms per instruction (v_pk_*_f32)
                    const f32x2 rm2 = {rmv, rmv}, x02 = {x0v, x0v}, s2 = {s, s};
                    f32x2 pac2 = {0.0f, 0.0f}, paa2 = {0.0f, 0.0f}, pe2 = {0.0f, 0.0f};
                    f32x2 psa2 = {0.0f, 0.0f}, psc2 = {0.0f, 0.0f}, pscc2 = {0.0f, 0.0f};
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 c4 = cq[g];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x2 g2 = {acc[4 * g + 2 * h], acc[4 * g + 2 * h + 1]};
                            const f32x2 x2 = {xq[g][2 * h], xq[g][2 * h + 1]};
                            const f32x2 c2 = {c4[2 * h], c4[2 * h + 1]};
                            const f32x2 av = g2 - rm2;                         // structure.py:985
                            const f32x2 cv = x2 - x02;                         // structure.py:987, up to the shift x0 - xm
                            const f32x2 ev = (g2 - c2) - s2 * x2;              // structure.py:943, 949
#if MFCD_UVT_EXP == 2
                            pe2 += av + cv + ev;
                            continue;
#endif
                            if constexpr ((WHAT & 1) != 0) {
                                pac2 = __builtin_elementwise_fma(av, cv, pac2);
                                paa2 = __builtin_elementwise_fma(av, av, paa2);
                                psa2 += av;
                                psc2 += cv;
                                pscc2 = __builtin_elementwise_fma(cv, cv, pscc2);
                            }
                            if constexpr ((WHAT & 2) != 0) pe2 = __builtin_elementwise_fma(ev, ev, pe2);
                            if constexpr (WHAT == 2) pscc2 = __builtin_elementwise_fma(x2, x2, pscc2);   // sum x^2 for ||sX||
                        }
                    }
                    pac = pac2.x + pac2.y;
                    paa = paa2.x + paa2.y;
                    pe = pe2.x + pe2.y;
                    psa = psa2.x + psa2.y;
                    psc = psc2.x + psc2.y;
                    pscc = pscc2.x + pscc2.y;
                } else {      // ragged last tile of the split: per-term column masks
                    pac = paa = pe = psa = psc = pscc = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool ok = cb + tile_row(r, half) < c_end;
                        const float gv = acc[r], xv = xq[r >> 2][r & 3];
                        const float cmv = cq[r >> 2][r & 3];
                        const float av = ok ? gv - rmv : 0.0f;
                        const float cv = ok ? xv - x0v : 0.0f;
                        const float ev = ok ? (gv - cmv) - s * xv : 0.0f;
                        if constexpr ((WHAT & 1) != 0) {
                            pac = fmaf(av, cv, pac);
                            paa = fmaf(av, av, paa);
                            psa += av;
                            psc += cv;
                            pscc = fmaf(cv, cv, pscc);
                        }
                        if constexpr ((WHAT & 2) != 0) pe = fmaf(ev, ev, pe);
                        if constexpr (WHAT == 2) pscc = ok ? fmaf(xv, xv, pscc) : pscc;
                    }
                }
                sac += (double)pac;
                saa += (double)paa;
                err2 += (double)pe;
                ssa += (double)psa;
                ssc += (double)psc;
                sscc += (double)pscc;
#if MFCD_UVT_STAMPS
                asm volatile("" : "+v"(sac), "+v"(saa), "+v"(err2));
                MFCD_STAMP(tt2);
                cyc_chain += tt1 - tt0;
                cyc_epi += tt2 - tt1;
                n_tiles += 1;
#endif
    };

    if constexpr (PFX && TC == 32) {
        // ONE tile per stage (d = 256): the two X register sets alternate from stage to stage (the loop is unrolled by
        // two so that their roles stay fixed: no copy of a register whose load is still in flight)
        auto one_stage = [&](int c0, f32x4 (&xq)[4], f32x4 (&xn)[4]) __attribute__((always_inline)) {
            const float *cms = cmss[buf];
            if (c0 + TC < c_end) issue_stage(c0 + TC, vts[buf ^ 1], cmss[buf ^ 1]);
            if (active) tile(c0, c0 + TC < c_end ? c0 + TC : c0, 0, cms, xq, xn);
            if (active) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPend) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            buf ^= 1;
        };
#pragma unroll 1
        for (int c0 = c_begin; c0 < c_end; c0 += 2 * TC) {
            one_stage(c0, xa, xb);
            if (c0 + TC >= c_end) break;
            one_stage(c0 + TC, xb, xa);
        }
    } else {
    for (int c0 = c_begin; c0 < c_end; c0 += TC) {
        const float *cms = cmss[buf];
        MFCD_STAMP(td0);
        if (c0 + TC < c_end) issue_stage(c0 + TC, vts[buf ^ 1], cmss[buf ^ 1]);
        MFCD_STAMP(td1);
#if MFCD_UVT_STAMPS
        cyc_dma += td1 - td0;
#endif
        if (active) {
            if constexpr (PFX) {
                // tiles in PAIRS, straight-line: the first uses xa and requests the second's values into xb, the second
                // uses xb and requests the next pair's (or the next stage's first tile's) into xa — fixed roles, no copy
                static_assert((TC / 32) % 2 == 0 || TC == 32, "the X prefetch ping-pong needs an even number of tiles per stage");
#pragma unroll 1
                for (int j = 0; j < TC / 32; j += 2) {
                    const int cbA = c0 + 32 * j, cbB = cbA + 32;
                    if (cbA >= c_end) break;
                    const bool hasB = cbB < c_end;
                    tile(cbA, hasB ? cbB : cbA, j, cms, xa, xb);
                    if (hasB) {
                        const int after = (j + 2 < TC / 32 && cbB + 32 < c_end) ? cbB + 32 : (c0 + TC < c_end ? c0 + TC : cbB);
                        tile(cbB, after, j + 1, cms, xb, xa);
                    }
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < TC / 32; ++j) {
                    const int cb = c0 + 32 * j;
                    if (cb >= c_end) continue;
                    tile(cb, cb, j, cms, xa, xb);
                }
            }
        }
        MFCD_STAMP(ts0);
        // this wave's pieces of the next stage have landed: everything but the X prefetch of the stage's last tile,
        // which is younger than the DMA (vmcnt counts in issue order) and stays in flight across the barrier
        if (PFX && active) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPend) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... and everybody's; this buffer may be overwritten
        MFCD_STAMP(ts1);
#if MFCD_UVT_STAMPS
        cyc_sync += ts1 - ts0;
#endif
        buf ^= 1;
    }
    }
#if MFCD_UVT_STAMPS
    if (lane == 0 && active) {
        atomicAdd(&mfcd_uvt_dbg[0], cyc_chain);
        atomicAdd(&mfcd_uvt_dbg[1], cyc_epi);
        atomicAdd(&mfcd_uvt_dbg[2], cyc_sync);
        atomicAdd(&mfcd_uvt_dbg[3], cyc_dma);
        atomicAdd(&mfcd_uvt_dbg[4], n_tiles);
        atomicAdd(&mfcd_uvt_dbg[5], 1ull);
        atomicAdd(&mfcd_uvt_dbg[6], (unsigned long long)__builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(&mfcd_uvt_dbg[7], (unsigned long long)__builtin_amdgcn_s_memrealtime() - rt0);
    }
#endif
    if (active) {
    // a row's columns are split over the two lane halves
    sac += __shfl_xor(sac, 32, MFCD_WAVE);
    saa += __shfl_xor(saa, 32, MFCD_WAVE);
    ssa += __shfl_xor(ssa, 32, MFCD_WAVE);
    ssc += __shfl_xor(ssc, 32, MFCD_WAVE);
    sscc += __shfl_xor(sscc, 32, MFCD_WAVE);
    if (!rowok) err2 = 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off, MFCD_WAVE);
    if ((WHAT & 1) != 0 && half == 0 && rowok) {   // kTiledRowSums doubles per (split, row): sum a c', sum a a, sum a, sum c', sum c'c', x0
        double *o = part_rows + ((size_t)split * n + row0 + l31) * kTiledRowSums;
        st_agent(o + 0, sac);
        st_agent(o + 1, saa);
        st_agent(o + 2, ssa);
        st_agent(o + 3, ssc);
        st_agent(o + 4, sscc);
        st_agent(o + 5, (double)x0v);
    }
    if ((WHAT & 2) != 0 && lane == 0) st_agent(part_err + (size_t)split * ((n + 31) / 32) + rt, err2);
    if constexpr (WHAT == 2) {   // error-only pass: sum x^2 of the wave's rows (the other forms get it from the row sums)
        if (!rowok) sscc = 0.0;       // (the two lane halves of a row were combined above)
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sscc += __shfl_xor(sscc, off, MFCD_WAVE);
        if (lane == 0) st_agent(part_xx + (size_t)split * ((n + 31) / 32) + rt, sscc);
    }
    }

    // ---------------- tail: the last workgroup of a row block finishes its rows (see UvtTail) ----------------
    // The tiled kernel's per-(split, row) sums are relative to the split's shift x0 (and to nothing for U V^T): they are
    // re-centred in f64.  Per split s with n_s columns and c' = x - x0_s:  sum x = S_c' + n_s x0;
    // sum x^2 = S_c'c' + 2 x0 S_c' + n_s x0^2;  with the fp32-rounded row mean xm the reference centres with
    // (structure.py:987) and t = xm - x0_s:  sum (x - xm)^2 = S_c'c' - 2 t S_c' + n_s t^2,  sum a (x - xm) = S_ac' - t S_a
    if (tail.cnt == nullptr) return;       // large pass: uvt_final_tiled_kernel finishes (whole workgroup, wave-uniform)
    __shared__ int tail_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial sums have reached the memory side ...
    __syncthreads();                                   // ... and every wave's, before the workgroup counts itself in
    if (tid == 0)
        tail_last = __hip_atomic_fetch_add(&tail.cnt[1 + rb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                            (unsigned)(splits - 1) ? 1 : 0;
    __syncthreads();
    if (!tail_last) return;
    double *const tred = reinterpret_cast<double *>(&vts[0][0]);   // [2][NW * 64]; the stages are done with (barriers above)
    const int rtiles = (n + 31) / 32;
    double q = 0.0, e = 0.0;
    if constexpr ((WHAT & 1) != 0) {
        const int r = rb * NW * 32 + tid;
        if (tid < NW * 32 && r < n) {
            // ONE pass over the splits (independent loads): the sums above are polynomials in mu, so their coefficients
            // are accumulated first and mu (which needs sum x) is applied at the end, all in f64
            double sx = 0.0, s_ac = 0.0, s_a = 0.0, s_ax0 = 0.0, aa = 0.0, s_cc = 0.0, s_c = 0.0, s_cx0 = 0.0, s_n = 0.0,
                   s_nx0 = 0.0, s_nx00 = 0.0;
            for (int sp0 = 0; sp0 < splits; sp0 += 8) {    // eight splits' sums requested together, added in split order
                double tv[8][kTiledRowSums];
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const double *t = part_rows + ((size_t)min(sp0 + x, splits - 1) * n + r) * kTiledRowSums;
#pragma unroll
                    for (int y = 0; y < kTiledRowSums; ++y) tv[x][y] = ld_agent(t + y);
                }
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const int sp = sp0 + x;
                    if (sp < splits) {
                        const int c0 = sp * cols_per_split;
                        const double ns = (double)(min(m, c0 + cols_per_split) - c0);
                        const double t0 = tv[x][0], t1 = tv[x][1], t2 = tv[x][2], t3 = tv[x][3], t4 = tv[x][4], x0 = tv[x][5];
                        sx += t3 + ns * x0;
                        s_ac += t0; s_a += t2; s_ax0 += x0 * t2;
                        aa += t1;
                        s_cc += t4; s_c += t3; s_cx0 += x0 * t3;
                        s_n += ns; s_nx0 += ns * x0; s_nx00 += ns * x0 * x0;
                    }
                }
            }
            const float xmean = (float)(sx / (double)m);
            const double mu = (double)xmean;
            const double ac = s_ac - mu * s_a + s_ax0;                                           // sum a (x - mu)
            const double cc = s_cc - 2.0 * (mu * s_c - s_cx0) + (mu * mu * s_n - 2.0 * mu * s_nx0 + s_nx00);   // sum (x - mu)^2
            const double qr = s_cc + 2.0 * s_cx0 + s_nx00;                                       // sum x^2
            double *o = tail.row_stats + (size_t)r * 8;
            o[0] = ac; o[1] = aa; o[2] = fmax(0.0, cc); o[3] = (double)rm[r]; o[4] = mu;
            o[5] = qr; o[6] = 0.0; o[7] = 0.0;
            q = qr;
        }
    }
    if constexpr ((WHAT & 2) != 0) {
        // the block's share of the two global sums: its waves' error sums (and, error-only pass, their sum x^2), split-major
        for (int k = tid; k < splits * NW; k += NW * 64) {
            const int sp = k / NW, t = rb * NW + k % NW;
            if (t < rtiles) {
                e += ld_agent(part_err + (size_t)sp * rtiles + t);
                if constexpr (WHAT == 2) q += ld_agent(part_xx + (size_t)sp * rtiles + t);
            }
        }
        tred[tid] = e;
        tred[NW * 64 + tid] = q;
        __syncthreads();
        for (int w = NW * 32; w > 0; w >>= 1) {
            if (tid < w) {
                tred[tid] += tred[tid + w];
                tred[NW * 64 + tid] += tred[NW * 64 + tid + w];
            }
            __syncthreads();
        }
        if (tid == 0) {
            st_agent(tail.blk + 2 * rb + 0, tred[0]);
            st_agent(tail.blk + 2 * rb + 1, tred[NW * 64]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tail_last = __hip_atomic_fetch_add(&tail.cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                                (unsigned)(row_blocks - 1) ? 1 : 0;
        }
        __syncthreads();
        if (!tail_last) return;
        e = q = 0.0;
        for (int k = tid; k < row_blocks; k += NW * 64) {
            e += ld_agent(tail.blk + 2 * k + 0);
            q += ld_agent(tail.blk + 2 * k + 1);
        }
        __syncthreads();
        tred[tid] = e;
        tred[NW * 64 + tid] = q;
        __syncthreads();
        for (int w = NW * 32; w > 0; w >>= 1) {
            if (tid < w) {
                tred[tid] += tred[tid + w];
                tred[NW * 64 + tid] += tred[NW * 64 + tid + w];
            }
            __syncthreads();
        }
        if (tid == 0) {
            tail.scal[0] = tred[0];
            tail.scal[1] = tail.s * tail.s * tred[NW * 64];   // ||sX||_F^2  (structure.py:946)
            tail.scal[2] = 0.0;
            tail.scal[3] = 0.0;
        }
    }
}

// row_stats[r][8] from the partials, plus this block's share of the two global sums (fixed order → deterministic):
// blk[b] = {sum of part_err over the block's slice, sum of sxx over the block's rows}
__global__ __launch_bounds__(256) void uvt_final_kernel(const double *__restrict__ part_rows,
                                                        const double *__restrict__ part_err,
                                                        const float *__restrict__ rm, const float *__restrict__ xm,
                                                        const double *__restrict__ scc, const double *__restrict__ sxx,
                                                        int n, int splits, int n_err,
                                                        double *__restrict__ row_stats, double *__restrict__ blk)
{
    __shared__ double red[2][256];
    const int r = blockIdx.x * 256 + threadIdx.x;
    double q = 0.0, e = 0.0;
    if (r < n) {
        double ac = 0.0, aa = 0.0;
        for (int sp = 0; sp < splits; ++sp) {
            const double2 t = *reinterpret_cast<const double2 *>(part_rows + ((size_t)sp * n + r) * 2);
            ac += t.x;
            aa += t.y;
        }
        q = sxx[r];
        double *o = row_stats + (size_t)r * 8;
        o[0] = ac; o[1] = aa; o[2] = scc[r]; o[3] = (double)rm[r]; o[4] = (double)xm[r];
        o[5] = q; o[6] = 0.0; o[7] = 0.0;
    }
    const int per = (n_err + gridDim.x - 1) / gridDim.x;
    const int k1 = min(n_err, ((int)blockIdx.x + 1) * per);
    for (int k = blockIdx.x * per + threadIdx.x; k < k1; k += 256) e += part_err[k];
    red[0][threadIdx.x] = e;
    red[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        blk[2 * blockIdx.x + 0] = red[0][0];
        blk[2 * blockIdx.x + 1] = red[1][0];
    }
}

// The tiled kernel's per-(split, row) sums are relative to the split's shift x0 (and to nothing for U V^T): this
// kernel (LARGE passes; short ones finish in the tiled kernel's tail, same arithmetic) re-centres them in f64 and writes the same row_stats / block shares as uvt_final_kernel.  Per split s with
// n_s columns and c' = x - x0_s:  sum x = S_c' + n_s x0;  sum x^2 = S_c'c' + 2 x0 S_c' + n_s x0^2;  with the fp32-rounded
// row mean xm the reference centres with (structure.py:987) and t = xm - x0_s:
//   sum (x - xm)^2 = S_c'c' - 2 t S_c' + n_s t^2        sum a (x - xm) = S_ac' - t S_a
// scal != nullptr: ONE workgroup does all rows and writes the two global sums itself (small n: saves a launch).
__global__ __launch_bounds__(1024) void uvt_final_tiled_kernel(const double *__restrict__ part_rows,
                                                              const double *__restrict__ part_err,
                                                              const double *__restrict__ part_xx,
                                                              const float *__restrict__ rm, int n, int m, int splits,
                                                              int cols_per_split, int n_err,
                                                              double *__restrict__ row_stats, double *__restrict__ blk,
                                                              double s, double *__restrict__ scal, int what)
{
    __shared__ double red[2][1024];
    const int nthr = blockDim.x;   // 256 (one workgroup per 256 rows) or 1024 (single-workgroup form)
    double q = 0.0, e = 0.0;
    for (int r = blockIdx.x * nthr + threadIdx.x; (what & 1) && r < n; r += gridDim.x * nthr) {
        // ONE pass over the splits (independent loads): the sums above are polynomials in mu, so their coefficients are
        // accumulated first and mu (which needs sum x) is applied at the end, all in f64
        double sx = 0.0, s_ac = 0.0, s_a = 0.0, s_ax0 = 0.0, aa = 0.0, s_cc = 0.0, s_c = 0.0, s_cx0 = 0.0, s_n = 0.0,
               s_nx0 = 0.0, s_nx00 = 0.0;
        for (int sp = 0; sp < splits; ++sp) {
            const double *t = part_rows + ((size_t)sp * n + r) * kTiledRowSums;
            const int c0 = sp * cols_per_split;
            const double ns = (double)(min(m, c0 + cols_per_split) - c0);
            const double t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], x0 = t[5];
            sx += t3 + ns * x0;
            s_ac += t0; s_a += t2; s_ax0 += x0 * t2;
            aa += t1;
            s_cc += t4; s_c += t3; s_cx0 += x0 * t3;
            s_n += ns; s_nx0 += ns * x0; s_nx00 += ns * x0 * x0;
        }
        const float xmean = (float)(sx / (double)m);
        const double mu = (double)xmean;
        const double ac = s_ac - mu * s_a + s_ax0;                                           // sum a (x - mu)
        const double cc = s_cc - 2.0 * (mu * s_c - s_cx0) + (mu * mu * s_n - 2.0 * mu * s_nx0 + s_nx00);   // sum (x - mu)^2
        const double qr = s_cc + 2.0 * s_cx0 + s_nx00;                                       // sum x^2
        double *o = row_stats + (size_t)r * 8;
        o[0] = ac; o[1] = aa; o[2] = fmax(0.0, cc); o[3] = (double)rm[r]; o[4] = mu;
        o[5] = qr; o[6] = 0.0; o[7] = 0.0;
        q += qr;
    }
    const int per = (n_err + gridDim.x - 1) / gridDim.x;
    const int k1 = min(n_err, ((int)blockIdx.x + 1) * per);
    for (int k = blockIdx.x * per + threadIdx.x; (what & 2) && k < k1; k += nthr) {
        e += part_err[k];
        if (!(what & 1)) q += part_xx[k];
    }
    red[0][threadIdx.x] = e;
    red[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = nthr >> 1; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (scal) {
            scal[0] = red[0][0];
            scal[1] = s * s * red[1][0];  // ||sX||_F^2  (structure.py:946)
            scal[2] = 0.0;
            scal[3] = 0.0;
        } else {
            blk[2 * blockIdx.x + 0] = red[0][0];
            blk[2 * blockIdx.x + 1] = red[1][0];
        }
    }
}

// scal[4] from the per-block shares (one workgroup, fixed order)
__global__ __launch_bounds__(256) void uvt_scal_kernel(const double *__restrict__ blk, int nblk, double s,
                                                       double *__restrict__ scal)
{
    __shared__ double red[2][256];
    double e = 0.0, q = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) {
        e += blk[2 * k + 0];
        q += blk[2 * k + 1];
    }
    red[0][threadIdx.x] = e;
    red[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scal[0] = red[0][0];
        scal[1] = s * s * red[1][0];  // ||sX||_F^2  (structure.py:946)
        scal[2] = 0.0;
        scal[3] = 0.0;
    }
}

// k rows of UV^T (structure.py:389-392): one wave per (row, 64-column chunk)
__global__ __launch_bounds__(256) void uvt_rows_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                       const int32_t *__restrict__ row_ids, int k, int n, int m,
                                                       int d, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    (void)lane;
    if (r >= k || c >= m) return;
    const int row = row_ids[r];
    if (row < 0 || row >= n) {   // never read outside the table: a bad id shows as a row of NaN
        out[(int64_t)r * m + c] = __builtin_nanf("");
        return;
    }
    const float *ur = U + (int64_t)row * d, *vc = V + (int64_t)c * d;
    float acc = 0.0f;
    for (int q = 0; q < d; ++q) acc += ur[q] * vc[q];
    out[(int64_t)r * m + c] = acc;
}

struct UvtWs {
    mfcd_bf16 *vsplit;   // split-product form: V as two bf16 terms, fragment-major, padded to whole stages (else nullptr)
    double *colpart;  // [2][kSlices][d]
    float *bar;       // [2][d]
    float *rm, *cm, *xm;
    double *scc, *sxx, *part_rows, *part_err, *part_xx, *blk, *dummy_rows, *dummy_scal;
    unsigned *cnt;    // [1 + row tiles] arrival counters of the tiled kernel's tail (zeroed by the pass's first kernel)
    int ncnt;
    int splits, cols_per_split, n_err, nblk;
    bool tiled;
    size_t bytes;
};

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct TiledCfg {
    int NW, TC;   // waves per workgroup (32 rows each), columns per LDS stage; NW == 0: no tiled form for this d
};

TiledCfg tiled_cfg(int d)
{
    switch (d) {
    case 256: return {4, 32};    // 64 KiB of stages: two workgroups (two waves per SIMD) per CU
    case 128: return {4, 64};    // two tiles per stage: the X prefetch alternates between two register sets
    case 64: return {4, 64};
    case 32: return {4, 128};
    default: return {0, 0};
    }
}

int g_uvt_split = 1;           // mfcd_set_tuning(MFCD_TUNE_UVT_SPLIT): 1 = bf16x3 split-product main kernel where it applies
bool split_form_applies(int d) { return g_uvt_split != 0 && (d == 32 || d == 64 || d == 128 || d == 256); }

int g_uvt_target_wgs = 512;    // mfcd_set_tuning(MFCD_TUNE_UVT_TARGET_WGS): workgroups the tiled form aims for: one round of the chip (two
                               // per CU), then whatever the L2 rule below adds (C3: 617 us at 1024 workgroups, 639 us at 4096)

int g_uvt_min_stages = 8;      // mfcd_set_tuning(MFCD_TUNE_UVT_MIN_STAGES): stages a workgroup sweeps at least

UvtWs plan_ws(char *base, int n, int m, int d)
{
    UvtWs w;
    const int rtiles = (n + 31) / 32;
    const TiledCfg tc = n >= 32 && (int64_t)m * d < (int64_t)0x7fff0000 ? tiled_cfg(d) : TiledCfg{0, 0};
    if (tc.NW) {
        // tiled form: several rounds of workgroups over the chip (the hardware balances them), a split's V rows
        // small enough for one XCD's L2 (4 MiB), >= 2 stages per split; 8 or more splits (a multiple of 8 where the
        // column count allows) so that the XCD-aware mapping applies
        const int row_blocks = (n + tc.NW * 32 - 1) / (tc.NW * 32);
        const int stages = (m + tc.TC - 1) / tc.TC;
        int64_t want = (g_uvt_target_wgs + row_blocks - 1) / row_blocks;
        const int64_t by_l2 = ((int64_t)m * d * 4 + (2 << 20) - 1) / (2 << 20);
        if (by_l2 > want) want = by_l2;
        int splits = want <= 1 ? 1 : (int)((want + 7) / 8 * 8);
        // at least 8 stages per workgroup where the column count allows: the prologue (U fragment, first stage) is worth
        // ~1.5 stages (C2: 8 splits 77.8 us, 32 splits 83.6 us for the whole pass)
        const int ms_ = g_uvt_min_stages;
        const int max_splits = stages / ms_ > 0 ? stages / ms_ : (stages / 2 > 0 ? stages / 2 : 1);
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
    w.tiled = tc.NW != 0;
    w.n_err = w.splits * rtiles;
    w.nblk = (n + 255) / 256;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base ? base + off : nullptr; off += al(b); return p; };
    w.colpart = (double *)take(sizeof(double) * 2 * kSlices * (size_t)d);
    w.bar = (float *)take(sizeof(float) * 2 * (size_t)d);
    w.rm = (float *)take(sizeof(float) * (size_t)n);
    w.cm = (float *)take(sizeof(float) * (size_t)m);
    w.xm = (float *)take(sizeof(float) * (size_t)n);
    w.scc = (double *)take(sizeof(double) * (size_t)n);
    w.sxx = (double *)take(sizeof(double) * (size_t)n);
    w.part_rows = (double *)take(sizeof(double) * (w.tiled ? kTiledRowSums : 2) * (size_t)n * w.splits);
    w.part_err = (double *)take(sizeof(double) * (size_t)w.n_err);
    w.part_xx = (double *)take(sizeof(double) * (size_t)w.n_err);
    w.blk = (double *)take(sizeof(double) * 2 * (size_t)(w.nblk > rtiles ? w.nblk : rtiles));
    w.ncnt = 1 + rtiles;
    w.cnt = (unsigned *)take(sizeof(unsigned) * (size_t)w.ncnt);
    w.dummy_rows = (double *)take(sizeof(double) * 8 * (size_t)n);   // output the caller did not ask for (select entry)
    w.dummy_scal = (double *)take(sizeof(double) * 4);
    // split-product form (d in {32, 64, 128, 256}): 4 d bytes per V row like the fp32 table, rows padded to whole stages plus one
    // (the stage DMA never leaves the buffer)
    w.vsplit = nullptr;
    if (w.tiled && split_form_applies(d)) {
        const size_t rows = ((size_t)m + tc.TC - 1) / tc.TC * tc.TC + tc.TC;
        w.vsplit = (mfcd_bf16 *)take(rows * (size_t)d * 4);
        if (!base) w.vsplit = (mfcd_bf16 *)(uintptr_t)1;   // size query: "would be allocated"
    }
    w.bytes = off;
    return w;
}

int g_uvt_wpe128 = 2;   // mfcd_set_tuning(MFCD_TUNE_UVT_WPE128): 2 = X prefetch at 2 waves/SIMD (default), 3 = round-1 form

template <int DD, int NW, int TC, int WPE, bool PFX, int WHAT>
void launch_tiled_what(const UvtWs &w, const float *U, const float *V, const float *X, int n, int m, float s, bool xv,
                       const UvtTail &tail, hipStream_t st)
{
    const int row_blocks = (n + NW * 32 - 1) / (NW * 32);
    const unsigned blocks = (unsigned)row_blocks * (w.splits >= 8 ? 8u * ((w.splits + 7) / 8) : (unsigned)w.splits);
    {
        if (xv && w.vsplit) {   // bf16x3 split product: V comes from the split table centre_vectors_kernel wrote
            hipLaunchKernelGGL((uvt_tiled_kernel<DD, NW, TC, true, 2, true, WHAT, true>), dim3(blocks), dim3(NW * 64), 0, st, U,
                               (const float *)w.vsplit, X, w.rm, w.cm, n, m, s, w.cols_per_split, w.splits, row_blocks,
                               w.part_rows, w.part_err, w.part_xx, tail);
            return;
        }
    }
    if (xv)
        hipLaunchKernelGGL((uvt_tiled_kernel<DD, NW, TC, true, WPE, PFX, WHAT>), dim3(blocks), dim3(NW * 64), 0, st, U, V,
                           X, w.rm, w.cm, n, m, s, w.cols_per_split, w.splits, row_blocks, w.part_rows, w.part_err,
                           w.part_xx, tail);
    else
        hipLaunchKernelGGL((uvt_tiled_kernel<DD, NW, TC, false, WPE, false, WHAT>), dim3(blocks), dim3(NW * 64), 0, st, U,
                           V, X, w.rm, w.cm, n, m, s, w.cols_per_split, w.splits, row_blocks, w.part_rows, w.part_err,
                           w.part_xx, tail);
}

template <int DD, int NW, int TC, int WPE = 2, bool PFX = (DD <= 128)>
int launch_tiled(const UvtWs &w, const float *U, const float *V, const float *X, int n, int m, float s, bool xv,
                 int what, const UvtTail &tail, hipStream_t st)
{
    if (what == 1) launch_tiled_what<DD, NW, TC, WPE, PFX, 1>(w, U, V, X, n, m, s, xv, tail, st);
    else if (what == 2) launch_tiled_what<DD, NW, TC, WPE, PFX, 2>(w, U, V, X, n, m, s, xv, tail, st);
    else launch_tiled_what<DD, NW, TC, WPE, PFX, 3>(w, U, V, X, n, m, s, xv, tail, st);
    return 0;
}

}  // namespace

namespace mfcd_detail {
int set_uvt_target_wgs(int v)
{
    if (v < 256 || v > (1 << 20)) return MFCD_EINVAL;
    g_uvt_target_wgs = v;
    return 0;
}

int set_uvt_min_stages(int v)
{
    if (v < 1 || v > 64) return MFCD_EINVAL;
    g_uvt_min_stages = v;
    return 0;
}

int set_uvt_split(int v)
{
    if (v != 0 && v != 1) return MFCD_EINVAL;
    g_uvt_split = v;
    return 0;
}

int set_uvt_wpe128(int v)
{
    if (v != 2 && v != 3) return MFCD_EINVAL;
    g_uvt_wpe128 = v;
    return 0;
}
}  // namespace mfcd_detail

extern "C" size_t mfcd_uvt_workspace_bytes(int n, int m, int d)
{
    if (n <= 0 || m <= 0 || d <= 0) return 0;
    return plan_ws(nullptr, n, m, d).bytes;
}

extern "C" size_t mfcd_uvt_slab_workspace_bytes(int n, int m, int d, int nrows)
{
    if (n <= 0 || m <= 0 || d <= 0 || nrows <= 0 || nrows > n) return 0;
    return plan_ws(nullptr, n, m, d).bytes + plan_ws(nullptr, nrows, m, d).bytes;
}

namespace {

// The pass over rows [row0, row0 + nrows) of U against the slab Xs [nrows][m]; the centring vectors come from ALL n rows
// of U (wf: the plan for n rows holds rm, cm; ws: the plan for nrows rows holds the slab's partial sums; the two are the
// same plan when the slab is the whole matrix).
int run_uvt(const float *U, const float *V, const float *Xs, int n, int m, int d, double s, int what, int row0,
            int nrows, double *row_stats, double *scal, const UvtWs &wf, const UvtWs &ws, hipStream_t st)
{
    const bool xv = (reinterpret_cast<uintptr_t>(Xs) & 15u) == 0 && m % 4 == 0;
    const bool al16 = ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(V)) & 15u) == 0;
    // fused form: X read ONCE, 3 launches (round 1: 8 launches, X read twice; round 2: 4-5); tables off a 16-byte boundary take the
    // generic form (the plan's column split suits both)
    const bool tiled = ws.tiled && al16;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(slices_for(n > m ? n : m), 2), dim3(256), 0, st, U, V, n, m, d,
                       wf.colpart, ws.cnt, ws.ncnt);
    {   // 16 rows per workgroup (4 per wave).  Small tables: every workgroup reduces the few partial sums itself
        // (one launch less); large ones: one small kernel reduces them once.
        const int64_t rows = (int64_t)n + m;
        const bool merged = (int64_t)(slices_for(n) + slices_for(m)) * d <= 8192;
        if (!merged)
            hipLaunchKernelGGL(colsum_final_kernel, dim3((d + 63) / 64, 2), dim3(256), 0, st, wf.colpart, n, m, d, wf.bar);
        const int rpb = 16;
        hipLaunchKernelGGL(centre_vectors_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256),
                           sizeof(float) * 2 * (size_t)(d + 1) + (merged ? sizeof(double) * 8 * (size_t)d : 0), st, U, V, wf.colpart, merged ? (const float *)nullptr : wf.bar, n,
                           m, d, rpb, wf.rm, wf.cm, (ws.tiled && al16 && xv && !(d == 128 && g_uvt_wpe128 == 3)) ? wf.vsplit : nullptr);
    }
    const float *Us = U + (int64_t)row0 * d;
    UvtWs w = ws;            // the slab's partial-sum arrays, with the full problem's centring vectors
    w.rm = wf.rm + row0;
    w.cm = wf.cm;
    w.vsplit = (ws.tiled && al16 && xv && !(d == 128 && g_uvt_wpe128 == 3)) ? wf.vsplit : nullptr;
    const int nn = nrows;
    if (tiled) {
        // short passes (C2: 37 us), three launches: column sums, centring vectors (+ split table), main kernel — whose
        // tail finishes the rows and the two global sums (round 2: two more launches for those).  Long passes keep the
        // separate final kernel(s): counting every workgroup in (a wait for its stores and an atomic's round trip, ~3 us
        // of a workgroup's ~150) and the finishing workgroups at the very end cost C3 2.5 % and C5 4 %.
        const bool fold = (int64_t)nn * m <= ((int64_t)1 << 26);
        const UvtTail tail{row_stats, w.blk, scal, fold ? w.cnt : nullptr, s};
        int rc = 0;
        if (d == 256) rc = launch_tiled<256, 4, 32>(w, Us, V, Xs, nn, m, (float)s, xv, what, tail, st);
        else if (d == 128 && g_uvt_wpe128 == 3) rc = launch_tiled<128, 4, 32, 3, false>(w, Us, V, Xs, nn, m, (float)s, xv, what, tail, st);
        else if (d == 128) rc = launch_tiled<128, 4, 64, 2, true>(w, Us, V, Xs, nn, m, (float)s, xv, what, tail, st);
        else if (d == 64) rc = launch_tiled<64, 4, 64>(w, Us, V, Xs, nn, m, (float)s, xv, what, tail, st);
        else rc = launch_tiled<32, 4, 128>(w, Us, V, Xs, nn, m, (float)s, xv, what, tail, st);
        if (rc) return rc;
        if (fold) {
            MFCD_HIP_TRY(hipGetLastError());
            return 0;
        }
        // tiny n (one row per thread), or only the global sums wanted and few enough shares: one workgroup finishes the
        // rows AND the two global sums (no uvt_scal_kernel launch)
        if (nn <= 1024 || (what == 2 && w.n_err <= 16384)) {
            hipLaunchKernelGGL(uvt_final_tiled_kernel, dim3(1), dim3(1024), 0, st, w.part_rows, w.part_err, w.part_xx, w.rm, nn,
                               m, w.splits, w.cols_per_split, w.n_err, row_stats, w.blk, s, scal, what);
            MFCD_HIP_TRY(hipGetLastError());
            return 0;
        }
        hipLaunchKernelGGL(uvt_final_tiled_kernel, dim3(w.nblk), dim3(256), 0, st, w.part_rows, w.part_err, w.part_xx, w.rm, nn,
                           m, w.splits, w.cols_per_split, w.n_err, row_stats, w.blk, s, (double *)nullptr, what);
    } else {
        // generic form (any d, tables of fewer than 32 rows): X row statistics from a sweep of their own
        if (xv) hipLaunchKernelGGL(x_rows_kernel<true>, dim3((nn + 3) / 4), dim3(256), 0, st, Xs, nn, m, w.xm, w.scc, w.sxx);
        else hipLaunchKernelGGL(x_rows_kernel<false>, dim3((nn + 3) / 4), dim3(256), 0, st, Xs, nn, m, w.xm, w.scc, w.sxx);
        const dim3 grid(((nn + 31) / 32 + 3) / 4, w.splits);
#define MFCD_UVT(DD)                                                                                               \
    hipLaunchKernelGGL((uvt_main_kernel<DD>), grid, dim3(256), 0, st, Us, V, Xs, w.rm, w.cm, w.xm, nn, m, d, (float)s, \
                       w.cols_per_split, w.part_rows, w.part_err)
        if (al16 && d == 8) MFCD_UVT(8);
        else if (al16 && d == 16) MFCD_UVT(16);
        else if (al16 && d == 32) MFCD_UVT(32);
        else if (al16 && d == 64) MFCD_UVT(64);
        else if (al16 && d == 128) MFCD_UVT(128);
        else if (al16 && d == 256) MFCD_UVT(256);
        else MFCD_UVT(0);
#undef MFCD_UVT
        hipLaunchKernelGGL(uvt_final_kernel, dim3(w.nblk), dim3(256), 0, st, w.part_rows, w.part_err, w.rm, w.xm,
                           w.scc, w.sxx, nn, w.splits, w.n_err, row_stats, w.blk);
    }
    if (what & 2) hipLaunchKernelGGL(uvt_scal_kernel, dim3(1), dim3(256), 0, st, w.blk, w.nblk, s, scal);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" int mfcd_uvt_stats_select(const float *U, const float *V, const float *X, int n, int m, int d, double s,
                                     int what, double *row_stats, double *scal, void *workspace, size_t workspace_bytes,
                                     void *stream)
{
    if (!U || !V || !X || !workspace || n <= 0 || m <= 0 || d <= 0 || d > MFCD_MAX_D || what < 1 || what > 3)
        return MFCD_EINVAL;
    if (((what & 1) && !row_stats) || ((what & 2) && !scal)) return MFCD_EINVAL;
    const UvtWs w = plan_ws((char *)workspace, n, m, d);
    if (workspace_bytes < w.bytes) return MFCD_EWORKSPACE;
    if (!(what & 1)) row_stats = w.dummy_rows;   // not asked for: written to scratch (generic form) or not at all (tiled)
    if (!(what & 2)) scal = w.dummy_scal;
    return run_uvt(U, V, X, n, m, d, s, what, 0, n, row_stats, scal, w, w, (hipStream_t)stream);
}

extern "C" int mfcd_uvt_stats(const float *U, const float *V, const float *X, int n, int m, int d, double s,
                              double *row_stats, double *scal, void *workspace, size_t workspace_bytes, void *stream)
{
    return mfcd_uvt_stats_select(U, V, X, n, m, d, s, 3, row_stats, scal, workspace, workspace_bytes, stream);
}

extern "C" int mfcd_uvt_stats_slab(const float *U, const float *V, const float *X_slab, int n, int m, int d, double s,
                                   int what, int row0, int nrows, double *row_stats_slab, double *scal_slab,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    if (!U || !V || !X_slab || !workspace || n <= 0 || m <= 0 || d <= 0 || d > MFCD_MAX_D || what < 1 || what > 3)
        return MFCD_EINVAL;
    if (row0 < 0 || nrows <= 0 || (int64_t)row0 + nrows > n) return MFCD_EINVAL;
    if (((what & 1) && !row_stats_slab) || ((what & 2) && !scal_slab)) return MFCD_EINVAL;
    const UvtWs wf = plan_ws((char *)workspace, n, m, d);
    const UvtWs ws = plan_ws((char *)workspace + wf.bytes, nrows, m, d);
    if (workspace_bytes < wf.bytes + ws.bytes) return MFCD_EWORKSPACE;
    if (!(what & 1)) row_stats_slab = ws.dummy_rows;
    if (!(what & 2)) scal_slab = ws.dummy_scal;
    return run_uvt(U, V, X_slab, n, m, d, s, what, row0, nrows, row_stats_slab, scal_slab, wf, ws, (hipStream_t)stream);
}

#if MFCD_UVT_STAMPS
// diagnostic build only: read and clear the in-kernel cycle accounting (not declared in include/mfcd.h)
extern "C" int mfcd_uvt_debug_read(unsigned long long *out8_host)
{
    MFCD_HIP_TRY(hipDeviceSynchronize());
    MFCD_HIP_TRY(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(mfcd_uvt_dbg), 8 * sizeof(unsigned long long)));
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    MFCD_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mfcd_uvt_dbg), zero, sizeof(zero)));
    return 0;
}
#endif

extern "C" int mfcd_uvt_rows(const float *U, const float *V, const int32_t *row_ids, int k, int n, int m, int d,
                             float *out, void *stream)
{
    if (!U || !V || !out || k < 0 || n <= 0 || m <= 0 || d <= 0) return MFCD_EINVAL;
    if (k == 0) return 0;
    if (!row_ids) return MFCD_EINVAL;
    hipLaunchKernelGGL(uvt_rows_kernel, dim3((m + 255) / 256, k), dim3(256), 0, (hipStream_t)stream, U, V, row_ids, k,
                       n, m, d, out);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}
