// Shared device helpers for the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfcd.h"

// No implicit FMA contraction anywhere in the device code: the reference rounds every product before it is
// summed (it materialises u*(vi-vj), g*(vi-vj), g*u as tensors), and contraction chosen per inlining context
// would make two copies of the same source round differently.  fma is written explicitly where ATen uses it.
#pragma clang fp contract(off)

#define MFCD_WAVE 64

#define MFCD_HIP_TRY(expr)                        \
    do {                                          \
        hipError_t e__ = (expr);                  \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// DPP lane move inside a row of 16 (no LDS round trip, unlike __shfl / ds_bpermute).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes; every lane gets the same value, fixed order:
// quad (xor 1, xor 2) -> 8 (row_half_mirror) -> 16 (row_mirror) with DPP, then the four row sums through
// v_readlane as (r0 + r1) + (r2 + r3).  After each stage the partner lanes hold identical partial sums,
// so the mirrored reads are equivalent to xor 4 / xor 8 butterflies.
__device__ __forceinline__ float wave_sum64(float x)
{
    x += dpp_move<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_move<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_move<0x141>(x);  // row_half_mirror
    x += dpp_move<0x140>(x);  // row_mirror
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ int wave_sum64_i(int x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, MFCD_WAVE);
    return x;
}

// torch.sigmoid in fp32 (structure.py:795): 1/(1+exp(-x)); rounds to exactly 0/1 when saturated.
__device__ __forceinline__ float sigmoid_f32(float x) { return 1.0f / (1.0f + expf(-x)); }

// F.binary_cross_entropy element term with ATen's clamp of the logs at -100 (structure.py:849).
__device__ __forceinline__ float bce_term_f32(float p, float z)
{
#pragma clang fp contract(off)
    const float l1 = fmaxf(log1pf(-p), -100.0f);
    const float l0 = fmaxf(logf(p), -100.0f);
    return (z - 1.0f) * l1 - z * l0;
}

// Backward coefficient dL/dx for one sample, divisor = batch size (structure.py:850):
// BCE backward (grad/B)*(p-z)/max((1-p)p,1e-12) followed by sigmoid backward *(1-p)*p.
__device__ __forceinline__ float bce_sigmoid_backward_f32(float p, float z, float inv_batch)
{
#pragma clang fp contract(off)
    const float den = fmaxf((1.0f - p) * p, 1e-12f);
    const float a = inv_batch * (p - z) / den;
    return a * (1.0f - p) * p;
}

// Factor tables are stored as fp32 or bf16 (BASELINE configs[2]: "bf16 factors"); arithmetic is always fp32.
typedef __bf16 mfcd_bf16;
__device__ __forceinline__ float ldf(const float *p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const mfcd_bf16 *p, int64_t i) { return (float)p[i]; }

// x_t = sum_k U[u][k] * (V[i][k] - V[j][k]) computed by one whole wave; all lanes return x_t.
template <typename TP>
__device__ __forceinline__ float wave_score(const TP *__restrict__ U, const TP *__restrict__ V, int u, int i, int j,
                                            int d, int lane)
{
    const TP *ur = U + (int64_t)u * d, *vi = V + (int64_t)i * d, *vj = V + (int64_t)j * d;
    float acc = 0.0f;
    for (int k = lane; k < d; k += MFCD_WAVE) acc += ldf(ur, k) * (ldf(vi, k) - ldf(vj, k));  // product rounded, then summed
    return wave_sum64(acc);
}
