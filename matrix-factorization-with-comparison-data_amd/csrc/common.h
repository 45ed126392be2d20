// Shared device helpers for the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfcd.h"

#define MFCD_WAVE 64

#define MFCD_HIP_TRY(expr)                        \
    do {                                          \
        hipError_t e__ = (expr);                  \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// Sum over the 64 lanes; every lane gets the same value (butterfly, fixed order).
__device__ __forceinline__ float wave_sum64(float x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, MFCD_WAVE);
    return x;
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
    const float l1 = fmaxf(log1pf(-p), -100.0f);
    const float l0 = fmaxf(logf(p), -100.0f);
    return (z - 1.0f) * l1 - z * l0;
}

// Backward coefficient dL/dx for one sample, divisor = batch size (structure.py:850):
// BCE backward (grad/B)*(p-z)/max((1-p)p,1e-12) followed by sigmoid backward *(1-p)*p.
__device__ __forceinline__ float bce_sigmoid_backward_f32(float p, float z, float inv_batch)
{
    const float den = fmaxf((1.0f - p) * p, 1e-12f);
    const float a = inv_batch * (p - z) / den;
    return a * (1.0f - p) * p;
}

// x_t = sum_k U[u][k] * (V[i][k] - V[j][k]) computed by one whole wave; all lanes return x_t.
__device__ __forceinline__ float wave_score(const float *__restrict__ U, const float *__restrict__ V,
                                            int u, int i, int j, int d, int lane)
{
    const float *ur = U + (int64_t)u * d, *vi = V + (int64_t)i * d, *vj = V + (int64_t)j * d;
    float acc = 0.0f;
    for (int k = lane; k < d; k += MFCD_WAVE) acc += ur[k] * (vi[k] - vj[k]);
    return wave_sum64(acc);
}
