// Diagnostic (tools/ only): issue cost of wave64 fp32 vector instructions on gfx950 with 1, 2 and 4 waves per SIMD.
// Each wave runs N iterations of an unrolled block of 32 instructions over 8 independent accumulators and reports
// shader cycles (s_memtime) per instruction *per SIMD* (cycles of the slowest wave x 1 / (instructions x waves per SIMD)).
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float *out, unsigned long long *cyc, float s0, float s1, int iters)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f + 1.0f;
    float sg0 = __builtin_amdgcn_readfirstlane(s0), sg1 = __builtin_amdgcn_readfirstlane(s1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(sg0), "v"(a[(i + 2) & 7]));
                if constexpr (KIND == 2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sg1));
                if constexpr (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (KIND == 4) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if constexpr (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if constexpr (KIND == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if constexpr (KIND == 7) {   // dependent chain on ONE accumulator
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(a[1]), "v"(a[2]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
__global__ __launch_bounds__(256) void rate_pk_kernel(float *out, unsigned long long *cyc, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)(threadIdx.x + i) * 1e-3f + 1.0f, 0.5f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if constexpr (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <typename K>
static void run(const char *name, K kern, int waves_per_simd, float *out, unsigned long long *cyc)
{
    // 256 CUs x 4 SIMDs; blocks of 256 threads = 4 waves = one wave per SIMD of a CU; `waves_per_simd` blocks per CU
    const int blocks = 256 * waves_per_simd, iters = 2000;
    kern(blocks, iters);
    (void)hipDeviceSynchronize();
    kern(blocks, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    double sum = 0;
    for (auto v : h) { mx = v > mx ? v : mx; sum += (double)v; }
    const double insts = (double)iters * 32;
    printf("%-34s %d waves/SIMD: %.2f cycles per instruction per wave (mean), %.2f per SIMD-issued instruction (max wave)\n", name,
           waves_per_simd, sum / h.size() / insts, (double)mx / insts / waves_per_simd);
}

int main()
{
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    (void)hipMalloc(&cyc, 256 * 8 * 4 * sizeof(unsigned long long));
#define RUN(NAME, ...) for (int w : {1, 2, 4}) run(NAME, [&](int b, int it) { __VA_ARGS__; }, w, out, cyc);
    RUN("v_fma_f32 vgpr,vgpr,vgpr", hipLaunchKernelGGL(rate_kernel<0>, dim3(b), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f, it))
    RUN("v_fma_f32 sgpr operand", hipLaunchKernelGGL(rate_kernel<1>, dim3(b), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f, it))
    RUN("v_mul_f32 sgpr operand", hipLaunchKernelGGL(rate_kernel<2>, dim3(b), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f, it))
    RUN("v_add_f32", hipLaunchKernelGGL(rate_kernel<3>, dim3(b), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f, it))
    RUN("v_fmac_f32", hipLaunchKernelGGL(rate_kernel<6>, dim3(b), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f, it))
    RUN("v_sqrt_f32", hipLaunchKernelGGL(rate_kernel<4>, dim3(b), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f, it))
    RUN("v_rcp_f32", hipLaunchKernelGGL(rate_kernel<5>, dim3(b), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f, it))
    RUN("v_fma_f32 dependent chain", hipLaunchKernelGGL(rate_kernel<7>, dim3(b), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f, it))
    RUN("v_pk_fma_f32", hipLaunchKernelGGL(rate_pk_kernel<0>, dim3(b), dim3(256), 0, 0, out, cyc, it))
    RUN("v_pk_mul_f32", hipLaunchKernelGGL(rate_pk_kernel<1>, dim3(b), dim3(256), 0, 0, out, cyc, it))
    return 0;
}
