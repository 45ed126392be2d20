// eval.hip — no-grad forward + BCE pass (validation loop structure.py:858-868, evaluate_model
// structure.py:899-916) and sample validation.  One workgroup of 16 waves per batch of B samples; one wave per
// sample for the row gathers (coalesced d-float rows) and the 64-lane shuffle reduction (a wave handles every 16th
// sample of the batch: the pass is a chain of dependent gathers, so it wants waves, not bytes); the batch
// mean and the match count are reduced in a fixed order, so results are run-to-run identical.
#include "common.h"

namespace {

constexpr int kEvalWaves = 16;

template <typename TP>
__global__ __launch_bounds__(kEvalWaves * 64) void eval_batches_kernel(const TP *__restrict__ U, const TP *__restrict__ V,
                                                           const mfcd_sample *__restrict__ samples, int64_t N, int B,
                                                           int d, float *__restrict__ loss_per_batch,
                                                           int32_t *__restrict__ correct_per_batch,
                                                           float *__restrict__ p_out)
{
    extern __shared__ __attribute__((aligned(16))) float terms[];  // [B] BCE terms, then [B] matches
    int *match = reinterpret_cast<int *>(terms + B);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t off = (int64_t)blockIdx.x * B;
    const int b = (int)((N - off) < B ? (N - off) : B);
    for (int t = wave; t < b; t += kEvalWaves) {
        const mfcd_sample s = samples[off + t];
        const float p = sigmoid_f32(wave_score(U, V, s.u, s.i, s.j, d, lane));
        if (lane == 0) {
            terms[t] = bce_term_f32(p, s.z);
            match[t] = ((p > 0.5f ? 1.0f : 0.0f) == s.z) ? 1 : 0;  // structure.py:912-915
            if (p_out) p_out[off + t] = p;
        }
    }
    __syncthreads();
    if (wave == 0) {
        float acc = 0.0f;
        int c = 0;
        for (int t = lane; t < b; t += MFCD_WAVE) {
            acc += terms[t];
            c += match[t];
        }
        acc = wave_sum64(acc);
        c = wave_sum64_i(c);
        if (lane == 0) {
            loss_per_batch[blockIdx.x] = acc / (float)b;
            if (correct_per_batch) correct_per_batch[blockIdx.x] = c;
        }
    }
}

__global__ __launch_bounds__(256) void check_samples_kernel(const mfcd_sample *__restrict__ samples, int64_t N, int n,
                                                            int m, int32_t *__restrict__ bad)
{
    int local = 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < N; t += (int64_t)gridDim.x * 256) {
        const mfcd_sample s = samples[t];
        local += (s.u < 0 || s.u >= n || s.i < 0 || s.i >= m || s.j < 0 || s.j >= m) ? 1 : 0;
    }
    local = wave_sum64_i(local);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(bad, local);
}

}  // namespace

namespace {
template <typename TP>
int eval_batches_impl(const TP *U, const TP *V, const mfcd_sample *samples, int64_t N, int B, int n, int m, int d,
                      float *loss_per_batch, int32_t *correct_per_batch, float *p_out, void *stream)
{
    if (!U || !V || n <= 0 || m <= 0 || d <= 0 || d > MFCD_MAX_D || N < 0 || B <= 0 || B > 16384) return MFCD_EINVAL;
    if (N == 0) return 0;
    if (!samples || !loss_per_batch) return MFCD_EINVAL;
    const int64_t nb = (N + B - 1) / B;
    hipLaunchKernelGGL((eval_batches_kernel<TP>), dim3((unsigned)nb), dim3(kEvalWaves * 64), sizeof(float) * 2 * (size_t)B,
                       (hipStream_t)stream, U, V, samples, N, B, d, loss_per_batch, correct_per_batch, p_out);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int mfcd_eval_batches(const float *U, const float *V, const mfcd_sample *samples, int64_t N, int B, int n,
                                 int m, int d, float *loss_per_batch, int32_t *correct_per_batch, float *p_out,
                                 void *stream)
{
    return eval_batches_impl<float>(U, V, samples, N, B, n, m, d, loss_per_batch, correct_per_batch, p_out, stream);
}

extern "C" int mfcd_eval_batches_bf16(const uint16_t *U, const uint16_t *V, const mfcd_sample *samples, int64_t N,
                                      int B, int n, int m, int d, float *loss_per_batch, int32_t *correct_per_batch,
                                      float *p_out, void *stream)
{
    return eval_batches_impl<mfcd_bf16>((const mfcd_bf16 *)U, (const mfcd_bf16 *)V, samples, N, B, n, m, d,
                                        loss_per_batch, correct_per_batch, p_out, stream);
}

extern "C" int mfcd_check_samples(const mfcd_sample *samples, int64_t N, int n, int m, int32_t *bad_count_dev,
                                  void *stream)
{
    if (!bad_count_dev || N < 0 || n <= 0 || m <= 0) return MFCD_EINVAL;
    MFCD_HIP_TRY(hipMemsetAsync(bad_count_dev, 0, sizeof(int32_t), (hipStream_t)stream));
    if (N == 0) return 0;
    if (!samples) return MFCD_EINVAL;
    int64_t blocks = (N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(check_samples_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, samples, N, n,
                       m, bad_count_dev);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}
