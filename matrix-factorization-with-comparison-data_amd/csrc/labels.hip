// labels.hip — Bradley-Terry-Luce label generation on the device.
//
// Replaces BTLPreferenceDataset._generate_labels (structure.py:493-519), a Python loop with one torch.bernoulli call
// per row (0.96 s for C2's 67 k rows, ~50 s at C4), and the host-side record packing behind it: one thread per
// triplet gathers X[u,i] - X[u,j] (dense X, or A[u].(B[i]-B[j]) for a factored X = A B^T that is never formed),
// score = sigmoid(scale * diff) in fp32 as at structure.py:509, draws K Bernoulli(score) labels and writes the
// 16-byte mfcd_sample records the training step consumes:
//   hard labels  K records per triplet, consecutive (structure.py:516-518)
//   soft labels  one record, z = mean of the K draws (structure.py:510-513; train sets only — the caller decides)
// Randomness: Philox4x32-10 (counter-based, Salmon et al. SC'11) keyed by the caller's 64-bit seed, counter =
// (triplet index, draw group): the stream is a function of (seed, triplet index) alone, so the output does not depend
// on launch geometry.  It is NOT the reference's CPU Mersenne-Twister stream: parity with the reference is
// distributional (every label is Bernoulli(score) with the same score), as SURVEY N1 states; bit-level replay of a
// reference run keeps using the host path (structure.BTLPreferenceDataset default).
#include "common.h"

namespace {

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

__global__ __launch_bounds__(256) void generate_labels_kernel(const int32_t *__restrict__ trip, int64_t T,
                                                              const float *__restrict__ X, int m,
                                                              const float *__restrict__ A, const float *__restrict__ B,
                                                              int dx, float scale, int K, int soft, unsigned seed_lo,
                                                              unsigned seed_hi, mfcd_sample *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int u = trip[3 * t], i = trip[3 * t + 1], j = trip[3 * t + 2];
    float diff;
    if (X) {
        diff = X[(int64_t)u * m + i] - X[(int64_t)u * m + j];            // structure.py:509
    } else {
        const float *a = A + (int64_t)u * dx, *bi = B + (int64_t)i * dx, *bj = B + (int64_t)j * dx;
        float xi = 0.0f, xj = 0.0f;
        for (int k = 0; k < dx; ++k) {
            xi = fmaf(a[k], bi[k], xi);
            xj = fmaf(a[k], bj[k], xj);
        }
        diff = xi - xj;
    }
    const float score = sigmoid_f32(scale * diff);
    int ones = 0;
    for (int k0 = 0; k0 < K; k0 += 4) {
        unsigned c[4] = {(unsigned)t, (unsigned)((unsigned long long)t >> 32), (unsigned)(k0 >> 2), 0x6d666364u};
        philox4x32_10(c, seed_lo, seed_hi);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (k0 + q >= K) break;
            const float uni = (float)(c[q] >> 8) * (1.0f / 16777216.0f);   // uniform on [0, 1), 24 bits
            const int z = uni < score ? 1 : 0;
            ones += z;
            if (!soft) {
                mfcd_sample s;
                s.u = u; s.i = i; s.j = j; s.z = (float)z;
                out[t * K + k0 + q] = s;
            }
        }
    }
    if (soft) {
        mfcd_sample s;
        s.u = u; s.i = i; s.j = j; s.z = (float)ones / (float)K;            // torch.mean of K fp32 0/1 draws
        out[t] = s;
    }
}

}  // namespace

extern "C" int mfcd_generate_labels(const int32_t *triplets, int64_t T, const float *X, int n, int m, const float *A,
                                    const float *B, int dx, double scale, int K, int soft, uint64_t seed,
                                    mfcd_sample *out, void *stream)
{
    if (T < 0 || n <= 0 || m <= 0 || K < 1) return MFCD_EINVAL;
    if (T == 0) return 0;
    if (!triplets || !out) return MFCD_EINVAL;
    if (!X && (!A || !B || dx <= 0)) return MFCD_EINVAL;
    hipLaunchKernelGGL(generate_labels_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       triplets, T, X, m, X ? nullptr : A, X ? nullptr : B, dx, (float)scale, K, soft ? 1 : 0,
                       (unsigned)seed, (unsigned)(seed >> 32), out);
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}
