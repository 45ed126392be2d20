/*
 * mfcd.h — C-ABI of libmfcd_hip.so: the MI355X (gfx950) hot path of triplet-comparison matrix
 * factorisation.  Plain pointers and sizes only; no torch types.
 *
 * The reference (MayeulCassier/Matrix-Factorization-With-Comparison-Data) is pure Python on
 * PyTorch and has no FFI layer of its own; each entry point below replaces the PyTorch op
 * sequence of the cited reference lines (paths into the reference repository).  The Python
 * binding a maintainer would add is shown in INTEGRATION.md; the in-tree one is
 * matrix-factorization-with-comparison-data_amd/mfcd/_lib.py (ctypes).
 *
 * Conventions
 *  - every pointer except `scalars_host`-style arguments is a DEVICE pointer borrowed from the
 *    caller; the library allocates nothing persistent and frees nothing it did not allocate;
 *  - every entry returns 0 on success, a positive hipError_t, or a negative MFCD_E* code;
 *    mfcd_error_string() renders either;
 *  - all work is enqueued on `stream` (a hipStream_t, may be NULL = default stream); no entry
 *    synchronises the host with the device;
 *  - factor tables are row-major contiguous fp32: U [n][d], V [m][d];
 *  - a sample is the 16-byte record mfcd_sample {int32 u, i, j; float z} (reference batch tuple
 *    (u, i, j, z), structure.py:527-531, with the label already cast to fp32 as at 849);
 *    0 <= u < n, 0 <= i,j < m are validated by mfcd_check_samples, not by the hot kernels.
 */
#ifndef MFCD_H
#define MFCD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFCD_ABI_VERSION 1

#define MFCD_EINVAL (-1)   /* bad argument (null pointer, non-positive size, d out of range)   */
#define MFCD_EWORKSPACE (-2) /* workspace smaller than mfcd_*_workspace_bytes says             */
#define MFCD_EALIGN (-3)   /* a table pointer is not 4-byte aligned                            */
#define MFCD_EINDEX (-4)   /* a sample indexes outside [0,n) x [0,m)^2 (mfcd_check_samples)     */
#define MFCD_ERCCL (-5)    /* RCCL is not loadable in this process, or an RCCL call failed      */

#define MFCD_MAX_D 1024

typedef struct mfcd_sample {
    int32_t u, i, j;
    float z;
} mfcd_sample;

int mfcd_abi_version(void);
const char *mfcd_error_string(int code);

/*
 * Validates 0<=u<n, 0<=i<m, 0<=j<m for N samples.  Writes the number of bad records to
 * *bad_count_dev (device int32).  The reference raises IndexError from U[u]/V[i]
 * (structure.py:787-789); the Python host turns a non-zero count into the same exception.
 */
int mfcd_check_samples(const mfcd_sample *samples, int64_t N, int n, int m, int32_t *bad_count_dev,
                       void *stream);

/*
 * Forward + BCE over N samples in consecutive batches of B (last one short), no gradient.
 * Replaces, per batch, model(u,i,j) (structure.py:773-795) + F.binary_cross_entropy(pred,
 * z.float()) (structure.py:864 validation loop, 908 evaluate_model) + (pred > 0.5) == z
 * (structure.py:912-915).
 *   loss_per_batch[k]    fp32 mean BCE of batch k (what loss.item() returns)
 *   correct_per_batch[k] int32 count of matches in batch k            (nullable)
 *   p_out[t]             fp32 sigmoid output per sample               (nullable)
 * Number of batches = ceil(N/B).
 */
int mfcd_eval_batches(const float *U, const float *V, const mfcd_sample *samples, int64_t N, int B,
                      int n, int m, int d, float *loss_per_batch, int32_t *correct_per_batch,
                      float *p_out, void *stream);

/*
 * Bytes of device workspace mfcd_train_steps needs for these sizes.  The first 4 bytes of the
 * workspace are an int32 status word written by the call: 0 = ok, 1 = a bounded in-kernel wait
 * expired (resident form only; U, V, m, v are then undefined).  Read it after the stream has drained.
 */
size_t mfcd_train_workspace_bytes(int64_t N, int B, int n, int m, int d);

/*
 * Which form of the fused step mfcd_train_steps uses (process-wide):
 *   0 auto (default)  resident when it applies, else streaming
 *   1 streaming       one launch per optimiser step, state streamed through HBM (any size, any d)
 *   2 resident        one persistent launch per call, p/m/v held in registers, rows exchanged through
 *                     tagged 8-byte granules; needs d a power of two <= 256 and 12*(n+m)*d bytes of
 *                     state to fit the register files; MFCD_EINVAL from mfcd_train_steps otherwise
 *   3 local           tiny problems ((n+m)*d <= 16384, B <= 4096, any d): one persistent launch of ONE
 *                     workgroup, parameters in LDS, moments in registers, three barriers per step;
 *                     MFCD_EINVAL otherwise.  Measured no faster than the resident form (one CU's ALU and
 *                     the serial per-row accumulation bound it), so "auto" does not select it.
 * Both forms compute the same step (same summation order per row); results agree to fp32 rounding.
 */
int mfcd_set_train_path(int mode);

/*
 * Arithmetic flavour of Adam inside the RESIDENT form, where the step is bound by vector-ALU cycles
 * (process-wide; the streaming form is HBM-bound and always uses the IEEE flavour):
 *   1 fast (default)  same operation order, but sqrt is the hardware v_sqrt_f32 (<= 1 ulp) and the two
 *                     divisions are reciprocal-multiply with one Newton correction (<= 1 ulp) instead of
 *                     the IEEE-rounded expansions; the update differs by a few ulp (~1e-10 per step)
 *   0 ieee            every operation IEEE-rounded, as ATen's CPU kernels
 * Either flavour is deterministic and keeps every parity test within the stated tolerances.
 */
int mfcd_set_resident_math(int fast);

/*
 * Runs ceil(N/B) sequential optimiser steps on the device, consuming `samples` in order in
 * batches of B (last one short, divisor = actual batch size).  One step replaces
 * structure.py:847-851: zero_grad, forward, BCE(mean), backward (gather + scatter-add of row
 * gradients into U and V) and torch.optim.Adam.step() (coupled L2 weight decay, bias-corrected,
 * dense over every row; torch/optim/adam.py _single_tensor_adam).
 *   U,V,mU,vU,mV,vV  parameters and Adam moments (exp_avg, exp_avg_sq), updated in place
 *   step0            optimiser steps already taken (Adam's `step` before this call)
 *   loss_per_step[k] fp32 batch-mean BCE of step k (structure.py:852 loss.item())
 * No dense gradient is materialised and there is no host synchronisation.
 */
int mfcd_train_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                     const mfcd_sample *samples, int64_t N, int B, int64_t step0, int n, int m,
                     int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                     float *loss_per_step, void *workspace, size_t workspace_bytes, void *stream);

/*
 * bf16 factor storage (BASELINE.json configs[2]; the reference has no such mode, the rounding points are defined
 * here and in oracle/mfcd_oracle.c): U [n][d], V [m][d] are bf16 in HBM; each step reads them as such, does all
 * arithmetic and keeps the Adam moments in fp32, and rounds the updated parameters to the nearest bf16 (ties to
 * even) once, when they are written back.  Always the streaming form; 20 bytes per element per step instead of 24.
 */
int mfcd_train_steps_bf16(uint16_t *U, uint16_t *V, float *mU, float *vU, float *mV, float *vV,
                          const mfcd_sample *samples, int64_t N, int B, int64_t step0, int n, int m,
                          int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                          float *loss_per_step, void *workspace, size_t workspace_bytes, void *stream);
int mfcd_eval_batches_bf16(const uint16_t *U, const uint16_t *V, const mfcd_sample *samples, int64_t N,
                           int B, int n, int m, int d, float *loss_per_batch,
                           int32_t *correct_per_batch, float *p_out, void *stream);

/*
 * Diagnostic twin of mfcd_train_steps for bench.py's roofline figure: identical work, but every step
 * launch is bracketed by its own pair of HIP events on `stream`, and the call WAITS for the last one.
 * kernel_us_host[3] (HOST memory) receives the average / min / max step-kernel duration in microseconds.
 * Not for the training loop (it synchronises and the events perturb launch pacing).
 */
int mfcd_train_steps_timed(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                           const mfcd_sample *samples, int64_t N, int B, int64_t step0, int n, int m,
                           int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                           float *loss_per_step, void *workspace, size_t workspace_bytes, void *stream,
                           float *kernel_us_host);

/*
 * Split form for data-parallel training (one exchange step between the two calls):
 * mfcd_batch_coefficients computes, for B samples of ONE batch, the sigmoid output, the BCE
 * term and the backward coefficient g_t = dL/dx_t with divisor `batch_divisor` (the GLOBAL
 * batch size), i.e. structure.py:848-850 up to the scalar per sample.
 */
int mfcd_batch_coefficients(const float *U, const float *V, const mfcd_sample *samples, int B,
                            int n, int m, int d, int batch_divisor, float *g_out, float *term_out,
                            float *p_out, void *stream);

/*
 * One dense Adam step given per-sample coefficients for a (global) batch of B samples:
 * applies  dU[u]+=g(V[i]-V[j]), dV[i]+=gU[u], dV[j]-=gU[u]  in batch order on top of the
 * weight-decay gradient and updates U,V,m,v in place (structure.py:850-851).
 * `step` is Adam's 1-based step number.  workspace: mfcd_train_workspace_bytes(B,B,n,m,d).
 */
int mfcd_apply_step(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                    const mfcd_sample *samples, const float *g, int B, int64_t step, int n, int m,
                    int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                    void *workspace, size_t workspace_bytes, void *stream);

/*
 * Data-parallel form with the dense exchange the north star names (RCCL all-reduce of the factor
 * gradients): mfcd_dense_grad overwrites gradU [n][d], gradV [m][d] with THIS rank's share of the batch
 * gradient (structure.py:848-850; divisor = the GLOBAL batch size; rows no local sample touches are 0);
 * term_out[t] (nullable) receives the BCE term of local sample t.  After the caller has summed the
 * buffers over ranks, mfcd_adam_dense applies torch.optim.Adam's step (structure.py:851) from them.
 */
int mfcd_dense_grad(const float *U, const float *V, const mfcd_sample *samples, int B, int n, int m,
                    int d, int batch_divisor, float *gradU, float *gradV, float *term_out,
                    void *stream);
int mfcd_adam_dense(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                    const float *gradU, const float *gradV, int64_t step, int n, int m, int d,
                    double lr, double beta1, double beta2, double eps, double weight_decay,
                    void *stream);

/*
 * Data-parallel training loop, native (one process per GPU; RCCL is bound at run time, the library has no link-time
 * dependency on it).  Sharding as SURVEY section 8e: a GLOBAL batch of B*world samples per optimiser step, rank r owns
 * the contiguous slice [r*B, (r+1)*B) of it, the divisor of the mean is the global batch size, every rank applies the
 * same gathered quantities in the same order, so replicas stay bit-identical and the run equals the single-GPU run
 * with batch_size = B*world (structure.py:840-852 with a larger DataLoader batch).  Per step: one coefficient
 * kernel over the rank's shard, ONE in-place ncclAllGather of B {g_t, BCE term_t} float pairs per rank, one fused
 * step kernel over the global batch; everything is enqueued on `stream`, nothing synchronises the host.
 *
 *   mfcd_dp_unique_id      rank 0: fills a 128-byte ncclUniqueId, which the caller distributes to the other ranks
 *   mfcd_dp_comm_create    every rank (current HIP device): ncclCommInitRank -> opaque communicator handle
 *   mfcd_dp_comm_destroy
 *   mfcd_dp_train_steps    `samples` is the GLOBAL stream of N samples (identical on every rank);
 *                          loss_per_step[k] = mean BCE of global batch k (identical on every rank);
 *                          comm == NULL: no collective is issued and this process computes every rank's shard
 *                          itself (exact, because replicas are identical): single-process rehearsal of any world
 *                          size, and the whole of the work when world == 1.
 */
int mfcd_dp_unique_id(void *id_out, size_t id_bytes);
int mfcd_dp_comm_create(const void *id, size_t id_bytes, int rank, int world, void **comm_out);
int mfcd_dp_comm_destroy(void *comm);
size_t mfcd_dp_workspace_bytes(int64_t N, int B, int world, int n, int m, int d);
int mfcd_dp_train_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                        const mfcd_sample *samples, int64_t N, int B, int rank, int world, int64_t step0, int n,
                        int m, int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                        float *loss_per_step, void *workspace, size_t workspace_bytes, void *comm, void *stream);

/*
 * Dense UV^T pass against X [n][m] fp32 without materialising UV^T (MFMA fp32 tiles, fused
 * epilogue).  Replaces the GEMM + reductions of compute_reconstruction_error
 * (structure.py:940-952) and of compute_alpha_and_norm_ratios (structure.py:982-996, 1003-1009,
 * 1038-1064):
 *   row_stats [n][8] f64, with a = (UV^T)[r][.] - rowmean(UV^T)[r], c = X[r][.] - rowmean(X)[r]:
 *       [0] sum a*c   [1] sum a*a   [2] sum c*c   [3] rowmean(UV^T)[r]   [4] rowmean(X)[r]
 *       [5] sum x*x   [6],[7] reserved (0)
 *   scal [4] f64: [0] ||(UV^T - colmean) - sX||_F^2   [1] ||sX||_F^2   [2],[3] reserved
 * workspace: mfcd_uvt_workspace_bytes(n,m,d).
 */
size_t mfcd_uvt_workspace_bytes(int n, int m, int d);
int mfcd_uvt_stats(const float *U, const float *V, const float *X, int n, int m, int d, double s,
                   double *row_stats, double *scal, void *workspace, size_t workspace_bytes,
                   void *stream);

/*
 * k rows of UV^T: out[r][c] = sum_k U[row_ids[r]][k] * V[c][k]   (structure.py:389-392 computes
 * the full product to read two rows).
 */
int mfcd_uvt_rows(const float *U, const float *V, const int32_t *row_ids, int k, int n, int m,
                  int d, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MFCD_H */
