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
 *    synchronises the host with the device except where its comment says so (mfcd_train_steps_timed;
 *    the bounded run-ahead of mfcd_train_steps; mfcd_shard_train_steps reads its collision marks once per call;
 *    mfcd_train_steps_big copies its per-step table from the stack);
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

#define MFCD_ABI_VERSION 4

#define MFCD_EINVAL (-1)   /* bad argument (null pointer, non-positive size, d out of range)   */
#define MFCD_EWORKSPACE (-2) /* workspace smaller than mfcd_*_workspace_bytes says             */
#define MFCD_EALIGN (-3)   /* a table pointer is not 4-byte aligned                            */
#define MFCD_EINDEX (-4)   /* a sample indexes outside [0,n) x [0,m)^2 (mfcd_check_samples)     */
#define MFCD_ERCCL (-5)    /* RCCL is not loadable in this process, or an RCCL call failed      */
#define MFCD_ESTATE (-6)   /* the workspace was not initialised (mfcd_train_workspace_init) or was
                              planned for other n, m, d                                         */

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
 * Workspace of mfcd_train_steps: caller-owned device memory, PLANNED once for a capacity and then reused by every
 * call that fits it (no per-call re-initialisation, no per-call allocation):
 *
 *   bytes = mfcd_train_workspace_bytes(N_cap, B, n, m, d)       N_cap = most samples one call will pass (an epoch)
 *   mfcd_train_workspace_init(ws, bytes, N_cap, B, n, m, d, stream)
 *        once after allocating it (256-byte aligned), and again after a reported abort; zero-fills it on `stream`
 *        and registers the host-side state that belongs to it (pinned staging ring, launch counter).  One workspace
 *        serves one model on one stream and is driven by one host thread at a time; use one per model / stream.
 *   mfcd_train_steps(..., ws, bytes, stream)                    any N <= N_cap with ceil(N/B) <= ceil(N_cap/B_plan),
 *        the same n, m, d; MFCD_ESTATE for an unregistered workspace or other n, m, d, MFCD_EWORKSPACE past the capacity
 *   mfcd_train_workspace_release(ws)                            before freeing it
 *
 * The first 4 bytes are an int32 status word: 0 = ok, 1 = a bounded in-kernel wait of the resident form expired
 * (U, V, m, v are then undefined).  It is STICKY: no call clears it, so an abort in any earlier call is still
 * visible when the host finally looks (after the stream has drained); only mfcd_train_workspace_init resets it.
 */
size_t mfcd_train_workspace_bytes(int64_t N_cap, int B, int n, int m, int d);
int mfcd_train_workspace_init(void *workspace, size_t workspace_bytes, int64_t N_cap, int B, int n, int m,
                              int d, void *stream);
int mfcd_train_workspace_release(void *workspace);

/*
 * Which form of the fused step mfcd_train_steps uses (process-wide):
 *   0 auto (default)  local where it applies; else resident where it applies and the call has at least
 *                     MFCD_TUNE_SHORT_CALL_STEPS (3) steps; else streaming
 *   1 streaming       one launch per optimiser step, state streamed through HBM (any size, any d)
 *   2 resident        one persistent launch per call, p/m/v held in registers, rows exchanged through
 *                     tagged 8-byte granules; needs d a power of two <= 256, 12*(n+m)*d bytes of state that
 *                     fit the register files, and every wave of the grid resident at once (asked of the
 *                     runtime's occupancy query per instantiation); MFCD_EINVAL from mfcd_train_steps otherwise
 *   3 local           tiny problems ((n+m)*d <= 8192, B <= 4096, any d): one persistent launch of ONE
 *                     workgroup, parameters in LDS, moments in registers, workgroup barriers only;
 *                     1.2-5.7x faster than the resident form wherever it applies; MFCD_EINVAL otherwise
 * All forms compute the same step (same summation order per row); results agree to fp32 rounding.
 */
int mfcd_set_train_path(int mode);

/*
 * Arithmetic flavour of Adam inside the RESIDENT and LOCAL forms, where the step is bound by vector-ALU cycles
 * (process-wide; the streaming form is HBM-bound and always uses the IEEE flavour):
 *   1 fast (default)  same operation order, but sqrt is the hardware v_sqrt_f32 (<= 1 ulp) and the two
 *                     divisions are reciprocal-multiply with one Newton correction (<= 1 ulp) instead of
 *                     the IEEE-rounded expansions; the update differs by a few ulp (~1e-10 per step)
 *   0 ieee            every operation IEEE-rounded, as ATen's CPU kernels
 * Either flavour is deterministic and keeps every parity test within the stated tolerances.
 */
int mfcd_set_resident_math(int fast);

/*
 * Tuning knobs for experiments and tests (process-wide; the defaults are the measured best and what every
 * published number uses).  They replace the environment variables the round-1 build read on every launch.
 */
#define MFCD_TUNE_RESIDENT_Q 1          /* 0 = smallest slice that fits (default); 1, 2, 4, 16 force it          */
#define MFCD_TUNE_RESIDENT_WPC 2        /* waves per CU for slices of <= 2 registers: 16 (default) or 8        */
#define MFCD_TUNE_RESIDENT_LOOKAHEAD 3  /* -1 auto (default: 4), 0 off, 2 .. 16 = depth of the window in steps */
#define MFCD_TUNE_RESIDENT_LDS_PAD 4    /* unused dynamic LDS per workgroup, bytes (default 0)                 */
#define MFCD_TUNE_RESIDENT_SPIN_LIMIT 5 /* polls before a wave gives up; 0 = default (2^22)                    */
#define MFCD_TUNE_SHORT_CALL_STEPS 6    /* "auto": calls of fewer steps stream instead (default 3)             */
#define MFCD_TUNE_UVT_WPE128 7          /* waves per SIMD of the d = 128 UV^T kernel: 2 (default) or 3        */
#define MFCD_TUNE_STREAM_CHUNKS 8       /* streaming form: 16-byte chunks per thread and array; 0 = auto      */
#define MFCD_TUNE_UVT_TARGET_WGS 9      /* UV^T pass: workgroups the column split aims for (default 512)       */
#define MFCD_TUNE_UVT_SPLIT 11          /* UV^T pass, d in {32, 64, 128}: 1 (default) = bf16x3 split product on the bf16 matrix pipe, 0 = fp32 MFMA */
#define MFCD_TUNE_RANK_SORT 12          /* Spearman kernel's sort: 1 (default) = block radix sort, 0 = bitonic network in LDS */
#define MFCD_TUNE_SHARD_PIPELINE 13     /* row-sharded native loop: 1 (default) = exchange of batch k+1 under step k when world > 1, 2 = always, 0 = strict chain */
#define MFCD_TUNE_UVT_MIN_STAGES 10     /* UV^T pass: column stages a workgroup sweeps at least (default 8)    */
int mfcd_set_tuning(int key, int64_t value);

/*
 * What mfcd_train_steps would do for a call of these sizes under the current settings (no launch; for
 * benchmarks and logs, so that they do not re-derive the plan).
 */
typedef struct mfcd_train_plan {
    int32_t form;                 /* 1 streaming, 2 resident, 3 local */
    int32_t resident_q;           /* registers per array and lane (slice = 64*q elements)        */
    int32_t resident_waves;       /* owner waves = slices                                         */
    int32_t resident_blocks;      /* workgroups of 4 waves                                        */
    int32_t resident_lookahead;   /* 0, 4 or 8                                                    */
    int32_t fast_math;            /* resident / local: Adam flavour                               */
    int32_t streaming_vec;        /* floats per lane and access (4 unless d % 4 != 0)             */
    int32_t streaming_chunks;
    int32_t streaming_blocks;     /* workgroups per step launch                                   */
    int32_t device_cus;
    int32_t reserved[6];
} mfcd_train_plan;
int mfcd_train_plan_query(int64_t N, int B, int n, int m, int d, int bf16_factors, mfcd_train_plan *out);

/*
 * Runs ceil(N/B) sequential optimiser steps on the device, consuming `samples` in order in
 * batches of B (last one short, divisor = actual batch size).  One step replaces
 * structure.py:847-851: zero_grad, forward, BCE(mean), backward (gather + scatter-add of row
 * gradients into U and V) and torch.optim.Adam.step() (coupled L2 weight decay, bias-corrected,
 * dense over every row; torch/optim/adam.py _single_tensor_adam).
 *   U,V,mU,vU,mV,vV  parameters and Adam moments (exp_avg, exp_avg_sq), updated in place
 *   step0            optimiser steps already taken (Adam's `step` before this call)
 *   loss_per_step[k] fp32 batch-mean BCE of step k (structure.py:852 loss.item())
 * No dense gradient is materialised.  The host is never made to wait for the device, with one bound: a workspace's
 * fifth queued call waits until its first has started (the pinned staging ring of per-step scalars has four slots).
 */
int mfcd_train_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                     const mfcd_sample *samples, int64_t N, int B, int64_t step0, int n, int m,
                     int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                     float *loss_per_step, void *workspace, size_t workspace_bytes, void *stream);

/*
 * bf16 factor storage (BASELINE.json configs[2]; the reference has no such mode, the rounding points are defined
 * here and in oracle/mfcd_oracle.c): U [n][d], V [m][d] are bf16 in HBM; each step reads them as such, does all
 * arithmetic and keeps the Adam moments in fp32, and rounds the updated parameters to the nearest bf16 (ties to
 * even) once, when they are written back.  Streaming form (20 bytes per element per step instead of 24) or, where
 * it applies, the resident form: the register copy is rounded after every update, the same
 * rounding point, so both forms agree with the oracle's definition.
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
 * Prepared calls: everything about a training call that does not change from call to call — the six table pointers,
 * the table shape and dtype, the batch size, the Adam hyper-parameters and the planned workspace — bound ONCE into a
 * handle, so that the per-call boundary is five scalars (round-2 review: a 20-step call spent ~24 us of its ~50 us in
 * argument marshalling on the host side of the boundary).  mfcd_train_call_run(handle, ...) is mfcd_train_steps /
 * mfcd_train_steps_bf16 with the bound arguments: same forms, same results, same stream semantics.  The handle borrows
 * the pointers (nothing is copied or owned); release it before the tables, the moments or the workspace go away, and
 * prepare a new one when any bound value changes (a learning-rate schedule, a re-planned workspace).
 * Replaces, on the reference side, the per-epoch body of train_model (structure.py:845-852).
 */
int mfcd_train_call_prepare(void *U, void *V, float *mU, float *vU, float *mV, float *vV, int bf16_factors, int B,
                            int n, int m, int d, double lr, double beta1, double beta2, double eps,
                            double weight_decay, void *workspace, size_t workspace_bytes, void **handle_out);
int mfcd_train_call_run(void *handle, const mfcd_sample *samples, int64_t N, int64_t step0, float *loss_per_step,
                        void *stream);
/*
 * Stage a LATER call of the same handle: run its prologue (stage table, sample translation, per-wave event lists of the
 * resident form) NOW on `side_stream`, into the workspace's second set of prologue regions, so that it overlaps the step
 * kernel of the call that is running.  The matching mfcd_train_call_run (same samples, N, step0, loss_per_step) then
 * launches its step kernel only.  The CALLER orders the streams: `side_stream` must not start this before the launch
 * two calls back has finished (it used the same set), and the main stream must wait for `side_stream` before the
 * matching run.  A no-op (returns 0) for calls that would not take the resident look-ahead form; a staged prologue that
 * is never run is simply overwritten by the next one.
 */
int mfcd_train_call_stage(void *handle, const mfcd_sample *samples, int64_t N, int64_t step0, float *loss_per_step,
                          void *side_stream);
int mfcd_train_call_release(void *handle);

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
 * `step` is Adam's 1-based step number.  workspace: mfcd_train_workspace_bytes(B,B,n,m,d) bytes of scratch
 * (no initialisation needed: this entry keeps no state in it).
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
/*
 * Backward of the model's forward for ANY loss on its output (autograd support of MatrixFactorization.forward,
 * structure.py:773-795): g[t] = dLoss/dx_t for B samples (x_t = the pre-sigmoid score); overwrites gradU [n][d],
 * gradV [m][d] with  dU[u]+=g(V[i]-V[j]), dV[i]+=g U[u], dV[j]-=g U[u]  accumulated in batch order (what
 * autograd's index_put_(accumulate=True) builds); rows no sample touches are 0.
 */
int mfcd_dense_grad_from_coefficients(const float *U, const float *V, const mfcd_sample *samples,
                                      const float *g, int B, int n, int m, int d, float *gradU,
                                      float *gradV, void *stream);
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
/* the same loop over bf16 factor tables (BASELINE configs[2]; moments, wire format and arithmetic stay fp32; equals
 * mfcd_train_steps_bf16's streaming form with batch_size = B * world) */
int mfcd_dp_train_steps_bf16(uint16_t *U, uint16_t *V, float *mU, float *vU, float *mV, float *vV,
                        const mfcd_sample *samples, int64_t N, int B, int rank, int world, int64_t step0, int n,
                        int m, int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                        float *loss_per_step, void *workspace, size_t workspace_bytes, void *comm, void *stream);

/*
 * Row-sharded training: STRONG scaling with the reference's batch size (structure.py:668, B = 64), results equal to
 * the single-GPU run.  Rank r of `world` holds rows [lo_r, hi_r) = mfcd_shard_rows(rows, r, world) of U and of V and of
 * their Adam moments (1/world of the state and of the dense Adam sweep); every rank sees the same sample stream.
 * Per optimiser step: the ranks write the rows of the batch they own into an exchange buffer xbuf[3][B][d] (role u, i,
 * j; zeros elsewhere), ONE all-reduce(sum) of that buffer taken as 32-bit integers reproduces every row bit for bit
 * (exactly one rank contributes non-zero bits per row), then the fused step runs over the shard in place, reading the
 * samples' rows from the buffer.  Every rank forms every sample's BCE term, so the step losses need no collective.
 * The arithmetic per element is that of mfcd_train_steps' streaming form: results are bit-identical to it.
 *
 *   mfcd_shard_pack / mfcd_shard_apply   the two halves of a step, for a caller that owns the collective
 *                                        (mfcd/dist.py over torch.distributed: RCCL, or gloo in the CPU tests)
 *   mfcd_shard_train_steps               the native loop (RCCL bound at run time, communicator from
 *                                        mfcd_dp_comm_create); table pointers are the rank's SHARDS.
 *                                        comm == NULL: the pointers are the FULL tables and this process plays every
 *                                        rank in turn (single-process rehearsal of any world size).
 *   mfcd_shard_pack_ahead                PIPELINED exchange (round 3): the rows of the NEXT batch as they will be
 *                                        after the step `step` that has not run yet — a row the current batch does
 *                                        not name changes in that step by the dense update with a zero sparse
 *                                        gradient, a pure function of its (p, m, v), computed here with the step
 *                                        kernel's own arithmetic (bit-identical) — so the all-reduce of batch k+1 can
 *                                        run underneath step k.  Only legal when no row of the next batch is named
 *                                        by the current one:
 *   mfcd_shard_collisions                flags[k] (uint8, device) = 1 when batch k+1 shares a row with batch k.
 * mfcd_shard_train_steps takes the pipelined chain by default when world > 1 (MFCD_TUNE_SHARD_PIPELINE 0 restores
 * pack -> all-reduce -> step on one stream everywhere, 2 pipelines on a one-rank communicator too): the collectives of
 * free pairs on a side stream, two exchange buffers, the strict chain — on the main stream, no stream hops — across
 * colliding pairs; it reads the collision flags on the host once per call (its one host wait).  Results are
 * bit-identical in both chains.
 * xbuf / workspace: mfcd_shard_workspace_bytes(N, B, d) bytes (the buffers come first, 3*B*d floats each).
 * u_lo..v_hi are GLOBAL row bounds of the shard; batch records name global rows.
 */
int mfcd_shard_rows(int rows, int rank, int world, int *lo, int *hi);
size_t mfcd_shard_workspace_bytes(int64_t N, int B, int d);
int mfcd_shard_pack(const float *U_shard, const float *V_shard, const mfcd_sample *batch, int Bk, int B, int d,
                    int u_lo, int u_hi, int v_lo, int v_hi, float *xbuf, void *stream);
int mfcd_shard_apply(float *U_shard, float *V_shard, float *mU, float *vU, float *mV, float *vV,
                     const mfcd_sample *batch, int Bk, int B, const float *xbuf, int64_t step, int d, int u_lo,
                     int u_hi, int v_lo, int v_hi, double lr, double beta1, double beta2, double eps,
                     double weight_decay, float *loss_terms, void *stream);
int mfcd_shard_collisions(const mfcd_sample *samples, int64_t N, int B, uint8_t *flags_dev, void *stream);
int mfcd_shard_pack_ahead(const float *U_shard, const float *V_shard, const float *mU, const float *vU,
                          const float *mV, const float *vV, const mfcd_sample *next_batch, int Bk, int B, int64_t step,
                          int d, int u_lo, int u_hi, int v_lo, int v_hi, double lr, double beta1, double beta2,
                          double eps, double weight_decay, float *xbuf, void *stream);
int mfcd_shard_train_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV,
                           const mfcd_sample *samples, int64_t N, int B, int rank, int world, int64_t step0, int n,
                           int m, int d, double lr, double beta1, double beta2, double eps, double weight_decay,
                           float *loss_per_step, void *workspace, size_t workspace_bytes, void *comm, void *stream);
/* the native loop over bf16 factor shards (exchange buffer, moments and arithmetic stay fp32; bit-identical to
 * mfcd_train_steps_bf16's streaming form with the same batch size) */
int mfcd_shard_train_steps_bf16(uint16_t *U, uint16_t *V, float *mU, float *vU, float *mV, float *vV,
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
 * The same pass computing only what the caller reads: what = 1 the per-row sums (compute_alpha_and_norm_ratios,
 * structure.py:982-1064; `scal` may be NULL), what = 2 the global sums (compute_reconstruction_error,
 * structure.py:940-952; `row_stats` may be NULL), what = 3 both (= mfcd_uvt_stats).  The epilogue's vector work shares
 * its lanes with the fp32 MFMA, so the narrower passes are faster (7 / 4 / 10 packed operations per pair of outputs).
 */
int mfcd_uvt_stats_select(const float *U, const float *V, const float *X, int n, int m, int d, double s,
                          int what, double *row_stats, double *scal, void *workspace,
                          size_t workspace_bytes, void *stream);

/*
 * The pass in row SLABS, for a ground truth kept as factors (X = A B^T at BASELINE C4 / C5 size would be 16 / 7.5 GiB;
 * SURVEY 8f N3): the caller forms rows [row0, row0 + nrows) of X with a plain library GEMM into X_slab [nrows][m] and
 * calls this once per slab.  U, V are the FULL tables (the column centring of U V^T needs every row of U);
 * row_stats_slab [nrows][8] are final for those rows; scal_slab [4] holds this slab's SHARE of the two global sums —
 * the caller adds the shares (f64, slab order) to get what mfcd_uvt_stats returns.
 * workspace: mfcd_uvt_slab_workspace_bytes(n, m, d, nrows) for the largest nrows used.
 */
size_t mfcd_uvt_slab_workspace_bytes(int n, int m, int d, int nrows);
int mfcd_uvt_stats_slab(const float *U, const float *V, const float *X_slab, int n, int m, int d, double s,
                        int what, int row0, int nrows, double *row_stats_slab, double *scal_slab,
                        void *workspace, size_t workspace_bytes, void *stream);

/*
 * k rows of UV^T: out[r][c] = sum_k U[row_ids[r]][k] * V[c][k]   (structure.py:389-392 computes
 * the full product to read two rows).  row_ids is a device array; a row id outside [0, n) yields a row of NaN
 * (the Python host validates the ids and raises IndexError, as U[row] would).
 */
int mfcd_uvt_rows(const float *U, const float *V, const int32_t *row_ids, int k, int n, int m,
                  int d, float *out, void *stream);

/*
 * BTL label generation on the device (replaces BTLPreferenceDataset._generate_labels,
 * structure.py:493-519, and the host-side packing of the records): for each of T triplets (int32 u, i, j)
 * score = sigmoid(scale * (X[u][i] - X[u][j])) in fp32, K Bernoulli(score) draws, and
 *   soft == 0: K consecutive mfcd_sample records per triplet with labels 0/1 (out holds T*K records)
 *   soft != 0: one record per triplet with z = mean of the K draws       (out holds T records)
 * X is dense [n][m] fp32, or NULL with the factors A [n][dx], B [m][dx] of X = A B^T given instead (C4/C5 sizes).
 * Randomness: Philox4x32-10 keyed by `seed`, counter = (triplet index, draw group): reproducible for a seed,
 * independent of launch geometry; NOT the reference's CPU generator stream (distributional parity only).
 * Indices are not validated here (mfcd_check_samples on the output does that).
 */
int mfcd_generate_labels(const int32_t *triplets, int64_t T, const float *X, int n, int m, const float *A,
                         const float *B, int dx, double scale, int K, int soft, uint64_t seed,
                         mfcd_sample *out, void *stream);

/*
 * Per-row Spearman rank correlation: rho[r] = Pearson correlation of the average ranks
 * (scipy.stats.rankdata semantics: ties share the mean of their positions; -0.0 ties with +0.0) of row r of A
 * [rows][lda] and row r of X [rows][ldx], m <= mfcd_spearman_max_columns() (20448: BASELINE configs[4]'s 20000 fits) columns each.  Replaces the
 * reference's Python loop of scipy.stats.spearmanr over the rows of the centred U V^T and X (structure.py:1023-1031);
 * the caller forms the rows of U V^T with a plain library GEMM.  One workgroup per row: bitonic sort in LDS, exact
 * integer sums of the doubled centred ranks, rho in f64 (NaN for a constant row, as scipy).  Deterministic.
 */
int mfcd_spearman_max_columns(void);
int mfcd_spearman_rows(const float *A, int64_t lda, const float *X, int64_t ldx, int rows, int m,
                       double *rho, void *stream);
/*
 * The same for rows of ANY length (BASELINE configs[3]: m = 65536 items do not fit a workgroup's LDS): blocks of rows
 * are sorted in global memory by one device-wide segmented radix sort of (key, column) pairs per matrix, then one
 * workgroup per row forms the run-averaged ranks and the exact integer sums as above (bit-identical to
 * mfcd_spearman_rows where both apply).  workspace: mfcd_spearman_long_workspace_bytes(rows, m) bytes (28 bytes per
 * element of a row block of about 256 MiB; 0 = sizes out of range).
 */
size_t mfcd_spearman_long_workspace_bytes(int rows, int m);
int mfcd_spearman_rows_long(const float *A, int64_t lda, const float *X, int64_t ldx, int rows, int m, double *rho,
                            void *workspace, size_t workspace_bytes, void *stream);

/*
 * Triplet sampling on the device (replaces the per-attempt rejection loops of generation_data.py:16-224
 * — choose_items_random 16-26, _by_proximity 29-43, _by_margin 46-84, _by_variance 87-99, _by_popularity 103-128,
 * _by_svd_projection 164-174 (the draw loop), _top_k 205-219).  One call evaluates `attempts` attempts
 * [attempt0, attempt0 + attempts) of the law and keeps what the reference's loop keeps: the first `want` triplets, in
 * attempt order, that pass the law's filter, are not among `barred_keys` and were not produced by an earlier attempt.
 * A triplet's key is (u * m + i) * m + j (int64); barred_keys holds the caller's `exclude` set plus the keys returned
 * by earlier calls of the same request, in any order.
 *
 *   law        MFCD_LAW_UNIFORM   u ~ U[0,n), i, j ~ U[0,m)                         (random; margin with use_margin)
 *              MFCD_LAW_ITEM_CDF  i, j from the item law whose cumulative sums are cdf[m] (f64, cdf[m-1] == 1):
 *                                 pair_rule 0 = numpy choice(m, size=2, replace=False, p) (popularity),
 *                                 pair_rule 1 = sequential draw without replacement (torch.multinomial; variance)
 *              MFCD_LAW_LISTS     i = list_i[u * list_row_stride + U[0,k)], j likewise from list_j; pair_rule 1 makes
 *                                 the two positions distinct (top_k, svd); list_row_stride = k for per-user tables
 *                                 (proximity, top_k), 0 for one list shared by all users (svd)
 *   users      NULL (u over all n users) or n_users user ids to draw u from (svd: the top users)
 *   use_margin keep only |X[u][i] - X[u][j]| <= margin (fp32 difference, as generation_data.py:72-73); X dense
 *              [n][m] fp32, or NULL with the factors A [n][dx], B [m][dx] of X = A B^T
 * Attempts with i == j are rejected in every law.  Outputs (device): triplets_out [want][3] int32 and keys_out [want]
 * in attempt order; counts_out[0] = triplets written (<= want), counts_out[1] = attempts consumed (the attempt that
 * completed the request + 1, or `attempts`).  Randomness: Philox4x32-10 keyed by `seed`, counter = (attempt index, draw
 * group): reproducible, independent of launch geometry and of how a request is cut into calls; NOT the reference's
 * generator streams — distributional parity (the seeded host forms in generation_data.py replay those).
 * attempts + n_barred < 2^32 - 1.  workspace: mfcd_sample_workspace_bytes(attempts, n_barred) bytes (0 = bad sizes).
 */
#define MFCD_LAW_UNIFORM 0
#define MFCD_LAW_ITEM_CDF 1
#define MFCD_LAW_LISTS 2

typedef struct mfcd_sampler {
    int32_t law, n, m, pair_rule;
    const double *cdf;
    const int32_t *list_i, *list_j;
    int32_t k, list_row_stride;
    const int32_t *users;
    int32_t n_users, use_margin;
    double margin;
    const float *X, *A, *B;
    int32_t dx, reserved;
} mfcd_sampler;

size_t mfcd_sample_workspace_bytes(int64_t attempts, int64_t n_barred);
int mfcd_sample_triplets(const mfcd_sampler *law, const int64_t *barred_keys, int64_t n_barred, int64_t attempt0,
                         int64_t attempts, uint64_t seed, int64_t want, int32_t *triplets_out, int64_t *keys_out,
                         int64_t *counts_out, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Register-resident optimiser steps for states of up to 8 388 608 elements at d = 64 (round 3, opt-in: BASELINE
 * configs[3], n = m = 65536, d = 64, is exactly 1 024 SIMDs x 64 lanes x 128 rows).  The step of mfcd_train_steps
 * (structure.py:845-852) with the arithmetic of its streaming form (bit-identical results), as ONE persistent launch of
 * two waves per SIMD: exp_avg / exp_avg_sq of a wave's 64 rows in 128 registers per lane, the
 * parameters in LDS, rows exchanged per step through tagged granules (publish right before use).  fp32 tables, d == 64,
 * n + m <= 131072, B <= 64; a batch may name at most mfcd_train_big_slots() (8) distinct rows of one wave's 64-row slice (else status 2 and the call is void:
 * streams that concentrate on a few rows belong to the streaming form).  The call waits once on the host (its per-step
 * table is copied from the stack).  workspace: mfcd_train_big_workspace_bytes(N, B) — status word (int32, first 4
 * bytes; mfcd_train_big_status reads it: 0 ok, 1 a bounded wait expired, 2 too many rows of one wave in a batch), the
 * per-step scalars, N loss terms and the 1536-byte mailbox slot of every sample.
 */
size_t mfcd_train_big_workspace_bytes(int64_t N, int B);
int mfcd_train_steps_big(float *U, float *V, float *mU, float *vU, float *mV, float *vV, const mfcd_sample *samples,
                         int64_t N, int B, int64_t step0, int n, int m, int d, double lr, double beta1, double beta2,
                         double eps, double weight_decay, float *loss_per_step, void *workspace, size_t workspace_bytes,
                         void *stream);
int mfcd_train_big_status(const void *workspace, int *status_out, void *stream);
/* pre-check of a sample stream for the form above: *max_out_dev (device int32) = the largest number of row references
 * one wave's 64-row slice receives from one batch (an upper bound of the distinct rows); the form takes the call when
 * it is <= mfcd_train_big_slots(). */
int mfcd_train_big_slots(void);
int mfcd_train_big_check(const mfcd_sample *samples, int64_t N, int B, int n, int m, int *max_out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MFCD_H */
