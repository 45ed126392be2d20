// Shared between the streaming (train.hip) and resident (resident.hip) forms of the optimiser step.
#pragma once
#include "common.h"

struct AdamStatic {
    float w1;   // 1 - beta1
    float b2;   // beta2
    float w2;   // 1 - beta2
    float eps;
    float wd;
};

// Per-step bias-correction scalars, computed on the host in f64 exactly as Python does in
// torch/optim/adam.py (1 - beta**step, lr / bc1, bc2 ** 0.5) and rounded to fp32 where ATen would.
struct StepScalars {
    float neg_step_size;  // -(lr / (1 - beta1^t))
    float bc2_sqrt;       // sqrt(1 - beta2^t)
    float inv_bc2_sqrt;   // 1 / sqrt(1 - beta2^t), rounded from f64 (used by the fast flavour only)
    float pad;
};

// One element of torch.optim.Adam's single-tensor step (coupled L2).  The operation sequence is pinned
// (explicit fmaf, contraction off) so that every inlined copy rounds identically — the resident kernel relies on
// a rolled-forward copy of a row matching the in-place update bit for bit — and it mirrors ATen's CPU kernels:
//   grad.add(param, alpha=wd)            -> fma(wd, p, g)                    (vec::fmadd)
//   exp_avg.lerp_(grad, 1-b1)            -> fma(1-b1, g - m, m)              (weight < 0.5 branch, vec::fmadd)
//   exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2) -> (v*b2) + ((1-b2)*g)*g          (no fma)
//   denom = sqrt(v)/bc2_sqrt + eps ;  param.addcdiv_(m, denom, -step_size)  -> p + ((-step_size)*m)/denom
__device__ __forceinline__ void adam_update(float &p, float &m1, float &m2, float gsparse, const AdamStatic &ac,
                                            const StepScalars &sc)
{
#pragma clang fp contract(off)
    const float g = __builtin_fmaf(ac.wd, p, gsparse);
    m1 = __builtin_fmaf(ac.w1, g - m1, m1);
    const float v_scaled = m2 * ac.b2;
    const float gg = (ac.w2 * g) * g;
    m2 = v_scaled + gg;
    const float den = sqrtf(m2) / sc.bc2_sqrt + ac.eps;
    const float num = sc.neg_step_size * m1;
    p = p + num / den;
}

// Fast flavour of the same update for the register-resident kernel, where the step is bound by VALU cycles:
// identical op order, but the square root is the hardware v_sqrt_f32 (<= 1 ulp) and the two divisions are a
// reciprocal multiply with one Newton correction of the quotient (q = q0 + (a - b*q0)*r, <= 1 ulp, almost always
// the correctly rounded quotient) instead of the ~52-cycle IEEE expansions.  The update term therefore differs
// from the IEEE flavour by at most a few ulp (~1e-10 absolute per step at lr = 1e-3).
__device__ __forceinline__ float div_newton(float a, float b, float r /* ~ 1/b */)
{
#pragma clang fp contract(off)
    const float q0 = a * r;
    const float e = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(e, r, q0);
}

__device__ __forceinline__ void adam_update_fast(float &p, float &m1, float &m2, float gsparse, const AdamStatic &ac,
                                                 const StepScalars &sc)
{
#pragma clang fp contract(off)
    const float g = __builtin_fmaf(ac.wd, p, gsparse);
    m1 = __builtin_fmaf(ac.w1, g - m1, m1);
    const float v_scaled = m2 * ac.b2;
    const float gg = (ac.w2 * g) * g;
    m2 = v_scaled + gg;
    const float sq = __builtin_amdgcn_sqrtf(m2);
    const float den = div_newton(sq, sc.bc2_sqrt, sc.inv_bc2_sqrt) + ac.eps;
    const float num = sc.neg_step_size * m1;
    p = p + div_newton(num, den, __builtin_amdgcn_rcpf(den));
}

// The fast flavour on TWO elements at once with gfx950's packed fp32 instructions (v_pk_fma_f32, v_pk_mul_f32,
// v_pk_add_f32: one issue slot for both halves).  Every packed operation rounds each half exactly like its scalar
// counterpart and the operation sequence is the one above, so the results are bit-identical to two calls of
// adam_update_fast; only the square roots and reciprocals stay scalar.  20 issue slots per pair instead of 36.
typedef float mfcd_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mfcd_f2 f2_splat(float x) { return (mfcd_f2){x, x}; }
__device__ __forceinline__ mfcd_f2 f2_fma(mfcd_f2 a, mfcd_f2 b, mfcd_f2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ mfcd_f2 div_newton2(mfcd_f2 a, mfcd_f2 b, mfcd_f2 r)
{
#pragma clang fp contract(off)
    const mfcd_f2 q0 = a * r;
    const mfcd_f2 e = f2_fma(-b, q0, a);
    return f2_fma(e, r, q0);
}

__device__ __forceinline__ void adam_update_fast2(float &pa, float &pb, float &m1a, float &m1b, float &m2a, float &m2b,
                                                  float ga, float gb, const AdamStatic &ac, const StepScalars &sc)
{
#pragma clang fp contract(off)
    mfcd_f2 p = {pa, pb}, m1 = {m1a, m1b}, m2 = {m2a, m2b};
    const mfcd_f2 g = f2_fma(f2_splat(ac.wd), p, (mfcd_f2){ga, gb});
    m1 = f2_fma(f2_splat(ac.w1), g - m1, m1);
    const mfcd_f2 v_scaled = m2 * f2_splat(ac.b2);
    const mfcd_f2 gg = (f2_splat(ac.w2) * g) * g;
    m2 = v_scaled + gg;
    const mfcd_f2 sq = {__builtin_amdgcn_sqrtf(m2.x), __builtin_amdgcn_sqrtf(m2.y)};
    const mfcd_f2 den = div_newton2(sq, f2_splat(sc.bc2_sqrt), f2_splat(sc.inv_bc2_sqrt)) + f2_splat(ac.eps);
    const mfcd_f2 num = f2_splat(sc.neg_step_size) * m1;
    const mfcd_f2 rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    p = p + div_newton2(num, den, rc);
    pa = p.x; pb = p.y; m1a = m1.x; m1b = m1.y; m2a = m2.x; m2b = m2.y;
}

// Q elements of one thread: pairs through the packed form when FAST && PACKED, scalar otherwise.
template <bool FAST, int Q, bool PACKED>
__device__ __forceinline__ void adam_update_q(float (&p)[Q], float (&m1)[Q], float (&m2)[Q], const float (&g)[Q],
                                              const AdamStatic &ac, const StepScalars &sc);

template <bool FAST>
__device__ __forceinline__ void adam_update_t(float &p, float &m1, float &m2, float gsparse, const AdamStatic &ac,
                                              const StepScalars &sc)
{
    if constexpr (FAST) adam_update_fast(p, m1, m2, gsparse, ac, sc);
    else adam_update(p, m1, m2, gsparse, ac, sc);
}

template <bool FAST, int Q, bool PACKED>
__device__ __forceinline__ void adam_update_q(float (&p)[Q], float (&m1)[Q], float (&m2)[Q], const float (&g)[Q],
                                              const AdamStatic &ac, const StepScalars &sc)
{
    if constexpr (FAST && PACKED && Q % 2 == 0) {
#pragma unroll
        for (int q = 0; q < Q; q += 2)
            adam_update_fast2(p[q], p[q + 1], m1[q], m1[q + 1], m2[q], m2[q + 1], g[q], g[q + 1], ac, sc);
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) adam_update_t<FAST>(p[q], m1[q], m2[q], g[q], ac, sc);
    }
}

namespace mfcd_detail {

// ---- local form (local.hip): one workgroup, parameters in LDS ----
constexpr int64_t kLocalMaxElems = 8192;   // (n+m)*d: above this one CU's vector ALU is slower than the multi-CU resident form
bool local_applies(int64_t N, int B, int n, int m, int d);
int launch_local_steps(float *U, float *V, float *mU, float *vU, float *mV, float *vV, const mfcd_sample *samples,
                       int64_t N, int B, int n, int m, int d, const StepScalars *sc_dev, const AdamStatic &ac,
                       float *loss_terms, int K, hipStream_t st);

struct ResidentPlan {
    bool ok;
    int Q, NW, blocks;
    int lookahead;   // 0, 4 or 8: the instantiation the launch uses
    bool fast_math;
    bool bf16;       // bf16 factor tables
    int tshift;      // look-ahead form: log2(steps per chunk of the per-wave event lists)
};

constexpr unsigned kSpinLimitDefault = 1u << 22;  // polls before a wave gives up (~seconds); sets status = 1

// process-wide tuning knobs (mfcd_set_tuning; experiments and tests only, defaults are the measured best)
struct Tuning {
    int resident_q = 0;          // 0 = smallest slice that fits; else force Q
    int resident_wpc = 16;       // waves per CU bound for Q <= 2
    int lookahead = -1;          // -1 auto (4, or 0 for tiny tables), 0 off, else the window depth 2 .. 16
    int lds_pad = 0;             // unused dynamic LDS per workgroup (bytes)
    unsigned spin_limit = kSpinLimitDefault;   // polls before a wave gives up and sets the status word
    int stream_chunks = 0;       // streaming form: 16-byte chunks per thread and array (0 = by table size)
    int short_call_steps = 3;    // "auto": calls of fewer steps than this stream (one launch per step) instead of
                                 // paying the persistent launch's fixed cost
    int shard_pipeline = 1;      // row-sharded loop: 1 = exchange of batch k+1 under step k where no row is shared
};
extern Tuning g_tune;

extern int g_resident_math;

int set_uvt_wpe128(int v);   // uvt.hip
int set_uvt_split(int v);    // uvt.hip
int set_rank_sort(int v);    // rank.hip
int set_uvt_target_wgs(int v);   // uvt.hip
int set_uvt_min_stages(int v);   // uvt.hip

// ev_tshift: chunk length of the event lists the WORKSPACE was laid out for (ResidentEvents::tshift; 0 = no lists, the
// look-ahead form is then not planned)
ResidentPlan plan_resident(int64_t N, int B, int n, int m, int d, int num_cus, bool bf16 = false, int ev_tshift = -1);
int resident_lookahead(int64_t N, int B, int n, int m);

// Geometry of the per-wave event lists of the look-ahead form (resident_kernel.h) for tables of this shape and batch
// size: a function of the shape alone (the smallest slice that fits, never of a tuning knob), so that a workspace
// planned once serves every call.  tshift = 0: the lists do not apply (a wave would see a hit nearly every step).
struct ResidentEvents {
    int tshift;        // log2(steps per chunk), 4 .. 8
    int waves;         // owner waves the arrays are laid out for
    int rows_per_wave;
};
ResidentEvents resident_events(int B, int n, int m, int d, int num_cus);
constexpr int kResidentEventCap = 64;          // entries per (wave, chunk) list = one per lane (resident_kernel.h)
constexpr int kResidentEventLook = 16;         // deepest look-ahead window: boundary copies per chunk
inline int64_t resident_event_chunks(int64_t K, int tshift) { return (K >> tshift) + 2; }

// stage tables of up to this many bytes travel in the prologue kernel's own argument segment: the host buffer they are
// built in is read at launch time only (no pinned slot, no event)
size_t train_inline_stage_bytes();

// One kernel in front of a resident / local launch: pinned staging slot -> workspace, and (xs != nullptr) the
// translated samples + (look > 0) the per-wave event lists of the resident form.
int launch_train_prologue(const void *stage_host, const void *stage_host_devview, void *stage_dev, size_t stage_bytes,
                          const mfcd_sample *samples, int64_t N, int B, int n, int m, int rows_per_wave, int tshift,
                          int look, int64_t nch_cap, mfcd_sample *xs, unsigned *ev_cnt, void *ev_ent, hipStream_t st);

// cold_dev: device copy of ResidentCold (resident_kernel.h: table pointers, status word, spin limit, touch strings);
// xs: the call's samples translated to virtual row ids
int launch_resident_steps(const ResidentPlan &pl, const void *cold_dev, const mfcd_sample *xs, int64_t N, int B, int n,
                          int m, int d, const StepScalars *sc_dev, const AdamStatic &ac, unsigned long long *mailbox,
                          unsigned tag_base, void *loss_terms, unsigned long long *dbg, int K, hipStream_t st);
constexpr int kResidentMaxWaves = 4096;   // 256 CUs x 16 waves: upper bound of ResidentPlan::NW (workspace sizing)

}  // namespace mfcd_detail
