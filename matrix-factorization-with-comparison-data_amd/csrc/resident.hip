// resident.hip — persistent, register-resident form of the fused optimiser step (gfx950).
//
// Regime: the whole Adam state (p, m, v of U and V: 12*(n+m)*d bytes) fits in the chip's register
// files (256 CUs x 512 KiB).  Then one launch runs ALL optimiser steps of a call (an epoch), the
// state never leaves VGPRs between steps, and HBM sees it twice per launch instead of twice per step.
//
// Decomposition (wave = owner, no workgroup barrier, no grid barrier):
//   * the concatenated table [U;V] is cut into NW contiguous slices of EW = 64*Q elements (a multiple
//     of d, so rows are never split); wave w keeps p, m, v of slice w in 3*Q VGPRs per lane
//     (lane l, register q <-> element w*EW + q*64 + l);
//   * per step a wave scans the batch (B 16-byte records).  Samples that touch one of ITS rows are its
//     "hits"; for a hit it needs the two/three rows of that sample owned by other waves, as they are
//     after the previous step.  Owners publish exactly those rows right after their own update:
//     each value travels as ONE naturally aligned 8-byte {tag = step+1, fp32 value} granule written
//     with an agent-scope relaxed atomic store (write-through, sc1) into a per-sample mailbox slot;
//     consumers poll the granules with agent-scope relaxed loads until every tag matches
//     (cdna_hip_programming.md Guideline 16, form R2: the data is the flag; no fence, no flag word).
//     A wave's own row comes straight from its registers.
//   * a wave without a hit in a step waits for nobody: it applies the weight-decay gradient, runs Adam
//     on its registers and moves on, so producers run ahead of consumers.  Dependencies only point to
//     earlier steps, every wave is resident (grid <= what the chip holds, checked on the host), every
//     spin is bounded and observes a shared abort word: no deadlock, bounded run time.
//   * summation order per row is batch order, as in the streaming kernel; results are deterministic.
//
// The arithmetic is the same as train.hip's (common.h helpers): structure.py:847-851 per step.
#include <cstdlib>
#include <vector>

#include "resident_kernel.h"

// One launcher per factor width lives in its own translation unit (resident_inst.hip, -DMFCD_RES_D=<d>);
// a width that was not built is simply absent (weak symbol) and the streaming form is used for it.
#define MFCD_DECL(d) \
    extern "C" int mfcd_resident_launch_d##d(const mfcd_detail::ResidentArgs *, int, int, void *) __attribute__((weak));
MFCD_DECL(2) MFCD_DECL(4) MFCD_DECL(8) MFCD_DECL(16) MFCD_DECL(32) MFCD_DECL(64) MFCD_DECL(128) MFCD_DECL(256)
#undef MFCD_DECL

namespace mfcd_detail {

int g_resident_math = 1;   // mfcd_set_resident_math: 1 = fast flavour (default), 0 = IEEE-rounded

typedef int (*ResidentLauncher)(const ResidentArgs *, int, int, void *);

static ResidentLauncher launcher_for(int d)
{
    switch (d) {
        case 2: return mfcd_resident_launch_d2;
        case 4: return mfcd_resident_launch_d4;
        case 8: return mfcd_resident_launch_d8;
        case 16: return mfcd_resident_launch_d16;
        case 32: return mfcd_resident_launch_d32;
        case 64: return mfcd_resident_launch_d64;
        case 128: return mfcd_resident_launch_d128;
        case 256: return mfcd_resident_launch_d256;
        default: return nullptr;
    }
}

// Resident-path plan: Q registers per array, NW owner waves, or ok=false when the path does not apply.
// Every wave of the grid must be resident at once (waves wait on each other), so the wave count is bounded
// by what every instantiation's register allocation admits per CU (kernel-resource-usage: Q<=2 ~70 VGPRs ->
// 4 workgroups of 4 waves; Q=4 ~85, Q=16 ~176 VGPRs -> 2 workgroups of 4 waves).
ResidentPlan plan_resident(int n, int m, int d, int num_cus)
{
    ResidentPlan pl{};
    pl.ok = false;
    if (d < 2 || d > 256 || (d & (d - 1)) != 0 || num_cus <= 0 || !launcher_for(d)) return pl;
    const int64_t T = (int64_t)(n + m) * d;
    // tuning override for experiments (tools/): MFCD_RESIDENT_Q forces the slice size
    const char *eq = getenv("MFCD_RESIDENT_Q");
    const int forced_q = eq ? atoi(eq) : 0;
    // Q <= 2 (<= ~70 VGPRs): 4 workgroups of 4 waves per CU are resident and hide each other's hand-off
    // latency (measured +7 % at C2 over 2 per CU); larger slices stay at 2 workgroups per CU.
    const char *ew = getenv("MFCD_RESIDENT_WPC");   // experiment knob: waves per CU bound for Q <= 2 (8 or 16)
    const int wpc = ew ? atoi(ew) : 16;
    static const int kQ[4] = {1, 2, 4, 16};
    for (int qi = 0; qi < 4; ++qi) {
        const int Q = kQ[qi];
        if ((64 * Q) % d != 0 || (forced_q && Q != forced_q)) continue;
        const int64_t nw = (T + 64 * (int64_t)Q - 1) / (64 * Q);
        if (nw <= (int64_t)num_cus * (Q <= 2 ? wpc : 8) && nw <= kResidentMaxWaves) {
            pl.ok = true;
            pl.Q = Q;
            pl.NW = (int)nw;
            pl.blocks = (pl.NW + 3) / 4;
            return pl;
        }
    }
    return pl;
}

// ---- row order ----
// The kernel only sees VIRTUAL row ids: the order is written once per launch as two tables (vrow: table row -> virtual
// row, inv: the inverse), the samples are translated with them, and the slice load / store goes through `inv`.  The
// order itself is the static interleave of RowMap (resident_kernel.h).  A per-launch order rebuilt from the launch's own
// touch counts (rows sorted by count, dealt to the waves in snake order: 49 +- 2 hits per wave instead of 49 +- 7) was
// measured and dropped: the kernel ran within 1 % of the static order and the sort cost 63 us per launch.
__global__ __launch_bounds__(256) void resident_order_kernel(int n, int m, int *__restrict__ vrow, int *__restrict__ inv)
{
    const RowMap rm = make_row_map(n, m);
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n + m) return;
    const bool item = r >= n;
    const int vr = item ? rm.vrow_v(r - n) : rm.vrow_u(r);
    vrow[r] = vr;
    inv[vr] = (item ? (int)0x80000000 : 0) | (item ? r - n : r);
}

// Translates the call's samples to virtual row ids (xs) and builds the per-wave touch strings:
// touch[w][k >> 5] bit (k & 31) = batch k holds a sample with a row of wave w (rows_per_wave consecutive virtual rows
// per wave).  One thread per sample; the strings are zero-filled before.
__global__ __launch_bounds__(256) void resident_translate_kernel(const mfcd_sample *__restrict__ samples, int64_t N,
                                                                 int B, int n, const int *__restrict__ vrow,
                                                                 int rows_per_wave, int KW, int want_touch,
                                                                 mfcd_sample *__restrict__ xs,
                                                                 unsigned *__restrict__ touch)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= N) return;
    mfcd_sample s = samples[t];
    s.u = vrow[s.u];
    s.i = vrow[n + s.i];
    s.j = vrow[n + s.j];
    xs[t] = s;
    if (!want_touch) return;
    const int k = (int)(t / B);
    const unsigned bit = 1u << (k & 31);
    const int word = k >> 5;
    // one string per wave, or (rows_per_wave <= 4) one per virtual row: the kernel's ROWWIN
    const int gr = rows_per_wave <= 4 ? 1 : rows_per_wave;
    const int wu = s.u / gr, wi = s.i / gr, wj = s.j / gr;
    atomicOr(touch + (size_t)wu * KW + word, bit);
    if (wi != wu) atomicOr(touch + (size_t)wi * KW + word, bit);
    if (wj != wi && wj != wu) atomicOr(touch + (size_t)wj * KW + word, bit);
}

int resident_touch_words(int K) { return (K + 31) / 32 + 3; }

size_t resident_aux_bytes(int64_t N, int n, int m, int K)
{
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return al(sizeof(int) * (size_t)(n + m)) * 2 + al(sizeof(mfcd_sample) * (size_t)(N > 0 ? N : 1)) +
           al(sizeof(unsigned) * (size_t)(4 * kResidentMaxWaves + 4) * (size_t)resident_touch_words(K));   // <= 4 rows/wave
}

int launch_resident_steps(const ResidentPlan &pl, const void *cold_dev,
                          const mfcd_sample *samples, int64_t N, int B, int n, int m, int d, const StepScalars *sc_dev,
                          const AdamStatic &ac, unsigned long long *mailbox, float *loss_terms, int *status,
                          unsigned long long *dbg, void *aux, int K, hipStream_t st)
{
    if (!aux) return MFCD_EINVAL;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char *ap = (char *)aux;                     // carve-up of the aux region (resident_aux_bytes)
    int *vrow = (int *)ap; ap += al(sizeof(int) * (size_t)(n + m));
    int *inv = (int *)ap;  ap += al(sizeof(int) * (size_t)(n + m));
    mfcd_sample *xs = (mfcd_sample *)ap; ap += al(sizeof(mfcd_sample) * (size_t)(N > 0 ? N : 1));
    unsigned *touch = (unsigned *)ap;

    ResidentArgs a;
    a.dbg = dbg;
    a.cold = (const ResidentCold *)cold_dev;
    a.samples = xs; a.sc = sc_dev; a.mailbox = mailbox; a.loss_terms = loss_terms; a.status = status;
    a.N = N; a.B = B; a.n = n; a.m = m; a.K = K; a.NW = pl.NW; a.ac = ac;
    const char *la = getenv("MFCD_RESIDENT_LOOKAHEAD");  // tuning/test override: 0 disables look-ahead publishing
    a.lookahead = la ? atoi(la) : 4;
    if (N + 64 * 16 >= ((int64_t)1 << 31)) a.lookahead = 0;   // the look-ahead form indexes samples with 32 bits
    // tiny tables: a batch touches so large a share of the rows that nearly every row recurs inside the window
    // and each publish takes the deferred (slow) path; publishing right before use is faster there
    if (!la && (int64_t)(n + m) < (int64_t)96 * B) a.lookahead = 0;
    a.inv = inv;
    a.touch = touch;
    a.KW = resident_touch_words(K);
    // row order tables, translated samples, touch strings
    const int rpw = 64 * pl.Q / d;
    const unsigned sblocks = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(resident_order_kernel, dim3((unsigned)((n + m + 255) / 256)), dim3(256), 0, st, n, m, vrow, inv);
    const int want_touch = a.lookahead > 0 && B <= 64;   // the look-ahead form reads the per-wave touch strings
    if (want_touch)   // one string per wave, or per row for waves of up to 4 rows (resident_kernel.h ROWWIN)
        MFCD_HIP_TRY(hipMemsetAsync(touch, 0, sizeof(unsigned) * (size_t)pl.NW * (rpw <= 4 ? rpw : 1) * a.KW, st));
    hipLaunchKernelGGL(resident_translate_kernel, dim3(sblocks), dim3(256), 0, st, samples, N, B, n, vrow, rpw, a.KW,
                       want_touch, xs, touch);
    const char *fm = getenv("MFCD_RESIDENT_MATH");   // experiment knob: "ieee" / "fast" overrides mfcd_set_resident_math
    a.fast_math = fm ? (fm[0] == 'f') : g_resident_math;
    const char *lp = getenv("MFCD_RESIDENT_LDS_PAD");   // experiment knob (bytes)
    a.lds_pad = lp ? atoi(lp) : 0;
    ResidentLauncher fn = launcher_for(d);
    if (!fn) return MFCD_EINVAL;
    return fn(&a, pl.Q, pl.blocks, (void *)st);
}

}  // namespace mfcd_detail
