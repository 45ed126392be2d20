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
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "resident_kernel.h"

// One launcher per factor width lives in its own translation unit (resident_inst.hip, -DMFCD_RES_D=<d>);
// a width that was not built is simply absent (weak symbol) and the streaming form is used for it.
#define MFCD_DECL(d)                                                                                                  \
    extern "C" int mfcd_resident_launch_d##d(const mfcd_detail::ResidentArgs *, int, int, void *) __attribute__((weak)); \
    extern "C" int mfcd_resident_occupancy_d##d(int, int, int, int, int) __attribute__((weak));
MFCD_DECL(2) MFCD_DECL(4) MFCD_DECL(8) MFCD_DECL(16) MFCD_DECL(32) MFCD_DECL(64) MFCD_DECL(128) MFCD_DECL(256)
#undef MFCD_DECL

namespace mfcd_detail {

int g_resident_math = 1;   // mfcd_set_resident_math: 1 = fast flavour (default), 0 = IEEE-rounded
Tuning g_tune;             // mfcd_set_tuning

typedef int (*ResidentLauncher)(const ResidentArgs *, int, int, void *);
typedef int (*ResidentOccupancy)(int, int, int, int, int);

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

static ResidentOccupancy occupancy_for(int d)
{
    switch (d) {
        case 2: return mfcd_resident_occupancy_d2;
        case 4: return mfcd_resident_occupancy_d4;
        case 8: return mfcd_resident_occupancy_d8;
        case 16: return mfcd_resident_occupancy_d16;
        case 32: return mfcd_resident_occupancy_d32;
        case 64: return mfcd_resident_occupancy_d64;
        case 128: return mfcd_resident_occupancy_d128;
        case 256: return mfcd_resident_occupancy_d256;
        default: return nullptr;
    }
}

// Look-ahead depth the launch of a call with these sizes uses (0 = publish right before use).
int resident_lookahead(int64_t N, int B, int n, int m)
{
    int la = g_tune.lookahead >= 0 ? g_tune.lookahead : 4;
    if (B > 64) la = 0;                                      // the look-ahead form is the reference's B <= 64
    if (N + 64 * 16 >= ((int64_t)1 << 31)) la = 0;           // it indexes samples with 32 bits
    // tiny tables: a batch touches so large a share of the rows that nearly every row recurs inside the window
    // and each publish takes the deferred (slow) path; publishing right before use is faster there
    if (g_tune.lookahead < 0 && (int64_t)(n + m) < (int64_t)96 * B) la = 0;
    return la > 16 ? 16 : (la < 0 ? 0 : la);
}

// Workgroups of the instantiation (d, Q, look, fast) one CU holds at once, as the runtime reports it for the actual
// code object (register and LDS use), capped by the hardware's own admission rule for 256-thread workgroups
// (MI355X_MICROARCH.md, Residency: min(API, 8, ...)); cached per instantiation.  0 = unknown (no device / query failed).
static int resident_blocks_per_cu(int d, int Q, int look, int fast, int bf16, int lds_pad)
{
    static std::mutex mu;
    static std::vector<std::pair<long long, int>> cache;
    const long long key = (((((long long)d * 64 + Q) * 16 + look) * 2 + fast) * 2 + bf16) * 262144 + lds_pad;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &e : cache)
        if (e.first == key) return e.second;
    int occ = 0;
    if (ResidentOccupancy fn = occupancy_for(d)) occ = fn(Q, look, fast, bf16, lds_pad);
    if (occ > 8) occ = 8;
    if (occ < 0) occ = 0;
    cache.emplace_back(key, occ);
    return occ;
}

// Resident-path plan: Q registers per array, NW owner waves, or ok=false when the path does not apply.
// Every wave of the grid must be resident at once (waves wait on each other).  The wave count per CU the design wants
// is 16 for Q <= 2 (4 workgroups of 4 waves hide each other's hand-off latency: +7 % at C2 over 2 per CU) and 8 above;
// whether the code object really admits that many is asked of the runtime (resident_blocks_per_cu) and the plan is
// refused otherwise, so that a compiler that allocates more registers ends in the streaming form, not in a spin.
// smallest slice (registers per array) whose wave count the chip can hold: geometry only (no tuning knob, no occupancy
// query), the same rule the plan applies first
static int resident_default_q(int n, int m, int d, int num_cus)
{
    if (d < 2 || d > 256 || (d & (d - 1)) != 0 || num_cus <= 0) return 0;
    const int64_t T = (int64_t)(n + m) * d;
    static const int kQ[5] = {1, 2, 4, 16, 32};
    for (int Q : kQ) {
        if ((64 * Q) % d != 0) continue;
        const int64_t nw = (T + 64 * (int64_t)Q - 1) / (64 * Q);
        if (nw <= (int64_t)num_cus * (Q <= 2 ? 16 : 8) && nw <= kResidentMaxWaves) return Q;
    }
    return 0;
}

ResidentEvents resident_events(int B, int n, int m, int d, int num_cus)
{
    ResidentEvents ev{0, 0, 0};
    const int Q = resident_default_q(n, m, d, num_cus);
    if (!Q || B > 64) return ev;
    ev.rows_per_wave = 64 * Q / d;
    ev.waves = (int)(((int64_t)(n + m) * d + 64 * (int64_t)Q - 1) / (64 * Q));
    // expected list entries per wave and step for uniformly drawn rows; a chunk of T steps (plus the boundary copies of
    // the deepest window) should average <= 20 of the 64 slots, so that an overflow is a property of the data (a row
    // most batches name), not of chance: P[Poisson(20) > 64] ~ 1e-14
    // (priced for slices of TWICE the smallest size: the plan may take the next slice size up — waves-per-CU knob,
    // occupancy of the actual code object — on the same workspace)
    const double h = 2.0 * 3.0 * B * ev.rows_per_wave / (double)(n + m);
    for (int ts = 8; ts >= 4; --ts)
        if (h * ((1 << ts) + kResidentEventLook) <= 20.0) {
            ev.tshift = ts;
            break;
        }
    return ev;
}

ResidentPlan plan_resident(int64_t N, int B, int n, int m, int d, int num_cus, bool bf16, int ev_tshift)
{
    ResidentPlan pl{};
    pl.ok = false;
    if (d < 2 || d > 256 || (d & (d - 1)) != 0 || num_cus <= 0 || !launcher_for(d)) return pl;
    const int64_t T = (int64_t)(n + m) * d;
    const int forced_q = g_tune.resident_q;
    const int wpc = g_tune.resident_wpc > 0 ? g_tune.resident_wpc : 16;
    int look = resident_lookahead(N, B, n, m);
    const bool fast = g_resident_math != 0;
    if (ev_tshift < 0) ev_tshift = resident_events(B, n, m, d, num_cus).tshift;
    if (ev_tshift == 0) look = 0;          // no event lists for this shape: publish right before use
    static const int kQ[5] = {1, 2, 4, 16, 32};
    for (int qi = 0; qi < 5; ++qi) {
        const int Q = kQ[qi];
        if ((64 * Q) % d != 0 || (forced_q && Q != forced_q)) continue;
        const int64_t nw = (T + 64 * (int64_t)Q - 1) / (64 * Q);
        const int want_wpc = Q <= 2 ? wpc : 8;
        if (nw <= (int64_t)num_cus * want_wpc && nw <= kResidentMaxWaves) {
            const int occ = resident_blocks_per_cu(d, Q, look, fast, bf16, g_tune.lds_pad);
            const int blocks = (int)((nw + 3) / 4);
            if (occ > 0 && (int64_t)blocks > (int64_t)occ * num_cus) continue;   // would not be resident: next Q or none
            pl.ok = true;
            pl.Q = Q;
            pl.NW = (int)nw;
            pl.blocks = blocks;
            pl.lookahead = look;
            pl.fast_math = fast;
            pl.bf16 = bf16;
            pl.tshift = ev_tshift;
            return pl;
        }
    }
    return pl;
}

// ---- prologue of a resident / local call: ONE kernel ----
//  (a) copies the call's host-built table (ResidentCold + per-step Adam scalars) into the workspace: out of the
//      kernel-argument segment for calls of up to 223 steps, else from the pinned staging slot, which the kernel reads
//      directly over the host link (no separate H2D copy on the stream either way);
//  (b) resident form only: translates the call's samples to VIRTUAL row ids (xs) and, for the look-ahead form, appends
//      every sample to the event list of each wave that owns one of its rows (resident_kernel.h: list (wave, chunk of
//      2^tshift steps); the first `look` steps of a chunk are also copied into the previous chunk's list).  The list
//      counters are all-zero on entry (workspace init + every wave clears its own counters at the end of a launch);
//      an entry that finds its list full is dropped and the count says so (the wave then takes the generic loop).
// INLINE: the table travels in the kernel-argument segment itself (first parameter, so it sits at offset 0 of the
// segment, which every lane can address); short calls then need no read over the host link at all.
#ifndef MFCD_PROLOGUE_THREADS
#define MFCD_PROLOGUE_THREADS 64
#endif
constexpr int kInlineStageUnits = 232;   // 16-byte units: ResidentCold (8) + 224 step scalars; 3712 bytes of kernarg
struct InlineStage {
    uint4 v[kInlineStageUnits];
};

size_t train_inline_stage_bytes() { return (size_t)kInlineStageUnits * 16; }

template <bool INLINE>
__global__ __launch_bounds__(256) void train_prologue_kernel(InlineStage inl, const uint4 *__restrict__ stage_host,
                                                             uint4 *__restrict__ stage_dev, int stage_units,
                                                             const mfcd_sample *__restrict__ samples, int64_t N, int B,
                                                             int n, int m, int rows_per_wave, int tshift, int look,
                                                             int64_t nch_cap, mfcd_sample *__restrict__ xs,
                                                             unsigned *__restrict__ ev_cnt, uint4 *__restrict__ ev_ent)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < stage_units) {
        if constexpr (INLINE) {
            const uint4 *ka = (const uint4 *)__builtin_amdgcn_kernarg_segment_ptr();
            stage_dev[t] = ka[t];
        } else {
            stage_dev[t] = stage_host[t];
        }
    }
    if (!xs || t >= N) return;
    const RowMap rm = make_row_map(n, m);
    mfcd_sample s = samples[t];
    s.u = rm.vrow_u(s.u);
    s.i = rm.vrow_v(s.i);
    s.j = rm.vrow_v(s.j);
    xs[t] = s;
    if (look <= 0) return;
    const int k = (int)(t / B);
    const unsigned tl = (unsigned)(t - (int64_t)k * B);
    const int w[3] = {s.u / rows_per_wave, s.i / rows_per_wave, s.j / rows_per_wave};
    const unsigned lr[3] = {(unsigned)(s.u - w[0] * rows_per_wave), (unsigned)(s.i - w[1] * rows_per_wave),
                            (unsigned)(s.j - w[2] * rows_per_wave)};
    const int c = k >> tshift;
    const bool copy_back = c > 0 && (k & ((1 << tshift) - 1)) < look;
    // one entry per distinct owner wave of the sample; all list slots are requested first (independent returning
    // atomics in flight together), the entries are stored once the slots are known
    uint4 e[3];
    size_t list[3];
    bool need[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        bool first = true;
#pragma unroll
        for (int r2 = 0; r2 < r; ++r2) first = first && w[r2] != w[r];
        need[r] = first;
        unsigned own = 0u, rows = 0u;
#pragma unroll
        for (int r2 = r; r2 < 3; ++r2)
            if (w[r2] == w[r]) {
                own |= 1u << r2;
                rows |= lr[r2] << (10 * r2);
            }
        e[r] = make_uint4(((unsigned)k << 9) | (tl << 3) | own, rows, __float_as_uint(s.z), 0u);
        list[r] = (size_t)w[r] * (size_t)nch_cap + (size_t)c;
    }
    unsigned slot[3] = {0u, 0u, 0u}, slot_back[3] = {0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (need[r]) slot[r] = atomicAdd(ev_cnt + list[r], 1u);
        if (need[r] && copy_back) slot_back[r] = atomicAdd(ev_cnt + list[r] - 1, 1u);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        // an entry that finds its list full is dropped: the count says so and the wave takes the generic loop
        if (need[r] && slot[r] < (unsigned)kEventCap - 1u) ev_ent[list[r] * kEventCap + slot[r]] = e[r];
        if (need[r] && copy_back && slot_back[r] < (unsigned)kEventCap - 1u)
            ev_ent[(list[r] - 1) * kEventCap + slot_back[r]] = e[r];
    }
}

int launch_train_prologue(const void *stage_host, const void *stage_host_devview, void *stage_dev, size_t stage_bytes,
                          const mfcd_sample *samples, int64_t N, int B, int n, int m, int rows_per_wave, int tshift,
                          int look, int64_t nch_cap, mfcd_sample *xs, unsigned *ev_cnt, void *ev_ent, hipStream_t st)
{
    const int units = (int)((stage_bytes + 15) / 16);
    const int64_t items = xs ? (N > units ? N : units) : units;
    // 64-thread workgroups: the kernel is a chain of dependent round trips per thread (record load, returning atomics,
    // entry stores), so it wants many workgroups per CU in flight, not few wide ones
    constexpr int kPT = MFCD_PROLOGUE_THREADS;
    const dim3 grid((unsigned)((items + kPT - 1) / kPT));
    if (units <= kInlineStageUnits) {
        InlineStage inl;
        std::memcpy(inl.v, stage_host, (size_t)units * 16);
        hipLaunchKernelGGL(train_prologue_kernel<true>, grid, dim3(kPT), 0, st, inl, (const uint4 *)nullptr,
                           (uint4 *)stage_dev, units, samples, N, B, n, m, rows_per_wave, tshift, look, nch_cap, xs,
                           ev_cnt, (uint4 *)ev_ent);
    } else {
        static const InlineStage none{};
        hipLaunchKernelGGL(train_prologue_kernel<false>, grid, dim3(kPT), 0, st, none, (const uint4 *)stage_host_devview,
                           (uint4 *)stage_dev, units, samples, N, B, n, m, rows_per_wave, tshift, look, nch_cap, xs,
                           ev_cnt, (uint4 *)ev_ent);
    }
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_resident_steps(const ResidentPlan &pl, const void *cold_dev, const mfcd_sample *xs, int64_t N, int B, int n,
                          int m, int d, const StepScalars *sc_dev, const AdamStatic &ac, unsigned long long *mailbox,
                          unsigned tag_base, void *loss_terms, unsigned long long *dbg, int K, hipStream_t st)
{
    ResidentArgs a;
    a.dbg = dbg;
    a.cold = (const ResidentCold *)cold_dev;
    a.samples = xs; a.sc = sc_dev; a.mailbox = mailbox; a.loss_terms = loss_terms;
    a.N = N; a.B = B; a.n = n; a.m = m; a.K = K; a.NW = pl.NW; a.ac = ac;
    a.lookahead = pl.lookahead;
    a.tag_base = tag_base;
    a.fast_math = pl.fast_math;
    a.bf16 = pl.bf16;
    a.lds_pad = g_tune.lds_pad;
    ResidentLauncher fn = launcher_for(d);
    if (!fn) return MFCD_EINVAL;
    return fn(&a, pl.Q, pl.blocks, (void *)st);
}

}  // namespace mfcd_detail
