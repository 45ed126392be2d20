// resident_kernel.h — the persistent register-resident step kernel (template); see resident.hip for the design.
#pragma once
#include "common.h"
#include "train_common.h"

namespace mfcd_detail {

typedef unsigned long long u64;

// Every lambda of the kernel must be inlined: a call would materialise its closure — references to the slice registers,
// the list registers, the kernel arguments — in scratch memory and turn the step loop into a walk over private memory
// (seen once: 0.64 -> 2.8 us per step when one lambda acquired a second call site).
#define MFCD_LAMBDA_INLINE __attribute__((always_inline))

// Diagnostic build (-DMFCD_RES_STATS, tools/diag_resident_stats.py only): per-wave cycle / event accounting of the
// look-ahead loop, written to dbg[8 + wave*8 ..]:
//   0 whole step loop  1 event steps (everything between "k == next_evt" and the dense update)  2 granule polls that
//   had to wait (from the first failed poll to success)  3 publish passes  4 hits | hits whose first poll succeeded << 40   5 failed polls | XCC_ID << 32  6 kernel start -> first step | HW_ID << 32
//   7 rows published | event steps << 32
#ifdef MFCD_RES_STATS
#define RS_NOW() ((u64)__builtin_amdgcn_s_memtime())
#define RS_ADD(slot, val) rs_acc[slot] += (u64)(val)
#else
#define RS_NOW() ((u64)0)
#define RS_ADD(slot, val) ((void)0)
#endif

// The per-step Adam scalars are written by the prologue kernel of the call and only read here: reading them through
// the constant address space makes every such load a SCALAR load (s_load through the scalar cache, values in SGPRs,
// lgkmcnt), whatever the kernel stores elsewhere.  As plain global loads they were vector loads whose vmcnt(0) wait at
// the end of every step also drained the granule requests issued one step ahead.
typedef const __attribute__((address_space(4))) float *ScalarTablePtr;
__device__ __forceinline__ StepScalars load_step_scalars(const StepScalars *table, int k)
{
    const ScalarTablePtr f = (ScalarTablePtr)(table + k);
    StepScalars s;
    s.neg_step_size = f[0];
    s.bc2_sqrt = f[1];
    s.inv_bc2_sqrt = f[2];
    s.pad = 0.0f;
    return s;
}


__device__ __forceinline__ u64 pack_granule(unsigned tag, float v)
{
    return ((u64)tag << 32) | (u64)__float_as_uint(v);
}

__device__ __forceinline__ u64 load_granule(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_granule(u64 *p, unsigned tag, float v)
{
    __hip_atomic_store(p, pack_granule(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Row order of the virtual table the waves cut into slices.  Item rows are touched twice as often as user rows (every
// sample names one user and two items), so with the plain concatenation [U; V] the waves that own item rows carry twice
// the hits of the others and set the pace (measured: the same state size split 2731 users / 5461 items, where all rows
// are touched equally often, ran 12 % faster than 4096 / 4096).  The virtual order therefore INTERLEAVES the two tables
// over their common length c = min(n, m): user r -> 2r, item r -> 2r + 1 (r < c); the rest of the longer table follows.
// The order is closed-form in both directions, so no table is kept: the prologue kernel translates the call's samples
// to virtual row ids with vrow_u / vrow_v, and the slice load / store of the step kernel maps back with is_item /
// table_row.  (A per-launch order rebuilt from the launch's own touch counts was measured in round 1 and dropped: the
// kernel ran within 1 % of this static order.)
struct RowMap {
    int n, m, c;   // c = min(n, m)
    __host__ __device__ int vrow_u(int u) const { return u < c ? 2 * u : 2 * c + (u - c); }
    __host__ __device__ int vrow_v(int i) const { return i < c ? 2 * i + 1 : 2 * c + (i - c); }
    // number of user / item rows whose virtual index is < x  (0 <= x <= n + m)
    __host__ __device__ int users_below(int x) const
    {
        if (x <= 2 * c) return (x + 1) >> 1;
        return n > c ? c + (x - 2 * c) : c;
    }
    __host__ __device__ int items_below(int x) const
    {
        if (x <= 2 * c) return x >> 1;
        return m > c ? c + (x - 2 * c) : c;
    }
    // virtual row -> table (true: V) and row id inside it
    __host__ __device__ bool is_item(int vr) const { return vr < 2 * c ? (vr & 1) != 0 : m > n; }
    __host__ __device__ int table_row(int vr) const { return vr < 2 * c ? vr >> 1 : c + (vr - 2 * c); }
};
__host__ __device__ inline RowMap make_row_map(int n, int m)
{
    RowMap r;
    r.n = n; r.m = m; r.c = n < m ? n : m;
    return r;
}

// ---- per-wave event lists (look-ahead form) ----
// The prologue kernel of a call hands every owner wave the list of the samples that name one of its rows, cut into
// CHUNKS of T = 2^tshift optimiser steps: list (wave w, chunk c) holds one 16-byte entry per such sample of steps
// [cT, (c+1)T), plus a copy of the entries of the first LOOK steps of chunk c + 1 (the look-ahead window reaches across
// the boundary), at most kEventCap of them, in arrival order of the prologue's atomics.  The wave loads a chunk's list
// into ONE entry per lane, sorts it by (step, slot in the batch) and from then on decides everything — is there a hit at
// this step, which row is due for publishing and into which mailbox slot — with v_readlane on its registers and two
// scalar cursors: no record load sits in front of a hit or a publish any more (round 2: two
// dependent memory round trips per hit), and the per-step path of a wave without events is one scalar compare.
//   word 0: step k << 9 | slot in the batch << 3 | own-role mask (bit r: the sample's role r = u, i, j is a row of mine)
//   word 1: my local row index of role 0 | role 1 << 10 | role 2 << 20   (valid where the own bit is set)
//   word 2: the sample's label z (fp32 bits);  word 3: unused
// A list that does not fit (a row that most batches name: popularity-sampled heads) makes its wave take the generic
// publish-right-before-use loop for the whole launch; the granule protocol is the same, so the two kinds of wave mix.
constexpr int kEventCap = 64;          // slots per list; a list that is USED holds at most kEventCap - 1 entries
constexpr unsigned kEventNone = 0xFFFFFFF8u;   // sorts last, step 0x7FFFFF, no own role
constexpr int kEventMaxLocalRows = 1024;

// Pointers needed only at a few points of a launch.  They live in device memory (workspace) and are (re)read with
// scalar loads there, so they do not occupy SGPRs during the step loop (with them passed by value the kernel needed
// > 102 SGPRs and spilled scalars into VGPR lanes on every step).
struct ResidentCold {
    float *U, *V, *mU, *vU, *mV, *vV;
    int *status;                     // 0 = ok, 1 = a bounded spin expired (sticky: never cleared by a launch)
    unsigned long long spin_limit;   // polls before a wave gives up
    unsigned *ev_cnt;                // [waves][nch_cap] entries appended to list (wave, chunk); all-zero between launches
                                     // (every wave clears its own counters at the end of a launch)
    uint4 *ev_ent;                   // [waves][nch_cap][kEventCap] entries
    long long nch_cap;               // chunks per wave the two arrays are laid out for
    long long tshift;                // log2(steps per chunk)
    float *loss_out;                 // [K] batch-mean BCE per step, formed inside the launch (look-ahead form); may be null
    unsigned long long pad[3];       // 128 bytes
};
static_assert(sizeof(ResidentCold) == 128, "train.hip fills this block as sixteen 8-byte words (kColdBytes)");

struct ResidentArgs {
    const ResidentCold *cold;
    const mfcd_sample *samples;   // the call's samples with u, i, j already translated to VIRTUAL row ids
    const StepScalars *sc;   // [K + 1]
    u64 *mailbox;            // [N][3][D] granules; a granule is valid when its tag == tag_base + step + 1
    void *loss_terms;        // look-ahead form: u64 [N] granules {tag, sigmoid output}; generic form: float [N] sigmoid outputs
    u64 *dbg;                // [8] who gave up first (diagnostics): {wave, step, sample position, own mask}
    unsigned tag_base;       // launch id << 21 (train.hip): granules left behind by earlier launches never match
    int64_t N;
    int B, n, m, K, NW;
    int lookahead;           // 0: publish right before use (any B); >0: look-ahead form (B <= 64)
    int fast_math;           // Adam arithmetic flavour: 0 IEEE-rounded, 1 v_sqrt / Newton-corrected rcp
    int bf16;                // factor tables are bf16 in HBM (cold->U / V then point to 2-byte elements)
    int lds_pad;             // unused dynamic LDS per workgroup: caps workgroups per CU so placement is even
    AdamStatic ac;
};

// This lane's record of chunk `base` of the batch that starts at sample pos0 (Bk samples); inert when past the end.
__device__ __forceinline__ mfcd_sample load_record(const mfcd_sample *__restrict__ samples, int64_t pos0, int Bk,
                                                   int base, int lane)
{
    mfcd_sample s;
    s.u = s.i = s.j = -0x40000000;
    s.z = 0.0f;
    if (base + lane < Bk) s = samples[pos0 + base + lane];
    return s;
}

struct Masks {
    u64 mu, mi, mj;  // lanes (samples) of a 64-record chunk whose u / i / j row I own
};

// D: factor width (power of two <= 256).  Q: registers per array per lane (slice = 64*Q elements).
// LOOK = 0: any batch size, chunked scan, rows published right after the step that precedes their use.
// LOOK = W > 0 (B <= 64, the reference's fixed batch size): look-ahead publishing.  The value a consumer needs
//   at step k is S_{k-1}(R); if batches j+1..k-1 do not touch R it is a pure function of the owner's state S_j
//   (k-1-j dense-only Adam updates), so the owner publishes it at the end of step j = max(k-W, last touch of R
//   before k) from a rolled-forward register copy.  The memory-side hand-off then overlaps W-1 steps of work
//   instead of sitting on every step's critical chain.  Same arithmetic in the same order -> identical bits.
// FAST: Adam arithmetic flavour (train_common.h): false = IEEE-rounded div/sqrt, true = v_sqrt / Newton-corrected rcp.
// BF16: the factor tables are bf16 in HBM (BASELINE configs[2]); the slice lives in registers as fp32 values that are
//   bf16-representable: every Adam update is followed by ONE round-to-nearest-even, the rounding point the streaming
//   bf16 form and the oracle define (mfcd_train_steps_bf16), so published rows are exactly what a step would re-read.
// Slices of Q >= 16 registers per array (C3: Q = 32, 2048 waves at two per SIMD) keep their row-gradient accumulators in
//   LDS instead of Q more registers: they are touched only in a hit step (the common step runs Adam with g = 0 and
//   never reads them), and a hit adds to the S registers of its row by address instead of a Q-long predicated chain.
template <int D, int Q, int LOOK, bool FAST, bool BF16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(Q >= 16 ? 2 : 1)))
void resident_train_kernel(ResidentArgs a)
{
    constexpr int S = D >= 64 ? D / 64 : 1;    // registers per row (gathered layout: lane <-> column lane + 64*s)
    constexpr int RPR = D < 64 ? 64 / D : 1;   // rows per register when D < 64
    constexpr int EW = 64 * Q;
    static_assert(EW % D == 0, "a wave's slice must hold whole rows");
    static_assert(EW / D <= kEventMaxLocalRows, "local row indices are 10-bit fields of an event entry");
    constexpr bool GRL = Q >= 16;             // row-gradient accumulators in LDS
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];   // [4 waves][64*Q] when GRL (+ the lds_pad knob)
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (gw >= a.NW) return;  // whole wave

    const int64_t T = (int64_t)(a.n + a.m) * D;
    const int64_t ebase = (int64_t)gw * EW;
    const int Rlo = (int)(ebase / D);
    const int64_t eend = (ebase + EW < T) ? ebase + EW : T;
    const int Rhi = (int)(eend / D);  // my rows are VIRTUAL row ids [Rlo, Rhi)
    const int lcol = lane & (D - 1);  // column of my lane when D < 64
    // element e of the virtual table -> table and address offset inside U or V (kernel start and end only)
    auto elem_is_item = [&](int64_t e) MFCD_LAMBDA_INLINE { return make_row_map(a.n, a.m).is_item((int)(e / D)); };
    auto elem_offset = [&](int64_t e) MFCD_LAMBDA_INLINE { return (int64_t)make_row_map(a.n, a.m).table_row((int)(e / D)) * D + (e % D); };

#ifdef MFCD_RES_STATS
    u64 rs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const u64 rs_t0 = RS_NOW();
    // -DMFCD_RES_STAMPS (tools/diag_short_call_stamps.py): absolute times (constant 100 MHz counter) of wave start, loop
    // start, loop end, wave end instead of slots 1, 2, 3, 7, and a second bank (dbg[8 + 4096*8 + wave*8 ..], s_memtime
    // ticks summed over the wave's hit steps): 0 event-step start -> first granule load  1 load -> every tag right
    // 2 -> end of the hit arithmetic  3 -> dense update done  4 -> touched rows published  5 hit steps
    // 6 the same as 1 for hits whose first look succeeded  7 their number
    [[maybe_unused]] const u64 rs_abs_t0 = __builtin_amdgcn_s_memrealtime();
    [[maybe_unused]] u64 rs_abs_l0 = 0, rs_abs_l1 = 0;
    [[maybe_unused]] u64 rs_hit[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] u64 rs_mark = 0;
#endif
    // ---- load my slice of the state into registers ----
    float p[Q], m1[Q], m2[Q], gr[GRL ? 1 : Q];
    float *const lgr = lds_dyn + (threadIdx.x >> 6) * (GRL ? EW : 0);   // this wave's accumulators: element q*64 + lane
    auto post = [](float x) MFCD_LAMBDA_INLINE {   // the rounding point of bf16 factor storage: once per update
        if constexpr (BF16) return (float)(mfcd_bf16)x;
        else return x;
    };
    {
        const ResidentCold c = *a.cold;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t e = ebase + q * 64 + lane;
            p[q] = m1[q] = m2[q] = 0.0f;
            if constexpr (GRL) lgr[q * 64 + lane] = 0.0f;
            if (e < T) {
                const int64_t o = elem_offset(e);
                if (!elem_is_item(e)) { p[q] = BF16 ? (float)((const mfcd_bf16 *)c.U)[o] : c.U[o]; m1[q] = c.mU[o]; m2[q] = c.vU[o]; }
                else { p[q] = BF16 ? (float)((const mfcd_bf16 *)c.V)[o] : c.V[o]; m1[q] = c.mV[o]; m2[q] = c.vV[o]; }
            }
        }
    }

    auto batch_size = [&](int step) MFCD_LAMBDA_INLINE {
        const int64_t pos0 = (int64_t)step * a.B;
        return (int)((a.N - pos0) < a.B ? (a.N - pos0) : a.B);
    };
    auto scan = [&](const mfcd_sample &s, int Bk, int base) MFCD_LAMBDA_INLINE {
        const bool valid = base + lane < Bk;
        const int ru = s.u, ri = s.i, rj = s.j;   // virtual ids
        Masks M;
        M.mu = __ballot(valid && ru >= Rlo && ru < Rhi);
        M.mi = __ballot(valid && ri >= Rlo && ri < Rhi);
        M.mj = __ballot(valid && rj >= Rlo && rj < Rhi);
        return M;
    };
    // first register of my LOCAL row lr (virtual row Rlo + lr): wave-uniform
    auto reg_of = [&](int lr) MFCD_LAMBDA_INLINE { return D >= 64 ? lr * S : lr / RPR; };

    // a bounded wait expired (or somebody else's did): make it known and remember who gave up first
    auto give_up = [&](int step, int64_t pos, int what) MFCD_LAMBDA_INLINE {
        int *const status = a.cold->status;
        if (lane == 0) {
            if (__hip_atomic_exchange(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && a.dbg) {
                a.dbg[0] = (u64)gw; a.dbg[1] = (u64)step; a.dbg[2] = (u64)pos; a.dbg[3] = (u64)what;
            }
        }
    };
    // one failed poll: returns true when the wave must give up (limit reached, or the abort word is set)
    auto poll_failed = [&](unsigned &spins, unsigned &limit) MFCD_LAMBDA_INLINE -> bool {
        if (spins == 0) limit = (unsigned)a.cold->spin_limit;   // first failed poll: the limit lives behind the cold pointer
        ++spins;
        if (spins > limit || (spins & 255u) == 0) {   // rare
            if (spins > limit ||
                __hip_atomic_load(a.cold->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                return true;
        }
        return false;
    };

    // write my local row lr as tagged granules into mailbox slot (pos*3 + role)
    auto store_row = [&](int lr, int64_t slot, unsigned tag) MFCD_LAMBDA_INLINE {
        u64 *dst = a.mailbox + slot * D;
        const int q0 = reg_of(lr);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (q >= q0 && q < q0 + S) {
                if constexpr (D >= 64) {
                    store_granule(dst + (q - q0) * 64 + lane, tag, p[q]);
                } else {
                    if (lane / D == lr % RPR) store_granule(dst + lcol, tag, p[q]);
                }
            }
        }
    };

    // One sample that names rows of mine (own0..2: which roles; lr0..2: my local rows of those roles): fetch the other
    // rows, form g, accumulate the row gradients of my rows.  `pos` is the sample's position in the call, `tag` its
    // step's tag.  Returns false when a bounded wait expired.
    // (Requesting the granules one step ahead — an extra event step per hit that only issues the loads — was built
    // and measured in round 3: same step time with and without it at every window depth: the rows a waiting hit waits
    // for do not exist yet — the launch is bound by the sample stream's dependency chain, one producer -> consumer
    // hand-off per link (tools/exp_chain_depth.py), not by the round trip of a request — so the simpler form stays.)
    auto process_hit = [&](bool own0, bool own1, bool own2, int lr0, int lr1, int lr2, float zz, int64_t pos,
                           unsigned tag, float inv_batch, int step) MFCD_LAMBDA_INLINE -> bool {
        const bool own[3] = {own0, own1, own2};
        const int lrs[3] = {lr0, lr1, lr2};
        const u64 *slot = a.mailbox + pos * 3 * D;

        // the three rows in gathered layout (zero beyond column D)
        float row[3][S];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) row[r][s2] = 0.0f;
            if (!own[r]) continue;
            // a row of mine comes straight from my registers
            const int q0 = reg_of(lrs[r]);
            if constexpr (D >= 64) {
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    if (q >= q0 && q < q0 + S) row[r][(q * 64 % D) / 64] = p[q];
            } else {
                float sel = 0.0f;
#pragma unroll
                for (int q = 0; q < Q; ++q) sel = (q == q0) ? p[q] : sel;
                const float v = __shfl(sel, (lrs[r] % RPR) * D + lcol, MFCD_WAVE);
                row[r][0] = lane < D ? v : 0.0f;
            }
        }
        // rows owned by other waves: their granules, polled until every tag is this step's
        unsigned spins = 0, limit = 0;
        [[maybe_unused]] u64 rs_w0 = 0;
        RS_ADD(4, 1ull);
#ifdef MFCD_RES_STAMPS
        { const u64 t = RS_NOW(); rs_hit[0] += t - rs_mark; rs_mark = t; }
#endif
#if defined(MFCD_RES_STAGGER) && MFCD_RES_STAGGER > 0
        // EXPERIMENT (make exp EXPFLAGS=-DMFCD_RES_STAGGER=<sleep units>): after a failed first look two sets of poll
        // loads stay in flight half a round trip apart, so that the memory is sampled twice per round trip
        {
            u64 ga[3][S], gb[3][S];
            auto issue = [&](u64 (&g)[3][S]) MFCD_LAMBDA_INLINE {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (own[r]) continue;
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2) {
                        const int c = lane + 64 * s2;
                        if (c < D) g[r][s2] = load_granule(slot + (int64_t)r * D + c);
                    }
                }
            };
            auto check = [&](const u64 (&g)[3][S]) MFCD_LAMBDA_INLINE -> bool {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (own[r]) continue;
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2) {
                        const int c = lane + 64 * s2;
                        if (c < D) {
                            ok = ok && ((unsigned)(g[r][s2] >> 32) == tag);
                            row[r][s2] = __uint_as_float((unsigned)g[r][s2]);
                        }
                    }
                }
                return __all(ok);
            };
            issue(ga);
            if (!check(ga)) {
#ifdef MFCD_RES_STATS
                rs_w0 = RS_NOW();
#endif
                issue(ga);
                __builtin_amdgcn_s_sleep(MFCD_RES_STAGGER);
                issue(gb);
                while (true) {
                    RS_ADD(5, 1);
                    if (poll_failed(spins, limit)) {
                        give_up(step, pos, (own0 ? 1 : 0) | (own1 ? 2 : 0) | (own2 ? 4 : 0));
                        return false;
                    }
                    if (check(ga)) break;
                    issue(ga);
                    if (check(gb)) break;
                    issue(gb);
                }
            }
        }
#else
        while (true) {
            bool ok = true;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (own[r]) continue;
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) {
                    const int c = lane + 64 * s2;
                    if (c < D) {
                        const u64 gq = load_granule(slot + (int64_t)r * D + c);
                        ok = ok && ((unsigned)(gq >> 32) == tag);
                        row[r][s2] = __uint_as_float((unsigned)gq);
                    }
                }
            }
            if (__all(ok)) break;
#ifdef MFCD_RES_STATS
            if (spins == 0) rs_w0 = RS_NOW();
            RS_ADD(5, 1);
#endif
            if (poll_failed(spins, limit)) {
                give_up(step, pos, (own0 ? 1 : 0) | (own1 ? 2 : 0) | (own2 ? 4 : 0));
                return false;
            }
        }
#endif
#ifdef MFCD_RES_STATS
        if (spins) RS_ADD(2, RS_NOW() - rs_w0);
        else RS_ADD(4, 1ull << 40);
#endif
#ifdef MFCD_RES_STAMPS
        { const u64 t = RS_NOW(); rs_hit[1] += t - rs_mark; if (!spins) { rs_hit[6] += t - rs_mark; rs_hit[7] += 1; } rs_mark = t; }
#endif

        float acc = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) acc += row[0][s2] * (row[1][s2] - row[2][s2]);
        float pr, g;
        if constexpr (FAST) {
            // fast flavour: hardware exponential and reciprocals (<= 1 ulp each) with one Newton correction of each
            // quotient instead of expf and two IEEE divisions (~60 fewer vector instructions per hit, all of them on the
            // producer -> consumer chain the launch is bound by).  Same operation order as the reference sequence (common.h).
            const float x = wave_sum64(acc);
            const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896340736f);      // exp(-x)
            const float dn = 1.0f + e;
            pr = div_newton(1.0f, dn, __builtin_amdgcn_rcpf(dn));
            pr = (e == __builtin_inff()) ? 0.0f : pr;                                  // exp overflow: 1/inf = 0 (NaN from the correction step)
            const float den = fmaxf((1.0f - pr) * pr, 1e-12f);
            const float num = inv_batch * (pr - zz);
            g = div_newton(num, den, __builtin_amdgcn_rcpf(den)) * (1.0f - pr) * pr;
        } else {
            pr = sigmoid_f32(wave_sum64(acc));
            g = bce_sigmoid_backward_f32(pr, zz, inv_batch);
        }
        // u's owner records the sigmoid output; the BCE term is formed off this path (in-launch batch means of the
        // look-ahead form, batch_mean_kernel otherwise)
        if (own[0] && lane == 0) {
            if constexpr (LOOK > 0) store_granule((u64 *)a.loss_terms + pos, tag, pr);
            else ((float *)a.loss_terms)[pos] = pr;
        }

        float du[S], dv[S];  // g*(V[i]-V[j]) and g*U[u] in gathered layout
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
            du[s2] = g * (row[1][s2] - row[2][s2]);
            dv[s2] = g * row[0][s2];
        }
        if constexpr (D < 64) {  // bring column (lane % D) to every lane
            du[0] = __shfl(du[0], lcol, MFCD_WAVE);
            dv[0] = __shfl(dv[0], lcol, MFCD_WAVE);
        }
        if constexpr (GRL) {
            // by address: the S accumulators of each owned row (role order u, i, j = the register form's order per element)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (!own[r]) continue;
                const int q0 = reg_of(lrs[r]);
                if constexpr (D >= 64) {
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2)
                        lgr[(q0 + s2) * 64 + lane] += r == 0 ? du[s2] : (r == 1 ? dv[s2] : -dv[s2]);
                } else {
                    if (lane / D == lrs[r] % RPR) lgr[q0 * 64 + lane] += r == 0 ? du[0] : (r == 1 ? dv[0] : -dv[0]);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int rowq = (q * 64 + lane) / D;          // local row of this lane's element of register q
                const int sq = D >= 64 ? (q * 64 % D) / 64 : 0;
                if (own[0] && rowq == lrs[0]) gr[q] += du[sq];
                if (own[1] && rowq == lrs[1]) gr[q] += dv[sq];
                if (own[2] && rowq == lrs[2]) gr[q] += -dv[sq];
            }
        }
#ifdef MFCD_RES_STAMPS
        { const u64 t = RS_NOW(); rs_hit[2] += t - rs_mark; rs_mark = t; }
#endif
        return true;
    };

    // dense Adam over the slice for one step; `hit`: this wave accumulated row gradients in this step (wave-uniform)
    auto step_update = [&](bool hit, const StepScalars &sc) MFCD_LAMBDA_INLINE {
        if constexpr (GRL) {
            if (hit) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float g = lgr[q * 64 + lane];
                    lgr[q * 64 + lane] = 0.0f;
                    adam_update_t<FAST>(p[q], m1[q], m2[q], g, a.ac, sc);
                    p[q] = post(p[q]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    adam_update_t<FAST>(p[q], m1[q], m2[q], 0.0f, a.ac, sc);
                    p[q] = post(p[q]);
                }
            }
        } else {
            // scalar, not packed pairs: v_pk_* measured 2.3x slower here (0.43 -> 0.99 us per common-path step at C2)
            adam_update_q<FAST, Q, false>(p, m1, m2, gr, a.ac, sc);
            if constexpr (BF16) {
#pragma unroll
                for (int q = 0; q < Q; ++q) p[q] = post(p[q]);
            }
        }
    };

    // ================= generic loop (any B): chunked scan, rows published right before their use =================
    // The whole launch of the LOOK = 0 instantiations, and the whole launch of a single WAVE of the look-ahead form whose
    // event list did not fit.  Returns false when a bounded wait expired.
    auto generic_loop = [&]() MFCD_LAMBDA_INLINE -> bool {
        auto publish = [&](int step) MFCD_LAMBDA_INLINE {
            const int64_t pos0 = (int64_t)step * a.B;
            const int Bk = batch_size(step);
            for (int base = 0; base < Bk; base += MFCD_WAVE) {
                const mfcd_sample s = load_record(a.samples, pos0, Bk, base, lane);
                const Masks M = scan(s, Bk, base);
                u64 mask = M.mu | M.mi | M.mj;
                while (mask) {
                    const int tl = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const int64_t slot0 = (pos0 + base + tl) * 3;
                    if ((M.mu >> tl) & 1ull) store_row(__shfl(s.u, tl, MFCD_WAVE) - Rlo, slot0 + 0, a.tag_base + (unsigned)step + 1u);
                    if ((M.mi >> tl) & 1ull) store_row(__shfl(s.i, tl, MFCD_WAVE) - Rlo, slot0 + 1, a.tag_base + (unsigned)step + 1u);
                    if ((M.mj >> tl) & 1ull) store_row(__shfl(s.j, tl, MFCD_WAVE) - Rlo, slot0 + 2, a.tag_base + (unsigned)step + 1u);
                }
            }
        };
        publish(0);
        for (int k = 0; k < a.K; ++k) {
            const int64_t pos0 = (int64_t)k * a.B;
            const int Bk = batch_size(k);
            const float inv_batch = 1.0f / (float)Bk;
            const StepScalars sc = load_step_scalars(a.sc, k);
            if constexpr (!GRL) {
#pragma unroll
                for (int q = 0; q < Q; ++q) gr[q] = 0.0f;
            }
            bool any_hit = false;
            for (int base = 0; base < Bk; base += MFCD_WAVE) {
                const mfcd_sample s = load_record(a.samples, pos0, Bk, base, lane);
                const Masks M = scan(s, Bk, base);
                u64 mask = M.mu | M.mi | M.mj;
                any_hit = any_hit || mask != 0ull;
                while (mask) {
                    const int tl = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const int hu = __shfl(s.u, tl, MFCD_WAVE) - Rlo, hi = __shfl(s.i, tl, MFCD_WAVE) - Rlo,
                              hj = __shfl(s.j, tl, MFCD_WAVE) - Rlo;
                    const float hz = __shfl(s.z, tl, MFCD_WAVE);
                    const bool ok = process_hit((bool)((M.mu >> tl) & 1ull), (bool)((M.mi >> tl) & 1ull),
                                                (bool)((M.mj >> tl) & 1ull), hu, hi, hj, hz, pos0 + base + tl,
                                                a.tag_base + (unsigned)k + 1u, inv_batch, k);
                    if (__builtin_amdgcn_readfirstlane((int)!ok)) return false;
                }
            }
            step_update(any_hit, sc);
            if (k + 1 < a.K) publish(k + 1);
        }
        return true;
    };

    bool alive = true;   // false: a bounded wait expired; the launch is winding down (parameters undefined, status set)

    if constexpr (LOOK > 0) {
        // ================= B <= 64: look-ahead publishing driven by the wave's event list =================
        // depth of the look-ahead window: a launch argument (<= the boundary copies the lists were built with)
        const int W = __builtin_amdgcn_readfirstlane(a.lookahead);
        const int N32 = (int)a.N;                                  // host guarantees N + 64*(W+2) < 2^31
        const int gws = __builtin_amdgcn_readfirstlane(gw);
        const int tshift = (int)a.cold->tshift;
        const int NCH = ((a.K - 1) >> tshift) + 1;                 // chunks of this call

        // ---- does every list of mine fit ?  (wave-uniform) ----
        bool fits = true;
        {
            const unsigned *cnt = a.cold->ev_cnt + (size_t)gws * (size_t)a.cold->nch_cap;
            for (int c = lane; c < NCH; c += MFCD_WAVE) fits = fits && cnt[c] < (unsigned)kEventCap;   // lane 63 stays a sentinel
            fits = __all(fits);
        }
        if (!__builtin_amdgcn_readfirstlane((int)fits)) {
            alive = generic_loop();
        } else {
            // this lane's entry of the current chunk's list, sorted by word 0 (kEventNone past the end; lane 63 always)
            unsigned ekey = kEventNone, erow = 0u;
            int estep = 0x7FFFFF;                                  // ekey >> 9
            float ez = 0.0f;
            int chunk_end = 0;                                     // first step that is not in the current chunk

            auto load_chunk = [&](int c) MFCD_LAMBDA_INLINE {
                const ResidentCold *cz = a.cold;
                const size_t li = (size_t)gws * (size_t)cz->nch_cap + (size_t)c;
                const uint4 e = cz->ev_ent[li * kEventCap + lane];   // past the count: stale bytes, masked below
                const unsigned n_raw = cz->ev_cnt[li];
                const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)n_raw);
                const unsigned key = (unsigned)lane < n ? e.x : kEventNone;
                // rank of my key among the n entries (keys are unique: one entry per (step, slot)); lanes past the
                // end keep their place, so `dest` is a permutation of the 64 lanes
                unsigned rank = 0u;
                for (unsigned l = 0; l < n; ++l) {
                    const unsigned kl = (unsigned)__builtin_amdgcn_readlane((int)key, (int)l);
                    rank += kl < key ? 1u : 0u;
                }
                const int dest = (int)((unsigned)lane < n ? rank : (unsigned)lane) << 2;
                ekey = (unsigned)__builtin_amdgcn_ds_permute(dest, (int)key);
                erow = (unsigned)__builtin_amdgcn_ds_permute(dest, (int)e.y);
                ez = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dest, (int)e.z));
                estep = (int)(ekey >> 9);
                const int ce = (c + 1) << tshift;
                chunk_end = ce < a.K ? ce : a.K;
            };
            // The sorted list is walked with two cursors (scalars): hc = the first entry whose step is >= the current
            // step (the next hit), pc = the first entry that has not entered the look-ahead window yet.  A list holds at
            // most 63 entries, so lane 63 is always a sentinel and a cursor never leaves the wave.
            auto step_at = [&](int i) MFCD_LAMBDA_INLINE -> int { return __builtin_amdgcn_readlane(estep, i); };
            auto lanes_in = [](int lo, int hi) MFCD_LAMBDA_INLINE -> u64 { return ((1ull << hi) - 1ull) & ~((1ull << lo) - 1ull); };   // 0 <= lo, hi <= 63
            // lanes whose entry names my local row lr (in a role I own)
            auto rowmask = [&](int lr) MFCD_LAMBDA_INLINE -> u64 {
                const bool t = ((ekey & 1u) && (int)(erow & 1023u) == lr) ||
                               ((ekey & 2u) && (int)((erow >> 10) & 1023u) == lr) ||
                               ((ekey & 4u) && (int)((erow >> 20) & 1023u) == lr);
                return __ballot(t);
            };

            // Publishing rule.  The registers hold the state after step jbase (-1: the initial state).  Entry e (step ke,
            // role r, my row lr) is due now iff no entry with a step in (jbase, ke) names lr, and
            //   after the update of a hit step jbase:  ke <= jbase + W and batch jbase named lr (its value for ke could
            //          not be known earlier); at the start (jbase = -1) every row counts as just touched;
            //   before the hits and the update of step jbase + 1:  ke == jbase + 1 + W — the entry has just entered the
            //          window (a row that batch jbase + 1 names lies in (jbase, ke) and waits for its update).
            // Every (entry, role) is published exactly once: by the step of its row's previous touch if that lies
            // within W steps, else W steps ahead.  The row is rolled forward over steps jbase+1 .. ke-1 (dense-only
            // updates, same arithmetic in the same order as the slice will see: identical bits).
            // publish_entry: entry idx; entries [conf_lo, conf_hi) are the ones with a step in (jbase, ke), entries
            // [touch_lo, touch_hi) those of step jbase (need_touch).
            auto publish_entry = [&](int idx, int jbase, int conf_lo, int conf_hi, int touch_lo, int touch_hi,
                                     bool need_touch) MFCD_LAMBDA_INLINE {
                const unsigned key_l = (unsigned)__builtin_amdgcn_readlane((int)ekey, idx);
                const unsigned row_l = (unsigned)__builtin_amdgcn_readlane((int)erow, idx);
                const int ke = (int)(key_l >> 9);
                const int tl = (int)((key_l >> 3) & 63u);
#pragma unroll 1
                for (int r = 0; r < 3; ++r) {
                    if (!((key_l >> r) & 1u)) continue;
                    const int lr = (int)((row_l >> (10 * r)) & 1023u);
                    if (conf_lo < conf_hi || need_touch) {
                        const u64 rm = rowmask(lr);
                        if (rm & lanes_in(conf_lo, conf_hi)) continue;
                        if (need_touch && !(rm & lanes_in(touch_lo, touch_hi))) continue;
                    }
                    // roll the registers of my row lr forward over steps jbase+1 .. ke-1; the scalars of step b+1 are
                    // requested (scalar cache) before step b is computed
                    const int q0 = reg_of(lr);
                    float pp[S], mm1[S], mm2[S];
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2) pp[s2] = mm1[s2] = mm2[s2] = 0.0f;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if (q >= q0 && q < q0 + S) {
                            const int s2 = D >= 64 ? (q * 64 % D) / 64 : 0;
                            pp[s2] = p[q];
                            mm1[s2] = m1[q];
                            mm2[s2] = m2[q];
                        }
                    }
                    StepScalars scb = load_step_scalars(a.sc, jbase + 1);          // (the table holds K+1 entries; ke <= K-1)
#pragma unroll 1
                    for (int st = jbase + 1; st < ke; ++st) {
                        const StepScalars scn = load_step_scalars(a.sc, st + 1);
#pragma unroll
                        for (int s2 = 0; s2 < S; ++s2) {
                            adam_update_t<FAST>(pp[s2], mm1[s2], mm2[s2], 0.0f, a.ac, scb);
                            pp[s2] = post(pp[s2]);
                        }
                        scb = scn;
                    }
                    u64 *dst = a.mailbox + (((int64_t)ke * a.B + tl) * 3 + r) * D;
                    const unsigned tag = a.tag_base + (unsigned)ke + 1u;
                    if constexpr (D >= 64) {
#pragma unroll
                        for (int s2 = 0; s2 < S; ++s2) store_granule(dst + s2 * 64 + lane, tag, pp[s2]);
                    } else {
                        if (lane / D == lr % RPR) store_granule(dst + lcol, tag, pp[0]);
                    }
                    RS_ADD(7, 1);
                }
            };
            // entries [lo, hi) (steps in (jbase, jbase + W]) from the state after step jbase; the rows must have been
            // named by the entries [touch_lo, touch_hi) of step jbase (jbase >= 0)
            auto publish_fresh = [&](int jbase, int lo, int hi, int touch_lo, int touch_hi) MFCD_LAMBDA_INLINE {
                if (lo >= hi) return;
                [[maybe_unused]] const u64 rs_p0 = RS_NOW();
                int g0 = lo, gstep = step_at(lo);                  // first entry of the current entry's step
                for (int idx = lo; idx < hi; ++idx) {
                    const int st = step_at(idx);
                    if (st != gstep) { g0 = idx; gstep = st; }
                    publish_entry(idx, jbase, lo, g0, touch_lo, touch_hi, jbase >= 0);
                }
                RS_ADD(3, RS_NOW() - rs_p0);
            };
            load_chunk(0);
            int hc = 0, pc = __builtin_popcountll(__ballot(estep < W));
            publish_fresh(-1, 0, pc, 0, 0);                        // first uses within the first W steps, from the initial state
            int nh = step_at(0), np = step_at(pc);
            // the next step at which this wave has anything to do besides its dense update: a hit, an entry entering
            // the window, the end of the chunk
            auto next_event = [&]() MFCD_LAMBDA_INLINE -> int {
                const int cp = np - W;
                const int e = nh < cp ? nh : cp;
                return e < chunk_end ? e : chunk_end;
            };
            int next_evt = next_event();

            // The step loop is cut at this wave's events: between two of them the wave runs QUIET steps — scalars of the
            // next step requested, dense update with a literal zero gradient, counter — in a loop of its own with no
            // flag, no gradient register and no event test inside (every instruction of this loop is paid by all
            // waves at every step: it sets the floor under the launch, and a wave on the dependency chain shares its
            // SIMD's issue slots with three waves running it).
            ScalarTablePtr scp = (ScalarTablePtr)a.sc;             // the table holds K+1 entries (host pads one)
            auto load_scalars_at = [&](ScalarTablePtr f) MFCD_LAMBDA_INLINE {
                StepScalars x;
                x.neg_step_size = f[0];
                x.bc2_sqrt = f[1];
                x.inv_bc2_sqrt = f[2];
                x.pad = 0.0f;
                return x;
            };
            StepScalars sc_cur = load_scalars_at(scp);
            int chunk = 0;
#ifdef MFCD_RES_STATS
            const u64 rs_l0 = RS_NOW();
            rs_acc[6] = rs_l0 - rs_t0;
            rs_abs_l0 = __builtin_amdgcn_s_memrealtime();
#endif
            int k = 0;
            while (true) {
                const int kq = next_evt < a.K ? next_evt : a.K;
                while (k < kq) {                                   // quiet steps
                    scp += 4;
                    const StepScalars sc_next = load_scalars_at(scp);
                    if constexpr (GRL) {
                        step_update(false, sc_cur);
                    } else {
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            adam_update_t<FAST>(p[q], m1[q], m2[q], 0.0f, a.ac, sc_cur);
                            p[q] = post(p[q]);
                        }
                    }
                    sc_cur = sc_next;
                    ++k;
                }
                if (k >= a.K) break;
                // ---- event step k ----
                scp += 4;
                const StepScalars sc_next = load_scalars_at(scp);
                if constexpr (!GRL) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) gr[q] = 0.0f;
                }
                bool hit = false;
                int h0 = 0;
                {
                    // a wave with an event is on somebody's critical chain, the waves on the common path have slack:
                    // it issues ahead of them until its step is done
                    __builtin_amdgcn_s_setprio(3);
                    [[maybe_unused]] const u64 rs_e0 = RS_NOW();
#ifdef MFCD_RES_STAMPS
                    rs_mark = rs_e0;
#endif
                    RS_ADD(7, 1ull << 32);
                    if (k == chunk_end) {
                        load_chunk(++chunk);
                        hc = 0;
                        nh = step_at(0);
                        pc = __builtin_popcountll(__ballot(estep < k + W));   // (earlier steps entered the window in the last chunk)
                        np = step_at(pc);
                    }
                    if (np == k + W) {                             // entries that enter the window now
                        [[maybe_unused]] const u64 rs_p0 = RS_NOW();
                        const int g0 = pc;
                        do {
                            publish_entry(pc, k - 1, hc, g0, 0, 0, false);
                            ++pc;
                            np = step_at(pc);
                        } while (np == k + W);
                        RS_ADD(3, RS_NOW() - rs_p0);
                    }
                    h0 = hc;
                    if (nh == k) {                                 // my hits of this step, in batch order
                        hit = true;
                        const int bk = (N32 - k * a.B) < a.B ? (N32 - k * a.B) : a.B;
                        const float inv_batch = 1.0f / (float)bk;
                        const unsigned tag = a.tag_base + (unsigned)k + 1u;
                        do {
                            const unsigned key_l = (unsigned)__builtin_amdgcn_readlane((int)ekey, hc);
                            const unsigned row_l = (unsigned)__builtin_amdgcn_readlane((int)erow, hc);
                            const float z_l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ez), hc));
                            // the abort decision is wave-uniform; saying so keeps the step loop free of exec-mask bookkeeping
                            const bool ok = process_hit((bool)(key_l & 1u), (bool)(key_l & 2u), (bool)(key_l & 4u),
                                                        (int)(row_l & 1023u), (int)((row_l >> 10) & 1023u),
                                                        (int)((row_l >> 20) & 1023u), z_l,
                                                        (int64_t)k * a.B + (int)((key_l >> 3) & 63u), tag, inv_batch, k);
                            if (__builtin_amdgcn_readfirstlane((int)!ok)) { alive = false; break; }
                            ++hc;
                            nh = step_at(hc);
                        } while (nh == k);
                        if (!alive) break;
                    }
                    next_evt = next_event();
                    if (!hit) __builtin_amdgcn_s_setprio(0);
                    RS_ADD(1, RS_NOW() - rs_e0);
                }
                step_update(hit, sc_cur);
#ifdef MFCD_RES_STAMPS
                if (hit) { const u64 t = RS_NOW(); rs_hit[3] += t - rs_mark; rs_mark = t; }
#endif
                if (__builtin_expect(hit, 0)) {
                    publish_fresh(k, hc, pc, h0, hc);              // rows batch k touched: their next use, if within W steps
                    __builtin_amdgcn_s_setprio(0);
#ifdef MFCD_RES_STAMPS
                    { const u64 t = RS_NOW(); rs_hit[4] += t - rs_mark; rs_hit[5] += 1; }
#endif
                }
                sc_cur = sc_next;
                ++k;
            }
            __builtin_amdgcn_s_setprio(0);
#ifdef MFCD_RES_STATS
            rs_abs_l1 = __builtin_amdgcn_s_memrealtime();
            rs_acc[0] = RS_NOW() - rs_l0;
            rs_acc[6] |= (u64)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32;    // HW_ID: wave / SIMD / CU / SH / SE
            rs_acc[5] |= (u64)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32;   // XCC_ID
            if (lane == 0 && a.dbg)
                for (int x = 0; x < 8; ++x) a.dbg[8 + (int64_t)gw * 8 + x] = rs_acc[x];
#endif
        }

        // ---- batch means, inside the launch: wave w forms the mean BCE of steps w, w + NW, ... from the tagged sigmoid
        // outputs their u-owners recorded (same terms, same summation order as batch_mean_kernel).  Nobody waits for
        // this wave, so the polls are off every critical chain; they end when the slowest owner has passed the step.
        {
            float *const loss_out = a.cold->loss_out;
            if (loss_out != nullptr) {
                for (int k = gws; k < a.K && alive; k += a.NW) {
                    const int64_t off = (int64_t)k * a.B;
                    const int b = (int)((a.N - off) < a.B ? (a.N - off) : a.B);
                    const unsigned tag = a.tag_base + (unsigned)k + 1u;
                    const float zz = lane < b ? a.samples[off + lane].z : 0.0f;
                    unsigned spins = 0, limit = 0;
                    float pr = 0.0f;
                    while (true) {
                        bool ok = true;
                        if (lane < b) {
                            const u64 gq = load_granule((const u64 *)a.loss_terms + off + lane);
                            ok = (unsigned)(gq >> 32) == tag;
                            pr = __uint_as_float((unsigned)gq);
                        }
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(8);
                        if (poll_failed(spins, limit)) {
                            give_up(k, off, 8);
                            alive = false;
                            break;
                        }
                    }
                    if (!alive) break;
                    float acc = lane < b ? bce_term_f32(pr, zz) : 0.0f;
                    acc = wave_sum64(acc);
                    if (lane == 0) loss_out[k] = acc / (float)b;
                }
            }
        }
        // my list counters are read by nobody else: leave them all-zero for the next launch's prologue (atomicAdd)
        {
            unsigned *const cnt = a.cold->ev_cnt + (size_t)gws * (size_t)a.cold->nch_cap;
            for (int c = lane; c < NCH; c += MFCD_WAVE) cnt[c] = 0u;
        }
    } else {
        alive = generic_loop();
    }

    // ---- write my slice back (the table pointers are re-read: they were not kept live across the loop) ----
    // (after an expired wait the values are undefined but the status word says so; the store keeps the exit path single)
    const ResidentCold c = *a.cold;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t e = ebase + q * 64 + lane;
        if (e < T) {
            const int64_t o = elem_offset(e);
            if (!elem_is_item(e)) {
                if constexpr (BF16) ((mfcd_bf16 *)c.U)[o] = (mfcd_bf16)p[q];   // exact: p is bf16-representable
                else c.U[o] = p[q];
                c.mU[o] = m1[q]; c.vU[o] = m2[q];
            } else {
                if constexpr (BF16) ((mfcd_bf16 *)c.V)[o] = (mfcd_bf16)p[q];
                else c.V[o] = p[q];
                c.mV[o] = m1[q]; c.vV[o] = m2[q];
            }
        }
    }
#ifdef MFCD_RES_STAMPS
    if (lane == 0 && a.dbg) {
        u64 *const o = a.dbg + 8 + (int64_t)gw * 8;
        o[1] = rs_abs_t0; o[2] = rs_abs_l0; o[3] = rs_abs_l1; o[7] = __builtin_amdgcn_s_memrealtime();
        for (int x = 0; x < 8; ++x) o[4096 * 8 + x] = rs_hit[x];
    }
#endif
}

}  // namespace mfcd_detail
