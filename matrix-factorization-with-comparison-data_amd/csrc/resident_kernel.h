// resident_kernel.h — the persistent register-resident step kernel (template); see resident.hip for the design.
#pragma once
#include "common.h"
#include "train_common.h"

namespace mfcd_detail {

typedef unsigned long long u64;


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

// Diagnostic build (-DMFCD_STAMPS, tools/ only): per-wave cycle accounting written to a debug region.
// dbg[gw*8 + {0 total, 1 poll-wait, 2 hit-compute, 3 adam, 4 publish, 5 hits, 6 polls, 7 hit-steps}]
#ifdef MFCD_STAMPS
#define STAMP() ((u64)__builtin_amdgcn_s_memtime())
#define DBG_ADD(slot, val) dbg_acc[slot] += (val)
// MFCD_STAMPS=2 ("light"): no stamp on the common path — slot 3 = whole hit block of a step (record load .. last hit),
// slot 4 = publish slow path only, slot 7 = number of publish slow paths — so that the step loop keeps its timing
#else
#define STAMP() ((u64)0)
#define DBG_ADD(slot, val) ((void)0)
#endif

// Pointers needed only when the slice is loaded at kernel start and stored at kernel end.  They live in device
// memory (workspace) and are (re)read with scalar loads at those two points, so they do not occupy SGPRs during
// the step loop (with them passed by value the kernel needed > 102 SGPRs and spilled scalars into VGPR lanes on
// every step).
struct ResidentCold {
    float *U, *V, *mU, *vU, *mV, *vV;
    int *status;                     // 0 = ok, 1 = a bounded spin expired (sticky: never cleared by a launch)
    unsigned long long spin_limit;   // polls before a wave gives up
    unsigned *touch;                 // [strings][KW]: bit s of a string = "batch s touches a row of that string" (look-
                                     // ahead form); all-zero between launches: every wave clears its own strings
    long long KW;                    // dwords per string (covers K + 64 steps; bits past K are zero)
    unsigned long long pad[6];       // 128 bytes
};
static_assert(sizeof(ResidentCold) == 128, "train.hip fills this block as sixteen 8-byte words (kColdBytes)");

struct ResidentArgs {
    const ResidentCold *cold;
    const mfcd_sample *samples;   // the call's samples with u, i, j already translated to VIRTUAL row ids
    const StepScalars *sc;   // [K]
    u64 *mailbox;            // [N][3][D] granules; a granule is valid when its tag == tag_base + step + 1
    float *loss_terms;       // [N]
    u64 *dbg;                // [NW][8] cycle accounting (diagnostic build only)
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
    auto elem_is_item = [&](int64_t e) { return make_row_map(a.n, a.m).is_item((int)(e / D)); };
    auto elem_offset = [&](int64_t e) { return (int64_t)make_row_map(a.n, a.m).table_row((int)(e / D)) * D + (e % D); };

    // ---- load my slice of the state into registers ----
    float p[Q], m1[Q], m2[Q], gr[GRL ? 1 : Q];
    float *const lgr = lds_dyn + (threadIdx.x >> 6) * (GRL ? EW : 0);   // this wave's accumulators: element q*64 + lane
    auto post = [](float x) {   // the rounding point of bf16 factor storage: once per update
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
#ifdef MFCD_TRACE
    u64 dbg_t_arrive = 0;
    int dbg_nhit = 0;
#endif
#ifdef MFCD_STAMPS
    [[maybe_unused]] u64 dbg_rt_age = 0, dbg_rt_n = 0;
    u64 dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const u64 t_start = STAMP();
#endif

    auto batch_size = [&](int step) {
        const int64_t pos0 = (int64_t)step * a.B;
        return (int)((a.N - pos0) < a.B ? (a.N - pos0) : a.B);
    };
    auto scan = [&](const mfcd_sample &s, int Bk, int base) {
        const bool valid = base + lane < Bk;
        const int ru = s.u, ri = s.i, rj = s.j;   // virtual ids
        Masks M;
        M.mu = __ballot(valid && ru >= Rlo && ru < Rhi);
        M.mi = __ballot(valid && ri >= Rlo && ri < Rhi);
        M.mj = __ballot(valid && rj >= Rlo && rj < Rhi);
        return M;
    };
    // first register of global row R (one of mine): wave-uniform
    auto reg_of = [&](int R) { return D >= 64 ? (R - Rlo) * S : (R - Rlo) / RPR; };

    // write row R (mine) as tagged granules into mailbox slot (pos*3 + role)
    auto store_row = [&](int R, int64_t slot, unsigned tag) {
        u64 *dst = a.mailbox + slot * D;
        const int q0 = reg_of(R);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (q >= q0 && q < q0 + S) {
                if constexpr (D >= 64) {
                    store_granule(dst + (q - q0) * 64 + lane, tag, p[q]);
                } else {
                    if (lane / D == (R - Rlo) % RPR) store_granule(dst + lcol, tag, p[q]);
                }
            }
        }
    };

    // One sample (lane tl of chunk `base` of batch k) that touches my rows: fetch the other rows, form g,
    // accumulate the row gradients of my rows into gr[].  Returns false when a bounded wait expired.
    // `hs` is the sample itself (wave-uniform), `M`/`tl` say which of its rows are mine.
    auto process_hit = [&](const mfcd_sample &hs, const Masks &M, int tl, int64_t pos, unsigned tag,
                           float inv_batch) -> bool {
        const int rows[3] = {hs.u, hs.i, hs.j};
        const bool own[3] = {(bool)((M.mu >> tl) & 1ull), (bool)((M.mi >> tl) & 1ull), (bool)((M.mj >> tl) & 1ull)};
        const float zz = hs.z;
        const u64 *slot = a.mailbox + pos * 3 * D;

        // the three rows in gathered layout (zero beyond column D)
        float row[3][S];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) row[r][s2] = 0.0f;
            if (!own[r]) continue;
            // a row of mine comes straight from my registers
            const int q0 = reg_of(rows[r]);
            if constexpr (D >= 64) {
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    if (q >= q0 && q < q0 + S) row[r][(q * 64 % D) / 64] = p[q];
            } else {
                float sel = 0.0f;
#pragma unroll
                for (int q = 0; q < Q; ++q) sel = (q == q0) ? p[q] : sel;
                const float v = __shfl(sel, ((rows[r] - Rlo) % RPR) * D + lcol, MFCD_WAVE);
                row[r][0] = lane < D ? v : 0.0f;
            }
        }
        // rows owned by other waves: poll their granules until every tag is this step's
        unsigned spins = 0, limit = 0;
#ifdef MFCD_TRACE
        dbg_t_arrive = (u64)__builtin_amdgcn_s_memrealtime();
#endif
        [[maybe_unused]] const u64 t_poll0 = STAMP();
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
            if (spins == 0) limit = (unsigned)a.cold->spin_limit;   // first failed poll: the limit lives behind the cold pointer
            ++spins;
            if (spins > limit || (spins & 255u) == 0) {   // rare
                int *const status = a.cold->status;
                if (spins > limit || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
        }
        [[maybe_unused]] const u64 t_poll1 = STAMP();
        DBG_ADD(1, t_poll1 - t_poll0);
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
        if (spins > 0) {   // waited: how old is the publish, and was it issued after I started polling ?
            const u64 now_rt = (u64)__builtin_amdgcn_s_memrealtime();
            u64 newest = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (!own[r]) {
                    const u64 tp = __hip_atomic_load(a.mailbox + (int64_t)a.N * 3 * D + pos * 3 + r, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                    newest = tp > newest ? tp : newest;
                }
            dbg_rt_age += now_rt - newest;                      // success time - (latest) publish time, 10 ns units
            dbg_rt_n += 1;
        }
#ifdef MFCD_TRACE
        // event trace (tools/trace_resident.py): one 32-byte record per hit behind the publish-time array:
        // {sample position | own flags << 56, arrival (real-time clock, 10 ns), poll success, spins}
        if (lane == 0 && dbg_nhit < 128) {
            u64 *rec = a.mailbox + (int64_t)a.N * 3 * D + (int64_t)a.N * 3 + ((int64_t)gw * 128 + dbg_nhit) * 4;
            rec[0] = (u64)pos | ((u64)(own[0] | (own[1] << 1) | (own[2] << 2)) << 56);
            rec[1] = dbg_t_arrive;
            rec[2] = (u64)__builtin_amdgcn_s_memrealtime();
            rec[3] = spins;
        }
        dbg_nhit += 1;
#endif
#endif
        DBG_ADD(5, 1);
        DBG_ADD(6, spins);

        float acc = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) acc += row[0][s2] * (row[1][s2] - row[2][s2]);
        const float pr = sigmoid_f32(wave_sum64(acc));
        const float g = bce_sigmoid_backward_f32(pr, zz, inv_batch);
        // u's owner records the sigmoid output; the BCE term is formed by the finalize kernel (off this path)
        if (own[0] && lane == 0) a.loss_terms[pos] = pr;

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
                const int q0 = reg_of(rows[r]);
                if constexpr (D >= 64) {
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2)
                        lgr[(q0 + s2) * 64 + lane] += r == 0 ? du[s2] : (r == 1 ? dv[s2] : -dv[s2]);
                } else {
                    if (lane / D == (rows[r] - Rlo) % RPR) lgr[q0 * 64 + lane] += r == 0 ? du[0] : (r == 1 ? dv[0] : -dv[0]);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int rowq = Rlo + (q * 64 + lane) / D;
                const int sq = D >= 64 ? (q * 64 % D) / 64 : 0;
                if (own[0] && rowq == rows[0]) gr[q] += du[sq];
                if (own[1] && rowq == rows[1]) gr[q] += dv[sq];
                if (own[2] && rowq == rows[2]) gr[q] += -dv[sq];
            }
        }
        DBG_ADD(2, STAMP() - t_poll1);
        return true;
    };

    // dense Adam over the slice for one step; `hit`: this wave accumulated row gradients in this step (wave-uniform)
    auto step_update = [&](bool hit, const StepScalars &sc) {
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

    if constexpr (LOOK > 0) {
        // ================= B <= 64: look-ahead publishing over a window of LOOK batches =================
        // Hot path per step and wave (no hit, no publish candidate — about nine wave-steps in ten): one prefetched
        // 16-byte record load, one ballot, a handful of scalar moves, Adam on the slice.  Only ONE 64-bit mask per
        // window batch lives in registers ("some sample of that batch touches a row of mine"); role masks and the
        // records themselves are re-derived from a reload only when such a sample exists.
        constexpr int W = LOOK;
        const int N32 = (int)a.N;                                  // host guarantees N + 64*(W+2) < 2^31
        const unsigned rcnt = (unsigned)(Rhi - Rlo);   // my rows: virtual ids [Rlo, Rlo + rcnt)
        // this lane's record of batch `step`: branch-free (clamped address, then inert ids for lanes past the batch
        // or past the stream), 32-bit byte offset from a scalar base
        const char *const sbase = (const char *)a.samples;
        auto load_rec = [&](int step) {
            const int pos = step * a.B + lane;
            const bool valid = lane < a.B && pos < N32;
            const unsigned off = (unsigned)(valid ? pos : 0) * 16u;
            mfcd_sample s = *(const mfcd_sample *)(sbase + off);
            s.u = valid ? s.u : -0x40000000;
            s.i = valid ? s.i : -0x40000000;
            s.j = valid ? s.j : -0x40000000;
            return s;
        };
        auto role_masks = [&](const mfcd_sample &s) {
            Masks M;
            M.mu = __ballot((unsigned)(s.u - Rlo) < rcnt);
            M.mi = __ballot((unsigned)(s.i - Rlo) < rcnt);
            M.mj = __ballot((unsigned)(s.j - Rlo) < rcnt);
            return M;
        };
        auto lane_sample = [&](const mfcd_sample &s, int tl) {     // record of lane tl as wave-uniform values
            mfcd_sample r;
            r.u = __builtin_amdgcn_readlane(s.u, tl);
            r.i = __builtin_amdgcn_readlane(s.i, tl);
            r.j = __builtin_amdgcn_readlane(s.j, tl);
            r.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.z), tl));
            return r;
        };
        auto touches = [&](const mfcd_sample &s, int R) {          // does the batch held in `s` touch virtual row R ?
            return __ballot(s.u == R || s.i == R || s.j == R) != 0ull;
        };

        // Which batches touch a row of mine is known before the launch (resident_translate_kernel: one pass over the
        // call's samples).  A sliding window over such a bit string — bit b <-> batch j + b, j = the step just finished —
        // replaces a per-step record load + range test + ballot per wave (every wave scanning every batch).
        // Waves of up to 4 rows keep one window PER ROW (a.touch then holds one string per virtual row): which row is
        // due for publishing, and for which step, is then decided with scalar bit tests alone, and the publish path
        // fetches only the records of the batches it actually publishes for.
        constexpr int RPW = EW / D;                 // rows per wave
        constexpr bool ROWWIN = RPW <= 4;
        constexpr int NWIN = ROWWIN ? RPW : 1;
        const int gws = __builtin_amdgcn_readfirstlane(gw);
        const unsigned *tw[NWIN];
        u64 winR[NWIN];
        const int KW0 = (int)a.cold->KW;
#pragma unroll
        for (int r = 0; r < NWIN; ++r) {
            tw[r] = a.cold->touch + (size_t)(ROWWIN ? gws * RPW + r : gws) * KW0;
            // (the strings are wave-uniform data, but they are written at the end of this kernel, so the compiler issues
            // a vector load for them; without the readfirstlane the windows live in VGPRs and every test on them below
            // becomes exec-mask control flow: that was 56 scalar instructions per wave-step)
            winR[r] = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)tw[r][0]) << 1;   // j = -1: "batch -1" is empty
        }
        u64 win = winR[0];                          // any of my rows
#pragma unroll
        for (int r = 1; r < NWIN; ++r) win |= winR[r];
        int fill = 33, widx = 1;     // bits 0 .. fill-1 of the windows are valid; the next dword of a string goes to bit `fill`

        // publish, from the state after step j, every (k, R) whose turn it is (see the template comment)
        // phase (per-row windows only): 0 everything, from the state after step j (the initial call);
        //   1 BEFORE step j's own hits and update: the rows batch j does not touch — their value after step j is a
        //     dense-only update away, so it is published now and does not wait behind this wave's hit of the step;
        //   2 after the update: the rows batch j touched.
        auto publish_phase = [&](int j, bool first, int phase, const StepScalars &sc_j) {
            u64 cand = win & (1ull << W);
            if (first || (win & 1ull)) cand |= win & ((1ull << W) - 2ull);   // bits 1 .. W-1
            if (phase == 1) cand = win & (1ull << W);
            if (phase == 2 && !(win & 1ull)) cand = 0ull;
            if (__builtin_expect(cand == 0ull, 1)) return;         // the common case (laid out as the fall-through)
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
            const u64 t_pub0 = STAMP();
#endif
            if constexpr (ROWWIN) {
                // per-row windows: (row r, step k = j + kk) is due iff batch k touches the row, no batch in (j, k) does,
                // and either k has just entered the window (kk == W) or the row was touched by batch j itself (its
                // value for k could not be known earlier) — the same rule as below, read off the bit strings
                StepScalars scw[W];
                bool have_sc = false;
#pragma unroll
                for (int r = 0; r < NWIN; ++r) {
                    const u64 wr = winR[r];
                    const bool fresh = first || (wr & 1ull);
                    if (phase == 1 && (wr & 1ull)) continue;               // touched by batch j: after the update
                    if (phase == 2 && !(wr & 1ull)) continue;              // done before the update
#pragma unroll
                    for (int kk = 1; kk <= W; ++kk) {
                        const int k = j + kk;
                        if (!((wr >> kk) & 1ull) || k >= a.K) continue;
                        if (wr & ((1ull << kk) - 2ull)) continue;          // touched again in (j, k): published later
                        if (kk < W && !fresh) continue;                    // published when k entered the window
                        const int R = Rlo + r;
                        const mfcd_sample rk = load_rec(k);
                        if (!have_sc) {                                    // roll-forward scalars, same burst
#pragma unroll
                            for (int b2 = 1; b2 < W; ++b2) scw[b2] = a.sc[(j + b2) < a.K ? (j + b2) : a.K];
                            have_sc = true;
                        }
                        // roll the registers of row R forward over steps j+1 .. k-1 (dense-only updates)
                        const int q0 = reg_of(R);
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
                        if (phase == 1) {                                  // step j itself: dense-only for this row
#pragma unroll
                            for (int s2 = 0; s2 < S; ++s2) {
                                adam_update_t<FAST>(pp[s2], mm1[s2], mm2[s2], 0.0f, a.ac, sc_j);
                                pp[s2] = post(pp[s2]);
                            }
                        }
#pragma unroll
                        for (int b2 = 1; b2 < kk; ++b2) {
#pragma unroll
                            for (int s2 = 0; s2 < S; ++s2) {
                                adam_update_t<FAST>(pp[s2], mm1[s2], mm2[s2], 0.0f, a.ac, scw[b2]);
                                pp[s2] = post(pp[s2]);
                            }
                        }
                        const u64 mr[3] = {(u64)__ballot(rk.u == R), (u64)__ballot(rk.i == R), (u64)__ballot(rk.j == R)};
                        const unsigned tag = a.tag_base + (unsigned)k + 1u;
#pragma unroll
                        for (int role = 0; role < 3; ++role) {
                            u64 pm = mr[role];
                            while (pm) {
                                const int tl = __ffsll((long long)pm) - 1;
                                pm &= pm - 1;
                                u64 *dst = a.mailbox + (((int64_t)k * a.B + tl) * 3 + role) * D;
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
                                if (lane == 0)   // publish time (100 MHz real-time clock), behind the mailbox
                                    __hip_atomic_store(a.mailbox + (int64_t)a.N * 3 * D + ((int64_t)k * a.B + tl) * 3 + role,
                                                       (u64)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
#endif
                                if constexpr (D >= 64) {
#pragma unroll
                                    for (int s2 = 0; s2 < S; ++s2) store_granule(dst + s2 * 64 + lane, tag, pp[s2]);
                                } else {
                                    if (lane / D == (R - Rlo) % RPR) store_granule(dst + lcol, tag, pp[0]);
                                }
                            }
                        }
                    }
                }
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
                DBG_ADD(4, STAMP() - t_pub0);
                DBG_ADD(7, 1);
#endif
                return;
            }
            // slow path: bring the whole window's records in with ONE burst of independent loads
            mfcd_sample wrec[W + 1];
#pragma unroll
            for (int b2 = 0; b2 <= W; ++b2) wrec[b2] = load_rec(j + b2 >= 0 ? j + b2 : a.K + 1);
            // the per-step scalars of the roll-forward travel in the same burst (inside the loops below each would be a
            // dependent load on the publish path); the table holds K+1 entries
            StepScalars scw[W];
#pragma unroll
            for (int b2 = 1; b2 < W; ++b2) scw[b2] = a.sc[(j + b2) < a.K ? (j + b2) : a.K];
#pragma unroll
            for (int kk = 1; kk <= W; ++kk) {
                const int k = j + kk;
                if (!((win >> kk) & 1ull) || k >= a.K) continue;
                if (kk < W && !first && !(win & 1ull)) continue;   // none of my rows was touched by batch j
                const mfcd_sample rk = wrec[kk];
                const Masks Mk = role_masks(rk);
                u64 pm = Mk.mu | Mk.mi | Mk.mj;
                while (pm) {
                    const int tl = __ffsll((long long)pm) - 1;
                    pm &= pm - 1;
                    const mfcd_sample sk = lane_sample(rk, tl);
                    const int rows[3] = {sk.u, sk.i, sk.j};
                    const bool fl[3] = {(bool)((Mk.mu >> tl) & 1ull), (bool)((Mk.mi >> tl) & 1ull),
                                        (bool)((Mk.mj >> tl) & 1ull)};
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        if (!fl[r]) continue;
                        const int R = rows[r];
                        bool later_touch = false;   // a batch in (j, k) touches R -> published after that step instead
#pragma unroll
                        for (int b2 = 1; b2 < kk; ++b2) later_touch = later_touch | touches(wrec[b2], R);
                        if (later_touch) continue;
                        if (kk < W && !first && !touches(wrec[0], R)) continue;  // already published when k entered the window
                        // roll the registers of row R forward over steps j+1 .. k-1 (dense-only updates)
                        const int q0 = reg_of(R);
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
#pragma unroll
                        for (int b2 = 1; b2 < kk; ++b2) {
#pragma unroll
                            for (int s2 = 0; s2 < S; ++s2) {
                                adam_update_t<FAST>(pp[s2], mm1[s2], mm2[s2], 0.0f, a.ac, scw[b2]);
                                pp[s2] = post(pp[s2]);
                            }
                        }
                        u64 *dst = a.mailbox + (((int64_t)k * a.B + tl) * 3 + r) * D;
                        const unsigned tag = a.tag_base + (unsigned)k + 1u;
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
                        if (lane == 0)   // publish time (100 MHz real-time clock), behind the mailbox
                            __hip_atomic_store(a.mailbox + (int64_t)a.N * 3 * D + ((int64_t)k * a.B + tl) * 3 + r,
                                               (u64)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
#endif
                        if constexpr (D >= 64) {
#pragma unroll
                            for (int s2 = 0; s2 < S; ++s2) store_granule(dst + s2 * 64 + lane, tag, pp[s2]);
                        } else {
                            if (lane / D == (R - Rlo) % RPR) store_granule(dst + lcol, tag, pp[0]);
                        }
                    }
                }
            }
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
            DBG_ADD(4, STAMP() - t_pub0);
            DBG_ADD(7, 1);
#endif
        };

        publish_phase(-1, true, 0, a.sc[0]);

        StepScalars sc_cur = a.sc[0];
        const StepScalars *sc_ptr = a.sc + 1;      // running pointer: the table holds K+1 entries (host pads one)
        for (int k = 0; k < a.K; ++k) {
            // slide the windows: bit 0 becomes batch k
            --fill;
#pragma unroll
            for (int r = 0; r < NWIN; ++r) winR[r] >>= 1;
            if (__builtin_expect(fill < 16, 0)) {
#pragma unroll
                for (int r = 0; r < NWIN; ++r)
                    winR[r] |= (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)tw[r][widx]) << fill;
                ++widx;
                fill += 32;
            }
            win = winR[0];
#pragma unroll
            for (int r = 1; r < NWIN; ++r) win |= winR[r];
#if defined(MFCD_RES_EXP) && MFCD_RES_EXP == 1   // timing experiment (tools/): no per-step scalar load
            const StepScalars sc_next = sc_cur;
#else
            const StepScalars sc_next = *sc_ptr++;
#endif

            if constexpr (!GRL) {
#pragma unroll
                for (int q = 0; q < Q; ++q) gr[q] = 0.0f;
            }
            if constexpr (ROWWIN) publish_phase(k, false, 1, sc_cur);
            // a wave with a hit or a fresh publish is on somebody's critical chain, the waves on the common path have
            // slack: it issues ahead of them until its step is done
            const bool urgent = (win & 1ull) != 0ull;
            if (__builtin_expect(urgent, 0)) __builtin_amdgcn_s_setprio(3);
            if (__builtin_expect((win & 1ull) != 0ull, 0)) {
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
                const u64 t_hit0 = STAMP();
#endif
                const mfcd_sample rk = load_rec(k);
                const Masks M0 = role_masks(rk);
                const int64_t pos0 = (int64_t)k * a.B;
                const int bk = (N32 - k * a.B) < a.B ? (N32 - k * a.B) : a.B;
                const float inv_batch = 1.0f / (float)bk;
                u64 mask = M0.mu | M0.mi | M0.mj;
                while (mask) {
                    const int tl = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    // the abort decision is wave-uniform; saying so keeps the step loop free of exec-mask bookkeeping
                    const bool ok = process_hit(lane_sample(rk, tl), M0, tl, pos0 + tl, a.tag_base + (unsigned)k + 1u, inv_batch);
                    if (__builtin_amdgcn_readfirstlane((int)!ok)) return;
                }
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
                DBG_ADD(3, STAMP() - t_hit0);
#endif
            }
#if defined(MFCD_STAMPS) && MFCD_STAMPS == 2
            step_update(urgent, sc_cur);
            publish_phase(k, false, ROWWIN ? 2 : 0, sc_cur);
            if (__builtin_expect(urgent, 0)) __builtin_amdgcn_s_setprio(0);
#else
            [[maybe_unused]] const u64 t_adam0 = STAMP();
            step_update(urgent, sc_cur);
            [[maybe_unused]] const u64 t_adam1 = STAMP();
            DBG_ADD(3, t_adam1 - t_adam0);
            publish_phase(k, false, ROWWIN ? 2 : 0, sc_cur);
            if (__builtin_expect(urgent, 0)) __builtin_amdgcn_s_setprio(0);
            DBG_ADD(4, STAMP() - t_adam1);
#endif
            sc_cur = sc_next;
        }
        // my touch strings are read by nobody else: leave them all-zero for the next launch's prologue (atomicOr).
        // Their address is re-derived from fresh scalar loads so that nothing stays live across the step loop for it.
        {
            const ResidentCold *cz = a.cold;
            asm volatile("" : "+s"(cz));
            unsigned *const tz = cz->touch;
            const int KWz = (int)cz->KW;
#pragma unroll
            for (int r = 0; r < NWIN; ++r)
                for (int w = lane; w < KWz; w += MFCD_WAVE) tz[(size_t)(ROWWIN ? gws * RPW + r : gws) * KWz + w] = 0u;
        }
    } else {
        // ================= any B: chunked scan, publish after the whole slice is updated =================
        auto publish = [&](int step) {
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
                    if ((M.mu >> tl) & 1ull) store_row(__shfl(s.u, tl, MFCD_WAVE), slot0 + 0, a.tag_base + (unsigned)step + 1u);
                    if ((M.mi >> tl) & 1ull) store_row(__shfl(s.i, tl, MFCD_WAVE), slot0 + 1, a.tag_base + (unsigned)step + 1u);
                    if ((M.mj >> tl) & 1ull) store_row(__shfl(s.j, tl, MFCD_WAVE), slot0 + 2, a.tag_base + (unsigned)step + 1u);
                }
            }
        };
        publish(0);
        for (int k = 0; k < a.K; ++k) {
            const int64_t pos0 = (int64_t)k * a.B;
            const int Bk = batch_size(k);
            const float inv_batch = 1.0f / (float)Bk;
            const StepScalars sc = a.sc[k];
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
                    mfcd_sample hs;
                    hs.u = __shfl(s.u, tl, MFCD_WAVE);
                    hs.i = __shfl(s.i, tl, MFCD_WAVE);
                    hs.j = __shfl(s.j, tl, MFCD_WAVE);
                    hs.z = __shfl(s.z, tl, MFCD_WAVE);
                    const bool ok = process_hit(hs, M, tl, pos0 + base + tl, a.tag_base + (unsigned)k + 1u, inv_batch);
                    if (__builtin_amdgcn_readfirstlane((int)!ok)) return;
                }
            }
            step_update(any_hit, sc);
            if (k + 1 < a.K) publish(k + 1);
        }
    }

#ifdef MFCD_STAMPS
    dbg_acc[0] = STAMP() - t_start;
#if MFCD_STAMPS == 2
    dbg_acc[2] = dbg_rt_age;   // light mode: slot 2 = sum of (success - publish) over waited polls [10 ns], slot 6 spins
    dbg_acc[5] = dbg_acc[5] | (dbg_rt_n << 32);   // high half: number of waited polls
#endif
    if (lane == 0 && a.dbg)
        for (int x = 0; x < 8; ++x) a.dbg[(int64_t)gw * 8 + x] = dbg_acc[x];
#endif
    // ---- write my slice back (the table pointers are re-read: they were not kept live across the loop) ----
    const ResidentCold *cp = a.cold;
    asm volatile("" : "+s"(cp));   // opaque to the optimiser: forces fresh scalar loads instead of 12 live SGPRs
    const ResidentCold c = *cp;
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
}

}  // namespace mfcd_detail
