// resident_kernel.h — the persistent register-resident step kernel (template); see resident.hip for the design.
#pragma once
#include "common.h"
#include "train_common.h"

namespace mfcd_detail {

typedef unsigned long long u64;

constexpr unsigned kSpinLimit = 1u << 22;  // polls before a wave gives up (~seconds); sets status = 1

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

// Diagnostic build (-DMFCD_STAMPS, tools/ only): per-wave cycle accounting written to a debug region.
// dbg[gw*8 + {0 total, 1 poll-wait, 2 hit-compute, 3 adam, 4 publish, 5 hits, 6 polls, 7 hit-steps}]
#ifdef MFCD_STAMPS
#define STAMP() ((u64)__builtin_amdgcn_s_memtime())
#define DBG_ADD(slot, val) dbg_acc[slot] += (val)
#else
#define STAMP() ((u64)0)
#define DBG_ADD(slot, val) ((void)0)
#endif

struct ResidentArgs {
    float *U, *V, *mU, *vU, *mV, *vV;
    const mfcd_sample *samples;
    const StepScalars *sc;   // [K]
    u64 *mailbox;            // [N][3][D] granules, zero-filled before the launch
    float *loss_terms;       // [N]
    int *status;             // 0 = ok, 1 = a bounded spin expired
    u64 *dbg;                // [NW][8] cycle accounting (diagnostic build only)
    int64_t N;
    int B, n, m, K, NW;
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
// SMALLB: every batch fits one 64-record chunk (B <= 64, the reference's fixed batch size): the masks of
// the next batch are computed off the critical path and the rows the next batch needs are updated and
// published before the rest of the slice.
template <int D, int Q, bool SMALLB>
__global__ __launch_bounds__(256) void resident_train_kernel(ResidentArgs a)
{
    constexpr int S = D >= 64 ? D / 64 : 1;    // registers per row (gathered layout: lane <-> column lane + 64*s)
    constexpr int RPR = D < 64 ? 64 / D : 1;   // rows per register when D < 64
    constexpr int EW = 64 * Q;
    static_assert(EW % D == 0, "a wave's slice must hold whole rows");
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (gw >= a.NW) return;  // whole wave

    const int64_t TU = (int64_t)a.n * D, T = (int64_t)(a.n + a.m) * D;
    const int64_t ebase = (int64_t)gw * EW;
    const int Rlo = (int)(ebase / D);
    const int64_t eend = (ebase + EW < T) ? ebase + EW : T;
    const int Rhi = (int)(eend / D);  // my rows are global row ids [Rlo, Rhi); V rows are offset by n
    const int lcol = lane & (D - 1);  // column of my lane when D < 64

    // ---- load my slice of the state into registers ----
    float p[Q], m1[Q], m2[Q], gr[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t e = ebase + q * 64 + lane;
        p[q] = m1[q] = m2[q] = 0.0f;
        if (e < T) {
            if (e < TU) { p[q] = a.U[e]; m1[q] = a.mU[e]; m2[q] = a.vU[e]; }
            else { p[q] = a.V[e - TU]; m1[q] = a.mV[e - TU]; m2[q] = a.vV[e - TU]; }
        }
    }
#ifdef MFCD_STAMPS
    u64 dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const u64 t_start = STAMP();
#endif

    auto batch_size = [&](int step) {
        const int64_t pos0 = (int64_t)step * a.B;
        return (int)((a.N - pos0) < a.B ? (a.N - pos0) : a.B);
    };
    auto scan = [&](const mfcd_sample &s, int Bk, int base) {
        const bool valid = base + lane < Bk;
        const int ru = s.u, ri = s.i + a.n, rj = s.j + a.n;
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
    auto process_hit = [&](const mfcd_sample &s, const Masks &M, int tl, int64_t pos, unsigned tag,
                           float inv_batch) -> bool {
        const int rows[3] = {__shfl(s.u, tl, MFCD_WAVE), __shfl(s.i, tl, MFCD_WAVE) + a.n,
                             __shfl(s.j, tl, MFCD_WAVE) + a.n};
        const bool own[3] = {(bool)((M.mu >> tl) & 1ull), (bool)((M.mi >> tl) & 1ull), (bool)((M.mj >> tl) & 1ull)};
        const float zz = __shfl(s.z, tl, MFCD_WAVE);
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
        unsigned spins = 0;
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
            ++spins;
            if (spins > kSpinLimit ||
                ((spins & 255u) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        [[maybe_unused]] const u64 t_poll1 = STAMP();
        DBG_ADD(1, t_poll1 - t_poll0);
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
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int rowq = Rlo + (q * 64 + lane) / D;
            const int sq = D >= 64 ? (q * 64 % D) / 64 : 0;
            if (own[0] && rowq == rows[0]) gr[q] += du[sq];
            if (own[1] && rowq == rows[1]) gr[q] += dv[sq];
            if (own[2] && rowq == rows[2]) gr[q] += -dv[sq];
        }
        DBG_ADD(2, STAMP() - t_poll1);
        return true;
    };

    if constexpr (SMALLB) {
        // ================= B <= 64: software-pipelined masks, critical rows first =================
        mfcd_sample rec_cur = load_record(a.samples, 0, batch_size(0), 0, lane);
        mfcd_sample rec_next = rec_cur;
        if (a.K > 1) rec_next = load_record(a.samples, (int64_t)a.B, batch_size(1), 0, lane);
        StepScalars sc = a.sc[0];
        Masks Mc = scan(rec_cur, batch_size(0), 0);
        {   // initial values of the rows batch 0 needs, tag 1
            u64 mask = Mc.mu | Mc.mi | Mc.mj;
            while (mask) {
                const int tl = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if ((Mc.mu >> tl) & 1ull) store_row(__shfl(rec_cur.u, tl, MFCD_WAVE), (int64_t)tl * 3 + 0, 1u);
                if ((Mc.mi >> tl) & 1ull) store_row(__shfl(rec_cur.i, tl, MFCD_WAVE) + a.n, (int64_t)tl * 3 + 1, 1u);
                if ((Mc.mj >> tl) & 1ull) store_row(__shfl(rec_cur.j, tl, MFCD_WAVE) + a.n, (int64_t)tl * 3 + 2, 1u);
            }
        }
        Masks Mn = Mc;
        if (a.K > 1) Mn = scan(rec_next, batch_size(1), 0);

        for (int k = 0; k < a.K; ++k) {
            const int64_t pos0 = (int64_t)k * a.B;
            const float inv_batch = 1.0f / (float)batch_size(k);
            // prefetch two steps ahead: consumed (scanned) at the end of this step, off the critical path
            mfcd_sample rec_nn = rec_next;
            if (k + 2 < a.K) rec_nn = load_record(a.samples, (int64_t)(k + 2) * a.B, batch_size(k + 2), 0, lane);
            const StepScalars sc_next = a.sc[k + 1 < a.K ? k + 1 : k];
#pragma unroll
            for (int q = 0; q < Q; ++q) gr[q] = 0.0f;

            // ---- critical: this step's hits ----
            u64 mask = Mc.mu | Mc.mi | Mc.mj;
            while (mask) {
                const int tl = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if (!process_hit(rec_cur, Mc, tl, pos0 + tl, (unsigned)k + 1u, inv_batch)) return;
            }
            [[maybe_unused]] const u64 t_adam0 = STAMP();
            // ---- critical: rows the next batch needs -> Adam on just those registers, then publish ----
            bool done[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) done[q] = false;
            if (k + 1 < a.K) {
                u64 pm = Mn.mu | Mn.mi | Mn.mj;
                const int64_t npos0 = pos0 + a.B;
                while (pm) {
                    const int tl = __ffsll((long long)pm) - 1;
                    pm &= pm - 1;
                    const int rows[3] = {__shfl(rec_next.u, tl, MFCD_WAVE), __shfl(rec_next.i, tl, MFCD_WAVE) + a.n,
                                         __shfl(rec_next.j, tl, MFCD_WAVE) + a.n};
                    const bool fl[3] = {(bool)((Mn.mu >> tl) & 1ull), (bool)((Mn.mi >> tl) & 1ull),
                                        (bool)((Mn.mj >> tl) & 1ull)};
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        if (!fl[r]) continue;
                        const int q0 = reg_of(rows[r]);
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            if (q >= q0 && q < q0 + S && !done[q]) {
                                adam_update(p[q], m1[q], m2[q], gr[q], a.ac, sc);
                                done[q] = true;
                            }
                        }
                        store_row(rows[r], (npos0 + tl) * 3 + r, (unsigned)k + 2u);
                    }
                }
            }
            [[maybe_unused]] const u64 t_adam1 = STAMP();
            DBG_ADD(4, t_adam1 - t_adam0);
            // ---- off the critical path: the rest of my slice, then the masks of batch k+2 ----
            // branch-free so the Q independent update chains interleave (a branch per register serialises them);
            // registers already updated above keep their values
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                float pn = p[q], m1n = m1[q], m2n = m2[q];
                adam_update(pn, m1n, m2n, gr[q], a.ac, sc);
                p[q] = done[q] ? p[q] : pn;
                m1[q] = done[q] ? m1[q] : m1n;
                m2[q] = done[q] ? m2[q] : m2n;
            }
            DBG_ADD(3, STAMP() - t_adam1);
            rec_cur = rec_next;
            rec_next = rec_nn;
            Mc = Mn;
            if (k + 2 < a.K) Mn = scan(rec_next, batch_size(k + 2), 0);
            sc = sc_next;
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
                    if ((M.mu >> tl) & 1ull) store_row(__shfl(s.u, tl, MFCD_WAVE), slot0 + 0, (unsigned)step + 1u);
                    if ((M.mi >> tl) & 1ull) store_row(__shfl(s.i, tl, MFCD_WAVE) + a.n, slot0 + 1, (unsigned)step + 1u);
                    if ((M.mj >> tl) & 1ull) store_row(__shfl(s.j, tl, MFCD_WAVE) + a.n, slot0 + 2, (unsigned)step + 1u);
                }
            }
        };
        publish(0);
        for (int k = 0; k < a.K; ++k) {
            const int64_t pos0 = (int64_t)k * a.B;
            const int Bk = batch_size(k);
            const float inv_batch = 1.0f / (float)Bk;
            const StepScalars sc = a.sc[k];
#pragma unroll
            for (int q = 0; q < Q; ++q) gr[q] = 0.0f;
            for (int base = 0; base < Bk; base += MFCD_WAVE) {
                const mfcd_sample s = load_record(a.samples, pos0, Bk, base, lane);
                const Masks M = scan(s, Bk, base);
                u64 mask = M.mu | M.mi | M.mj;
                while (mask) {
                    const int tl = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    if (!process_hit(s, M, tl, pos0 + base + tl, (unsigned)k + 1u, inv_batch)) return;
                }
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) adam_update(p[q], m1[q], m2[q], gr[q], a.ac, sc);
            if (k + 1 < a.K) publish(k + 1);
        }
    }

#ifdef MFCD_STAMPS
    dbg_acc[0] = STAMP() - t_start;
    if (lane == 0 && a.dbg)
        for (int x = 0; x < 8; ++x) a.dbg[(int64_t)gw * 8 + x] = dbg_acc[x];
#endif
    // ---- write my slice back ----
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t e = ebase + q * 64 + lane;
        if (e < T) {
            if (e < TU) { a.U[e] = p[q]; a.mU[e] = m1[q]; a.vU[e] = m2[q]; }
            else { a.V[e - TU] = p[q]; a.mV[e - TU] = m1[q]; a.vV[e - TU] = m2[q]; }
        }
    }
}

}  // namespace mfcd_detail
