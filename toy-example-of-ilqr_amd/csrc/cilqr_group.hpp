// cilqr_group.hpp — the solve kernel for large batches of short horizons: G trajectories per wavefront, ONE rollout pass
// for all of them.
//
// Why.  The line search's forward pass (cs:442-461) is a serial chain over the horizon; k_solve runs it with lane =
// trial step size, and 72 % of the iterations of the headline workload roll out alpha = 1 alone: one live lane of 64
// through ~170 vector instructions x 50 steps, a third of an iteration's cycles and 40 % of its vector instructions
// (profiles/r03_v6_phase_config5.json) — and a wave64 FP64 instruction costs the same issue cycles with one live lane as
// with 64.  Here a wavefront owns G trajectories at once.  Everything that is parallel over the horizon or over matrix
// elements (cost expansion, backward sweep, trial costs) runs for one trajectory at a time, exactly as in k_solve, in
// "segments"; a trajectory whose backward sweep has produced its gains REQUESTS a rollout and yields; after every
// trajectory of the wavefront has had its segment, one rollout pass serves all requests — lane = (trajectory, step size):
// G lanes for G first trials, 20 per trajectory whose search goes deeper.  The serial chain per trajectory-iteration
// shrinks by (1 - 1/G) of the rollout, and so does its share of vector instructions.
//
// What moves.  LDS holds per trajectory only what persists from segment to segment: x, u, the lane indices, the cost
// model's constants and the solve's scalars (GrpSt) — 3.2 KB at N = 50 — and ONE copy of what lives inside a segment:
// the model Jacobians, the cost expansion, the stage-cost scratch, the lane window (restaged when a segment begins).
// The gains (K, d) must survive until the pass after the segment: they go to global memory ([N][10] doubles per
// trajectory, written by the sweep's lanes with two buffer stores per step, read back by the rollout lanes one step ahead
// through L1 / L2).  The expansion of an unchanged trajectory (cs:469-475) is recomputed instead of kept — the same bits.
//
// Bit-exactness: every number is produced by the same device functions in the same order as in k_solve; only WHEN a
// trajectory's phases run changes.  Results, counters and traces equal k_solve's and the oracle's.
#pragma once
#include "cilqr_device.hpp"

namespace cilqr {

// in-kernel cycle accounting of the grouped build: development library only (cilqr_set_phase_profiling)
#ifdef CILQR_DEV_BUILD
#define CILQR_GPROF 1
#else
#define CILQR_GPROF 0
#endif

#ifndef CILQR_GRP_QLDS
#define CILQR_GRP_QLDS false /* true: Q_uu to the lanes through the LDS crossbar instead of v_readlane — measured 1 % slower (r04_experiments) */
#endif

enum { GP_EMPTY = 0, GP_ITER = 1, GP_SEARCH = 2, GP_DONE = 3, GP_STOLEN = 4 /* b names a parked trajectory to take over */ };

// the scalars cs:110-141 carries from one iteration to the next, plus where the line search stands
struct GrpSt {
    double J_cur, J_init, lamb, dV0, dV1, new_J, dt, wb;
    long long tl_start;
    double J_pair; // the second cost of a paired costing pass (grp_cost_trials2)
    int b, phase, status, iters, ls_trials, cost_evals, tl, flag, deep_next, idx0, t0, have_all, trials, req, nfb, rp;
    int small_steps; // (development aid) steps of the last rollout pass that ran the straight-line small-angle form
    int t_done;      // closed loop in one launch: ticks of this ego that are done
    int pad1, pad2;
};
static_assert(sizeof(GrpSt) == 160, "GrpSt layout");
#define CILQR_GRPST_DOUBLES 20
static_assert(sizeof(Cst) == 88, "Cst layout");
#define CILQR_CST_DOUBLES 12 /* the by-value constants (Cst) of the trajectory, kept so that a segment need not walk the tables again */

// ridx + two rows of tidx (the trials costed in pairs keep one row of lane-index guesses each): 3 (N + 2) ints, 16-byte granules
__host__ __device__ inline int grp_idx_doubles(int N) { return ((3 * (N + 2) + 1) / 2 + 1) & ~1; }
__host__ __device__ inline int grp_pg_doubles(int N) { // per trajectory
    return 4 * (N + 1) + 2 * N + grp_idx_doubles(N) + CILQR_CSTK_DOUBLES + CILQR_GRPST_DOUBLES + CILQR_CST_DOUBLES +
           (CILQR_GPROF ? CILQR_PROF_SLOTS + 1 : 0);
}
// One copy per wavefront: the Jacobians / stage-cost scratch, then an area that holds the cost expansion (+ the sweep's
// constants) from the expansion to the end of the backward sweep and the LANE WINDOW the rest of the time — the window is
// used where the expansion is dead (initial trajectory, line-search costs); the expansion's own single lane lookup per
// row goes to global memory next to the lane record it needs from there anyway.
__host__ __device__ inline int grp_expansion_doubles(int N) { return (4 * (N + 1) + 2 * N + 7 * (N + 1) + 2 * N) + CILQR_XCH; }
__host__ __device__ inline int grp_shared_doubles(int N, int W) {
    const int e = grp_expansion_doubles(N), w = 2 * W;
    return kd_doubles(N, 1) + (e > w ? e : w);
}
__host__ __device__ inline size_t grp_lds_bytes(int N, int W, int G) {
    return sizeof(double) * ((size_t)G * grp_pg_doubles(N) + (size_t)grp_shared_doubles(N, W));
}
// global scratch per trajectory slot: slab | first-trial buffer | gains, 128-byte granules
__host__ __device__ inline size_t grp_scratch_doubles(int N) {
    const size_t d = slab_doubles(N) + first_trial_doubles(N) + (size_t)CILQR_KD * (size_t)N;
    return (d + 15) / 16 * 16;
}

typedef double __attribute__((ext_vector_type(2))) f64x2;
typedef const f64x2 __attribute__((address_space(3))) lds_cf64x2;
typedef const f64x2 __attribute__((address_space(1))) f64x2g;

__device__ inline GrpSt* grp_state(double* base, int N, int g) {
    return reinterpret_cast<GrpSt*>(base + (size_t)g * grp_pg_doubles(N) + 4 * (N + 1) + 2 * N + grp_idx_doubles(N) + CILQR_CSTK_DOUBLES);
}

__device__ inline Cst* grp_cst(double* base, int N, int g) {
    return reinterpret_cast<Cst*>(reinterpret_cast<double*>(grp_state(base, N, g)) + CILQR_GRPST_DOUBLES);
}
__device__ inline long long* grp_prof(double* base, int N, int g) { // (development library) cycles per phase of the trajectory in slot g
    return reinterpret_cast<long long*>(reinterpret_cast<double*>(grp_state(base, N, g)) + CILQR_GRPST_DOUBLES + CILQR_CST_DOUBLES);
}
// the constants back from LDS, wave-uniform (scalar registers: addresses built from them stay scalar)
__device__ inline void load_cst_lds(Cst& c, const Cst* p) {
    const int* w = reinterpret_cast<const int*>(p);
    int v[sizeof(Cst) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Cst) / 4); ++i) v[i] = __builtin_amdgcn_readfirstlane(w[i]);
    __builtin_memcpy(&c, v, sizeof(Cst));
}

// ---------------------------------------------------------------------------------------------
// The tail of a launch.  When the trajectory counter has run dry a wavefront that still holds two trajectories advances each
// at half speed while wavefronts that have finished theirs sit idle — with solves that differ by a factor of ten in length,
// the launch would end with a few pairs of long ones (measured before this: config 3, two rounds of resident wavefronts,
// 13.3 ms in pairs against 12.0 one per wavefront).  So a wavefront that runs out of work asks for more: it leaves a ticket
// (SH_HELPING) and polls the queue of parked trajectories; a wavefront that holds two and sees a ticket takes it and PARKS
// one of them between two iterations — x, u, the lane indices and the scalars cs:110-141 carries over, 2.8 KB, written with
// 8-byte agent-scope atomics as in k_solve's resumable solves (rq_push / rq_pop, park_st / park_ld) — and the idle wavefront
// carries on with it from the next expansion.  Whoever runs an iteration computes the same bits.  One ticket, one parked
// trajectory; at most CILQR_GRP_MAX_WAITING wavefronts wait at a time (the others leave: a few polled lines must not be
// hammered by two thousand wavefronts); every wait is bounded.
#ifndef CILQR_GRP_MAX_WAITING
#define CILQR_GRP_MAX_WAITING 256
#endif
__host__ __device__ inline size_t grp_park_doubles(int N) { // x | u | GrpSt | lane indices
    return (size_t)(4 * (N + 1) + 2 * N + CILQR_GRPST_DOUBLES + (N + 2) / 2 + 1);
}
__device__ __attribute__((noinline)) void grp_park_copy(double* pk, double* lx, double* lu, int* ridx, GrpSt* st, int N, int lane,
                                                         int store) {
    double* const pk_sc = pk + 4 * (N + 1) + 2 * N;
    constexpr int NSC = CILQR_GRPST_DOUBLES;
    unsigned* const pk_ix = reinterpret_cast<unsigned*>(pk_sc + NSC);
    double* const sc = reinterpret_cast<double*>(st); // GrpSt as 18 eight-byte words
    if (store) {
        for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) park_st(pk + e, lx[e]);
        for (int e = lane; e < 2 * N; e += CILQR_WAVE) park_st(pk + 4 * (N + 1) + e, lu[e]);
        for (int k = lane; k <= N; k += CILQR_WAVE) sh_st(pk_ix + k, (unsigned)ridx[k]);
        if (lane < NSC) park_st(pk_sc + lane, sc[lane]);
    } else {
        for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) lx[e] = park_ld(pk + e);
        for (int e = lane; e < 2 * N; e += CILQR_WAVE) lu[e] = park_ld(pk + 4 * (N + 1) + e);
        for (int k = lane; k <= N; k += CILQR_WAVE) ridx[k] = (int)sh_ld(pk_ix + k);
        if (lane < NSC) sc[lane] = park_ld(pk_sc + lane);
    }
    wave_sync();
}

// An idle wavefront: leave a ticket, wait for a parked trajectory.  Returns its number, or -1 when the launch is over (every
// trajectory finished), too many are waiting already, or the bound on the wait is reached.
__device__ __attribute__((noinline)) int grp_wait_for_work(unsigned* ctl, const unsigned long long* q, unsigned cap, unsigned B, int lane) {
    if (sh_ld_u(ctl + SH_FINISHED, lane) >= B) return -1;
    if (sh_ld_u(ctl + SH_HELPING, lane) >= (unsigned)CILQR_GRP_MAX_WAITING) return -1; // (a look first)
    if (sh_add_u(ctl + SH_HELPING, 1u, lane) >= (unsigned)CILQR_GRP_MAX_WAITING) {
        (void)sh_add_u(ctl + SH_HELPING, 0u - 1u, lane);
        return -1;
    }
    (void)sh_add_u(ctl + SH_HELPERS, 1u, lane);
    for (int spin = 0; spin < (1 << 20); ++spin) { // (a launch lasts milliseconds; this bound is seconds)
        const int pb = rq_pop(ctl, q, cap, lane);
        if (pb >= 0) return pb; // (the ticket was taken by whoever parked it)
        if (sh_ld_u(ctl + SH_FINISHED, lane) >= B) return -1;
        __builtin_amdgcn_s_sleep(127);
    }
    if (lane == 0) sh_st(ctl + SH_ERROR, 1u);
    return -1;
}
// a wavefront with two trajectories: is somebody waiting?  Takes the ticket if so.
__device__ inline bool grp_take_ticket(unsigned* ctl, int lane) {
    for (int tries = 0; tries < 4; ++tries) {
        const unsigned v = sh_ld_u(ctl + SH_HELPING, lane);
        if (v == 0u || v > 0x7fffffffu) return false;
        if (sh_cas_u(ctl + SH_HELPING, v, v - 1u, lane) == v) return true;
    }
    return false;
}


__device__ inline void carve_group(Lds& l, double* base, int N, int G, int g) {
    double* p = base + (size_t)g * grp_pg_doubles(N);
    l.x = p; p += 4 * (N + 1);
    l.u = p; p += 2 * N;
    l.ridx = reinterpret_cast<int*>(p);
    l.tidx = l.ridx + (N + 2);
    p += grp_idx_doubles(N);
    l.ck = reinterpret_cast<CstK*>(p);
    double* s = base + (size_t)G * grp_pg_doubles(N);
    l.kd = s; s += kd_doubles(N, 1);
    l.cs = l.kd;
    l.lxs = 7;
    l.lx = s; s += 4 * (N + 1);
    l.lu = s; s += 2 * N;
    l.lxx = s; s += 7 * (N + 1);
    l.luu = s; s += 2 * N;
    l.xch = s; s += CILQR_XCH;
    l.win = l.lx; // (shares the expansion's area, see grp_shared_doubles)
    l.gl = nullptr;
    l.ring = nullptr;
    l.ctld = nullptr;
    l.ctli = nullptr;
    l.prof = nullptr;
    l.w0 = 0;
    l.W = 0;
}

// ---------------------------------------------------------------------------------------------
// backward_sweep_lanes (cilqr_device.hpp) for NS trajectories AT ONCE: the same step body, once per trajectory, on register
// sets of its own inside one loop, so that the wavefront has NS independent dependent chains to issue from (a step of one
// sweep is ~145 instructions on one chain: W gather, two 4-term passes, the 2 x 2 inverse with its IEEE division, the rank-2
// update — a lone wavefront keeps its SIMD half busy with it).  Every element is the expression backward_sweep_lanes
// evaluates, in its order: same bits.  base[s] = the trajectory's Lds::x (its arrays are addressed relative to it, as in
// make_lane_map), gains[s] = where its gains go (global memory), alive: a sweep that meets a non-PD Q_uu (cs:415-420) stops
// there — its later steps keep their Jacobians, as in the one-trajectory form — while the others go on.
// Returns the mask of the sweeps that completed.  (A sweep that fails — non-PD Q_uu, cs:415-420 — is only marked.)
template <int NS>
__device__ inline unsigned backward_sweep_lanes_multi(const Cst* c, const Lds* l, const double* lamb, int lane, double (*dV)[2],
                                                      double* const* gains) {
    const int N = c[0].N; // (one horizon per handle)
    const int rp = (lane >> 3) % 6, cc = lane & 7;
    LaneMap mp[NS];
    __amdgpu_buffer_rsrc_t ggr[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        make_lane_map<false>(l[s], lane, mp[s]);
        if (lane == 0) {
            l[s].xch[CILQR_XCH_CONST + 0] = 0.0;
            l[s].xch[CILQR_XCH_CONST + 1] = 1.0;
            l[s].xch[CILQR_XCH_CONST + 2] = c[s].dt;
            l[s].xch[CILQR_XCH_CONST + 3] = 0.0;
        }
        ggr[s] = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(gains[s]), 0, N * CILQR_KD * (int)sizeof(double), 0x00020000);
    }
    wave_sync();
    const int wc = (cc <= 4) ? cc : 4;
    const bool diag = (rp >= 4) && (cc == rp);
    const int r4 = rp & 3;
    const int src_c0 = 32 + wc, src_c1 = 40 + wc;
    const int src_r0 = (rp >= 4) ? 36 : 32 + r4, src_r1 = (rp >= 4) ? 44 : 40 + r4;
    const double krf = (rp == 4) ? 0.5 : 1.0;
    double wn[NS], dvacc[NS];
    unsigned am1[NS][4], am2[NS][4], alq[NS], alv[NS];
    unsigned dm1[NS][4], dm2[NS][4], dlq[NS], dlv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double* const base = l[s].x;
        wn[s] = (cc < 4) ? base[mp[s].lq + mp[s].slq * N] : base[mp[s].lv + mp[s].slv * N];
        dvacc[s] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            am1[s][k] = lds_addr(base + mp[s].m1[k] + mp[s].s1[k] * (N - 1));
            am2[s][k] = lds_addr(base + mp[s].m2[k] + mp[s].s2[k] * (N - 1));
            dm1[s][k] = 8u * (unsigned)mp[s].s1[k];
            dm2[s][k] = 8u * (unsigned)mp[s].s2[k];
        }
        alq[s] = lds_addr(base + mp[s].lq + mp[s].slq * (N - 1));
        alv[s] = lds_addr(base + mp[s].lv + mp[s].slv * (N - 1));
        dlq[s] = 8u * (unsigned)mp[s].slq;
        dlv[s] = 8u * (unsigned)mp[s].slv;
    }
    // One straight-line block per step: phase by phase over the sweeps, so that the scheduler can interleave their chains.  A
    // sweep whose Q_uu turns out not to be positive definite is only MARKED: it keeps running on whatever numbers it has (its
    // gains are never used: the iteration ends as BACKWARD_PASS_FAIL and expands afresh; no address leaves its arrays), which
    // keeps the loop free of per-sweep control flow.  The exact pivot test (sqrt, quotient) is one rare branch for all sweeps.
    unsigned failed = 0u;
    for (int i = N - 1; i >= 0; --i) {
        double Q[NS], Zv[NS], S[NS], c0[NS], c1[NS], r0[NS], r1[NS], invdet[NS], Quu0[NS], Quu1[NS], Quu2[NS], Quu3[NS];
        double m1[NS][4], m2[NS][4], Lq[NS], lv[NS], w0[NS], w1[NS], w2[NS], w3[NS], X[NS], Xq[NS];
        unsigned need_exact = 0u;
        int ccv = cc;
        __asm__("" : "+v"(ccv));
        // (sub-phase by sub-phase over the sweeps — written out in the order the chains should be issued: every LDS round
        //  trip of one sweep in flight together with the other's)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m1[s][k] = lds_load(am1[s][k]);
                m2[s][k] = lds_load(am2[s][k]);
                am1[s][k] -= dm1[s][k];
                am2[s][k] -= dm2[s][k];
            }
            Lq[s] = lds_load(alq[s]);
            lv[s] = lds_load(alv[s]);
            alq[s] -= dlq[s];
            alv[s] -= dlv[s];
            w0[s] = lane_gather(wn[s], wc); w1[s] = lane_gather(wn[s], 8 + wc);
            w2[s] = lane_gather(wn[s], 16 + wc); w3[s] = lane_gather(wn[s], 24 + wc);
        }
        // (statement by statement over the sweeps: consecutive instructions belong to different chains)
#define EACH_S for (int s = 0; s < NS; ++s)
        double tX[NS], x0[NS], x1[NS], x2[NS], x3[NS], tY[NS], det[NS];
#pragma unroll
        EACH_S tX[s] = m1[s][0] * w0[s];
#pragma unroll
        EACH_S tX[s] = CQ_MADD(m1[s][1], w1[s], tX[s]);
#pragma unroll
        EACH_S tX[s] = CQ_MADD(m1[s][2], w2[s], tX[s]);
#pragma unroll
        EACH_S X[s] = CQ_MADD(m1[s][3], w3[s], tX[s]);
#pragma unroll
        EACH_S Zv[s] = lv[s] + X[s];
#pragma unroll
        EACH_S Xq[s] = dpp_move_banks<0x114, 0xA>(X[s]);
#pragma unroll
        EACH_S x0[s] = dpp_move<0x00>(Xq[s]);
#pragma unroll
        EACH_S x1[s] = dpp_move<0x55>(Xq[s]);
#pragma unroll
        EACH_S x2[s] = dpp_move<0xAA>(Xq[s]);
#pragma unroll
        EACH_S x3[s] = dpp_move<0xFF>(Xq[s]);
#pragma unroll
        EACH_S tY[s] = x0[s] * m2[s][0];
#pragma unroll
        EACH_S tY[s] = CQ_MADD(x1[s], m2[s][1], tY[s]);
#pragma unroll
        EACH_S tY[s] = CQ_MADD(x2[s], m2[s][2], tY[s]);
#pragma unroll
        EACH_S tY[s] = CQ_MADD(x3[s], m2[s][3], tY[s]);
#pragma unroll
        EACH_S Q[s] = Lq[s] + tY[s];
#pragma unroll
        EACH_S { if (diag) Q[s] = Q[s] + lamb[s]; }
#pragma unroll
        EACH_S S[s] = (ccv == 4) ? Zv[s] : Q[s];
#pragma unroll
        EACH_S { c0[s] = lane_gather(S[s], src_c0); c1[s] = lane_gather(S[s], src_c1); }
#pragma unroll
        EACH_S { r0[s] = lane_gather(S[s], src_r0); r1[s] = lane_gather(S[s], src_r1); }
#pragma unroll
        EACH_S { Quu0[s] = lane_bcast<36>(Q[s]); Quu1[s] = lane_bcast<37>(Q[s]); Quu2[s] = lane_bcast<44>(Q[s]); Quu3[s] = lane_bcast<45>(Q[s]); }
#pragma unroll
        EACH_S det[s] = Quu0[s] * Quu3[s] - Quu2[s] * Quu1[s];
#pragma unroll
        EACH_S invdet[s] = 1.0 / det[s];
#pragma unroll
        EACH_S {
            const unsigned h0 = (unsigned)(dm_to_bits(Quu0[s]) >> 32), h3 = (unsigned)(dm_to_bits(Quu3[s]) >> 32);
            const bool ordinary = ((h0 - 0x2B300000u) < 0x29800000u) && ((h3 - 0x2B300000u) < 0x29800000u);
            const bool surely_pd = ordinary && (Quu0[s] * Quu3[s] > (Quu2[s] * Quu2[s]) * 1.0000000000009095);
            if (!surely_pd) need_exact |= (1u << s);
        }
        if (need_exact != 0u) { // Eigen::LLT's verdict evaluated exactly (see backward_sweep_lanes)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!(need_exact & (1u << s)) || (failed & (1u << s))) continue;
                bool fail = false;
                if (Quu0[s] <= 0.0) {
                    fail = true;
                } else {
                    double l00 = dm_sqrt(Quu0[s]);
                    double l10 = Quu2[s] / l00;
                    double piv1 = Quu3[s] - l10 * l10;
                    if (piv1 <= 0.0) fail = true;
                }
                if (fail) failed |= (1u << s);
            }
            if (failed == (1u << NS) - 1u) break; // nobody left
        }
        double kc0[NS], kc1[NS], n00[NS], n01[NS], n10[NS], n11[NS], kr0[NS], kr1[NS], p0[NS], p1[NS], ta[NS], tb[NS], tc[NS];
        int rpv = rp;
        __asm__("" : "+v"(rpv));
#pragma unroll
        EACH_S { n00[s] = -(Quu3[s] * invdet[s]); n01[s] = -(-Quu1[s] * invdet[s]); n10[s] = -(-Quu2[s] * invdet[s]); n11[s] = -(Quu0[s] * invdet[s]); }
#pragma unroll
        EACH_S { kc0[s] = n00[s] * c0[s]; kc1[s] = n10[s] * c0[s]; kr0[s] = n00[s] * r0[s]; kr1[s] = n10[s] * r0[s]; }
#pragma unroll
        EACH_S { kc0[s] = CQ_MADD(n01[s], c1[s], kc0[s]); kc1[s] = CQ_MADD(n11[s], c1[s], kc1[s]);
                 kr0[s] = CQ_MADD(n01[s], r1[s], kr0[s]); kr1[s] = CQ_MADD(n11[s], r1[s], kr1[s]); }
#pragma unroll
        EACH_S { kr0[s] = kr0[s] * krf; kr1[s] = kr1[s] * krf; }
#pragma unroll
        EACH_S { p0[s] = kr0[s] * Quu0[s]; p1[s] = kr0[s] * Quu1[s]; tb[s] = kr0[s] * c0[s]; tc[s] = r0[s] * kc0[s]; }
#pragma unroll
        EACH_S { p0[s] = CQ_MADD(kr1[s], Quu2[s], p0[s]); p1[s] = CQ_MADD(kr1[s], Quu3[s], p1[s]);
                 tb[s] = CQ_MADD(kr1[s], c1[s], tb[s]); tc[s] = CQ_MADD(r1[s], kc1[s], tc[s]); }
#pragma unroll
        EACH_S ta[s] = p0[s] * kc0[s];
#pragma unroll
        EACH_S ta[s] = CQ_MADD(p1[s], kc1[s], ta[s]);
#pragma unroll
        EACH_S { const double own = (ccv < 4) ? Q[s] : Zv[s]; wn[s] = own + ta[s]; }
#pragma unroll
        EACH_S wn[s] = wn[s] + tb[s];
#pragma unroll
        EACH_S wn[s] = wn[s] + tc[s];
#pragma unroll
        EACH_S dvacc[s] = dvacc[s] + ((rpv == 4) ? ta[s] : tb[s]);
#undef EACH_S
        int lanev = lane;
        __asm__("" : "+v"(lanev));
        if (lanev < 5) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, kc0[s]), ggr[s], 8u * (unsigned)lane, i * (CILQR_KD * 8), 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, kc1[s]), ggr[s], 8u * (unsigned)lane + 8u * CILQR_KD_ROW,
                                                      i * (CILQR_KD * 8), 0);
            }
        }
    }
    const unsigned alive = ((1u << NS) - 1u) & ~failed;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        dV[s][0] = lane_bcast<36>(dvacc[s]);
        dV[s][1] = lane_bcast<44>(dvacc[s]);
    }
    wave_sync();
    return alive;
}

// ---------------------------------------------------------------------------------------------
// Expansion + backward sweep of the trajectory in slot g (cs:463-690, cs:383-440), OUT OF LINE: everything it needs is in
// LDS (x, u, lane indices, constants) or an argument, everything it produces goes to LDS (Jacobians, expansion — consumed
// inside —, the expected cost reduction) or global memory (the gains), so the call carries nothing and the sweep's serial
// loop gets a register allocation of its own.  Inlined into the kernel's state machine the loop picked up a scratch reload
// per step whenever the code around it grew (152 -> 164 instructions a step, backward 69 k -> 85 k cycles per iteration).
// prof: development library, cycle accounting (PH_DERIV = 1, PH_BACKWARD = 2, PH_TOTAL = 6 of the slot's accumulators).
template <int NC, int G>
__device__ __attribute__((noinline)) bool grp_expand_backward(double* lds, int g, int n_rt, int lane, double lamb, double* gains,
                                                               long long* prof, int dual_probe = 0) {
    const int N = NC ? NC : uniform_int(n_rt); // (an argument: in a vector register — scalar again, or descriptors built from it count as divergent)
    Lds l;
    carve_group(l, lds, N, G, g);
    Cst c;
    load_cst_lds(c, grp_cst(lds, N, g));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    long long t0 = (CILQR_GPROF && prof) ? (long long)__builtin_readcyclecounter() : 0;
    l.W = 0; // the lane window gives way to the expansion (the rows' one lane lookup each: global memory)
    // (always expanded afresh: the expansion area is shared by the wavefront's trajectories; after a failed pass the
    //  reference keeps the old one, cs:469-475 — same trajectory, same bits)
    cost_and_model_derivatives<false, false>(c, l, al, lane);
    if (CILQR_GPROF && prof) {
        const long long t1 = (long long)__builtin_readcyclecounter();
        if (lane == 0) { prof[1] += t1 - t0; prof[6] += t1 - t0; }
        t0 = t1;
    }
    double dV[2];
    bool ok;
    if (CILQR_GPROF && dual_probe) {
        // development probe (profiles/r04_experiments): the SAME sweep twice in one loop, two independent chains — what would
        // two trajectories' sweeps cost side by side?  (identical inputs, identical outputs, the gains stored twice)
        const Cst c2[2] = {c, c};
        const Lds l2[2] = {l, l};
        const double lamb2[2] = {lamb, lamb};
        double dV2[2][2];
        double* const gains2[2] = {gains, gains};
        const unsigned done = backward_sweep_lanes_multi<2>(c2, l2, lamb2, lane, dV2, gains2);
        ok = (done & 1u) != 0u;
        dV[0] = dV2[0][0];
        dV[1] = dV2[0][1];
    } else {
        ok = backward_sweep_lanes<0, true, CILQR_GRP_QLDS>(c, l, lamb, lane, dV, nullptr, gains);
    }
    if (lane == 0) {
        GrpSt* st = grp_state(lds, N, g);
        st->dV0 = dV[0];
        st->dV1 = dV[1];
    }
    wave_sync();
    if (CILQR_GPROF && prof) {
        const long long t1 = (long long)__builtin_readcyclecounter();
        if (lane == 0) { prof[2] += t1 - t0; prof[6] += t1 - t0; }
    }
    return ok;
}

// get_total_cost (cs:199-287) of trial t of the trajectory in slot g — out of line for the same reason: the trial lives in
// global memory (src / as: the slab or the first-trial buffer), x's lane window is staged (w0, W), the result is the return
// value; serial reference-point chains that had to be run are counted in GrpSt::nfb.
template <int NC, int G>
__device__ __attribute__((noinline)) double grp_cost_trial(double* lds, int g, int n_rt, int lane, const double* src, int t, int as,
                                                            int w0, int W) {
    const int N = NC ? NC : uniform_int(n_rt); // (an argument: in a vector register — scalar again, or descriptors built from it count as divergent)
    Lds l;
    carve_group(l, lds, N, G, g);
    l.w0 = w0;
    l.W = W;
    Cst c;
    load_cst_lds(c, grp_cst(lds, N, g));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    int nfb = 0;
    double J1[1];
    total_cost_trials<false, 1, false, 1>(c, l, al, src, t, 1, lane, w0, 0, &nfb, J1, nullptr, 0, as);
    if (nfb != 0 && lane == 0) grp_state(lds, N, g)->nfb += nfb;
    return J1[0];
}

// Two trials at once, t and t + 1 (slab only): their reference-point searches, row loads and barrier chains overlap inside
// the wavefront — the second cost comes back in GrpSt::J_pair.  A search that has gone past its first trial usually goes on
// (14 % of the headline's iterations try all 20 step sizes: 55 % of all trial costs), so the second cost is rarely wasted;
// k_solve's two-per-SIMD builds could not afford the registers of the paired form (NTP = 2: 68 spilled), a function of
// its own can.
template <int NC, int G>
__device__ __attribute__((noinline)) double grp_cost_trials2(double* lds, int g, int n_rt, int lane, const double* src, int t,
                                                              int w0, int W) {
    const int N = NC ? NC : uniform_int(n_rt); // (an argument: in a vector register — scalar again, or descriptors built from it count as divergent)
    Lds l;
    carve_group(l, lds, N, G, g);
    l.w0 = w0;
    l.W = W;
    Cst c;
    load_cst_lds(c, grp_cst(lds, N, g));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    int nfb = 0;
    double J2[2];
    total_cost_trials<false, 1, false, 2>(c, l, al, src, t, 2, lane, w0, 0, &nfb, J2, nullptr, 0, CILQR_MAX_ALPHA_TRIALS);
    if (lane == 0) {
        GrpSt* st = grp_state(lds, N, g);
        st->J_pair = J2[1];
        if (nfb != 0) st->nfb += nfb;
    }
    wave_sync();
    return J2[0];
}

// The initial trajectory of the trajectory in slot g and its cost (cs:155-197, cs:104): fills x, u, the lane indices and
// the trial-index seeds; the row-0 lane index comes back in *idx0_out (LDS: GrpSt::idx0).  Once per solve: out of line so
// that its serial rollout's register needs stay out of the kernel's.
template <int NC, int G>
__device__ __attribute__((noinline)) double grp_init(double* lds, int g, int n_rt, int lane, double xs0, double xs1, double xs2,
                                                      double xs3, const double* last_u, int Wcap) {
    const int N = NC ? NC : uniform_int(n_rt); // (an argument: in a vector register — scalar again, or descriptors built from it count as divergent)
    Lds l;
    carve_group(l, lds, N, G, g);
    Cst c;
    load_cst_lds(c, grp_cst(lds, N, g));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    const double xs[4] = {xs0, xs1, xs2, xs3};
    int idx0 = 0;
    init_trajectory(c, l, xs, last_u, lane, idx0, Wcap);
    seed_trial_indices(l, N, 2, lane);
    const double J = total_cost_lds<false>(c, l, al, lane);
    if (lane == 0) grp_state(lds, N, g)->idx0 = idx0;
    wave_sync();
    return J;
}

// stage_window() for a window that is staged once per SEGMENT instead of once per solve: 16-byte loads, all of a lane's
// loads in flight before its first LDS store (the plain loop is one 8-byte load, one wait, one store per trip: 14 round
// trips to L2 for a 432-sample window — 9.7 k cycles per iteration of the grouped build before this).
__device__ inline void stage_window_fast(const Cst& c, Lds& l, int w0, int Wcap, int lane) {
    int W = c.L - w0;
    W = (W < Wcap) ? W : Wcap;
    const f64x2g* src = reinterpret_cast<const f64x2g*>(c.lane_xy + 2 * (size_t)w0); // (16-byte aligned: lane_xy is, a sample is 16 bytes)
    f64x2* dst = reinterpret_cast<f64x2*>(l.win);
    constexpr int CH = 5; // samples per lane and round: 320 per round
    for (int base = 0; base < W; base += CH * CILQR_WAVE) {
        f64x2 v[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            const int e = base + t * CILQR_WAVE + lane;
            v[t] = (e < W) ? src[e] : f64x2{0.0, 0.0};
        }
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            const int e = base + t * CILQR_WAVE + lane;
            if (e < W) dst[e] = v[t];
        }
    }
    l.w0 = w0;
    l.W = W;
    wave_sync();
}

// ordering point between the lanes of one wavefront for LDS traffic only: the wave's LDS operations complete, global
// stores still in flight (gains, trace records) are NOT waited for
__device__ inline void lds_sync() {
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __asm__ volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// forward_pass (cs:442-461) for the lanes of several trajectories at once.  Everything rollout_trials_rp keeps in scalar
// registers because it is the same for every lane — where the nominal trajectory and the gains are, where the trial goes,
// dt and the wheelbase — is a per-lane value here.  Same operations in the same order: same bits.
struct GrpRoll {
    double alpha, dt, wb;
    unsigned xaddr, uaddr; // LDS byte addresses of the trajectory's x, u
    unsigned kaddr;        // ... of its gains when they are staged for the pass (rollout_group), else:
    unsigned goff;         // byte offset of its gains in the block's scratch area
    unsigned vrow0;        // ... of row 0, pair 0 of this lane's trial
    unsigned pairb, tileb; // pair / row-tile stride of the destination (slab: 20 step sizes per tile, first-trial buffer: 1)
};

template <bool STAGE>
__device__ inline void roll_fetch_g(RollIn& g, __amdgpu_buffer_rsrc_t rs, const GrpRoll& q, int i) {
    const unsigned xa = q.xaddr, ua = q.uaddr, ka = q.kaddr;
    const f64x2 a = *(lds_cf64x2*)(size_t)(xa + 32u * (unsigned)i);
    const f64x2 b = *(lds_cf64x2*)(size_t)(xa + 32u * (unsigned)i + 16u);
    const f64x2 u = *(lds_cf64x2*)(size_t)(ua + 16u * (unsigned)i);
    g.x[0] = a.x; g.x[1] = a.y; g.x[2] = b.x; g.x[3] = b.y;
    g.u[0] = u.x; g.u[1] = u.y;
#pragma unroll
    for (int j = 0; j < CILQR_KD / 2; ++j) {
        if (STAGE) {
            const f64x2 v = *(lds_cf64x2*)(size_t)(ka + (unsigned)(CILQR_KD * 8) * (unsigned)i + 16u * (unsigned)j);
            g.k[2 * j] = v.x;
            g.k[2 * j + 1] = v.y;
        } else { // straight from global memory (L2), one step ahead like the rest
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, q.goff, i * (CILQR_KD * 8) + 16 * j, 0);
            u32x2 lo, hi;
            lo.x = v.x; lo.y = v.y; hi.x = v.z; hi.y = v.w;
            g.k[2 * j] = __builtin_bit_cast(double, lo);
            g.k[2 * j + 1] = __builtin_bit_cast(double, hi);
        }
    }
}

struct GrpOut {
    unsigned bx0, bx1, bu; // byte offsets of row 0 of the pairs (x0 x1), (x2 x3), (u0 u1) of this lane's trial
};
// row k of a pair: its tile (a per-lane stride: the lanes of a pass write slabs and first-trial buffers side by side) in the
// vector offset, its place inside the tile — the same for every lane — in the scalar offset
__device__ inline void slab_st2_row(__amdgpu_buffer_rsrc_t rs, unsigned b, unsigned tileb, int k, double v0, double v1) {
    const u32x2 lo_ = __builtin_bit_cast(u32x2, v0), hi_ = __builtin_bit_cast(u32x2, v1);
    u32x4 q_;
    q_.x = lo_.x; q_.y = lo_.y; q_.z = hi_.x; q_.w = hi_.y;
#ifdef CILQR_LOSTROWS_REPRO
    const int ku = k; // (the form that lost rows, see rollout_group)
#else
    const int ku = opaque_uniform(k); // (see slab_row_off)
#endif
    const unsigned voff = __umul24((unsigned)(ku / CILQR_SLAB_TILE), tileb) + b;
    __builtin_amdgcn_raw_buffer_store_b128(q_, rs, voff, (ku % CILQR_SLAB_TILE) * 16, 0);
}
__device__ inline void slab_st2(__amdgpu_buffer_rsrc_t rs, unsigned voff, double v0, double v1) {
    const u32x2 lo_ = __builtin_bit_cast(u32x2, v0), hi_ = __builtin_bit_cast(u32x2, v1);
    u32x4 q_;
    q_.x = lo_.x; q_.y = lo_.y; q_.z = hi_.x; q_.w = hi_.y;
    __builtin_amdgcn_raw_buffer_store_b128(q_, rs, voff, 0, 0);
}
template <int RP, bool SMALL, int PIN>
__device__ inline bool roll_step_g(const GrpRoll& q, __amdgpu_buffer_rsrc_t rs, const DmPinned& pk, const RollIn& g, double xc[4],
                                   const GrpOut& o, int i) {
    const double dx0 = xc[0] - g.x[0], dx1 = xc[1] - g.x[1], dx2 = xc[2] - g.x[2], dx3 = xc[3] - g.x[3];
    const double k0 = CQ_MADD(g.k[3], dx3, CQ_MADD(g.k[2], dx2, CQ_MADD(g.k[1], dx1, g.k[0] * dx0)));
    const double k1 = CQ_MADD(g.k[8], dx3, CQ_MADD(g.k[7], dx2, CQ_MADD(g.k[6], dx1, g.k[5] * dx0)));
    double un[2];
    un[0] = CQ_MADD(q.alpha, g.k[CILQR_KD_D(0)], g.u[0] + k0);
    un[1] = CQ_MADD(q.alpha, g.k[CILQR_KD_D(1)], g.u[1] + k1);
    double xn[4];
    if (SMALL) {
        if (!DM_WAVE_ALL(__builtin_fabs(xc[3]) < 0.785 && __builtin_fabs(un[1]) < 0.7)) return false;
        if (!propagate_small_v<RP, PIN>(q.dt, q.wb, xc, un, xn, &pk)) return false;
    } else {
        propagate_v<RP, PIN | DM_NOSHORT>(q.dt, q.wb, xc, un, xn, &pk);
    }
    slab_st2_row(rs, o.bu, q.tileb, i, un[0], un[1]);
    slab_st2_row(rs, o.bx0, q.tileb, i + 1, xn[0], xn[1]);
    slab_st2_row(rs, o.bx1, q.tileb, i + 1, xn[2], xn[3]);
    xc[0] = xn[0]; xc[1] = xn[1]; xc[2] = xn[2]; xc[3] = xn[3];
    return true;
}

// the lanes of one vehicle model (the others are masked off by the caller's branch)
template <int RP, int PIN, bool STAGE>
__device__ inline int rollout_group_rp(int N, __amdgpu_buffer_rsrc_t rs, const GrpRoll& q) {
    const f64x2 a0 = *(lds_cf64x2*)(size_t)(q.xaddr);
    const f64x2 b0 = *(lds_cf64x2*)(size_t)(q.xaddr + 16u);
    double xc[4] = {a0.x, a0.y, b0.x, b0.y};
    slab_st2(rs, q.vrow0, xc[0], xc[1]);
    slab_st2(rs, q.vrow0 + q.pairb, xc[2], xc[3]);
    GrpOut o;
    o.bx0 = q.vrow0;
    o.bx1 = q.vrow0 + q.pairb;
    o.bu = q.vrow0 + 2u * q.pairb;
    // as in rollout_trials_rp: a straight-line small-angle loop that hands over to the general loop at the first step that
    // does not qualify on some lane; gains and nominal point of step i + 1 fetched while step i computes; two register sets
    DmPinned pk;
    if (PIN) dm_pin_load(pk);
    int i = 0;
    {
        RollIn ga, gb;
        roll_fetch_g<STAGE>(ga, rs, q, 0);
        for (;;) {
            if (i >= N) break;
            if (i + 1 < N) roll_fetch_g<STAGE>(gb, rs, q, i + 1);
            if (!roll_step_g<RP, true, PIN>(q, rs, pk, ga, xc, o, i)) break;
            ++i;
            if (i >= N) break;
            if (i + 1 < N) roll_fetch_g<STAGE>(ga, rs, q, i + 1);
            if (!roll_step_g<RP, true, PIN>(q, rs, pk, gb, xc, o, i)) break;
            ++i;
        }
    }
    const int i_small = i;
    if (i < N) {
        RollIn ga, gb;
        roll_fetch_g<STAGE>(ga, rs, q, i);
        for (;;) {
            if (i + 1 < N) roll_fetch_g<STAGE>(gb, rs, q, i + 1);
            roll_step_g<RP, false, PIN>(q, rs, pk, ga, xc, o, i);
            ++i;
            if (i >= N) break;
            if (i + 1 < N) roll_fetch_g<STAGE>(ga, rs, q, i + 1);
            roll_step_g<RP, false, PIN>(q, rs, pk, gb, xc, o, i);
            ++i;
            if (i >= N) break;
        }
    }
    return i_small;
}

// One pass for every trajectory of the wavefront that has asked for one (GrpSt::req: 1 = the first trial alone into the
// first-trial buffer, 2 = all 20 step sizes into the slab).  Lanes are dealt out in trajectory order; the requests of a
// pass never exceed 64 lanes for G <= 3.  Returns false when nobody asked.
// Out of line: at the call site nothing of a solve is live in registers (a trajectory's state is in LDS between its
// segments), so the call costs nothing and the pass gets a register allocation of its own — inlined into the kernel its
// loops carried scratch reloads (2 per step) and 16 lane moves of spilled scalars per step.
// STAGE: the gains of the pass are copied into the wavefront's shared LDS area first (two trajectories fit); otherwise the
// lanes read them from global memory one step ahead.
template <int G, int PIN, bool STAGE>
__device__ __attribute__((noinline)) bool rollout_group(double* lds_base, double* scr_blk, int N_arg, int lane) {
    // (arguments of an out-of-line function arrive in vector registers: without this the horizon — and with it the loop
    //  counters and the buffer descriptor built from it — counts as divergent, and every slab store becomes a loop over the
    //  distinct descriptors of the lanes)
#ifdef CILQR_LOSTROWS_REPRO
    // EXPERIMENT ONLY (-DCILQR_LOSTROWS_REPRO, never in a shipped library): the pass as round 4's first tiled build had it —
    // the horizon left in its vector register, so that the descriptor built from it counts as divergent and every slab store
    // sits in a waterfall loop — the shape that delivered a later content of a store's data register on gfx950 with XNACK off
    // (profiles/r04_experiments/tiled_slab_lost_rows.txt; scripts/probes/lost_rows_failing_pass.s is this function's code)
    const int N = N_arg;
#else
    const int N = uniform_int(N_arg);
#endif
    const int R = N + 1;
    int start = 0, gl = -1, al = 0, rq = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        GrpSt* st = grp_state(lds_base, N, g);
        const int req = uniform_int(st->req);
        const int n = (req == 2) ? CILQR_MAX_ALPHA_TRIALS : (req == 1 ? 1 : 0);
        if (lane >= start && lane < start + n) { gl = g; al = lane - start; rq = req; }
        start += n;
    }
    if (start == 0) return false;
    wave_sync(); // (the sweeps' gain stores have completed: they were issued a segment ago)
    // The gains of the pass, from global memory into the wavefront's shared LDS area — free between segments: Jacobians,
    // expansion and stage-cost scratch are dead — in one coalesced sweep, so that the serial loop reads them like k_solve's
    // does.  (Read straight from global memory one step ahead, the loop ran at the latency of a load from the fabric behind
    // the slab stores: SQ_WAIT_ANY + 59 %, the launch 4 % SLOWER than one trajectory per wavefront.)
    double* const stage = lds_base + (size_t)G * grp_pg_doubles(N);
    static_assert(G <= 3, "lanes: three searches of 20 step sizes fit a wavefront");
    static_assert(!STAGE || G <= 2, "the shared area holds the gains of two trajectories (2 x 10 N <= 10 N + 15 N + 15 doubles)");
#pragma unroll
    for (int g = 0; STAGE && g < G; ++g) {
        if (uniform_int(grp_state(lds_base, N, g)->req) == 0) continue;
        const double* src = scr_blk + (size_t)g * grp_scratch_doubles(N) + slab_doubles(N) + first_trial_doubles(N);
        const f64x2* s2 = reinterpret_cast<const f64x2*>(src);
        f64x2* d2 = reinterpret_cast<f64x2*>(stage + (size_t)g * CILQR_KD * N);
        for (int e = lane; e < CILQR_KD * N / 2; e += CILQR_WAVE) d2[e] = s2[e];
    }
    wave_sync();
    if (gl >= 0) {
        double* pg = lds_base + (size_t)gl * grp_pg_doubles(N);
        const GrpSt* st = grp_state(lds_base, N, gl);
        GrpRoll q;
        q.alpha = dm_pow2i(-al);
        q.dt = st->dt;
        q.wb = st->wb;
        const int rp = st->rp;
        q.xaddr = lds_addr(pg);
        q.uaddr = lds_addr(pg + 4 * R);
        const unsigned as = (rq == 2) ? (unsigned)CILQR_MAX_ALPHA_TRIALS : 1u;
        const unsigned gbase = (unsigned)gl * (unsigned)(grp_scratch_doubles(N) * sizeof(double));
        q.kaddr = lds_addr(stage + (size_t)gl * CILQR_KD * N);
        q.goff = gbase + (unsigned)((slab_doubles(N) + first_trial_doubles(N)) * sizeof(double));
        q.tileb = as * (unsigned)(2 * CILQR_SLAB_TILE * sizeof(double));
        q.pairb = (unsigned)CILQR_SLAB_RT(R) * q.tileb;
        q.vrow0 = gbase + ((rq == 2) ? 0u : (unsigned)(slab_doubles(N) * sizeof(double))) +
                  (unsigned)(2 * CILQR_SLAB_TILE * sizeof(double)) * (unsigned)al;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)uniform_ptr(scr_blk), 0, (int)(G * grp_scratch_doubles(N) * sizeof(double)), 0x00020000);
        // one loop pair per vehicle model: only that model's polynomial constants are live inside it
        int i_small;
        if (rp == 0) i_small = rollout_group_rp<0, PIN, STAGE>(N, rs, q);
        else i_small = rollout_group_rp<1, PIN, STAGE>(N, rs, q);
        if (CILQR_GPROF && al == 0) grp_state(lds_base, N, gl)->small_steps = i_small;
    }
    wave_sync();
    if (lane < G) grp_state(lds_base, N, lane)->req = 0;
    wave_sync();
    return true;
}

} // namespace cilqr
