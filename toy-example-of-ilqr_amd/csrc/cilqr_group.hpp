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

enum { GP_EMPTY = 0, GP_ITER = 1, GP_SEARCH = 2, GP_DONE = 3, GP_STOLEN = 4 /* b names a parked trajectory to take over */,
       GP_CLAIMED = 7 /* pad2 = the slot's place in the queue of parked trajectories (rq_claim): its entry is awaited */,
       GP_EXPAND = 5 /* at the head of an iteration: its expansion and sweep run after every trajectory's segment (grp_expand, grp_sweep) */,
       GP_BPF = 6 /* that sweep met a non-PD Q_uu: back in solve with BACKWARD_PASS_FAIL */ };

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
// (+ the ring through which the OTHER trajectory's rows stream during a sweep of both, see backward_sweep_pair: it sits behind the
//  expansion, inside the room the lane window uses the rest of the time)
__host__ __device__ inline int grp_expansion_doubles(int N) { return (4 * (N + 1) + 2 * N + 7 * (N + 1) + 2 * N) + CILQR_XCH + CILQR_GL_RING; }
__host__ __device__ inline int grp_shared_doubles(int N, int W) {
    const int e = grp_expansion_doubles(N), w = 2 * W;
    return kd_doubles(N, 1) + (e > w ? e : w);
}
__host__ __device__ inline size_t grp_lds_bytes(int N, int W, int G) {
    return sizeof(double) * ((size_t)G * grp_pg_doubles(N) + (size_t)grp_shared_doubles(N, W));
}
// global scratch per trajectory slot: slab | first-trial buffer | gains, 128-byte granules
// ... | rows of the expansion + Jacobians when the trajectory's sweep input streams from global memory (backward_sweep_pair)
__host__ __device__ inline size_t grp_rows_offset(int N) {
    const size_t d = slab_doubles(N) + first_trial_doubles(N) + (size_t)CILQR_KD * (size_t)N;
    return (d + 31) / 32 * 32;
}
__host__ __device__ inline size_t grp_scratch_doubles(int N) {
    return grp_rows_offset(N) + (size_t)CILQR_GRP_ROW * (size_t)(N + 1);
}

// ---------------------------------------------------------------------------------------------
// Round 5: two trajectories per wavefront for horizons of 64 ... 127 as well (two rows per lane) — the LONG layout.
// At N = 100 what persists per trajectory is 6.6 KB; a second copy of k_solve's 8 KB Jacobian / gain array and a 12 KB expansion
// would leave a CU three or four wavefronts.  So nothing of a horizon's length lives in the shared area any more:
//   * BOTH expansions (and Jacobians) go to their 256-byte rows in global memory (grp_expand<STREAM>) and the sweep streams
//     both halves through a ring each (backward_sweep_pair<BOTH>); a trajectory that sweeps alone runs the same code on both halves;
//   * the gains stay in global memory and reach the rollout pass through a ring of two chunks of CILQR_GRPL_CHUNK steps
//     (rollout_group_long), refilled by all 64 lanes one chunk ahead;
//   * trial costs run one at a time (one stage-cost slot, two rows per lane).
// Shared area = max(stage-cost scratch + lane window | the sweep's constants + two rings | the gains ring): 20 KB a wavefront
// at N = 100 with a window of 304 samples — eight wavefronts = sixteen trajectories per CU (k_solve's build: eight).
#define CILQR_GRPL_CHUNK 8
#define CILQR_GRPL_CHUNK_BYTES (CILQR_GRPL_CHUNK * CILQR_KD * 8)
__host__ __device__ inline int grpl_cs_doubles(int N) { return (3 * (N + 1) + 1) & ~1; }
__host__ __device__ inline int grpl_gring_doubles(int G) { return 2 * G * CILQR_GRPL_CHUNK * CILQR_KD; }
__host__ __device__ inline int grpl_shared_doubles(int N, int W, int G) {
    int s = grpl_cs_doubles(N) + 2 * W;
    const int sweep = CILQR_XCH + 2 * CILQR_GL_RING, roll = grpl_gring_doubles(G);
    s = s > sweep ? s : sweep;
    return s > roll ? s : roll;
}
__host__ __device__ inline size_t grpl_lds_bytes(int N, int W, int G) {
    return sizeof(double) * ((size_t)G * grp_pg_doubles(N) + (size_t)grpl_shared_doubles(N, W, G));
}

typedef double __attribute__((ext_vector_type(2))) f64x2;
typedef const f64x2 __attribute__((address_space(3))) lds_cf64x2;
typedef const f64x2 __attribute__((address_space(1))) f64x2g;

// Augmented Lagrangian in pairs (long layout only): the trajectory's penalty weight rho (hpp:106-112) lives in GrpSt::J_pair — the
// long layout costs one trial per pass, the slot is free —, its multipliers stay in HBM ([N][C] per trajectory, BatchArgs::alm_mu)
struct GrpAlm {
    const double* mu;   // this trajectory's multipliers
    double* mu_next;    // ... and the proposal cost_and_model_derivatives writes
    int C;
};
__device__ inline AlmSt grp_almst(const GrpAlm& ga, double rho) {
    AlmSt al;
    al.mu = (double*)uniform_ptr(const_cast<double*>(ga.mu));
    al.mu_next = (double*)uniform_ptr(ga.mu_next);
    al.rho = rho;
    al.C = uniform_int(ga.C);
    return al;
}
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

// An idle wavefront: leave a ticket, take a place in the queue (rq_claim: the next push that has no taker is this wavefront's)
// and wait for it.  claim >= 0: a place one of its slots holds already — no new ticket.  Returns the trajectory, or -1 when the
// launch is over (every trajectory finished), too many are waiting already, or the bound on the wait is reached (flagged:
// the push that comes for an abandoned place would be lost).
__device__ __attribute__((noinline)) int grp_wait_for_work(unsigned* ctl, const unsigned long long* q, unsigned cap, unsigned B, int lane,
                                                           long long claim = -1) {
    if (sh_ld_u(ctl + SH_FINISHED, lane) >= B) return -1;
    unsigned h;
    if (claim < 0) {
        if (sh_ld_u(ctl + SH_HELPING, lane) >= (unsigned)CILQR_GRP_MAX_WAITING) return -1; // (a look first)
        if (sh_add_u(ctl + SH_HELPING, 1u, lane) >= (unsigned)CILQR_GRP_MAX_WAITING) {
            (void)sh_add_u(ctl + SH_HELPING, 0u - 1u, lane);
            return -1;
        }
        (void)sh_add_u(ctl + SH_HELPERS, 1u, lane);
        h = rq_claim(ctl, lane);
    } else {
        // (a wavefront that holds a place never leaves before the launch is over: the push that comes for it would be lost)
        h = (unsigned)claim;
        (void)sh_add_u(ctl + SH_HELPING, 1u, lane);
        (void)sh_add_u(ctl + SH_HELPERS, 1u, lane);
    }
#ifdef CILQR_DEV_BUILD
    // (the test of what a lost hand-over looks like from outside forces the expiry: tests/test_gpu_parity.py::test_a_lost_hand_over_is_loud)
    const unsigned forced = sh_ld_u(ctl + SH_TEST_SPINS, lane);
    const int bound = forced ? (int)forced : (1 << 20);
#else
    constexpr int bound = 1 << 20;
#endif
    for (int spin = 0; spin < bound; ++spin) { // (a launch lasts milliseconds; this bound is seconds)
        const int pb = rq_poll(q, cap, h, lane);
        if (pb >= 0) return pb;
        // (every wavefront polls its OWN place; the one word they all share is looked at every eighth time: the wavefronts
        //  that hold a place are not capped at CILQR_GRP_MAX_WAITING, and a line serves ~90 atomics per microsecond)
        if ((spin & 7) == 7 && sh_ld_u(ctl + SH_FINISHED, lane) >= B) return -1;
        __builtin_amdgcn_s_sleep(127);
    }
    if (lane == 0) sh_st(ctl + SH_ERROR, 1u);
    return -1;
}
// Sliced solves (round 5; k_solve's resumable solves are the model, cilqr_device.hpp): a solve runs `res_iters` iterations at a
// time.  At the end of a slice the trajectory is parked and queued — and the slot takes the next one — when somebody else is
// in need of the slot: parked trajectories wait, or the LAST fresh trajectories are about to be handed out (fewer than
// `window` left: the launch's final round — before that a fresh trajectory finds a slot soon enough anyway and the hand-over,
// 2.9 KB out and in again plus the set-up of a segment, would be paid by every long solve of a large batch for nothing).
// Once the counter is dry every slot that falls empty takes a parked trajectory, so the long solves of the final round
// advance side by side, a slice at a time, instead of finishing one by one on an emptying chip.
// The queue of the grouped build is NOT reused within a launch (CILQR_GRP_Q_PER_TRAJECTORY entries per trajectory, position =
// push number): a place that was claimed is read whenever its owner next looks, and a ring that wrapped could have been
// overwritten by then (found with slices of ONE iteration: 100 pushes per trajectory, the push cap positions later landed
// before a slot's once-a-turn look; the trajectory was lost and the launch flagged).  Hand-overs of either kind simply stop
// when the room is used up (grp_queue_room; two batches' worth of margin for pushes that have looked already): solves then
// run on where they are.
#define CILQR_GRP_Q_PER_TRAJECTORY 16
__device__ inline bool grp_queue_room(const unsigned* ctl, unsigned B, unsigned cap, int lane) {
    return sh_ld_u(ctl + SH_Q_RESV, lane) + 2u * B < cap;
}
__device__ __attribute__((noinline)) bool grp_slot_wanted(const unsigned* next, unsigned B, unsigned window, int fresh_left,
                                                          const unsigned* ctl, const unsigned long long* q, unsigned cap, int lane) {
    if (!grp_queue_room(ctl, B, cap, lane)) return false;
    if (fresh_left) {
        const unsigned nx = sh_ld_u(next, lane);
        if (nx < B && B - nx <= window) return true;
    }
    return rq_avail(ctl, lane);
}
// A slot in need of work: a parked trajectory (>= 0); -1: none is queued; -2: the place taken (*claim) has no entry yet — it was
// claimed by a faster slot in between, or its push is still on the way: the slot keeps the place (GP_CLAIMED) and looks again
// every turn.
__device__ __attribute__((noinline)) int grp_take_parked(unsigned* ctl, const unsigned long long* q, unsigned cap, int lane, int* claim) {
    if (!rq_avail(ctl, lane)) return -1;
    const unsigned h = rq_claim(ctl, lane);
    for (int t = 0; t < 4; ++t) {
        const int pb = rq_poll(q, cap, h, lane);
        if (pb >= 0) return pb;
        __builtin_amdgcn_s_sleep(8);
    }
    *claim = (int)h;
    return -2;
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
    l.ring = s; s += CILQR_GL_RING;
    l.ctld = nullptr;
    l.ctli = nullptr;
    l.prof = nullptr;
    l.w0 = 0;
    l.W = 0;
}

__device__ inline void carve_group_long(Lds& l, double* base, int N, int G, int g) {
    double* p = base + (size_t)g * grp_pg_doubles(N);
    l.x = p; p += 4 * (N + 1);
    l.u = p; p += 2 * N;
    l.ridx = reinterpret_cast<int*>(p);
    l.tidx = l.ridx + (N + 2);
    p += grp_idx_doubles(N);
    l.ck = reinterpret_cast<CstK*>(p);
    double* s = base + (size_t)G * grp_pg_doubles(N);
    l.kd = s; // (only as the stage-cost scratch: Jacobians and gains live in global memory)
    l.cs = s;
    l.lxs = 7;
    l.lx = l.lu = l.lxx = l.luu = nullptr;
    l.xch = s;                // during a sweep
    l.ring = s + CILQR_XCH;   // ... two rings, one per half of the wavefront
    l.win = s + grpl_cs_doubles(N);
    l.gl = nullptr;
    l.ctld = nullptr;
    l.ctli = nullptr;
    l.prof = nullptr;
    l.w0 = 0;
    l.W = 0;
}
template <bool LONG>
__device__ inline void carve_group_t(Lds& l, double* base, int N, int G, int g) {
    if (LONG) carve_group_long(l, base, N, G, g);
    else carve_group(l, base, N, G, g);
}

// ---------------------------------------------------------------------------------------------
// Round 5: the backward sweeps (cs:383-440) of BOTH trajectories of a wavefront in ONE instruction stream.
//
// backward_sweep_lanes lays the step's 34 useful elements out on a 6 x 8 grid, 48 of 64 lanes, and the grouped kernel ran
// it once per trajectory: two sweeps of ~144 instructions a step on a vector unit that two wavefronts keep busy (the sweep
// is the phase that is bound by issue slots: round 4's dual-chain probe, two sweeps interleaved in one loop, took 1.95-2.11 x
// one).  Here a trajectory gets HALF the wavefront — 32 lanes, a 4 x 8 grid — and both halves execute the same instructions:
//   (r, c), c < 4   pass 1: X[r][c]      pass 2: Y[r][c] -> Q_xx[r][c]          owns W[r][c] = V_xx[r][c]
//   (r, 4)          pass 1: X[r][4] -> Q_x[r] = l_x[r] + X[r][4]               owns W[r][4] = V_x[r]
//   (k, 5)          pass 1: X[4][k]      pass 2: Y[4][k] -> Q_ux[0][k]          (0, 5): delta_V[0] += (0.5 d)^T Q_uu d
//   (k, 6)          pass 1: X[5][k]      pass 2: Y[5][k] -> Q_ux[1][k]          (1, 5): delta_V[1] += d^T Q_u
//   (0, 7) (1, 7)   pass 1: X[4][4], X[5][4] -> Q_u[0], Q_u[1]
//   (r, 7)          pass 2: Y[4 + r / 2][4 + r % 2] -> the four entries of Q_uu (+ lambda on its diagonal)
// — the rows 4 and 5 of the big grid live in the columns 5 .. 7 that the small one has to spare.  Every element is the
// expression backward_sweep_lanes evaluates, four (or two) products summed in index order, zeros included: same bits.  What
// changes is how operands travel: every cross-lane move is a ds_bpermute with a per-lane source (32 per step; the 6 x 8 form
// mixes 16 of them with 10 DPP moves and 8 v_readlane), every coefficient a per-lane LDS address that walks backwards with
// the step — so the two halves may read from different places:
//   half 0: trajectory A, its Jacobians and expansion in the wavefront's shared LDS arrays (as for a lone sweep);
//   half 1: trajectory B, whose expansion phase wrote 256-byte ROWS to global memory (CILQR_GRP_ROW: the 16 slots of
//           CILQR_GL_ROW, then the step's eight Jacobian entries) because LDS holds one set — streamed through a ring of four
//           rows in LDS, 64 doubles (two rows) per refill, fetched two steps before the sweep gets there.
// A sweep that meets a non-PD Q_uu (cs:415-420) is only MARKED and runs on (its gains are never used: the iteration ends as
// BACKWARD_PASS_FAIL); returns bit 0 / bit 1 = trajectory A / B completed.  gr: buffer over the block's scratch, goff: byte
// offsets of the two gain arrays in it.
struct PairArgs {
    double lambA, lambB, dtA, dtB;
    const double* rowsB; // [(N + 1)][CILQR_GRP_ROW] in global memory
    double* ringB;       // [CILQR_GL_RING] in LDS
    unsigned goffA, goffB;
    // BOTH (the long layout): trajectory A streams too — byte offsets of the two trajectories' rows in the block's scratch
    // area (read through the one descriptor over it) and A's ring
    unsigned roffA, roffB;
    double* ringA;
};
// ALM (with BOTH): the rows are the augmented-Lagrangian ones — dense, NOT symmetric l_xx (cs:701-713) at CILQR_GLA_LXX + 4 r + c,
// l_uu / l_x / l_u at their CILQR_GLA_* slots, the Jacobians at CILQR_GLA_JAC
template <bool BOTH = false, bool ALM = false>
__device__ inline unsigned backward_sweep_pair(int N, const Lds& lA, const PairArgs& pa, double* scr_blk, unsigned scr_bytes, int lane,
                                               double dVA[2], double dVB[2]) {
    static_assert(!ALM || BOTH, "augmented-Lagrangian rows: both halves stream");
    constexpr int JAC = ALM ? CILQR_GLA_JAC : CILQR_GRP_ROW_JAC;
    const int h = lane >> 5, q = lane & 31, r = q >> 3, cg = q & 7, hb = lane & 32;
    // this lane's two elements
    int R1, C1, R2, C2;
    if (cg <= 4) { R1 = r; C1 = cg; R2 = r; C2 = (cg < 4) ? cg : 0; }
    else if (cg == 5) { R1 = 4; C1 = r; R2 = 4; C2 = r; }
    else if (cg == 6) { R1 = 5; C1 = r; R2 = 5; C2 = r; }
    else { R1 = 4 + (r & 1); C1 = 4; R2 = 4 + (r >> 1); C2 = 4 + (r & 1); }
    const bool diag = (R2 >= 4) && (C2 == R2);
    const bool isvec = (cg == 4) || (cg == 7);
    if (lane == 0) {
        lA.xch[CILQR_XCH_CONST + 0] = 0.0;
        lA.xch[CILQR_XCH_CONST + 1] = 1.0;
        lA.xch[CILQR_XCH_CONST + 2] = pa.dtA;
        lA.xch[CILQR_XCH_CONST + 3] = pa.dtB;
    }
    wave_sync();
    const double* const base = lA.x;
    const int CCo = (int)(lA.xch - lA.x) + CILQR_XCH_CONST; // ZERO; + 1 ONE; + 2 dt of A; + 3 dt of B
    const int A5 = (int)(lA.kd - lA.x);
    // ten coefficient addresses (LDS bytes) at step N - 1, their per-step decrements, and what a ring address gets back
    // every fourth step; a[0..3] = M[k][R1], a[4..7] = M[k][C2], a[8] = L[R2][C2], a[9] = l[R1]
    unsigned a[10], d[10], w[10];
    const unsigned ring0 = lds_addr((BOTH && h == 0) ? pa.ringA : pa.ringB);
    constexpr unsigned ROWB = CILQR_GRP_ROW * 8u;
    auto place = [&](int j, int off, int stride, int slotB) {
        // off / stride: doubles relative to lA.x (half 0); slotB: slot in trajectory B's row, -1 = a constant (off applies)
        if ((!BOTH && h == 0) || slotB < 0) {
            int o = off;
            if (h == 1 && off == CCo + 2) o = CCo + 3; // (B's own dt)
            const int st = (h == 0) ? stride : 0;
            a[j] = lds_addr(base + o + st * (N - 1));
            d[j] = 8u * (unsigned)st;
            w[j] = 0u;
        } else {
            a[j] = ring0 + 8u * (unsigned)slotB + (unsigned)((N - 1) & 3) * ROWB;
            d[j] = ROWB;
            w[j] = 4u * ROWB;
        }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int off, stride;
        lane_map_M(lA, k, R1, off, stride);
        place(k, off, stride, stride ? JAC + (off - A5) : -1);
        lane_map_M(lA, k, C2, off, stride);
        place(4 + k, off, stride, stride ? JAC + (off - A5) : -1);
    }
    int slotq = -1, slotv;
    {
        // L[R2][C2]: l_xx (7 packed entries 00 01 03 11 13 33 22), l_uu diagonal, zero elsewhere
        int off = CCo, stride = 0;
        if (ALM) {
            if (R2 < 4 && C2 < 4) { stride = 16; slotq = CILQR_GLA_LXX + 4 * R2 + C2; }
            else if (diag) { stride = 2; slotq = CILQR_GLA_LUU + (R2 - 4); }
        } else if (R2 < 4 && C2 < 4) {
            const int lo = (R2 < C2) ? R2 : C2, hi = (R2 < C2) ? C2 : R2;
            int e = -1;
            if (lo == 0 && hi == 0) e = 0;
            if (lo == 0 && hi == 1) e = 1;
            if (lo == 0 && hi == 3) e = 2;
            if (lo == 1 && hi == 1) e = 3;
            if (lo == 1 && hi == 3) e = 4;
            if (lo == 3 && hi == 3) e = 5;
            if (lo == 2 && hi == 2) e = 6;
            if (e >= 0) { off = BOTH ? 0 : (int)(lA.lxx - lA.x) + e; stride = 7; slotq = (e < 6) ? CILQR_GL_LXX + e : CILQR_GL_LXX22; }
        } else if (diag) {
            off = BOTH ? 0 : (int)(lA.luu - lA.x) + (R2 - 4); stride = 2; slotq = CILQR_GL_LUU + (R2 - 4);
        }
        place(8, off, stride, slotq);
        if (R1 < 4) { off = BOTH ? 0 : (int)(lA.lx - lA.x) + R1; stride = 4; slotv = (ALM ? CILQR_GLA_LX : CILQR_GL_LX) + R1; }
        else { off = BOTH ? 0 : (int)(lA.lu - lA.x) + (R1 - 4); stride = 2; slotv = (ALM ? CILQR_GLA_LU : CILQR_GL_LU) + (R1 - 4); }
        place(9, off, stride, slotv);
    }
    // cross-lane sources (ds_bpermute byte addresses)
    int sw[4], sx[4], sq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sw[k] = (hb + 8 * k + C1) << 2;
        sx[k] = ((R2 < 4) ? hb + 8 * R2 + k : hb + 8 * k + (R2 == 4 ? 5 : 6)) << 2;
        sq[k] = (hb + 8 * k + 7) << 2;
    }
    int sc0, sc1, sr0, sr1;
    if (cg <= 4) {
        sc0 = (cg < 4) ? hb + 8 * cg + 5 : hb + 7;
        sc1 = (cg < 4) ? hb + 8 * cg + 6 : hb + 15;
        sr0 = hb + 8 * r + 5;
        sr1 = hb + 8 * r + 6;
    } else { // (only (0, 5) and (1, 5) matter: r = c = Q_u, the expected cost reduction)
        sc0 = hb + 7; sc1 = hb + 15; sr0 = hb + 7; sr1 = hb + 15;
    }
    sc0 <<= 2; sc1 <<= 2; sr0 <<= 2; sr1 <<= 2;
    const double krf = (q == 5) ? 0.5 : 1.0;
    const double lamb = h ? pa.lambB : pa.lambA;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(scr_blk), 0, (int)scr_bytes, 0x00020000);
    const unsigned goff = (h ? pa.goffB : pa.goffA) + 8u * (unsigned)q;
    // trajectory B's rows: a ring of two chunks (two rows each)
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(const_cast<double*>(pa.rowsB)), 0,
                                                                          (N + 1) * (int)ROWB, 0x00020000);
    constexpr int CHB = 2 * (int)ROWB; // bytes per chunk
    double chunk = 0.0, chunkA = 0.0;
    const unsigned laneA = pa.roffA + 8u * (unsigned)lane, laneB = pa.roffB + 8u * (unsigned)lane; // (BOTH: through gr)
    {
        const int c0 = (N - 1) / 2;
        if (BOTH) {
            const double fa = gl_load(gr, laneA, c0 * CHB), fb = gl_load(gr, laneB, c0 * CHB);
            pa.ringA[(c0 & 1) * CILQR_WAVE + lane] = fa;
            pa.ringB[(c0 & 1) * CILQR_WAVE + lane] = fb;
            if (c0 > 0) { chunkA = gl_load(gr, laneA, (c0 - 1) * CHB); chunk = gl_load(gr, laneB, (c0 - 1) * CHB); }
        } else {
            const double first = gl_load(grs, 8u * (unsigned)lane, c0 * CHB);
            pa.ringB[(c0 & 1) * CILQR_WAVE + lane] = first;
            if (c0 > 0) chunk = gl_load(grs, 8u * (unsigned)lane, (c0 - 1) * CHB);
        }
    }
    // W = [l_xx[N] | l_x[N]] on the lanes (r, c <= 4)
    double wn = 0.0;
    if (cg <= 4) {
        if (BOTH) {
            const int slot = (cg < 4) ? slotq : slotv;
            wn = (slot >= 0) ? gl_load(gr, (h ? pa.roffB : pa.roffA) + 8u * (unsigned)slot, N * (int)ROWB) : 0.0;
        } else if (h == 0) {
            // (the addresses above are at step N - 1: one stride further is row N)
            wn = (cg < 4) ? lds_load(a[8] + d[8]) : lds_load(a[9] + d[9]);
        } else {
            const int slot = (cg < 4) ? slotq : slotv;
            wn = (slot >= 0) ? gl_load(grs, 8u * (unsigned)slot, N * (int)ROWB) : 0.0;
        }
    }
    wave_sync(); // (the ring's first chunk is in place)
    double dvacc = 0.0;
    unsigned failed = 0u;
    for (int i = N - 1; i >= 0; --i) {
        if ((i & 1) == 1 && i != N - 1) {
            // the sweep enters chunk c of B's rows: they arrived while chunk c + 1 was computed; fetch chunk c - 1
            const int cch = i >> 1;
            pa.ringB[(cch & 1) * CILQR_WAVE + lane] = chunk;
            if (BOTH) {
                pa.ringA[(cch & 1) * CILQR_WAVE + lane] = chunkA;
                if (cch > 0) { chunkA = gl_load(gr, laneA, (cch - 1) * CHB); chunk = gl_load(gr, laneB, (cch - 1) * CHB); }
            } else if (cch > 0) chunk = gl_load(grs, 8u * (unsigned)lane, (cch - 1) * CHB);
        }
        double m1[4], m2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m1[k] = lds_load(a[k]);
            m2[k] = lds_load(a[4 + k]);
        }
        const double Lq = lds_load(a[8]), lv = lds_load(a[9]);
#pragma unroll
        for (int j = 0; j < 10; ++j) a[j] -= d[j];
        if ((i & 3) == 0) {
#pragma unroll
            for (int j = 0; j < 10; ++j) a[j] += w[j];
        }
        auto gather = [](double v, int src4) {
            const unsigned long long u = dm_to_bits(v);
            int lo = (int)(unsigned)(u & 0xffffffffULL), hi = (int)(unsigned)(u >> 32);
            lo = __builtin_amdgcn_ds_bpermute(src4, lo);
            hi = __builtin_amdgcn_ds_bpermute(src4, hi);
            return dm_from_bits(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
        };
        // pass 1
        const double w0 = gather(wn, sw[0]), w1 = gather(wn, sw[1]), w2 = gather(wn, sw[2]), w3 = gather(wn, sw[3]);
        const double X = CQ_MADD(m1[3], w3, CQ_MADD(m1[2], w2, CQ_MADD(m1[1], w1, m1[0] * w0)));
        const double Zv = lv + X;
        // pass 2
        const double x0 = gather(X, sx[0]), x1 = gather(X, sx[1]), x2 = gather(X, sx[2]), x3 = gather(X, sx[3]);
        const double Y = CQ_MADD(x3, m2[3], CQ_MADD(x2, m2[2], CQ_MADD(x1, m2[1], x0 * m2[0])));
        double Q = Lq + Y;
        if (diag) Q = Q + lamb;
        const double Quu0 = gather(Q, sq[0]), Quu1 = gather(Q, sq[1]), Quu2 = gather(Q, sq[2]), Quu3 = gather(Q, sq[3]);
        const double S = isvec ? Zv : Q;
        const double c0 = gather(S, sc0), c1 = gather(S, sc1), r0 = gather(S, sr0), r1 = gather(S, sr1);
        const double det = Quu0 * Quu3 - Quu2 * Quu1;
        const double invdet = 1.0 / det;
        {
            // Eigen::LLT's verdict (see backward_sweep_lanes): decided without the square root on ordinary magnitudes
            const unsigned h0 = (unsigned)(dm_to_bits(Quu0) >> 32), h3 = (unsigned)(dm_to_bits(Quu3) >> 32);
            const bool ordinary = ((h0 - 0x2B300000u) < 0x29800000u) && ((h3 - 0x2B300000u) < 0x29800000u);
            const bool surely_pd = ordinary && (Quu0 * Quu3 > (Quu2 * Quu2) * 1.0000000000009095);
            unsigned long long need = __ballot(!surely_pd);
            if (failed & 1u) need &= 0xffffffff00000000ULL;
            if (failed & 2u) need &= 0x00000000ffffffffULL;
            if (need != 0ULL) {
                bool fail = false;
                if (!surely_pd) {
                    if (Quu0 <= 0.0) {
                        fail = true;
                    } else {
                        const double l00 = dm_sqrt(Quu0);
                        const double l10 = Quu2 / l00;
                        const double piv1 = Quu3 - l10 * l10;
                        if (piv1 <= 0.0) fail = true;
                    }
                }
                const unsigned long long fb = __ballot(fail);
                if ((fb & 0x00000000ffffffffULL) != 0ULL) failed |= 1u;
                if ((fb & 0xffffffff00000000ULL) != 0ULL) failed |= 2u;
                if (failed == 3u) break;
            }
        }
        const double n00 = -(Quu3 * invdet), n01 = -(-Quu1 * invdet), n10 = -(-Quu2 * invdet), n11 = -(Quu0 * invdet);
        const double kc0 = CQ_MADD(n01, c1, n00 * c0), kc1 = CQ_MADD(n11, c1, n10 * c0); // (K | d)[:, c]
        double kr0 = CQ_MADD(n01, r1, n00 * r0), kr1 = CQ_MADD(n11, r1, n10 * r0);       // K[:, r]
        kr0 = kr0 * krf; // (exact: the factor is 1, or 0.5 on lane (0, 5) — hd = 0.5 d of cs:435)
        kr1 = kr1 * krf;
        const double p0 = CQ_MADD(kr1, Quu2, kr0 * Quu0);
        const double p1 = CQ_MADD(kr1, Quu3, kr0 * Quu1);
        const double ta = CQ_MADD(p1, kc1, p0 * kc0);
        const double tb = CQ_MADD(kr1, c1, kr0 * c0);
        const double tc = CQ_MADD(r1, kc1, r0 * kc0);
        int cgv = cg;
        __asm__("" : "+v"(cgv));
        const double own = (cgv < 4) ? Q : Zv;
        wn = ((own + ta) + tb) + tc;
        int qv = q;
        __asm__("" : "+v"(qv));
        if (qv < 5) { // row 0 of a half holds (K | d) column c
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, kc0), gr, goff, i * (CILQR_KD * 8), 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, kc1), gr, goff + 8u * CILQR_KD_ROW, i * (CILQR_KD * 8), 0);
        }
        dvacc = dvacc + ((qv == 5) ? ta : tb);
    }
    dVA[0] = lane_bcast<5>(dvacc);
    dVA[1] = lane_bcast<13>(dvacc);
    dVB[0] = lane_bcast<37>(dvacc);
    dVB[1] = lane_bcast<45>(dvacc);
    wave_sync();
    return 3u & ~failed;
}

// ---------------------------------------------------------------------------------------------
// The head of an iteration in two steps (round 5), so that the sweeps of the wavefront's two trajectories can run as one:
// grp_expand — cost expansion + model Jacobians of the trajectory in slot g (cs:463-690, ut:285-342), into the wavefront's
// shared LDS arrays (STREAM = false: the first trajectory of a turn that gets here) or into its rows in global memory (STREAM =
// true: the second one; LDS holds one set) — and grp_sweep, after every trajectory of the turn has had its segment: the
// backward sweep (cs:383-440) of the one trajectory that is waiting, or of both in one instruction stream.  Both read
// everything from LDS / their arguments and leave their results in GrpSt (dV, and for a completed sweep the request for
// the search's first rollout pass) and global memory (gains): the calls carry nothing.
template <int NC, int G, bool STREAM, bool LONG = false, bool ALM = false>
__device__ __attribute__((noinline)) void grp_expand(double* lds, int g, int n_rt, int lane, double* rows_rt, long long* prof,
                                                     GrpAlm ga = GrpAlm{nullptr, nullptr, 0}) {
    static_assert(STREAM || !LONG, "long layout: every expansion goes to its rows in global memory");
    static_assert(!ALM || LONG, "augmented Lagrangian in pairs: the long layout");
    const int N = NC ? NC : uniform_int(n_rt);
    Lds l;
    carve_group_t<LONG>(l, lds, N, G, g);
    Cst c;
    load_cst_lds(c, grp_cst(lds, N, g));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    if (ALM) al = grp_almst(ga, grp_state(lds, N, g)->J_pair);
    const long long t0 = (CILQR_GPROF && prof) ? (long long)__builtin_readcyclecounter() : 0;
    l.W = 0; // the lane window gives way to the expansion (the rows' one lane lookup each: global memory)
    if (STREAM && ALM) {
        double* const rows = (double*)uniform_ptr(rows_rt);
        l.gl = rows;
        l.lxs = 16;
        l.kd = rows + CILQR_GLA_JAC;
        cost_and_model_derivatives<true, true, CILQR_GL_ROW, true>(c, l, al, lane);
    } else if (STREAM) {
        double* const rows = (double*)uniform_ptr(rows_rt);
        l.gl = rows;
        l.kd = rows + CILQR_GRP_ROW_JAC; // (the Jacobians of step k at slot 16 of row k)
        cost_and_model_derivatives<false, true, CILQR_GRP_ROW>(c, l, al, lane);
    } else {
        cost_and_model_derivatives<false, false>(c, l, al, lane);
    }
    if (CILQR_GPROF && prof) {
        const long long t1 = (long long)__builtin_readcyclecounter();
        if (lane == 0) { prof[1] += t1 - t0; prof[6] += t1 - t0; }
    }
}

// what a completed sweep leaves behind for the trajectory: the head of its line search (cs:349-354)
__device__ inline void grp_after_sweep(GrpSt* st, bool ok, double dV0, double dV1, int tier) {
    st->dV0 = dV0;
    st->dV1 = dV1;
    if (ok) {
        const int all = (tier == 0) || (tier < 0 && st->deep_next != 0);
        st->status = CILQR_RUNNING;
        st->new_J = st->J_cur;
        st->trials = 0;
        st->flag = 0;
        st->have_all = all;
        st->t0 = 0;
        st->req = all ? 2 : 1;
        st->phase = GP_SEARCH;
    } else {
        st->phase = GP_BPF;
    }
}

// gA: the trajectory whose expansion is in LDS; gB: the one whose rows are in global memory, or -1.  Returns the number of
// trajectories that now wait for a rollout pass.
template <int NC, int G, bool LONG = false, bool ALM = false>
__device__ __attribute__((noinline)) int grp_sweep(double* lds, int gA_rt, int gB_rt, int n_rt, int lane, double* scr_blk_rt, int tier_rt,
                                                   long long* profA, long long* profB) {
    const int N = NC ? NC : uniform_int(n_rt);
    const int gA = uniform_int(gA_rt), gB = uniform_int(gB_rt), tier = uniform_int(tier_rt);
    double* const scr_blk = (double*)uniform_ptr(scr_blk_rt);
    Lds l;
    carve_group_t<LONG>(l, lds, N, G, gA);
    GrpSt* const stA = grp_state(lds, N, gA);
    const long long t0 = (CILQR_GPROF && profA) ? (long long)__builtin_readcyclecounter() : 0;
    const size_t slot_d = grp_scratch_doubles(N);
    const size_t gains_d = slab_doubles(N) + first_trial_doubles(N);
    int asked = 0;
    if (LONG) {
        // both halves stream from the rows in global memory; a trajectory on its own occupies both halves (the same numbers
        // twice: its gains are stored by two lanes each, identically)
        const int gb = (gB < 0) ? gA : gB;
        GrpSt* const stB = grp_state(lds, N, gb);
        PairArgs pa;
        pa.lambA = stA->lamb; pa.lambB = stB->lamb;
        pa.dtA = stA->dt; pa.dtB = stB->dt;
        pa.rowsB = scr_blk + (size_t)gb * slot_d + grp_rows_offset(N);
        pa.ringA = l.ring;
        pa.ringB = l.ring + CILQR_GL_RING;
        pa.goffA = (unsigned)(((size_t)gA * slot_d + gains_d) * sizeof(double));
        pa.goffB = (unsigned)(((size_t)gb * slot_d + gains_d) * sizeof(double));
        pa.roffA = (unsigned)(((size_t)gA * slot_d + grp_rows_offset(N)) * sizeof(double));
        pa.roffB = (unsigned)(((size_t)gb * slot_d + grp_rows_offset(N)) * sizeof(double));
        double dVA[2], dVB[2];
        const unsigned ok = backward_sweep_pair<true, ALM>(N, l, pa, scr_blk, (unsigned)(G * slot_d * sizeof(double)), lane, dVA, dVB);
        if (lane == 0) {
            grp_after_sweep(stA, (ok & 1u) != 0u, dVA[0], dVA[1], tier);
            if (gB >= 0) grp_after_sweep(stB, (ok & 2u) != 0u, dVB[0], dVB[1], tier);
        }
        asked = (int)(ok & 1u) + ((gB >= 0) ? (int)((ok >> 1) & 1u) : 0);
    } else if (gB < 0) {
        Cst c;
        load_cst_lds(c, grp_cst(lds, N, gA));
        double dV[2];
        const bool ok = backward_sweep_lanes<0, true, CILQR_GRP_QLDS>(c, l, stA->lamb, lane, dV, nullptr, scr_blk + (size_t)gA * slot_d + gains_d);
        if (lane == 0) grp_after_sweep(stA, ok, dV[0], dV[1], tier);
        asked = ok ? 1 : 0;
    } else {
        GrpSt* const stB = grp_state(lds, N, gB);
        PairArgs pa;
        pa.lambA = stA->lamb; pa.lambB = stB->lamb;
        pa.dtA = stA->dt; pa.dtB = stB->dt;
        pa.rowsB = scr_blk + (size_t)gB * slot_d + grp_rows_offset(N);
        pa.ringB = l.ring;
        pa.goffA = (unsigned)(((size_t)gA * slot_d + gains_d) * sizeof(double));
        pa.goffB = (unsigned)(((size_t)gB * slot_d + gains_d) * sizeof(double));
        double dVA[2], dVB[2];
        const unsigned ok = backward_sweep_pair(N, l, pa, scr_blk, (unsigned)(G * slot_d * sizeof(double)), lane, dVA, dVB);
        if (lane == 0) {
            grp_after_sweep(stA, (ok & 1u) != 0u, dVA[0], dVA[1], tier);
            grp_after_sweep(stB, (ok & 2u) != 0u, dVB[0], dVB[1], tier);
        }
        asked = (int)(ok & 1u) + (int)((ok >> 1) & 1u);
    }
    wave_sync();
    if (CILQR_GPROF && profA) {
        const long long t1 = (long long)__builtin_readcyclecounter();
        const long long dt = (t1 - t0) / ((gB >= 0) ? 2 : 1); // (a sweep of both: shared out)
        if (lane == 0) {
            profA[2] += dt; profA[6] += dt;
            if (gB >= 0 && profB) { profB[2] += dt; profB[6] += dt; }
        }
    }
    return asked;
}

// (Round 4's grp_expand_backward — expansion + sweep of one trajectory in one call — and its development probe
//  backward_sweep_lanes_multi, two sweeps interleaved in one loop on separate register sets: 1.95-2.11 x the time of one —
//  are gone: grp_expand + grp_sweep do the same for one trajectory, and the sweep of two is backward_sweep_pair.)

// get_total_cost (cs:199-287) of trial t of the trajectory in slot g — out of line for the same reason: the trial lives in
// global memory (src / as: the slab or the first-trial buffer), x's lane window is staged (w0, W), the result is the return
// value; serial reference-point chains that had to be run are counted in GrpSt::nfb.
template <int NC, int G, int NCH = 1, bool LONG = (NCH > 1), bool ALM = false>
__device__ __attribute__((noinline)) double grp_cost_trial(double* lds, int g, int n_rt, int lane, const double* src, int t, int as,
                                                            int w0, int W, GrpAlm ga = GrpAlm{nullptr, nullptr, 0}) {
    static_assert(!ALM || LONG, "augmented Lagrangian in pairs: the long layout");
    const int N = NC ? NC : uniform_int(n_rt); // (an argument: in a vector register — scalar again, or descriptors built from it count as divergent)
    Lds l;
    carve_group_t<LONG>(l, lds, N, G, g);
    l.w0 = w0;
    l.W = W;
    Cst c;
    load_cst_lds(c, grp_cst(lds, N, g));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    if (ALM) al = grp_almst(ga, grp_state(lds, N, g)->J_pair);
    int nfb = 0;
    double J1[1];
    total_cost_trials<false, NCH, ALM, 1>(c, l, al, src, t, 1, lane, w0, 0, &nfb, J1, nullptr, 0, as);
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
template <int NC, int G, bool LONG = false, bool ALM = false>
__device__ __attribute__((noinline)) double grp_init(double* lds, int g, int n_rt, int lane, double xs0, double xs1, double xs2,
                                                      double xs3, const double* last_u, int Wcap, GrpAlm ga = GrpAlm{nullptr, nullptr, 0}) {
    const int N = NC ? NC : uniform_int(n_rt); // (an argument: in a vector register — scalar again, or descriptors built from it count as divergent)
    Lds l;
    carve_group_t<LONG>(l, lds, N, G, g);
    Cst c;
    load_cst_lds(c, grp_cst(lds, N, g));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    const double xs[4] = {xs0, xs1, xs2, xs3};
    int idx0 = 0;
    init_trajectory(c, l, xs, last_u, lane, idx0, Wcap);
    seed_trial_indices(l, N, 2, lane);
    if (ALM) al = grp_almst(ga, grp_state(lds, N, g)->J_pair);
    const double J = total_cost_lds<ALM>(c, l, al, lane);
    if (lane == 0) grp_state(lds, N, g)->idx0 = idx0;
    wave_sync();
    return J;
}

// get_total_cost of the CURRENT trajectory with the current multipliers (cs:342 at the head of every iteration, cs:104 at the
// end of the solve: under the augmented Lagrangian the multipliers may have moved since the cost was last taken).  The lane
// window (w0, W) must be staged: the caller is on the search's side of the turn, or restages it.
template <int NC, int G>
__device__ __attribute__((noinline)) double grp_recost_alm(double* lds, int g, int n_rt, int lane, int w0, int W, GrpAlm ga) {
    const int N = NC ? NC : uniform_int(n_rt);
    Lds l;
    carve_group_long(l, lds, N, G, g);
    l.w0 = w0;
    l.W = W;
    Cst c;
    load_cst_lds(c, grp_cst(lds, N, g));
    const AlmSt al = grp_almst(ga, grp_state(lds, N, g)->J_pair);
    return total_cost_lds<true>(c, l, al, lane);
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

// ---------------------------------------------------------------------------------------------
// The rollout pass of the LONG layout (horizons of 64 ... 127): as rollout_group, but the gains of a whole horizon do not fit
// the shared LDS area (2 x 10 N doubles = 16 KB at N = 100), so they come in through a ring of two chunks of CILQR_GRPL_CHUNK
// steps per trajectory: chunk c + 1 is fetched from global memory — by all 64 lanes, 16 bytes each (+ 16 on the first lanes) —
// while the steps of chunk c run, and dropped into the ring half (c + 1) & 1 when the loop first asks for one of its steps.
// Every lane of the wavefront runs the loop (the refill needs them all): a lane that has no trial of its own — or a trial of
// the other vehicle model — shadows a lane that has, with its stores sent outside the buffer's range (dropped by the
// hardware's range check), so the wave-uniform decisions (small-angle form or general form) see only real trials.
struct LongStage {
    unsigned src1, src2; // byte offsets (block scratch) of this lane's 16 bytes of chunk 0: element lane, element lane + 64
    unsigned ring0;      // LDS byte address of ring half 0 (wave-uniform: element e of a chunk lands at + 16 e)
    bool two;            // lane + 64 is an element of the chunk
};
typedef unsigned __attribute__((address_space(3))) lds_u32;

template <int G>
__device__ inline void roll_fetch_l(RollIn& g, const GrpRoll& q, int i) {
    const unsigned xa = q.xaddr, ua = q.uaddr;
    const unsigned ka = q.kaddr + (unsigned)(((i / CILQR_GRPL_CHUNK) & 1) * (G * CILQR_GRPL_CHUNK_BYTES) + (i % CILQR_GRPL_CHUNK) * (CILQR_KD * 8));
    const f64x2 a = *(lds_cf64x2*)(size_t)(xa + 32u * (unsigned)i);
    const f64x2 b = *(lds_cf64x2*)(size_t)(xa + 32u * (unsigned)i + 16u);
    const f64x2 u = *(lds_cf64x2*)(size_t)(ua + 16u * (unsigned)i);
    g.x[0] = a.x; g.x[1] = a.y; g.x[2] = b.x; g.x[3] = b.y;
    g.u[0] = u.x; g.u[1] = u.y;
#pragma unroll
    for (int j = 0; j < CILQR_KD / 2; ++j) {
        const f64x2 v = *(lds_cf64x2*)(size_t)(ka + 16u * (unsigned)j);
        g.k[2 * j] = v.x;
        g.k[2 * j + 1] = v.y;
    }
}

template <int RP, int PIN, int G>
__device__ inline int rollout_long_rp(int N, __amdgpu_buffer_rsrc_t rs, const GrpRoll& q, const LongStage& sg) {
    const int nch = (N + CILQR_GRPL_CHUNK - 1) / CILQR_GRPL_CHUNK;
    // The gains ring is filled by LDS-DMA (buffer_load ... lds: global memory -> LDS without passing through registers; the
    // element lane e of a chunk lands at ring half + 16 e, which is the chunk's layout): no registers held for a chunk in
    // flight, and — the point — no compiler-placed wait.  Through registers the compiler, which cannot count the slab stores
    // between a load in one trip of the loop and its use eight steps later, put s_waitcnt vmcnt(0) in front of the LDS
    // write: every eighth step waited for all its predecessors' slab stores to reach L2 (the pass 50 % longer per step than
    // the short horizons').  The vector-memory counter retires in order, so waiting until at most 20 operations are
    // outstanding when a chunk is first needed covers its DMA: 23 (first chunk) or 24 slab stores were issued behind it.
    int have = 0; // chunks 0 .. have have been asked for and waited for (ring half = chunk & 1)
    auto issue = [&](int c) {
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (LDS reads of the half about to be overwritten have returned)
        lds_u32* const d1 = (lds_u32*)(size_t)(sg.ring0 + (unsigned)((c & 1) * (G * CILQR_GRPL_CHUNK_BYTES)));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, sg.src1, c * CILQR_GRPL_CHUNK_BYTES, 0, 0);
        if (sg.two) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1 + 256, 16, sg.src2, c * CILQR_GRPL_CHUNK_BYTES, 0, 0);
    };
    auto need = [&](int j) { // before the first read of step j's gains
        const int c = j / CILQR_GRPL_CHUNK;
        if (c > have) {
            __asm__ volatile("s_waitcnt vmcnt(20)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            have = c;
            if (c + 1 < nch) issue(c + 1);
        }
    };
    issue(0);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (nch > 1) issue(1);
    const f64x2 a0 = *(lds_cf64x2*)(size_t)(q.xaddr);
    const f64x2 b0 = *(lds_cf64x2*)(size_t)(q.xaddr + 16u);
    double xc[4] = {a0.x, a0.y, b0.x, b0.y};
    slab_st2(rs, q.vrow0, xc[0], xc[1]);
    slab_st2(rs, q.vrow0 + q.pairb, xc[2], xc[3]);
    GrpOut o;
    o.bx0 = q.vrow0;
    o.bx1 = q.vrow0 + q.pairb;
    o.bu = q.vrow0 + 2u * q.pairb;
    DmPinned pk;
    if (PIN) dm_pin_load(pk);
    // straight-line small-angle loop handing over to the general one, two register sets, inputs of step i + 1 fetched while
    // step i computes (rollout_group_rp) — but the hand-over keeps both sets instead of fetching step i again: its chunk's
    // ring half may already be the target of the next DMA
    int i = 0;
    RollIn ga, gb;
    bool second = false; // gb (not ga) holds the step the small-angle loop stopped at
    roll_fetch_l<G>(ga, q, 0);
    for (;;) {
        if (i >= N) break;
        if (i + 1 < N) { need(i + 1); roll_fetch_l<G>(gb, q, i + 1); }
        if (!roll_step_g<RP, true, PIN>(q, rs, pk, ga, xc, o, i)) break;
        ++i;
        if (i >= N) break;
        if (i + 1 < N) { need(i + 1); roll_fetch_l<G>(ga, q, i + 1); }
        if (!roll_step_g<RP, true, PIN>(q, rs, pk, gb, xc, o, i)) { second = true; break; }
        ++i;
    }
    const int i_small = i;
    if (i < N) {
        if (second) { const RollIn t = ga; ga = gb; gb = t; } // ga: step i, gb: step i + 1
        for (;;) {
            roll_step_g<RP, false, PIN>(q, rs, pk, ga, xc, o, i);
            ++i;
            if (i >= N) break;
            if (i + 1 < N) { need(i + 1); roll_fetch_l<G>(ga, q, i + 1); }
            roll_step_g<RP, false, PIN>(q, rs, pk, gb, xc, o, i);
            ++i;
            if (i >= N) break;
            if (i + 1 < N) { need(i + 1); roll_fetch_l<G>(gb, q, i + 1); }
        }
    }
    return i_small;
}

template <int G, int PIN>
__device__ __attribute__((noinline)) bool rollout_group_long(double* lds_base, double* scr_blk, int N_arg, int lane) {
    const int N = uniform_int(N_arg); // (see rollout_group: the horizon back into a scalar register)
    const int R = N + 1;
    static_assert(G <= 3, "lanes: three searches of 20 step sizes fit a wavefront; a chunk of three trajectories is 120 elements");
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
    double* const ring = lds_base + (size_t)G * grp_pg_doubles(N);
    const unsigned slot_b = (unsigned)(grp_scratch_doubles(N) * sizeof(double));
    const unsigned gains_b = (unsigned)((slab_doubles(N) + first_trial_doubles(N)) * sizeof(double));
    constexpr int EPG = CILQR_GRPL_CHUNK * CILQR_KD / 2; // 16-byte elements of a chunk per trajectory
    LongStage sg;
    {
        const int e1 = lane, e2 = lane + CILQR_WAVE;
        const int g1 = e1 / EPG, r1 = e1 % EPG;
        const int g2 = (e2 < EPG * G) ? e2 / EPG : 0, r2 = e2 % EPG;
        sg.two = e2 < EPG * G;
        sg.src1 = (unsigned)g1 * slot_b + gains_b + 16u * (unsigned)r1;
        sg.src2 = (unsigned)g2 * slot_b + gains_b + 16u * (unsigned)r2;
        static_assert(EPG * G > CILQR_WAVE && EPG * G <= 2 * CILQR_WAVE, "a chunk = one full DMA of 64 elements + a partial one");
        sg.ring0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_cdouble*)ring);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uniform_ptr(scr_blk), 0, (int)(G * grp_scratch_doubles(N) * sizeof(double)), 0x00020000);
    GrpRoll q;
    q.alpha = 1.0; q.dt = 0.0; q.wb = 1.0;
    q.xaddr = q.uaddr = q.kaddr = lds_addr(lds_base);
    q.goff = 0u; q.vrow0 = 0u; q.pairb = 0u; q.tileb = 0u;
    int rp = -1;
    if (gl >= 0) {
        double* pg = lds_base + (size_t)gl * grp_pg_doubles(N);
        const GrpSt* st = grp_state(lds_base, N, gl);
        q.alpha = dm_pow2i(-al);
        q.dt = st->dt;
        q.wb = st->wb;
        rp = st->rp;
        q.xaddr = lds_addr(pg);
        q.uaddr = lds_addr(pg + 4 * R);
        const unsigned as = (rq == 2) ? (unsigned)CILQR_MAX_ALPHA_TRIALS : 1u;
        const unsigned gbase = (unsigned)gl * slot_b;
        q.kaddr = lds_addr(ring) + (unsigned)gl * CILQR_GRPL_CHUNK_BYTES;
        q.goff = gbase + gains_b;
        q.tileb = as * (unsigned)(2 * CILQR_SLAB_TILE * sizeof(double));
        q.pairb = (unsigned)CILQR_SLAB_RT(R) * q.tileb;
        q.vrow0 = gbase + ((rq == 2) ? 0u : (unsigned)(slab_doubles(N) * sizeof(double))) +
                  (unsigned)(2 * CILQR_SLAB_TILE * sizeof(double)) * (unsigned)al;
    }
    for (int model = 0; model < 2; ++model) {
        const bool mine = (gl >= 0) && (rp == model);
        const unsigned long long m = __ballot(mine);
        if (m == 0ULL) continue;
        const int proto = __ffsll((long long)m) - 1;
        GrpRoll qq;
        qq.alpha = __shfl(q.alpha, proto, CILQR_WAVE);
        qq.dt = __shfl(q.dt, proto, CILQR_WAVE);
        qq.wb = __shfl(q.wb, proto, CILQR_WAVE);
        qq.xaddr = __shfl(q.xaddr, proto, CILQR_WAVE);
        qq.uaddr = __shfl(q.uaddr, proto, CILQR_WAVE);
        qq.kaddr = __shfl(q.kaddr, proto, CILQR_WAVE);
        qq.goff = __shfl(q.goff, proto, CILQR_WAVE);
        qq.pairb = __shfl(q.pairb, proto, CILQR_WAVE);
        qq.tileb = __shfl(q.tileb, proto, CILQR_WAVE);
        qq.vrow0 = 0x40000000u; // (a shadow lane: its stores fall outside the buffer)
        if (mine) qq = q;
        int i_small;
        if (model == 0) i_small = rollout_long_rp<0, PIN, G>(N, rs, qq, sg);
        else i_small = rollout_long_rp<1, PIN, G>(N, rs, qq, sg);
        if (CILQR_GPROF && mine && al == 0) grp_state(lds_base, N, gl)->small_steps = i_small;
        wave_sync();
    }
    if (lane < G) grp_state(lds_base, N, lane)->req = 0;
    wave_sync();
    return true;
}

} // namespace cilqr
