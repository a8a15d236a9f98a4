// cilqr_kernels.hpp — the fused solve kernel k_solve (one block = one trajectory, or persistent blocks pulling
// trajectories) and the table of its builds.  Included by cilqr_amd.hip (host side, piecewise kernels) and by
// cilqr_solve_inst.hip, which is compiled once per group of builds so that the library builds in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "cilqr_device.hpp"
#include "cilqr_group.hpp"

using namespace cilqr;

extern __shared__ double g_lds[];

#ifndef CILQR_OOL_TWO_ROWS
#define CILQR_OOL_TWO_ROWS 0 /* 1: the two-rows-per-lane large-batch builds of k_solve run their heavy phases out of line (ool_expand_sweep) — measured 1-2 % slower on configs[3] (r04_experiments): off */
#endif
#ifndef CILQR_OOL_SWEEP
#define CILQR_OOL_SWEEP 1
#endif
#ifndef CILQR_OOL_COST
#define CILQR_OOL_COST 1
#endif
#ifndef CILQR_GRP_PAIR_SWEEP
#define CILQR_GRP_PAIR_SWEEP 1 /* the grouped kernel sweeps both trajectories of a wavefront in one instruction stream (backward_sweep_pair) */
#endif
#ifndef CILQR_SOLVE_WAVES_PER_SIMD
#define CILQR_SOLVE_WAVES_PER_SIMD 1
#endif

struct BatchArgs {
    const cilqr_params* params;
    const DevScene* scenes;
    const int32_t* scenario_id; // may be null
    const int32_t* param_id;    // may be null
    const int32_t* tick;        // may be null
    double* scratch;            // [grid][scratch_doubles(N)]
    long long* prof;            // optional [B][CILQR_PROF_SLOTS] cycles per phase + counters (null = off)
    int B;
    int N;
    int n_params, n_scenes;     // table sizes: the fused solve checks its ids against them
    int W;                      // capacity (samples) of the per-trajectory LDS lane window
    int flags;                  // CILQR_DBG_* (testing aids)
    int tier;                   // line-search rollouts: -1 adaptive (default), 0 always all 20 step sizes in one
                                // pass, 1 always the first trial alone first (cilqr_set_rollout_mode)
    // augmented-Lagrangian state kept by the handle (solve_type alm): [B][N][alm_C], [B]
    double* alm_mu;
    double* alm_mu_next;
    double* alm_rho;
    int alm_C;
    int alm;                    // 1 when the handle's parameter sets use the ALM solve type
    // work sharing between blocks (k_solve's SHARE; null = off): counters + slots, one request and one row of
    // reference-index hints per trajectory
    unsigned* sh_ctl;
    ShareReq* sh_req;
    int* sh_hints;              // [B][N + 2]
    int sh_max_helpers;         // blocks that stay to help (the others leave when they are done)
    int sh_min_t0;              // a search is announced once this many of its trials have been rejected
    int sh_backoff;             // an idle helper looks again after 1 us, doubling up to 2^sh_backoff us
    unsigned* next;             // persistent blocks (large batches): the next trajectory to hand out; null = one block
                                // per trajectory
    long long* timeline;        // optional [B][4]: start / end of the solve of trajectory b (constant 100 MHz clock), the
                                // index of the block that solved it, the XCC it ran on (cilqr_set_block_timeline; null = off)
    // resumable solves (k_solve's RES; null / 0 = every solve runs to its end in one go)
    double* park;               // [B][park_doubles(N)] state of the solves that were interrupted
    unsigned long long* rq;     // [rq_cap] queue of their numbers (counters in the control words: SH_Q_*)
    unsigned* ctl;              // the launch's control words
    int rq_cap;
    int res_iters;              // iterations per slice
    int res_window;             // (grouped build) slices end in a hand-over only once fewer fresh trajectories than this are left
    // closed planning loop on the device (cilqr_closed_loop_batch_device): every ego runs `loop_ticks` ticks back to back —
    // solve, ego <- x.row(1), obstacle window one tick on, warm start from the plan just made (mp:180-197, cs:163-180)
    int loop_ticks;             // 0 / 1 = one solve per trajectory
    double* loop_x0;            // [B][4] in / out: the ego states (the solve reads them through its x0 argument)
    int32_t* loop_tick;         // [B] in / out
    double* loop_states;        // optional [B][loop_ticks][4]: the ego state after every tick
    int32_t* loop_iters;        // optional [loop_ticks][B]: iterations of every tick's solve
    int pair_costs;             // grouped build: line-search trials after the first costed two per pass
    int pair_sweep;             // grouped build: the backward sweeps of a wavefront's two trajectories in one instruction stream
                                // (1; 0 = one after the other as in round 4 — development library, CILQR_TUNE=pair_sweep=0)
};

__device__ inline AlmSt load_alm(const BatchArgs& a, int b, int N) {
    AlmSt al;
    al.C = a.alm_C;
    al.mu = a.alm_mu ? a.alm_mu + (size_t)b * N * a.alm_C : nullptr;
    al.mu_next = a.alm_mu_next ? a.alm_mu_next + (size_t)b * N * a.alm_C : nullptr;
    al.rho = a.alm_rho ? a.alm_rho[b] : 1.0;
    return al;
}

// phase ids of the optional in-kernel cycle accounting
enum { PH_INIT = 0, PH_DERIV = 1, PH_BACKWARD = 2, PH_ROLLOUT = 3, PH_TRIAL_COST = 4, PH_ACCEPT = 5, PH_TOTAL = 6, PH_ITERS = 7, PH_REF_FALLBACKS = 8, PH_TRIALS = 9, PH_TC_REF = 10, PH_TC_STAGE = 11, PH_TC_SUM = 12, PH_TC_SAMPLED = 13, PH_ROLL_FIRST = 14, PH_ROLL_ALL = 15, PH_ROLL_SECOND = 16 };
#define PROF_T0() long long t_ph_ = (PROF && a.prof) ? (long long)__builtin_readcyclecounter() : 0
#define PROF_ADD(ph)                                                  \
    do {                                                              \
        if (PROF && a.prof) {                                         \
            long long t_now_ = (long long)__builtin_readcyclecounter(); \
            if (lane == 0) ph_acc[ph] += t_now_ - t_ph_;              \
            t_ph_ = t_now_;                                           \
        }                                                             \
    } while (0)

template <bool LOOP = false>
__device__ inline void load_cst(Cst& c, const BatchArgs& a, int b, const Lds& l, int lane, bool fill = true) {
    int pid = a.param_id ? a.param_id[b] : 0;
    int sid = a.scenario_id ? a.scenario_id[b] : 0;
    int tk = a.tick ? a.tick[b] : 0;
    if (LOOP) tk = uniform_int((int)sh_ld(reinterpret_cast<const unsigned*>(a.loop_tick + b))); // (advanced inside this launch)
    make_cst(c, a.params[pid], a.scenes[sid], tk, l.ck, lane, fill);
}

// The device-pointer entry point cannot check its index arrays on the host.  A trajectory whose ids point
// outside the tables, or whose obstacle routes end before tick + N + 1 (upstream: RoutingLine::operator[]
// throws std::out_of_range, ut:52-58), is not solved: NaN outputs, end_reason CILQR_END_BAD_INPUT.
template <bool LOOP = false>
__device__ inline bool ids_valid(const BatchArgs& a, int b) {
    const int pid = a.param_id ? a.param_id[b] : 0;
    const int sid = a.scenario_id ? a.scenario_id[b] : 0;
    int tk = a.tick ? a.tick[b] : 0;
    if (LOOP) tk = uniform_int((int)sh_ld(reinterpret_cast<const unsigned*>(a.loop_tick + b)));
    if ((unsigned)pid >= (unsigned)a.n_params || (unsigned)sid >= (unsigned)a.n_scenes || tk < 0) return false;
    const DevScene& s = a.scenes[sid];
    return !(s.M > 0 && (long long)tk + a.N + 1 > (long long)s.T);
}

// A block whose own trajectory is done costs open line-search trials of the blocks still running (see ShareReq)
// until every trajectory of the launch has finished.  `me` only spreads the helpers over the open searches.
// (not inlined: it runs once, after the solve, with nothing live — inlined, its copy of the trial costing costs the
//  solve loop 27 more spilled vector registers)
template <int NCH, int NC, bool ALM>
__device__ __attribute__((noinline)) void share_help(BatchArgs a, Lds l, int N, int lane, int me) {
    unsigned* const ctl = a.sh_ctl;
    // enough helpers already?  (a look first: the read-modify-write only when there is a chance)
    if (sh_ld_u(ctl + SH_HELPING, lane) >= (unsigned)a.sh_max_helpers) return; // (a look first: no read-modify-write without a chance)
    if (sh_add_u(ctl + SH_HELPING, 1u, lane) >= (unsigned)a.sh_max_helpers) {
        (void)sh_add_u(ctl + SH_HELPING, 0u - 1u, lane);
        return;
    }
    (void)sh_add_u(ctl + SH_HELPERS, 1u, lane);
    int cur_b = -1, idx0h = 0, nfb = 0, idle = 0, slot_b = 0;
    unsigned cur_seq = 0;
    Cst c2;
    AlmSt al2;
    al2.mu = nullptr; al2.mu_next = nullptr; al2.rho = 1.0; al2.C = 0;
    for (int spin = 0; spin < (1 << 22); ++spin) { // (bounded: a launch lasts milliseconds, this is seconds)
        if (sh_ld_u(ctl + SH_FINISHED, lane) >= (unsigned)a.B) break;
        const unsigned sv = sh_ld(ctl + SH_SLOT0 + lane); // the 64 slots, one per lane
        unsigned long long open = __ballot(sv != 0u);
        const int rot = (me * 7 + spin) & 63;
        if (rot) open = (open >> rot) | (open << (64 - rot));
        bool worked = false;
        while (open && !worked) {
            const int j = __builtin_ctzll(open);
            open &= open - 1;
            const int bb = (int)__shfl((int)sv, (j + rot) & 63, CILQR_WAVE) - 1;
            if (bb < 0 || bb >= a.B) continue;
            ShareReq* rq = a.sh_req + bb;
            const unsigned v = sh_ld_u(&rq->claim, lane);
            const unsigned next = v & 0xffu;
            if (next >= (unsigned)CILQR_MAX_ALPHA_TRIALS) continue; // closed, or every trial handed out
            if (sh_cas_u(&rq->claim, v, v + 1u, lane) != v) continue; // somebody else moved it: look again later
            const int t = (int)next;
            const unsigned seq = v >> 16;
            if (bb != cur_b || seq != cur_seq) {
                sh_acquire(); // the owner's slab, hints and row-0 index of this search
                load_cst(c2, a, bb, l, lane);
                if (NC) c2.N = NC;
                wave_sync();
                idx0h = rq->idx0;
                slot_b = rq->slot; // the owner block's scratch area (its slab)
                if (ALM) { // the owner's multipliers (global memory) and its penalty weight of this search
                    al2 = load_alm(a, bb, N);
                    al2.rho = dm_from_bits(rq->rho_bits);
                }
                stage_window(c2, l, idx0h, a.W, lane);
                const int* hints = a.sh_hints + (size_t)bb * (N + 2);
                for (int k = lane; k <= N; k += CILQR_WAVE) { l.ridx[k] = hints[k]; l.tidx[k] = hints[k]; }
                wave_sync();
                cur_b = bb;
                cur_seq = seq;
            }
            double J1[1];
            total_cost_trials<false, NCH, ALM, 1>(c2, l, al2, a.scratch + (size_t)slot_b * scratch_doubles(N), t, 1, lane, idx0h,
                                                  0, &nfb, J1, nullptr, 0, CILQR_MAX_ALPHA_TRIALS);
            if (lane == 0) sh_st64(&rq->J[t], dm_to_bits(J1[0]));
            (void)sh_add_u(ctl + SH_HELPED, 1u, lane);
            worked = true;
        }
        // nothing to do: look again after 1 us, backing off
        if (worked) idle = 0;
        else {
            idle = (idle < a.sh_backoff) ? idle + 1 : a.sh_backoff;
            for (int r = 0; r < (1 << idle); ++r) __builtin_amdgcn_s_sleep(32);
        }
    }
    (void)sh_add_u(ctl + SH_HELPING, 0u - 1u, lane);
}

// Which trajectory a block solves.  Workgroups go to the chip's eight XCDs round-robin (block i to XCD i mod 8) and
// each XCD works through its own share, so a batch whose work per trajectory correlates with the index mod 8 — a
// parameter sweep laid out setting-fastest (config 5: setting = b mod 16), scenarios dealt out in turn (config 4:
// b mod 4) — loads the XCDs unevenly: measured 0.85 ... 1.27 of the mean, the launch waiting for the fullest.  Each
// group of eight consecutive trajectories is therefore rotated by a hash of its number before it is dealt out to the
// eight XCDs (a bijection; the last, partial group keeps its order): config 5 92.8 -> 80.6 ms, config 4 67.3 -> 56 ms.
__device__ inline int trajectory_of_block(unsigned blk, int B) {
    const unsigned q = blk >> 3, k = blk & 7u;
    if ((q + 1u) * 8u > (unsigned)B) return (int)blk;
    const unsigned h = (q * 0x9E3779B1u) >> 29;
    return (int)(8u * q + ((k + h) & 7u));
}

// CILQRSolver::solve (cs:85-153) + iter_step (cs:337-381)
// DBG = true compiles the testing-aid paths in (cilqr_set_debug_flags); the production
// instantiation carries neither their code nor their registers.
// HELP = true: the block has a second wavefront that does nothing but cost every other trial of the
// line search (slot 1) while the main wavefront costs the ones in between (slot 0); used when the
// batch is too small to fill the chip with one wavefront per trajectory.  Control words in LDS:
// CTL_MODE of an iteration: 0 = no line search (the backward pass failed), 1 = the slab holds all 20 trial
// trajectories, 2 = only the first trial exists so far (in the first-trial buffer; the main wave costs it alone)
enum { CTL_MODE = 0, CTL_EXIT = 2, CTL_IDX0 = 3, CTL_W0 = 4, CTL_W = 5 };
// doubles: the helper's / the main wave's cost of the current pass (two slots each, alternating), and what the
// helper needs to reach the main wave's verdicts on its own
enum { CTLD_JH = 0, CTLD_JM = 2, CTLD_JCUR = 4, CTLD_DV = 5, CTLD_RHO = 7 };
#define BLOCK_BAR()            \
    do {                       \
        if (HELP) __syncthreads(); \
    } while (0)

// WPS = waves per SIMD the register allocation must allow (2 for big batches: more resident
// trajectories at the price of a few spills)
// NTP = trials costed per pass after the first when there is no helper wavefront: 2 overlaps the two
// trials' latencies inside one wavefront (pays while SIMDs hold one or two wavefronts: +5 % at B = 1536-2048);
// 1 keeps fewer values live (27 instead of 68 spilled registers) and is the faster choice once every SIMD
// holds two wavefronts anyway (+1.5 % at B >= 3072)
// NC = horizon known at compile time (0: taken from the parameter table): every LDS offset but the lane window's
// and every row count become constants — fewer live scalar registers, addresses folded into instruction offsets
// LG = the cost expansion (l_x, l_u, l_xx, l_uu) lives in global memory instead of LDS (CILQR_GL_ROW): large batches
// of long horizons, where the LDS block of a trajectory would cap the CU at 5 wavefronts
// SHARE = blocks that have finished help the ones still running with their line searches (ShareReq; lone wavefronts,
// barrier mode, one trial per pass); switched on per launch by a.sh_ctl
// solve_one = the solve of trajectory b by the calling block; `slot` = which scratch area (slab, ...) it uses
// A parked solve's state between global memory (8-byte agent-scope atomics, see rq_push) and the block's LDS: x, u, the
// lane indices, and the solve's scalars through 12 doubles of LDS (`sc`).  Out of line: it runs once per slice, and
// inlined its loops raise the register pressure of the whole solve (80 -> 153 spilled vector registers).
__device__ __attribute__((noinline)) void park_copy(double* pk, double* lx, double* lu, int* ridx, double* sc, int N, int lane,
                                                    int store) {
    double* const pk_sc = pk + 4 * (N + 1) + 2 * N;
    unsigned* const pk_ix = reinterpret_cast<unsigned*>(pk_sc + CILQR_PARK_SCALARS);
    if (store) {
        for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) park_st(pk + e, lx[e]);
        for (int e = lane; e < 2 * N; e += CILQR_WAVE) park_st(pk + 4 * (N + 1) + e, lu[e]);
        for (int k = lane; k <= N; k += CILQR_WAVE) sh_st(pk_ix + k, (unsigned)ridx[k]);
        if (lane < 12) park_st(pk_sc + lane, sc[lane]);
    } else {
        for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) lx[e] = park_ld(pk + e);
        for (int e = lane; e < 2 * N; e += CILQR_WAVE) lu[e] = park_ld(pk + 4 * (N + 1) + e);
        for (int k = lane; k <= N; k += CILQR_WAVE) ridx[k] = (int)sh_ld(pk_ix + k);
        if (lane < 12) sc[lane] = park_ld(pk_sc + lane);
    }
    wave_sync();
}


// ------------------------------------------------------------------------------------------------
// Round 4: the lesson of the grouped build (cilqr_group.hpp) applied to the two-rows-per-lane builds of k_solve (horizons
// above 63 in large batches: work sharing, resumable solves, expansion in LDS or in global memory).  Those builds carried 99
// spilled vector registers and a scratch access inside the backward step: everything a solve keeps in registers is live across
// its phases.  With OOL the three heavy phases — expansion + sweep, trial cost — are functions of their own that rebuild
// what they need from LDS (the trajectory's arrays through carve(), its by-value constants from a copy parked in the unused
// cycle-accounting slots) and a handful of arguments, and hand their results back through LDS.  Same device functions, same
// bits.
template <int NCH, int NC, bool LG>
__device__ __attribute__((noinline)) bool ool_expand_sweep(double* lds, int n_rt, int Wcap, double* gl, int w0, int Wcur, double lamb,
                                                            int lane, int expand) {
    const int N = NC ? NC : uniform_int(n_rt); // (an argument: in a vector register — scalar again, or descriptors built from it count as divergent)
    Lds l;
    carve(l, lds, N, Wcap, 0, 1, LG ? 1 : 0);
    l.gl = gl;
    l.w0 = w0;
    l.W = Wcur;
    Cst c;
    load_cst_lds(c, reinterpret_cast<const Cst*>(l.prof));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    if (expand) cost_and_model_derivatives<false, LG>(c, l, al, lane);
    else model_jacobians(c, l, lane); // (cs:469-475: the expansion of the unchanged trajectory is kept)
    double dV[2];
    const bool ok = backward_sweep_lanes<LG ? CILQR_GL_ROW : 0>(c, l, lamb, lane, dV);
    wave_sync();
    if (lane == 0) { l.ctld[CTLD_DV] = dV[0]; l.ctld[CTLD_DV + 1] = dV[1]; }
    wave_sync();
    return ok;
}
template <int NCH, int NC, bool LG>
__device__ __attribute__((noinline)) double ool_cost_trial(double* lds, int n_rt, int Wcap, int w0, int Wcur, const double* src, int t,
                                                            int as, int idx0, int lane) {
    const int N = NC ? NC : uniform_int(n_rt); // (an argument: in a vector register — scalar again, or descriptors built from it count as divergent)
    Lds l;
    carve(l, lds, N, Wcap, 0, 1, LG ? 1 : 0);
    l.w0 = w0;
    l.W = Wcur;
    Cst c;
    load_cst_lds(c, reinterpret_cast<const Cst*>(l.prof));
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    int nfb = 0;
    double J1[1];
    total_cost_trials<false, NCH, false, 1>(c, l, al, src, t, 1, lane, idx0, 0, &nfb, J1, nullptr, 0, as);
    if (nfb != 0 && lane == 0) l.ctli[7] += nfb;
    return J1[0];
}

// what solve_one returns besides a finished solve's iteration count
enum { SOLVE_PARKED = -1 /* parked again: its number has been queued */, SOLVE_BAD_INPUT = -2 };

// RES = resumable: the solve runs a.res_iters iterations at a time and is parked in between (see rq_push); `resumed` = this
// call continues a parked solve.
// LOOP = the launch runs a closed planning loop (cilqr_closed_loop_batch_device): x0 and tick are advanced inside it
template <bool DBG, int NCH, bool ALM, bool HELP, bool PROF, int WPS, int NTP, int NC, bool LG, bool SHARE, bool RES, bool LOOP>
__device__ __forceinline__ int solve_one(const BatchArgs& a, const int b, const int slot, const bool resumed,
                                          const double* __restrict__ x0,
                                          const double* last_u, double* u_out, // (not restrict: a later tick of the closed
                                                                               //  loop warm-starts from the plan in u_out)
                                          double* __restrict__ x_out, cilqr_result* __restrict__ res_out,
                                          cilqr_trace_rec* __restrict__ trace_out, int trace_cap) {
    const int lane = threadIdx.x & (CILQR_WAVE - 1);
    const int wave = HELP ? (threadIdx.x >> 6) : 0;
    const long long tl_start = a.timeline ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
    static_assert(!SHARE || (!HELP && !PROF && NTP == 1), "work sharing: lone wavefronts costing one trial per pass");
    static_assert(!RES || (!HELP && !PROF && !ALM), "resumable solves: lone wavefronts, barrier mode (the multipliers of the "
                                                      "augmented Lagrangian are large arrays written with plain stores)");
    static_assert(!LOOP || (!RES && !SHARE && !PROF), "closed loop: plain builds");
    const bool res_on = RES && a.park != nullptr;
    const bool share = SHARE && a.sh_ctl != nullptr;
    // the heavy phases out of line (see ool_expand_sweep): the two-rows-per-lane barrier builds of the large batches
    constexpr bool OOL = CILQR_OOL_TWO_ROWS && NCH == 2 && !DBG && !ALM && !HELP && !PROF && !LOOP && NTP == 1;
    const int N = NC ? NC : a.N; // one horizon per handle
    if (!ids_valid<LOOP>(a, b)) { // wave-uniform, before the wavefronts of a helper-mode block part ways
        if (share || res_on) (void)sh_add_u(a.ctl + SH_FINISHED, 1u, lane);
        if (wave == 0) {
            const double qnan = dm_from_bits(0x7ff8000000000000ULL);
            for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) x_out[(size_t)b * 4 * (N + 1) + e] = qnan;
            for (int e = lane; e < 2 * N; e += CILQR_WAVE) u_out[(size_t)b * 2 * N + e] = qnan;
            if (lane == 0 && res_out) {
                cilqr_result r;
                r.J_init = qnan; r.J_final = qnan; r.iters = 0; r.end_reason = CILQR_END_BAD_INPUT;
                r.final_status = CILQR_RUNNING; r.ls_trials = 0; r.cost_evals = 0; r.trace_len = 0;
                res_out[b] = r;
            }
        }
        return SOLVE_BAD_INPUT;
    }
    Lds l;
    constexpr int SLOTS = (HELP || NTP == 2) ? 2 : 1; // trials costed concurrently (the host sizes the LDS block alike)
    carve(l, g_lds, N, a.W, ALM ? 1 : 0, SLOTS, LG ? 1 : 0);
    Cst c;
    load_cst<LOOP>(c, a, b, l, lane);
    if (NC) c.N = NC;
    if (OOL) {
        if (lane == 0) { *reinterpret_cast<Cst*>(l.prof) = c; l.ctli[7] = 0; }
        wave_sync();
    }
    double* scr = a.scratch + (size_t)slot * scratch_doubles(N);
    if (LG) l.gl = scr + scratch_gl_offset(N);
    double* first = scr + slab_doubles(N); // the first-trial buffer
    AlmSt al = load_alm(a, b, N);
    if (HELP && wave == 1) {
        // ---- helper wavefront: costs trial 2p + 1 of every pass p, mirrors the main wave's control flow ----
        __syncthreads(); // B0: window, ridx, x, u staged by the main wave
        const int idx0h = l.ctli[CTL_IDX0];
        l.w0 = l.ctli[CTL_W0];
        l.W = l.ctli[CTL_W];
        int nfb = 0;
        for (int itr = 0; itr < c.max_iter; ++itr) {
            __syncthreads(); // B1: K, d, trial slab of this iteration are ready (or the backward pass failed)
            const int mode = l.ctli[CTL_MODE];
            if (mode) {
                if (ALM) al.rho = l.ctld[CTLD_RHO];
                const double Jc = l.ctld[CTLD_JCUR], dV0 = l.ctld[CTLD_DV], dV1 = l.ctld[CTLD_DV + 1];
                const double conv_thr = c.k->conv_thr, accept_thr = c.k->accept_thr;
                int t0 = 0;
                bool over = false;
                if (mode == 2) {
                    // the main wave costs the first trial on its own; if the search goes on it rolls the others out
                    __syncthreads(); // B2s: its cost is in LDS
                    if (trial_verdict(Jc, l.ctld[CTLD_JM], 0, dV0, dV1, conv_thr, accept_thr) != 0) over = true;
                    else { __syncthreads(); /* B3: the slab is filled */ t0 = 1; }
                }
                if (!over) {
                    for (int par = 0; t0 < CILQR_MAX_ALPHA_TRIALS; t0 += 2, par ^= 1) {
                        double J1[1] = {0.0};
                        const bool mine = (t0 + 1 < CILQR_MAX_ALPHA_TRIALS);
                        if (mine) {
                            total_cost_trials<DBG, NCH, ALM, 1>(c, l, al, scr, t0 + 1, 1, lane, idx0h, a.flags, &nfb, J1, nullptr, 1);
                            if (lane == 0) l.ctld[CTLD_JH + par] = J1[0];
                        }
                        __syncthreads(); // B2: both costs of this pass are in LDS (slots alternate between passes)
                        // the same verdict the main wave reaches (trial_verdict is a pure function of these numbers)
                        const double J0 = l.ctld[CTLD_JM + par];
                        if (trial_verdict(Jc, J0, t0, dV0, dV1, conv_thr, accept_thr) != 0) break;
                        if (mine && trial_verdict(Jc, J1[0], t0 + 1, dV0, dV1, conv_thr, accept_thr) != 0) break;
                    }
                }
            }
            __syncthreads(); // B4
            if (l.ctli[CTL_EXIT]) break;
        }
        return 0;
    }
    // the main wavefront carries the serial chain of the solve: it wins issue arbitration against the helper
    // wavefront (of another block) it shares its SIMD with
    if (HELP) __builtin_amdgcn_s_setprio(2);
    if (ALM && last_u == nullptr && !(RES && resumed)) {
        // cs:88-93: fresh multipliers unless this call continues a previous solution
        al.rho = c.k->alm_rho_init;
        for (int e = lane; e < N * al.C; e += CILQR_WAVE) { al.mu[e] = 0.0; al.mu_next[e] = 0.0; }
        wave_sync();
    }

    // the cycle accounting lives in LDS (written by lane 0 only): as a register array it would cost the profiling
    // build 34 vector registers and push spill reloads into the backward loop
    long long* const ph_acc = l.prof;
    if (PROF && a.prof) {
        for (int e = lane; e < CILQR_PROF_SLOTS; e += CILQR_WAVE) ph_acc[e] = 0;
        wave_sync();
    }
    const long long t_begin = (PROF && a.prof) ? (long long)__builtin_readcyclecounter() : 0;
    PROF_T0();
    int idx0;
    double J_cur, J_init;
    double r_lamb = 0.0;  // the other scalars of a resumed solve, applied where the solve's own are set up
    int r_status = 0, r_iters = 0, r_trials = 0, r_evals = 0, r_tl = 0, r_flag = 0, r_deep = 0, r_seq = 1;
    double* const pk = (RES && res_on) ? a.park + (size_t)b * park_doubles(N) : nullptr;
    if (RES && resumed) {
        // the parked state: x, u, lane indices (the window is staged again from the same row-0 index)
        double* const sc_ = l.xch; // (12 doubles: the sweep's constants and, behind them, the helper-mode words — free here)
        park_copy(pk, l.x, l.u, l.ridx, sc_, N, lane, 0);
        idx0 = uniform_int((int)sc_[10]);
        stage_window(c, l, idx0, a.W, lane);
        seed_trial_indices(l, N, SLOTS, lane);
        J_cur = sc_[0];
        J_init = sc_[1];
        r_lamb = sc_[2]; r_status = (int)sc_[3]; r_iters = (int)sc_[4]; r_trials = (int)sc_[5]; r_evals = (int)sc_[6];
        r_tl = (int)sc_[7]; r_flag = (int)sc_[8]; r_deep = (int)sc_[9]; r_seq = (int)sc_[11];
        wave_sync();
    } else {
        double xs[4];
        if (LOOP) { // closed loop on the device: the ego state is advanced inside this launch — not through the
                                // read-only kernel argument, whose loads the compiler may hoist out of the tick loop
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[e] = park_ld(a.loop_x0 + 4 * (size_t)b + e);
        } else {
            xs[0] = x0[4 * b]; xs[1] = x0[4 * b + 1]; xs[2] = x0[4 * b + 2]; xs[3] = x0[4 * b + 3];
        }
        init_trajectory(c, l, xs, last_u ? last_u + (size_t)b * N * 2 : nullptr, lane, idx0, a.W);
        seed_trial_indices(l, N, SLOTS, lane);
        J_cur = total_cost_lds<ALM>(c, l, al, lane);
        J_init = J_cur;
    }
    PROF_ADD(PH_INIT);
    if (HELP) {
        if (lane == 0) { l.ctli[CTL_IDX0] = idx0; l.ctli[CTL_W0] = l.w0; l.ctli[CTL_W] = l.W; }
        __syncthreads(); // B0
    }

    double lamb = c.k->init_lamb;
    int status = CILQR_RUNNING;
    int iters = 0, ls_trials = 0, cost_evals = 1, tl = 0;
    int end_reason = CILQR_END_MAX_ITER;
    int flag = 0;
    int n_fallback = 0;
    int itr0 = 0;              // resumable solves: the iteration this slice starts at
    bool fresh_expansion = false; // ... and its first iteration expands afresh whatever the status says (see below)
    // Line-search rollouts.  88 % / 71 % of the iterations of BASELINE configs 2 / 5 accept the first trial
    // (profiles/r02_trial_depth_histogram.json), 6-14 % try all 20 and the rest is spread evenly between.  An
    // iteration therefore either rolls out alpha = 1 alone into the small first-trial buffer and only on
    // rejection all step sizes into the slab ("shallow"), or all of them at once ("deep") when the previous
    // iteration's search went beyond its first trial — failed searches come in runs.  One extra rollout pass on
    // ~2 % of the iterations buys slab writes on 13-30 % of them instead of all.
    bool deep_next = false;
    ShareReq* const rq = share ? a.sh_req + b : nullptr;
    unsigned sh_seq = 1; // the searches of this trajectory that were announced, counted (a closed request holds the next number)
    if (RES && resumed) {
        lamb = r_lamb;
        status = uniform_int(r_status);
        iters = uniform_int(r_iters);
        ls_trials = uniform_int(r_trials);
        cost_evals = uniform_int(r_evals);
        tl = uniform_int(r_tl);
        flag = uniform_int(r_flag);
        deep_next = uniform_int(r_deep) != 0;
        sh_seq = (unsigned)uniform_int(r_seq);
        itr0 = iters;
        fresh_expansion = true; // the expansion a failed pass keeps (cs:469-475) stayed behind with the block that parked
    }
    bool parked = false;
    for (int itr = itr0; itr < c.max_iter; ++itr) {
        // are there idle blocks?  (asked here, needed after the backward sweep: the answer's latency is hidden)
        unsigned sh_probe = 0;
        if (share && lane == 0) sh_probe = sh_ld(a.sh_ctl + SH_HELPING);
        // ---- iter_step ----
        cost_evals += 1; // ori_cost (cs:342) — equals J_cur bit for bit, not recomputed (barrier mode)
        if (ALM) J_cur = total_cost_lds<ALM>(c, l, al, lane); // the multipliers may have moved since
        // cs:469-475: in barrier mode the expansion of the unchanged trajectory is kept after a failed pass
        const bool expand = ALM || status == CILQR_RUNNING || status == CILQR_FORWARD_PASS_SMALL_STEP || (RES && fresh_expansion);
        double dV[2];
        bool ok;
        if (OOL && CILQR_OOL_SWEEP) {
            ok = ool_expand_sweep<NCH, NC, LG>(g_lds, N, a.W, l.gl, l.w0, l.W, lamb, lane, expand ? 1 : 0);
            dV[0] = l.ctld[CTLD_DV];
            dV[1] = l.ctld[CTLD_DV + 1];
            status = CILQR_RUNNING;
            if (RES) fresh_expansion = false;
        } else {
        if (expand) {
            // (resumed after a failed pass: the same trajectory expanded again — the same bits as the kept expansion)
            cost_and_model_derivatives<ALM, LG>(c, l, al, lane);
        } else {
            model_jacobians(c, l, lane); // the gains of the failed pass sit where A, B were
        }
        PROF_ADD(PH_DERIV);
        status = CILQR_RUNNING;
        if (RES) fresh_expansion = false;
        ok = backward_sweep<(DBG && !ALM), LG ? (ALM ? CILQR_GL_ROW_ALM : CILQR_GL_ROW) : 0>(c, l, lamb, lane, dV, a.flags);
        wave_sync();
        PROF_ADD(PH_BACKWARD);
        }
        double new_J = J_cur;
        int trials = 0, alpha_idx = -1;
        if (!ok) {
            status = CILQR_BACKWARD_PASS_FAIL;
            if (HELP) {
                if (lane == 0) l.ctli[CTL_MODE] = 0;
                __syncthreads(); // B1
            }
        } else {
            flag = 0;
            const bool deep = (a.tier == 0) || (a.tier < 0 && deep_next);
            bool have_all = false; // the slab holds all 20 trial trajectories of this iteration
            bool done = false;
            int t0 = 0;
            unsigned sh_st = 0;    // work sharing: state of this search's announcement (SH_ST_*), 0 = not announced
            bool sh_local = false; // cost the trial here although a helper delivered it
            int par = 0; // helper mode: which pair of cost slots this pass uses
            // The line search of cs:354-372.  The costs are produced pass by pass — alpha = 1 alone (usually
            // accepted), then NTP trials per pass (two with a helper wavefront) — and consumed strictly in order.
            // One call site each for the rollout, the costing and the acceptance keeps the loop body small.
            while (t0 < CILQR_MAX_ALPHA_TRIALS && !done) {
                if (t0 == 0 || !have_all) {
                    // t0 == 0: the iteration's first rollout pass; t0 == 1 without the slab: the first trial
                    // was rejected, now the other step sizes (lane 0 repeats the first trial: the same bits)
                    const bool all = deep || t0 == 1;
                    if (t0 == 1) restore_gains_head(l, first, N, lane);
                    rollout_trials<(!HELP && WPS == 2 && NCH == 1) ? 0 : DM_PIN>(c, l, all ? scr : first, lane, all ? CILQR_MAX_ALPHA_TRIALS : 1,
                                                                                      all ? CILQR_MAX_ALPHA_TRIALS : 1);
                    have_all = all;
                    PROF_ADD(PH_ROLLOUT);
                    if (PROF && a.prof && lane == 0) ph_acc[t0 == 1 ? PH_ROLL_SECOND : (all ? PH_ROLL_ALL : PH_ROLL_FIRST)] += 1;
                    if (HELP) {
                        if (t0 == 0 && lane == 0) {
                            l.ctli[CTL_MODE] = all ? 1 : 2;
                            l.ctld[CTLD_RHO] = al.rho; l.ctld[CTLD_JCUR] = J_cur; l.ctld[CTLD_DV] = dV[0]; l.ctld[CTLD_DV + 1] = dV[1];
                        }
                        __syncthreads(); // B1 (first pass) / B3 (second pass)
                    }
                    // the stage-cost scratch lies over the gains of the first steps: a shallow iteration may
                    // still need them for its second pass
                    if (!all) save_gains_head(l, first, N, lane);
                }
                const double* src = have_all ? scr : first;
                const int as = have_all ? CILQR_MAX_ALPHA_TRIALS : 1;
                double Jp[CILQR_NT];
                int nt = (t0 == 0) ? 1 : NTP;
                if (HELP && have_all) nt = 2; // this wave costs trial t0 (slot 0), the helper trial t0 + 1 (slot 1)
                if (t0 + nt > CILQR_MAX_ALPHA_TRIALS) nt = CILQR_MAX_ALPHA_TRIALS - t0;
                bool foreign = false; // the cost of trial t0 comes from another block
                if (SHARE && share && have_all && t0 >= a.sh_min_t0 && !sh_local &&
                    (sh_st != 0u || (CILQR_MAX_ALPHA_TRIALS - t0 >= CILQR_SH_MIN_OPEN &&
                                     __builtin_amdgcn_readfirstlane((int)sh_probe) != 0))) {
                    sh_st = sh_owner_step(a.sh_ctl, rq, a.sh_hints + (size_t)b * (N + 2), l.ridx, &l.ctld[CTLD_JM], b, slot, N, t0,
                                          idx0, ALM ? dm_to_bits(al.rho) : 0ULL, sh_seq, sh_st, lane);
                    foreign = (sh_st & SH_ST_FOREIGN) != 0u;
                    if (foreign) Jp[0] = l.ctld[CTLD_JM];
                }
                if (SHARE && foreign) {
                    // nothing to compute
                } else if (OOL && CILQR_OOL_COST) {
                    Jp[0] = ool_cost_trial<NCH, NC, LG>(g_lds, N, a.W, l.w0, l.W, src, t0, as, idx0, lane);
                } else if (HELP || nt == 1) {
                    double J1[1];
                    total_cost_trials<DBG, NCH, ALM, 1>(c, l, al, src, t0, 1, lane, idx0, a.flags, &n_fallback, J1,
                                                        (PROF && a.prof) ? &ph_acc[PH_TC_REF] : nullptr, 0, as);
                    Jp[0] = J1[0];
                    if (HELP) {
                        if (lane == 0) l.ctld[CTLD_JM + par] = J1[0];
                        __syncthreads(); // B2 (B2s for the lone first trial of a shallow iteration)
                        if (have_all) {
                            Jp[1] = l.ctld[CTLD_JH + par];
                            par ^= 1;
                        }
                    }
                } else if (NTP > 1) {
                    total_cost_trials<DBG, NCH, ALM, CILQR_NT>(c, l, al, src, t0, nt, lane, idx0, a.flags, &n_fallback,
                                                               Jp, (PROF && a.prof) ? &ph_acc[PH_TC_REF] : nullptr);
                }
                PROF_ADD(PH_TRIAL_COST);
                if (SHARE && foreign && trial_verdict(J_cur, Jp[0], t0, dV[0], dV[1], c.k->conv_thr, c.k->accept_thr) == 2) {
                    // the accepted trial's lane indices are needed too: cost it here (the same bits) and accept then
                    sh_local = true;
                    continue;
                }
                sh_local = false;
                for (int tt = 0; tt < nt && !done; ++tt) {
                    const int t = t0 + tt;
                    new_J = Jp[tt];
                    trials++;
                    const int verdict = trial_verdict(J_cur, new_J, t, dV[0], dV[1], c.k->conv_thr, c.k->accept_thr);
                    if (verdict == 1) {
                        status = CILQR_CONVERGED;
                        alpha_idx = t;
                        done = true;
                    } else if (verdict == 2) {
                        if (t != 0) status = CILQR_FORWARD_PASS_SMALL_STEP;
                        flag = 1;
                        alpha_idx = t;
                        accept_trial(c, l, src, t, tt, lane, as);
                        PROF_ADD(PH_ACCEPT);
                        J_cur = new_J;
                        done = true;
                    }
                }
                t0 += nt;
            }
            if (SHARE && sh_st != 0u) { // (before anything touches the slab again)
                sh_owner_close(a.sh_ctl, rq, b, sh_seq, sh_st, (t0 > 0 ? t0 - 1 : 0), lane);
                sh_seq++;
            }
            deep_next = (trials > 1);
            if (!done) {
                status = CILQR_FORWARD_PASS_FAIL;
                if (ALM) { // cs:377-378
                    for (int e = lane; e < N * al.C; e += CILQR_WAVE) al.mu[e] = al.mu_next[e];
                    double r = (1 + c.k->alm_gamma) * al.rho;
                    al.rho = (c.k->max_rho < r) ? c.k->max_rho : r;
                    wave_sync();
                }
            }
        }
        // ---- back in solve (cs:113-141) ----
        iters++;
        ls_trials += trials;
        cost_evals += trials;
        if (status == CILQR_BACKWARD_PASS_FAIL || status == CILQR_FORWARD_PASS_FAIL) {
            double la = lamb * c.k->lamb_amplify;
            lamb = (c.k->lamb_amplify < la) ? la : c.k->lamb_amplify;
        } else if (status == CILQR_RUNNING) {
            lamb *= c.k->lamb_decay;
        }
        if (trace_out && tl < trace_cap && lane == 0) {
            cilqr_trace_rec r;
            r.status = status; r.trials = trials; r.accepted = flag; r.alpha_idx = alpha_idx;
            r.lamb = lamb; r.new_J = new_J;
            trace_out[(size_t)b * trace_cap + tl] = r;
        }
        tl++;
        bool leave = false;
        if (lamb > c.k->max_lamb) { end_reason = CILQR_END_MAX_LAMB; leave = true; }
        else if (status == CILQR_CONVERGED) { end_reason = CILQR_END_CONVERGED; leave = true; }
        if (HELP) {
            if (lane == 0) l.ctli[CTL_EXIT] = (leave || itr + 1 >= c.max_iter) ? 1 : 0;
            __syncthreads(); // B4
        }
        if (leave) break;
        if (RES && res_on && itr + 1 - itr0 >= a.res_iters && itr + 1 < c.max_iter) {
            // the slice is over: park, unless nothing else wants this block (then carry on for another slice)
            if (sh_ld_u(a.next, lane) < (unsigned)a.B || rq_nonempty(a.ctl, a.rq, (unsigned)a.rq_cap, lane)) { parked = true; break; }
            itr0 = itr + 1;
        }
    }
    if (RES && parked) {
        // park: what cs:110-141 carries from one iteration to the next
        double* const sc_ = l.xch;
        if (lane == 0) {
            sc_[0] = J_cur; sc_[1] = J_init; sc_[2] = lamb; sc_[3] = (double)status; sc_[4] = (double)iters;
            sc_[5] = (double)ls_trials; sc_[6] = (double)cost_evals; sc_[7] = (double)tl; sc_[8] = (double)flag;
            sc_[9] = deep_next ? 1.0 : 0.0; sc_[10] = (double)idx0; sc_[11] = (double)sh_seq;
        }
        wave_sync();
        park_copy(pk, l.x, l.u, l.ridx, sc_, N, lane, 1);
        if (a.timeline && lane == 0) { // resumable solves: first start, and in [2] minus the busy time so far
            if (!resumed) a.timeline[4 * (size_t)b] = tl_start;
            (void)__hip_atomic_fetch_add(a.timeline + 4 * (size_t)b + 2, tl_start - (long long)__builtin_amdgcn_s_memrealtime(),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (slices run on different XCDs)
        }
        rq_push(a.ctl, a.rq, (unsigned)a.rq_cap, (unsigned)b, lane);
        return SOLVE_PARKED;
    }
    if (ALM) {
        J_cur = total_cost_lds<ALM>(c, l, al, lane); // J_final := get_total_cost(u_ret, x_ret) with the final multipliers
        if (lane == 0) a.alm_rho[b] = al.rho;
    }
    // results: u, x of the last accepted trajectory
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        double* xo = x_out + ((size_t)b * (N + 1) + k) * 4;
        xo[0] = l.x[4 * k]; xo[1] = l.x[4 * k + 1]; xo[2] = l.x[4 * k + 2]; xo[3] = l.x[4 * k + 3];
        if (k < N) {
            double* uo = u_out + ((size_t)b * N + k) * 2;
            uo[0] = l.u[2 * k]; uo[1] = l.u[2 * k + 1];
        }
    }
    if (PROF && a.prof && lane == 0) {
        ph_acc[PH_TOTAL] = (long long)__builtin_readcyclecounter() - t_begin;
        ph_acc[PH_ITERS] = iters;
        ph_acc[PH_REF_FALLBACKS] = n_fallback;
        ph_acc[PH_TRIALS] = ls_trials;
        for (int e = 0; e < CILQR_PROF_SLOTS; ++e) a.prof[(size_t)b * CILQR_PROF_SLOTS + e] = ph_acc[e];
    }
    if (lane == 0 && res_out) {
        cilqr_result r;
        r.J_init = J_init; r.J_final = J_cur; r.iters = iters; r.end_reason = end_reason;
        r.final_status = status; r.ls_trials = ls_trials; r.cost_evals = cost_evals;
        r.trace_len = (trace_out && tl > trace_cap) ? trace_cap : tl;
        res_out[b] = r;
    }
    if (a.timeline && lane == 0) {
        long long* tl_rec = a.timeline + 4 * (size_t)b;
        if (!(RES && resumed)) tl_rec[0] = tl_start;  // (a resumed solve keeps the start of its first slice)
        tl_rec[1] = (long long)__builtin_amdgcn_s_memrealtime();
        if (RES && res_on) // (a sliced solve has no one block: minus its busy time instead)
            (void)__hip_atomic_fetch_add(tl_rec + 2, tl_start - tl_rec[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else tl_rec[2] = blockIdx.x;
        tl_rec[3] = __builtin_amdgcn_s_getreg((20 /* XCC_ID */) | (0 << 6) | (3 << 11)) & 0xf; // hwreg(HW_REG_XCC_ID, 0, 4)
    }
    if ((SHARE && share) || (RES && res_on)) (void)sh_add_u(a.ctl + SH_FINISHED, 1u, lane);
    return iters;
}

// The solve kernel.  Small batches: one block per trajectory (in the XCD-aware order above).  Large batches (a.next
// set): PERSISTENT blocks — as many as the chip holds at once — that pull trajectories from a counter until none is
// left.  The hardware's workgroup dispatcher hands blocks out in order, round-robin over the XCDs, and waits whenever
// the XCD whose turn it is has no room: with solves of 1 to 8 ms a freed slot stood empty for 157 us (median; 7 % of
// all slot time) before the next block began; pulling, the next solve begins 3 us after the last (measured with
// cilqr_set_block_timeline, scripts/block_timeline.py: config 5 82.4 -> 75.8 ms).  Pulling also evens out the XCDs,
// and a launch touches one scratch area per resident block instead of one per trajectory.  A block that finds no
// trajectory left turns to the line searches of the blocks still running (SHARE builds).
template <bool DBG, int NCH, bool ALM, bool HELP, bool PROF, int WPS = 1, int NTP = CILQR_NT, int NC = 0, bool LG = false,
          bool SHARE = false, bool RES = false, bool LOOP = false>
__global__ void __launch_bounds__(HELP ? 2 * CILQR_WAVE : CILQR_WAVE, HELP ? 2 : WPS)
k_solve(BatchArgs a, const double* __restrict__ x0, const double* last_u,
        double* u_out, double* __restrict__ x_out, cilqr_result* __restrict__ res_out,
        cilqr_trace_rec* __restrict__ trace_out, int trace_cap) {
    const int lane = threadIdx.x & (CILQR_WAVE - 1);
    const bool persistent = !HELP && a.next != nullptr;
    if (!persistent && (int)blockIdx.x >= a.B) return;
    const bool res_on = RES && persistent && a.park != nullptr;
    bool fresh_left = true;
    const int T = LOOP ? a.loop_ticks : 1; // closed loop on the device: ticks per ego
    int t_done = 0;                                    // ... ticks of the current ego that are done
    unsigned b_cur = 0;
    for (;;) { // (one call site: a second inlined copy of the solve costs the loop ~70 spilled vector registers; out of
               //  line, with the arguments on the stack, a solve takes 7 % longer)
        unsigned b = (unsigned)a.B;
        bool resumed = false;
        if (LOOP && t_done > 0) b = b_cur; // the next tick of the ego this block is driving
        else if (!persistent) b = (unsigned)trajectory_of_block(blockIdx.x, a.B);
        else if (fresh_left) b = sh_add_u(a.next, 1u, lane);
        if (b >= (unsigned)a.B) {
            if (!res_on) break;
            // resumable solves: no fresh trajectory left — a parked one, else this block is done.  (No solve is stranded:
            // a block only gets here after FINISHING a solve, so every unfinished solve is either running on a block that
            // is still in this loop or queued; and a block that parks a solve comes straight back for one.)
            fresh_left = false;
            const int pb = rq_pop(a.ctl, a.rq, (unsigned)a.rq_cap, lane);
            if (pb < 0) break;
            b = (unsigned)pb;
            resumed = true;
        }
        const double* lu = last_u;
        if (LOOP && t_done > 0) {
            // a later tick of the loop: warm from the plan just stored if the ego's configuration says so, cold (and, under
            // the augmented Lagrangian, with fresh multipliers) otherwise — cs:88-101.  (The ids were valid a tick ago.)
            const int pid = a.param_id ? a.param_id[b] : 0;
            lu = a.params[pid].use_last_solution ? u_out : nullptr;
        }
        const int it = solve_one<DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP>(
            a, (int)b, persistent ? (int)blockIdx.x : (int)b, resumed, x0, lu, u_out, x_out, res_out, trace_out, trace_cap);
        if (LOOP) {
            // the step after the path (mp:181,197), for this ego: ego_state = new_x.row(1), the obstacle window one tick on;
            // the next solve starts warm from the plan just stored (cs:163-180: d_last_u = d_u_out) if use_last_solution
            if (HELP) __syncthreads();
            const int N = NC ? NC : a.N;
            if ((!HELP || threadIdx.x < CILQR_WAVE) && it >= 0) {
                Lds l;
                carve(l, g_lds, N, a.W, ALM ? 1 : 0, (HELP || NTP == 2) ? 2 : 1, LG ? 1 : 0);
                if (lane < 4) {
                    const double v = l.x[4 + lane]; // row 1 of the plan, still in LDS
                    a.loop_x0[4 * (size_t)b + lane] = v;
                    if (a.loop_states) a.loop_states[((size_t)b * T + t_done) * 4 + lane] = v;
                }
                if (lane == 0) {
                    a.loop_tick[b] += 1;
                    if (a.loop_iters) a.loop_iters[(size_t)t_done * a.B + b] = it;
                }
            }
            // This block's own stores are in L2 before it reads x0 / tick / u again, and not shadowed by its CU's L1: stores
            // complete, THEN (helper mode: both wavefronts past this point, THEN) the L1 is invalidated.  In that order: a
            // neighbouring ego's block on the same CU may re-fill the shared line of tick[] / x0[] at any time, and a fill
            // that predates the store must not survive the invalidate.
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (HELP) __syncthreads();
            sh_acquire();
            __builtin_amdgcn_s_dcache_inv(); // (tick[b] is wave-uniform: the scalar cache may hold it)
            t_done = (it >= 0 && t_done + 1 < T) ? t_done + 1 : 0; // (an ego whose routes have run out stops there)
            b_cur = b;
            if (t_done > 0) continue;
        }
        if (!persistent) break;
    }
    if (SHARE && persistent && a.sh_ctl != nullptr) {
        const int N = NC ? NC : a.N;
        Lds l;
        carve(l, g_lds, N, a.W, ALM ? 1 : 0, (HELP || NTP == 2) ? 2 : 1, LG ? 1 : 0);
        share_help<NCH, NC, ALM>(a, l, N, lane, (int)blockIdx.x);
    }
}


// ------------------------------------------------------------------------------------------------
// k_solve_grp: the solve for large batches of horizons up to 63, barrier mode — G trajectories per wavefront, one rollout
// pass for all of them (cilqr_group.hpp).  Persistent blocks pulling trajectories from a.next, like k_solve's.
// Every trajectory runs CILQRSolver::solve (cs:85-153) + iter_step (cs:337-381) through the same device functions as in
// solve_one, cut into segments at the two points where its line search needs a rollout pass:
//   GP_ITER   -> expansion, backward sweep; gains to global memory; ask for the first trial alone (or all 20 step sizes
//                when the previous search went deep) and yield                                   [phase GP_SEARCH, t0 = 0]
//   GP_SEARCH -> cost trials t0, t0 + 1, ... in order; first trial rejected and only it rolled out: ask for all 20 and
//                yield [t0 = 1]; verdict reached or all 20 rejected: back in solve (cs:113-141), then GP_ITER or done
// A trajectory that ends is replaced at once (pull, initial trajectory, first expansion) inside the same segment.
// LOOP: the closed planning loop in one launch (cilqr_closed_loop_batch_device; mp:180-197, cs:163-180): a trajectory slot keeps
// its ego for all its ticks — solve, ego <- x.row(1), tick + 1, next solve warm from the plan just stored where the ego's
// configuration says so — as k_solve's LOOP builds do, but two egos per wavefront share their rollout passes.
template <int NC, int G, bool LOOP = false, int NCH = 1, bool LAY = (NCH > 1), bool ALM = false>
__global__ void __launch_bounds__(CILQR_WAVE, 2)
k_solve_grp(BatchArgs a, const double* __restrict__ x0, const double* last_u, double* u_out, double* __restrict__ x_out,
            cilqr_result* __restrict__ res_out, cilqr_trace_rec* __restrict__ trace_out, int trace_cap) {
    // NCH = 2: horizons of 64 ... 127, two rows per lane — the LONG layout of cilqr_group.hpp (both expansions and the gains in
    // global memory, streamed; trial costs one at a time)
    // LAY: the long layout for any horizon (NCH = 1: one row per lane) — what the augmented Lagrangian (ALM) runs in: its dense
    // l_xx never fits LDS twice, streamed rows take it as they are; the multipliers stay in HBM, rho in GrpSt::J_pair; no
    // hand-overs between wavefronts (the multipliers are written with plain stores): the host passes a.park = nullptr
    constexpr bool LONG = LAY;
    static_assert(!(ALM && LOOP), "the closed loop in one launch has no augmented-Lagrangian grouped build");
    static_assert(!ALM || LONG, "augmented Lagrangian in pairs: the long layout");
    const int lane = threadIdx.x & (CILQR_WAVE - 1);
    const int N = NC ? NC : a.N;
    double* const scr_blk = a.scratch + (size_t)blockIdx.x * G * grp_scratch_doubles(N);
    auto alm_of = [&](int bb) {
        GrpAlm ga{nullptr, nullptr, 0};
        if (ALM) {
            ga.mu = a.alm_mu + (size_t)bb * N * a.alm_C;
            ga.mu_next = a.alm_mu_next + (size_t)bb * N * a.alm_C;
            ga.C = a.alm_C;
        }
        return ga;
    };
    if (lane < G) {
        GrpSt* st = grp_state(g_lds, N, lane);
        st->phase = GP_EMPTY;
        st->req = 0;
    }
    wave_sync();
    bool fresh_left = true;
    const int T = LOOP ? a.loop_ticks : 1;
    const bool steal = a.park != nullptr; // the tail of the launch: idle wavefronts take over trajectories of wavefronts that hold two
    const bool slice = steal && a.res_iters > 0; // ... and solves run a.res_iters iterations at a time (see the end of an iteration)
    AlmSt al;
    al.mu = nullptr; al.mu_next = nullptr; al.rho = 1.0; al.C = 0;
    // Round 5: a turn of the wavefront = (pass 0) every trajectory's segment up to the point where it needs a rollout pass OR
    // stands at the head of an iteration (GP_EXPAND); then the expansions of those — the first into the shared LDS arrays, the
    // second into its rows in global memory — and ONE backward sweep for both (grp_sweep: two trajectories in one instruction
    // stream); (pass 1, rare) the segments of trajectories whose sweep met a non-PD Q_uu, which carry on as in round 4 with
    // expansion + sweep in one call; then the rollout pass.  All searches of a turn come before all expansions because a
    // search's lane window and the first expansion share LDS.  CILQR_GRP_PAIR_SWEEP = 0: round 4's turn (A/B builds).
    for (;;) {
        int n_live = 0;
        for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            // ---- the heads of the iterations that are due: expansions, then the sweep(s) ----
            int gA = -1, gB = -1;
            for (int g = 0; g < G; ++g)
                if (uniform_int(grp_state(g_lds, N, g)->phase) == GP_EXPAND) { if (gA < 0) gA = g; else gB = g; }
            if (gA < 0) break;
            const bool prof2 = CILQR_GPROF && a.prof != nullptr;
            if (LONG)
                grp_expand<NC, G, true, LONG, ALM>(g_lds, gA, N, lane, scr_blk + (size_t)gA * grp_scratch_doubles(N) + grp_rows_offset(N),
                                                   prof2 ? grp_prof(g_lds, N, gA) : nullptr, alm_of(uniform_int(grp_state(g_lds, N, gA)->b)));
            else
                grp_expand<NC, G, false>(g_lds, gA, N, lane, nullptr, prof2 ? grp_prof(g_lds, N, gA) : nullptr);
            if (gB >= 0)
                grp_expand<NC, G, true, LONG, ALM>(g_lds, gB, N, lane, scr_blk + (size_t)gB * grp_scratch_doubles(N) + grp_rows_offset(N),
                                                   prof2 ? grp_prof(g_lds, N, gB) : nullptr, alm_of(uniform_int(grp_state(g_lds, N, gB)->b)));
            const int asked = uniform_int(grp_sweep<NC, G, LONG, ALM>(g_lds, gA, gB, N, lane, scr_blk, a.tier, prof2 ? grp_prof(g_lds, N, gA) : nullptr,
                                                                      (prof2 && gB >= 0) ? grp_prof(g_lds, N, gB) : nullptr));
            n_live += asked;
            if (asked == ((gB >= 0) ? 2 : 1)) break; // (nobody failed: no second pass)
        }
        for (int g = 0; g < G; ++g) {
            Lds l;
            carve_group_t<LONG>(l, g_lds, N, G, g);
            GrpSt* const st = grp_state(g_lds, N, g);
            int phase = uniform_int(st->phase);
            if (phase == GP_CLAIMED) { // the slot holds a place in the queue: has its entry arrived?
                if (pass != 0) continue;
                const int pb = rq_poll(a.rq, (unsigned)a.rq_cap, (unsigned)uniform_int(st->pad2), lane);
                if (pb < 0) continue;
                if (lane == 0) { st->b = pb; st->req = 0; }
                lds_sync();
                phase = GP_STOLEN;
            }
            if (phase == GP_DONE) {
                // (sliced solves: a slot that ran dry takes a parked trajectory as soon as one is queued)
                if (LOOP || !slice || pass != 0) continue;
                int claim = -1;
                const int pb = uniform_int(grp_take_parked(a.ctl, a.rq, (unsigned)a.rq_cap, lane, &claim));
                if (pb == -2) { if (lane == 0) { st->phase = GP_CLAIMED; st->pad2 = uniform_int(claim); } lds_sync(); }
                if (pb < 0) continue;
#ifdef CILQR_DEV_BUILD
                (void)sh_add_u(a.ctl + SH_HELPED, 1u, lane); // (slots revived)
#endif
                if (lane == 0) { st->b = pb; st->req = 0; }
                lds_sync();
                phase = GP_STOLEN;
            }
            if (pass == 1 && phase != GP_BPF) continue; // (second pass: only the trajectories whose sweep failed)
            const bool split = CILQR_GRP_PAIR_SWEEP && a.pair_sweep != 0 && pass == 0;
            double* const scr = scr_blk + (size_t)g * grp_scratch_doubles(N);
            double* const first = scr + slab_doubles(N);
            long long* const pacc = grp_prof(g_lds, N, g);
            const bool prof = CILQR_GPROF && a.prof != nullptr;
            long long t_ph = prof ? (long long)__builtin_readcyclecounter() : 0;
#define GPROF_ADD(ph)                                                        \
    do {                                                                     \
        if (prof) {                                                          \
            const long long t_now_ = (long long)__builtin_readcyclecounter(); \
            if (lane == 0) { pacc[ph] += t_now_ - t_ph; pacc[PH_TOTAL] += t_now_ - t_ph; } \
            t_ph = t_now_;                                                   \
        }                                                                    \
    } while (0)
            Cst c;
            int b = 0, idx0 = 0, status = CILQR_RUNNING, iters = 0, ls_trials = 0, cost_evals = 0, tl = 0, flag = 0, t0 = 0, trials = 0;
            bool deep_next = false, have_all = false;
            int keep_b = -1, t_done = 0; // closed loop: the slot's next trajectory is the same ego, its next tick
            int it0 = 0;                 // sliced solves: the iteration count at which this slice began
            bool win_ok = false;         // (ALM) this trajectory's lane window is staged in the shared area right now
            double J_cur = 0.0, J_init = 0.0, lamb = 0.0, new_J = 0.0, dV[2] = {0.0, 0.0};
            long long tl_start = 0;
            int entry = (phase == GP_SEARCH) ? 1 : (phase == GP_BPF ? 2 : 0); // where the segment takes the solve up again
            // is a wavefront waiting for work?  (asked here, needed when this trajectory is back in solve: the answer's
            // latency is hidden; only a wavefront that holds two trajectories will act on it)
            unsigned waiting_probe = 0;
            if (steal && lane == 0) waiting_probe = sh_ld(a.ctl + SH_HELPING);
            for (;;) { // (a slot that takes a parked trajectory off the queue in mid-segment comes round once more)
            if (phase == GP_STOLEN) {
                // a trajectory another wavefront parked between two iterations: its x, u, lane indices and scalars
                b = uniform_int(st->b);
                if (LOOP) { // its earlier ticks' plan (the warm start of the next tick) was written by another wavefront
                    sh_acquire();
                    __builtin_amdgcn_s_dcache_inv();
                }
                if (prof) {
                    for (int e = lane; e < CILQR_PROF_SLOTS; e += CILQR_WAVE) pacc[e] = 0; // (its cycles so far stay behind)
                    wave_sync();
                }
                grp_park_copy(a.park + (size_t)b * grp_park_doubles(N), l.x, l.u, l.ridx, st, N, lane, 0);
                idx0 = uniform_int(st->idx0);
                status = uniform_int(st->status); iters = uniform_int(st->iters); ls_trials = uniform_int(st->ls_trials);
                cost_evals = uniform_int(st->cost_evals); tl = uniform_int(st->tl); flag = uniform_int(st->flag);
                deep_next = uniform_int(st->deep_next) != 0;
                J_cur = st->J_cur; J_init = st->J_init; lamb = st->lamb;
                tl_start = st->tl_start;
                t_done = LOOP ? uniform_int(st->t_done) : 0;
                it0 = iters;
                load_cst<LOOP>(c, a, b, l, lane); // (fills this slot's copy of the cost model's constants)
                if (NC) c.N = NC;
                if (lane == 0) { *grp_cst(g_lds, N, g) = c; st->b = b; st->req = 0; }
                seed_trial_indices(l, N, 2, lane);
                phase = GP_ITER;
                GPROF_ADD(PH_TC_REF);
            } else if (phase != GP_EMPTY) {
                b = uniform_int(st->b);
                idx0 = uniform_int(st->idx0);
                status = uniform_int(st->status); iters = uniform_int(st->iters); ls_trials = uniform_int(st->ls_trials);
                cost_evals = uniform_int(st->cost_evals); tl = uniform_int(st->tl); flag = uniform_int(st->flag);
                t0 = uniform_int(st->t0); trials = uniform_int(st->trials);
                deep_next = uniform_int(st->deep_next) != 0; have_all = uniform_int(st->have_all) != 0;
                J_cur = st->J_cur; J_init = st->J_init; lamb = st->lamb; new_J = st->new_J; dV[0] = st->dV0; dV[1] = st->dV1;
                tl_start = st->tl_start;
                t_done = LOOP ? uniform_int(st->t_done) : 0;
                it0 = uniform_int(st->pad1);
                load_cst_lds(c, grp_cst(g_lds, N, g));
                if (entry == 1) { stage_window_fast(c, l, idx0, a.W, lane); win_ok = true; } // (the window area belongs to whoever's segment it is)
                GPROF_ADD(PH_TC_REF); // (grouped build: slot 10 = the segment's set-up — state, constants, lane window)
            }
            for (;;) {
                int alpha_idx = -1;
                if (entry == 2) {
                    // the sweep of this trajectory met a non-PD Q_uu (cs:345-347): back in solve with what iter_step hands back
                    entry = 0;
                    status = CILQR_BACKWARD_PASS_FAIL;
                    new_J = J_cur;
                    trials = 0;
                } else if (entry == 0) {
                    if (phase == GP_EMPTY) {
                        unsigned nb = (unsigned)a.B;
                        if (LOOP && keep_b >= 0) { nb = (unsigned)keep_b; keep_b = -1; } // the next tick of the ego this slot is driving
                        else {
                            t_done = 0;
                            if (fresh_left) nb = sh_add_u(a.next, 1u, lane);
                        }
                        if (nb >= (unsigned)a.B) {
                            fresh_left = false;
                            phase = GP_DONE;
                            if (!LOOP && slice) { // sliced solves: no fresh trajectory left — a parked one, if any
                                int claim = -1;
                                const int pb = uniform_int(grp_take_parked(a.ctl, a.rq, (unsigned)a.rq_cap, lane, &claim));
                                if (pb >= 0) {
                                    if (lane == 0) { st->b = pb; st->req = 0; }
                                    lds_sync();
                                    phase = GP_STOLEN;
                                } else if (pb == -2) {
                                    if (lane == 0) st->pad2 = uniform_int(claim);
                                    phase = GP_CLAIMED;
                                }
                            }
                            break;
                        }
                        b = (int)nb;
                        if (prof) {
                            for (int e = lane; e < CILQR_PROF_SLOTS; e += CILQR_WAVE) pacc[e] = 0;
                            wave_sync();
                            t_ph = (long long)__builtin_readcyclecounter();
                        }
                        tl_start = a.timeline ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
                        if (!ids_valid<LOOP>(a, b)) { // as in solve_one: NaN outputs, CILQR_END_BAD_INPUT; the slot takes the next trajectory
                            const double qnan = dm_from_bits(0x7ff8000000000000ULL);
                            for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) x_out[(size_t)b * 4 * (N + 1) + e] = qnan;
                            for (int e = lane; e < 2 * N; e += CILQR_WAVE) u_out[(size_t)b * 2 * N + e] = qnan;
                            if (lane == 0 && res_out) {
                                cilqr_result r;
                                r.J_init = qnan; r.J_final = qnan; r.iters = 0; r.end_reason = CILQR_END_BAD_INPUT;
                                r.final_status = CILQR_RUNNING; r.ls_trials = 0; r.cost_evals = 0; r.trace_len = 0;
                                res_out[b] = r;
                            }
                            if (steal) (void)sh_add_u(a.ctl + SH_FINISHED, 1u, lane);
                            continue; // (closed loop: an ego whose routes have run out stops there)
                        }
                        load_cst<LOOP>(c, a, b, l, lane);
                        if (NC) c.N = NC;
                        if (lane == 0) { st->dt = c.dt; st->wb = c.wb; st->rp = c.rp; st->nfb = 0; *grp_cst(g_lds, N, g) = c; }
                        lds_sync();
                        double xs0, xs1, xs2, xs3;
                        const double* lu = last_u;
                        if (LOOP) { // the ego state is advanced inside this launch: not through the read-only kernel argument
                            xs0 = park_ld(a.loop_x0 + 4 * (size_t)b); xs1 = park_ld(a.loop_x0 + 4 * (size_t)b + 1);
                            xs2 = park_ld(a.loop_x0 + 4 * (size_t)b + 2); xs3 = park_ld(a.loop_x0 + 4 * (size_t)b + 3);
                            if (t_done > 0) { // a later tick: warm from the plan just stored if the ego's configuration says so (cs:88-101)
                                const int pid = a.param_id ? a.param_id[b] : 0;
                                lu = a.params[pid].use_last_solution ? u_out : nullptr;
                            }
                        } else {
                            xs0 = x0[4 * b]; xs1 = x0[4 * b + 1]; xs2 = x0[4 * b + 2]; xs3 = x0[4 * b + 3];
                        }
                        if (ALM) { // cs:88-93: fresh multipliers and penalty weight unless this call continues a previous solution
                            const GrpAlm ga0 = alm_of(b);
                            double rho0 = a.alm_rho[b];
                            if (lu == nullptr) {
                                rho0 = c.k->alm_rho_init;
                                for (int e = lane; e < N * ga0.C; e += CILQR_WAVE) { const_cast<double*>(ga0.mu)[e] = 0.0; ga0.mu_next[e] = 0.0; }
                            }
                            if (lane == 0) st->J_pair = rho0;
                            wave_sync();
                        }
                        J_cur = grp_init<NC, G, LONG, ALM>(g_lds, g, N, lane, xs0, xs1, xs2, xs3, lu ? lu + (size_t)b * N * 2 : nullptr, a.W, alm_of(b));
                        win_ok = true;
                        idx0 = uniform_int(st->idx0);
                        if (ALM) { l.w0 = idx0; l.W = (c.L - idx0 < a.W) ? c.L - idx0 : a.W; } // (what init_trajectory staged)
                        J_init = J_cur;
                        lamb = c.k->init_lamb;
                        status = CILQR_RUNNING;
                        iters = 0; ls_trials = 0; cost_evals = 1; tl = 0; flag = 0;
                        it0 = 0;
                        deep_next = false;
                        phase = GP_ITER;
                        GPROF_ADD(PH_INIT);
                    }
                    if (iters < c.max_iter) {
                        // ---- iter_step ----
                        cost_evals += 1; // ori_cost (cs:342) — equals J_cur bit for bit, not recomputed
                        if (ALM) { // ... under the augmented Lagrangian it IS recomputed: the multipliers may have moved since
                            if (!win_ok) { stage_window_fast(c, l, idx0, a.W, lane); win_ok = true; }
                            J_cur = grp_recost_alm<NC, G>(g_lds, g, N, lane, l.w0, l.W, alm_of(b));
                        }
                        if (prof) { GPROF_ADD(PH_TC_STAGE); }
                        if (split) { phase = GP_EXPAND; break; } // its expansion and sweep run after every trajectory's segment
                        // (a trajectory on its own — second pass after a failed sweep, or round 4's turn: the same two functions, one
                        //  trajectory; they read lambda and leave the request in its block)
                        if (lane == 0) { st->lamb = lamb; st->deep_next = deep_next ? 1 : 0; st->J_cur = J_cur; }
                        lds_sync();
                        if (LONG) grp_expand<NC, G, true, LONG, ALM>(g_lds, g, N, lane, scr + grp_rows_offset(N), prof ? pacc : nullptr, alm_of(b));
                        else grp_expand<NC, G, false>(g_lds, g, N, lane, nullptr, prof ? pacc : nullptr);
                        win_ok = false; // (the sweep's rings have been over the window's area)
                        const bool ok = uniform_int(grp_sweep<NC, G, LONG, ALM>(g_lds, g, -1, N, lane, scr_blk, a.tier, prof ? pacc : nullptr, nullptr)) == 1;
                        if (prof) t_ph = (long long)__builtin_readcyclecounter(); // (booked inside)
                        status = CILQR_RUNNING;
                        dV[0] = st->dV0;
                        dV[1] = st->dV1;
                        new_J = J_cur;
                        trials = 0;
                        if (ok) {
                            flag = 0;
                            have_all = (a.tier == 0) || (a.tier < 0 && deep_next);
                            t0 = 0;
                            if (lane == 0) st->req = have_all ? 2 : 1;
                            phase = GP_SEARCH;
                            break; // until the pass has run
                        }
                        status = CILQR_BACKWARD_PASS_FAIL;
                    }
                } else {
                    entry = 0;
                    // ---- the line search of cs:354-372, from trial t0 on ----
                    bool done = false, again = false;
                    while (t0 < CILQR_MAX_ALPHA_TRIALS && !done) {
                        if (t0 == 1 && !have_all) { // the first trial was rejected and only it exists: all step sizes, next pass
                            have_all = true;
                            again = true;
                            break;
                        }
                        const double* src = have_all ? scr : first;
                        const int as = have_all ? CILQR_MAX_ALPHA_TRIALS : 1;
                        // past the first trial the costs come two per pass (the searches that get here mostly go on)
                        const int nt = (G > 1 && !LONG && a.pair_costs && have_all && t0 >= 1 && t0 + 1 < CILQR_MAX_ALPHA_TRIALS) ? 2 : 1; // (G = 1: three wavefronts per SIMD, 168 VGPRs — the paired form needs 208)
                        double Jp[2];
                        if (G > 1 && !LONG && nt == 2) {
                            Jp[0] = grp_cost_trials2<NC, (G > 1 ? G : 2)>(g_lds, g, N, lane, src, t0, l.w0, l.W);
                            Jp[1] = st->J_pair;
                        } else {
                            Jp[0] = grp_cost_trial<NC, G, NCH, LONG, ALM>(g_lds, g, N, lane, src, t0, as, l.w0, l.W, alm_of(b));
                            Jp[1] = 0.0;
                        }
                        GPROF_ADD(PH_TRIAL_COST);
                        for (int tt = 0; tt < nt && !done; ++tt) {
                            const int t = t0 + tt;
                            new_J = Jp[tt];
                            trials++;
                            const int verdict = trial_verdict(J_cur, new_J, t, dV[0], dV[1], c.k->conv_thr, c.k->accept_thr);
                            if (verdict == 1) {
                                status = CILQR_CONVERGED;
                                alpha_idx = t;
                                done = true;
                            } else if (verdict == 2) {
                                if (t != 0) status = CILQR_FORWARD_PASS_SMALL_STEP;
                                flag = 1;
                                alpha_idx = t;
                                accept_trial(c, l, src, t, tt, lane, as);
                                GPROF_ADD(PH_ACCEPT);
                                J_cur = new_J;
                                done = true;
                            }
                        }
                        t0 += nt;
                    }
                    if (again) {
                        if (lane == 0) st->req = 2;
                        break; // phase stays GP_SEARCH, t0 = 1
                    }
                    deep_next = (trials > 1);
                    if (!done) {
                        status = CILQR_FORWARD_PASS_FAIL;
                        if (ALM) { // cs:377-378
                            const GrpAlm ga1 = alm_of(b);
                            for (int e = lane; e < N * ga1.C; e += CILQR_WAVE) const_cast<double*>(ga1.mu)[e] = ga1.mu_next[e];
                            const double r = (1 + c.k->alm_gamma) * st->J_pair;
                            wave_sync();
                            if (lane == 0) st->J_pair = (c.k->max_rho < r) ? c.k->max_rho : r;
                            wave_sync();
                        }
                    }
                }
                bool leave = true; // (max_iter reached before the first iteration: nothing below applies)
                if (iters < c.max_iter) {
                    // ---- back in solve (cs:113-141) ----
                    iters++;
                    ls_trials += trials;
                    cost_evals += trials;
                    if (status == CILQR_BACKWARD_PASS_FAIL || status == CILQR_FORWARD_PASS_FAIL) {
                        double la = lamb * c.k->lamb_amplify;
                        lamb = (c.k->lamb_amplify < la) ? la : c.k->lamb_amplify;
                    } else if (status == CILQR_RUNNING) {
                        lamb *= c.k->lamb_decay;
                    }
                    if (trace_out && tl < trace_cap && lane == 0) {
                        cilqr_trace_rec r;
                        r.status = status; r.trials = trials; r.accepted = flag; r.alpha_idx = alpha_idx;
                        r.lamb = lamb; r.new_J = new_J;
                        trace_out[(size_t)b * trace_cap + tl] = r;
                    }
                    tl++;
                    leave = false;
                }
                int end_reason = CILQR_END_MAX_ITER;
                if (!leave) {
                    if (lamb > c.k->max_lamb) { end_reason = CILQR_END_MAX_LAMB; leave = true; }
                    else if (status == CILQR_CONVERGED) { end_reason = CILQR_END_CONVERGED; leave = true; }
                    else if (iters >= c.max_iter) leave = true;
                }
                GPROF_ADD(PH_TC_STAGE); // (slot 11 = back in solve: counters, lambda schedule, trace record)
                if (!leave) {
                    phase = GP_ITER;
                    if (steal && __builtin_amdgcn_readfirstlane((int)waiting_probe) != 0) {
                        // somebody is waiting for work and this wavefront holds another live trajectory: hand this one over
                        bool other = false;
                        for (int h = 0; h < G; ++h) {
                            const int ph = uniform_int(grp_state(g_lds, N, h)->phase);
                            if (h != g && (ph == GP_SEARCH || ph == GP_EXPAND)) other = true; // (live: waits for a pass, or for its sweep)
                        }
                        if (other && grp_queue_room(a.ctl, (unsigned)a.B, (unsigned)a.rq_cap, lane) && grp_take_ticket(a.ctl, lane)) {
                            if (lane == 0) {
                                st->b = b; st->idx0 = idx0; st->status = status; st->iters = iters; st->ls_trials = ls_trials;
                                st->cost_evals = cost_evals; st->tl = tl; st->flag = flag;
                                st->deep_next = deep_next ? 1 : 0;
                                st->J_cur = J_cur; st->J_init = J_init; st->lamb = lamb; st->tl_start = tl_start;
                                st->t_done = t_done;
                            }
                            lds_sync();
                            // Closed loop: this wavefront has stored the ego's plan, state and result at every tick it ran, with
                            // plain stores that sit dirty in ITS XCD's L2; whoever takes the ego over stores the same addresses
                            // from another XCD, and the two L2s write back in no particular order (found by the loop's own test:
                            // final states that were tick 10's).  Write this L2's lines back before the ego changes hands.
                            if (LOOP) sh_release();
                            grp_park_copy(a.park + (size_t)b * grp_park_doubles(N), l.x, l.u, l.ridx, st, N, lane, 1);
                            rq_push(a.ctl, a.rq, (unsigned)a.rq_cap, (unsigned)b, lane);
                            phase = GP_DONE; // (the counter is dry — or somebody would not be waiting: nothing to pull)
                            break;
                        }
                        waiting_probe = 0; // (asked once per segment)
                    }
                    if (!LOOP && slice && iters - it0 >= a.res_iters) {
                        // Sliced solves (as k_solve's resumable ones): the slice is over.  While fresh trajectories are left, or
                        // parked ones wait, this one goes to the back of the queue and the slot takes the next — every long solve
                        // is well under way when the short ones are done, instead of finishing alone at the end of the launch.
                        if (grp_slot_wanted(a.next, (unsigned)a.B, (unsigned)a.res_window, fresh_left ? 1 : 0, a.ctl, a.rq, (unsigned)a.rq_cap, lane)) {
                            if (lane == 0) {
                                st->b = b; st->idx0 = idx0; st->status = status; st->iters = iters; st->ls_trials = ls_trials;
                                st->cost_evals = cost_evals; st->tl = tl; st->flag = flag;
                                st->deep_next = deep_next ? 1 : 0;
                                st->J_cur = J_cur; st->J_init = J_init; st->lamb = lamb; st->tl_start = tl_start;
                                st->t_done = t_done;
                            }
                            lds_sync();
                            grp_park_copy(a.park + (size_t)b * grp_park_doubles(N), l.x, l.u, l.ridx, st, N, lane, 1);
                            rq_push(a.ctl, a.rq, (unsigned)a.rq_cap, (unsigned)b, lane);
                            (void)grp_take_ticket(a.ctl, lane); // (if a wavefront waits, this push is the one its ticket asked for)
                            phase = GP_EMPTY;
                        } else {
                            it0 = iters; // (nothing else wants the slot: another slice)
                        }
                    }
                    continue;
                }
                if (ALM) { // J_final := get_total_cost(u_ret, x_ret) with the final multipliers; the penalty weight back to the handle
                    if (!win_ok) { stage_window_fast(c, l, idx0, a.W, lane); win_ok = true; }
                    J_cur = grp_recost_alm<NC, G>(g_lds, g, N, lane, l.w0, l.W, alm_of(b));
                    if (lane == 0) a.alm_rho[b] = st->J_pair;
                }
                // results: u, x of the last accepted trajectory
                for (int k = lane; k <= N; k += CILQR_WAVE) {
                    double* xo = x_out + ((size_t)b * (N + 1) + k) * 4;
                    xo[0] = l.x[4 * k]; xo[1] = l.x[4 * k + 1]; xo[2] = l.x[4 * k + 2]; xo[3] = l.x[4 * k + 3];
                    if (k < N) {
                        double* uo = u_out + ((size_t)b * N + k) * 2;
                        uo[0] = l.u[2 * k]; uo[1] = l.u[2 * k + 1];
                    }
                }
                if (prof) {
                    GPROF_ADD(PH_ACCEPT);
                    if (lane == 0) {
                        pacc[PH_ITERS] = iters; pacc[PH_REF_FALLBACKS] = st->nfb; pacc[PH_TRIALS] = ls_trials;
                        for (int e = 0; e < CILQR_PROF_SLOTS; ++e) a.prof[(size_t)b * CILQR_PROF_SLOTS + e] = pacc[e];
                    }
                }
                if (lane == 0 && res_out) {
                    cilqr_result r;
                    r.J_init = J_init; r.J_final = J_cur; r.iters = iters; r.end_reason = end_reason;
                    r.final_status = status; r.ls_trials = ls_trials; r.cost_evals = cost_evals;
                    r.trace_len = (trace_out && tl > trace_cap) ? trace_cap : tl;
                    res_out[b] = r;
                }
                if (a.timeline && lane == 0) {
                    long long* tl_rec = a.timeline + 4 * (size_t)b;
                    tl_rec[0] = tl_start;
                    tl_rec[1] = (long long)__builtin_amdgcn_s_memrealtime();
                    tl_rec[2] = blockIdx.x;
                    tl_rec[3] = __builtin_amdgcn_s_getreg((20 /* XCC_ID */) | (0 << 6) | (3 << 11)) & 0xf;
                }
                if (LOOP) {
                    // the step after the path (mp:181,197), for this ego: ego_state = new_x.row(1), the obstacle window one tick on
                    if (lane < 4) {
                        const double v = l.x[4 + lane];
                        a.loop_x0[4 * (size_t)b + lane] = v;
                        if (a.loop_states) a.loop_states[((size_t)b * T + t_done) * 4 + lane] = v;
                    }
                    if (lane == 0) {
                        // (an agent-scope add: the ego may have come from another wavefront, whose increments this CU's L1 —
                        //  which may hold the line for a neighbouring ego's sake — has not seen)
                        (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned*>(a.loop_tick + b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (a.loop_iters) a.loop_iters[(size_t)t_done * a.B + b] = iters;
                    }
                    // this wavefront's stores (plan, ego state, tick) are in L2 before it — or whoever takes the ego over — reads
                    // them again, and not shadowed by the CU's L1 / scalar cache
                    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    sh_acquire();
                    __builtin_amdgcn_s_dcache_inv();
                    t_done += 1;
                }
                wave_sync(); // (x, u are read before the slot's next trajectory overwrites them)
                phase = GP_EMPTY;
                if (LOOP && t_done < T) { keep_b = b; continue; }
                if (steal) (void)sh_add_u(a.ctl + SH_FINISHED, 1u, lane);
            }
            if (phase != GP_STOLEN) break;
            } // (the slot's restart loop)
            // what the trajectory carries to its next segment
            if (lane == 0) {
                st->phase = phase;
                if (phase == GP_SEARCH || phase == GP_EXPAND) {
                    st->b = b; st->idx0 = idx0; st->status = status; st->iters = iters; st->ls_trials = ls_trials;
                    st->cost_evals = cost_evals; st->tl = tl; st->flag = flag; st->t0 = t0; st->trials = trials;
                    st->deep_next = deep_next ? 1 : 0; st->have_all = have_all ? 1 : 0;
                    st->J_cur = J_cur; st->J_init = J_init; st->lamb = lamb; st->new_J = new_J; st->dV0 = dV[0]; st->dV1 = dV[1];
                    st->tl_start = tl_start;
                    st->t_done = t_done;
                    st->pad1 = it0;
                }
            }
            lds_sync(); // (not wave_sync: the sweep's gain stores may still be in flight; rollout_group waits for them)
            GPROF_ADD(PH_TC_SUM); // (slot 12 = parking the state)
#undef GPROF_ADD
            if (phase == GP_SEARCH) n_live++;
        }
        }
        if (n_live == 0) {
            if (!steal) break;
            // a place in the queue one of the slots holds already, else (sliced solves) a parked trajectory if one is queued,
            // else a ticket and a place (grp_wait_for_work)
            int pb = -1, gw = 0;
            long long claim = -1;
            for (int g = G - 1; g >= 0; --g)
                if (uniform_int(grp_state(g_lds, N, g)->phase) == GP_CLAIMED) { gw = g; claim = (long long)(unsigned)uniform_int(grp_state(g_lds, N, g)->pad2); }
            if (claim < 0 && !LOOP && slice) {
                int c2 = -1;
                pb = uniform_int(grp_take_parked(a.ctl, a.rq, (unsigned)a.rq_cap, lane, &c2));
                if (pb == -2) claim = (long long)(unsigned)uniform_int(c2);
            }
            if (pb < 0) pb = grp_wait_for_work(a.ctl, a.rq, (unsigned)a.rq_cap, (unsigned)a.B, lane, claim);
#ifdef CILQR_DEV_BUILD
            if (pb < 0 && sh_ld_u(a.ctl + SH_FINISHED, lane) < (unsigned)a.B) (void)sh_add_u(a.ctl + SH_ANNOUNCED, 1u, lane); // (left before the end)
#endif
            if (pb < 0) break;
            if (lane == 0) { GrpSt* st0 = grp_state(g_lds, N, gw); st0->b = pb; st0->phase = GP_STOLEN; st0->req = 0; }
            lds_sync();
            continue;
        }
        // one rollout pass for every trajectory that asked (the polynomial coefficients as scalar-register literals: the
        // flavour the two-per-SIMD builds of k_solve measured best with)
        if (CILQR_GPROF && a.prof) {
            // the pass's cycles, shared out among the trajectories it served
            int reqs[G], n_req = 0;
            for (int g = 0; g < G; ++g) { reqs[g] = uniform_int(grp_state(g_lds, N, g)->req); n_req += reqs[g] != 0; }
            const long long t0_ = (long long)__builtin_readcyclecounter();
            if (LONG) rollout_group_long<G, 0>(g_lds, scr_blk, N, lane);
            else rollout_group<G, 0, (G <= 2)>(g_lds, scr_blk, N, lane);
            const long long dt_ = ((long long)__builtin_readcyclecounter() - t0_) / (n_req > 0 ? n_req : 1);
            if (lane == 0)
                for (int g = 0; g < G; ++g)
                    if (reqs[g]) {
                        long long* pa = grp_prof(g_lds, N, g);
                        pa[PH_ROLLOUT] += dt_; pa[PH_TOTAL] += dt_;
                        pa[PH_TC_SAMPLED] += grp_state(g_lds, N, g)->small_steps; // (slot 13: straight-line steps of its passes)
                        pa[reqs[g] == 1 ? PH_ROLL_FIRST : (uniform_int(grp_state(g_lds, N, g)->t0) == 1 ? PH_ROLL_SECOND : PH_ROLL_ALL)] += 1;
                    }
            wave_sync();
        } else if (LONG) {
            rollout_group_long<G, 0>(g_lds, scr_blk, N, lane);
        } else {
            rollout_group<G, 0, (G <= 2)>(g_lds, scr_blk, N, lane);
        }
    }
}

#define CILQR_GRP_SIGNATURE                                                                                  \
    (BatchArgs, const double* __restrict__, const double*, double*, double* __restrict__, cilqr_result* __restrict__, \
     cilqr_trace_rec* __restrict__, int)

// ------------------------------------------------------------------------------------------------
// The builds of k_solve the library carries: X(group, DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP).
// `group` = which compilation of cilqr_solve_inst.hip instantiates it (toy-example-of-ilqr_amd/build.py runs the
// groups in parallel).  The production library carries neither the testing-aid builds (DBG: cilqr_set_debug_flags)
// nor the cycle-accounting builds (PROF: cilqr_set_phase_profiling); the development library, libcilqr_amd_dev.so
// (-DCILQR_DEV_BUILD: what tests/ and scripts/ load for those aids), carries both, and its augmented-Lagrangian
// builds understand the debug flags too.
#ifdef CILQR_DEV_BUILD
#define CILQR_ALM_DBG true
#else
#define CILQR_ALM_DBG false
#endif
// Round 5: the production table holds what the dispatcher in cilqr_amd.hip reaches with default settings, nothing else (the
// builds that only tuning switches could select — lone wavefronts one per SIMD, the lone N = 50 / N = 100 builds that the
// grouped kernel and the global-expansion build replaced, the helper build of N = 100 — are gone; DESIGN.md lists the
// survivors with the reason each exists).
#define CILQR_SOLVE_VARIANTS_PROD(X)                                            \
    /* small batches: main + helper wavefront; any horizon (one / two rows per lane), BASELINE's 50, the reference YAMLs' 30 */ \
    X(0, false, 1, false, true, false, 1, CILQR_NT, 0, false, false, false, false)            \
    X(1, false, 2, false, true, false, 1, CILQR_NT, 0, false, false, false, false)            \
    X(2, false, 1, false, true, false, 1, CILQR_NT, 50, false, false, false, false)           \
    X(3, false, 1, false, true, false, 1, CILQR_NT, 30, false, false, false, false)           \
    /* large batches, lone wavefronts two per SIMD — what runs when the grouped kernel is switched off (cilqr_set_group_mode(0), \
       a compiler other than the validated one): N <= 63; N >= 64 with the expansion in global memory, work sharing and \
       resumable solves (round 5: the grouped kernel's long layout took the large batches of these horizons over, the builds \
       with the expansion in LDS for N 64 ... 75 and the compile-time N = 100 one are gone) */ \
    X(4, false, 1, false, false, false, 2, 1, 0, false, false, false, false)                  \
    X(6, false, 2, false, false, false, 2, 1, 0, true, true, true, false)                    \
    /* augmented Lagrangian: helper wavefronts, lone two per SIMD, long horizons with the expansion in global memory */ \
    X(0, CILQR_ALM_DBG, 1, true, true, false, 1, CILQR_NT, 0, false, false, false, false)     \
    X(1, CILQR_ALM_DBG, 2, true, true, false, 1, CILQR_NT, 0, false, false, false, false)     \
    X(2, CILQR_ALM_DBG, 1, true, false, false, 2, 1, 0, false, false, false, false)           \
    X(3, CILQR_ALM_DBG, 2, true, false, false, 2, 1, 0, false, false, false, false)           \
    X(4, CILQR_ALM_DBG, 2, true, false, false, 2, 1, 0, true, true, false, false)             \
    /* closed planning loop in one launch: the plain builds (helper wavefront / lone, two per SIMD), both solve types */ \
    X(5, false, 1, false, true, false, 1, CILQR_NT, 0, false, false, false, true)             \
    X(6, false, 2, false, true, false, 1, CILQR_NT, 0, false, false, false, true)             \
    X(7, false, 1, false, false, false, 2, 1, 0, false, false, false, true)                   \
    X(0, false, 2, false, false, false, 2, 1, 0, false, false, false, true)                   \
    X(1, CILQR_ALM_DBG, 1, true, true, false, 1, CILQR_NT, 0, false, false, false, true)      \
    X(2, CILQR_ALM_DBG, 2, true, true, false, 1, CILQR_NT, 0, false, false, false, true)      \
    X(3, CILQR_ALM_DBG, 1, true, false, false, 2, 1, 0, false, false, false, true)            \
    X(4, CILQR_ALM_DBG, 2, true, false, false, 2, 1, 0, false, false, false, true)
#ifdef CILQR_DEV_BUILD
#define CILQR_SOLVE_VARIANTS_DEV(X)                                             \
    X(3, true, 1, false, false, false, 1, CILQR_NT, 0, false, false, false, false)            \
    X(4, true, 2, false, false, false, 1, CILQR_NT, 0, false, false, false, false)            \
    X(5, false, 1, false, true, true, 1, CILQR_NT, 0, false, false, false, false)             \
    X(6, false, 2, false, true, true, 1, CILQR_NT, 0, false, false, false, false)             \
    X(7, false, 1, false, false, true, 1, CILQR_NT, 0, false, false, false, false)            \
    X(7, false, 2, false, false, true, 1, CILQR_NT, 0, false, false, false, false)            \
    /* the headline build with the cycle accounting: two wavefronts per SIMD, as it runs (the others account at one) */ \
    X(4, false, 1, false, false, true, 2, 1, 50, false, false, false, false)
#else
#define CILQR_SOLVE_VARIANTS_DEV(X)
#endif
#define CILQR_SOLVE_VARIANTS(X) CILQR_SOLVE_VARIANTS_PROD(X) CILQR_SOLVE_VARIANTS_DEV(X)
#define CILQR_SOLVE_GROUPS 8
#define CILQR_SOLVE_SIGNATURE                                                                                      \
    (BatchArgs, const double* __restrict__, const double*, double*, double* __restrict__, \
     cilqr_result* __restrict__, cilqr_trace_rec* __restrict__, int)
