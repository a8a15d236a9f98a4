/*
 * cilqr_amd.h — C-ABI of the MI355X-native batched CILQR solver (libcilqr_amd.so).
 *
 * The reference has no FFI: its boundary is the C++ class CILQRSolver
 * (/root/reference/include/cilqr_solver.hpp:31-148) whose only public method is
 *   std::tuple<MatrixX2d, MatrixX4d> solve(x0, ref_waypoints, ref_velo, obs_preds, road_boaders)
 * (hpp:37-41), called once per planning tick from /root/reference/src/motion_planning.cpp:194-196.
 * This header is what a binding for that path would bind: plain pointers and sizes, no C++/torch
 * types, error codes instead of exceptions.  Every entry point cites the reference member it
 * replaces.  All floating-point data is IEEE binary64; arrays are time-major and row-major:
 *   u[B][N][2] = (acc, steer),  x[B][N+1][4] = (px, py, v, yaw),
 *   K[B][N][2][4], d[B][N][2], l_x[B][N+1][4], l_u[B][N][2], l_xx[B][N+1][4][4], l_uu[B][N][2][2],
 *   A[B][N][4][4], Bm[B][N][4][2].
 *
 * Threading: a handle is not thread-safe (the reference instance is not re-entrant either); use
 * one handle per host thread / per GPU.  Calls are synchronous unless the name ends in _async.
 * Functions ending in _device take DEVICE pointers for the bulk arrays and a hipStream_t (as
 * void*); everything else takes HOST pointers and stages through the handle's own buffers.
 */
#ifndef CILQR_AMD_H
#define CILQR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CILQR_OK 0
#define CILQR_ERR_BAD_ARG (-1)
#define CILQR_ERR_OBSTACLE_HORIZON (-2) /* an obstacle route is shorter than tick+N+1 (upstream: std::out_of_range, src/utils.cpp:52-58) */
#define CILQR_ERR_DEVICE (-3)           /* HIP runtime error; see cilqr_last_error() */
#define CILQR_ERR_UNSUPPORTED (-4)
#define CILQR_ERR_NO_DEVICE (-5)

/* Largest lqr/N a handle accepts (cilqr_set_params answers CILQR_ERR_BAD_ARG beyond it).  Upstream's N is any int
 * (src/cilqr_solver.cpp:19); here the horizon-parallel phases hold one, two or — round 6: horizons of 128 ... 255, one family
 * of builds: two trajectories per wavefront, both solve types — four rows of the trajectory per lane of a 64-lane wavefront
 * (N + 1 <= 256).  At horizons above 127 the stage-by-stage entry points and the closed loop in one launch UNDER THE AUGMENTED
 * LAGRANGIAN answer CILQR_ERR_UNSUPPORTED (cilqr_solve, cilqr_solve_batch, cilqr_solve_batch_device and
 * cilqr_advance_batch_device cover that planning loop tick by tick; in barrier mode cilqr_closed_loop_batch_device runs).  The YAMLs upstream ships use 30; BASELINE's configurations 50 and 100.  A
 * documented limit of this drop-in, not a silent one: nothing is truncated.  All parameter sets of one handle share N
 * and the solve type (one kernel build per launch); use one handle per (N, solve type). */
#define CILQR_MAX_HORIZON 255
#define CILQR_MAX_ALPHA_TRIALS 20 /* alpha = 1, 1/2, ... while > 1e-6 (src/cilqr_solver.cpp:354) */

/* The scalars CILQRSolver::CILQRSolver copies out of GlobalConfig (src/cilqr_solver.cpp:17-83);
 * field names are the config keys. */
typedef struct cilqr_params {
    int32_t N;               /* lqr/N */
    int32_t max_iter;        /* iteration/max_iter */
    int32_t solve_type;      /* lqr/slove_type: 0 = "barrier", 1 = "alm" */
    int32_t reference_point; /* vehicle/reference_point: 0 = "rear_center", 1 = "gravity_center" */
    int32_t use_last_solution;
    int32_t reserved0;
    double dt; /* delta_t */
    double w_pos, w_vel, w_yaw, w_acc, w_stl;
    double obstacle_exp_q1, obstacle_exp_q2, state_exp_q1, state_exp_q2;
    double alm_rho_init, alm_gamma, max_rho, max_mu;
    double init_lamb, lamb_decay, lamb_amplify, max_lamb;
    double convergence_threshold, accept_step_threshold;
    double wheelbase, width, length, velo_max, velo_min, yaw_lim, acc_max, acc_min, stl_lim, d_safe;
} cilqr_params;

/* LQRSolveStatus (include/cilqr_solver.hpp:23-29) */
enum {
    CILQR_RUNNING = 0,
    CILQR_CONVERGED = 1,
    CILQR_BACKWARD_PASS_FAIL = 2,
    CILQR_FORWARD_PASS_FAIL = 3,
    CILQR_FORWARD_PASS_SMALL_STEP = 4
};

/* why solve() left its loop (src/cilqr_solver.cpp:127-148) */
enum {
    CILQR_END_CONVERGED = 0,
    CILQR_END_MAX_LAMB = 1,
    CILQR_END_MAX_ITER = 2,
    /* not solved: scenario_id / param_id outside the tables, tick < 0, or an obstacle route shorter than
     * tick + N + 1 (upstream: RoutingLine::operator[] throws std::out_of_range, src/utils.cpp:52-58).  Only the
     * device-pointer entry point reports it this way (its index arrays cannot be checked on the host): u, x
     * and the costs of that trajectory are NaN, iters = 0.  The host-pointer entry points return
     * CILQR_ERR_BAD_ARG / CILQR_ERR_OBSTACLE_HORIZON before launching anything. */
    CILQR_END_BAD_INPUT = 3,
    /* not solved, and not by the caller's doing: the launch never finished this trajectory.  Every launch that hands
     * trajectories from wavefront to wavefront (pairs per wavefront: sliced solves, idle wavefronts at the tail; resumable solves
     * and work sharing of the long-horizon builds) first marks EVERY result of the batch with this value (J_init = J_final = NaN,
     * iters = 0) on the launch stream; whoever finishes a trajectory overwrites the mark.  A bounded wait of the hand-over
     * protocol that expired leaves the trajectory that was in transit marked — it cannot be mistaken for a result (upstream,
     * cs:144-152, returns every solve's (u, x): a missing one must be loud) — and the same launch is reported by cilqr_wait /
     * cilqr_solve_batch as CILQR_ERR_DEVICE.  Never observed outside the test that forces it (development library,
     * CILQR_TUNE=grp_wait_spins). */
    CILQR_END_NOT_SOLVED = 4
};

/* The non-ego arguments of one solve() call, shared by many trajectories of a batch:
 * ref_waypoints (.x/.y/.yaw of a ReferenceLine, include/utils.hpp:32-51), the full obstacle
 * routes (RoutingLine, include/utils.hpp:53-68; obs_preds[j][k] == obs[j][tick + k] as in
 * utils::get_sub_routing_lines, src/utils.cpp:88-103), road_boaders and ref_velo. */
typedef struct cilqr_scenario_desc {
    const double* lane_x;
    const double* lane_y;
    const double* lane_yaw;
    int32_t L; /* lane samples, 1 <= L <= 65535 (uint16_t indices upstream, cs:291-295) */
    int32_t M; /* obstacles */
    const double* obs; /* [M][T][3] = (x, y, yaw) */
    int32_t T;
    int32_t reserved0;
    double road_borders[2]; /* (max border offset, min border offset), motion_planning.cpp:101-103 */
    double ref_velo;        /* vehicle/target_velocity */
} cilqr_scenario_desc;

/* one record per executed trip of the loop at src/cilqr_solver.cpp:110 */
typedef struct cilqr_trace_rec {
    int32_t status;    /* current_solve_status after iter_step */
    int32_t trials;    /* forward_pass+get_total_cost evaluations the reference would have made */
    int32_t accepted;  /* effective_flag */
    int32_t alpha_idx; /* accepted / converged alpha = 2^-idx, -1 if none */
    double lamb;       /* after the update at :118-125 */
    double new_J;      /* cost returned by iter_step */
} cilqr_trace_rec;

typedef struct cilqr_result {
    double J_init;  /* cost of the initial trajectory — the value the reference logs as "final cost" */
    double J_final; /* get_total_cost(u_ret, x_ret) */
    int32_t iters;
    int32_t end_reason;
    int32_t final_status;
    int32_t ls_trials;
    int32_t cost_evals; /* get_total_cost calls the reference would have made inside solve() */
    int32_t trace_len;
} cilqr_result;

typedef struct cilqr_handle cilqr_handle;

/* ---- lifetime ------------------------------------------------------------------------- */
/* device = HIP device ordinal.  Fails with CILQR_ERR_NO_DEVICE when no GPU is visible: there
 * is no CPU fallback in this library. */
int cilqr_create(int device, cilqr_handle** out);
/* HIP devices visible to the process (one handle per device shards a batch: cilqr_amd::ShardedSolver, cilqr_solver_shim.hpp) */
int cilqr_device_count(int32_t* n);
int cilqr_destroy(cilqr_handle* h);
const char* cilqr_last_error(void);
const char* cilqr_version(void);

/* Replaces CILQRSolver::CILQRSolver(config) (cs:17-83).  A table of parameter sets so that a
 * batch can sweep settings (BASELINE config 5); all entries must share N. */
int cilqr_set_params(cilqr_handle* h, const cilqr_params* params, int32_t n_params);
/* Uploads the scenario tables to HBM (copied; the caller's arrays are not retained). */
int cilqr_set_scenarios(cilqr_handle* h, const cilqr_scenario_desc* scen, int32_t n_scen);

/* ---- the path: CILQRSolver::solve for a batch (cs:85-153) --------------------------------- */
/* x0[B][4]; scenario_id/param_id/tick [B] (NULL = all zeros).  last_u: NULL for a cold start
 * (get_init_traj, cs:155-161) or [B][N][2] = the previous solution of each trajectory for a
 * warm start (get_init_traj_increment, cs:163-180: shifted by one step, last row repeated).
 * trace: NULL or [B][trace_cap].  Returns when the host buffers are filled. */
int cilqr_solve_batch(cilqr_handle* h, int32_t B, const double* x0, const int32_t* scenario_id,
                      const int32_t* param_id, const int32_t* tick, const double* last_u,
                      double* u_out, double* x_out, cilqr_result* res_out,
                      cilqr_trace_rec* trace_out, int32_t trace_cap);

/* CILQRSolver::solve exactly as main() calls it (include/cilqr_solver.hpp:37-41, src/motion_planning.cpp:194-196):
 * ONE ego, every argument handed over on every call.  x0[4]; sc = (ref_waypoints, obs_preds from the current tick
 * on, road_boaders, ref_velo); last_u NULL or [N][2] (warm start, cs:163-180); u_out[N][2], x_out[N+1][4].
 * The tables stay in HBM between calls and are uploaded again only when their contents differ bitwise from what
 * is there; obstacle predictions that are the tail of the routes uploaded earlier — what
 * utils::get_sub_routing_lines (src/utils.cpp:88-103) hands over tick after tick — only move a tick offset.
 * Uses scenario slot 0 of the handle (replacing whatever cilqr_set_scenarios put there) and parameter set 0. */
int cilqr_solve(cilqr_handle* h, const double* x0, const cilqr_scenario_desc* sc, const double* last_u,
                double* u_out, double* x_out, cilqr_result* res_out);
/* how often cilqr_solve uploaded tables / re-used the resident ones (either pointer may be NULL) */
int cilqr_solve_cache_stats(cilqr_handle* h, int64_t* uploads, int64_t* reuses);

/* Same as cilqr_solve_batch with every array already resident in HBM; enqueues on `stream` (hipStream_t) and
 * returns without synchronising.  The index arrays are checked on the device: a trajectory with an id outside the
 * tables or too short an obstacle route is not solved and ends with CILQR_END_BAD_INPUT.
 * ONE LAUNCH PER HANDLE AT A TIME: the scratch areas, the persistent blocks' trajectory counter and the work-sharing
 * state belong to the handle.  Launches of one handle are therefore ordered on the device — a launch on another
 * stream than the handle's previous one waits for it (hipStreamWaitEvent) — and do not overlap; independent batches
 * that should overlap take one handle each (and one stream each).  A hand-off failure inside a launch (a bounded wait that
 * ran out — work sharing between blocks: the owner then costs the trial itself, results stay correct; a trajectory handed from
 * wavefront to wavefront: its result keeps the CILQR_END_NOT_SOLVED mark) is latched in the handle whichever launch slot it
 * happened in and however many launches followed: cilqr_solve_batch and the next cilqr_wait return CILQR_ERR_DEVICE (once;
 * the latch is cleared by the report), cilqr_work_sharing_stats()[3] shows it without clearing. */
int cilqr_solve_batch_device(cilqr_handle* h, int32_t B, const double* d_x0,
                             const int32_t* d_scenario_id, const int32_t* d_param_id,
                             const int32_t* d_tick, const double* d_last_u, double* d_u_out,
                             double* d_x_out, cilqr_result* d_res_out, cilqr_trace_rec* d_trace_out,
                             int32_t trace_cap, void* stream);

/* The step after the path for a whole batch, on the device (src/motion_planning.cpp:181,197): ego_state =
 * new_x.row(1) and the obstacle window one tick on: d_x0[b] = d_x[b][1], d_tick[b] += 1 (d_tick may be NULL).
 * Together with cilqr_solve_batch_device — whose d_last_u may be the d_u_out buffer of the previous tick, also
 * when it is this tick's d_u_out: a block reads its rows before it writes them — a closed planning loop over
 * thousands of egos runs without a host round trip per tick. */
int cilqr_advance_batch_device(cilqr_handle* h, int32_t B, const double* d_x, double* d_x0, int32_t* d_tick,
                               void* stream);

/* The whole closed planning loop of src/motion_planning.cpp:180-197 for B egos in ONE launch: every ego runs `ticks`
 * ticks back to back on the block that picked it up — CILQRSolver::solve, ego_state = new_x.row(1), obstacle window one
 * tick on, the next solve warm-started from the plan just made where the ego's parameter set has use_last_solution
 * (cilqr_solver.cpp:95-101, 163-180), cold otherwise as in the reference (under "alm" with fresh multipliers: cs:88-93);
 * the first tick starts from d_last_u, NULL = cold — with no synchronisation between egos: a tick-by-tick loop of cilqr_solve_batch_device +
 * cilqr_advance_batch_device ends every tick with the batch's slowest solves on a mostly idle chip, the fused loop ends
 * once.  Same numbers as that loop (d_last_u = the previous d_u_out where use_last_solution, else NULL), ego by ego.  d_x0[B][4] and d_tick[B] (required) are read AND advanced; d_u_out /
 * d_x_out / d_res_out hold the LAST tick's plan; optional d_states[B][ticks][4] = the ego state after every tick,
 * d_iters[ticks][B] = iterations of every tick's solve.  An ego whose obstacle routes run out (tick + N + 1 > T) stops there
 * with CILQR_END_BAD_INPUT.  Resumable solves do not apply here (an ego's ticks are its slices). */
int cilqr_closed_loop_batch_device(cilqr_handle* h, int32_t B, int32_t ticks, double* d_x0, const int32_t* d_scenario_id,
                                   const int32_t* d_param_id, int32_t* d_tick, const double* d_last_u, double* d_u_out,
                                   double* d_x_out, cilqr_result* d_res_out, double* d_states, int32_t* d_iters,
                                   void* stream);

/* Batches in flight (round 5).  k = 1 (default): the behaviour described above — one launch of the handle at a time, on the
 * caller's stream.  k = 2 ... 4: the handle owns k LAUNCH SLOTS, each with an internal stream, trial slabs, control words and
 * parked-solve state of its own (the parameter and scenario tables are shared), and consecutive cilqr_solve_batch_device /
 * cilqr_closed_loop_batch_device calls go to the slots round robin.  A launch ends with its longest solves on a chip that is
 * emptying (the tail: 5 % of a 65 536-trajectory launch, a third of an 8 192-trajectory one); with batches in flight the
 * persistent blocks of the next batch move into the wave slots the tail frees.  Semantics in this mode:
 *   - a launch starts after everything that was enqueued on `stream` before the call (an event: the caller's copies into
 *     x0 / tick are seen), after the previous launch of its slot, and after every launch still in flight whose buffers
 *     overlap its own with a write on either side (address ranges of x0, ids, tick, last_u, u_out, x_out, res, trace,
 *     states, iters: a warm start from the previous call's u_out, or the same output arrays, orders the two automatically);
 *   - `stream` does NOT wait for the launch: results are complete on a stream after cilqr_join_device(h, that stream), on the
 *     host after cilqr_wait(h).  Every other entry point of the handle (host-buffer calls, cilqr_advance_batch_device,
 *     piecewise calls, cilqr_set_params / _scenarios) joins first, by itself;
 *   - handles in "alm" mode (multipliers per trajectory live in the handle) and the development aids keep one launch at a
 *     time whatever k;
 *   - SIDE EFFECT on the host application: the slots' internal streams are created at the device's HIGHEST stream priority (the
 *     runtime spreads the streams of a priority level over that level's hardware queues; at normal priority the third slot
 *     shared a queue with the caller's stream and every third launch waited behind a 60 ms kernel).  While solves are in flight
 *     their wavefronts are therefore scheduled ahead of the process's other work on that GPU — kernels of other libraries,
 *     copies feeding the next batch — whenever both are ready.  A planner that must keep other streams responsive uses k = 1
 *     (the launch then rides on the caller's own stream and priority) or one handle per stream.
 * Results do not depend on k.  cilqr_slot_kernel_ms: duration of slot k's last launch when cilqr_set_timing is on. */
int cilqr_set_batches_in_flight(cilqr_handle* h, int32_t k);
int cilqr_join_device(cilqr_handle* h, void* stream);
int cilqr_wait(cilqr_handle* h);
int cilqr_slot_kernel_ms(cilqr_handle* h, int32_t k, float* ms);

/* Wall time of the most recent solve kernel measured with HIP events on its own stream (ms). */
int cilqr_last_kernel_ms(cilqr_handle* h, float* ms);
/* Shape of the most recent fused launch: out = { trajectories per wavefront (1: k_solve, 2 or 3: k_solve_grp), blocks in the
 * grid, threads per block (128 = main + helper wavefront), lane-window samples in LDS }. */
int cilqr_last_launch_info(cilqr_handle* h, int32_t out[4]);
/* When enabled, every cilqr_solve_batch*_ call brackets its kernel with HIP events. */
int cilqr_set_timing(cilqr_handle* h, int32_t enabled);

/* solve_type "alm" (cs:88-93, 253-277, 377-378, 581-643, 665-680): the multipliers alm_mu / alm_mu_next
 * ([B][N][cols], cols = 8 + 2 * max M over the scenarios) and alm_rho ([B]) live in the handle, as they
 * live in the reference instance; a solve with last_u == NULL resets them, a warm-started solve keeps
 * them.  These two calls expose them (tests; resuming a batch on another handle).  Any pointer may be NULL. */
int cilqr_set_alm_state(cilqr_handle* h, int32_t B, const double* mu, const double* rho);
int cilqr_get_alm_state(cilqr_handle* h, int32_t B, double* mu, double* mu_next, double* rho, int32_t* cols);

/* Helper wavefronts: -1 (default) = automatic — a second wavefront per trajectory costs every other
 * line-search trial when the batch is small enough for that to pay (<= 1536 trajectories; <= 512 for horizons of 96
 * and more in barrier mode, where work sharing between blocks does better): up to 1024 one wavefront each would leave
 * SIMD slots empty, a little beyond that the quicker stragglers still outweigh the second round of blocks;
 * 0 = never; 1 = always.  Results are identical in every mode. */
int cilqr_set_helper_mode(cilqr_handle* h, int32_t mode);

/* Trajectories per wavefront (barrier mode, every horizon the library takes — up to 63 with one row per lane, 64 ... 127
 * with two and both trajectories' expansions streamed from global memory): -1 (default) = automatic — batches beyond the
 * helper range run two trajectories per wavefront, whose line-search rollouts (src/cilqr_solver.cpp:442-461, a serial chain
 * over the horizon) share one pass of the instruction stream and whose backward sweeps (cs:383-440) run as one, a
 * trajectory per half-wavefront; 0 or 1 = one trajectory per wavefront everywhere; 2 = two wherever that build can run (any
 * batch size).  Results are identical in every mode.  Handles in "alm" mode run in pairs only when mode 2 is set explicitly
 * (round 6: those two kernels are bit-exact under the wave64 emulator of tests/ but have not yet run on a GPU; -1 keeps such batches
 * on the lone-wavefront builds).  A launch in pairs hands trajectories from wavefront to wavefront
 * (sliced solves, idle wavefronts at its tail); should a bounded wait inside it ever expire, the trajectory that was in transit
 * keeps end_reason = CILQR_END_NOT_SOLVED in its cilqr_result (every result is pre-marked on the launch stream), and
 * cilqr_solve_batch / the next cilqr_wait return CILQR_ERR_DEVICE — whichever launch slot it happened in.  The bound is 2^20 polls of
 * ~5 us: an idle wavefront gives up when the REST of a launch lasts longer than about five seconds — launches are milliseconds;
 * a closed loop of many ticks at horizons in the hundreds is the one shape that can approach it: cut such loops into several calls (never observed in
 * 52 k stress launches; forced by tests/test_gpu_parity.py::test_a_lost_hand_over_is_loud). */
int cilqr_set_group_mode(cilqr_handle* h, int32_t mode);

/* Work sharing between blocks (horizons above 63, batches beyond the helper range): 1 (default) = blocks that find no
 * trajectory left to solve (large batches run persistent blocks that pull trajectories from a counter) cost
 * line-search trials of the trajectories still being solved; 0 = off.  A cost is a function of the trial trajectory alone, so the results
 * are identical either way; what changes is how long a launch waits for its slowest trajectories. */
int cilqr_set_work_sharing(cilqr_handle* h, int32_t mode);
/* Counters of the last launch that shared work (waits for the device): out = { line searches announced, trial costs
 * delivered by other blocks, blocks that stayed to help, non-zero if a bounded wait expired in that launch OR in any launch of
 * any slot since the latch was last reported (see cilqr_solve_batch_device) }. */
int cilqr_work_sharing_stats(cilqr_handle* h, uint32_t out[4]);
/* Resumable solves (long horizons — two rows per lane — in batches that take more than one round of the chip's resident
 * blocks): a solve runs `iters` iterations at a time; in between its state (x, u, lane indices, a dozen scalars: what
 * cilqr_solver.cpp:110-141 carries from one iteration to the next) is parked in HBM and its number queued, and blocks pull
 * fresh trajectories before parked ones — so the solves that will take 100 iterations are well under way when the short
 * ones are done, instead of starting late and finishing alone.  Same results bit for bit (whoever resumes a solve
 * computes the same numbers).  The launches that run two trajectories per wavefront (large batches, barrier mode) slice
 * their solves the same way once the last round of fresh trajectories is being handed out.  -1 (default) = automatic: 32
 * iterations per slice for lone wavefronts, 16 (horizons up to 63) / 12 (longer) for pairs; 0 = every solve runs to its end
 * in one go; n > 0 = n iterations per slice everywhere.  cilqr_resume_stats: how many times a solve was parked (end of a
 * slice, or handed to an idle wavefront at the tail of the launch) in the handle's last such launch. */
int cilqr_set_resume_iters(cilqr_handle* h, int32_t iters);
int cilqr_resume_stats(cilqr_handle* h, uint32_t* parked);

/* Line-search rollouts (forward_pass for the step sizes of cs:354): -1 (default) = adaptive — an iteration rolls
 * out alpha = 1 alone and the other 19 step sizes only once that trial is rejected, unless the previous
 * iteration's search went beyond its first trial, in which case all 20 are rolled out in one pass; 0 = always all
 * 20 in one pass; 1 = always the first trial alone first.  Results are identical in every mode. */
int cilqr_set_rollout_mode(cilqr_handle* h, int32_t mode);

/* Optional in-kernel cycle accounting of the fused solve (development aid): when enabled, the next
 * solve records, per trajectory, shader-clock cycles spent in
 * [0] initial trajectory + cost, [1] cost/model derivatives, [2] backward sweep, [3] trial rollouts,
 * [4] trial cost evaluations, [5] accepting a trial, [6] whole solve, [7] iterations,
 * [8] trial cost evaluations that fell back to the serial reference-point chain, [9] trials,
 * [10..12] split of [4]: reference points, stage costs, ordered sum, [13] trial cost evaluations in which
 * some row's reference-point proof had to sample the lane interval (convexity certificate not
 * applicable), [14] rollout passes of the first trial alone, [15] rollout passes of all 20 step sizes at once,
 * [16] of those, second passes after a rejected first trial.  out[B][17]. */
#define CILQR_PROF_SLOTS 17
/* (development library only — libcilqr_amd_dev.so, the same sources built with -DCILQR_DEV_BUILD: the production
 *  library carries neither the cycle-accounting nor the testing-aid builds of the solve kernel and answers
 *  CILQR_ERR_UNSUPPORTED to a request to switch them on; it does not read the CILQR_TUNE environment variable either) */
int cilqr_set_phase_profiling(cilqr_handle* h, int32_t enabled);
/* Testing aid (development library only).  bit 0: always use the serial reference-point chain (cs:289-314 as written) instead
 * of the lane-parallel search + proof; bit 1: wave-uniform backward sweep instead of the
 * lane-parallel one.  Results must be identical either way. */
int cilqr_set_debug_flags(cilqr_handle* h, int32_t flags);
int cilqr_get_phase_cycles(cilqr_handle* h, int64_t* out, int32_t B);
/* Development aid: when enabled, the fused solve records per trajectory when the block that solved it started and
 * ended (constant 100 MHz clock), that block's index and the XCC (chiplet) it ran on — out[B][4] — which shows how
 * the launch fills the chip over time (scripts/block_timeline.py).  A launch that runs resumable solves has no one
 * block per trajectory: there the record holds the start of the first slice, the end of the last, MINUS the time the
 * solve was actually running (100 MHz ticks, summed over its slices) in place of the block index, and the last XCC. */
int cilqr_set_block_timeline(cilqr_handle* h, int32_t enabled);
int cilqr_get_block_timeline(cilqr_handle* h, int64_t* out, int32_t B);

/* ---- the pieces of the path, exported so each can be parity-checked on its own -------------- */
/* get_init_traj / const_velo_prediction (cs:155-161,182-197): x_out[B][N+1][4] */
int cilqr_init_traj_batch(cilqr_handle* h, int32_t B, const double* x0, const int32_t* param_id,
                          double* x_out);
/* get_ref_exact_points (cs:289-314): ref_out[B][N+1][3], idx_out[B][N+1] */
int cilqr_ref_points_batch(cilqr_handle* h, int32_t B, const double* x, const int32_t* scenario_id,
                           const int32_t* param_id, double* ref_out, int32_t* idx_out);
/* get_total_cost (cs:199-287): J_out[B] */
int cilqr_total_cost_batch(cilqr_handle* h, int32_t B, const double* u, const double* x,
                           const int32_t* scenario_id, const int32_t* param_id, const int32_t* tick,
                           double* J_out);
/* forward_pass (cs:442-461) for every trial step size at once: alpha_idx a -> alpha = 2^-a.
 * new_u[B][n_alpha][N][2], new_x[B][n_alpha][N+1][4], J_out[B][n_alpha] (get_total_cost of each) */
int cilqr_forward_pass_batch(cilqr_handle* h, int32_t B, const double* u, const double* x,
                             const double* d, const double* K, const int32_t* scenario_id,
                             const int32_t* param_id, const int32_t* tick, int32_t n_alpha,
                             double* new_u, double* new_x, double* J_out);
/* get_total_cost_derivatives_and_Hessians (cs:463-690) + utils::get_kinematic_model_derivatives
 * (utils.cpp:285-342); any output may be NULL */
int cilqr_cost_derivatives_batch(cilqr_handle* h, int32_t B, const double* u, const double* x,
                                 const int32_t* scenario_id, const int32_t* param_id,
                                 const int32_t* tick, double* l_x, double* l_u, double* l_xx,
                                 double* l_uu, double* A, double* Bm);
/* backward_pass (cs:383-440): lamb[B] -> d, K, dV[B][2], status[B] (RUNNING or BACKWARD_PASS_FAIL;
 * on failure d/K hold the rows computed before the failing step, zeros elsewhere) */
int cilqr_backward_pass_batch(cilqr_handle* h, int32_t B, const double* u, const double* x,
                              const double* lamb, const int32_t* scenario_id,
                              const int32_t* param_id, const int32_t* tick, double* d, double* K,
                              double* dV, int32_t* status);
/* elementary functions as evaluated on the device (csrc/detmath.h):
 * func 0 exp, 1 sin, 2 cos, 3 tan, 4 atan, 5 hypot(x,y), 6 x/y, 7 sqrt(|x|),
 * 8 sin, 9 cos, 10 tan in the register flavour the rollout uses (same values) */
int cilqr_detmath_eval(cilqr_handle* h, int32_t func, const double* x, const double* y, int32_t n,
                       double* out);

/* ---- scenario construction on the host (no GPU needed) -------------------------------------- */
/* ReferenceLine::ReferenceLine(_x, _y, width, accuracy) (src/utils.cpp:21-35) on top of
 * CubicSpline2D (src/cubic_spline.cpp:130-157).  Writes up to cap samples of x/y/yaw/longitude
 * and the total count to *count. */
int cilqr_reference_line_build(const double* wx, const double* wy, int32_t n, double width,
                               double accuracy, double* x, double* y, double* yaw, double* s,
                               int32_t cap, int32_t* count);
/* ReferenceLine::calc_position (src/utils.cpp:60-67): out = (x, y, yaw) */
int cilqr_reference_line_position(const double* wx, const double* wy, int32_t n, double width,
                                  double cur_s, double out[3]);
/* The route fabrication of main() (src/motion_planning.cpp:121-173) without the random noise:
 * init_cond[V][4] = (x, y, v, yaw); center_widths[n_center]; routes[V][T_cap][3]; *T_out = samples
 * per route ( t = 0; t < max_simulation_time + 10; t += delta_t ).  line_num/start_s may be NULL. */
int cilqr_build_routes(const double* wx, const double* wy, int32_t n, const double* center_widths,
                       int32_t n_center, double accuracy, const double* init_cond, int32_t V,
                       double max_simulation_time, double delta_t, double* routes, int32_t T_cap,
                       int32_t* T_out, int32_t* line_num, double* start_s);

/* Initial states of the synthetic benchmark batches (BASELINE configs 2-5): row b of a batch is a
 * function of (seed, first + b) alone — splitmix64 counter generator -> uniform -> Box-Muller — so shards of a
 * batch regenerate exactly their own rows.  x0_out[B][4] = base + (U(-5,5), +-U(0.05,1.0), N(0,0.5), N(0,0.02));
 * |dy| >= 0.05 keeps the starts off the singular reference line.  Host C++ twin of workloads.py::perturbed_starts. */
int cilqr_perturbed_starts(const double base[4], int32_t B, uint64_t seed, int64_t first, double* x0_out);

#ifdef __cplusplus
}
#endif
#endif /* CILQR_AMD_H */
