/*
 * cilqr_oracle.h — CPU ORACLE for the CILQR solve path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm
 *   /root/reference/src/cilqr_solver.cpp:85-739  (CILQRSolver::solve and everything it calls)
 *   /root/reference/src/utils.cpp:262-439        (kinematic model, circle centres, ellipse margin)
 * written from the reference's semantics, function by function (each function cites the lines it
 * follows).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * shipped HIP path never calls into it.
 *
 * PARITY STATUS: "parity unpinned" against the reference C++ binary — the reference cannot be
 * built in this environment (Eigen / yaml-cpp / spdlog / fmt absent, no network) and ships no
 * tests or golden vectors.  What IS pinned (tests/test_oracle_golden.py):
 *   - every leaf function against vectors produced by importing the reference's own Python
 *     modules scripts/utils/kinematic.py and scripts/utils/constraint.py (tests/golden/);
 *   - gradients against finite differences of the oracle's own cost;
 *   - solve-level statistics against the survey's independent NumPy reading.
 *
 * Two builds of the same source:
 *   liboracle_libm.so  — elementary functions from glibc libm, as the reference uses them;
 *   liboracle_det.so   — -DORC_DETMATH: elementary functions from csrc/detmath.h, the same code
 *                        the HIP kernels use, so oracle and device agree bit-for-bit and
 *                        decision traces can be compared exactly.
 * Floating-point expression order follows the reference's Eigen expressions evaluated left to
 * right, inner products accumulated in index order, no fused multiply-add (-ffp-contract=off).
 */
#ifndef CILQR_ORACLE_H
#define CILQR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors the members CILQRSolver copies from GlobalConfig (src/cilqr_solver.cpp:17-83) */
typedef struct orc_params {
    int32_t N;               /* lqr/N */
    int32_t max_iter;        /* iteration/max_iter */
    int32_t solve_type;      /* 0 = barrier, 1 = alm (lqr/slove_type) */
    int32_t reference_point; /* 0 = rear_center, 1 = gravity_center (vehicle/reference_point) */
    int32_t use_last_solution;
    int32_t reserved0;
    double dt;
    double w_pos, w_vel, w_yaw, w_acc, w_stl;
    double obstacle_exp_q1, obstacle_exp_q2, state_exp_q1, state_exp_q2;
    double alm_rho_init, alm_gamma, max_rho, max_mu;
    double init_lamb, lamb_decay, lamb_amplify, max_lamb;
    double convergence_threshold, accept_step_threshold;
    double wheelbase, width, length, velo_max, velo_min, yaw_lim, acc_max, acc_min, stl_lim, d_safe;
} orc_params;

/* LQRSolveStatus (include/cilqr_solver.hpp:23-29) */
enum {
    ORC_RUNNING = 0,
    ORC_CONVERGED = 1,
    ORC_BACKWARD_PASS_FAIL = 2,
    ORC_FORWARD_PASS_FAIL = 3,
    ORC_FORWARD_PASS_SMALL_STEP = 4
};

/* why solve() left its loop (src/cilqr_solver.cpp:127-148) */
enum { ORC_END_CONVERGED = 0, ORC_END_MAX_LAMB = 1, ORC_END_MAX_ITER = 2 };

/* scenario inputs of one solve() call */
typedef struct orc_scene {
    const double* lane_x;   /* ReferenceLine::x   [L] */
    const double* lane_y;   /* ReferenceLine::y   [L] */
    const double* lane_yaw; /* ReferenceLine::yaw [L] */
    int32_t L;
    int32_t M;              /* number of obstacles */
    const double* obs;      /* [M][T][3] (x, y, yaw) full routes */
    int32_t T;              /* samples per route */
    int32_t tick;           /* obs_preds[j][k] == obs[j][tick + k]  (utils.cpp:88-103) */
    double road_borders[2]; /* (max border offset, min border offset) */
    double ref_velo;
} orc_scene;

/* per-iteration decision trace record */
typedef struct orc_trace_rec {
    int32_t status;     /* current_solve_status after iter_step */
    int32_t trials;     /* forward_pass + get_total_cost evaluations in this iteration */
    int32_t accepted;   /* effective_flag after iter_step */
    int32_t alpha_idx;  /* index of the accepted/converged alpha = 2^-idx, -1 if none */
    double lamb;        /* lamb after the update at :118-125 */
    double new_J;       /* cost returned by iter_step */
} orc_trace_rec;

typedef struct orc_result {
    double J_init;   /* get_total_cost of the initial trajectory (what the reference logs) */
    double J_final;  /* get_total_cost(u_ret, x_ret)   */
    int32_t iters;   /* trips of the loop at :110 that were executed */
    int32_t end_reason;
    int32_t final_status;
    int32_t ls_trials;    /* sum of trials over iterations */
    int32_t cost_evals;   /* calls of get_total_cost inside solve (excl. the J_final recomputation) */
    int32_t trace_len;
} orc_result;

typedef struct orc_solver orc_solver;

/* How close the discrete decisions of ONE iteration (one trip of the loop at cs:110) came to going the other
 * way: the smallest relative distance |a - b| / scale over every comparison a < b of that kind evaluated in the
 * iteration (DBL_MAX if none was).  ls: the line-search verdicts (cs:358-365: |decay| < threshold, decay > 0,
 * approx < 0, decay / approx > threshold), scale = |ori_cost|; pd: the two Cholesky pivots of every step
 * (cs:415-416), scale = sum of the magnitudes of the terms of the pivot; ref: every `cur < min_distance` of the
 * lane scans (cs:300), scale = the larger distance.  Record 0 also covers the initial trajectory's scan. */
typedef struct orc_margin_rec {
    double ls_margin, pd_margin, ref_margin;
} orc_margin_rec;
void orc_set_margin_buffer(orc_solver* s, orc_margin_rec* buf, int32_t cap);

int orc_math_mode(void); /* 0 = libm, 1 = detmath */
int orc_fused(void);     /* 1 in the fused-flavour build (-DORC_FUSED: an experiment, see cilqr_oracle.c) */

orc_solver* orc_create(const orc_params* p);
void orc_destroy(orc_solver* s);
void orc_reset(orc_solver* s); /* is_first_solve = true, forget warm start */

/* CILQRSolver::solve.  u_out[N*2], x_out[(N+1)*4] row-major (time-major).
 * trace may be NULL; at most trace_cap records are written.
 * returns 0, or -1 if an obstacle route is shorter than tick+N+1 (std::out_of_range upstream). */
int orc_solve(orc_solver* s, const double x0[4], const orc_scene* sc, double* u_out, double* x_out,
              orc_result* res, orc_trace_rec* trace, int32_t trace_cap);

/* batch driver for the CPU baseline: B independent fresh solves, OpenMP over trajectories.
 * x0[B*4]; scene_id[B] indexes scenes[]; param_id[B] indexes params[]; tick[B] overrides
 * scenes[].tick.  Outputs time-major per trajectory. */
int orc_solve_batch(const orc_params* params, int32_t n_params, const orc_scene* scenes,
                    int32_t n_scenes, int32_t B, const double* x0, const int32_t* scene_id,
                    const int32_t* param_id, const int32_t* tick, int32_t n_threads, double* u_out,
                    double* x_out, orc_result* res);

/* ---- piecewise entry points (stateless unless noted) ---- */
void orc_kinematic_propagate(const double x[4], const double u[2], double dt, double wheelbase,
                             int32_t reference_point, double out[4]);
/* A[N][4][4], B[N][4][2] */
void orc_model_derivatives(const double* x, const double* u, double dt, double wheelbase, int32_t N,
                           int32_t reference_point, double* A, double* B);
void orc_front_rear_centers(const double state[4], double wheelbase, int32_t reference_point,
                            double front[2], double rear[2]);
/* front_over_state[4][2], rear_over_state[4][2] */
void orc_front_rear_center_derivatives(double yaw, double wheelbase, int32_t reference_point,
                                       double* front_over_state, double* rear_over_state);
void orc_ellipsoid_scales(const double obs_attr[3], double ego_pnt_radius, double ab[2]);
double orc_ellipsoid_safety_margin(const double pnt[2], const double obs_state[3], const double ab[2]);
void orc_ellipsoid_safety_margin_derivatives(const double pnt[2], const double obs_state[3],
                                             const double ab[2], double out[2]);
double orc_exp_barrier(double c, double q1, double q2);
/* c_dot[n]; b_dot[n]; b_ddot[n][n] */
void orc_exp_barrier_derivative_and_Hessian(double c, const double* c_dot, int32_t n, double q1,
                                            double q2, double* b_dot, double* b_ddot);
void orc_obstacle_constr(const orc_params* p, const double ego[4], const double obs[3], double out[2]);
void orc_obstacle_constr_derivatives(const orc_params* p, const double ego[4], const double obs[3],
                                     double front_over_state[4], double rear_over_state[4]);
/* ref[(N+1)][3], idx[(N+1)] (idx may be NULL) */
void orc_ref_exact_points(const double* x, int32_t rows, const orc_scene* sc, double* ref, int32_t* idx);
void orc_const_velo_prediction(const orc_params* p, const double x0[4], double* x_out);
/* uses/updates the solver's ALM state when solve_type == alm */
double orc_total_cost(orc_solver* s, const double* u, const double* x, const orc_scene* sc);
/* forced recomputation (status set to RUNNING first); copies out the l_* members */
void orc_cost_derivatives(orc_solver* s, const double* u, const double* x, const orc_scene* sc,
                          double* l_x, double* l_u, double* l_xx, double* l_uu);
/* returns status (ORC_RUNNING or ORC_BACKWARD_PASS_FAIL); d[N][2], K[N][2][4], dV[2] */
int orc_backward_pass(orc_solver* s, const double* u, const double* x, double lamb,
                      const orc_scene* sc, double* d, double* K, double* dV);
void orc_forward_pass(const orc_params* p, const double* u, const double* x, const double* d,
                      const double* K, double alpha, double* new_u, double* new_x);

/* test hooks for the ALM state */
void orc_set_alm_state(orc_solver* s, const double* mu, double rho, int32_t cols);
void orc_get_alm_next(orc_solver* s, double* mu_next);

/* detmath / libm elementary functions as used by this build (for tests) */
double orc_m_exp(double x);
double orc_m_sin(double x);
double orc_m_cos(double x);
double orc_m_tan(double x);
double orc_m_atan(double x);
double orc_m_hypot(double x, double y);
void orc_m_vec(int32_t func, const double* x, const double* y, int64_t n, double* out);
double orc_alm_item(double c, double rho, double mu);                       /* hpp:81-83 */
void orc_lagrangian_derivative_and_Hessian(double c, const double* c_dot, int32_t n, double rho, double mu, double* b_dot,
                                           double* b_ddot);                  /* cs:701-713 */
#ifdef ORC_RECORD
void orc_record_math(double* buf, long cap); /* liboracle_rec.so: record (function, x, y) of every elementary-function call */
long orc_record_count(void);
#endif

#ifdef __cplusplus
}
#endif
#endif
