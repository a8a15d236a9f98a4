/*
 * cilqr_oracle.c — CPU ORACLE (test infrastructure, see cilqr_oracle.h for status and rules).
 *
 * Plain-C restatement of /root/reference/src/cilqr_solver.cpp:85-739 and
 * /root/reference/src/utils.cpp:262-439.  Citations "cs:" = src/cilqr_solver.cpp,
 * "ut:" = src/utils.cpp, "hpp:" = include/cilqr_solver.hpp, all under /root/reference.
 *
 * Matrices are dense and row-major; products accumulate in inner-index order starting from the
 * first term, evaluated left to right exactly as the reference's Eigen expressions associate.
 * Build with -ffp-contract=off.
 */
#include "cilqr_oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_DETMATH
#include "../toy-example-of-ilqr_amd/csrc/detmath.h"
#define M_EXP(x) dm_exp(x)
#define M_SIN(x) dm_sin(x)
#define M_COS(x) dm_cos(x)
#define M_TAN(x) dm_tan(x)
#define M_ATAN(x) dm_atan(x)
#define M_HYPOT(x, y) dm_hypot((x), (y))
#define M_SQRT(x) dm_sqrt(x)
int orc_math_mode(void) { return 1; }
#else
#include <math.h>
#define M_EXP(x) exp(x)
#define M_SIN(x) sin(x)
#define M_COS(x) cos(x)
#define M_TAN(x) tan(x)
#define M_ATAN(x) atan(x)
#define M_HYPOT(x, y) hypot((x), (y))
#define M_SQRT(x) sqrt(x)
int orc_math_mode(void) { return 0; }
#endif

/* Optional recorder of the ARGUMENTS the path hands to its elementary functions (liboracle_rec.so only, -DORC_RECORD: the
 * libm flavour plus this): tests/test_pins.py dumps them from real solves and measures detmath against glibc exactly
 * there.  Single-threaded use only. */
#ifdef ORC_RECORD
static double* orc_rec_buf = 0; /* [cap][3] = (function code 0..5, x, y) */
static long orc_rec_cap = 0, orc_rec_n = 0;
void orc_record_math(double* buf, long cap) { orc_rec_buf = buf; orc_rec_cap = cap; orc_rec_n = 0; }
long orc_record_count(void) { return orc_rec_n; }
static inline double orc_rec(int f, double x, double y, double r) {
    if (orc_rec_buf && orc_rec_n < orc_rec_cap) {
        double* o = orc_rec_buf + 3 * orc_rec_n;
        o[0] = (double)f; o[1] = x; o[2] = y;
    }
    orc_rec_n++;
    return r;
}
#undef M_EXP
#undef M_SIN
#undef M_COS
#undef M_TAN
#undef M_ATAN
#undef M_HYPOT
#define M_EXP(x) orc_rec(0, (x), 0.0, exp(x))
#define M_SIN(x) orc_rec(1, (x), 0.0, sin(x))
#define M_COS(x) orc_rec(2, (x), 0.0, cos(x))
#define M_TAN(x) orc_rec(3, (x), 0.0, tan(x))
#define M_ATAN(x) orc_rec(4, (x), 0.0, atan(x))
#define M_HYPOT(x, y) orc_rec(5, (x), (y), hypot((x), (y)))
#endif

#define ORC_EPS 1e-5 /* include/utils.hpp:28 */

/* The FUSED flavour (round 4, an experiment the review of round 3 asked for: what does bit-identity with an unfused
 * reference cost, and what would an explicitly fused contract inside north_star's 1e-5 buy?).  -DORC_FUSED turns the
 * multiply-adds of four NAMED groups of sites into explicit fma() — the same sites, in the same association, carry
 * CQ_MADD in csrc/cilqr_device.hpp (-DCILQR_FUSED), so the device stays bit-identical to THIS build:
 *   F1 the quadratic forms of get_total_cost (cs:211-212), F2 every product of backward_pass (cs:400-436),
 *   F3 forward_pass's K dx and + alpha d (cs:452-455), F4 the affine updates of kinematic_propagate (ut:266-281).
 * Everything else — barrier terms, cost derivatives, geometry, the elementary functions — is untouched. */
#ifdef ORC_FUSED
#define CQ_MADD(a, b, c) __builtin_fma((a), (b), (c))
int orc_fused(void) { return 1; }
#else
#define CQ_MADD(a, b, c) ((a) * (b) + (c))
int orc_fused(void) { return 0; }
#endif

double orc_m_exp(double x) { return M_EXP(x); }
double orc_m_sin(double x) { return M_SIN(x); }
double orc_m_cos(double x) { return M_COS(x); }
double orc_m_tan(double x) { return M_TAN(x); }
double orc_m_atan(double x) { return M_ATAN(x); }
double orc_m_hypot(double x, double y) { return M_HYPOT(x, y); }
/* the same, n at a time: func 0 exp, 1 sin, 2 cos, 3 tan, 4 atan, 5 hypot(x, y) */
void orc_m_vec(int32_t func, const double* x, const double* y, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) {
        const double a = x[i], b = y ? y[i] : 0.0;
        switch (func) {
            case 0: out[i] = M_EXP(a); break;
            case 1: out[i] = M_SIN(a); break;
            case 2: out[i] = M_COS(a); break;
            case 3: out[i] = M_TAN(a); break;
            case 4: out[i] = M_ATAN(a); break;
            default: out[i] = M_HYPOT(a, b); break;
        }
    }
}

struct orc_solver {
    orc_params p;
    int is_first_solve;
    int status;          /* current_solve_status */
    double* last_solve_u; /* [N][2] */
    /* cost expansion members (hpp:136-141) */
    double* l_x;  /* [N+1][4]    */
    double* l_u;  /* [N][2]      */
    double* l_xx; /* [N+1][4][4] */
    double* l_uu; /* [N][2][2]   */
    /* ALM state (hpp:106-112) */
    double alm_rho;
    double* alm_mu;      /* [N][alm_cols] */
    double* alm_mu_next; /* [N][alm_cols] */
    int alm_cols;
    int cost_evals;
    /* optional decision-margin recorder (orc_set_margin_buffer): test infrastructure on top of test
     * infrastructure — how close every discrete decision of an iteration came to going the other way */
    orc_margin_rec* mg_buf;
    int mg_cap;
    orc_margin_rec mg_cur;
    /* work space, allocated once (keeps the batch driver free of malloc traffic) */
    double *ws_ref, *ws_lub, *ws_luub, *ws_lxb, *ws_lxxb, *ws_dfdx, *ws_dfdu;
    double *ws_u, *ws_x, *ws_nu, *ws_nx, *ws_d, *ws_K;
};

static void ensure_alm(struct orc_solver* s, int cols);

/* ---------- small dense helpers: C(m x n) = A(m x k) * B(k x n), row-major ---------- */
/* ORC_SUM4 (round 6, an EXPERIMENT flag; the shipped checkers are built without it): how a four-term inner product is
 * associated.  The reference leaves that to Eigen (cs:211-212, 400-436, 449-451: products of 4 x 4 / 4 x 2 blocks and 4-vectors)
 * and Eigen is neither vendored nor pinned upstream, so which of its evaluators the reference binary runs cannot be checked here:
 *   0 (default)  index order        ((p0 + p1) + p2) + p3   Eigen's coefficient-based product without vectorisation
 *   1            adjacent pairs     (p0 + p1) + (p2 + p3)   the shape of Eigen's redux_novec_unroller (balanced tree)
 *   2            interleaved pairs  (p0 + p2) + (p1 + p3)   two-lane SSE2 packets accumulating even / odd terms, then predux
 * Two-term products are the same in every mode (a + b is commutative).  tests/sum_order_tolerance.py measures how far the
 * solves move between the modes. */
#ifndef ORC_SUM4
#define ORC_SUM4 0
#endif
int orc_sum4(void) { return ORC_SUM4; }

#if ORC_SUM4 == 1
#define ORC_SUM4_OF(p0, p1, p2, p3) (((p0) + (p1)) + ((p2) + (p3)))
#elif ORC_SUM4 == 2
#define ORC_SUM4_OF(p0, p1, p2, p3) (((p0) + (p2)) + ((p1) + (p3)))
#endif

static void matmul(const double* A, const double* B, double* C, int m, int k, int n) {
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) {
#ifdef ORC_SUM4_OF
            if (k == 4) {
                C[i * n + j] = ORC_SUM4_OF(A[i * 4] * B[j], A[i * 4 + 1] * B[n + j], A[i * 4 + 2] * B[2 * n + j],
                                           A[i * 4 + 3] * B[3 * n + j]);
                continue;
            }
#endif
            double acc = A[i * k] * B[j];
            for (int t = 1; t < k; ++t) {
                acc = acc + A[i * k + t] * B[t * n + j];
            }
            C[i * n + j] = acc;
        }
    }
}

/* the same product with the accumulation written as CQ_MADD: the sites F1-F3 (identical to matmul unless ORC_FUSED) */
static void matmul_f(const double* A, const double* B, double* C, int m, int k, int n) {
#ifdef ORC_SUM4_OF
    if (k == 4) {
        matmul(A, B, C, m, k, n);
        return;
    }
#endif
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) {
            double acc = A[i * k] * B[j];
            for (int t = 1; t < k; ++t) {
                acc = CQ_MADD(A[i * k + t], B[t * n + j], acc);
            }
            C[i * n + j] = acc;
        }
    }
}

static void transpose(const double* A, double* At, int m, int n) {
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) {
            At[j * m + i] = A[i * n + j];
        }
    }
}

/* utils::sign (include/utils.hpp:110-117): -1 for negative, +1 otherwise (also for 0 and NaN) */
static int sign_of(double v) { return (v < 0) ? -1 : 1; }

/* ---------- ut:262-283 kinematic_propagate ---------- */
void orc_kinematic_propagate(const double x[4], const double u[2], double dt, double wheelbase,
                             int32_t reference_point, double out[4]) {
    if (reference_point == 0) { /* RearCenter */
        double n0 = CQ_MADD(x[2] * M_COS(x[3]), dt, x[0]);
        double n1 = CQ_MADD(x[2] * M_SIN(x[3]), dt, x[1]);
        double n2 = CQ_MADD(u[0], dt, x[2]);
        double n3 = x[3] + x[2] * M_TAN(u[1]) * dt / wheelbase;
        out[0] = n0; out[1] = n1; out[2] = n2; out[3] = n3;
    } else { /* GravityCenter */
        double beta = M_ATAN(M_TAN(u[1]) / 2);
        double n0 = CQ_MADD(x[2] * M_COS(beta + x[3]), dt, x[0]);
        double n1 = CQ_MADD(x[2] * M_SIN(beta + x[3]), dt, x[1]);
        double n2 = CQ_MADD(u[0], dt, x[2]);
        double n3 = x[3] + 2 * x[2] * M_SIN(beta) * dt / wheelbase;
        out[0] = n0; out[1] = n1; out[2] = n2; out[3] = n3;
    }
}

/* ---------- ut:285-342 get_kinematic_model_derivatives ---------- */
void orc_model_derivatives(const double* x, const double* u, double dt, double wheelbase, int32_t N,
                           int32_t reference_point, double* A, double* B) {
    for (int i = 0; i < N; ++i) {
        double velo = x[i * 4 + 2];
        double yaw = x[i * 4 + 3];
        double delta = u[i * 2 + 1];
        double* Ai = A + i * 16;
        double* Bi = B + i * 8;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) Ai[r * 4 + c] = (r == c) ? 1.0 : 0.0;
        for (int e = 0; e < 8; ++e) Bi[e] = 0.0;
        if (reference_point == 0) {
            Ai[0 * 4 + 2] = M_COS(yaw) * dt;
            Ai[0 * 4 + 3] = velo * (-M_SIN(yaw)) * dt;
            Ai[1 * 4 + 2] = M_SIN(yaw) * dt;
            Ai[1 * 4 + 3] = velo * M_COS(yaw) * dt;
            Ai[3 * 4 + 2] = M_TAN(delta) * dt / wheelbase;
            Bi[2 * 2 + 0] = dt;
            Bi[3 * 2 + 1] = (velo * dt / wheelbase) / (M_COS(delta) * M_COS(delta));
        } else {
            /* note: beta here is atan(tan(delta/2)) (ut:291), NOT the atan(tan(delta)/2) of ut:265 */
            double beta = M_ATAN(M_TAN(delta / 2));
            double td = M_TAN(delta);
            double beta_over_stl = 0.5 * (1 + td * td) / (1 + 0.25 * (td * td));
            Ai[0 * 4 + 2] = M_COS(beta + yaw) * dt;
            Ai[0 * 4 + 3] = velo * (-M_SIN(beta + yaw)) * dt;
            Ai[1 * 4 + 2] = M_SIN(beta + yaw) * dt;
            Ai[1 * 4 + 3] = velo * M_COS(beta + yaw) * dt;
            Ai[3 * 4 + 2] = 2 * M_SIN(beta) * dt / wheelbase;
            Bi[0 * 2 + 1] = velo * (-M_SIN(beta + yaw)) * dt * beta_over_stl;
            Bi[1 * 2 + 1] = velo * M_COS(beta + yaw) * dt * beta_over_stl;
            Bi[2 * 2 + 0] = dt;
            Bi[3 * 2 + 1] = (2 * velo * dt / wheelbase) * M_COS(beta) * beta_over_stl;
        }
    }
}

/* ---------- ut:344-361 get_vehicle_front_and_rear_centers ---------- */
void orc_front_rear_centers(const double state[4], double wheelbase, int32_t reference_point,
                            double front[2], double rear[2]) {
    double yaw = state[3];
    double wv0 = wheelbase * M_COS(yaw);
    double wv1 = wheelbase * M_SIN(yaw);
    if (reference_point == 0) {
        front[0] = state[0] + wv0;
        front[1] = state[1] + wv1;
        rear[0] = state[0];
        rear[1] = state[1];
    } else {
        front[0] = state[0] + 0.5 * wv0;
        front[1] = state[1] + 0.5 * wv1;
        rear[0] = state[0] - 0.5 * wv0;
        rear[1] = state[1] - 0.5 * wv1;
    }
}

/* ---------- ut:363-385 get_vehicle_front_and_rear_center_derivatives (4x2 each) ---------- */
void orc_front_rear_center_derivatives(double yaw, double wheelbase, int32_t reference_point,
                                       double* front_over_state, double* rear_over_state) {
    double half_whba = 0.5 * wheelbase;
    double f[8] = {1, 0, 0, 1, 0, 0, half_whba * (-M_SIN(yaw)), half_whba * M_COS(yaw)};
    double r[8] = {1, 0, 0, 1, 0, 0, -half_whba * (-M_SIN(yaw)), -half_whba * M_COS(yaw)};
    if (reference_point == 0) {
        f[6] = wheelbase * (-M_SIN(yaw));
        f[7] = wheelbase * M_COS(yaw);
        r[6] = 0;
        r[7] = 0;
    }
    memcpy(front_over_state, f, sizeof(f));
    memcpy(rear_over_state, r, sizeof(r));
}

/* ---------- ut:387-393 get_ellipsoid_obstacle_scales ---------- */
void orc_ellipsoid_scales(const double obs_attr[3], double ego_pnt_radius, double ab[2]) {
    ab[0] = 0.5 * obs_attr[1] + obs_attr[2] * 6 + ego_pnt_radius;
    ab[1] = 0.5 * obs_attr[0] + obs_attr[2] + ego_pnt_radius;
}

/* ---------- ut:395-407 ellipsoid_safety_margin ---------- */
double orc_ellipsoid_safety_margin(const double pnt[2], const double obs_state[3], const double ab[2]) {
    double theta = obs_state[2];
    double d0 = pnt[0] - obs_state[0];
    double d1 = pnt[1] - obs_state[1];
    double c = M_COS(theta), s = M_SIN(theta);
    double p0 = c * d0 + s * d1;
    double p1 = (-s) * d0 + c * d1;
    return 1 - ((p0 * p0) / (ab[0] * ab[0]) + (p1 * p1) / (ab[1] * ab[1]));
}

/* ---------- ut:409-439 ellipsoid_safety_margin_derivatives ---------- */
void orc_ellipsoid_safety_margin_derivatives(const double pnt[2], const double obs_state[3],
                                             const double ab[2], double out[2]) {
    double theta = obs_state[2];
    double d0 = pnt[0] - obs_state[0];
    double d1 = pnt[1] - obs_state[1];
    double c = M_COS(theta), s = M_SIN(theta);
    double rot[4] = {c, s, -s, c};
    double p0 = rot[0] * d0 + rot[1] * d1;
    double p1 = rot[2] * d0 + rot[3] * d1;
    double res_over_pnt_std[2] = {-2 * p0 / (ab[0] * ab[0]), -2 * p1 / (ab[1] * ab[1])};
    double pnt_std_over_diff[4];
    transpose(rot, pnt_std_over_diff, 2, 2);
    double diff_over_pnt[4] = {1, 0, 0, 1};
    double tmp[4];
    matmul(diff_over_pnt, pnt_std_over_diff, tmp, 2, 2, 2);
    matmul(tmp, res_over_pnt_std, out, 2, 2, 1);
}

/* ---------- hpp:80 exp_barrier, hpp:81-83 augmented_lagrangian_item ---------- */
double orc_exp_barrier(double c, double q1, double q2) { return q1 * M_EXP(q2 * c); }

static double augmented_lagrangian_item(double c, double rho, double mu) {
    double t = c + mu / rho;
    double m = (t > 0.0) ? t : 0.0; /* std::max(c + mu/rho, 0.0) */
    return rho * (m * m) / 2;
}

/* ---------- cs:692-699 exp_barrier_derivative_and_Hessian ---------- */
void orc_exp_barrier_derivative_and_Hessian(double c, const double* c_dot, int32_t n, double q1,
                                            double q2, double* b_dot, double* b_ddot) {
    double b = orc_exp_barrier(c, q1, q2);
    for (int i = 0; i < n; ++i) b_dot[i] = q2 * b * c_dot[i];
    double s = (q2 * q2) * b;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) b_ddot[i * n + j] = s * (c_dot[i] * c_dot[j]);
}

/* ---------- cs:701-713 lagrangian_derivative_and_Hessian ---------- */
static void lagrangian_derivative_and_Hessian(double c, const double* c_dot, int n, double rho,
                                              double mu, double* b_dot, double* b_ddot) {
    for (int i = 0; i < n; ++i) b_dot[i] = 0.0;
    for (int i = 0; i < n * n; ++i) b_ddot[i] = 0.0;
    if ((c + mu / rho) > 0) {
        double s = rho * (c + mu / rho);
        for (int i = 0; i < n; ++i) b_dot[i] = s * c_dot[i];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) b_ddot[i * n + j] = b_dot[i] * c_dot[j];
    }
}

/* the two ALM leaf functions, exported for tests/test_pins.py */
double orc_alm_item(double c, double rho, double mu) { return augmented_lagrangian_item(c, rho, mu); }
void orc_lagrangian_derivative_and_Hessian(double c, const double* c_dot, int32_t n, double rho, double mu, double* b_dot,
                                           double* b_ddot) {
    lagrangian_derivative_and_Hessian(c, c_dot, n, rho, mu, b_dot, b_ddot);
}

/* ---------- cs:316-324 get_bound_constr ---------- */
static double bound_upper(double v, double b) { return v - b; }
static double bound_lower(double v, double b) { return b - v; }

/* ---------- cs:326-335 get_obstacle_avoidance_constr ---------- */
void orc_obstacle_constr(const orc_params* p, const double ego[4], const double obs[3], double out[2]) {
    double front[2], rear[2], ab[2];
    double obs_attr[3] = {p->width, p->length, p->d_safe}; /* cs:78 */
    orc_front_rear_centers(ego, p->wheelbase, p->reference_point, front, rear);
    orc_ellipsoid_scales(obs_attr, 0.5 * p->width, ab);
    out[0] = orc_ellipsoid_safety_margin(front, obs, ab);
    out[1] = orc_ellipsoid_safety_margin(rear, obs, ab);
}

/* ---------- cs:715-739 get_obstacle_avoidance_constr_derivatives ---------- */
void orc_obstacle_constr_derivatives(const orc_params* p, const double ego[4], const double obs[3],
                                     double front_over_state[4], double rear_over_state[4]) {
    double front[2], rear[2], ab[2], gf[2], gr[2], Ff[8], Fr[8];
    double obs_attr[3] = {p->width, p->length, p->d_safe};
    orc_front_rear_centers(ego, p->wheelbase, p->reference_point, front, rear);
    orc_ellipsoid_scales(obs_attr, 0.5 * p->width, ab);
    orc_ellipsoid_safety_margin_derivatives(front, obs, ab, gf);
    orc_ellipsoid_safety_margin_derivatives(rear, obs, ab, gr);
    orc_front_rear_center_derivatives(ego[3], p->wheelbase, p->reference_point, Ff, Fr);
    matmul(Ff, gf, front_over_state, 4, 2, 1);
    matmul(Fr, gr, rear_over_state, 4, 2, 1);
}

/* ---------- cs:289-314 get_ref_exact_points ---------- */
static void mg_note(double* slot, double m) {
    if (!(m >= *slot)) *slot = m; /* also takes NaN: a NaN decision is as fragile as it gets */
}

static void ref_exact_points_mg(const double* x, int32_t rows, const orc_scene* sc, double* ref, int32_t* idx,
                                double* mg_ref) {
    int start_index = 0;
    for (int i = 0; i < rows; ++i) {
        int min_idx = -1;
        double min_distance = DBL_MAX;
        for (int j = start_index; j < sc->L; ++j) {
            double cur = M_HYPOT(x[i * 4 + 0] - sc->lane_x[j], x[i * 4 + 1] - sc->lane_y[j]);
            if (mg_ref && min_idx >= 0) {
                /* the comparison cur < min_distance, relative to the distances compared */
                double sc_ = (min_distance > cur) ? min_distance : cur;
                double df = cur - min_distance;
                mg_note(mg_ref, ((df < 0) ? -df : df) / ((sc_ > 1e-300) ? sc_ : 1e-300));
            }
            if (min_idx < 0 || cur < min_distance) {
                min_idx = j;
                min_distance = cur;
            } else {
                break;
            }
        }
        ref[i * 3 + 0] = sc->lane_x[min_idx];
        ref[i * 3 + 1] = sc->lane_y[min_idx];
        ref[i * 3 + 2] = sc->lane_yaw[min_idx];
        if (idx) idx[i] = min_idx;
        start_index = min_idx;
    }
}

void orc_ref_exact_points(const double* x, int32_t rows, const orc_scene* sc, double* ref, int32_t* idx) {
    ref_exact_points_mg(x, rows, sc, ref, idx, NULL);
}

static const double* obs_at(const orc_scene* sc, int j, int k) {
    return sc->obs + ((size_t)j * sc->T + (size_t)(sc->tick + k)) * 3;
}

/* ---------- cs:182-197 const_velo_prediction ---------- */
void orc_const_velo_prediction(const orc_params* p, const double x0[4], double* x_out) {
    double cur_u[2] = {0.0, 0.0};
    memcpy(x_out, x0, 4 * sizeof(double));
    for (int i = 0; i < p->N; ++i) {
        orc_kinematic_propagate(x_out + i * 4, cur_u, p->dt, p->wheelbase, p->reference_point,
                                x_out + (i + 1) * 4);
    }
}

/* ---------- cs:199-287 get_total_cost ---------- */
double orc_total_cost(orc_solver* s, const double* u, const double* x, const orc_scene* sc) {
    const orc_params* p = &s->p;
    const int N = p->N;
    const int M = sc->M;
    s->cost_evals++;
    if (p->solve_type == 1) ensure_alm(s, 8 + 2 * M);
    double* ref = s->ws_ref;
    ref_exact_points_mg(x, N + 1, sc, ref, NULL, s->mg_buf ? &s->mg_cur.ref_margin : NULL);

    double W[16] = {0}, R[4] = {0};
    W[0] = p->w_pos; W[5] = p->w_pos; W[10] = p->w_vel; W[15] = p->w_yaw; /* cs:23-27 */
    R[0] = p->w_acc; R[3] = p->w_stl;                                      /* cs:28-30 */

    /* part 1 (cs:205-213): trace((x-ref) W (x-ref)^T) + trace(u R u^T) */
    double states_devt = 0.0;
    for (int k = 0; k <= N; ++k) {
        double e[4] = {x[k * 4 + 0] - ref[k * 3 + 0], x[k * 4 + 1] - ref[k * 3 + 1],
                       x[k * 4 + 2] - sc->ref_velo, x[k * 4 + 3] - ref[k * 3 + 2]};
        double t[4], dk;
        matmul_f(e, W, t, 1, 4, 4);
        matmul_f(t, e, &dk, 1, 4, 1);
        states_devt = (k == 0) ? dk : (states_devt + dk);
    }
    double ctrl_energy = 0.0;
    for (int k = 0; k < N; ++k) {
        double t[2], dk;
        matmul_f(u + k * 2, R, t, 1, 2, 2);
        matmul_f(t, u + k * 2, &dk, 1, 2, 1);
        ctrl_energy = (k == 0) ? dk : (ctrl_energy + dk);
    }
    double J_prime = states_devt + ctrl_energy;

    /* part 2 (cs:215-283) */
    double J_barrier = 0.;
    for (int k = 1; k < N + 1; ++k) {
        const double* u_k = u + (k - 1) * 2;
        const double* x_k = x + k * 4;
        const double* ref_x_k = ref + k * 3;

        double acc_up_constr = bound_upper(u_k[0], p->acc_max);
        double acc_lo_constr = bound_lower(u_k[0], p->acc_min);
        double stl_up_constr = bound_upper(u_k[1], p->stl_lim);
        double stl_lo_constr = bound_lower(u_k[1], -p->stl_lim);
        double velo_up_constr = bound_upper(x_k[2], p->velo_max);
        double velo_lo_constr = bound_lower(x_k[2], p->velo_min);

        double d_sign = (x_k[1] - ref_x_k[1]) * M_COS(ref_x_k[2]) - (x_k[0] - ref_x_k[0]) * M_SIN(ref_x_k[2]);
        double cur_d = sign_of(d_sign) * M_HYPOT(x_k[0] - ref_x_k[0], x_k[1] - ref_x_k[1]);
        double pos_up_constr = bound_upper(cur_d, sc->road_borders[0] - p->width / 2);
        double pos_lo_constr = bound_lower(cur_d, sc->road_borders[1] + p->width / 2);

        double J_barrier_k = 0.0;
        if (p->solve_type == 0) {
            J_barrier_k = orc_exp_barrier(acc_up_constr, p->state_exp_q1, p->state_exp_q2) +
                          orc_exp_barrier(acc_lo_constr, p->state_exp_q1, p->state_exp_q2) +
                          orc_exp_barrier(stl_up_constr, p->state_exp_q1, p->state_exp_q2) +
                          orc_exp_barrier(stl_lo_constr, p->state_exp_q1, p->state_exp_q2) +
                          orc_exp_barrier(velo_up_constr, p->state_exp_q1, p->state_exp_q2) +
                          orc_exp_barrier(velo_lo_constr, p->state_exp_q1, p->state_exp_q2) +
                          orc_exp_barrier(pos_up_constr, p->state_exp_q1, p->state_exp_q2) +
                          orc_exp_barrier(pos_lo_constr, p->state_exp_q1, p->state_exp_q2);
        } else {
            const double* mu = s->alm_mu + (size_t)(k - 1) * s->alm_cols;
            J_barrier_k = augmented_lagrangian_item(acc_up_constr, s->alm_rho, mu[0]) +
                          augmented_lagrangian_item(acc_lo_constr, s->alm_rho, mu[1]) +
                          augmented_lagrangian_item(stl_up_constr, s->alm_rho, mu[2]) +
                          augmented_lagrangian_item(stl_lo_constr, s->alm_rho, mu[3]) +
                          augmented_lagrangian_item(velo_up_constr, s->alm_rho, mu[4]) +
                          augmented_lagrangian_item(velo_lo_constr, s->alm_rho, mu[5]) +
                          augmented_lagrangian_item(pos_up_constr, s->alm_rho, mu[6]) +
                          augmented_lagrangian_item(pos_lo_constr, s->alm_rho, mu[7]);
        }
        for (int j = 0; j < M; ++j) {
            double c2[2];
            orc_obstacle_constr(p, x_k, obs_at(sc, j, k), c2);
            if (p->solve_type == 0) {
                J_barrier_k += orc_exp_barrier(c2[0], p->obstacle_exp_q1, p->obstacle_exp_q2);
                J_barrier_k += orc_exp_barrier(c2[1], p->obstacle_exp_q1, p->obstacle_exp_q2);
            } else {
                const double* mu = s->alm_mu + (size_t)(k - 1) * s->alm_cols;
                J_barrier_k += augmented_lagrangian_item(c2[0], s->alm_rho, mu[8 + 2 * j]);
                J_barrier_k += augmented_lagrangian_item(c2[1], s->alm_rho, mu[9 + 2 * j]);
            }
        }
        J_barrier += J_barrier_k;
    }
    return J_prime + J_barrier;
}

/* ---------- cs:463-690 get_total_cost_derivatives_and_Hessians ---------- */
static void add_vec(double* dst, const double* a, int n) {
    for (int i = 0; i < n; ++i) dst[i] = dst[i] + a[i];
}

static void cost_derivatives_and_Hessians(orc_solver* s, const double* u, const double* x,
                                          const orc_scene* sc) {
    const orc_params* p = &s->p;
    const int N = p->N;
    const int M = sc->M;
    if (p->solve_type == 1) ensure_alm(s, 8 + 2 * M);
    /* cs:469-475: after a failed pass the trajectory is unchanged, keep the previous expansion */
    if (p->solve_type == 0 && s->status != ORC_RUNNING && s->status != ORC_FORWARD_PASS_SMALL_STEP) {
        s->status = ORC_RUNNING;
        return;
    }
    s->status = ORC_RUNNING;

    double* ref = s->ws_ref;
    ref_exact_points_mg(x, N + 1, sc, ref, NULL, s->mg_buf ? &s->mg_cur.ref_margin : NULL);

    double W[16] = {0}, R[4] = {0};
    W[0] = p->w_pos; W[5] = p->w_pos; W[10] = p->w_vel; W[15] = p->w_yaw;
    R[0] = p->w_acc; R[3] = p->w_stl;

    double* l_u_barrier = s->ws_lub;
    double* l_uu_barrier = s->ws_luub;
    double* l_x_barrier = s->ws_lxb;
    double* l_xx_barrier = s->ws_lxxb;
    memset(l_u_barrier, 0, sizeof(double) * (size_t)N * 2);
    memset(l_uu_barrier, 0, sizeof(double) * (size_t)N * 4);
    memset(l_x_barrier, 0, sizeof(double) * (size_t)(N + 1) * 4);
    memset(l_xx_barrier, 0, sizeof(double) * (size_t)(N + 1) * 16);

    for (int k = 1; k < N + 1; ++k) {
        const double* u_k = u + (k - 1) * 2;
        const double* x_k = x + k * 4;
        const double* ref_x_k = ref + k * 3;

        double d_sign = (x_k[1] - ref_x_k[1]) * M_COS(ref_x_k[2]) - (x_k[0] - ref_x_k[0]) * M_SIN(ref_x_k[2]);
        double cur_d = sign_of(d_sign) * M_HYPOT(x_k[0] - ref_x_k[0], x_k[1] - ref_x_k[1]);
        double acc_up_constr = bound_upper(u_k[0], p->acc_max);
        double acc_lo_constr = bound_lower(u_k[0], p->acc_min);
        double stl_up_constr = bound_upper(u_k[1], p->stl_lim);
        double stl_lo_constr = bound_lower(u_k[1], -p->stl_lim);
        double velo_up_constr = bound_upper(x_k[2], p->velo_max);
        double velo_lo_constr = bound_lower(x_k[2], p->velo_min);
        double pos_up_constr = bound_upper(cur_d, sc->road_borders[0] - p->width / 2);
        double pos_lo_constr = bound_lower(cur_d, sc->road_borders[1] + p->width / 2);

        double acc_up_over_u[2] = {1.0, 0.0};
        double acc_lo_over_u[2] = {-1, 0};
        double stl_up_over_u[2] = {0.0, 1.0};
        double stl_lo_over_u[2] = {0, -1.0};
        double velo_up_over_x[4] = {0, 0, 1, 0};
        double velo_lo_over_x[4] = {0, 0, -1, 0};
        double pos_up_over_x[4] = {
            (x_k[0] - ref_x_k[0]) / M_HYPOT(x_k[0] - ref_x_k[0], x_k[1] - ref_x_k[1]),
            (x_k[1] - ref_x_k[1]) / M_HYPOT(x_k[0] - ref_x_k[0], x_k[1] - ref_x_k[1]), 0, 0};
        if (d_sign < 0) {
            for (int i = 0; i < 4; ++i) pos_up_over_x[i] = -1 * pos_up_over_x[i];
        }
        double pos_lo_over_x[4];
        for (int i = 0; i < 4; ++i) pos_lo_over_x[i] = -1 * pos_up_over_x[i];

        double bu[4][2], buu[4][4];   /* acc_up, acc_lo, stl_up, stl_lo */
        double bx[4][4], bxx[4][16];  /* velo_up, velo_lo, pos_up, pos_lo */
        double* mu_next = NULL;
        const double* mu = NULL;
        if (p->solve_type == 0) {
            const double q1 = p->state_exp_q1, q2 = p->state_exp_q2;
            orc_exp_barrier_derivative_and_Hessian(acc_up_constr, acc_up_over_u, 2, q1, q2, bu[0], buu[0]);
            orc_exp_barrier_derivative_and_Hessian(acc_lo_constr, acc_lo_over_u, 2, q1, q2, bu[1], buu[1]);
            orc_exp_barrier_derivative_and_Hessian(stl_up_constr, stl_up_over_u, 2, q1, q2, bu[2], buu[2]);
            orc_exp_barrier_derivative_and_Hessian(stl_lo_constr, stl_lo_over_u, 2, q1, q2, bu[3], buu[3]);
            orc_exp_barrier_derivative_and_Hessian(velo_up_constr, velo_up_over_x, 4, q1, q2, bx[0], bxx[0]);
            orc_exp_barrier_derivative_and_Hessian(velo_lo_constr, velo_lo_over_x, 4, q1, q2, bx[1], bxx[1]);
            orc_exp_barrier_derivative_and_Hessian(pos_up_constr, pos_up_over_x, 4, q1, q2, bx[2], bxx[2]);
            orc_exp_barrier_derivative_and_Hessian(pos_lo_constr, pos_lo_over_x, 4, q1, q2, bx[3], bxx[3]);
        } else {
            mu = s->alm_mu + (size_t)(k - 1) * s->alm_cols;
            mu_next = s->alm_mu_next + (size_t)(k - 1) * s->alm_cols;
            const double rho = s->alm_rho;
            lagrangian_derivative_and_Hessian(acc_up_constr, acc_up_over_u, 2, rho, mu[0], bu[0], buu[0]);
            lagrangian_derivative_and_Hessian(acc_lo_constr, acc_lo_over_u, 2, rho, mu[1], bu[1], buu[1]);
            lagrangian_derivative_and_Hessian(stl_up_constr, stl_up_over_u, 2, rho, mu[2], bu[2], buu[2]);
            lagrangian_derivative_and_Hessian(stl_lo_constr, stl_lo_over_u, 2, rho, mu[3], bu[3], buu[3]);
            lagrangian_derivative_and_Hessian(velo_up_constr, velo_up_over_x, 4, rho, mu[4], bx[0], bxx[0]);
            lagrangian_derivative_and_Hessian(velo_lo_constr, velo_lo_over_x, 4, rho, mu[5], bx[1], bxx[1]);
            lagrangian_derivative_and_Hessian(pos_up_constr, pos_up_over_x, 4, rho, mu[6], bx[2], bxx[2]);
            lagrangian_derivative_and_Hessian(pos_lo_constr, pos_lo_over_x, 4, rho, mu[7], bx[3], bxx[3]);
            /* cs:622-637 */
            const double cs8[8] = {acc_up_constr, acc_lo_constr, stl_up_constr, stl_lo_constr,
                                   velo_up_constr, velo_lo_constr, pos_up_constr, pos_lo_constr};
            for (int c = 0; c < 8; ++c) {
                double v = mu[c] + rho * cs8[c];
                v = (v > 0.0) ? v : 0.0;       /* std::max(v, 0.0) */
                v = (p->max_mu < v) ? p->max_mu : v; /* std::min(v, max_mu) */
                mu_next[c] = v;
            }
        }
        /* cs:553-558 / 599-604: sums in source order */
        double* lub = l_u_barrier + (k - 1) * 2;
        double* luub = l_uu_barrier + (k - 1) * 4;
        for (int i = 0; i < 2; ++i) lub[i] = bu[0][i] + bu[1][i] + bu[2][i] + bu[3][i];
        for (int i = 0; i < 4; ++i) luub[i] = buu[0][i] + buu[1][i] + buu[2][i] + buu[3][i];
        /* cs:576-580 / 639-643 */
        double* lxb = l_x_barrier + k * 4;
        double* lxxb = l_xx_barrier + k * 16;
        for (int i = 0; i < 4; ++i) lxb[i] = bx[0][i] + bx[1][i] + bx[2][i] + bx[3][i];
        for (int i = 0; i < 16; ++i) lxxb[i] = bxx[0][i] + bxx[1][i] + bxx[2][i] + bxx[3][i];

        /* cs:647-683 obstacle terms */
        for (int j = 0; j < M; ++j) {
            const double* ob = obs_at(sc, j, k);
            double c2[2], fos[4], ros[4], fb[4], fbb[16], rb[4], rbb[16], sumv[4], summ[16];
            orc_obstacle_constr(p, x_k, ob, c2);
            orc_obstacle_constr_derivatives(p, x_k, ob, fos, ros);
            if (p->solve_type == 0) {
                orc_exp_barrier_derivative_and_Hessian(c2[0], fos, 4, p->obstacle_exp_q1, p->obstacle_exp_q2, fb, fbb);
                orc_exp_barrier_derivative_and_Hessian(c2[1], ros, 4, p->obstacle_exp_q1, p->obstacle_exp_q2, rb, rbb);
            } else {
                const double rho = s->alm_rho;
                lagrangian_derivative_and_Hessian(c2[0], fos, 4, rho, mu[8 + 2 * j], fb, fbb);
                lagrangian_derivative_and_Hessian(c2[1], ros, 4, rho, mu[9 + 2 * j], rb, rbb);
                for (int c = 0; c < 2; ++c) {
                    double v = mu[8 + 2 * j + c] + rho * c2[c];
                    v = (v > 0.0) ? v : 0.0;
                    v = (p->max_mu < v) ? p->max_mu : v;
                    mu_next[8 + 2 * j + c] = v;
                }
            }
            for (int i = 0; i < 4; ++i) sumv[i] = fb[i] + rb[i];
            for (int i = 0; i < 16; ++i) summ[i] = fbb[i] + rbb[i];
            add_vec(lxb, sumv, 4);
            add_vec(lxxb, summ, 16);
        }
    }

    /* cs:491-494 prime parts and cs:686-689 totals */
    for (int k = 0; k < N; ++k) {
        double t[2];
        matmul(u + k * 2, R, t, 1, 2, 2);
        for (int i = 0; i < 2; ++i) s->l_u[k * 2 + i] = 2 * t[i] + l_u_barrier[k * 2 + i];
        for (int i = 0; i < 4; ++i) s->l_uu[k * 4 + i] = 2 * R[i] + l_uu_barrier[k * 4 + i];
    }
    for (int k = 0; k <= N; ++k) {
        double e2[4] = {2 * (x[k * 4 + 0] - ref[k * 3 + 0]), 2 * (x[k * 4 + 1] - ref[k * 3 + 1]),
                        2 * (x[k * 4 + 2] - sc->ref_velo), 2 * (x[k * 4 + 3] - ref[k * 3 + 2])};
        double t[4];
        matmul(e2, W, t, 1, 4, 4);
        for (int i = 0; i < 4; ++i) s->l_x[k * 4 + i] = t[i] + l_x_barrier[k * 4 + i];
        for (int i = 0; i < 16; ++i) s->l_xx[k * 16 + i] = 2 * W[i] + l_xx_barrier[k * 16 + i];
    }
}

void orc_cost_derivatives(orc_solver* s, const double* u, const double* x, const orc_scene* sc,
                          double* l_x, double* l_u, double* l_xx, double* l_uu) {
    const int N = s->p.N;
    s->status = ORC_RUNNING;
    cost_derivatives_and_Hessians(s, u, x, sc);
    memcpy(l_x, s->l_x, sizeof(double) * 4 * (N + 1));
    memcpy(l_u, s->l_u, sizeof(double) * 2 * N);
    memcpy(l_xx, s->l_xx, sizeof(double) * 16 * (N + 1));
    memcpy(l_uu, s->l_uu, sizeof(double) * 4 * N);
}

/* ---------- cs:383-440 backward_pass ---------- */
static int backward_pass(orc_solver* s, const double* u, const double* x, double lamb,
                         const orc_scene* sc, double* d, double* K, double* delta_V) {
    const orc_params* p = &s->p;
    const int N = p->N;
    cost_derivatives_and_Hessians(s, u, x, sc);
    double* df_dx = s->ws_dfdx;
    double* df_du = s->ws_dfdu;
    orc_model_derivatives(x, u, p->dt, p->wheelbase, N, p->reference_point, df_dx, df_du);

    delta_V[0] = 0.0;
    delta_V[1] = 0.0;
    memset(d, 0, sizeof(double) * 2 * N);
    memset(K, 0, sizeof(double) * 8 * N);

    double V_x[4], V_xx[16];
    memcpy(V_x, s->l_x + N * 4, sizeof(V_x));
    memcpy(V_xx, s->l_xx + N * 16, sizeof(V_xx));

    for (int i = N - 1; i >= 0; --i) {
        const double* A = df_dx + i * 16; /* 4x4 */
        const double* B = df_du + i * 8;  /* 4x2 */
        double At[16], Bt[8];
        transpose(A, At, 4, 4);
        transpose(B, Bt, 4, 2); /* 2x4 */

        double tmp4[4], tmp2[2], AtV[16], BtV[8], prod16[16], prod4[4], prod8[8];
        double Q_x[4], Q_u[2], Q_xx[16], Q_uu[4], Q_ux[8];

        matmul_f(At, V_x, tmp4, 4, 4, 1);
        for (int e = 0; e < 4; ++e) Q_x[e] = s->l_x[i * 4 + e] + tmp4[e];
        matmul_f(Bt, V_x, tmp2, 2, 4, 1);
        for (int e = 0; e < 2; ++e) Q_u[e] = s->l_u[i * 2 + e] + tmp2[e];
        matmul_f(At, V_xx, AtV, 4, 4, 4);
        matmul_f(AtV, A, prod16, 4, 4, 4);
        for (int e = 0; e < 16; ++e) Q_xx[e] = s->l_xx[i * 16 + e] + prod16[e];
        matmul_f(Bt, V_xx, BtV, 2, 4, 4);
        matmul_f(BtV, B, prod4, 2, 4, 2);
        for (int e = 0; e < 4; ++e) {
            double lam_e = (e == 0 || e == 3) ? lamb * 1.0 : lamb * 0.0;
            Q_uu[e] = (s->l_uu[i * 4 + e] + prod4[e]) + lam_e;
        }
        matmul_f(BtV, A, prod8, 2, 4, 4);
        for (int e = 0; e < 8; ++e) Q_ux[e] = 0.0 + prod8[e]; /* l_ux is zero (cs:79-80) */

        /* Eigen::LLT on the lower triangle (cs:415-420): fails iff a pivot is <= 0 */
        {
            double piv0 = Q_uu[0];
            int fail = 0;
            if (s->mg_buf) {
                /* piv0 <= 0 relative to the terms it is the sum of */
                double a0 = s->l_uu[i * 4 + 0], b0 = prod4[0];
                double sc0 = ((a0 < 0) ? -a0 : a0) + ((b0 < 0) ? -b0 : b0) + ((lamb < 0) ? -lamb : lamb);
                mg_note(&s->mg_cur.pd_margin, ((piv0 < 0) ? -piv0 : piv0) / ((sc0 > 1e-300) ? sc0 : 1e-300));
            }
            if (piv0 <= 0.0) {
                fail = 1;
            } else {
                double l00 = M_SQRT(piv0);
                double l10 = Q_uu[2] / l00;
                double piv1 = Q_uu[3] - l10 * l10;
                if (s->mg_buf) {
                    double q3 = (Q_uu[3] < 0) ? -Q_uu[3] : Q_uu[3];
                    double sc1 = q3 + l10 * l10;
                    mg_note(&s->mg_cur.pd_margin, ((piv1 < 0) ? -piv1 : piv1) / ((sc1 > 1e-300) ? sc1 : 1e-300));
                }
                if (piv1 <= 0.0) fail = 1;
            }
            if (fail) {
                s->status = ORC_BACKWARD_PASS_FAIL;
                return s->status;
            }
        }
        /* Matrix2d::inverse() (cs:421): adjugate times 1/det */
        double det = Q_uu[0] * Q_uu[3] - Q_uu[2] * Q_uu[1];
        double invdet = 1.0 / det;
        double Q_uu_inv[4] = {Q_uu[3] * invdet, -Q_uu[1] * invdet, -Q_uu[2] * invdet, Q_uu[0] * invdet};
        double neg_inv[4] = {-Q_uu_inv[0], -Q_uu_inv[1], -Q_uu_inv[2], -Q_uu_inv[3]};

        double* d_i = d + i * 2;
        double* K_i = K + i * 8; /* 2x4 */
        matmul_f(neg_inv, Q_u, d_i, 2, 2, 1);
        matmul_f(neg_inv, Q_ux, K_i, 2, 2, 4);

        /* cs:427-432 value function update */
        double Kt[8], Quxt[8], KtQuu[8], t4a[4], t4b[4], t4c[4], t16a[16], t16b[16], t16c[16];
        transpose(K_i, Kt, 2, 4);    /* 4x2 */
        transpose(Q_ux, Quxt, 2, 4); /* 4x2 */
        matmul_f(Kt, Q_uu, KtQuu, 4, 2, 2);
        matmul_f(KtQuu, d_i, t4a, 4, 2, 1);
        matmul_f(Kt, Q_u, t4b, 4, 2, 1);
        matmul_f(Quxt, d_i, t4c, 4, 2, 1);
        for (int e = 0; e < 4; ++e) V_x[e] = ((Q_x[e] + t4a[e]) + t4b[e]) + t4c[e];
        matmul_f(KtQuu, K_i, t16a, 4, 2, 4);
        matmul_f(Kt, Q_ux, t16b, 4, 2, 4);
        matmul_f(Quxt, K_i, t16c, 4, 2, 4);
        for (int e = 0; e < 16; ++e) V_xx[e] = ((Q_xx[e] + t16a[e]) + t16b[e]) + t16c[e];

        /* cs:435-436 expected cost reduction */
        double hd[2] = {0.5 * d_i[0], 0.5 * d_i[1]};
        double hdQ[2], q0, q1;
        matmul_f(hd, Q_uu, hdQ, 1, 2, 2);
        matmul_f(hdQ, d_i, &q0, 1, 2, 1);
        matmul_f(d_i, Q_u, &q1, 1, 2, 1);
        delta_V[0] += q0;
        delta_V[1] += q1;
    }
    return s->status;
}

int orc_backward_pass(orc_solver* s, const double* u, const double* x, double lamb,
                      const orc_scene* sc, double* d, double* K, double* dV) {
    s->status = ORC_RUNNING;
    return backward_pass(s, u, x, lamb, sc, d, K, dV);
}

/* ---------- cs:442-461 forward_pass ---------- */
void orc_forward_pass(const orc_params* p, const double* u, const double* x, const double* d,
                      const double* K, double alpha, double* new_u, double* new_x) {
    const int N = p->N;
    memcpy(new_x, x, 4 * sizeof(double));
    for (int i = 0; i < N; ++i) {
        double dx[4], Kdx[2];
        for (int e = 0; e < 4; ++e) dx[e] = new_x[i * 4 + e] - x[i * 4 + e];
        matmul_f(K + i * 8, dx, Kdx, 2, 4, 1);
        for (int e = 0; e < 2; ++e) new_u[i * 2 + e] = CQ_MADD(alpha, d[i * 2 + e], u[i * 2 + e] + Kdx[e]);
        orc_kinematic_propagate(new_x + i * 4, new_u + i * 2, p->dt, p->wheelbase, p->reference_point,
                                new_x + (i + 1) * 4);
    }
}

/* ---------- cs:337-381 iter_step ---------- */
typedef struct iter_out {
    double new_J;
    int trials;
    int alpha_idx;
} iter_out;

static iter_out iter_step(orc_solver* s, const double* u, const double* x, double lamb,
                          const orc_scene* sc, int* effective_flag, double* new_u, double* new_x,
                          double* d, double* K) {
    const orc_params* p = &s->p;
    const int N = p->N;
    iter_out out = {0.0, 0, -1};
    double delta_item[2];
    double ori_cost = orc_total_cost(s, u, x, sc);
    backward_pass(s, u, x, lamb, sc, d, K, delta_item);
    if (s->status == ORC_BACKWARD_PASS_FAIL) {
        memcpy(new_u, u, sizeof(double) * 2 * N);
        memcpy(new_x, x, sizeof(double) * 4 * (N + 1));
        out.new_J = ori_cost;
        return out;
    }
    double new_J = DBL_MAX;
    *effective_flag = 0;
    int idx = 0;
    for (double alpha = 1.0; alpha > 1e-6; alpha *= 0.5, ++idx) {
        orc_forward_pass(p, u, x, d, K, alpha, new_u, new_x);
        new_J = orc_total_cost(s, new_u, new_x, sc);
        out.trials++;
        const double actual_cost_decay = ori_cost - new_J;
        double am1 = alpha - 1.0;
        am1 = (am1 < 0) ? -am1 : am1;
        double acd_abs = (actual_cost_decay < 0) ? -actual_cost_decay : actual_cost_decay;
        if (s->mg_buf) {
            /* every comparison of the verdict on this trial, in evaluation order, relative to the cost level the
             * compared numbers are differences of */
            double lvl = (ori_cost < 0) ? -ori_cost : ori_cost;
            lvl = (lvl > 1e-300) ? lvl : 1e-300;
            double m;
            if (am1 < ORC_EPS) {
                m = acd_abs - p->convergence_threshold;
                mg_note(&s->mg_cur.ls_margin, ((m < 0) ? -m : m) / lvl);
            }
            if (!(am1 < ORC_EPS && acd_abs < p->convergence_threshold)) {
                mg_note(&s->mg_cur.ls_margin, acd_abs / lvl); /* actual_cost_decay > 0 */
                if (actual_cost_decay > 0.0) {
                    double apx = -(alpha * alpha * delta_item[0] + alpha * delta_item[1]);
                    mg_note(&s->mg_cur.ls_margin, ((apx < 0) ? -apx : apx) / lvl); /* approx < 0 */
                    if (!(apx < 0.0)) {
                        m = actual_cost_decay - p->accept_step_threshold * apx; /* decay / approx > threshold */
                        mg_note(&s->mg_cur.ls_margin, ((m < 0) ? -m : m) / lvl);
                    }
                }
            }
        }
        if (am1 < ORC_EPS && acd_abs < p->convergence_threshold) {
            s->status = ORC_CONVERGED;
            out.new_J = new_J;
            out.alpha_idx = idx;
            return out;
        }
        double approx_cost_decay = -(alpha * alpha * delta_item[0] + alpha * delta_item[1]);
        if (actual_cost_decay > 0.0 &&
            (approx_cost_decay < 0.0 || actual_cost_decay / approx_cost_decay > p->accept_step_threshold)) {
            if (am1 > ORC_EPS) {
                s->status = ORC_FORWARD_PASS_SMALL_STEP;
            }
            *effective_flag = 1;
            out.new_J = new_J;
            out.alpha_idx = idx;
            return out;
        }
    }
    /* cs:377-378 (only meaningful in ALM mode; in BARRIER mode these members are never read) */
    if (p->solve_type == 1) {
        memcpy(s->alm_mu, s->alm_mu_next, sizeof(double) * (size_t)N * s->alm_cols);
        double r = (1 + p->alm_gamma) * s->alm_rho;
        s->alm_rho = (p->max_rho < r) ? p->max_rho : r;
    }
    s->status = ORC_FORWARD_PASS_FAIL;
    out.new_J = new_J;
    return out;
}

/* ---------- ctor cs:17-83 ---------- */
orc_solver* orc_create(const orc_params* p) {
    orc_solver* s = (orc_solver*)calloc(1, sizeof(orc_solver));
    s->p = *p;
    const int N = p->N;
    s->is_first_solve = 1;
    s->status = ORC_RUNNING;
    s->last_solve_u = (double*)calloc((size_t)N * 2, sizeof(double));
    s->l_x = (double*)calloc((size_t)(N + 1) * 4, sizeof(double));
    s->l_u = (double*)calloc((size_t)N * 2, sizeof(double));
    s->l_xx = (double*)calloc((size_t)(N + 1) * 16, sizeof(double));
    s->l_uu = (double*)calloc((size_t)N * 4, sizeof(double));
    s->ws_ref = (double*)calloc((size_t)(N + 1) * 3, sizeof(double));
    s->ws_lub = (double*)calloc((size_t)N * 2, sizeof(double));
    s->ws_luub = (double*)calloc((size_t)N * 4, sizeof(double));
    s->ws_lxb = (double*)calloc((size_t)(N + 1) * 4, sizeof(double));
    s->ws_lxxb = (double*)calloc((size_t)(N + 1) * 16, sizeof(double));
    s->ws_dfdx = (double*)calloc((size_t)N * 16, sizeof(double));
    s->ws_dfdu = (double*)calloc((size_t)N * 8, sizeof(double));
    s->ws_u = (double*)calloc((size_t)N * 2, sizeof(double));
    s->ws_x = (double*)calloc((size_t)(N + 1) * 4, sizeof(double));
    s->ws_nu = (double*)calloc((size_t)N * 2, sizeof(double));
    s->ws_nx = (double*)calloc((size_t)(N + 1) * 4, sizeof(double));
    s->ws_d = (double*)calloc((size_t)N * 2, sizeof(double));
    s->ws_K = (double*)calloc((size_t)N * 8, sizeof(double));
    s->alm_mu = NULL;
    s->alm_mu_next = NULL;
    s->alm_cols = 0;
    s->alm_rho = p->alm_rho_init;
    return s;
}

void orc_destroy(orc_solver* s) {
    if (!s) return;
    free(s->last_solve_u);
    free(s->l_x);
    free(s->l_u);
    free(s->l_xx);
    free(s->l_uu);
    free(s->alm_mu);
    free(s->alm_mu_next);
    free(s->ws_ref); free(s->ws_lub); free(s->ws_luub); free(s->ws_lxb); free(s->ws_lxxb);
    free(s->ws_dfdx); free(s->ws_dfdu);
    free(s->ws_u); free(s->ws_x); free(s->ws_nu); free(s->ws_nx); free(s->ws_d); free(s->ws_K);
    free(s);
}

void orc_reset(orc_solver* s) {
    s->is_first_solve = 1;
    s->status = ORC_RUNNING;
}

static void ensure_alm(orc_solver* s, int cols) {
    if (s->alm_cols != cols || !s->alm_mu) {
        free(s->alm_mu);
        free(s->alm_mu_next);
        s->alm_cols = cols;
        s->alm_mu = (double*)calloc((size_t)s->p.N * cols, sizeof(double));
        s->alm_mu_next = (double*)calloc((size_t)s->p.N * cols, sizeof(double));
    }
}

/* ---------- cs:85-153 solve ---------- */
int orc_solve(orc_solver* s, const double x0[4], const orc_scene* sc, double* u_out, double* x_out,
              orc_result* res, orc_trace_rec* trace, int32_t trace_cap) {
    const orc_params* p = &s->p;
    const int N = p->N;
    if (sc->M > 0 && sc->tick + N + 1 > sc->T) return -1; /* RoutingLine::operator[] would throw (ut:52-58) */
    if (sc->L < 1) return -2;

    if (p->solve_type == 1 && (!p->use_last_solution || (p->use_last_solution && s->is_first_solve))) {
        ensure_alm(s, 8 + 2 * sc->M);
        s->alm_rho = p->alm_rho_init;
        memset(s->alm_mu, 0, sizeof(double) * (size_t)N * s->alm_cols);
        memset(s->alm_mu_next, 0, sizeof(double) * (size_t)N * s->alm_cols);
    } else if (p->solve_type == 1) {
        ensure_alm(s, 8 + 2 * sc->M);
    }
    s->status = ORC_RUNNING;
    s->cost_evals = 0;

    double *u = s->ws_u, *x = s->ws_x, *new_u = s->ws_nu, *new_x = s->ws_nx, *d = s->ws_d, *K = s->ws_K;
    memset(u, 0, sizeof(double) * (size_t)N * 2);
    memset(x, 0, sizeof(double) * (size_t)(N + 1) * 4);

    if (!s->is_first_solve && p->use_last_solution) {
        /* cs:163-180 get_init_traj_increment */
        for (int i = 0; i < N - 1; ++i) {
            u[i * 2 + 0] = s->last_solve_u[(i + 1) * 2 + 0];
            u[i * 2 + 1] = s->last_solve_u[(i + 1) * 2 + 1];
        }
        u[(N - 1) * 2 + 0] = s->last_solve_u[(N - 1) * 2 + 0];
        u[(N - 1) * 2 + 1] = s->last_solve_u[(N - 1) * 2 + 1];
        memcpy(x, x0, 4 * sizeof(double));
        for (int i = 0; i < N; ++i) {
            orc_kinematic_propagate(x + i * 4, u + i * 2, p->dt, p->wheelbase, p->reference_point,
                                    x + (i + 1) * 4);
        }
    } else {
        /* cs:155-161 get_init_traj */
        orc_const_velo_prediction(p, x0, x);
        s->is_first_solve = 0;
    }

    s->mg_cur.ls_margin = s->mg_cur.pd_margin = s->mg_cur.ref_margin = DBL_MAX; /* the initial cost's scan joins record 0 */
    double J = orc_total_cost(s, u, x, sc);
    double lamb = p->init_lamb;
    int is_exceed_max_itr = 1;
    int iter_effective_flag = 0;
    int iters = 0, ls_trials = 0, tl = 0;
    int end_reason = ORC_END_MAX_ITER;
    for (int itr = 0; itr < p->max_iter; ++itr) {
        iter_out io = iter_step(s, u, x, lamb, sc, &iter_effective_flag, new_u, new_x, d, K);
        iters++;
        ls_trials += io.trials;
        if (iter_effective_flag) {
            memcpy(x, new_x, sizeof(double) * 4 * (N + 1));
            memcpy(u, new_u, sizeof(double) * 2 * N);
        }
        if (s->status == ORC_BACKWARD_PASS_FAIL || s->status == ORC_FORWARD_PASS_FAIL) {
            double la = lamb * p->lamb_amplify;
            lamb = (p->lamb_amplify < la) ? la : p->lamb_amplify; /* std::max(lamb_amplify, lamb*lamb_amplify) */
        } else if (s->status == ORC_RUNNING) {
            lamb *= p->lamb_decay;
        }
        if (s->mg_buf) {
            if (tl < s->mg_cap) s->mg_buf[tl] = s->mg_cur;
            s->mg_cur.ls_margin = s->mg_cur.pd_margin = s->mg_cur.ref_margin = DBL_MAX;
        }
        if (trace && tl < trace_cap) {
            trace[tl].status = s->status;
            trace[tl].trials = io.trials;
            trace[tl].accepted = iter_effective_flag;
            trace[tl].alpha_idx = io.alpha_idx;
            trace[tl].lamb = lamb;
            trace[tl].new_J = io.new_J;
        }
        tl++;
        if (lamb > p->max_lamb) {
            is_exceed_max_itr = 0;
            end_reason = ORC_END_MAX_LAMB;
            break;
        } else if (s->status == ORC_CONVERGED) {
            is_exceed_max_itr = 0;
            end_reason = ORC_END_CONVERGED;
            break;
        }
    }
    (void)is_exceed_max_itr;
    memcpy(s->last_solve_u, u, sizeof(double) * 2 * N);
    memcpy(u_out, u, sizeof(double) * 2 * N);
    memcpy(x_out, x, sizeof(double) * 4 * (N + 1));
    if (res) {
        res->J_init = J;
        res->iters = iters;
        res->end_reason = end_reason;
        res->final_status = s->status;
        res->ls_trials = ls_trials;
        res->cost_evals = s->cost_evals;
        res->trace_len = (tl < trace_cap || !trace) ? tl : trace_cap;
        res->J_final = orc_total_cost(s, u, x, sc);
    }
    return 0;
}

int orc_solve_batch(const orc_params* params, int32_t n_params, const orc_scene* scenes,
                    int32_t n_scenes, int32_t B, const double* x0, const int32_t* scene_id,
                    const int32_t* param_id, const int32_t* tick, int32_t n_threads, double* u_out,
                    double* x_out, orc_result* res) {
    int rc_all = 0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        /* one solver instance per thread and parameter set, reset before every (cold-start) solve */
        orc_solver** pool = (orc_solver**)calloc((size_t)n_params, sizeof(orc_solver*));
#pragma omp for schedule(dynamic, 4)
        for (int b = 0; b < B; ++b) {
            int pid = param_id ? param_id[b] : 0;
            int sid = scene_id ? scene_id[b] : 0;
            if (pid < 0 || pid >= n_params || sid < 0 || sid >= n_scenes) {
#pragma omp atomic write
                rc_all = -3;
                continue;
            }
            const orc_params* p = params + pid;
            orc_scene sc = scenes[sid];
            if (tick) sc.tick = tick[b];
            if (!pool[pid]) pool[pid] = orc_create(p);
            orc_solver* s = pool[pid];
            orc_reset(s);
            int rc = orc_solve(s, x0 + (size_t)b * 4, &sc, u_out + (size_t)b * p->N * 2,
                               x_out + (size_t)b * (p->N + 1) * 4, res ? res + b : NULL, NULL, 0);
            if (rc != 0) {
#pragma omp atomic write
                rc_all = rc;
            }
        }
        for (int i = 0; i < n_params; ++i) orc_destroy(pool[i]);
        free(pool);
    }
    return rc_all;
}

/* decision-margin recorder: subsequent orc_solve calls write one record per iteration (NULL = off) */
void orc_set_margin_buffer(orc_solver* s, orc_margin_rec* buf, int32_t cap) {
    s->mg_buf = buf;
    s->mg_cap = buf ? cap : 0;
}

/* test hooks: inject / read the augmented-Lagrangian state of an instance */
void orc_set_alm_state(orc_solver* s, const double* mu, double rho, int32_t cols) {
    ensure_alm(s, cols);
    memcpy(s->alm_mu, mu, sizeof(double) * (size_t)s->p.N * cols);
    s->alm_rho = rho;
}

void orc_get_alm_next(orc_solver* s, double* mu_next) {
    memcpy(mu_next, s->alm_mu_next, sizeof(double) * (size_t)s->p.N * s->alm_cols);
}
