// cilqr_device.hpp — gfx950 device code of the batched CILQR solve path.
//
// Mapping (one wavefront = one trajectory, block = 64 threads, FP64 throughout, no MFMA):
//   * phases that are parallel over the horizon (stage cost, cost gradients/Hessians, model
//     Jacobians) run with lane = time step k;
//   * the line search runs with lane = trial step size: lane a rolls the closed-loop dynamics out
//     with alpha = 2^-a, so all 20 trial trajectories of cs:354 cost one pass of the serial
//     instruction stream; the trial costs are then evaluated (lane = k again) in the reference's
//     order until the first trial that the reference would have accepted;
//   * the backward Riccati-like sweep is serial over the horizon and is computed wave-uniformly
//     from LDS-resident stage data, exploiting the sparsity of df/dx = I + 5 entries and df/du
//     (3 entries + dt);
//   * x, u, K, d, l_*, A, B live in LDS; the 20 trial trajectories live in an L2-resident scratch
//     slab; lane table / obstacle routes are read through L1/L2 (they are shared by the batch).
//
// Arithmetic contract: every value is computed with the same IEEE operations in the same order
// as oracle/cilqr_oracle.c (which restates the reference's Eigen expressions), with structural
// zeros dropped (x + 0*y == x) — so results are bit-identical to the oracle's detmath build.
// Compile with -ffp-contract=off.
//
// Reference citations: "cs:" = /root/reference/src/cilqr_solver.cpp, "ut:" = /root/reference/src/utils.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cilqr_amd.h"
#include "detmath.h"

#define CILQR_WAVE 64
#define CILQR_EPS 1e-5 /* include/utils.hpp:28 */

namespace cilqr {

struct DevScene {
    const double* lane_xy;  // [L][2]
    const double* lane_yaw; // [L]
    const double* obs;      // [M][T][3]
    int L, M, T, pad;
    double border_hi, border_lo, ref_velo;
};

// wave-uniform constants of one trajectory
struct Cst {
    int N, rp, M, L, T, tick, max_iter, pad;
    double dt, wb, half_wb;
    double w_pos, w_vel, w_yaw, w_acc, w_stl;
    double sq1, sq2, oq1, oq2;
    double acc_max, acc_min, stl_lim, velo_max, velo_min;
    double pos_up_b, pos_lo_b;
    double ell_a2, ell_b2;
    double ref_velo;
    double init_lamb, lamb_decay, lamb_amplify, max_lamb, conv_thr, accept_thr;
    const double* lane_xy;
    const double* lane_yaw;
    const double* obs;
};

__device__ inline void make_cst(Cst& c, const cilqr_params& p, const DevScene& s, int tick) {
    c.N = p.N; c.rp = p.reference_point; c.M = s.M; c.L = s.L; c.T = s.T; c.tick = tick;
    c.max_iter = p.max_iter;
    c.dt = p.dt; c.wb = p.wheelbase; c.half_wb = 0.5 * p.wheelbase;
    c.w_pos = p.w_pos; c.w_vel = p.w_vel; c.w_yaw = p.w_yaw; c.w_acc = p.w_acc; c.w_stl = p.w_stl;
    c.sq1 = p.state_exp_q1; c.sq2 = p.state_exp_q2; c.oq1 = p.obstacle_exp_q1; c.oq2 = p.obstacle_exp_q2;
    c.acc_max = p.acc_max; c.acc_min = p.acc_min; c.stl_lim = p.stl_lim;
    c.velo_max = p.velo_max; c.velo_min = p.velo_min;
    c.pos_up_b = s.border_hi - p.width / 2; // cs:239
    c.pos_lo_b = s.border_lo + p.width / 2; // cs:241
    // ut:387-393 with obs_attr = {width, length, d_safe} (cs:78) and ego_pnt_radius = 0.5*width (cs:330)
    double a = 0.5 * p.length + p.d_safe * 6 + 0.5 * p.width;
    double b = 0.5 * p.width + p.d_safe + 0.5 * p.width;
    c.ell_a2 = a * a;
    c.ell_b2 = b * b;
    c.ref_velo = s.ref_velo;
    c.init_lamb = p.init_lamb; c.lamb_decay = p.lamb_decay; c.lamb_amplify = p.lamb_amplify;
    c.max_lamb = p.max_lamb; c.conv_thr = p.convergence_threshold; c.accept_thr = p.accept_step_threshold;
    c.lane_xy = s.lane_xy; c.lane_yaw = s.lane_yaw; c.obs = s.obs;
}

// LDS carve-out for one trajectory (offsets in doubles).  l_xx keeps the 7 entries that can be
// non-zero: (0,0) (0,1) (0,3) (1,1) (1,3) (3,3) (2,2); l_uu is diagonal in barrier mode.
struct Lds {
    double* x;   // [(N+1)][4]
    double* u;   // [N][2]
    double* K;   // [N][8]
    double* d;   // [N][2]
    double* lx;  // [(N+1)][4]
    double* lu;  // [N][2]
    double* lxx; // [(N+1)][7]
    double* luu; // [N][2]
    double* A5;  // [N][5]  a02 a03 a12 a13 a32
    double* B3;  // [N][3]  b01 b11 b31
    double* cs;  // [3][(N+1)] stage-cost scratch: state, ctrl, barrier
    int* ridx;   // [(N+1)] lane-sample index of every row of the current trajectory
};

__host__ __device__ inline int lds_doubles(int N) {
    return 4 * (N + 1) + 2 * N + 8 * N + 2 * N + 4 * (N + 1) + 2 * N + 7 * (N + 1) + 2 * N + 5 * N +
           3 * N + 3 * (N + 1);
}
__host__ __device__ inline size_t lds_bytes(int N) {
    return sizeof(double) * (size_t)lds_doubles(N) + sizeof(int) * (size_t)(N + 2);
}

__device__ inline void carve(Lds& l, double* base, int N) {
    double* p = base;
    l.x = p; p += 4 * (N + 1);
    l.u = p; p += 2 * N;
    l.K = p; p += 8 * N;
    l.d = p; p += 2 * N;
    l.lx = p; p += 4 * (N + 1);
    l.lu = p; p += 2 * N;
    l.lxx = p; p += 7 * (N + 1);
    l.luu = p; p += 2 * N;
    l.A5 = p; p += 5 * N;
    l.B3 = p; p += 3 * N;
    l.cs = p; p += 3 * (N + 1);
    l.ridx = reinterpret_cast<int*>(p);
}

// scratch slab of the trial trajectories: [alpha][7][(N+1)] doubles
// rows 0-3 = x' components, 4-5 = u' components, 6 = lane-sample index (stored as a double)
__host__ __device__ inline size_t scratch_doubles(int N) {
    return (size_t)CILQR_MAX_ALPHA_TRIALS * 7 * (size_t)(N + 1);
}

// ---------------------------------------------------------------------------------------------
// ut:262-283 kinematic_propagate
__device__ inline void propagate(const Cst& c, const double x[4], const double u[2], double xn[4]) {
    if (c.rp == 0) {
        double sn, cs;
        dm_sincos(x[3], &sn, &cs);
        double tn = dm_tan(u[1]);
        xn[0] = x[0] + x[2] * cs * c.dt;
        xn[1] = x[1] + x[2] * sn * c.dt;
        xn[2] = x[2] + u[0] * c.dt;
        xn[3] = x[3] + x[2] * tn * c.dt / c.wb;
    } else {
        double beta = dm_atan(dm_tan(u[1]) / 2);
        double sn, cs;
        dm_sincos(beta + x[3], &sn, &cs);
        double sb = dm_sin(beta);
        xn[0] = x[0] + x[2] * cs * c.dt;
        xn[1] = x[1] + x[2] * sn * c.dt;
        xn[2] = x[2] + u[0] * c.dt;
        xn[3] = x[3] + 2 * x[2] * sb * c.dt / c.wb;
    }
}

// cs:295-311 for row 0 (start_index = 0): all 64 lanes evaluate consecutive candidates.
// Returns the first j >= 0 at which the distance stops strictly decreasing (or L-1).
__device__ inline int ref_scan_row0(const Cst& c, double px, double py, int lane) {
    int s = 0;
    for (;;) {
        int j = s + lane;
        double D = dm_inf();
        if (j < c.L) D = dm_hypot(px - c.lane_xy[2 * j], py - c.lane_xy[2 * j + 1]);
        double Dn = __shfl_down(D, 1, CILQR_WAVE);
        bool stop = !(Dn < D) && (lane < CILQR_WAVE - 1);
        unsigned long long m = __ballot(stop);
        if (m != 0ULL) {
            int first = __ffsll((long long)m) - 1;
            return s + first;
        }
        s += CILQR_WAVE - 1;
    }
}

// "is hypot(cur) < hypot(best)" decided on the squared values whenever that is safe
__device__ inline bool dist_less(double cur2, double best2) {
    if (cur2 >= best2) return false;                          // sqrt is monotone
    if (cur2 < best2 * 0.99999999999999644729) return true;   // 1 - 2^-48: > 16 ulp apart after sqrt
    return dm_sqrt(cur2) < dm_sqrt(best2);                    // near tie (also the NaN path)
}

// cs:295-311 for one row, lane-private: first local minimum of the distance at or after s
__device__ inline int ref_scan_from(const Cst& c, double px, double py, int s) {
    int j = s;
    double bx = px - c.lane_xy[2 * j], by = py - c.lane_xy[2 * j + 1];
    double best2 = bx * bx + by * by;
    while (j + 1 < c.L) {
        double cx = px - c.lane_xy[2 * (j + 1)], cy = py - c.lane_xy[2 * (j + 1) + 1];
        double cur2 = cx * cx + cy * cy;
        if (!dist_less(cur2, best2)) break;
        best2 = cur2;
        ++j;
    }
    return j;
}

// ---------------------------------------------------------------------------------------------
// one obstacle against one ego state: margins (cs:326-335, ut:344-361,395-407) and, if GRAD, their
// gradients w.r.t. the state (cs:715-739, ut:363-385,409-439).
struct ObsOut {
    double mf, mr;        // front / rear safety margin
    double gf[3], gr[3];  // gradient components (x, y, yaw); the v component is structurally 0
};

template <bool GRAD>
__device__ inline void obstacle_terms(const Cst& c, const double xk[4], double sn_yaw, double cs_yaw,
                                      const double* ob, ObsOut& o) {
    double wv0 = c.wb * cs_yaw, wv1 = c.wb * sn_yaw;
    double fx, fy, rx, ry;
    if (c.rp == 0) {
        fx = xk[0] + wv0; fy = xk[1] + wv1; rx = xk[0]; ry = xk[1];
    } else {
        fx = xk[0] + 0.5 * wv0; fy = xk[1] + 0.5 * wv1;
        rx = xk[0] - 0.5 * wv0; ry = xk[1] - 0.5 * wv1;
    }
    double so, co;
    dm_sincos(ob[2], &so, &co);
    double dfx = fx - ob[0], dfy = fy - ob[1];
    double drx = rx - ob[0], dry = ry - ob[1];
    double fX = co * dfx + so * dfy, fY = (-so) * dfx + co * dfy;
    double rX = co * drx + so * dry, rY = (-so) * drx + co * dry;
    o.mf = 1 - ((fX * fX) / c.ell_a2 + (fY * fY) / c.ell_b2);
    o.mr = 1 - ((rX * rX) / c.ell_a2 + (rY * rY) / c.ell_b2);
    if (GRAD) {
        double f0 = -2 * fX / c.ell_a2, f1 = -2 * fY / c.ell_b2;
        double r0 = -2 * rX / c.ell_a2, r1 = -2 * rY / c.ell_b2;
        double gfx = co * f0 + (-so) * f1, gfy = so * f0 + co * f1;
        double grx = co * r0 + (-so) * r1, gry = so * r0 + co * r1;
        double f30, f31, r30, r31;
        if (c.rp == 0) {
            f30 = c.wb * (-sn_yaw); f31 = c.wb * cs_yaw; r30 = 0; r31 = 0;
        } else {
            f30 = c.half_wb * (-sn_yaw); f31 = c.half_wb * cs_yaw;
            r30 = -c.half_wb * (-sn_yaw); r31 = -c.half_wb * cs_yaw;
        }
        o.gf[0] = gfx; o.gf[1] = gfy; o.gf[2] = f30 * gfx + f31 * gfy;
        o.gr[0] = grx; o.gr[1] = gry; o.gr[2] = r30 * grx + r31 * gry;
    }
}

__device__ inline const double* obs_at(const Cst& c, int j, int k) {
    return c.obs + ((size_t)j * c.T + (size_t)(c.tick + k)) * 3;
}

// ---------------------------------------------------------------------------------------------
// Stage cost of row k (cs:199-287).  xk = x[k], uk = u[k] (k < N), ukm1 = u[k-1] (k >= 1).
// sd = k-th diagonal entry of (x-ref) W (x-ref)^T, ce = k-th of u R u^T, jb = J_barrier_k.
__device__ inline void stage_cost(const Cst& c, int k, const double xk[4], const double uk[2],
                                  const double ukm1[2], int ridx, double& sd, double& ce, double& jb) {
    double rx = c.lane_xy[2 * ridx], ry = c.lane_xy[2 * ridx + 1], ryaw = c.lane_yaw[ridx];
    double e0 = xk[0] - rx, e1 = xk[1] - ry, e2 = xk[2] - c.ref_velo, e3 = xk[3] - ryaw;
    sd = (((e0 * c.w_pos) * e0 + (e1 * c.w_pos) * e1) + (e2 * c.w_vel) * e2) + (e3 * c.w_yaw) * e3;
    ce = 0.0;
    if (k < c.N) ce = (uk[0] * c.w_acc) * uk[0] + (uk[1] * c.w_stl) * uk[1];
    jb = 0.0;
    if (k >= 1) {
        double acc_up = ukm1[0] - c.acc_max, acc_lo = c.acc_min - ukm1[0];
        double stl_up = ukm1[1] - c.stl_lim, stl_lo = -c.stl_lim - ukm1[1];
        double vel_up = xk[2] - c.velo_max, vel_lo = c.velo_min - xk[2];
        double sr, cr;
        dm_sincos(ryaw, &sr, &cr);
        double d_sign = e1 * cr - e0 * sr;
        double hyp = dm_hypot(e0, e1);
        double cur_d = (d_sign < 0) ? -hyp : hyp;
        double pos_up = cur_d - c.pos_up_b, pos_lo = c.pos_lo_b - cur_d;
        double j = c.sq1 * dm_exp(c.sq2 * acc_up) + c.sq1 * dm_exp(c.sq2 * acc_lo);
        j = j + c.sq1 * dm_exp(c.sq2 * stl_up);
        j = j + c.sq1 * dm_exp(c.sq2 * stl_lo);
        j = j + c.sq1 * dm_exp(c.sq2 * vel_up);
        j = j + c.sq1 * dm_exp(c.sq2 * vel_lo);
        j = j + c.sq1 * dm_exp(c.sq2 * pos_up);
        j = j + c.sq1 * dm_exp(c.sq2 * pos_lo);
        double sy, cy;
        dm_sincos(xk[3], &sy, &cy);
        for (int o = 0; o < c.M; ++o) {
            ObsOut t;
            obstacle_terms<false>(c, xk, sy, cy, obs_at(c, o, k), t);
            j = j + c.oq1 * dm_exp(c.oq2 * t.mf);
            j = j + c.oq1 * dm_exp(c.oq2 * t.mr);
        }
        jb = j;
    }
}

// J = (sum_k sd + sum_k ce) + sum_k jb, each sum sequential in k as Eigen's trace()/the loop at
// cs:217 accumulate.  All lanes compute the same value from LDS broadcasts.
__device__ inline double sum_stage_costs(const Lds& l, int N) {
    const double* sdv = l.cs;
    const double* cev = l.cs + (N + 1);
    const double* jbv = l.cs + 2 * (N + 1);
    double sd = sdv[0], ce = cev[0], jb = 0.0;
    for (int k = 1; k <= N; ++k) {
        sd = sd + sdv[k];
        jb = jb + jbv[k];
        if (k < N) ce = ce + cev[k];
    }
    return (sd + ce) + jb;
}

// get_total_cost of the trajectory held in LDS (x, u, ridx)
__device__ inline double total_cost_lds(const Cst& c, const Lds& l, int lane) {
    const int N = c.N;
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        double xk[4] = {l.x[4 * k], l.x[4 * k + 1], l.x[4 * k + 2], l.x[4 * k + 3]};
        double uk[2] = {0, 0}, um[2] = {0, 0};
        if (k < N) { uk[0] = l.u[2 * k]; uk[1] = l.u[2 * k + 1]; }
        if (k >= 1) { um[0] = l.u[2 * k - 2]; um[1] = l.u[2 * k - 1]; }
        double sd, ce, jb;
        stage_cost(c, k, xk, uk, um, l.ridx[k], sd, ce, jb);
        l.cs[k] = sd;
        l.cs[(N + 1) + k] = ce;
        l.cs[2 * (N + 1) + k] = jb;
    }
    __syncthreads();
    double J = sum_stage_costs(l, N);
    __syncthreads();
    return J;
}

// get_total_cost of trial trajectory `a` held in the scratch slab
__device__ inline double total_cost_trial(const Cst& c, const Lds& l, const double* scr, int a, int lane) {
    const int N = c.N;
    const int R = N + 1;
    const double* t = scr + (size_t)a * 7 * R;
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        double xk[4] = {t[k], t[R + k], t[2 * R + k], t[3 * R + k]};
        double uk[2] = {0, 0}, um[2] = {0, 0};
        if (k < N) { uk[0] = t[4 * R + k]; uk[1] = t[5 * R + k]; }
        if (k >= 1) { um[0] = t[4 * R + k - 1]; um[1] = t[5 * R + k - 1]; }
        int ridx = (int)t[6 * R + k];
        double sd, ce, jb;
        stage_cost(c, k, xk, uk, um, ridx, sd, ce, jb);
        l.cs[k] = sd;
        l.cs[R + k] = ce;
        l.cs[2 * R + k] = jb;
    }
    __syncthreads();
    double J = sum_stage_costs(l, N);
    __syncthreads();
    return J;
}

// ---------------------------------------------------------------------------------------------
// Initial trajectory (cs:155-197): cold start u = 0, or warm start from last_u shifted by one step;
// fills LDS x, u, ridx.  Wave-uniform serial rollout.
__device__ inline void init_trajectory(const Cst& c, const Lds& l, const double x0[4],
                                       const double* last_u, int lane, int& idx0) {
    const int N = c.N;
    for (int k = lane; k < N; k += CILQR_WAVE) {
        double a = 0.0, b = 0.0;
        if (last_u) {
            int src = (k < N - 1) ? (k + 1) : (N - 1);
            a = last_u[2 * src];
            b = last_u[2 * src + 1];
        }
        l.u[2 * k] = a;
        l.u[2 * k + 1] = b;
    }
    __syncthreads();
    idx0 = ref_scan_row0(c, x0[0], x0[1], lane);
    double xc[4] = {x0[0], x0[1], x0[2], x0[3]};
    int s = idx0;
    if (lane == 0) {
        l.x[0] = xc[0]; l.x[1] = xc[1]; l.x[2] = xc[2]; l.x[3] = xc[3];
        l.ridx[0] = s;
    }
    for (int i = 0; i < N; ++i) {
        double ui[2] = {l.u[2 * i], l.u[2 * i + 1]};
        double xn[4];
        propagate(c, xc, ui, xn);
        s = ref_scan_from(c, xn[0], xn[1], s);
        if (lane == 0) {
            l.x[4 * (i + 1)] = xn[0]; l.x[4 * (i + 1) + 1] = xn[1];
            l.x[4 * (i + 1) + 2] = xn[2]; l.x[4 * (i + 1) + 3] = xn[3];
            l.ridx[i + 1] = s;
        }
        xc[0] = xn[0]; xc[1] = xn[1]; xc[2] = xn[2]; xc[3] = xn[3];
    }
    __syncthreads();
}

// ridx for a trajectory already staged in LDS x (used by the piecewise kernels)
__device__ inline void ref_indices_lds(const Cst& c, const Lds& l, int lane, int& idx0) {
    const int N = c.N;
    idx0 = ref_scan_row0(c, l.x[0], l.x[1], lane);
    int s = idx0;
    if (lane == 0) l.ridx[0] = s;
    for (int i = 1; i <= N; ++i) {
        s = ref_scan_from(c, l.x[4 * i], l.x[4 * i + 1], s);
        if (lane == 0) l.ridx[i] = s;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// forward_pass (cs:442-461) for all trial step sizes: lane a < n_alpha uses alpha = 2^-a and also
// tracks the lane-sample index of every new row (cs:289-314) so that the costs can be evaluated
// afterwards without another serial scan.
__device__ inline void rollout_trials(const Cst& c, const Lds& l, double* scr, int lane, int idx0,
                                      int n_alpha) {
    const int N = c.N;
    const int R = N + 1;
    if (lane < n_alpha) {
        const double alpha = dm_pow2i(-lane);
        double* t = scr + (size_t)lane * 7 * R;
        double xc[4] = {l.x[0], l.x[1], l.x[2], l.x[3]};
        int s = idx0;
        t[0] = xc[0]; t[R] = xc[1]; t[2 * R] = xc[2]; t[3 * R] = xc[3];
        t[6 * R] = (double)s;
        for (int i = 0; i < N; ++i) {
            const double* Ki = l.K + 8 * i;
            const double* xi = l.x + 4 * i;
            double dx0 = xc[0] - xi[0], dx1 = xc[1] - xi[1], dx2 = xc[2] - xi[2], dx3 = xc[3] - xi[3];
            double k0 = ((Ki[0] * dx0 + Ki[1] * dx1) + Ki[2] * dx2) + Ki[3] * dx3;
            double k1 = ((Ki[4] * dx0 + Ki[5] * dx1) + Ki[6] * dx2) + Ki[7] * dx3;
            double un[2];
            un[0] = (l.u[2 * i] + k0) + alpha * l.d[2 * i];
            un[1] = (l.u[2 * i + 1] + k1) + alpha * l.d[2 * i + 1];
            double xn[4];
            propagate(c, xc, un, xn);
            s = ref_scan_from(c, xn[0], xn[1], s);
            t[4 * R + i] = un[0];
            t[5 * R + i] = un[1];
            t[i + 1] = xn[0];
            t[R + i + 1] = xn[1];
            t[2 * R + i + 1] = xn[2];
            t[3 * R + i + 1] = xn[3];
            t[6 * R + i + 1] = (double)s;
            xc[0] = xn[0]; xc[1] = xn[1]; xc[2] = xn[2]; xc[3] = xn[3];
        }
    }
    __syncthreads();
}

// copy trial `a` from the scratch slab into the current trajectory in LDS
__device__ inline void accept_trial(const Cst& c, const Lds& l, const double* scr, int a, int lane) {
    const int N = c.N;
    const int R = N + 1;
    const double* t = scr + (size_t)a * 7 * R;
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        l.x[4 * k] = t[k];
        l.x[4 * k + 1] = t[R + k];
        l.x[4 * k + 2] = t[2 * R + k];
        l.x[4 * k + 3] = t[3 * R + k];
        l.ridx[k] = (int)t[6 * R + k];
        if (k < N) {
            l.u[2 * k] = t[4 * R + k];
            l.u[2 * k + 1] = t[5 * R + k];
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// get_total_cost_derivatives_and_Hessians (cs:463-690, barrier mode) and
// get_kinematic_model_derivatives (ut:285-342), lane = k.  Row k holds l_x[k], l_xx[k] (from x[k],
// u[k-1]-independent), lane k also produces l_u[k-1], l_uu[k-1] (they depend on u[k-1]) and, for
// k < N, the model Jacobian entries of step k.
__device__ inline void cost_and_model_derivatives(const Cst& c, const Lds& l, int lane) {
    const int N = c.N;
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        double xk[4] = {l.x[4 * k], l.x[4 * k + 1], l.x[4 * k + 2], l.x[4 * k + 3]};
        int ridx = l.ridx[k];
        double rx = c.lane_xy[2 * ridx], ry = c.lane_xy[2 * ridx + 1], ryaw = c.lane_yaw[ridx];
        double e0 = xk[0] - rx, e1 = xk[1] - ry, e2 = xk[2] - c.ref_velo, e3 = xk[3] - ryaw;
        // prime parts (cs:493-494)
        double lx0 = (2 * e0) * c.w_pos, lx1 = (2 * e1) * c.w_pos, lx2 = (2 * e2) * c.w_vel, lx3 = (2 * e3) * c.w_yaw;
        double h00 = 0, h01 = 0, h03 = 0, h11 = 0, h13 = 0, h33 = 0, h22 = 0; // barrier Hessian
        double b0 = 0, b1 = 0, b2 = 0, b3 = 0;                                 // barrier gradient
        double sy, cy;
        dm_sincos(xk[3], &sy, &cy);
        if (k >= 1) {
            double um0 = l.u[2 * k - 2], um1 = l.u[2 * k - 1];
            // control bounds (cs:510-513, 537-558)
            double b_au = c.sq1 * dm_exp(c.sq2 * (um0 - c.acc_max));
            double b_al = c.sq1 * dm_exp(c.sq2 * (c.acc_min - um0));
            double b_su = c.sq1 * dm_exp(c.sq2 * (um1 - c.stl_lim));
            double b_sl = c.sq1 * dm_exp(c.sq2 * (-c.stl_lim - um1));
            double q22 = c.sq2 * c.sq2;
            double lub0 = (c.sq2 * b_au) - (c.sq2 * b_al);
            double lub1 = (c.sq2 * b_su) - (c.sq2 * b_sl);
            double luub0 = (q22 * b_au) + (q22 * b_al);
            double luub1 = (q22 * b_su) + (q22 * b_sl);
            // l_u = 2 (u R) + barrier, l_uu = 2 R + barrier (cs:491-492, 686-687)
            l.lu[2 * (k - 1)] = 2 * (um0 * c.w_acc) + lub0;
            l.lu[2 * (k - 1) + 1] = 2 * (um1 * c.w_stl) + lub1;
            l.luu[2 * (k - 1)] = 2 * c.w_acc + luub0;
            l.luu[2 * (k - 1) + 1] = 2 * c.w_stl + luub1;
            // velocity bounds and road borders (cs:507-533, 560-580)
            double b_vu = c.sq1 * dm_exp(c.sq2 * (xk[2] - c.velo_max));
            double b_vl = c.sq1 * dm_exp(c.sq2 * (c.velo_min - xk[2]));
            double sr, cr;
            dm_sincos(ryaw, &sr, &cr);
            double d_sign = e1 * cr - e0 * sr;
            double hyp = dm_hypot(e0, e1);
            double cur_d = (d_sign < 0) ? -hyp : hyp;
            double b_pu = c.sq1 * dm_exp(c.sq2 * (cur_d - c.pos_up_b));
            double b_pl = c.sq1 * dm_exp(c.sq2 * (c.pos_lo_b - cur_d));
            double px = e0 / hyp, py = e1 / hyp;
            if (d_sign < 0) { px = -px; py = -py; }
            double nx = -px, ny = -py; // pos_lo_constr_over_x = -1 * pos_up_constr_over_x
            double d_pu = c.sq2 * b_pu, d_pl = c.sq2 * b_pl;
            double s_pu = q22 * b_pu, s_pl = q22 * b_pl;
            b0 = d_pu * px + d_pl * nx;
            b1 = d_pu * py + d_pl * ny;
            b2 = (c.sq2 * b_vu) - (c.sq2 * b_vl);
            b3 = 0.0;
            h00 = s_pu * (px * px) + s_pl * (nx * nx);
            h01 = s_pu * (px * py) + s_pl * (nx * ny);
            h11 = s_pu * (py * py) + s_pl * (ny * ny);
            h22 = (q22 * b_vu) + (q22 * b_vl);
            // obstacles (cs:647-683)
            double oq22 = c.oq2 * c.oq2;
            for (int o = 0; o < c.M; ++o) {
                ObsOut t;
                obstacle_terms<true>(c, xk, sy, cy, obs_at(c, o, k), t);
                double bf = c.oq1 * dm_exp(c.oq2 * t.mf);
                double br = c.oq1 * dm_exp(c.oq2 * t.mr);
                double df = c.oq2 * bf, dr = c.oq2 * br;
                double sf = oq22 * bf, srr = oq22 * br;
                b0 = b0 + (df * t.gf[0] + dr * t.gr[0]);
                b1 = b1 + (df * t.gf[1] + dr * t.gr[1]);
                b3 = b3 + (df * t.gf[2] + dr * t.gr[2]);
                h00 = h00 + (sf * (t.gf[0] * t.gf[0]) + srr * (t.gr[0] * t.gr[0]));
                h01 = h01 + (sf * (t.gf[0] * t.gf[1]) + srr * (t.gr[0] * t.gr[1]));
                h03 = h03 + (sf * (t.gf[0] * t.gf[2]) + srr * (t.gr[0] * t.gr[2]));
                h11 = h11 + (sf * (t.gf[1] * t.gf[1]) + srr * (t.gr[1] * t.gr[1]));
                h13 = h13 + (sf * (t.gf[1] * t.gf[2]) + srr * (t.gr[1] * t.gr[2]));
                h33 = h33 + (sf * (t.gf[2] * t.gf[2]) + srr * (t.gr[2] * t.gr[2]));
            }
        }
        l.lx[4 * k] = lx0 + b0;
        l.lx[4 * k + 1] = lx1 + b1;
        l.lx[4 * k + 2] = lx2 + b2;
        l.lx[4 * k + 3] = lx3 + b3;
        double* hx = l.lxx + 7 * k;
        hx[0] = 2 * c.w_pos + h00;
        hx[1] = 0.0 + h01;
        hx[2] = 0.0 + h03;
        hx[3] = 2 * c.w_pos + h11;
        hx[4] = 0.0 + h13;
        hx[5] = 2 * c.w_yaw + h33;
        hx[6] = 2 * c.w_vel + h22;
        if (k < N) {
            // model Jacobians of step k (ut:285-342)
            double v = xk[2];
            double delta = l.u[2 * k + 1];
            double* A = l.A5 + 5 * k;
            double* B = l.B3 + 3 * k;
            if (c.rp == 0) {
                double td = dm_tan(delta);
                double cd = dm_cos(delta);
                A[0] = cy * c.dt;
                A[1] = v * (-sy) * c.dt;
                A[2] = sy * c.dt;
                A[3] = v * cy * c.dt;
                A[4] = td * c.dt / c.wb;
                B[0] = 0.0;
                B[1] = 0.0;
                B[2] = (v * c.dt / c.wb) / (cd * cd);
            } else {
                double beta = dm_atan(dm_tan(delta / 2)); // ut:291 (not the beta of ut:265)
                double td = dm_tan(delta);
                double g = 0.5 * (1 + td * td) / (1 + 0.25 * (td * td));
                double sby, cby;
                dm_sincos(beta + xk[3], &sby, &cby);
                double sb, cb;
                dm_sincos(beta, &sb, &cb);
                A[0] = cby * c.dt;
                A[1] = v * (-sby) * c.dt;
                A[2] = sby * c.dt;
                A[3] = v * cby * c.dt;
                A[4] = 2 * sb * c.dt / c.wb;
                B[0] = v * (-sby) * c.dt * g;
                B[1] = v * cby * c.dt * g;
                B[2] = (2 * v * c.dt / c.wb) * cb * g;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// backward_pass (cs:383-440) after the expansion above.  Wave-uniform; V_x, V_xx in registers.
// Returns true on success, false for a non-PD Q_uu (BACKWARD_PASS_FAIL); fills l.K, l.d, dV.
__device__ inline bool backward_sweep(const Cst& c, const Lds& l, double lamb, int lane, double dV[2]) {
    const int N = c.N;
    double Vx[4], V[16];
    {
        const double* hx = l.lxx + 7 * N;
        Vx[0] = l.lx[4 * N]; Vx[1] = l.lx[4 * N + 1]; Vx[2] = l.lx[4 * N + 2]; Vx[3] = l.lx[4 * N + 3];
        V[0] = hx[0]; V[1] = hx[1]; V[2] = 0.0; V[3] = hx[2];
        V[4] = hx[1]; V[5] = hx[3]; V[6] = 0.0; V[7] = hx[4];
        V[8] = 0.0; V[9] = 0.0; V[10] = hx[6]; V[11] = 0.0;
        V[12] = hx[2]; V[13] = hx[4]; V[14] = 0.0; V[15] = hx[5];
    }
    dV[0] = 0.0;
    dV[1] = 0.0;
    const double dt = c.dt;
    for (int i = N - 1; i >= 0; --i) {
        const double* Ap = l.A5 + 5 * i;
        const double* Bp = l.B3 + 3 * i;
        const double a02 = Ap[0], a03 = Ap[1], a12 = Ap[2], a13 = Ap[3], a32 = Ap[4];
        const double b01 = Bp[0], b11 = Bp[1], b31 = Bp[2];
        const double* hx = l.lxx + 7 * i;
        // Q_x = l_x + A^T V_x ; Q_u = l_u + B^T V_x
        double Qx[4];
        Qx[0] = l.lx[4 * i] + Vx[0];
        Qx[1] = l.lx[4 * i + 1] + Vx[1];
        Qx[2] = l.lx[4 * i + 2] + (((a02 * Vx[0] + a12 * Vx[1]) + Vx[2]) + a32 * Vx[3]);
        Qx[3] = l.lx[4 * i + 3] + ((a03 * Vx[0] + a13 * Vx[1]) + Vx[3]);
        double Qu[2];
        Qu[0] = l.lu[2 * i] + dt * Vx[2];
        Qu[1] = l.lu[2 * i + 1] + ((b01 * Vx[0] + b11 * Vx[1]) + b31 * Vx[3]);
        // T = A^T V
        double T[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T[j] = V[j];
            T[4 + j] = V[4 + j];
            T[8 + j] = ((a02 * V[j] + a12 * V[4 + j]) + V[8 + j]) + a32 * V[12 + j];
            T[12 + j] = (a03 * V[j] + a13 * V[4 + j]) + V[12 + j];
        }
        // Q_xx = l_xx + T A
        double Qxx[16];
        const double lxxd[16] = {hx[0], hx[1], 0.0, hx[2], hx[1], hx[3], 0.0, hx[4],
                                 0.0, 0.0, hx[6], 0.0, hx[2], hx[4], 0.0, hx[5]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double p0 = T[4 * r], p1 = T[4 * r + 1];
            double p2 = ((T[4 * r] * a02 + T[4 * r + 1] * a12) + T[4 * r + 2]) + T[4 * r + 3] * a32;
            double p3 = (T[4 * r] * a03 + T[4 * r + 1] * a13) + T[4 * r + 3];
            Qxx[4 * r] = lxxd[4 * r] + p0;
            Qxx[4 * r + 1] = lxxd[4 * r + 1] + p1;
            Qxx[4 * r + 2] = lxxd[4 * r + 2] + p2;
            Qxx[4 * r + 3] = lxxd[4 * r + 3] + p3;
        }
        // U = B^T V (2x4)
        double U[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            U[j] = dt * V[8 + j];
            U[4 + j] = (b01 * V[j] + b11 * V[4 + j]) + b31 * V[12 + j];
        }
        // Q_uu = l_uu + U B + lamb I
        double Quu[4];
        Quu[0] = (l.luu[2 * i] + U[2] * dt) + lamb;
        Quu[1] = (0.0 + ((U[0] * b01 + U[1] * b11) + U[3] * b31));
        Quu[2] = (0.0 + U[4 + 2] * dt);
        Quu[3] = (l.luu[2 * i + 1] + ((U[4] * b01 + U[5] * b11) + U[7] * b31)) + lamb;
        // Q_ux = U A (2x4)
        double Qux[8];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            Qux[4 * m] = U[4 * m];
            Qux[4 * m + 1] = U[4 * m + 1];
            Qux[4 * m + 2] = ((U[4 * m] * a02 + U[4 * m + 1] * a12) + U[4 * m + 2]) + U[4 * m + 3] * a32;
            Qux[4 * m + 3] = (U[4 * m] * a03 + U[4 * m + 1] * a13) + U[4 * m + 3];
        }
        // Eigen::LLT (lower) PD test
        bool fail = false;
        if (Quu[0] <= 0.0) {
            fail = true;
        } else {
            double l00 = dm_sqrt(Quu[0]);
            double l10 = Quu[2] / l00;
            double piv1 = Quu[3] - l10 * l10;
            if (piv1 <= 0.0) fail = true;
        }
        if (fail) return false;
        double det = Quu[0] * Quu[3] - Quu[2] * Quu[1];
        double invdet = 1.0 / det;
        double n00 = -(Quu[3] * invdet), n01 = -(-Quu[1] * invdet), n10 = -(-Quu[2] * invdet), n11 = -(Quu[0] * invdet);
        double dd[2];
        dd[0] = n00 * Qu[0] + n01 * Qu[1];
        dd[1] = n10 * Qu[0] + n11 * Qu[1];
        double Kk[8];
#pragma unroll
        for (int cidx = 0; cidx < 4; ++cidx) {
            Kk[cidx] = n00 * Qux[cidx] + n01 * Qux[4 + cidx];
            Kk[4 + cidx] = n10 * Qux[cidx] + n11 * Qux[4 + cidx];
        }
        if (lane == 0) {
            l.d[2 * i] = dd[0];
            l.d[2 * i + 1] = dd[1];
#pragma unroll
            for (int e = 0; e < 8; ++e) l.K[8 * i + e] = Kk[e];
        }
        // value function update (cs:427-432)
        double P[8]; // K^T Q_uu (4x2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            P[2 * r] = Kk[r] * Quu[0] + Kk[4 + r] * Quu[2];
            P[2 * r + 1] = Kk[r] * Quu[1] + Kk[4 + r] * Quu[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double ta = P[2 * r] * dd[0] + P[2 * r + 1] * dd[1];
            double tb = Kk[r] * Qu[0] + Kk[4 + r] * Qu[1];
            double tc = Qux[r] * dd[0] + Qux[4 + r] * dd[1];
            Vx[r] = ((Qx[r] + ta) + tb) + tc;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                double ta = P[2 * r] * Kk[cidx] + P[2 * r + 1] * Kk[4 + cidx];
                double tb = Kk[r] * Qux[cidx] + Kk[4 + r] * Qux[4 + cidx];
                double tc = Qux[r] * Kk[cidx] + Qux[4 + r] * Kk[4 + cidx];
                V[4 * r + cidx] = ((Qxx[4 * r + cidx] + ta) + tb) + tc;
            }
        }
        // expected cost reduction (cs:435-436)
        double hd0 = 0.5 * dd[0], hd1 = 0.5 * dd[1];
        double g0 = hd0 * Quu[0] + hd1 * Quu[2];
        double g1 = hd0 * Quu[1] + hd1 * Quu[3];
        dV[0] += g0 * dd[0] + g1 * dd[1];
        dV[1] += dd[0] * Qu[0] + dd[1] * Qu[1];
    }
    return true;
}

} // namespace cilqr
