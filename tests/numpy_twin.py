"""A second, independent restatement of the reference's CILQR path in NumPy — TEST INFRASTRUCTURE.

Why it exists: the reference's C++ (Eigen, yaml-cpp, spdlog, fmt) cannot be built in this image and upstream has
no tests, so the C oracle (oracle/cilqr_oracle.c) is "pinned by reading" for everything the reference's Python
modules do not cover: the RearCenter model, the road-border terms, the C++ line search / regularisation schedule /
termination rules, the first-local-minimum reference scan, the augmented-Lagrangian branch.  This module is a
second reading of the same sources, written matrix-style with NumPy the way the C++ is written with Eigen
(whole-matrix expressions, `@` products) instead of the oracle's scalar loops, so that a transcription error in
either restatement shows up as a disagreement (tests/test_numpy_twin.py).  It is not bit-compatible with anything:
NumPy associates sums differently from Eigen; agreement is checked to ~1e-10 on single stages and on decision
traces of well-conditioned solves.

Citations: cs = /root/reference/src/cilqr_solver.cpp, ut = /root/reference/src/utils.cpp,
hpp = /root/reference/include/cilqr_solver.hpp.
"""
import math

import numpy as np

RUNNING, CONVERGED, BACKWARD_PASS_FAIL, FORWARD_PASS_FAIL, FORWARD_PASS_SMALL_STEP = range(5)
EPS = 1e-5  # include/utils.hpp:28


def sign(v):  # utils.hpp:110-117
    return -1 if v < 0 else 1


# ---- ut:262-283 -------------------------------------------------------------------------------------
def kinematic_propagate(x, u, dt, wb, rear):
    beta = math.atan(math.tan(u[1]) / 2)
    if rear:
        return np.array([x[0] + x[2] * math.cos(x[3]) * dt,
                         x[1] + x[2] * math.sin(x[3]) * dt,
                         x[2] + u[0] * dt,
                         x[3] + x[2] * math.tan(u[1]) * dt / wb])
    return np.array([x[0] + x[2] * math.cos(beta + x[3]) * dt,
                     x[1] + x[2] * math.sin(beta + x[3]) * dt,
                     x[2] + u[0] * dt,
                     x[3] + 2 * x[2] * math.sin(beta) * dt / wb])


# ---- ut:285-342 -------------------------------------------------------------------------------------
def model_derivatives(x, u, dt, wb, N, rear):
    v, yaw, delta = x[:N, 2], x[:N, 3], u[:N, 1]
    beta = np.arctan(np.tan(u[:, 1] / 2))                                  # ut:291 (the beta of the Jacobian)
    g = 0.5 * (1 + np.tan(u[:, 1]) ** 2) / (1 + 0.25 * np.tan(u[:, 1]) ** 2)  # ut:292-293
    A = np.tile(np.eye(4), (N, 1, 1))
    B = np.zeros((N, 4, 2))
    for i in range(N):
        if rear:
            A[i, 0, 2] = math.cos(yaw[i]) * dt
            A[i, 0, 3] = v[i] * (-math.sin(yaw[i])) * dt
            A[i, 1, 2] = math.sin(yaw[i]) * dt
            A[i, 1, 3] = v[i] * math.cos(yaw[i]) * dt
            A[i, 3, 2] = math.tan(delta[i]) * dt / wb
            B[i, 2, 0] = dt
            B[i, 3, 1] = (v[i] * dt / wb) / (math.cos(delta[i]) * math.cos(delta[i]))
        else:
            a = beta[i] + yaw[i]
            A[i, 0, 2] = math.cos(a) * dt
            A[i, 0, 3] = v[i] * (-math.sin(a)) * dt
            A[i, 1, 2] = math.sin(a) * dt
            A[i, 1, 3] = v[i] * math.cos(a) * dt
            A[i, 3, 2] = 2 * math.sin(beta[i]) * dt / wb
            B[i, 0, 1] = v[i] * (-math.sin(a)) * dt * g[i]
            B[i, 1, 1] = v[i] * math.cos(a) * dt * g[i]
            B[i, 2, 0] = dt
            B[i, 3, 1] = (2 * v[i] * dt / wb) * math.cos(beta[i]) * g[i]
    return A, B


# ---- ut:344-439 -------------------------------------------------------------------------------------
def front_rear(state, wb, rear):
    w = wb * np.array([math.cos(state[3]), math.sin(state[3])])
    if rear:
        return state[:2] + w, state[:2].copy()
    return state[:2] + 0.5 * w, state[:2] - 0.5 * w


def front_rear_derivatives(yaw, wb, rear):
    h = 0.5 * wb
    f = np.array([[1, 0], [0, 1], [0, 0], [h * (-math.sin(yaw)), h * math.cos(yaw)]], dtype=float)
    r = np.array([[1, 0], [0, 1], [0, 0], [-h * (-math.sin(yaw)), -h * math.cos(yaw)]], dtype=float)
    if rear:
        f[3] = [wb * (-math.sin(yaw)), wb * math.cos(yaw)]
        r[3] = [0, 0]
    return f, r


def ellipse_ab(width, length, d_safe, radius):
    return np.array([0.5 * length + d_safe * 6 + radius, 0.5 * width + d_safe + radius])  # ut:389-390


def rot(theta):
    return np.array([[math.cos(theta), math.sin(theta)], [-math.sin(theta), math.cos(theta)]])


def safety_margin(pnt, obs, ab):
    s = rot(obs[2]) @ (pnt - obs[:2])
    return 1 - (s[0] ** 2 / ab[0] ** 2 + s[1] ** 2 / ab[1] ** 2)


def safety_margin_derivatives(pnt, obs, ab):
    R = rot(obs[2])
    s = R @ (pnt - obs[:2])
    return np.eye(2) @ R.T @ np.array([-2 * s[0] / ab[0] ** 2, -2 * s[1] / ab[1] ** 2])


class Twin:
    """State and methods of CILQRSolver (hpp:31-148); `p` is any object with the cilqr_params field names."""

    def __init__(self, p):
        self.p = p
        self.N = int(p.N)
        self.rear = int(p.reference_point) == 0
        self.alm = int(p.solve_type) == 1
        self.W = np.diag([p.w_pos, p.w_pos, p.w_vel, p.w_yaw])
        self.R = np.diag([p.w_acc, p.w_stl])
        self.is_first_solve = True
        self.status = RUNNING
        self.last_u = None
        self.alm_rho, self.alm_mu, self.alm_mu_next = None, None, None
        self.l_x = self.l_u = self.l_xx = self.l_uu = None
        self.ab = ellipse_ab(p.width, p.length, p.d_safe, 0.5 * p.width)  # cs:78, cs:330

    # -- cs:289-314 ----------------------------------------------------------------------------------
    def ref_exact_points(self, x, lane):
        lx, ly, lyaw = lane
        out = np.zeros((x.shape[0], 3))
        idx = np.zeros(x.shape[0], dtype=int)
        start = 0
        for i in range(x.shape[0]):
            min_idx, min_d = -1, np.finfo(float).max
            for j in range(start, len(lx)):
                d = math.hypot(x[i, 0] - lx[j], x[i, 1] - ly[j])
                if min_idx < 0 or d < min_d:
                    min_idx, min_d = j, d
                else:
                    break
            out[i] = (lx[min_idx], ly[min_idx], lyaw[min_idx])
            idx[i] = min_idx
            start = min_idx
        return out, idx

    def obstacle_constr(self, xk, obs):  # cs:326-335
        f, r = front_rear(xk, self.p.wheelbase, self.rear)
        return np.array([safety_margin(f, obs, self.ab), safety_margin(r, obs, self.ab)])

    def obstacle_constr_derivatives(self, xk, obs):  # cs:715-739
        f, r = front_rear(xk, self.p.wheelbase, self.rear)
        fs, rs = front_rear_derivatives(xk[3], self.p.wheelbase, self.rear)
        return fs @ safety_margin_derivatives(f, obs, self.ab), rs @ safety_margin_derivatives(r, obs, self.ab)

    def _bounds(self, uk, xk, refk, borders):
        p = self.p
        d_sign = (xk[1] - refk[1]) * math.cos(refk[2]) - (xk[0] - refk[0]) * math.sin(refk[2])
        cur_d = sign(d_sign) * math.hypot(xk[0] - refk[0], xk[1] - refk[1])
        c = [uk[0] - p.acc_max, p.acc_min - uk[0], uk[1] - p.stl_lim, -p.stl_lim - uk[1],
             xk[2] - p.velo_max, p.velo_min - xk[2],
             cur_d - (borders[0] - p.width / 2), (borders[1] + p.width / 2) - cur_d]
        return c, d_sign

    # -- cs:199-287 ----------------------------------------------------------------------------------
    def total_cost(self, u, x, lane, ref_velo, obs, tick, borders):
        p, N = self.p, self.N
        ref, _ = self.ref_exact_points(x, lane)
        ref_states = np.column_stack([ref[:, :2], np.full(N + 1, ref_velo), ref[:, 2]])
        e = x - ref_states
        J = np.trace(e @ self.W @ e.T) + np.trace(u @ self.R @ u.T)
        Jb = 0.0
        for k in range(1, N + 1):
            c, _ = self._bounds(u[k - 1], x[k], ref[k], borders)
            if self.alm:
                Jk = sum(self.alm_rho * max(ci + self.alm_mu[k - 1, i] / self.alm_rho, 0.0) ** 2 / 2 for i, ci in enumerate(c))
            else:
                Jk = sum(p.state_exp_q1 * math.exp(p.state_exp_q2 * ci) for ci in c)
            for j in range(obs.shape[0]):
                oc = self.obstacle_constr(x[k], obs[j, tick + k])
                for m in range(2):
                    if self.alm:
                        Jk += self.alm_rho * max(oc[m] + self.alm_mu[k - 1, 8 + 2 * j + m] / self.alm_rho, 0.0) ** 2 / 2
                    else:
                        Jk += p.obstacle_exp_q1 * math.exp(p.obstacle_exp_q2 * oc[m])
            Jb += Jk
        return J + Jb

    # -- cs:692-713 ----------------------------------------------------------------------------------
    def _term(self, c, c_dot, q1, q2, mu):
        c_dot = np.asarray(c_dot, dtype=float)
        if self.alm:
            if (c + mu / self.alm_rho) > 0:
                b_dot = self.alm_rho * (c + mu / self.alm_rho) * c_dot
                return b_dot, np.outer(b_dot, c_dot)
            return np.zeros_like(c_dot), np.zeros((c_dot.size, c_dot.size))
        b = q1 * math.exp(q2 * c)
        return q2 * b * c_dot, q2 ** 2 * b * np.outer(c_dot, c_dot)

    # -- cs:463-690 ----------------------------------------------------------------------------------
    def cost_derivatives(self, u, x, lane, ref_velo, obs, tick, borders):
        p, N = self.p, self.N
        if (not self.alm) and self.status not in (RUNNING, FORWARD_PASS_SMALL_STEP):
            self.status = RUNNING
            return
        self.status = RUNNING
        ref, _ = self.ref_exact_points(x, lane)
        ref_states = np.column_stack([ref[:, :2], np.full(N + 1, ref_velo), ref[:, 2]])
        l_u = 2 * (u @ self.R)
        l_uu = np.tile(2 * self.R, (N, 1, 1))
        l_x = 2 * (x - ref_states) @ self.W
        l_xx = np.tile(2 * self.W, (N + 1, 1, 1))
        for k in range(1, N + 1):
            uk, xk, rk = u[k - 1], x[k], ref[k]
            c, d_sign = self._bounds(uk, xk, rk, borders)
            h = math.hypot(xk[0] - rk[0], xk[1] - rk[1])
            up = np.array([(xk[0] - rk[0]) / h, (xk[1] - rk[1]) / h, 0, 0])
            if d_sign < 0:
                up = -1 * up
            lo = -1 * up
            dots_u = ([1.0, 0.0], [-1, 0], [0.0, 1.0], [0, -1.0])
            dots_x = ([0, 0, 1, 0], [0, 0, -1, 0], up, lo)
            mu = self.alm_mu[k - 1] if self.alm else np.zeros(8 + 2 * obs.shape[0])
            gu, Hu = np.zeros(2), np.zeros((2, 2))
            for i in range(4):
                g, H = self._term(c[i], dots_u[i], p.state_exp_q1, p.state_exp_q2, mu[i])
                gu, Hu = gu + g, Hu + H
            gx, Hx = np.zeros(4), np.zeros((4, 4))
            for i in range(4):
                g, H = self._term(c[4 + i], dots_x[i], p.state_exp_q1, p.state_exp_q2, mu[4 + i])
                gx, Hx = gx + g, Hx + H
            if self.alm:
                for i in range(8):
                    self.alm_mu_next[k - 1, i] = min(max(mu[i] + self.alm_rho * c[i], 0.0), p.max_mu)
            for j in range(obs.shape[0]):
                o = obs[j, tick + k]
                oc = self.obstacle_constr(xk, o)
                df, dr = self.obstacle_constr_derivatives(xk, o)
                g1, H1 = self._term(oc[0], df, p.obstacle_exp_q1, p.obstacle_exp_q2, mu[8 + 2 * j])
                g2, H2 = self._term(oc[1], dr, p.obstacle_exp_q1, p.obstacle_exp_q2, mu[9 + 2 * j])
                if self.alm:
                    self.alm_mu_next[k - 1, 8 + 2 * j] = min(max(mu[8 + 2 * j] + self.alm_rho * oc[0], 0.0), p.max_mu)
                    self.alm_mu_next[k - 1, 9 + 2 * j] = min(max(mu[9 + 2 * j] + self.alm_rho * oc[1], 0.0), p.max_mu)
                gx, Hx = gx + (g1 + g2), Hx + (H1 + H2)
            l_u[k - 1] += gu
            l_uu[k - 1] += Hu
            l_x[k] += gx
            l_xx[k] += Hx
        self.l_x, self.l_u, self.l_xx, self.l_uu = l_x, l_u, l_xx, l_uu

    # -- cs:383-440 ----------------------------------------------------------------------------------
    def backward_pass(self, u, x, lamb, lane, ref_velo, obs, tick, borders):
        p, N = self.p, self.N
        self.cost_derivatives(u, x, lane, ref_velo, obs, tick, borders)
        A, B = model_derivatives(x, u, p.dt, p.wheelbase, N, self.rear)
        dV = np.zeros(2)
        d = np.zeros((N, 2))
        K = np.zeros((N, 2, 4))
        V_x, V_xx = self.l_x[N].copy(), self.l_xx[N].copy()
        for i in range(N - 1, -1, -1):
            Q_x = self.l_x[i] + A[i].T @ V_x
            Q_u = self.l_u[i] + B[i].T @ V_x
            Q_xx = self.l_xx[i] + A[i].T @ V_xx @ A[i]
            Q_uu = self.l_uu[i] + B[i].T @ V_xx @ B[i] + lamb * np.eye(2)
            Q_ux = B[i].T @ V_xx @ A[i]
            # Eigen::LLT on the lower triangle: fails iff a pivot is <= 0 (NaN pivots do not fail)
            if Q_uu[0, 0] <= 0.0 or (Q_uu[1, 1] - (Q_uu[1, 0] / math.sqrt(Q_uu[0, 0])) ** 2) <= 0.0:
                self.status = BACKWARD_PASS_FAIL
                return d, K, dV
            det = Q_uu[0, 0] * Q_uu[1, 1] - Q_uu[1, 0] * Q_uu[0, 1]
            inv = np.array([[Q_uu[1, 1], -Q_uu[0, 1]], [-Q_uu[1, 0], Q_uu[0, 0]]]) * (1.0 / det)
            d[i] = -inv @ Q_u
            K[i] = -inv @ Q_ux
            V_x = Q_x + K[i].T @ Q_uu @ d[i] + K[i].T @ Q_u + Q_ux.T @ d[i]
            V_xx = Q_xx + K[i].T @ Q_uu @ K[i] + K[i].T @ Q_ux + Q_ux.T @ K[i]
            dV[0] += 0.5 * d[i] @ Q_uu @ d[i]
            dV[1] += d[i] @ Q_u
        return d, K, dV

    # -- cs:442-461 ----------------------------------------------------------------------------------
    def forward_pass(self, u, x, d, K, alpha):
        p, N = self.p, self.N
        nu, nx = np.zeros((N, 2)), np.zeros((N + 1, 4))
        nx[0] = x[0]
        for i in range(N):
            nu[i] = u[i] + K[i] @ (nx[i] - x[i]) + alpha * d[i]
            nx[i + 1] = kinematic_propagate(nx[i], nu[i], p.dt, p.wheelbase, self.rear)
        return nu, nx

    # -- cs:337-381 ----------------------------------------------------------------------------------
    def iter_step(self, u, x, lamb, args, flag):
        p = self.p
        ori = self.total_cost(u, x, *args)
        d, K, dV = self.backward_pass(u, x, lamb, *args)
        if self.status == BACKWARD_PASS_FAIL:
            return u, x, ori, flag, 0, -1
        flag = False
        alpha, trials, idx = 1.0, 0, 0
        new_J = np.finfo(float).max
        nu, nx = u, x
        while alpha > 1e-6:
            nu, nx = self.forward_pass(u, x, d, K, alpha)
            new_J = self.total_cost(nu, nx, *args)
            trials += 1
            decay = ori - new_J
            if abs(alpha - 1.0) < EPS and abs(decay) < p.convergence_threshold:
                self.status = CONVERGED
                return nu, nx, new_J, flag, trials, idx
            approx = -(alpha * alpha * dV[0] + alpha * dV[1])
            if decay > 0.0 and (approx < 0.0 or decay / approx > p.accept_step_threshold):
                if abs(alpha - 1.0) > EPS:
                    self.status = FORWARD_PASS_SMALL_STEP
                return nu, nx, new_J, True, trials, idx
            alpha *= 0.5
            idx += 1
        if self.alm:
            self.alm_mu = self.alm_mu_next.copy()
            self.alm_rho = min((1 + p.alm_gamma) * self.alm_rho, p.max_rho)
        self.status = FORWARD_PASS_FAIL
        return nu, nx, new_J, flag, trials, -1

    # -- cs:85-153 -----------------------------------------------------------------------------------
    def solve(self, x0, lane, ref_velo, obs, tick, borders):
        p, N = self.p, self.N
        M = obs.shape[0]
        if self.alm and ((not p.use_last_solution) or (p.use_last_solution and self.is_first_solve)):
            self.alm_rho = p.alm_rho_init
            self.alm_mu = np.zeros((N, 8 + 2 * M))
            self.alm_mu_next = np.zeros((N, 8 + 2 * M))
        self.status = RUNNING
        x0 = np.asarray(x0, dtype=float)
        if (not self.is_first_solve) and p.use_last_solution:
            u = np.zeros((N, 2))
            u[:N - 1] = self.last_u[1:]
            u[N - 1] = self.last_u[N - 1]
        else:
            u = np.zeros((N, 2))
            self.is_first_solve = False
        x = np.zeros((N + 1, 4))
        x[0] = x0
        for i in range(N):
            x[i + 1] = kinematic_propagate(x[i], u[i], p.dt, p.wheelbase, self.rear)
        args = (lane, ref_velo, obs, tick, borders)
        J_init = self.total_cost(u, x, *args)
        lamb = p.init_lamb
        flag = False
        trace = []
        end = "MAX_ITER"
        for _ in range(int(p.max_iter)):
            nu, nx, new_J, flag, trials, aidx = self.iter_step(u, x, lamb, args, flag)
            if flag:
                x, u = nx, nu
            if self.status in (BACKWARD_PASS_FAIL, FORWARD_PASS_FAIL):
                lamb = max(p.lamb_amplify, lamb * p.lamb_amplify)
            elif self.status == RUNNING:
                lamb *= p.lamb_decay
            trace.append((self.status, trials, int(flag), aidx, lamb, new_J))
            if lamb > p.max_lamb:
                end = "MAX_LAMB"
                break
            elif self.status == CONVERGED:
                end = "CONVERGED"
                break
        self.last_u = u.copy()
        return {"u": u, "x": x, "J_init": J_init, "J_final": self.total_cost(u, x, *args), "trace": trace, "end": end}
