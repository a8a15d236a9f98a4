// cilqr_device.hpp — gfx950 device code of the batched CILQR solve path.
//
// Mapping (one wavefront = one trajectory, FP64 throughout, no MFMA):
//   * phases that are parallel over the horizon (stage cost, cost gradients/Hessians, model
//     Jacobians, reference-point search + proof) run with lane = time step k;
//   * the line search runs with lane = trial step size: lane a rolls the closed-loop dynamics out
//     with alpha = 2^-a, so all 20 trial trajectories of cs:354 cost one pass of the serial
//     instruction stream; the trial costs are then evaluated (lane = k again) in the reference's
//     order until the first trial that the reference would have accepted;
//   * the backward Riccati-like sweep is serial over the horizon; inside a step lane = matrix
//     element of a 6x8 grid, operands move between lanes with DPP / ds_bpermute / v_readlane
//     (backward_sweep_lanes); a wave-uniform form (backward_sweep_uniform) is kept as a testing aid;
//   * x, u, l_*, the Jacobians A, B and — in the same array, once a step's Jacobians are consumed — the
//     gains K, d, the cost model's constants and a window of the lane table live in LDS; the 20 trial
//     trajectories live in an L2-resident scratch slab; obstacle routes are read through L1/L2 (shared
//     by the batch).
//
// Arithmetic contract: every value is computed with the same IEEE operations in the same order
// as oracle/cilqr_oracle.c (which restates the reference's Eigen expressions); where structural
// zeros are dropped (x + 0*y == x) the value is unchanged for finite data — so results are
// bit-identical to the oracle's detmath build.  Compile with -ffp-contract=off.
//
// Reference citations: "cs:" = /root/reference/src/cilqr_solver.cpp, "ut:" = /root/reference/src/utils.cpp,
// "hpp:" = /root/reference/include/cilqr_solver.hpp.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cilqr_amd.h"
#include "detmath.h"

#define CILQR_WAVE 64

// The FUSED flavour (round-4 experiment, see oracle/cilqr_oracle.c): -DCILQR_FUSED turns the multiply-adds of four named
// groups of sites — F1 the quadratic forms of the stage cost, F2 every product of the backward sweep, F3 the rollout's
// K dx + alpha d, F4 the affine updates of the kinematic step — into explicit fma, in the association the oracle's
// -DORC_FUSED build uses.  Default: the plain product and sum (the build the library ships).
#ifdef CILQR_FUSED
#define CQ_MADD(a, b, c) __builtin_fma((a), (b), (c))
#else
#define CQ_MADD(a, b, c) ((a) * (b) + (c))
#endif
#define CILQR_DBG_SERIAL_REF_SCAN 1 /* cilqr_set_debug_flags: always take the serial reference-point chain */
#define CILQR_DBG_UNIFORM_BACKWARD 2 /* use the wave-uniform backward sweep instead of the lane-parallel one */

namespace cilqr {

// Ordering point between the lanes of ONE wavefront that exchange data through LDS / the scratch
// slab: the wave's outstanding memory operations complete and the compiler may not move accesses
// across it.  No s_barrier: kernels may run more than one wavefront per block (helper waves), and a
// block-wide barrier inside single-wave phases would deadlock.
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// pointers into HBM are typed as address space 1 so that every access is a global_load (a generic
// pointer would compile to flat_load, which also ties up the LDS counter)
typedef const double __attribute__((address_space(1))) gdouble;
typedef double __attribute__((address_space(1))) gdouble_w;  // HBM, writable (global_store, not flat_store)
#define CILQR_OBS_STRIDE 5  /* device obstacle record: x, y, yaw, sin(yaw), cos(yaw) */
#define CILQR_AUX_STRIDE 4  /* device lane record: yaw, sin(yaw), cos(yaw), unused */

struct DevScene {
    const double* lane_xy;  // [L][2]
    const double* lane_aux; // [L][4]  yaw, sin, cos (filled on the device with dm_sincos: same bits as in-kernel)
    const double* obs;      // [M][T][5] x, y, yaw, sin, cos
    int L, M, T, pad;
    double border_hi, border_lo, ref_velo;
    double cert_rcap, cert_smax; // convexity certificate of the lane table (see convex_interior), filled by the host
};

// Wave-uniform constants of one trajectory.  The few that the serial sweeps and the control flow use
// travel by value (scalar registers); the cost model's constants sit in LDS (CstK) and are read where a
// cost or a derivative is evaluated — as one struct by value they would claim ~100 scalar registers for
// the whole solve and keep the register allocator spilling around every loop.
struct CstK {
    double w_pos, w_vel, w_yaw, w_acc, w_stl;
    double sq1, sq2, oq1, oq2;
    double acc_max, acc_min, stl_lim, velo_max, velo_min;
    double pos_up_b, pos_lo_b;
    double ell_a2, ell_b2;
    double ref_velo;
    double cert_rcap, cert_smax;
    double init_lamb, lamb_decay, lamb_amplify, max_lamb, conv_thr, accept_thr;
    double alm_rho_init, alm_gamma, max_rho, max_mu;
    double inv_a2, inv_b2; // correctly rounded reciprocals of the ellipse axes squared (div_by_const)
    double pad;
};
#define CILQR_CSTK_DOUBLES ((int)(sizeof(CstK) / sizeof(double)))

struct Cst {
    int N, rp, M, L, T, tick, max_iter, pad;
    double dt, wb, half_wb;
    gdouble* lane_xy;
    gdouble* lane_aux;
    gdouble* obs;
    const CstK* k;
};

// fills the by-value part on every lane and, from lane 0, the LDS part at `ck` (followed by a wave_sync:
// both wavefronts of a helper-mode block do this, writing identical values)
// (fill = false: the LDS part is there already — a trajectory of a grouped wavefront coming back for its next segment)
__device__ inline void make_cst(Cst& c, const cilqr_params& p, const DevScene& s, int tick, CstK* ck, int lane, bool fill = true);

__device__ inline void make_cst(Cst& c, const cilqr_params& p, const DevScene& s, int tick, CstK* ck, int lane, bool fill) {
    c.N = p.N; c.rp = p.reference_point; c.M = s.M; c.L = s.L; c.T = s.T; c.tick = tick;
    c.max_iter = p.max_iter; c.pad = 0;
    c.dt = p.dt; c.wb = p.wheelbase; c.half_wb = 0.5 * p.wheelbase;
    c.lane_xy = (gdouble*)s.lane_xy; c.lane_aux = (gdouble*)s.lane_aux; c.obs = (gdouble*)s.obs;
    c.k = ck;
    if (!fill) return;
    if (lane == 0) {
        CstK k;
        k.w_pos = p.w_pos; k.w_vel = p.w_vel; k.w_yaw = p.w_yaw; k.w_acc = p.w_acc; k.w_stl = p.w_stl;
        k.sq1 = p.state_exp_q1; k.sq2 = p.state_exp_q2; k.oq1 = p.obstacle_exp_q1; k.oq2 = p.obstacle_exp_q2;
        k.acc_max = p.acc_max; k.acc_min = p.acc_min; k.stl_lim = p.stl_lim;
        k.velo_max = p.velo_max; k.velo_min = p.velo_min;
        k.pos_up_b = s.border_hi - p.width / 2; // cs:239
        k.pos_lo_b = s.border_lo + p.width / 2; // cs:241
        // ut:387-393 with obs_attr = {width, length, d_safe} (cs:78) and ego_pnt_radius = 0.5*width (cs:330)
        double a = 0.5 * p.length + p.d_safe * 6 + 0.5 * p.width;
        double b = 0.5 * p.width + p.d_safe + 0.5 * p.width;
        k.ell_a2 = a * a;
        k.ell_b2 = b * b;
        k.ref_velo = s.ref_velo;
        k.cert_rcap = s.cert_rcap; k.cert_smax = s.cert_smax;
        k.init_lamb = p.init_lamb; k.lamb_decay = p.lamb_decay; k.lamb_amplify = p.lamb_amplify;
        k.max_lamb = p.max_lamb; k.conv_thr = p.convergence_threshold; k.accept_thr = p.accept_step_threshold;
        k.alm_rho_init = p.alm_rho_init; k.alm_gamma = p.alm_gamma; k.max_rho = p.max_rho; k.max_mu = p.max_mu;
        k.inv_a2 = 1.0 / k.ell_a2;
        k.inv_b2 = 1.0 / k.ell_b2;
        k.pad = 0.0;
        *ck = k;
    }
    wave_sync();
}

// LDS carve-out for one trajectory (offsets in doubles).  l_xx keeps the 7 entries that can be
// non-zero: (0,0) (0,1) (0,3) (1,1) (1,3) (3,3) (2,2); l_uu is diagonal in barrier mode.
struct Lds {
    double* x;   // [(N+1)][4]
    double* u;   // [N][2]
    double* kd;  // [N][10] per step: the model Jacobians (a02 a03 a12 a13 a32 | b01 b11 b31 | - -) until the
                 // backward sweep has consumed them, then the gains (K[0][:] d[0] | K[1][:] d[1]) of that step
    double* lx;  // [(N+1)][4]
    double* lu;  // [N][2]
    double* lxx; // [(N+1)][lxs]: 7 packed entries (barrier mode: symmetric) or 16 dense (ALM mode)
    int lxs;
    double* luu; // [N][2]
    double* gl;  // derivative rows in global memory instead of lx / lu / lxx / luu (see CILQR_GL_ROW), or nullptr
    double* ring; // [CILQR_GL_RING] rows of l.gl staged for the backward sweep, eight at a time (lg builds)
    CstK* ck;    // the cost model's constants (see Cst)
    double* xch; // [CILQR_XCH] constant block of the lane-parallel backward sweep (see backward_sweep_lanes)
    double* cs;  // [slots][3][(N+1)] stage-cost scratch: state, ctrl, barrier (slots = trials costed concurrently).
                 // ALIASES kd: the gains are dead once the rollout has produced the trial trajectories, and costs
                 // are only summed then.  An iteration that rolls out the first trial alone may still need them for
                 // a second pass: it parks the 3 (N+1) doubles its lone costing overwrites behind the first-trial
                 // buffer (save_gains_head / restore_gains_head)
    double* win; // [W][2] copy of lane_xy[w0 .. w0+W): the stretch of lane the horizon can reach
    int* ridx;   // [(N+1)] lane-sample index of every row of the current trajectory
    int* tidx;   // [slots][(N+2)] the same for the trial trajectories being costed
    int* ctli;   // [8]  control words shared by the main and the helper wavefront of a block
    double* ctld; // [CILQR_CTLD]
    long long* prof; // [CILQR_PROF_SLOTS] in-kernel cycle accounting (profiling builds)
    int w0;      // first lane sample held in win (the row-0 index: scans only move forward from it)
    int W;
};

// constants the backward sweep's per-lane address maps point at: 0.0, 1.0, dt, 0.0
#define CILQR_XCH_CONST 0
#define CILQR_XCH 4
#define CILQR_KD 10 /* doubles per step of Lds::kd */
#define CILQR_KD_B 5 /* offset of b01 b11 b31 */
/* gains of a step: (K[0][0..3] d[0] | K[1][0..3] d[1]) — column c of (K | d) is written by lane c with one
 * two-address store */
#define CILQR_KD_ROW 5
#define CILQR_KD_K(e) ((e) < 4 ? (e) : (e) + 1) /* K[e / 4][e % 4], e = 0..7 */
#define CILQR_KD_D(j) (4 + CILQR_KD_ROW * (j))  /* d[j] */
#define CILQR_CTLD 8 /* doubles shared by the main and the helper wavefront (k_solve) */
#define CILQR_NT 2 /* trial trajectories costed per pass (after the first): their memory latencies overlap */

// slots = trial trajectories costed concurrently by a block: 2 with a helper wavefront or paired passes, else 1
__host__ __device__ inline int kd_doubles(int N, int slots) {
    const int g = CILQR_KD * N, c = slots * 3 * (N + 1); // gains / stage-cost scratch share the array
    return g > c ? g : c;
}
// The cost expansion in global memory ("lg" builds: long horizons in large batches, barrier mode).  The expansion is
// 15 (N + 1) doubles — 12 KB of the 25.5 KB a horizon of 100 needs in LDS, which caps a CU at 5 trajectories.  It is
// written once per iteration by the lane = row phase and read by the backward sweep strictly in order, one row per
// step: exactly what a prefetched stream out of L2 serves.  One 128-byte row per step:
//   0-3 l_x | 4-5 l_u | 6-11 l_xx packed (00 01 03 11 13 33) | 12-13 l_uu diagonal | 14 l_xx (22) | 15 zero
// (slot 15 is what the lanes of the structurally zero entries of l read, in range and uniform with the others)
#define CILQR_GL_ROW 16
#define CILQR_GL_LX 0
#define CILQR_GL_LU 4
#define CILQR_GL_LXX 6
#define CILQR_GL_LUU 12
#define CILQR_GL_LXX22 14
#define CILQR_GL_ZERO 15
// augmented-Lagrangian builds (l_xx dense, 16 entries): one 256-byte row per step,
//   0-3 l_x | 4-5 l_u | 6-7 l_uu diagonal | 8-23 l_xx row-major | 24 zero | 25-31 unused
#define CILQR_GL_ROW_ALM 32
#define CILQR_GLA_LX 0
#define CILQR_GLA_LU 4
#define CILQR_GLA_LUU 6
#define CILQR_GLA_LXX 8
#define CILQR_GLA_ZERO 24
// The sweep reads them through a ring of 128 doubles in LDS (8 rows, or 4 of the 256-byte rows) that is refilled 64
// doubles — one per lane, four rows or two — at a time: the global load of a chunk is issued a chunk's steps before its
// rows are needed and costs two instructions.
#define CILQR_GL_RING 128
__host__ __device__ inline int lds_doubles(int N, int alm, int slots, int lg = 0) {
    const int expansion = lg ? CILQR_GL_RING : 4 * (N + 1) + 2 * N + (alm ? 16 : 7) * (N + 1) + 2 * N;
    return 4 * (N + 1) + 2 * N + kd_doubles(N, slots) + expansion + CILQR_XCH + CILQR_CTLD + CILQR_CSTK_DOUBLES + CILQR_PROF_SLOTS;
}
__host__ __device__ inline size_t lds_bytes(int N, int W, int alm, int slots, int lg = 0) {
    return sizeof(double) * ((size_t)lds_doubles(N, alm, slots, lg) + 2 * (size_t)W) + sizeof(int) * (size_t)(((1 + slots) * (N + 2) + 8 + 1) & ~1);
}

__device__ inline void carve(Lds& l, double* base, int N, int W, int alm, int slots, int lg = 0) {
    double* p = base;
    l.x = p; p += 4 * (N + 1);
    l.u = p; p += 2 * N;
    l.kd = p; p += kd_doubles(N, slots);
    l.lxs = alm ? 16 : 7;
    l.gl = nullptr;
    l.ring = nullptr;
    if (lg) {
        l.lx = l.lu = l.lxx = l.luu = nullptr; // the expansion lives in l.gl (set by the kernel)
        l.ring = p; p += CILQR_GL_RING;
    } else {
        l.lx = p; p += 4 * (N + 1);
        l.lu = p; p += 2 * N;
        l.lxx = p; p += l.lxs * (N + 1);
        l.luu = p; p += 2 * N;
    }
    l.xch = p; p += CILQR_XCH;
    l.cs = l.kd;
    l.ctld = p; p += CILQR_CTLD;
    l.ck = reinterpret_cast<CstK*>(p); p += CILQR_CSTK_DOUBLES;
    l.prof = reinterpret_cast<long long*>(p); p += CILQR_PROF_SLOTS;
    // the index arrays next (an even number of ints), the lane window — the only part whose size is not a function
    // of the horizon — last: with a compile-time horizon every other offset is a constant
    const int n_int = ((1 + slots) * (N + 2) + 8 + 1) & ~1;
    l.ridx = reinterpret_cast<int*>(p);
    l.tidx = l.ridx + (N + 2);
    l.ctli = l.tidx + slots * (N + 2);
    p += n_int / 2;
    l.win = p; p += 2 * W;
    l.w0 = 0;
    l.W = 0; // nothing staged yet: every lookup goes to global memory
}

// lane sample j: from the LDS window when it is inside, from global memory otherwise
__device__ inline void lane_point(const Cst& c, const Lds& l, int j, double& px, double& py) {
    unsigned o = (unsigned)(j - l.w0);
    if (o < (unsigned)l.W) {
        px = l.win[2 * o];
        py = l.win[2 * o + 1];
    } else {
        px = c.lane_xy[2 * j];
        py = c.lane_xy[2 * j + 1];
    }
}

// stage lane_xy[w0 .. w0+Wcap) into LDS (clipped to the table); call with all lanes
__device__ inline void stage_window(const Cst& c, Lds& l, int w0, int Wcap, int lane) {
    int W = c.L - w0;
    W = (W < Wcap) ? W : Wcap;
    for (int e = lane; e < 2 * W; e += CILQR_WAVE) l.win[e] = c.lane_xy[2 * (size_t)w0 + e];
    l.w0 = w0;
    l.W = W;
    wave_sync();
}

// scratch slab of the trial trajectories: [3 pairs][row tiles][20 alphas][4 rows][2] doubles — components in pairs (x0 x1 |
// x2 x3 | u0 u1); rows in tiles of CILQR_SLAB_TILE = 4, so that the four rows a step size owns in a tile are ONE 64-byte
// sector: a rollout lane fills it with four consecutive 16-byte stores (complete sectors leave L2, the 20 lanes of a store
// instruction write 20 neighbouring sectors), and the trial costs — lane = row, one step size at a time — fetch a sector
// per FOUR rows instead of one per row.  (Rows next to each other, alpha in between — the layout until round 4 — made every
// 16 bytes the costs read bring a whole sector in: 4 x the slab's bytes, the largest single part of the kernel's fetch
// traffic.)  Components 0-3 = x', 4-5 = u'.  TR(t, c, k) with t = TRIAL_AT(slab, alpha).
// Behind the slab, the same for the alpha = 1 trial alone (the "first-trial buffer", [3 pairs][N + 1 rows, padded to whole
// tiles][2]): most iterations accept that trial, so they roll out, cost and accept only it — 2.4 KB written and read back
// contiguously, L2-resident — and the slab is written only in iterations expected (or found) to search deeper.  Both are
// addressed as TRS(t, c, k, as): as = 20 inside the slab (t = TRIAL_AT(slab, alpha)), as = 1 in the first-trial buffer.
#define CILQR_TRIAL_ROWS 6
#ifndef CILQR_SLAB_TILE
#define CILQR_SLAB_TILE 4 // (1 = the untiled layout, kept for A/B runs)
#endif
#define CILQR_SLAB_RT(R) (((R) + CILQR_SLAB_TILE - 1) / CILQR_SLAB_TILE)
#define TRS(t, c, k, as)                                                                                              \
    (t)[(((size_t)((c) >> 1) * CILQR_SLAB_RT(R) + (size_t)((k) / CILQR_SLAB_TILE)) * (size_t)(as)) * (2 * CILQR_SLAB_TILE) + \
        (size_t)((k) % CILQR_SLAB_TILE) * 2 + (size_t)((c) & 1)]
#define TRIAL_AT(base, alpha) ((base) + 2 * CILQR_SLAB_TILE * (alpha))
#define TR(t, c, k) TRS(t, c, k, CILQR_MAX_ALPHA_TRIALS)
__host__ __device__ inline size_t first_trial_doubles(int N) { // one trial: 3 pairs of whole row tiles
    return (size_t)3 * CILQR_SLAB_RT(N + 1) * 2 * CILQR_SLAB_TILE;
}
__host__ __device__ inline size_t slab_doubles(int N) { return (size_t)CILQR_MAX_ALPHA_TRIALS * first_trial_doubles(N); }
__host__ __device__ inline size_t scratch_gl_offset(int N) { // rows of the cost expansion ("lg" builds), 128-byte aligned
    const size_t head = slab_doubles(N) + first_trial_doubles(N) + (size_t)3 * (size_t)(N + 1); // + first-trial buffer + parked gains
    return (head + 15) / 16 * 16;
}
__host__ __device__ inline size_t scratch_doubles(int N) {
    return scratch_gl_offset(N) + (size_t)CILQR_GL_ROW_ALM * (size_t)(N + 1); // (sized for the wider, ALM rows)
}

// ---------------------------------------------------------------------------------------------
// ut:262-283 kinematic_propagate
// TRIG = flavour of the elementary functions (detmath.h: DM_PIN, DM_NOSHORT, DM_SMALL)
// (dt, wb by value: wave-uniform scalars for the one-trajectory rollouts, per-lane values where the lanes of one pass
//  belong to different trajectories — rollout_group)
template <int RP, int TRIG = 0>
__device__ inline void propagate_v(const double dt, const double wb, const double x[4], const double u[2], double xn[4],
                                   const DmPinned* pk = nullptr) {
    if (RP == 0) {
        double sn, cs;
        dm_sincos<TRIG>(x[3], &sn, &cs, pk);
        double tn = dm_tan<TRIG>(u[1], pk);
        xn[0] = CQ_MADD(x[2] * cs, dt, x[0]);
        xn[1] = CQ_MADD(x[2] * sn, dt, x[1]);
        xn[2] = CQ_MADD(u[0], dt, x[2]);
        xn[3] = x[3] + x[2] * tn * dt / wb;
    } else {
        double beta = dm_atan<TRIG>(dm_tan<TRIG>(u[1], pk) / 2);
        double sn, cs;
        dm_sincos<TRIG>(beta + x[3], &sn, &cs, pk);
        double sb = dm_sin<TRIG>(beta, pk);
        xn[0] = CQ_MADD(x[2] * cs, dt, x[0]);
        xn[1] = CQ_MADD(x[2] * sn, dt, x[1]);
        xn[2] = CQ_MADD(u[0], dt, x[2]);
        xn[3] = x[3] + 2 * x[2] * sb * dt / wb;
    }
}
template <int RP, int TRIG = 0>
__device__ inline void propagate(const Cst& c, const double x[4], const double u[2], double xn[4],
                                 const DmPinned* pk = nullptr) {
    propagate_v<RP, TRIG>(c.dt, c.wb, x, u, xn, pk);
}

// The same step for small angles on every active lane, straight-line: all range reductions and
// interval selections are known to be the identity (detmath.h, DM_SMALL).  The caller has checked
// |yaw| < 0.785 and |steer| < 0.7 (so |tan(steer)| / 2 < 0.4375 and |beta| < 0.42); for the CoG model the
// angle beta + yaw is only known here: returns false, with xn untouched, when it is not small on
// some lane — the caller then redoes the step the general way.  pk: the pinned coefficients (dm_pin_load).
template <int RP, int PIN = DM_PIN>
__device__ inline bool propagate_small_v(const double dt, const double wb, const double x[4], const double u[2], double xn[4],
                                         const DmPinned* pk) {
    constexpr int T = PIN | DM_SMALL;
    if (RP == 0) {
        double sn, cs;
        dm_sincos<T>(x[3], &sn, &cs, pk);
        double tn = dm_tan<T>(u[1], pk);
        xn[0] = CQ_MADD(x[2] * cs, dt, x[0]);
        xn[1] = CQ_MADD(x[2] * sn, dt, x[1]);
        xn[2] = CQ_MADD(u[0], dt, x[2]);
        xn[3] = x[3] + x[2] * tn * dt / wb;
    } else {
        double beta = dm_atan<T>(dm_tan<T>(u[1], pk) / 2);
        const double ang = beta + x[3];
        if (!DM_WAVE_ALL(__builtin_fabs(ang) < 0.785)) return false;
        double sn, cs;
        dm_sincos<T>(ang, &sn, &cs, pk);
        double sb = dm_sin<T>(beta, pk);
        xn[0] = CQ_MADD(x[2] * cs, dt, x[0]);
        xn[1] = CQ_MADD(x[2] * sn, dt, x[1]);
        xn[2] = CQ_MADD(u[0], dt, x[2]);
        xn[3] = x[3] + 2 * x[2] * sb * dt / wb;
    }
    return true;
}
template <int RP, int PIN = DM_PIN>
__device__ inline bool propagate_small(const Cst& c, const double x[4], const double u[2], double xn[4],
                                       const DmPinned* pk) {
    return propagate_small_v<RP, PIN>(c.dt, c.wb, x, u, xn, pk);
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

// cs:295-311 for one row, lane-private: first local minimum of the distance at or after s.
// Four candidates are fetched and evaluated per trip (independent loads, one latency); candidate
// t is compared with candidate t-1 exactly as the reference's running minimum would be.
__device__ inline int ref_scan_from(const Cst& c, const Lds& l, double px, double py, int s) {
    int j = s;
    double qx, qy;
    lane_point(c, l, j, qx, qy);
    double bx = px - qx, by = py - qy;
    double best2 = bx * bx + by * by;
    for (;;) {
        double c2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int jj = j + 1 + t;
            double v = dm_inf();
            if (jj < c.L) {
                double ax, ay;
                lane_point(c, l, jj, ax, ay);
                double ex = px - ax, ey = py - ay;
                v = ex * ex + ey * ey;
            }
            c2[t] = v;
        }
        bool f0 = dist_less(c2[0], best2);
        bool f1 = dist_less(c2[1], c2[0]);
        bool f2 = dist_less(c2[2], c2[1]);
        bool f3 = dist_less(c2[3], c2[2]);
        int adv = f0 ? (f1 ? (f2 ? (f3 ? 4 : 3) : 2) : 1) : 0;
        j += adv;
        if (adv < 4) break;
        best2 = c2[3];
    }
    return j;
}

// squared distance from (px, py) to lane sample j (+inf past the end of the table)
__device__ inline double lane_d2(const Cst& c, const Lds& l, double px, double py, int j) {
    if (j >= c.L) return dm_inf();
    double ax, ay;
    lane_point(c, l, j, ax, ay);
    double ex = px - ax, ey = py - ay;
    return ex * ex + ey * ey;
}

// A local minimum of the distance profile of one row near `guess` (>= lo): walk forward while the
// distance strictly decreases, otherwise walk backward to where the strict decrease starts.  Only
// a CANDIDATE for cs:295-311 — but one that comes with two facts, established with the reference's own
// comparisons on the way: the returned m satisfies  not d(m+1) < d(m)  (the scan would stop at m; the
// table end counts as +inf), and, if m > lo,  d(m) < d(m-1)  (the scan would not have stopped at m-1).
// *q_m = squared distance at m.
__device__ inline int local_min_near(const Cst& c, const Lds& l, double px, double py, int guess, int lo,
                                     double* q_m) {
    int j = guess;
    // the three neighbours are fetched together: when the guess is already right this is all it takes
    double cur = lane_d2(c, l, px, py, j);
    double nxt = lane_d2(c, l, px, py, j + 1);
    double prv = lane_d2(c, l, px, py, (j > lo) ? j - 1 : j);
    // dec(i) = "d(i+1) < d(i)" with the reference's comparison.  A guess that is a few samples off (the small steps of
    // a line search) is settled by the first trips below; one that is far off — the first trial of an iteration of a
    // long horizon moves the late rows by a hundred samples and more — is found by doubling steps and bisection
    // between a sample where the distance still decreases and one where it does not (dec(a) && !dec(b) is kept
    // throughout, so whatever the profile looks like in between the result has the two facts above): O(log) round
    // trips to the lane table instead of one per four samples forward / one per sample backward.
    if (dist_less(nxt, cur)) {
        j += 1;
        cur = nxt;
        double n1 = lane_d2(c, l, px, py, j + 1);
        double n2 = lane_d2(c, l, px, py, j + 2);
        double n3 = lane_d2(c, l, px, py, j + 3);
        double n4 = lane_d2(c, l, px, py, j + 4);
        bool f1 = dist_less(n1, cur), f2 = dist_less(n2, n1), f3 = dist_less(n3, n2), f4 = dist_less(n4, n3);
        int adv = f1 ? (f2 ? (f3 ? (f4 ? 4 : 3) : 2) : 1) : 0;
        j += adv;
        cur = (adv == 0) ? cur : ((adv == 1) ? n1 : ((adv == 2) ? n2 : ((adv == 3) ? n3 : n4)));
        if (adv == 4) {
            // dec(j - 1) holds; look further out in doubling steps (the table end stops every walk: d(L) = +inf)
            int a = j - 1, step = 4;
            double qb;
            for (;;) {
                int b = a + step;
                b = (b > c.L - 1) ? c.L - 1 : b;
                qb = lane_d2(c, l, px, py, b);
                if (!dist_less(lane_d2(c, l, px, py, b + 1), qb)) { j = b; break; }
                a = b;
                step *= 2;
            }
            // dec(a), !dec(j), a < j
            while (j - a > 1) {
                const int mid = a + ((j - a) >> 1);
                const double qm = lane_d2(c, l, px, py, mid);
                if (dist_less(lane_d2(c, l, px, py, mid + 1), qm)) a = mid;
                else { j = mid; qb = qm; }
            }
            cur = qb;
        }
    } else if (j > lo && !dist_less(cur, prv)) {
        // !dec(j) and !dec(j - 1): the minimum lies behind the guess.  The next four samples back in one trip ...
        const int j1 = j - 1;
        const double p2 = lane_d2(c, l, px, py, (j1 - 1 > lo) ? j1 - 1 : lo);
        const double p3 = lane_d2(c, l, px, py, (j1 - 2 > lo) ? j1 - 2 : lo);
        const double p4 = lane_d2(c, l, px, py, (j1 - 3 > lo) ? j1 - 3 : lo);
        const double p5 = lane_d2(c, l, px, py, (j1 - 4 > lo) ? j1 - 4 : lo);
        // candidate m = j1 - t needs (m == lo or dec(m - 1)); !dec(m) is known for m = j1 and follows for each further
        // step back from the failed dec of the step before
        if (j1 == lo || dist_less(prv, p2)) { j = j1; cur = prv; }
        else if (j1 - 1 == lo || dist_less(p2, p3)) { j = j1 - 1; cur = p2; }
        else if (j1 - 2 == lo || dist_less(p3, p4)) { j = j1 - 2; cur = p3; }
        else if (j1 - 3 == lo || dist_less(p4, p5)) { j = j1 - 3; cur = p4; }
        else {
            // ... then doubling steps back: !dec(b) at b = j1 - 4 (> lo here), look for an a < b with dec(a), or reach lo
            int b = j1 - 4, step = 4;
            double qb = p5;
            int a;
            bool found = false;
            for (;;) {
                a = b - step;
                a = (a < lo) ? lo : a;
                const double qa = lane_d2(c, l, px, py, a);
                if (dist_less(lane_d2(c, l, px, py, a + 1), qa)) { found = true; break; }
                b = a;
                qb = qa;
                if (a == lo) break;
                step *= 2;
            }
            if (found) {
                while (b - a > 1) {
                    const int mid = a + ((b - a) >> 1);
                    const double qm = lane_d2(c, l, px, py, mid);
                    if (dist_less(lane_d2(c, l, px, py, mid + 1), qm)) a = mid;
                    else { b = mid; qb = qm; }
                }
            }
            j = b; // found: dec(b - 1) && !dec(b); not found: b == lo && !dec(lo)
            cur = qb;
        }
    }
    *q_m = cur;
    return j;
}

// Cheap sufficient form of verify_interval() for an interval that lies inside the LDS window: every
// interior comparison must hold with a safety factor (so that it also holds for the rounded
// square roots the reference compares) and d(b+1) >= d(b) (then hypot(b+1) < hypot(b) is false).
// A `false` only means "not proven this way".
__device__ inline bool verify_window_fast(const Lds& l, double px, double py, int a, int b) {
    const int o = a - l.w0, len = b - a;
    if (o < 0 || len < 0 || o + len + 1 >= l.W) return false;
    const double Ksafe = 0.99999999999999644729; // 1 - 2^-48
    const double* w = l.win + 2 * o;
    double ex = px - w[0], ey = py - w[1];
    double prev = ex * ex + ey * ey;
    bool good = true;
    for (int t = 1; t <= len; ++t) {
        ex = px - w[2 * t];
        ey = py - w[2 * t + 1];
        const double cur = ex * ex + ey * ey;
        good = good && (cur < prev * Ksafe);
        prev = cur;
    }
    ex = px - w[2 * (len + 1)];
    ey = py - w[2 * (len + 1) + 1];
    const double nxt = ex * ex + ey * ey;
    return good && (nxt >= prev);
}

// The interior of the same proof without looking at the interior, when the lane is locally convex for
// this point.  Given (from local_min_near) that d(b) < d(b-1) and not d(b+1) < d(b), is d strictly
// decreasing on all of [a, b], b - a >= 2?
//
// Write q(j) = |p - L_j|^2, D(j) = q(j+1) - q(j) = -2 (L_{j+1} - L_j).(p - M_j) with M_j the midpoint of
// segment j.  Then D(j+1) - D(j) = 2 [ (L_{j+2} - L_{j+1}).(M_{j+1} - M_j) - ((L_{j+2} - L_{j+1}) - (L_{j+1} - L_j)).(p - M_j) ]
//                                >= 2 [ g - h |p - M_j| ],
// g = min over the table of the first product, h = max over the table of the second-difference norm
// (both computed once per scenario on the host, rounded against us).  For j in [a, b] the point is within
// r = d(b) + (b - a + 1) s_max of M_j (s_max = longest segment).  The host hands over
// cert_rcap = min(1e3, (g - 1e-6) / h) shrunk by 1e-8: r <= cert_rcap gives D(j+1) - D(j) >= 2e-6 on the
// interval.  d(b) < d(b-1) as the reference computes it means D(b-1) <= 2^-49 q.  For j <= b - 2 this
// leaves D(j) <= 2^-49 q - 2e-6 with q <= r^2 <= 1e6, i.e. q(j+1) is below q(j) by more than 1e-6 absolute
// = 1e-12 relative — four orders of magnitude more than the rounding of the two squared distances and of
// their square roots (2^-50), so the reference's comparison hypot(j+1) < hypot(j) holds at every j in
// [a, b-2] without being evaluated.  A `false` only means "not proven this way" (kinked lane, far point, NaN).
__device__ inline bool convex_interior(const Cst& c, double q_b, int a, int b) {
    const double lim = c.k->cert_rcap - (double)(b - a + 1) * c.k->cert_smax;
    return (lim > 0.0) && (q_b <= lim * lim);
}

// With the comparisons the reference makes: does d strictly decrease on [a, b] (a <= b) and stop
// decreasing at b?  Eight candidates per trip are fetched together.
__device__ inline bool verify_interval(const Cst& c, const Lds& l, double px, double py, int a, int b) {
    double prev = lane_d2(c, l, px, py, a);
    int j = a;
    for (;;) {
        double v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = lane_d2(c, l, px, py, j + 1 + t);
        bool good = true, done = false;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int pos = j + 1 + t;
            const bool f = dist_less(v[t], (t == 0) ? prev : v[t - 1]);
            if (!done) {
                if (pos <= b) { if (!f) { good = false; done = true; } }
                else { good = !f; done = true; }   // pos == b + 1
            }
        }
        if (done) return good;
        j += 8;
        prev = v[7];
    }
}

// ---------------------------------------------------------------------------------------------
// one obstacle against one ego state: margins (cs:326-335, ut:344-361,395-407) and, if GRAD, their
// gradients w.r.t. the state (cs:715-739, ut:363-385,409-439).
struct ObsOut {
    double mf, mr;        // front / rear safety margin
    double gf[3], gr[3];  // gradient components (x, y, yaw); the v component is structurally 0
};

// What a row needs of an obstacle's record (x, y, sin yaw, cos yaw).  The rows fetch the record of obstacle o + 1 while they
// work on obstacle o, and the first one before the barrier terms of the row: a record is the same for every trial and
// every iteration of a solve and sits in L1 / L2, but its latency was paid once per obstacle and row.
struct ObsRec {
    double x, y, s, c;
};
__device__ inline void obs_fetch(ObsRec& r, gdouble* ob) {
    r.x = ob[0]; r.y = ob[1]; r.s = ob[3]; r.c = ob[4]; // (ob[3], ob[4]: dm_sincos(ob[2]) precomputed on the device at upload)
}
#ifndef CILQR_FAST_CONST_DIV
#define CILQR_FAST_CONST_DIV 0 /* measured: 18 % fewer vector instructions per trial cost, 0-1.6 % faster (r04_experiments): off */
#endif
// RN(x / c) for a constant c whose correctly rounded reciprocal rc = RN(1 / c) is at hand; see obstacle_terms
__device__ inline double div_by_const(double x, double c, double rc) {
    const double q0 = x * rc;
    const double e = __builtin_fma(-c, q0, x);
    return __builtin_fma(e, rc, q0);
}
template <bool GRAD>
__device__ inline void obstacle_terms(const Cst& c, const double xk[4], double sn_yaw, double cs_yaw,
                                      const ObsRec& ob, ObsOut& o) {
    double wv0 = c.wb * cs_yaw, wv1 = c.wb * sn_yaw;
    double fx, fy, rx, ry;
    if (c.rp == 0) {
        fx = xk[0] + wv0; fy = xk[1] + wv1; rx = xk[0]; ry = xk[1];
    } else {
        fx = xk[0] + 0.5 * wv0; fy = xk[1] + 0.5 * wv1;
        rx = xk[0] - 0.5 * wv0; ry = xk[1] - 0.5 * wv1;
    }
    const double so = ob.s, co = ob.c;
    double dfx = fx - ob.x, dfy = fy - ob.y;
    double drx = rx - ob.x, dry = ry - ob.y;
    double fX = co * dfx + so * dfy, fY = (-so) * dfx + co * dfy;
    double rX = co * drx + so * dry, rY = (-so) * drx + co * dry;
    if (!GRAD && CILQR_FAST_CONST_DIV &&
        DM_WAVE_ALL(__builtin_fabs(fX) < 1e150 && __builtin_fabs(fY) < 1e150 && __builtin_fabs(rX) < 1e150 && __builtin_fabs(rY) < 1e150)) {
        // The four quotients by the ellipse axes (ut:403-405: IEEE divisions upstream) through the correctly rounded
        // reciprocal: q0 = RN(x r), e = x - c q0 (exact in one fma), q = RN(q0 + e r) IS RN(x / c) for every x when
        // r = RN(1 / c) (Markstein 1990; q0 alone is off by one ulp on a quarter of the inputs, the corrected q on none of
        // 9.4e8 random and adversarial pairs incl. an all-ones significand: profiles/r04_experiments/) as long as nothing
        // overflows (the guard: squares below 1e300; a NaN fails it) or underflows — x below 2^-969, where the quotient is
        // below 2^-54 anyway and `1 - (qa + qb)` cannot see it.  3 vector instructions instead of ~15 per quotient; only in the
        // COST path (12 per row and trial cost) — the gradients divide signed small numbers and keep the division.
        const double a2 = c.k->ell_a2, b2 = c.k->ell_b2, ra = c.k->inv_a2, rb = c.k->inv_b2;
        o.mf = 1 - (div_by_const(fX * fX, a2, ra) + div_by_const(fY * fY, b2, rb));
        o.mr = 1 - (div_by_const(rX * rX, a2, ra) + div_by_const(rY * rY, b2, rb));
        return;
    }
    o.mf = 1 - ((fX * fX) / c.k->ell_a2 + (fY * fY) / c.k->ell_b2);
    o.mr = 1 - ((rX * rX) / c.k->ell_a2 + (rY * rY) / c.k->ell_b2);
    if (GRAD) {
        double f0 = -2 * fX / c.k->ell_a2, f1 = -2 * fY / c.k->ell_b2;
        double r0 = -2 * rX / c.k->ell_a2, r1 = -2 * rY / c.k->ell_b2;
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

__device__ inline gdouble* obs_at(const Cst& c, int j, int k) {
    return c.obs + ((size_t)j * c.T + (size_t)(c.tick + k)) * CILQR_OBS_STRIDE;
}

// Augmented-Lagrangian state of one trajectory (hpp:106-112): multipliers in HBM, [N][C], C = 8 + 2 M
struct AlmSt {
    double* mu;
    double* mu_next;
    double rho;
    int C;
};

// hpp:81-83 augmented_lagrangian_item
__device__ inline double alm_item(double cv, double rho, double mu) {
    double t = cv + mu / rho;
    double m = (t > 0.0) ? t : 0.0;
    return rho * (m * m) / 2;
}

// ---------------------------------------------------------------------------------------------
// Stage cost of row k (cs:199-287).  xk = x[k], uk = u[k] (k < N), ukm1 = u[k-1] (k >= 1).
// sd = k-th diagonal entry of (x-ref) W (x-ref)^T, ce = k-th of u R u^T, jb = J_barrier_k.
template <bool ALM>
__device__ inline void stage_cost(const Cst& c, const Lds& l, const AlmSt& al, int k, const double xk[4],
                                  const double uk[2], const double ukm1[2], int ridx, double& sd, double& ce,
                                  double& jb) {
    double rx, ry;
    lane_point(c, l, ridx, rx, ry);
    gdouble* aux = c.lane_aux + (size_t)ridx * CILQR_AUX_STRIDE;
    const double ryaw = aux[0], sr = aux[1], cr = aux[2];
    double e0 = xk[0] - rx, e1 = xk[1] - ry, e2 = xk[2] - c.k->ref_velo, e3 = xk[3] - ryaw;
    sd = CQ_MADD(e3 * c.k->w_yaw, e3, CQ_MADD(e2 * c.k->w_vel, e2, CQ_MADD(e1 * c.k->w_pos, e1, (e0 * c.k->w_pos) * e0)));
    ce = 0.0;
    if (k < c.N) ce = CQ_MADD(uk[1] * c.k->w_stl, uk[1], (uk[0] * c.k->w_acc) * uk[0]);
    jb = 0.0;
    if (k >= 1) {
        double acc_up = ukm1[0] - c.k->acc_max, acc_lo = c.k->acc_min - ukm1[0];
        double stl_up = ukm1[1] - c.k->stl_lim, stl_lo = -c.k->stl_lim - ukm1[1];
        double vel_up = xk[2] - c.k->velo_max, vel_lo = c.k->velo_min - xk[2];
        double d_sign = e1 * cr - e0 * sr;
        double hyp = dm_hypot(e0, e1);
        double cur_d = (d_sign < 0) ? -hyp : hyp;
        double pos_up = cur_d - c.k->pos_up_b, pos_lo = c.k->pos_lo_b - cur_d;
        const double* mu = ALM ? (al.mu + (size_t)(k - 1) * al.C) : nullptr;
        ObsRec rec = {0.0, 0.0, 0.0, 0.0};
        gdouble* po = obs_at(c, 0, k);
        const size_t po_step = (size_t)c.T * CILQR_OBS_STRIDE;
        if (c.M > 0) obs_fetch(rec, po);
        double j;
        if (ALM) {
            j = alm_item(acc_up, al.rho, mu[0]) + alm_item(acc_lo, al.rho, mu[1]);
            j = j + alm_item(stl_up, al.rho, mu[2]);
            j = j + alm_item(stl_lo, al.rho, mu[3]);
            j = j + alm_item(vel_up, al.rho, mu[4]);
            j = j + alm_item(vel_lo, al.rho, mu[5]);
            j = j + alm_item(pos_up, al.rho, mu[6]);
            j = j + alm_item(pos_lo, al.rho, mu[7]);
        } else {
            j = c.k->sq1 * dm_exp(c.k->sq2 * acc_up) + c.k->sq1 * dm_exp(c.k->sq2 * acc_lo);
            j = j + c.k->sq1 * dm_exp(c.k->sq2 * stl_up);
            j = j + c.k->sq1 * dm_exp(c.k->sq2 * stl_lo);
            j = j + c.k->sq1 * dm_exp(c.k->sq2 * vel_up);
            j = j + c.k->sq1 * dm_exp(c.k->sq2 * vel_lo);
            j = j + c.k->sq1 * dm_exp(c.k->sq2 * pos_up);
            j = j + c.k->sq1 * dm_exp(c.k->sq2 * pos_lo);
        }
        double sy, cy;
        dm_sincos(xk[3], &sy, &cy);
        for (int o = 0; o < c.M; ++o) {
            ObsRec nxt = rec;
            po += po_step;
            if (o + 1 < c.M) obs_fetch(nxt, po);
            ObsOut t;
            obstacle_terms<false>(c, xk, sy, cy, rec, t);
            if (ALM) {
                j = j + alm_item(t.mf, al.rho, mu[8 + 2 * o]);
                j = j + alm_item(t.mr, al.rho, mu[9 + 2 * o]);
            } else {
                j = j + c.k->oq1 * dm_exp(c.k->oq2 * t.mf);
                j = j + c.k->oq1 * dm_exp(c.k->oq2 * t.mr);
            }
            rec = nxt;
        }
        jb = j;
    }
}

// J = (sum_k sd + sum_k ce) + sum_k jb, each sum sequential in k as Eigen's trace()/the loop at
// cs:217 accumulate: three lanes per trial slot each run one of the three chains over its own row
// of l.cs (one instruction stream), then every lane combines them.  NTR trial slots at once: lanes
// 3*tt + {0,1,2} take slot tt; J[tt] for every slot on every lane.
template <int NTR>
__device__ inline void sum_stage_costs_multi(const Lds& l, int N, int lane, double J[NTR], int slot0 = 0) {
    const int R = N + 1;
    const int slot = (lane < 3 * NTR) ? lane / 3 : 0;
    const int row = (lane < 3 * NTR) ? lane % 3 : 0;
    const double* v = l.cs + (size_t)(slot0 + slot) * 3 * R + row * R;
    const int last = (row == 1) ? N - 1 : N;
    double acc = (row == 2) ? 0.0 : v[0];
    int k = 1;
    if (k + 7 <= last) {
        // the chain of additions is the critical path: the next eight terms are fetched while eight are added
        double a[8], b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = v[k + t];
        for (; k + 15 <= last; k += 8) {
#pragma unroll
            for (int t = 0; t < 8; ++t) b[t] = v[k + 8 + t];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = acc + a[t];
#pragma unroll
            for (int t = 0; t < 8; ++t) a[t] = b[t];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = acc + a[t];
        k += 8;
    }
    for (; k <= last; ++k) acc = acc + v[k];
#pragma unroll
    for (int tt = 0; tt < NTR; ++tt) {
        double sd = __shfl(acc, 3 * tt, CILQR_WAVE);
        double ce = __shfl(acc, 3 * tt + 1, CILQR_WAVE);
        double jb = __shfl(acc, 3 * tt + 2, CILQR_WAVE);
        J[tt] = (sd + ce) + jb;
    }
}

__device__ inline double sum_stage_costs(const Lds& l, int N, int lane) {
    double J[1];
    sum_stage_costs_multi<1>(l, N, lane, J);
    return J[0];
}

// get_total_cost of the trajectory held in LDS (x, u, ridx)
template <bool ALM>
__device__ inline double total_cost_lds(const Cst& c, const Lds& l, const AlmSt& al, int lane) {
    const int N = c.N;
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        double xk[4] = {l.x[4 * k], l.x[4 * k + 1], l.x[4 * k + 2], l.x[4 * k + 3]};
        double uk[2] = {0, 0}, um[2] = {0, 0};
        if (k < N) { uk[0] = l.u[2 * k]; uk[1] = l.u[2 * k + 1]; }
        if (k >= 1) { um[0] = l.u[2 * k - 2]; um[1] = l.u[2 * k - 1]; }
        double sd, ce, jb;
        stage_cost<ALM>(c, l, al, k, xk, uk, um, l.ridx[k], sd, ce, jb);
        l.cs[k] = sd;
        l.cs[(N + 1) + k] = ce;
        l.cs[2 * (N + 1) + k] = jb;
    }
    wave_sync();
    double J = sum_stage_costs(l, N, lane);
    wave_sync();
    return J;
}

// get_total_cost of trial trajectory `a` held in the scratch slab.  Also leaves the trial's lane
// indices in l.tidx (accept_trial copies them).
//
// Reference points (cs:289-314), lane = row.  The reference chains the rows:
// idx[k] = first j >= idx[k-1] with !(d_k(j+1) < d_k(j)).  Here every row first finds a candidate
// m[k] near the index the CURRENT trajectory has on that row, then row k checks — with the same
// comparisons the reference makes — that d_k strictly decreases on [m[k-1], m[k]] and stops
// decreasing at m[k].  If that holds for every row, idx == m by induction from idx[0] = idx0;
// otherwise (non-monotone candidates, an earlier minimum missed, ...) the serial chain runs.
// NCH = rows per lane: 1 when N + 1 <= 64, else 2 (N + 1 <= 128; cilqr_set_params caps N at 127)
// NTR = trials costed in this pass (a0, a0+1, ...; slots beyond `nt` are skipped): their loads and
// LDS round trips overlap, so a pass of two costs much less than two passes of one.
template <bool DBG, int NCH, bool ALM, int NTR>
__device__ inline void total_cost_trials(const Cst& c, const Lds& l, const AlmSt& al, const double* scr, int a0,
                                         int nt, int lane, int idx0, int flags_in, int* n_fallback, double J[NTR],
                                         long long* sub = nullptr, int slot0 = 0, int as = CILQR_MAX_ALPHA_TRIALS) {
    // scr/as: where the trials live — the slab (as = 20, trial a0 + tt at scr + a0 + tt) or, for the first
    // trial of a shallow iteration, the first-trial buffer (as = 1, a0 = 0)
    const int flags = DBG ? flags_in : 0;
    const int N = c.N;
    const int R = N + 1;
    long long t0 = sub ? (long long)__builtin_readcyclecounter() : 0;
    // this lane's rows of the trials, fetched once
    double xk[NTR][NCH][4], uk[NTR][NCH][2], um[NTR][NCH][2];
    // Where each row starts looking: at the index the trial costed LAST in this slot has on that row (the rows of
    // l.tidx are kept from trial to trial; the solve seeds them with the current trajectory's indices).  Inside a
    // search that is the trial one or two step sizes up — as close as the current trajectory, which the seed and every
    // accepted trial make it equal to.  It matters where the chain of cs:289-314 is unstable: a trajectory whose rows
    // see two local minima of the distance (a lane that swerves) can sit on one branch while every trial, however
    // small its step, chains along the other, 200 samples away; guessing from the current trajectory sends each of
    // those trials down the serial chain (config 4: 304 trials of one solve, 45 of the launch's 52 ms), guessing from
    // the previous trial only the first of a run.
    int guess[NTR][NCH];
#pragma unroll
    for (int tt = 0; tt < NTR; ++tt) {
        const int* tix = l.tidx + (slot0 + tt) * (N + 2);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int k = lane + CILQR_WAVE * ch;
            guess[tt][ch] = (k <= N) ? tix[k] : idx0;
        }
    }
#pragma unroll
    for (int tt = 0; tt < NTR; ++tt) {
        const double* t = TRIAL_AT(scr, a0 + tt);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int k = lane + CILQR_WAVE * ch;
            xk[tt][ch][0] = xk[tt][ch][1] = xk[tt][ch][2] = xk[tt][ch][3] = 0.0;
            uk[tt][ch][0] = uk[tt][ch][1] = um[tt][ch][0] = um[tt][ch][1] = 0.0;
            if (tt < nt && k <= N) {
                xk[tt][ch][0] = TRS(t, 0, k, as); xk[tt][ch][1] = TRS(t, 1, k, as);
                xk[tt][ch][2] = TRS(t, 2, k, as); xk[tt][ch][3] = TRS(t, 3, k, as);
                if (k < N) { uk[tt][ch][0] = TRS(t, 4, k, as); uk[tt][ch][1] = TRS(t, 5, k, as); }
                if (k >= 1) { um[tt][ch][0] = TRS(t, 4, k - 1, as); um[tt][ch][1] = TRS(t, 5, k - 1, as); }
            }
        }
    }
    bool proven[NTR];
#pragma unroll
    for (int tt = 0; tt < NTR; ++tt) proven[tt] = (tt >= nt);
    if (!(flags & CILQR_DBG_SERIAL_REF_SCAN)) {
        // level 1: every row looks for its candidate — a local minimum of its distance profile — near its guess
        // (unchanged for the small steps of a failing line search: three samples and done) ...
        double qm[NTR][NCH];
        int cand[NTR][NCH];
#pragma unroll
        for (int tt = 0; tt < NTR; ++tt) {
            if (tt >= nt) continue;
            int* tix = l.tidx + (slot0 + tt) * (N + 2);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int k = lane + CILQR_WAVE * ch;
                qm[tt][ch] = 0.0;
                cand[tt][ch] = 0;
                if (k <= N) {
                    int m = idx0;
                    if (k > 0) {
                        int g = guess[tt][ch];
                        g = (g < idx0) ? idx0 : g;
                        g = (g > c.L - 1) ? c.L - 1 : g;
                        m = local_min_near(c, l, xk[tt][ch][0], xk[tt][ch][1], g, idx0, &qm[tt][ch]);
                    }
                    cand[tt][ch] = m;
                    tix[k] = m;
                }
            }
        }
        wave_sync();
        // ... then the proof that the chain of cs:289-314 visits exactly these.  Row k's scan starts at idx[k-1] and
        // only moves forward.  Where the candidates ascend, idx[k] = m[k] iff the distance decreases strictly from
        // m[k-1] up to m[k] and stops there: the two comparisons at m[k] come with the candidate, the interior is
        // covered by the convexity certificate, else sample by sample.  Where a candidate lies BEHIND the chain — the
        // vehicle slows down, backs up or swerves and its nearest sample falls behind the one an earlier row has
        // reached — the chain stays where it is, provided the distance does not decrease there: idx = the running
        // maximum p of the candidates, and row k with m[k] <= p[k-1] has to show not d(p + 1) < d(p) at p = p[k-1]
        // (one comparison; it came with the candidate when m[k] = p).  By induction from idx[0] = idx0.
        // (config 4 has solves in which a score of rows stay behind on every trial trajectory: without the running
        //  maximum each of their 300 trial costs went down the serial chain, through global memory: 45 ms of a 52 ms launch)
#pragma unroll
        for (int tt = 0; tt < NTR; ++tt) {
            if (tt >= nt) continue;
            int* tix = l.tidx + (slot0 + tt) * (N + 2);
            int lo_[NCH], hi_[NCH];
            bool mono = true;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int k = lane + CILQR_WAVE * ch;
                lo_[ch] = idx0;
                hi_[ch] = cand[tt][ch];
                if (k >= 1 && k <= N) {
                    lo_[ch] = tix[k - 1];
                    mono = mono && (lo_[ch] <= hi_[ch]);
                }
            }
            if (__ballot(!mono) != 0ULL) {
                // some candidate lies behind: running maximum over the rows (lane order, chunk after chunk)
                wave_sync();
                int carry = idx0;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int k = lane + CILQR_WAVE * ch;
                    int v = (k <= N) ? cand[tt][ch] : 0;
#pragma unroll
                    for (int d = 1; d < CILQR_WAVE; d <<= 1) {
                        const int o = __shfl_up(v, d, CILQR_WAVE);
                        v = (lane >= d && o > v) ? o : v;
                    }
                    v = (carry > v) ? carry : v;
                    carry = __shfl(v, CILQR_WAVE - 1, CILQR_WAVE);
                    hi_[ch] = v;
                    if (k <= N) tix[k] = v;
                }
                wave_sync();
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int k = lane + CILQR_WAVE * ch;
                    if (k >= 1 && k <= N) lo_[ch] = tix[k - 1];
                }
            }
            bool ok = true, sampled = false;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int k = lane + CILQR_WAVE * ch;
                if (k >= 1 && k <= N) {
                    const int lo = lo_[ch], hi = hi_[ch];
                    bool good = true;
                    if (hi > lo) { // (then hi is this row's candidate)
                        if (hi - lo >= 2 && !convex_interior(c, qm[tt][ch], lo, hi)) {
                            sampled = true;
                            if (!verify_window_fast(l, xk[tt][ch][0], xk[tt][ch][1], lo, hi))
                                good = verify_interval(c, l, xk[tt][ch][0], xk[tt][ch][1], lo, hi);
                        }
                    } else if (cand[tt][ch] != lo) { // the chain stays at lo, ahead of this row's candidate
                        good = !dist_less(lane_d2(c, l, xk[tt][ch][0], xk[tt][ch][1], lo + 1),
                                          lane_d2(c, l, xk[tt][ch][0], xk[tt][ch][1], lo));
                    }
                    ok = ok && good;
                }
            }
            proven[tt] = (__ballot(!ok) == 0ULL);
            const bool any_sampled = (__ballot(sampled) != 0ULL);
            if (sub && lane == 0 && any_sampled) sub[3] += 1;
        }
    }
    // level 2: the serial chain of cs:289-314
#pragma unroll
    for (int tt = 0; tt < NTR; ++tt) {
        if (proven[tt]) continue;
        wave_sync();
        *n_fallback += 1;
        const double* t = TRIAL_AT(scr, a0 + tt);
        int* tix = l.tidx + (slot0 + tt) * (N + 2);
        int s = idx0;
        if (lane == 0) tix[0] = s;
        for (int i = 1; i <= N; ++i) {
            s = ref_scan_from(c, l, TRS(t, 0, i, as), TRS(t, 1, i, as), s);
            if (lane == 0) tix[i] = s;
        }
    }
    wave_sync();
    if (sub) { long long t1 = (long long)__builtin_readcyclecounter(); if (lane == 0) sub[0] += t1 - t0; t0 = t1; }
#pragma unroll
    for (int tt = 0; tt < NTR; ++tt) {
        if (tt >= nt) continue;
        const int* tix = l.tidx + (slot0 + tt) * (N + 2);
        double* cs = l.cs + (size_t)(slot0 + tt) * 3 * R;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int k = lane + CILQR_WAVE * ch;
            if (k <= N) {
                double sd, ce, jb;
                stage_cost<ALM>(c, l, al, k, xk[tt][ch], uk[tt][ch], um[tt][ch], tix[k], sd, ce, jb);
                cs[k] = sd;
                cs[R + k] = ce;
                cs[2 * R + k] = jb;
            }
        }
    }
    wave_sync();
    if (sub) { long long t1 = (long long)__builtin_readcyclecounter(); if (lane == 0) sub[1] += t1 - t0; t0 = t1; }
    sum_stage_costs_multi<NTR>(l, N, lane, J, slot0);
    wave_sync();
    if (sub) { long long t1 = (long long)__builtin_readcyclecounter(); if (lane == 0) sub[2] += t1 - t0; }
}

// single-trial form (piecewise kernel)
template <bool DBG, int NCH, bool ALM>
__device__ inline double total_cost_trial(const Cst& c, const Lds& l, const AlmSt& al, const double* scr, int a,
                                          int lane, int idx0, int flags_in, int* n_fallback,
                                          long long* sub = nullptr) {
    double J[1];
    total_cost_trials<DBG, NCH, ALM, 1>(c, l, al, scr, a, 1, lane, idx0, flags_in, n_fallback, J, sub);
    return J[0];
}

// ---------------------------------------------------------------------------------------------
// Reference points of the trajectory held in LDS x, rows in parallel: candidate per row from a guess
// (distance travelled from row 0 over the lane's sample spacing at idx0 — only speed depends on it), then
// the same proof as for the trial trajectories (see total_cost_trials).  false = not proven, l.ridx undefined.
__device__ inline bool ref_indices_parallel(const Cst& c, const Lds& l, int lane, int idx0) {
    const int N = c.N;
    double ds = 1.0;
    if (idx0 + 1 < c.L) {
        double ax, ay, bx, by;
        lane_point(c, l, idx0, ax, ay);
        lane_point(c, l, idx0 + 1, bx, by);
        ds = dm_hypot(bx - ax, by - ay);
    }
    const double x0 = l.x[0], y0 = l.x[1];
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        int m = idx0;
        double q = 0.0;
        if (k > 0) {
            const double px = l.x[4 * k], py = l.x[4 * k + 1];
            const double gd = dm_hypot(px - x0, py - y0) / ds;
            int g = idx0 + ((gd < 1.0e6) ? (int)gd : 0); // NaN / inf / absurd: start at idx0
            g = (g < idx0) ? idx0 : g;
            g = (g > c.L - 1) ? c.L - 1 : g;
            m = local_min_near(c, l, px, py, g, idx0, &q);
        }
        l.ridx[k] = m;
        l.cs[k] = q; // the stage-cost scratch is free here
    }
    wave_sync();
    bool ok = true;
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        if (k >= 1) {
            const int lo = l.ridx[k - 1], hi = l.ridx[k];
            bool good = (lo <= hi);
            if (good && hi - lo >= 2 && !convex_interior(c, l.cs[k], lo, hi)) {
                const double px = l.x[4 * k], py = l.x[4 * k + 1];
                if (!verify_window_fast(l, px, py, lo, hi)) good = verify_interval(c, l, px, py, lo, hi);
            }
            ok = ok && good;
        }
    }
    const bool proven = (__ballot(!ok) == 0ULL);
    wave_sync();
    return proven;
}

// Initial trajectory (cs:155-197): cold start u = 0, or warm start from last_u shifted by one step;
// fills LDS x, u, ridx.  Wave-uniform serial rollout.
__device__ inline void init_trajectory(const Cst& c, Lds& l, const double x0[4], const double* last_u,
                                       int lane, int& idx0, int Wcap) {
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
    wave_sync();
    idx0 = ref_scan_row0(c, x0[0], x0[1], lane);
    stage_window(c, l, idx0, Wcap, lane);
    // the rollout is one serial chain (wave-uniform); the reference points do not feed back into it, so they
    // are found afterwards for all rows at once
    double xc[4] = {x0[0], x0[1], x0[2], x0[3]};
    if (lane == 0) {
        l.x[0] = xc[0]; l.x[1] = xc[1]; l.x[2] = xc[2]; l.x[3] = xc[3];
    }
    for (int i = 0; i < N; ++i) {
        double ui[2] = {l.u[2 * i], l.u[2 * i + 1]};
        double xn[4];
        if (c.rp == 0) propagate<0>(c, xc, ui, xn);
        else propagate<1>(c, xc, ui, xn);
        if (lane == 0) {
            l.x[4 * (i + 1)] = xn[0]; l.x[4 * (i + 1) + 1] = xn[1];
            l.x[4 * (i + 1) + 2] = xn[2]; l.x[4 * (i + 1) + 3] = xn[3];
        }
        xc[0] = xn[0]; xc[1] = xn[1]; xc[2] = xn[2]; xc[3] = xn[3];
    }
    wave_sync();
    if (!ref_indices_parallel(c, l, lane, idx0)) {
        // the serial chain of cs:289-314
        int s = idx0;
        if (lane == 0) l.ridx[0] = s;
        for (int i = 1; i <= N; ++i) {
            s = ref_scan_from(c, l, l.x[4 * i], l.x[4 * i + 1], s);
            if (lane == 0) l.ridx[i] = s;
        }
        wave_sync();
    }
}

// ridx for a trajectory already staged in LDS x (used by the piecewise kernels)
__device__ inline void ref_indices_lds(const Cst& c, Lds& l, int lane, int& idx0, int Wcap) {
    const int N = c.N;
    idx0 = ref_scan_row0(c, l.x[0], l.x[1], lane);
    stage_window(c, l, idx0, Wcap, lane);
    int s = idx0;
    if (lane == 0) l.ridx[0] = s;
    for (int i = 1; i <= N; ++i) {
        s = ref_scan_from(c, l, l.x[4 * i], l.x[4 * i + 1], s);
        if (lane == 0) l.ridx[i] = s;
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// forward_pass (cs:442-461) for all trial step sizes at once: lane a < n_alpha uses alpha = 2^-a.
// The trial trajectories go to the scratch slab; their reference points are found later, only for
// the trials whose cost is actually needed.
// `scr`, `as`: destination — the slab (as = 20: lane a writes trial a) or the first-trial buffer (as = 1,
// n_alpha = 1: only alpha = 1 is rolled out).  A trial trajectory has the same bits whichever pass produced it.
//
// One step of the closed-loop rollout: the nominal point and the gains of the step arrive in `g` (fetched from LDS
// one step ahead, see RollIn), xc is the trial state on entry and the next state on exit.  SMALL = the straight-line
// small-angle step: returns false — nothing stored, xc untouched — when some active lane does not qualify.
struct RollIn {
    double k[CILQR_KD], x[4], u[2];
};
__device__ inline void roll_fetch(RollIn& g, const Lds& l, int i) {
#pragma unroll
    for (int e = 0; e < CILQR_KD; ++e) g.k[e] = l.kd[CILQR_KD * i + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) g.x[e] = l.x[4 * i + e];
    g.u[0] = l.u[2 * i];
    g.u[1] = l.u[2 * i + 1];
}
typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
// Where the rollout lanes store: a buffer descriptor of the destination (four scalar registers), wave-uniform byte
// offsets of the current rows (scalar registers, advanced by the scalar unit) and this lane's byte offset — one
// buffer_store per value, no address arithmetic in vector registers.
struct RollOut {
    __amdgpu_buffer_rsrc_t rsrc;
    int pairb, tileb; // pair / row-tile stride in bytes
    unsigned lane_off;
};
// A wave-uniform step index as the scalar unit holds it, its value hidden from the optimiser.
// Why (profiles/r04_experiments/tiled_slab_lost_rows.txt has the whole story): the first tiled build of the grouped rollout
// pass took its horizon as an argument — out of line, so in a VECTOR register — and the compiler, counting loop exits, step
// index and buffer descriptor as divergent, wrapped every slab store in a loop over the lanes' "distinct" descriptors
// (v_readfirstlane, compare, s_and_saveexec, store, s_xor exec, s_cbranch_execnz).  On gfx950 with XNACK off a 16-byte store
// in that shape delivered, for lanes 12-15, the content its first data register received six instructions LATER (the next
// store's offset: that is what landed in u0's low half; found with a shadow copy written by a second pass and compared lane by
// lane; wait states or s_waitcnt vmcnt(0) in between change nothing, HSA_XNACK=1 makes it vanish — the mechanism is open).  Costs off in the ninth digit, depending on which two
// trajectories shared a wavefront.  The cause removed: every descriptor is built from scalars (rollout_group and
// rollout_trials_rp take the horizon through v_readfirstlane; tests/test_cabi.py scans the shipped disassembly for such loops).
// This opaque copy changes nothing about that — it was one of the first day's "cures", by way of a different register
// allocation — and stays in the grouped pass because it is free there and keeps the offset's arithmetic on the scalar unit.
__device__ inline int opaque_uniform(int k) {
    int ku = __builtin_amdgcn_readfirstlane(k);
    __asm__ volatile("" : "+s"(ku));
    return ku;
}
// byte offset of row k of pair 0 for step size 0 (scalar arithmetic)
// (k_solve's step index is a clean scalar register — its loops have scalar bounds — and its tiled stores were right from the
//  first build on, checked against the oracle and, launch by launch, against both cured forms of the grouped pass; the opaque
//  copy costs configs[3]'s kernel 1.6 % here, so it stays with the grouped pass: slab_st2_row)
__device__ inline int slab_row_off(const RollOut& o, int k) {
    return (k / CILQR_SLAB_TILE) * o.tileb + (k % CILQR_SLAB_TILE) * 16;
}
template <int RP, bool SMALL, int PIN = DM_PIN>
__device__ inline bool roll_step(const Cst& c, const DmPinned& pk, const RollIn& g, double alpha, double xc[4], const RollOut& o, int i) {
    const double dx0 = xc[0] - g.x[0], dx1 = xc[1] - g.x[1], dx2 = xc[2] - g.x[2], dx3 = xc[3] - g.x[3];
    const double k0 = CQ_MADD(g.k[3], dx3, CQ_MADD(g.k[2], dx2, CQ_MADD(g.k[1], dx1, g.k[0] * dx0)));
    const double k1 = CQ_MADD(g.k[8], dx3, CQ_MADD(g.k[7], dx2, CQ_MADD(g.k[6], dx1, g.k[5] * dx0)));
    double un[2];
    un[0] = CQ_MADD(alpha, g.k[CILQR_KD_D(0)], g.u[0] + k0);
    un[1] = CQ_MADD(alpha, g.k[CILQR_KD_D(1)], g.u[1] + k1);
    double xn[4];
    if (SMALL) {
        if (!DM_WAVE_ALL(__builtin_fabs(xc[3]) < 0.785 && __builtin_fabs(un[1]) < 0.7)) return false;
        if (!propagate_small<RP, PIN>(c, xc, un, xn, &pk)) return false;
    } else {
        propagate<RP, PIN | DM_NOSHORT>(c, xc, un, xn, &pk);
    }
#define CILQR_SLAB_ST2(base, pair, v0, v1)                                                                           \
    {                                                                                                                \
        const u32x2 lo_ = __builtin_bit_cast(u32x2, (v0)), hi_ = __builtin_bit_cast(u32x2, (v1));                    \
        u32x4 q_;                                                                                                    \
        q_.x = lo_.x; q_.y = lo_.y; q_.z = hi_.x; q_.w = hi_.y;                                                      \
        __builtin_amdgcn_raw_buffer_store_b128(q_, o.rsrc, o.lane_off, (base) + (pair) * o.pairb, 0);                \
    }
    const int ou = slab_row_off(o, i), ox = slab_row_off(o, i + 1);
    CILQR_SLAB_ST2(ou, 2, un[0], un[1]);
    CILQR_SLAB_ST2(ox, 0, xn[0], xn[1]);
    CILQR_SLAB_ST2(ox, 1, xn[2], xn[3]);
#undef CILQR_SLAB_ST2
    xc[0] = xn[0]; xc[1] = xn[1]; xc[2] = xn[2]; xc[3] = xn[3];
    return true;
}

// a wave-uniform value the compiler cannot see to be uniform (it descends from vector comparisons), moved into
// scalar registers so that addresses built from it stay scalar
__device__ inline int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline gdouble_w* uniform_ptr(double* p) {
    const unsigned long long u = (unsigned long long)(size_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u & 0xffffffffULL));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return (gdouble_w*)(size_t)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}

template <int RP, int PIN = DM_PIN>
__device__ inline void rollout_trials_rp(const Cst& c, const Lds& l, double* scr_in, int lane, int n_alpha, int as_in) {
    // (scalar whatever the compiler thinks of where it came from — in the closed-loop builds the horizon descends from a table
    //  lookup by an index the loop carries, counted as divergent, the buffer descriptor built from it too, and every slab
    //  store became a loop over the lanes' "distinct" descriptors: the shape that lost store data in round 4, see opaque_uniform)
    const int N = __builtin_constant_p(c.N) ? c.N : uniform_int(c.N); // (the compile-time horizons stay compile-time)
    const int R = N + 1;
    const int as = uniform_int(as_in);
    gdouble_w* scr = uniform_ptr(scr_in);
    double* scr_in_uniform = (double*)(size_t)scr; // the same address as a generic pointer, for the descriptor
    if (lane < n_alpha) {
        const double alpha = dm_pow2i(-lane);
        gdouble_w* t = TRIAL_AT(scr, lane);
        double xc[4] = {l.x[0], l.x[1], l.x[2], l.x[3]};
        TRS(t, 0, 0, as) = xc[0]; TRS(t, 1, 0, as) = xc[1]; TRS(t, 2, 0, as) = xc[2]; TRS(t, 3, 0, as) = xc[3];
        RollOut o;
        o.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)scr_in_uniform, 0, (int)(first_trial_doubles(N) * as * sizeof(double)), 0x00020000);
        o.lane_off = (unsigned)(2 * CILQR_SLAB_TILE * sizeof(double)) * (unsigned)lane;
        o.tileb = as * 2 * CILQR_SLAB_TILE * (int)sizeof(double);
        o.pairb = CILQR_SLAB_RT(R) * o.tileb;
        // Two loops over the steps.  The first assumes small angles on all trial lanes (the usual case:
        // yaw relative to the x axis and steering below pi/4) and runs the straight-line step; the moment
        // a step does not qualify it hands over — nothing of that step has been stored yet — to the second,
        // general loop, which finishes the horizon.
        // In both, the gains and the nominal point of step i + 1 are fetched from LDS while step i computes;
        // the loops are unrolled by two over two register sets, so nothing is copied between steps.  The
        // polynomial coefficients of sin / cos / tan sit in vector registers for the whole pass (dm_pin_load).
        DmPinned pk;
        if (PIN) dm_pin_load(pk);
        int i = 0;
        {
            RollIn ga, gb;
            roll_fetch(ga, l, 0);
            for (;;) {
                if (i >= N) break;
                if (i + 1 < N) roll_fetch(gb, l, i + 1);
                if (!roll_step<RP, true, PIN>(c, pk, ga, alpha, xc, o, i)) break;
                ++i;
                if (i >= N) break;
                if (i + 1 < N) roll_fetch(ga, l, i + 1);
                if (!roll_step<RP, true, PIN>(c, pk, gb, alpha, xc, o, i)) break;
                ++i;
            }
        }
        if (i < N) {
            RollIn ga, gb;
            roll_fetch(ga, l, i);
            for (;;) {
                if (i + 1 < N) roll_fetch(gb, l, i + 1);
                roll_step<RP, false, PIN>(c, pk, ga, alpha, xc, o, i);
                ++i;
                if (i >= N) break;
                if (i + 1 < N) roll_fetch(ga, l, i + 1);
                roll_step<RP, false, PIN>(c, pk, gb, alpha, xc, o, i);
                ++i;
                if (i >= N) break;
            }
        }
    }
    wave_sync();
}

// PIN = DM_PIN: the polynomial coefficients of sin / cos and of the range reduction sit in vector registers for the whole
// pass (36 VGPRs; what the one-wavefront-per-SIMD builds and the two-rows-per-lane builds do best with); 0: they are
// scalar-register literals rebuilt where they are used (the scalar unit's issue slots) — the lone wavefronts that share a
// SIMD in the large-batch builds are faster that way, and spill 12 vector registers instead of 74.
template <int PIN = DM_PIN>
__device__ inline void rollout_trials(const Cst& c, const Lds& l, double* scr, int lane, int n_alpha,
                                      int as = CILQR_MAX_ALPHA_TRIALS) {
    // one loop per vehicle model: only that model's polynomial constants are live inside it
    if (c.rp == 0) rollout_trials_rp<0, PIN>(c, l, scr, lane, n_alpha, as);
    else rollout_trials_rp<1, PIN>(c, l, scr, lane, n_alpha, as);
}

// the rows of l.tidx start out as the current trajectory's indices (see the guesses of total_cost_trials)
__device__ inline void seed_trial_indices(const Lds& l, int N, int slots, int lane) {
    for (int s = 0; s < slots; ++s)
        for (int k = lane; k <= N; k += CILQR_WAVE) l.tidx[s * (N + 2) + k] = l.ridx[k];
    wave_sync();
}

// copy trial `a` (costed in slot `slot` of the last pass, whose index row is still in l.tidx) into the current trajectory
__device__ inline void accept_trial(const Cst& c, const Lds& l, const double* scr, int a, int slot, int lane,
                                    int as = CILQR_MAX_ALPHA_TRIALS) {
    const int N = c.N;
    const int R = N + 1;
    const double* t = TRIAL_AT(scr, a);
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        l.x[4 * k] = TRS(t, 0, k, as);
        l.x[4 * k + 1] = TRS(t, 1, k, as);
        l.x[4 * k + 2] = TRS(t, 2, k, as);
        l.x[4 * k + 3] = TRS(t, 3, k, as);
        l.ridx[k] = l.tidx[slot * (N + 2) + k];
        if (k < N) {
            l.u[2 * k] = TRS(t, 4, k, as);
            l.u[2 * k + 1] = TRS(t, 5, k, as);
        }
    }
    wave_sync();
}

// The head of l.kd that the lone costing of a first trial overwrites (slot 0 of the stage-cost scratch), parked
// behind the first-trial buffer and brought back if the search goes on to a second rollout pass.
__device__ inline void save_gains_head(const Lds& l, double* first, int N, int lane) {
    double* park = first + first_trial_doubles(N);
    for (int e = lane; e < 3 * (N + 1); e += CILQR_WAVE) park[e] = l.kd[e];
}
__device__ inline void restore_gains_head(const Lds& l, const double* first, int N, int lane) {
    const double* park = first + first_trial_doubles(N);
    for (int e = lane; e < 3 * (N + 1); e += CILQR_WAVE) l.kd[e] = park[e];
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// Work sharing between blocks (k_solve's SHARE).  A launch ends with a few long line searches on a mostly idle
// chip: the last trajectories each cost their 20 trial trajectories one after the other.  Blocks that find no
// trajectory left to pull (the large-batch builds run persistent blocks, k_solve) therefore stay and cost open trials
// of the blocks still running.  A cost is a pure function of the trial
// trajectory (in the global slab), the trajectory's tables and the row-0 lane index, so whoever computes it
// computes the same bits; the search's verdicts are still taken in order by the owner.
//
// Per trajectory one ShareReq; a search in need of help is announced in one of 64 slots that idle blocks poll.
// claim = seq << 16 | next: trials are handed out in ascending order — the order the verdicts need them in — by
// compare-and-swap on `next`, to helpers and to the owner alike (the owner keeps the first two of the search without
// asking); seq numbers the searches of the trajectory, a closed request has next = 255.  J[t] holds the bits of the
// cost of trial t or the pending mark.  The owner does not touch the slab again before every claimed trial has been
// delivered (sh_owner_close), so no helper ever reads a slab that is being rewritten or writes into a later search.
// Visibility across the chip's eight L2s: the owner's release (L2 write-back) before it opens the claim word, the
// helper's acquire (L2 invalidate) after it has claimed; claim words, slots, counters and results are agent-scope
// atomics.  Every wait is bounded: past the bound the owner costs the trial itself and the launch is flagged
// (SH_ERROR), it never hangs.
struct ShareReq {
    unsigned claim;
    int idx0;       // row-0 lane index of the trajectory (= first sample of its lane window)
    int slot;       // which scratch area holds its slab (the owner block's)
    int pad1;
    unsigned long long J[CILQR_MAX_ALPHA_TRIALS];
    unsigned long long rho_bits; // augmented Lagrangian: the owner's penalty weight (its multipliers are in global memory)
    unsigned long long pad2;
};
static_assert(sizeof(ShareReq) == 192, "ShareReq layout");
// (the words that many blocks hammer sit 64 bytes apart: one L2 line serves ~90 atomics per microsecond, and the
//  owners' once-per-iteration look at SH_HELPING must not queue behind two thousand idle blocks polling the queue)
enum { SH_NEXT = 0 /* persistent blocks: the next trajectory */, SH_FINISHED = 16, SH_HELPING = 32 /* blocks helping right now */,
       SH_Q_RESV = 48, SH_Q_HEAD = 64 /* queue of parked trajectories (resumable solves) */,
       SH_HELPERS = 80 /* blocks that went helping (count) */, SH_ERROR = 81, SH_ANNOUNCED = 82, SH_HELPED = 83, SH_PARKED = 84,
       SH_TEST_SPINS = 85 /* development library: bound of the grouped build's hand-over wait when non-zero (CILQR_TUNE=grp_wait_spins) */,
       SH_SLOT0 = 96 };
#define CILQR_SH_NSLOT 64
#define CILQR_SH_WORDS (SH_SLOT0 + CILQR_SH_NSLOT)
#define CILQR_SH_PENDING 0x7ff8c11a5ea7ed00ULL /* a NaN no arithmetic produces */
#define CILQR_SH_SPIN_CAP (1 << 22)
#define CILQR_SH_MIN_OPEN 4 /* trials still open for a search to be announced */

__device__ inline unsigned sh_ld(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void sh_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned long long sh_ld64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void sh_st64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave-uniform forms: lane 0 acts, every lane gets the result
__device__ inline unsigned sh_ld_u(const unsigned* p, int lane) {
    unsigned v = 0;
    if (lane == 0) v = sh_ld(p);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ inline unsigned sh_add_u(unsigned* p, unsigned inc, int lane) {
    unsigned v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(p, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// compare-and-swap by lane 0; returns the value found (== expected on success)
__device__ inline unsigned sh_cas_u(unsigned* p, unsigned expected, unsigned desired, int lane) {
    unsigned v = 0;
    if (lane == 0) {
        unsigned e = expected;
        __hip_atomic_compare_exchange_strong(p, &e, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = e;
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ inline void sh_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ inline void sh_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

// ---------------------------------------------------------------------------------------------
// Resumable solves (k_solve's RES; long horizons in large batches).  A launch of 8192 trajectories on 2048 resident
// wavefronts ends with a long tail: the solves that take 100 iterations are as long as a third of the launch, and the
// ones that happen to be pulled late finish alone.  A solve therefore runs at most `res_iters` iterations at a time;
// then its state — x, u, the lane indices and a dozen scalars, 5.3 KB at N = 100: exactly what cs:110-141 carries
// from one iteration to the next — is parked in global memory and its number queued; blocks pull fresh trajectories
// while there are any and parked ones after that, so every long solve is well under way when the short ones are done.
// Whoever resumes it computes the same bits: the cost expansion of an unchanged trajectory (kept in LDS after a failed
// pass, cs:469-475) is recomputed from the same inputs.  (scripts/schedule_sim.py replays measured launches through
// this policy: configs[3]'s shard 51.1 -> 47.0 ms at 32 iterations per slice.)
// Hand-over between blocks (which sit on different XCDs as a rule: their L2s are not coherent): every word of the
// parked state is stored and loaded as an 8-byte agent-scope atomic (write-through / L1-bypassing on gfx950) — a valid
// publication form on its own; an agent-scope release / acquire pair per park would write back and invalidate whole
// L2s hundreds of times per millisecond in mid-launch (measured: the launch three times as long).
// Queue: q[cap] 64-bit entries, cap >= batch (a trajectory is queued at most once at a time), entry = (push number + 1)
// << 32 | trajectory; the launch starts with q zeroed.  push: the state's stores have completed (s_waitcnt), take a
// number (SH_Q_RESV), store the entry.  pop: the entry at SH_Q_HEAD carries that head's number once it is there;
// compare-and-swap the head.  State layout per trajectory (doubles): x | u | scalars | ridx (two per double).
#define CILQR_PARK_SCALARS 16
__host__ __device__ inline size_t park_doubles(int N) {
    return (size_t)(4 * (N + 1) + 2 * N + CILQR_PARK_SCALARS + (N + 2) / 2 + 1);
}
__device__ inline void park_st(double* p, double v) { sh_st64(reinterpret_cast<unsigned long long*>(p), dm_to_bits(v)); }
__device__ inline double park_ld(const double* p) { return dm_from_bits(sh_ld64(reinterpret_cast<const unsigned long long*>(p))); }
__device__ inline void rq_push(unsigned* ctl, unsigned long long* q, unsigned cap, unsigned b, int lane) {
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every lane's stores of the parked state have completed
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        const unsigned s = __hip_atomic_fetch_add(ctl + SH_Q_RESV, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_st64(q + (s % cap), ((unsigned long long)(s + 1u) << 32) | (unsigned long long)b);
        (void)__hip_atomic_fetch_add(ctl + SH_PARKED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// a parked trajectory if one is queued, else -1; never waits
__device__ inline int rq_pop(unsigned* ctl, const unsigned long long* q, unsigned cap, int lane) {
    int b = -1;
    if (lane == 0) {
        for (int tries = 0; tries < 64; ++tries) {
            const unsigned h = sh_ld(ctl + SH_Q_HEAD);
            const unsigned long long e = sh_ld64(q + (h % cap));
            if ((unsigned)(e >> 32) != h + 1u) break; // nothing there (yet)
            unsigned expect = h;
            if (__hip_atomic_compare_exchange_strong(ctl + SH_Q_HEAD, &expect, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                b = (int)(unsigned)(e & 0xffffffffULL);
                break;
            }
        }
    }
    return __builtin_amdgcn_readfirstlane(b);
}
// The same queue taken from by CLAIM instead of compare-and-swap (the grouped build's sliced solves, cilqr_group.hpp): a taker
// reserves the next position with one fetch-and-add on SH_Q_HEAD and then owns it — it reads the entry when the push that
// was (or will be) given that number has stored it.  No retry loops: when thousands of slots reach the end of a slice within the
// same microseconds, compare-and-swap lets one of them through per round trip to memory (measured: 3 us per hand-over,
// config 3 three times as long), fetch-and-adds pipeline.  A claim beyond the pushes so far is a place in line for the next push.
__device__ inline bool rq_avail(const unsigned* ctl, int lane) { // pushes not yet claimed?
    int d = 0;
    if (lane == 0) d = ((int)(sh_ld(ctl + SH_Q_RESV) - sh_ld(ctl + SH_Q_HEAD)) > 0) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(d) != 0;
}
__device__ inline unsigned rq_claim(unsigned* ctl, int lane) {
    unsigned h = 0;
    if (lane == 0) h = __hip_atomic_fetch_add(ctl + SH_Q_HEAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)h);
}
__device__ inline int rq_poll(const unsigned long long* q, unsigned cap, unsigned h, int lane) { // the trajectory at position h, or -1: not there yet
    int b = -1;
    if (lane == 0) {
        const unsigned long long e = sh_ld64(q + (h % cap));
        if ((unsigned)(e >> 32) == h + 1u) b = (int)(unsigned)(e & 0xffffffffULL);
    }
    return __builtin_amdgcn_readfirstlane(b);
}
__device__ inline bool rq_nonempty(const unsigned* ctl, const unsigned long long* q, unsigned cap, int lane) {
    int d = 0;
    if (lane == 0) {
        const unsigned h = sh_ld(ctl + SH_Q_HEAD);
        d = ((unsigned)(sh_ld64(q + (h % cap)) >> 32) == h + 1u) ? 1 : 0;
    }
    return __builtin_amdgcn_readfirstlane(d) != 0;
}

// owner: announce the search whose trials t0 .. 19 sit in the slab; the owner keeps t0 and t0 + 1.  false = no slot free
__device__ inline bool sh_open(unsigned* ctl, ShareReq* rq, int* hints, const int* ridx, int b, int slot, int N, int t0,
                               int idx0, unsigned long long rho_bits, unsigned seq, int lane) {
    for (int k = lane; k <= N; k += CILQR_WAVE) hints[k] = ridx[k];
    if (lane < CILQR_MAX_ALPHA_TRIALS) rq->J[lane] = CILQR_SH_PENDING;
    if (lane == 0) { rq->idx0 = idx0; rq->slot = slot; rq->rho_bits = rho_bits; }
    sh_release(); // the slab, the hints and the marks are out before the claim word opens
    const unsigned own = (unsigned)((t0 + 2 < CILQR_MAX_ALPHA_TRIALS) ? t0 + 2 : CILQR_MAX_ALPHA_TRIALS);
    if (lane == 0) sh_st(&rq->claim, (seq << 16) | own);
    bool placed = false;
    for (int i = 0; i < 4 && !placed; ++i) {
        unsigned* slot = ctl + SH_SLOT0 + ((unsigned)(b + 17 * i) % CILQR_SH_NSLOT);
        placed = (sh_cas_u(slot, 0u, (unsigned)b + 1u, lane) == 0u);
    }
    if (placed) (void)sh_add_u(ctl + SH_ANNOUNCED, 1u, lane);
    return placed; // (not placed: the claim word stays open but nobody is pointed at it; sh_owner_close closes it)
}
// owner: take trial t for itself unless a helper has it (trials go out in ascending order: t is free iff next == t)
__device__ inline bool sh_take(ShareReq* rq, unsigned seq, int t, int lane) {
    for (int tries = 0; tries < 64; ++tries) {
        const unsigned v = sh_ld_u(&rq->claim, lane);
        const unsigned next = v & 0xffu;
        if (next > (unsigned)t) return false; // a helper's
        if (sh_cas_u(&rq->claim, v, (v & ~0xffu) | (unsigned)(t + 1), lane) == v) return true;
    }
    return false;
}
// owner: the cost a helper delivers for trial t; false = not delivered within the bound
__device__ inline bool sh_await(unsigned* ctl, ShareReq* rq, int t, int lane, double& J) {
    unsigned long long v = CILQR_SH_PENDING;
    for (int spin = 0; spin < CILQR_SH_SPIN_CAP; ++spin) {
        unsigned long long w = 0;
        if (lane == 0) w = sh_ld64(&rq->J[t]);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w & 0xffffffffULL));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w >> 32));
        v = ((unsigned long long)hi << 32) | lo;
        if (v != CILQR_SH_PENDING) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (v == CILQR_SH_PENDING) {
        if (lane == 0) sh_st(ctl + SH_ERROR, 1u);
        return false;
    }
    J = dm_from_bits(v);
    return true;
}
// The owner's side as the solve loop calls it, once per trial of an announced search.  Kept out of line: the loop
// pays for it in registers otherwise (26 -> 50 spilled vector registers), and it runs on a few searches per launch.
// State word: bit 0 announced, bit 1 in a slot, bits 8-15 first trial that is not the owner's, bit 31 (result only)
// the cost of trial t0 was delivered by a helper and is in *Jout (LDS).
#define SH_ST_ON 1u
#define SH_ST_PLACED 2u
#define SH_ST_FOREIGN 0x80000000u
__device__ __attribute__((noinline)) unsigned sh_owner_step(unsigned* ctl, ShareReq* rq, int* hints, const int* ridx,
                                                            double* Jout, int b, int slot, int N, int t0, int idx0,
                                                            unsigned long long rho_bits, unsigned seq, unsigned st, int lane) {
    st &= ~SH_ST_FOREIGN;
    if (!(st & SH_ST_ON)) {
        const bool placed = sh_open(ctl, rq, hints, ridx, b, slot, N, t0, idx0, rho_bits, seq, lane);
        const int own0 = (t0 + 2 < CILQR_MAX_ALPHA_TRIALS) ? t0 + 2 : CILQR_MAX_ALPHA_TRIALS;
        st = SH_ST_ON | (placed ? SH_ST_PLACED : 0u) | ((unsigned)own0 << 8);
    }
    const int own = (int)((st >> 8) & 0xffu);
    if (t0 < own) return st;
    if (sh_take(rq, seq, t0, lane)) return (st & ~0xff00u) | ((unsigned)(t0 + 1) << 8);
    double J;
    if (!sh_await(ctl, rq, t0, lane, J)) return st; // not delivered (flagged): the owner costs it
    if (lane == 0) *Jout = J;
    wave_sync();
    return st | SH_ST_FOREIGN;
}
// owner: close the search; returns when every trial a helper claimed has been delivered (the slab is free again)
// (t_last = the last trial whose verdict was taken: everything handed out beyond it went to helpers)
__device__ __attribute__((noinline)) void sh_owner_close(unsigned* ctl, ShareReq* rq, int b, unsigned seq, unsigned st,
                                                         int t_last, int lane) {
    const bool placed = (st & SH_ST_PLACED) != 0u;
    unsigned v = sh_ld_u(&rq->claim, lane);
    for (int tries = 0; tries < 1024; ++tries) {
        const unsigned f = sh_cas_u(&rq->claim, v, ((seq + 1u) << 16) | 0xffu, lane);
        if (f == v) break;
        v = f;
    }
    int next = (int)(v & 0xffu);
    next = (next < CILQR_MAX_ALPHA_TRIALS) ? next : CILQR_MAX_ALPHA_TRIALS;
    const int own = (int)((st >> 8) & 0xffu); // trials below this were the owner's
    for (int t = (t_last + 1 > own) ? t_last + 1 : own; t < next; ++t) {
        double dummy;
        (void)sh_await(ctl, rq, t, lane, dummy);
    }
    if (placed) {
        for (int i = 0; i < 4; ++i) {
            unsigned* slot = ctl + SH_SLOT0 + ((unsigned)(b + 17 * i) % CILQR_SH_NSLOT);
            if (sh_cas_u(slot, (unsigned)b + 1u, 0u, lane) == (unsigned)b + 1u) break;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// hpp/cs:701-713 lagrangian_derivative_and_Hessian: the scalar s with b_dot = s * c_dot,
// b_ddot = b_dot * c_dot^T (s = 0 when the constraint is inactive)
__device__ inline double alm_slope(double cv, double rho, double mu) {
    double t = cv + mu / rho;
    return (t > 0) ? rho * t : 0.0;
}
// cs:622-637 / 673-676 multiplier update
__device__ inline double alm_next_mu(const Cst& c, double mu, double rho, double cv) {
    double v = mu + rho * cv;
    v = (v > 0.0) ? v : 0.0;
    v = (c.k->max_mu < v) ? c.k->max_mu : v;
    return v;
}

// model Jacobians of step k (ut:285-342) into l.kd: a02 a03 a12 a13 a32 | b01 b11 b31.
// sy, cy = dm_sincos(yaw) of row k (the caller has them already).
__device__ inline void model_jacobians_row(const Cst& c, const Lds& l, int k, double v, double yaw, double sy, double cy,
                                           int kds = CILQR_KD) {
    const double delta = l.u[2 * k + 1];
    double* A = l.kd + kds * k; // (kds: doubles per step — CILQR_KD in LDS, the row length when l.kd points into global rows)
    double* B = A + CILQR_KD_B;
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
        dm_sincos(beta + yaw, &sby, &cby);
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

// The Jacobians alone: the backward sweep overwrites them with the gains, so an iteration that keeps
// the cost expansion of an unchanged trajectory (cs:469-475) recomputes them — same inputs, same bits.
__device__ inline void model_jacobians(const Cst& c, const Lds& l, int lane) {
    for (int k = lane; k < c.N; k += CILQR_WAVE) {
        const double* xk = l.x + 4 * k;
        double sy, cy;
        dm_sincos(xk[3], &sy, &cy);
        model_jacobians_row(c, l, k, xk[2], xk[3], sy, cy);
    }
    wave_sync();
}

// get_total_cost_derivatives_and_Hessians (cs:463-690) and get_kinematic_model_derivatives
// (ut:285-342), lane = k.  Row k holds l_x[k], l_xx[k]; lane k also produces l_u[k-1], l_uu[k-1]
// (they depend on u[k-1]) and, for k < N, the model Jacobian entries of step k.
// ALM = false: exponential barriers, l_xx packed (symmetric).  ALM = true: augmented Lagrangian
// terms (cs:581-643, 665-680), l_xx dense (b_dot c_dot^T is not bitwise symmetric), and the
// multiplier proposal alm_mu_next is written.
// LG = true (barrier mode only): the expansion goes to the 128-byte rows of l.gl in global memory (CILQR_GL_ROW)
// GROW (LG, barrier mode): doubles per row — CILQR_GL_ROW, or CILQR_GRP_ROW = 32 for the rows of a trajectory whose whole sweep
// input streams from global memory (cilqr_group.hpp, backward_sweep_pair): the same 16 slots, then the step's eight
// Jacobian entries at slot 16 (l.kd then points at slot 16 of row 0)
#define CILQR_GRP_ROW 32
#define CILQR_GRP_ROW_JAC 16
// SROW (ALM + LG, the grouped kernel): the 256-byte row also carries the step's eight Jacobian entries, at slot
// CILQR_GLA_JAC = 24 (l.kd then points at slot 24 of row 0) — the row's zero slot moves onto l_xx[0][2], a structural zero
#define CILQR_GLA_JAC 24
#define CILQR_GLA_ZERO_S (CILQR_GLA_LXX + 2)
template <bool ALM, bool LG = false, int GROW = CILQR_GL_ROW, bool SROW = false>
__device__ inline void cost_and_model_derivatives(const Cst& c, const Lds& l, const AlmSt& al, int lane) {
    const int N = c.N;
    static_assert(GROW == CILQR_GL_ROW || (LG && !ALM), "wide rows: barrier mode, expansion in global memory");
    static_assert(!SROW || (ALM && LG), "rows with the Jacobians inside: barrier mode says so through GROW");
    gdouble_w* const grows = LG ? (gdouble_w*)l.gl : nullptr;
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        double xk[4] = {l.x[4 * k], l.x[4 * k + 1], l.x[4 * k + 2], l.x[4 * k + 3]};
        int ridx = l.ridx[k];
        double rx, ry;
        lane_point(c, l, ridx, rx, ry);
        gdouble* aux = c.lane_aux + (size_t)ridx * CILQR_AUX_STRIDE;
        const double ryaw = aux[0], sr = aux[1], cr = aux[2];
        double e0 = xk[0] - rx, e1 = xk[1] - ry, e2 = xk[2] - c.k->ref_velo, e3 = xk[3] - ryaw;
        // prime parts (cs:493-494)
        double lx0 = (2 * e0) * c.k->w_pos, lx1 = (2 * e1) * c.k->w_pos, lx2 = (2 * e2) * c.k->w_vel, lx3 = (2 * e3) * c.k->w_yaw;
        double h00 = 0, h01 = 0, h03 = 0, h11 = 0, h13 = 0, h33 = 0, h22 = 0; // barrier Hessian
        double b0 = 0, b1 = 0, b2 = 0, b3 = 0;                                 // barrier gradient
        double sy, cy;
        dm_sincos(xk[3], &sy, &cy);
        ObsRec rec = {0.0, 0.0, 0.0, 0.0};
        gdouble* po = obs_at(c, 0, k);
        const size_t po_step = (size_t)c.T * CILQR_OBS_STRIDE;
        if (k >= 1 && c.M > 0) obs_fetch(rec, po);
        double g10 = 0, g30 = 0, g31 = 0; // ALM only: the other halves of the non-symmetric Hessian
        if (ALM && k >= 1) {
            const double um0 = l.u[2 * k - 2], um1 = l.u[2 * k - 1];
            const double* mu = al.mu + (size_t)(k - 1) * al.C;
            double* mun = al.mu_next + (size_t)(k - 1) * al.C;
            const double rho = al.rho;
            const double d_sign = e1 * cr - e0 * sr;
            const double hyp = dm_hypot(e0, e1);
            const double cur_d = (d_sign < 0) ? -hyp : hyp;
            const double cv[8] = {um0 - c.k->acc_max, c.k->acc_min - um0, um1 - c.k->stl_lim, -c.k->stl_lim - um1,
                                  xk[2] - c.k->velo_max, c.k->velo_min - xk[2], cur_d - c.k->pos_up_b, c.k->pos_lo_b - cur_d};
            double sl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sl[j] = alm_slope(cv[j], rho, mu[j]);
                mun[j] = alm_next_mu(c, mu[j], rho, cv[j]);
            }
            if (LG) {
                gdouble_w* rm = grows + (size_t)CILQR_GL_ROW_ALM * (k - 1);
                rm[CILQR_GLA_LU] = 2 * (um0 * c.k->w_acc) + (sl[0] - sl[1]);
                rm[CILQR_GLA_LU + 1] = 2 * (um1 * c.k->w_stl) + (sl[2] - sl[3]);
                rm[CILQR_GLA_LUU] = 2 * c.k->w_acc + (sl[0] + sl[1]);
                rm[CILQR_GLA_LUU + 1] = 2 * c.k->w_stl + (sl[2] + sl[3]);
            } else {
                l.lu[2 * (k - 1)] = 2 * (um0 * c.k->w_acc) + (sl[0] - sl[1]);
                l.lu[2 * (k - 1) + 1] = 2 * (um1 * c.k->w_stl) + (sl[2] - sl[3]);
                l.luu[2 * (k - 1)] = 2 * c.k->w_acc + (sl[0] + sl[1]);
                l.luu[2 * (k - 1) + 1] = 2 * c.k->w_stl + (sl[2] + sl[3]);
            }
            double px = e0 / hyp, py = e1 / hyp;
            if (d_sign < 0) { px = -px; py = -py; }
            const double nx = -px, ny = -py;
            const double ux = sl[6] * px, uy = sl[6] * py, vx = sl[7] * nx, vy = sl[7] * ny; // b_dot of pos_up / pos_lo
            b0 = ux + vx;
            b1 = uy + vy;
            b2 = sl[4] - sl[5];
            b3 = 0.0;
            h00 = ux * px + vx * nx;
            h01 = ux * py + vx * ny;
            g10 = uy * px + vy * nx;
            h11 = uy * py + vy * ny;
            h22 = sl[4] + sl[5];
            for (int o = 0; o < c.M; ++o) {
                ObsRec nxt = rec;
                po += po_step;
                if (o + 1 < c.M) obs_fetch(nxt, po);
                ObsOut t;
                obstacle_terms<true>(c, xk, sy, cy, rec, t);
                rec = nxt;
                const double sf = alm_slope(t.mf, rho, mu[8 + 2 * o]);
                const double sr2 = alm_slope(t.mr, rho, mu[9 + 2 * o]);
                mun[8 + 2 * o] = alm_next_mu(c, mu[8 + 2 * o], rho, t.mf);
                mun[9 + 2 * o] = alm_next_mu(c, mu[9 + 2 * o], rho, t.mr);
                const double f0 = sf * t.gf[0], f1 = sf * t.gf[1], f3 = sf * t.gf[2];
                const double r0 = sr2 * t.gr[0], r1 = sr2 * t.gr[1], r3 = sr2 * t.gr[2];
                b0 = b0 + (f0 + r0);
                b1 = b1 + (f1 + r1);
                b3 = b3 + (f3 + r3);
                h00 = h00 + (f0 * t.gf[0] + r0 * t.gr[0]);
                h01 = h01 + (f0 * t.gf[1] + r0 * t.gr[1]);
                h03 = h03 + (f0 * t.gf[2] + r0 * t.gr[2]);
                g10 = g10 + (f1 * t.gf[0] + r1 * t.gr[0]);
                h11 = h11 + (f1 * t.gf[1] + r1 * t.gr[1]);
                h13 = h13 + (f1 * t.gf[2] + r1 * t.gr[2]);
                g30 = g30 + (f3 * t.gf[0] + r3 * t.gr[0]);
                g31 = g31 + (f3 * t.gf[1] + r3 * t.gr[1]);
                h33 = h33 + (f3 * t.gf[2] + r3 * t.gr[2]);
            }
        }
        if (!ALM && k >= 1) {
            double um0 = l.u[2 * k - 2], um1 = l.u[2 * k - 1];
            // control bounds (cs:510-513, 537-558)
            // (no scheduling fences between the exponentials any more: round 1 had them to hold the register count
            //  down at one wavefront per SIMD; with two per SIMD leaving the order to the compiler is 2 % faster)
            double b_au = c.k->sq1 * dm_exp(c.k->sq2 * (um0 - c.k->acc_max));
            double b_al = c.k->sq1 * dm_exp(c.k->sq2 * (c.k->acc_min - um0));
            double b_su = c.k->sq1 * dm_exp(c.k->sq2 * (um1 - c.k->stl_lim));
            double b_sl = c.k->sq1 * dm_exp(c.k->sq2 * (-c.k->stl_lim - um1));
            double q22 = c.k->sq2 * c.k->sq2;
            double lub0 = (c.k->sq2 * b_au) - (c.k->sq2 * b_al);
            double lub1 = (c.k->sq2 * b_su) - (c.k->sq2 * b_sl);
            double luub0 = (q22 * b_au) + (q22 * b_al);
            double luub1 = (q22 * b_su) + (q22 * b_sl);
            // l_u = 2 (u R) + barrier, l_uu = 2 R + barrier (cs:491-492, 686-687)
            if (LG) {
                gdouble_w* rm = grows + (size_t)GROW * (k - 1);
                rm[CILQR_GL_LU] = 2 * (um0 * c.k->w_acc) + lub0;
                rm[CILQR_GL_LU + 1] = 2 * (um1 * c.k->w_stl) + lub1;
                rm[CILQR_GL_LUU] = 2 * c.k->w_acc + luub0;
                rm[CILQR_GL_LUU + 1] = 2 * c.k->w_stl + luub1;
            } else {
                l.lu[2 * (k - 1)] = 2 * (um0 * c.k->w_acc) + lub0;
                l.lu[2 * (k - 1) + 1] = 2 * (um1 * c.k->w_stl) + lub1;
                l.luu[2 * (k - 1)] = 2 * c.k->w_acc + luub0;
                l.luu[2 * (k - 1) + 1] = 2 * c.k->w_stl + luub1;
            }
            // velocity bounds and road borders (cs:507-533, 560-580)
            double b_vu = c.k->sq1 * dm_exp(c.k->sq2 * (xk[2] - c.k->velo_max));
            double b_vl = c.k->sq1 * dm_exp(c.k->sq2 * (c.k->velo_min - xk[2]));
            double d_sign = e1 * cr - e0 * sr;
            double hyp = dm_hypot(e0, e1);
            double cur_d = (d_sign < 0) ? -hyp : hyp;
            double b_pu = c.k->sq1 * dm_exp(c.k->sq2 * (cur_d - c.k->pos_up_b));
            double b_pl = c.k->sq1 * dm_exp(c.k->sq2 * (c.k->pos_lo_b - cur_d));
            double px = e0 / hyp, py = e1 / hyp;
            if (d_sign < 0) { px = -px; py = -py; }
            double nx = -px, ny = -py; // pos_lo_constr_over_x = -1 * pos_up_constr_over_x
            double d_pu = c.k->sq2 * b_pu, d_pl = c.k->sq2 * b_pl;
            double s_pu = q22 * b_pu, s_pl = q22 * b_pl;
            b0 = d_pu * px + d_pl * nx;
            b1 = d_pu * py + d_pl * ny;
            b2 = (c.k->sq2 * b_vu) - (c.k->sq2 * b_vl);
            b3 = 0.0;
            h00 = s_pu * (px * px) + s_pl * (nx * nx);
            h01 = s_pu * (px * py) + s_pl * (nx * ny);
            h11 = s_pu * (py * py) + s_pl * (ny * ny);
            h22 = (q22 * b_vu) + (q22 * b_vl);
            // obstacles (cs:647-683)
            double oq22 = c.k->oq2 * c.k->oq2;
            for (int o = 0; o < c.M; ++o) {
                ObsRec nxt = rec;
                po += po_step;
                if (o + 1 < c.M) obs_fetch(nxt, po);
                ObsOut t;
                obstacle_terms<true>(c, xk, sy, cy, rec, t);
                rec = nxt;
                double bf = c.k->oq1 * dm_exp(c.k->oq2 * t.mf);
                double br = c.k->oq1 * dm_exp(c.k->oq2 * t.mr);
                double df = c.k->oq2 * bf, dr = c.k->oq2 * br;
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
        if (LG && ALM) {
            gdouble_w* r = grows + (size_t)CILQR_GL_ROW_ALM * k;
            r[CILQR_GLA_LX] = lx0 + b0;
            r[CILQR_GLA_LX + 1] = lx1 + b1;
            r[CILQR_GLA_LX + 2] = lx2 + b2;
            r[CILQR_GLA_LX + 3] = lx3 + b3;
            gdouble_w* hx = r + CILQR_GLA_LXX;
            hx[0] = 2 * c.k->w_pos + h00; hx[1] = 0.0 + h01; hx[2] = 0.0; hx[3] = 0.0 + h03;
            hx[4] = 0.0 + g10; hx[5] = 2 * c.k->w_pos + h11; hx[6] = 0.0; hx[7] = 0.0 + h13;
            hx[8] = 0.0; hx[9] = 0.0; hx[10] = 2 * c.k->w_vel + h22; hx[11] = 0.0;
            hx[12] = 0.0 + g30; hx[13] = 0.0 + g31; hx[14] = 0.0; hx[15] = 2 * c.k->w_yaw + h33;
            if (!SROW) r[CILQR_GLA_ZERO] = 0.0;
            if (k < N) model_jacobians_row(c, l, k, xk[2], xk[3], sy, cy, SROW ? CILQR_GL_ROW_ALM : CILQR_KD);
            continue;
        }
        if (LG) {
            gdouble_w* r = grows + (size_t)GROW * k;
            r[CILQR_GL_LX] = lx0 + b0;
            r[CILQR_GL_LX + 1] = lx1 + b1;
            r[CILQR_GL_LX + 2] = lx2 + b2;
            r[CILQR_GL_LX + 3] = lx3 + b3;
            r[CILQR_GL_LXX] = 2 * c.k->w_pos + h00;
            r[CILQR_GL_LXX + 1] = 0.0 + h01;
            r[CILQR_GL_LXX + 2] = 0.0 + h03;
            r[CILQR_GL_LXX + 3] = 2 * c.k->w_pos + h11;
            r[CILQR_GL_LXX + 4] = 0.0 + h13;
            r[CILQR_GL_LXX + 5] = 2 * c.k->w_yaw + h33;
            r[CILQR_GL_LXX22] = 2 * c.k->w_vel + h22;
            r[CILQR_GL_ZERO] = 0.0;
            if (k < N) model_jacobians_row(c, l, k, xk[2], xk[3], sy, cy, GROW == CILQR_GL_ROW ? CILQR_KD : GROW);
            continue;
        }
        l.lx[4 * k] = lx0 + b0;
        l.lx[4 * k + 1] = lx1 + b1;
        l.lx[4 * k + 2] = lx2 + b2;
        l.lx[4 * k + 3] = lx3 + b3;
        if (ALM) {
            double* hx = l.lxx + 16 * k;
            hx[0] = 2 * c.k->w_pos + h00; hx[1] = 0.0 + h01; hx[2] = 0.0; hx[3] = 0.0 + h03;
            hx[4] = 0.0 + g10; hx[5] = 2 * c.k->w_pos + h11; hx[6] = 0.0; hx[7] = 0.0 + h13;
            hx[8] = 0.0; hx[9] = 0.0; hx[10] = 2 * c.k->w_vel + h22; hx[11] = 0.0;
            hx[12] = 0.0 + g30; hx[13] = 0.0 + g31; hx[14] = 0.0; hx[15] = 2 * c.k->w_yaw + h33;
        } else {
            double* hx = l.lxx + 7 * k;
            hx[0] = 2 * c.k->w_pos + h00;
            hx[1] = 0.0 + h01;
            hx[2] = 0.0 + h03;
            hx[3] = 2 * c.k->w_pos + h11;
            hx[4] = 0.0 + h13;
            hx[5] = 2 * c.k->w_yaw + h33;
            hx[6] = 2 * c.k->w_vel + h22;
        }
        if (k < N) model_jacobians_row(c, l, k, xk[2], xk[3], sy, cy);
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// backward_pass (cs:383-440) after the expansion above.  Wave-uniform; V_x, V_xx in registers.
// Returns true on success, false for a non-PD Q_uu (BACKWARD_PASS_FAIL); fills the gains in l.kd, dV.
__device__ inline bool backward_sweep_uniform(const Cst& c, const Lds& l, double lamb, int lane, double dV[2],
                                              int* fail_step = nullptr) {
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
        const double* Ap = l.kd + CILQR_KD * i;
        const double* Bp = Ap + CILQR_KD_B;
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
        if (fail) {
            if (fail_step) *fail_step = i; // steps i .. 0 of l.kd still hold Jacobians, not gains
            return false;
        }
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
            // over the Jacobians of this step, which are in registers by now
            double* kd = l.kd + CILQR_KD * i;
            kd[CILQR_KD_D(0)] = dd[0];
            kd[CILQR_KD_D(1)] = dd[1];
#pragma unroll
            for (int e = 0; e < 8; ++e) kd[CILQR_KD_K(e)] = Kk[e];
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

// ---------------------------------------------------------------------------------------------
// backward_pass (cs:383-440), lane-parallel form.  One backward step is two small matrix products
// and a rank-2 update; written with M = [A | B] (4x6) and W = [V_xx | V_x] (4x5) they become
//   pass 1:  X = M^T W            (6x5)   rows 0-3: A^T V_xx | A^T V_x,  rows 4-5: B^T V_xx | B^T V_x
//   pass 2:  Y = X[:, 0:4] M      (6x6)   (A^T V A, -, B^T V A, B^T V B)
//   Q = l + Y (+ lamb on the Q_uu diagonal);  column 4 of X + (l_x, l_u) = (Q_x, Q_u)
//   W' = [V_xx' | V_x'] = Q[0:4] + K^T Q_uu K_c + K^T Q_c + Q_ux^T K_c   with K_c = (K | d), Q_c = (Q_ux | Q_u)
// so every output element is the same expression of 4 (or 2) products summed in index order —
// the reference's dense Eigen evaluation order, zeros included.  Lane (r', c'') = (lane >> 3,
// lane & 7) owns one element; operands travel through the small LDS exchange area l.xch, the
// per-step coefficients come straight from the stage arrays through per-lane address maps.
// ~145 vector (~220 wave) instructions per step instead of ~540 for the wave-uniform form.
struct LaneMap {
    int m1[4];  // LDS double-offsets (at step 0) of M[k][r'], k = 0..3
    int s1[4];  // their per-step strides
    int m2[4];  // M[k][c'']
    int s2[4];
    int lq, slq;  // L[r'][c''] (l_xx / l_uu / zero)
    int lv, slv;  // l[r'] (l_x / l_u)
};

__device__ inline void lane_map_M(const Lds& l, int k, int j, int& off, int& stride) {
    // M = [A | B], A = I + {a02 a03 a12 a13 a32}, B = {b01 b11 dt b31}
    const int A5 = (int)(l.kd - l.x), B3 = A5 + CILQR_KD_B, CC = (int)(l.xch - l.x) + CILQR_XCH_CONST;
    const int ZERO = CC, ONE = CC + 1, DT = CC + 2;
    off = ZERO; stride = 0;
    if (j < 4 && k == j) off = ONE;
    if (j == 2 && k == 0) { off = A5 + 0; stride = CILQR_KD; }
    if (j == 2 && k == 1) { off = A5 + 2; stride = CILQR_KD; }
    if (j == 2 && k == 3) { off = A5 + 4; stride = CILQR_KD; }
    if (j == 3 && k == 0) { off = A5 + 1; stride = CILQR_KD; }
    if (j == 3 && k == 1) { off = A5 + 3; stride = CILQR_KD; }
    if (j == 4 && k == 2) off = DT;
    if (j == 5 && k == 0) { off = B3 + 0; stride = CILQR_KD; }
    if (j == 5 && k == 1) { off = B3 + 1; stride = CILQR_KD; }
    if (j == 5 && k == 3) { off = B3 + 2; stride = CILQR_KD; }
}

// the same for an expansion held in the rows of l.gl: slots (doubles) inside the 128-byte row of a step
template <int ROWD>
__device__ inline void lane_map_gl(int lane, int& slot_q, int& slot_v) {
    const int rp = (lane >> 3) % 6, cc = lane & 7;
    if (ROWD == CILQR_GL_ROW_ALM) { // dense l_xx
        slot_q = CILQR_GLA_ZERO;
        if (rp < 4 && cc < 4) slot_q = CILQR_GLA_LXX + 4 * rp + cc;
        else if (rp >= 4 && cc == rp) slot_q = CILQR_GLA_LUU + (rp - 4);
        slot_v = (rp < 4) ? CILQR_GLA_LX + rp : CILQR_GLA_LU + (rp - 4);
        return;
    }
    slot_q = CILQR_GL_ZERO;
    if (rp < 4 && cc < 4) {
        const int a = (rp < cc) ? rp : cc, b = (rp < cc) ? cc : rp;
        if (a == 0 && b == 0) slot_q = CILQR_GL_LXX + 0;
        if (a == 0 && b == 1) slot_q = CILQR_GL_LXX + 1;
        if (a == 0 && b == 3) slot_q = CILQR_GL_LXX + 2;
        if (a == 1 && b == 1) slot_q = CILQR_GL_LXX + 3;
        if (a == 1 && b == 3) slot_q = CILQR_GL_LXX + 4;
        if (a == 3 && b == 3) slot_q = CILQR_GL_LXX + 5;
        if (a == 2 && b == 2) slot_q = CILQR_GL_LXX22;
    } else if (rp >= 4 && cc == rp) {
        slot_q = CILQR_GL_LUU + (rp - 4);
    }
    slot_v = (rp < 4) ? CILQR_GL_LX + rp : CILQR_GL_LU + (rp - 4);
}

template <bool LG = false>
__device__ inline void make_lane_map(const Lds& l, int lane, LaneMap& m) {
    const bool dense_lxx = (l.lxs == 16);
    const int rp = (lane >> 3) % 6, cc = lane & 7;      // lanes >= 48 alias rows 0/1 (results unused)
    const int ccm = (cc < 6) ? cc : 5;
    const int CC = (int)(l.xch - l.x) + CILQR_XCH_CONST;
    for (int k = 0; k < 4; ++k) {
        lane_map_M(l, k, rp, m.m1[k], m.s1[k]);
        lane_map_M(l, k, ccm, m.m2[k], m.s2[k]);
    }
    m.lq = CC; m.slq = 0;
    m.lv = CC; m.slv = 0;
    if (LG) return; // the expansion is not in LDS (lane_map_gl)
    // L[r'][c'']: l_xx (7 packed entries 00 01 03 11 13 33 22), l_uu diagonal, zero elsewhere
    const int LXX = (int)(l.lxx - l.x), LUU = (int)(l.luu - l.x);
    if (rp < 4 && cc < 4) {
        int a = (rp < cc) ? rp : cc, b = (rp < cc) ? cc : rp, e = -1;
        if (a == 0 && b == 0) e = 0;
        if (a == 0 && b == 1) e = 1;
        if (a == 0 && b == 3) e = 2;
        if (a == 1 && b == 1) e = 3;
        if (a == 1 && b == 3) e = 4;
        if (a == 3 && b == 3) e = 5;
        if (a == 2 && b == 2) e = 6;
        if (e >= 0) { m.lq = LXX + e; m.slq = 7; }
        if (dense_lxx) { m.lq = LXX + 4 * rp + cc; m.slq = 16; }
    } else if (rp >= 4 && cc == rp) {
        m.lq = LUU + (rp - 4); m.slq = 2;
    }
    if (rp < 4) { m.lv = (int)(l.lx - l.x) + rp; m.slv = 4; }
    else { m.lv = (int)(l.lu - l.x) + (rp - 4); m.slv = 2; }
}

// ---- cross-lane moves of a double (two 32-bit halves) -------------------------------------------
// (every lane of the wave is active where this is used and every lane has a valid source lane, so no lane keeps
// its old value: the move without an `old` operand writes a fresh register and needs no copy of v first)
template <int CTRL>
__device__ inline double dpp_move(double v) {
    const unsigned long long u = dm_to_bits(v);
    const int lo = (int)(unsigned)(u & 0xffffffffULL), hi = (int)(unsigned)(u >> 32);
    const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return dm_from_bits(((unsigned long long)(unsigned)hi2 << 32) | (unsigned long long)(unsigned)lo2);
}
// the same move, taken only by the lanes of the banks in BANKS (bank = (lane % 16) / 4); the others keep v
template <int CTRL, int BANKS>
__device__ inline double dpp_move_banks(double v) {
    const unsigned long long u = dm_to_bits(v);
    int lo = (int)(unsigned)(u & 0xffffffffULL), hi = (int)(unsigned)(u >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, BANKS, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, BANKS, false);
    return dm_from_bits(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
// LDS byte address of a pointer into the block's LDS, made opaque to the optimiser: kept in a register and
// stepped there, instead of being rebuilt every step as (relocated base + offset)
typedef const double __attribute__((address_space(3))) lds_cdouble;
__device__ inline unsigned lds_addr(const double* p) {
    unsigned a = (unsigned)(size_t)(lds_cdouble*)p;
    __asm__("" : "+v"(a));
    return a;
}
__device__ inline double lds_load(unsigned a) { return *(lds_cdouble*)(size_t)a; }
// value of lane `src` (any lane of the wave, per-lane choice): ds_bpermute, no LDS memory involved
__device__ inline double lane_gather(double v, int src) {
    const unsigned long long u = dm_to_bits(v);
    int lo = (int)(unsigned)(u & 0xffffffffULL), hi = (int)(unsigned)(u >> 32);
    lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
    return dm_from_bits(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
// value of a lane known at compile time, delivered wave-uniformly (v_readlane -> SGPR)
template <int SRC>
__device__ inline double lane_bcast(double v) {
    const unsigned long long u = dm_to_bits(v);
    int lo = (int)(unsigned)(u & 0xffffffffULL), hi = (int)(unsigned)(u >> 32);
    lo = __builtin_amdgcn_readlane(lo, SRC);
    hi = __builtin_amdgcn_readlane(hi, SRC);
    return dm_from_bits(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

// The operands that pass 2 and the rank-2 update need live in other lanes' registers; they are moved
// with DPP (inside the 8-lane row of the grid), ds_bpermute (across rows) and v_readlane (the 2x2
// Q_uu and Q_u, needed by every lane) — no LDS round trips inside a step.
// LG: the cost expansion streams in from the rows of l.gl (global memory; this wave wrote them a phase ago) through
// the LDS ring l.ring: rows 4 c .. 4 c + 3 ("chunk" c, 64 doubles, one per lane) are fetched while the four steps of
// chunk c + 1 compute and dropped into the ring half c & 1 when the sweep gets there.  Row r sits at ring offset
// (r & 7) rows, so the per-lane read addresses are the row-independent slot plus a wave-uniform offset.
__device__ inline double gl_load(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, int row_off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_off, row_off, 0));
}
// ROWD = doubles per row of the expansion in global memory (CILQR_GL_ROW, CILQR_GL_ROW_ALM), 0 = expansion in LDS
// GGAIN: the gains go to `gains` in global memory — [N][CILQR_KD] doubles, the layout of Lds::kd — instead of over the
// step's Jacobians in LDS (the builds that hold several trajectories per wavefront keep one Jacobian array for all of
// them; the rollout reads the gains back through L1 / L2, one step ahead)
// QLDS: the four entries of Q_uu reach every lane through ds_bpermute (the LDS crossbar, issued next to the gathers of
// (Q_ux | Q_u) that leave at the same moment) instead of eight v_readlane: with two wavefronts per SIMD the sweep is bound by
// the vector unit's issue slots, and these are eight of them per step.
template <int ROWD = 0, bool GGAIN = false, bool QLDS = false>
__device__ inline bool backward_sweep_lanes(const Cst& c, const Lds& l, double lamb, int lane, double dV[2],
                                            int* fail_step = nullptr, double* gains = nullptr) {
    constexpr bool LG = ROWD != 0;
    constexpr int GL_CHUNK = LG ? CILQR_WAVE / ROWD : 4; // rows per chunk of 64 doubles
    const int N = c.N;
    const int rp = (lane >> 3) % 6, cc = lane & 7;
    const double* const base = l.x;
    LaneMap mp;
    make_lane_map<LG>(l, lane, mp);
    constexpr int ROWB = (LG ? ROWD : CILQR_GL_ROW) * (int)sizeof(double);
    __amdgpu_buffer_rsrc_t grs;
    unsigned goq = 0, gov = 0;
    if (LG) {
        int sq, sv;
        lane_map_gl<ROWD>(lane, sq, sv);
        goq = 8u * (unsigned)sq;
        gov = 8u * (unsigned)sv;
        grs = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(l.gl), 0, (N + 1) * ROWB, 0x00020000);
    }
    if (lane == 0) {
        l.xch[CILQR_XCH_CONST + 0] = 0.0;
        l.xch[CILQR_XCH_CONST + 1] = 1.0;
        l.xch[CILQR_XCH_CONST + 2] = c.dt;
        l.xch[CILQR_XCH_CONST + 3] = 0.0;
    }
    wave_sync();
    __amdgpu_buffer_rsrc_t ggr;
    if (GGAIN) ggr = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(gains), 0, N * CILQR_KD * (int)sizeof(double), 0x00020000);
    // W = [l_xx[N] | l_x[N]], element (r, c) in the register `wn` of lane 8 r + c (r < 4, c <= 4)
    double wn;
    if (LG) wn = gl_load(grs, (cc < 4) ? goq : gov, N * ROWB);
    else wn = (cc < 4) ? base[mp.lq + mp.slq * N] : base[mp.lv + mp.slv * N];
    dV[0] = 0.0;
    dV[1] = 0.0;
    const int wc = (cc <= 4) ? cc : 4;
    const bool diag = (rp >= 4) && (cc == rp);
    const int r4 = rp & 3;
    const int src_c0 = 32 + wc, src_c1 = 40 + wc;   // (Q_ux | Q_u)[0][c], [1][c]
    // Q_ux[0][r], [1][r] for the rows of W.  Rows 4 and 5 of the grid have no part in the update of W: they take
    // r = Q_u instead, which makes their K[:, r] the step d, and carry the expected cost reduction (cs:435-436):
    // with d halved on row 4 (krf), lane (4, 4) gets ta = (0.5 d)^T Q_uu d and lane (5, 4) gets tb = d^T Q_u out of
    // the very expressions the other lanes evaluate for W — the reference's operands in the reference's order
    const int src_r0 = (rp >= 4) ? 36 : 32 + r4, src_r1 = (rp >= 4) ? 44 : 40 + r4;
    const double krf = (rp == 4) ? 0.5 : 1.0;
    double dvacc = 0.0; // lane 36: delta_V[0], lane 44: delta_V[1]
    int q_src[4] = {36, 37, 44, 45};
    if (QLDS) __asm__("" : "+v"(q_src[0]), "+v"(q_src[1]), "+v"(q_src[2]), "+v"(q_src[3]));
    // this lane's ten coefficient addresses (LDS byte addresses) walk backwards with the step
    unsigned am1[4], am2[4], dm1[4], dm2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        am1[k] = lds_addr(base + mp.m1[k] + mp.s1[k] * (N - 1));
        am2[k] = lds_addr(base + mp.m2[k] + mp.s2[k] * (N - 1));
        dm1[k] = 8u * (unsigned)mp.s1[k];
        dm2[k] = 8u * (unsigned)mp.s2[k];
    }
    unsigned alq = lds_addr(base + mp.lq + mp.slq * (N - 1));
    unsigned alv = lds_addr(base + mp.lv + mp.slv * (N - 1));
    const unsigned dlq = 8u * (unsigned)mp.slq, dlv = 8u * (unsigned)mp.slv;
    constexpr int CHB = GL_CHUNK * ROWB; // bytes per chunk (512)
    double chunk = 0.0;                        // this lane's double of the chunk in flight
    unsigned rq = 0, rv = 0;                   // this lane's two read addresses inside ring row 0
    if (LG) {
        const int c0 = (N - 1) / GL_CHUNK;
        // (rows past N - 1 of the first chunk are not used; reads past row N return zero: the descriptor ends there)
        const double first = gl_load(grs, 8u * (unsigned)lane, c0 * CHB);
        ((double*)l.ring)[(c0 & 1) * CILQR_WAVE + lane] = first;
        if (c0 > 0) chunk = gl_load(grs, 8u * (unsigned)lane, (c0 - 1) * CHB);
        rq = lds_addr(l.ring) + goq;
        rv = lds_addr(l.ring) + gov;
    }
    for (int i = N - 1; i >= 0; --i) {
        if (LG && (i & (GL_CHUNK - 1)) == GL_CHUNK - 1 && i != N - 1) {
            // the sweep enters chunk c: its rows arrived while chunk c + 1 was computed; fetch chunk c - 1
            const int cch = i / GL_CHUNK;
            ((double*)l.ring)[(cch & 1) * CILQR_WAVE + lane] = chunk;
            if (cch > 0) chunk = gl_load(grs, 8u * (unsigned)lane, (cch - 1) * CHB);
        }
        // per-lane coefficients of this step (issued together with the cross-lane moves of pass 1, whose
        // latency they share; fetching them a step ahead was measured and is slower)
        double m1[4], m2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m1[k] = lds_load(am1[k]);
            m2[k] = lds_load(am2[k]);
            am1[k] -= dm1[k];
            am2[k] -= dm2[k];
        }
        double Lq, lv;
        if (LG) {
            const unsigned ro = (unsigned)(i & (2 * GL_CHUNK - 1)) * (unsigned)ROWB; // wave-uniform
            Lq = lds_load(rq + ro);
            lv = lds_load(rv + ro);
        } else {
            Lq = lds_load(alq);
            lv = lds_load(alv);
            alq -= dlq;
            alv -= dlv;
        }
        // pass 1: column wc of W from the lanes 8 k + wc
        const double w0 = lane_gather(wn, wc), w1 = lane_gather(wn, 8 + wc);
        const double w2 = lane_gather(wn, 16 + wc), w3 = lane_gather(wn, 24 + wc);
        const double X = CQ_MADD(m1[3], w3, CQ_MADD(m1[2], w2, CQ_MADD(m1[1], w1, m1[0] * w0)));
        const double Zv = lv + X; // (Q_x, Q_u) on the lanes of column 4
        // pass 2: row r' of X[:, 0:4] sits in lanes 8 r' + 0..3; bring it to both quads of the row, then
        // broadcast inside each quad
        const double Xq = dpp_move_banks<0x114, 0xA>(X);  // row_shr:4 on the lanes with cc >= 4 (banks 1 and 3)
        const double x0 = dpp_move<0x00>(Xq), x1 = dpp_move<0x55>(Xq);
        const double x2 = dpp_move<0xAA>(Xq), x3 = dpp_move<0xFF>(Xq);
        const double Y = CQ_MADD(x3, m2[3], CQ_MADD(x2, m2[2], CQ_MADD(x1, m2[1], x0 * m2[0])));
        double Q = Lq + Y;
        if (diag) Q = Q + lamb;
        // Q_uu, Q_u on every lane; PD test and inverse (cs:415-421)
        double Quu0, Quu1, Quu2, Quu3;
        if (QLDS) {
            // (source lanes in vector registers the optimiser cannot see through: a ds_bpermute from a constant lane would be
            //  turned back into v_readlane)
            Quu0 = lane_gather(Q, q_src[0]); Quu1 = lane_gather(Q, q_src[1]);
            Quu2 = lane_gather(Q, q_src[2]); Quu3 = lane_gather(Q, q_src[3]);
        } else {
            Quu0 = lane_bcast<36>(Q); Quu1 = lane_bcast<37>(Q);
            Quu2 = lane_bcast<44>(Q); Quu3 = lane_bcast<45>(Q);
        }
        // (the lane predicates are recomputed from a vector register each step: one compare instead of the two
        //  v_readlane a spilled scalar mask costs)
        int ccv = cc;
        __asm__("" : "+v"(ccv));
        const double S = (ccv == 4) ? Zv : Q;               // rows 4-5: (Q_ux | Q_u)
        const double c0 = lane_gather(S, src_c0), c1 = lane_gather(S, src_c1);
        const double r0 = lane_gather(S, src_r0), r1 = lane_gather(S, src_r1);
        // Matrix2d::inverse() (cs:421) is started before the verdict on Q_uu is in: its division is the longest
        // dependent chain of the step and the verdict's arithmetic fills its bubbles; a failed verdict discards it
        const double det = Quu0 * Quu3 - Quu2 * Quu1;
        const double invdet = 1.0 / det;
        // Eigen::LLT's verdict: Quu0 > 0 and the second pivot Quu3 - (Quu2 / sqrt(Quu0))^2 > 0.  The pivot as
        // computed is Quu3 - (Quu2^2 / Quu0)(1 + e), |e| < 2^-50 (one sqrt, one quotient, one square, and the
        // final subtraction keeps the sign).  So when Quu0 and Quu3 are positive and of ordinary size
        // (2^-332 .. 2^332: no product below leaves the normal numbers) and Quu0 Quu3 exceeds Quu2^2 by a factor
        // 1 + 2^-40, the pivot is positive and need not be evaluated.  Anything else — including NaN, for
        // which every comparison is false — takes the exact path.
        bool fail = false;
        {
            const unsigned h0 = (unsigned)(dm_to_bits(Quu0) >> 32), h3 = (unsigned)(dm_to_bits(Quu3) >> 32);
            const bool ordinary = ((h0 - 0x2B300000u) < 0x29800000u) && ((h3 - 0x2B300000u) < 0x29800000u);
            bool surely_pd = ordinary && (Quu0 * Quu3 > (Quu2 * Quu2) * 1.0000000000009095);
            if (QLDS) surely_pd = (__ballot(!surely_pd) == 0ULL); // (every lane holds the same four numbers: make the verdict scalar)
            if (!surely_pd) {
                if (Quu0 <= 0.0) {
                    fail = true;
                } else {
                    double l00 = dm_sqrt(Quu0);
                    double l10 = Quu2 / l00;
                    double piv1 = Quu3 - l10 * l10;
                    if (piv1 <= 0.0) fail = true;
                }
                if (QLDS) fail = (__ballot(fail) != 0ULL);
            }
        }
        if (fail) {
            if (fail_step) *fail_step = i; // steps i .. 0 of l.kd still hold Jacobians, not gains
            return false;
        }
        const double n00 = -(Quu3 * invdet), n01 = -(-Quu1 * invdet), n10 = -(-Quu2 * invdet), n11 = -(Quu0 * invdet);
        const double kc0 = CQ_MADD(n01, c1, n00 * c0), kc1 = CQ_MADD(n11, c1, n10 * c0); // (K | d)[:, c]
        double kr0 = CQ_MADD(n01, r1, n00 * r0), kr1 = CQ_MADD(n11, r1, n10 * r0); // K[:, r]
        kr0 = kr0 * krf; // (exact: the factor is 1, or 0.5 on row 4 — hd = 0.5 d of cs:435)
        kr1 = kr1 * krf;
        const double p0 = CQ_MADD(kr1, Quu2, kr0 * Quu0);                  // (K^T Q_uu)[r][:]
        const double p1 = CQ_MADD(kr1, Quu3, kr0 * Quu1);
        const double ta = CQ_MADD(p1, kc1, p0 * kc0);
        const double tb = CQ_MADD(kr1, c1, kr0 * c0);
        const double tc = CQ_MADD(r1, kc1, r0 * kc0);
        const double own = (ccv < 4) ? Q : Zv;
        wn = ((own + ta) + tb) + tc;
        int lanev = lane;
        __asm__("" : "+v"(lanev));
        if (lanev < 5) { // row r' = 0 holds (K | d) column c
            if (GGAIN) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, kc0), ggr, 8u * (unsigned)lane, i * (CILQR_KD * 8), 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, kc1), ggr, 8u * (unsigned)lane + 8u * CILQR_KD_ROW,
                                                      i * (CILQR_KD * 8), 0);
            } else {
                // over the Jacobians of this step: every lane loaded its coefficients at the top of the step
                double* kd = l.kd + CILQR_KD * i + lane;
                kd[0] = kc0;
                kd[CILQR_KD_ROW] = kc1;
            }
        }
        // expected cost reduction (cs:435-436): delta_V[0] += (0.5 d)^T Q_uu d on lane 36, delta_V[1] += d^T Q_u on lane 44
        int rpv = rp;
        __asm__("" : "+v"(rpv));
        dvacc = dvacc + ((rpv == 4) ? ta : tb);
    }
    dV[0] = lane_bcast<36>(dvacc);
    dV[1] = lane_bcast<44>(dvacc);
    wave_sync();
    return true;
}

template <bool DBG, int ROWD = 0>
__device__ inline bool backward_sweep(const Cst& c, const Lds& l, double lamb, int lane, double dV[2], int flags,
                                      int* fail_step = nullptr) {
    static_assert(!(DBG && ROWD != 0), "the wave-uniform twin reads the expansion from LDS");
    if (DBG && (flags & CILQR_DBG_UNIFORM_BACKWARD)) return backward_sweep_uniform(c, l, lamb, lane, dV, fail_step);
    return backward_sweep_lanes<ROWD>(c, l, lamb, lane, dV, fail_step);
}


// The line search's verdict on one trial (cs:356-371): 0 = go on, 1 = converged (first trial only),
// 2 = accepted.  A pure function of its arguments: the main and the helper wavefront both evaluate it on the
// same numbers and so agree on when a search ends without a second hand-shake.
__device__ inline int trial_verdict(double J_cur, double new_J, int t, double dV0, double dV1, double conv_thr,
                                    double accept_thr) {
    const double alpha = dm_pow2i(-t);
    const double decay = J_cur - new_J;
    const double adecay = (decay < 0) ? -decay : decay;
    if (t == 0 && adecay < conv_thr) return 1;
    const double approx = -(alpha * alpha * dV0 + alpha * dV1);
    if (decay > 0.0 && (approx < 0.0 || decay / approx > accept_thr)) return 2;
    return 0;
}

} // namespace cilqr
