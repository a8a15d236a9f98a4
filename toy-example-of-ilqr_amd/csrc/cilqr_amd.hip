// cilqr_amd.hip — kernels and C-ABI of libcilqr_amd.so (see include/cilqr_amd.h).
//
// One block = one wavefront = one trajectory.  The fused kernel runs the whole of
// CILQRSolver::solve (/root/reference/src/cilqr_solver.cpp:85-153) for its trajectory; the
// piecewise kernels expose the same device functions (cilqr_device.hpp) one at a time so that
// every stage can be compared with the oracle in isolation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cilqr_kernels.hpp"

// every build of k_solve is instantiated in cilqr_solve_inst.hip (one compilation per group, run in parallel)
#define CILQR_X_EXTERN(g, DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP) \
    extern template __global__ void k_solve<DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP> CILQR_SOLVE_SIGNATURE;
CILQR_SOLVE_VARIANTS(CILQR_X_EXTERN)
#undef CILQR_X_EXTERN
extern template __global__ void k_solve_grp<50, 2> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<30, 2> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<50, 2, true> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<30, 2, true> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2, true> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<100, 2, false, 2> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2, false, 2> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2, false, 1, true, true> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2, false, 2, true, true> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2, false, 4> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2, true, 4> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2, true, 2> CILQR_GRP_SIGNATURE;
extern template __global__ void k_solve_grp<0, 2, false, 4, true, true> CILQR_GRP_SIGNATURE;

// ------------------------------------------------------------------------------------------------
// piecewise kernels
// ------------------------------------------------------------------------------------------------

__device__ inline void stage_xu(const Lds& l, int N, const double* x, const double* u, int lane) {
    for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) l.x[e] = x[e];
    if (u)
        for (int e = lane; e < 2 * N; e += CILQR_WAVE) l.u[e] = u[e];
    wave_sync();
}

__global__ void __launch_bounds__(CILQR_WAVE)
k_init_traj(BatchArgs a, const double* __restrict__ x0, double* __restrict__ x_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N;
    Lds l; carve(l, g_lds, N, a.W, a.alm, 1);
    Cst c; load_cst(c, a, b, l, lane);
    const double xs[4] = {x0[4 * b], x0[4 * b + 1], x0[4 * b + 2], x0[4 * b + 3]};
    int idx0;
    init_trajectory(c, l, xs, nullptr, lane, idx0, a.W);
    for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) x_out[(size_t)b * 4 * (N + 1) + e] = l.x[e];
}

__global__ void __launch_bounds__(CILQR_WAVE)
k_ref_points(BatchArgs a, const double* __restrict__ x, double* __restrict__ ref_out,
             int32_t* __restrict__ idx_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N;
    Lds l; carve(l, g_lds, N, a.W, a.alm, 1);
    Cst c; load_cst(c, a, b, l, lane);
    stage_xu(l, N, x + (size_t)b * 4 * (N + 1), nullptr, lane);
    int idx0;
    ref_indices_lds(c, l, lane, idx0, a.W);
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        int j = l.ridx[k];
        if (idx_out) idx_out[(size_t)b * (N + 1) + k] = j;
        double* r = ref_out + ((size_t)b * (N + 1) + k) * 3;
        r[0] = c.lane_xy[2 * j]; r[1] = c.lane_xy[2 * j + 1]; r[2] = c.lane_aux[(size_t)j * CILQR_AUX_STRIDE];
    }
}

template <bool ALM>
__global__ void __launch_bounds__(CILQR_WAVE)
k_total_cost(BatchArgs a, const double* __restrict__ u, const double* __restrict__ x,
             double* __restrict__ J_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N;
    Lds l; carve(l, g_lds, N, a.W, a.alm, 1);
    Cst c; load_cst(c, a, b, l, lane);
    stage_xu(l, N, x + (size_t)b * 4 * (N + 1), u + (size_t)b * 2 * N, lane);
    int idx0;
    ref_indices_lds(c, l, lane, idx0, a.W);
    AlmSt al = load_alm(a, b, N);
    double J = total_cost_lds<ALM>(c, l, al, lane);
    if (lane == 0) J_out[b] = J;
}

template <bool ALM>
__global__ void __launch_bounds__(CILQR_WAVE)
k_forward_pass(BatchArgs a, const double* __restrict__ u, const double* __restrict__ x,
               const double* __restrict__ d, const double* __restrict__ K, int n_alpha,
               double* __restrict__ new_u, double* __restrict__ new_x, double* __restrict__ J_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N;
    const int R = N + 1;
    Lds l; carve(l, g_lds, N, a.W, a.alm, 1);
    Cst c; load_cst(c, a, b, l, lane);
    stage_xu(l, N, x + (size_t)b * 4 * R, u + (size_t)b * 2 * N, lane);
    for (int e = lane; e < 8 * N; e += CILQR_WAVE) l.kd[CILQR_KD * (e >> 3) + CILQR_KD_K(e & 7)] = K[(size_t)b * 8 * N + e];
    for (int e = lane; e < 2 * N; e += CILQR_WAVE) l.kd[CILQR_KD * (e >> 1) + CILQR_KD_D(e & 1)] = d[(size_t)b * 2 * N + e];
    __syncthreads();
    int idx0;
    ref_indices_lds(c, l, lane, idx0, a.W); // indices of the current trajectory: the guesses for its trials
    seed_trial_indices(l, N, 1, lane);
    double* scr = a.scratch + (size_t)b * scratch_doubles(N);
    AlmSt al = load_alm(a, b, N);
    rollout_trials(c, l, scr, lane, n_alpha);
    for (int t = 0; t < n_alpha; ++t) {
        const double* tr = TRIAL_AT(scr, t);
        for (int k = lane; k <= N; k += CILQR_WAVE) {
            double* xo = new_x + (((size_t)b * n_alpha + t) * R + k) * 4;
            xo[0] = TR(tr, 0, k); xo[1] = TR(tr, 1, k); xo[2] = TR(tr, 2, k); xo[3] = TR(tr, 3, k);
            if (k < N) {
                double* uo = new_u + (((size_t)b * n_alpha + t) * N + k) * 2;
                uo[0] = TR(tr, 4, k); uo[1] = TR(tr, 5, k);
            }
        }
        int nfb = 0;
        double J = total_cost_trial<true, 2, ALM>(c, l, al, scr, t, lane, idx0, a.flags, &nfb);
        if (lane == 0 && J_out) J_out[(size_t)b * n_alpha + t] = J;
    }
}

template <bool ALM>
__global__ void __launch_bounds__(CILQR_WAVE)
k_cost_derivatives(BatchArgs a, const double* __restrict__ u, const double* __restrict__ x,
                   double* __restrict__ o_lx, double* __restrict__ o_lu, double* __restrict__ o_lxx,
                   double* __restrict__ o_luu, double* __restrict__ o_A, double* __restrict__ o_B) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N;
    const int R = N + 1;
    Lds l; carve(l, g_lds, N, a.W, a.alm, 1);
    Cst c; load_cst(c, a, b, l, lane);
    stage_xu(l, N, x + (size_t)b * 4 * R, u + (size_t)b * 2 * N, lane);
    int idx0;
    ref_indices_lds(c, l, lane, idx0, a.W);
    AlmSt al = load_alm(a, b, N);
    cost_and_model_derivatives<ALM>(c, l, al, lane);
    for (int k = lane; k <= N; k += CILQR_WAVE) {
        if (o_lx) for (int e = 0; e < 4; ++e) o_lx[((size_t)b * R + k) * 4 + e] = l.lx[4 * k + e];
        if (o_lxx) {
            if (ALM) {
                for (int e = 0; e < 16; ++e) o_lxx[((size_t)b * R + k) * 16 + e] = l.lxx[16 * k + e];
            } else {
                const double* hx = l.lxx + 7 * k;
                const double dn[16] = {hx[0], hx[1], 0.0, hx[2], hx[1], hx[3], 0.0, hx[4],
                                       0.0, 0.0, hx[6], 0.0, hx[2], hx[4], 0.0, hx[5]};
                for (int e = 0; e < 16; ++e) o_lxx[((size_t)b * R + k) * 16 + e] = dn[e];
            }
        }
        if (k < N) {
            if (o_lu) for (int e = 0; e < 2; ++e) o_lu[((size_t)b * N + k) * 2 + e] = l.lu[2 * k + e];
            if (o_luu) {
                double* q = o_luu + ((size_t)b * N + k) * 4;
                q[0] = l.luu[2 * k]; q[1] = 0.0; q[2] = 0.0; q[3] = l.luu[2 * k + 1];
            }
            if (o_A) {
                const double* A = l.kd + CILQR_KD * k;
                const double dn[16] = {1, 0, A[0], A[1], 0, 1, A[2], A[3], 0, 0, 1, 0, 0, 0, A[4], 1};
                for (int e = 0; e < 16; ++e) o_A[((size_t)b * N + k) * 16 + e] = dn[e];
            }
            if (o_B) {
                const double* Bq = l.kd + CILQR_KD * k + CILQR_KD_B;
                const double dn[8] = {0, Bq[0], 0, Bq[1], c.dt, 0, 0, Bq[2]};
                for (int e = 0; e < 8; ++e) o_B[((size_t)b * N + k) * 8 + e] = dn[e];
            }
        }
    }
}

template <bool ALM>
__global__ void __launch_bounds__(CILQR_WAVE)
k_backward_pass(BatchArgs a, const double* __restrict__ u, const double* __restrict__ x,
                const double* __restrict__ lamb, double* __restrict__ o_d, double* __restrict__ o_K,
                double* __restrict__ o_dV, int32_t* __restrict__ o_status) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N;
    const int R = N + 1;
    Lds l; carve(l, g_lds, N, a.W, a.alm, 1);
    Cst c; load_cst(c, a, b, l, lane);
    stage_xu(l, N, x + (size_t)b * 4 * R, u + (size_t)b * 2 * N, lane);
    int idx0;
    ref_indices_lds(c, l, lane, idx0, a.W);
    AlmSt al = load_alm(a, b, N);
    cost_and_model_derivatives<ALM>(c, l, al, lane);
    double dV[2];
    int fail_step = -1; // gains exist for the steps after it; d, K start as zeros upstream (cs:392-393)
    bool ok = backward_sweep<!ALM>(c, l, lamb[b], lane, dV, a.flags, &fail_step);
    __syncthreads();
    for (int e = lane; e < 8 * N; e += CILQR_WAVE)
        o_K[(size_t)b * 8 * N + e] = ((e >> 3) > fail_step) ? l.kd[CILQR_KD * (e >> 3) + CILQR_KD_K(e & 7)] : 0.0;
    for (int e = lane; e < 2 * N; e += CILQR_WAVE)
        o_d[(size_t)b * 2 * N + e] = ((e >> 1) > fail_step) ? l.kd[CILQR_KD * (e >> 1) + CILQR_KD_D(e & 1)] : 0.0;
    if (lane == 0) {
        o_dV[2 * b] = dV[0];
        o_dV[2 * b + 1] = dV[1];
        o_status[b] = ok ? CILQR_RUNNING : CILQR_BACKWARD_PASS_FAIL;
    }
}

// fills sin/cos of the lane yaw and of the obstacle yaw with the device's own dm_sincos
__global__ void k_prepare_tables(const double* __restrict__ yaw_in, double* __restrict__ lane_aux, int L,
                                 const double* __restrict__ obs_in, double* __restrict__ obs_out, int MT) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) {
        double y = yaw_in[i], sn, cs;
        dm_sincos(y, &sn, &cs);
        double* o = lane_aux + (size_t)i * CILQR_AUX_STRIDE;
        o[0] = y; o[1] = sn; o[2] = cs; o[3] = 0.0;
    }
    if (i < MT) {
        const double* r = obs_in + (size_t)i * 3;
        double sn, cs;
        dm_sincos(r[2], &sn, &cs);
        double* o = obs_out + (size_t)i * CILQR_OBS_STRIDE;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = sn; o[4] = cs;
    }
}

// the step after the path, for a batch (mp:181,197): ego_state = new_x.row(1); the obstacle window moves one tick on
__global__ void k_advance(int B, int N, const double* __restrict__ x, double* __restrict__ x0, int32_t* __restrict__ tick) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const double* r = x + ((size_t)i * (N + 1) + 1) * 4;
    double* o = x0 + (size_t)i * 4;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
    if (tick) tick[i] += 1;
}

__global__ void k_detmath(int f, const double* __restrict__ x, const double* __restrict__ y,
                          double* __restrict__ o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i], b = y ? y[i] : 0.0, r = 0.0;
    switch (f) {
        case 0: r = dm_exp(a); break;
        case 1: r = dm_sin(a); break;
        case 2: r = dm_cos(a); break;
        case 3: r = dm_tan(a); break;
        case 4: r = dm_atan(a); break;
        case 5: r = dm_hypot(a, b); break;
        case 6: r = a / b; break;
        case 7: r = dm_sqrt(a < 0 ? -a : a); break;
        case 8: { DmPinned pk; dm_pin_load(pk); r = dm_sin<1>(a, &pk); break; }  // coefficients pinned to vector registers
        case 9: { DmPinned pk; dm_pin_load(pk); r = dm_cos<1>(a, &pk); break; }
        case 10: { DmPinned pk; dm_pin_load(pk); r = dm_tan<1>(a, &pk); break; }
        default: break;
    }
    o[i] = r;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(CILQR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// What belongs to ONE launch in flight: the trial slabs, the control words (trajectory counter of the persistent blocks,
// work-sharing counters and slots), the parked-solve state, the timing events.  A handle has one such set by default —
// its launches are then ordered, one at a time — and up to CILQR_MAX_IN_FLIGHT of them after
// cilqr_set_batches_in_flight(): independent batches then overlap on the device, the tables shared (round 5).
#define CILQR_MAX_IN_FLIGHT 4
struct MemRange { const char* p; size_t n; bool w; };
struct LaunchSlot {
    hipStream_t stream = nullptr;       // internal stream of the slot (in-flight mode only; null: the caller's stream carries the launch)
    hipEvent_t done = nullptr;          // recorded behind the slot's last launch (or behind anything else that used its buffers)
    hipStream_t last_stream = nullptr;  // the stream `done` was recorded on
    bool launched = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed_pending = false;
    float last_ms = 0.f;
    DevBuf scratch;
    DevBuf park, rq;
    int park_B = 0, park_N = 0;
    DevBuf sh_ctl, sh_req, sh_hints;
    int sh_B = 0, sh_N = 0;
    std::vector<MemRange> ranges;       // the caller's buffers the slot's last solve reads / writes (in-flight mode: hazards)
};
#define SL(h) ((h)->slot[(h)->cur])

struct cilqr_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    LaunchSlot slot[CILQR_MAX_IN_FLIGHT];
    int cur = 0;        // the slot the call in progress uses / the last launch used
    int in_flight = 1;  // cilqr_set_batches_in_flight
    int next_slot = 0;
    bool force_seq = false; // the call in progress is a host-buffer entry point: slot 0 on the handle's stream, ordered
    hipEvent_t ev_in = nullptr; // in-flight mode: marks the caller's stream at the call (its earlier work = the launch's inputs)
    bool timing = false;
    // tables
    std::vector<cilqr_params> params;
    DevBuf d_params;
    std::vector<DevScene> scenes;     // device pointers inside
    std::vector<int> scene_T, scene_M; // for validation
    std::vector<double> scene_spacing, scene_velo;
    std::vector<void*> scene_allocs;
    DevBuf d_scenes;
    // scratch + staging
    int win = 0;      // LDS lane-window capacity in samples (derived from the tables)
    int win_occ = 0;  // the same when two wavefronts per SIMD are wanted (large batches), one stage-cost slot
    int win_occ2 = 0; // ... with two stage-cost slots (paired passes)
    DevBuf alm_mu, alm_mu_next, alm_rho; // ALM solve type: per-trajectory multipliers carried across calls
    int alm_B = 0, alm_C = 0, alm_N = 0; // rows, columns and horizon the multiplier arrays were laid out for
    // cilqr_solve (one ego per call, the reference's own call shape): host copy of the tables that are in HBM,
    // so that a tick whose arguments are unchanged — or whose obstacle predictions are the tail of the routes
    // uploaded earlier — re-uses them; one pinned staging block and one device block for the call's buffers
    struct {
        bool valid = false;
        std::vector<double> lane_x, lane_y, lane_yaw, obs;
        int M = 0, T = 0, last_d = 0;
        double borders[2] = {0, 0}, ref_velo = 0;
        void* pinned = nullptr;
        void* dev = nullptr;
        size_t cap = 0;
        long long uploads = 0, reuses = 0;
    } one;
    DevBuf prof;      // [B][8] int64, filled when profiling is on
    bool profiling = false;
    int debug_flags = 0;
    int helper_mode = -1;      // -1 auto (by batch size), 0 never, 1 always
    int rollout_mode = -1;     // -1 adaptive, 0 all step sizes in one pass, 1 first trial alone first (BatchArgs::tier)
    // Largest batch that gets helper wavefronts.  Up to 1024 trajectories a lone wavefront per trajectory leaves
    // SIMD slots empty.  Beyond that the blocks take a second round, which pays while the batch's stragglers
    // dominate: measured +20 % at 1536 straight-lane trajectories, +3 % / -13 % at 2048 (straight / bend), and
    // +45 % at 2048 with two rows per lane (N = 100), whose lone-wavefront kernel is the slower one
    int helper_max_batch = 1536;
    int helper_max_batch_two_rows = -1; // horizons above 63 (two rows per lane): -1 = by horizon and solve type, see
                                        // wants_helper()
    int occ_floor_pct = 0;    // smallest lane window the occupancy-driven choice accepts, in % of the horizon's reach (never
                              // below 64 samples).  Round 2: occupancy beats the window — horizon 100 went from 4 blocks per
                              // CU with a 912-sample window to 6 with 64 samples: +18 %; horizon 50 fits 8 blocks either way
    int persistent_blocks = 1; // large batches: as many blocks as fit on the chip pull trajectories (0: one block each)
    int num_cus = 256;
    int share_backoff = 4;
    int share_max_helpers = 64, share_min_t0 = 1; // (measured: 64 helpers serve the few open searches of a launch's tail; 2048 polling blocks cost 9-16 %)
    bool timeline = false;     // record a block timeline with the next solves (development aid)
    DevBuf tl;
    int tl_B = 0;
    bool last_launch_shared = false;
    // Hand-over failures (a bounded wait of a launch's hand-over protocol that expired) are LATCHED: one device word the launch's
    // own stream ORs the launch's SH_ERROR into right behind the kernel (k_latch_launch) — it survives the slot's next launch,
    // which zeroes the control words, and covers every slot.  latch = { mask of slots with an error, launches with an error }
    DevBuf latch;
    bool latch_pending = false; // a launch that could have set it has been enqueued since the last look
    int grp_wait_spins = 0;     // development library: forced bound of the grouped build's hand-over wait (0 = the real one)
    int last_info[4] = {0, 0, 0, 0}; // cilqr_last_launch_info
    bool last_launch_reset_ctl = false; // the last fused launch zeroed the control words (persistent blocks): its counters are its own
    int share = 1;             // finished blocks help running ones with their line searches (k_solve's SHARE): 1 on, 0 off
    // resumable solves (k_solve's RES: the two-row builds in persistent launches): iterations per slice, 0 = off.  A
    // launch whose batch fits the chip at once (no second round of trajectories) has nothing to reorder and runs whole.
    int resume_iters = -1;     // -1 = automatic: 32 (k_solve), group_slice / group_slice_long (the grouped build's sliced solves)
    unsigned last_parked = 0;
    bool fused_call = false;   // the call in progress is a fused solve (not a piecewise entry point)
    bool looping = false;      // the call in progress is a closed loop in one launch: the plain builds (see the dispatch)
    int global_expansion = -1; // cost expansion in global memory (k_solve's LG): -1 = for horizons above 63 in batches of the
                               // two-wavefronts-per-SIMD range (barrier mode), 0 = never, 1 = wherever a build exists
    int win_lg = 0;            // lane window of those builds
                               // (round 5: every batch beyond the helper range runs the two-wavefronts-per-SIMD register build,
                               //  one trial per pass: the one-per-SIMD lone builds could only be reached through tuning switches
                               //  and are gone from the library)
    int prof_two_per_simd = 0; // development library: cycle accounting in the two-per-SIMD headline build (CILQR_TUNE=prof2=1)
    int group_mode = -1;       // trajectories per wavefront in the large-batch launches of horizons up to 63, barrier mode
                               // (k_solve_grp): -1 = 2 where that build applies, 0 / 1 = never (k_solve), 2 = wherever it can run
    int win_grp = 0;           // lane window of those launches
    int group_loop = 1;        // the closed loop in one launch runs the grouped build too (0: k_solve's LOOP builds)
    int group_pair_sweep = 1;  // ... the sweeps of a wavefront's two trajectories in one instruction stream (CILQR_TUNE=pair_sweep=0: round 4's turn)
    int poison_scratch = 0;    // development library: fill the kernels' scratch before every launch (CILQR_TUNE=poison=1: NaN
                               // patterns, 2: zeros) — results must not depend on what the scratch held
    int group_pair_costs = 1;  // ... line-search trials after the first costed two per pass
    int group_alm = 0;         // ... augmented-Lagrangian batches in pairs (the long layout at every horizon).  Round 6: the kernels were
                               // written while the GPU pool was closed and are bit-exact under the wave64 emulator (tests/test_emulator.py)
                               // but have never run on a GPU: 0 = only when pairs are asked for explicitly (cilqr_set_group_mode(2)),
                               // the default dispatch stays on k_solve's GPU-proven builds; 1 (CILQR_TUNE=group_alm=1) = by default
    int group_long = 1;        // ... horizons of 64 ... 127 run the grouped build too (its long layout; CILQR_TUNE=group_long=0: k_solve's
                               // two-rows-per-lane builds)
    int group_slice = 16;      // ... solves run this many iterations at a time while other trajectories wait (0: to their end in one go)
    int group_slice_long = 12; // ... the same for the long layout
    int group_slice_window_pct = 200; // ... hand-overs at the end of a slice begin when fewer fresh trajectories are left than this share of the resident slots
    int group_steal = 1;       // ... idle wavefronts take over trajectories of wavefronts that still hold two (the launch's tail)
    int prof_B = 0;
    DevBuf st[16];
    // resident blocks per CU of each persistent build (asked once per kernel and LDS size, not on every launch)
    struct Occ { const void* kern; size_t shm; int per_cu; };
    std::vector<Occ> occ;
};

// Persistent launches use one scratch area per resident block: never more than this many per CU, whatever the
// occupancy query says (8 = two wavefronts per SIMD; ensure_scratch sizes the areas with the same number)
#define CILQR_MAX_BLOCKS_PER_CU 8
#define CILQR_MAX_BLOCKS_PER_CU_G1 8
#define CILQR_GROUP_MAX 2 /* trajectories per wavefront of the grouped builds (k_solve_grp).  Three per wavefront were built and
                             measured in round 4 (-6 %: profiles/r04_experiments/three_trajectories_per_wavefront_ab.txt) and are no
                             longer instantiated; the templates still take G */
static int grp_n(const cilqr_handle* h);
static int blocks_per_cu(cilqr_handle* h, const void* kern, size_t shm, int* out) {
    for (const auto& e : h->occ)
        if (e.kern == kern && e.shm == shm) { *out = e.per_cu; return CILQR_OK; }
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, CILQR_WAVE, shm));
    per_cu = per_cu < 1 ? 1 : (per_cu > CILQR_MAX_BLOCKS_PER_CU_G1 ? CILQR_MAX_BLOCKS_PER_CU_G1 : per_cu);
    h->occ.push_back({kern, shm, per_cu});
    *out = per_cu;
    return CILQR_OK;
}

// LDS lane window: enough samples for the horizon at 1.5x the target speed (anything beyond it is
// still correct — lookups outside the window read global memory), a multiple of 64, <= 1024.
static void update_window(cilqr_handle* h) {
    if (h->params.empty() || h->scene_spacing.empty()) return;
    // samples the horizon covers at the target speed
    double base = 32;
    for (const auto& p : h->params)
        for (size_t i = 0; i < h->scene_spacing.size(); ++i) {
            double ds = h->scene_spacing[i] > 1e-6 ? h->scene_spacing[i] : 0.1;
            double n = p.N * p.dt * h->scene_velo[i] / ds;
            if (n > base) base = n;
        }
    const int want = std::min(1024, ((int)(base * 1.5 + 64) + 63) / 64 * 64); // generous
    const int floor_ok = std::min(want, ((int)(base * 1.2 + 32) + 63) / 64 * 64); // still comfortable
    // prefer a window that lets one more block fit into the CU's 160 KiB of LDS, as long as it stays
    // comfortable (anything outside the window is still read correctly, from global memory)
    const int N = h->params[0].N;
    const int alm = h->params[0].solve_type == 1 ? 1 : 0;
    size_t fixed = lds_bytes(N, 0, alm, 2);
    auto pick = [&](int floor_w) {
        for (int k = 8; k >= 1; --k) {
            const long budget = (long)(163840 / k) - (long)fixed;
            if (budget <= 0) continue;
            const int wk = (int)(budget / 16) / 8 * 8;
            if (wk >= floor_w) return std::min(want, wk);
        }
        return want;
    };
    h->win = pick(floor_ok);
    // batches that fill the chip several times over: occupancy (two wavefronts per SIMD hide each other's
    // latencies) is worth more than the far end of the window, which only the last rows at full speed reach;
    // their kernels cost one trial at a time (one stage-cost slot)
    const int occ_floor = std::min(floor_ok, std::max(64, ((int)(base * (h->occ_floor_pct / 100.0) + 16) + 7) / 8 * 8));
    h->win_occ2 = pick(occ_floor);
    fixed = lds_bytes(N, 0, alm, 1);
    h->win_occ = pick(occ_floor);
    fixed = lds_bytes(N, 0, alm, 1, 1);
    h->win_lg = pick(occ_floor);
    {
        // the grouped builds: the window shares the expansion's area (cilqr_group.hpp), so it is free up to that size and
        // otherwise bounded by 8 blocks per CU
        const long room = (long)(163840 / (grp_n(h) == 1 ? CILQR_MAX_BLOCKS_PER_CU_G1 : CILQR_MAX_BLOCKS_PER_CU)) - (long)grp_lds_bytes(N, 0, grp_n(h)) +
                          (long)sizeof(double) * grp_expansion_doubles(N);
        int wg = (int)(room / 16) / 8 * 8;
        wg = std::max(wg, grp_expansion_doubles(N) / 2 / 8 * 8);
        h->win_grp = std::max(8, std::min(want, wg));
        if (N + 1 > CILQR_WAVE || alm) {
            // the long layout (two rows per lane, or the augmented Lagrangian at any horizon): the window sits behind the stage-cost scratch; the largest that keeps 8
            // wavefronts on a CU, as long as it is comfortable — else 7, 6, ...
            fixed = grpl_lds_bytes(N, 0, grp_n(h)) - sizeof(double) * (size_t)grpl_shared_doubles(N, 0, grp_n(h)) +
                    sizeof(double) * (size_t)grpl_cs_doubles(N);
            h->win_grp = std::max(8, pick(occ_floor));
        }
    }
}

// One launch per SLOT at a time (scratch areas, control words and parked-solve state belong to the launch in flight; a handle
// has one slot unless cilqr_set_batches_in_flight() gave it more): work enqueued on another stream than the slot's previous
// launch first waits for it, on the device.
// Every result of a batch marked "not solved" before a launch that hands trajectories from wavefront to wavefront: whoever
// finishes a trajectory overwrites its record; one that is lost in transit keeps the mark (include/cilqr_amd.h, CILQR_END_NOT_SOLVED)
__global__ void k_mark_unsolved(cilqr_result* __restrict__ res, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    cilqr_result r;
    r.J_init = __builtin_nan("");
    r.J_final = __builtin_nan("");
    r.iters = 0;
    r.end_reason = CILQR_END_NOT_SOLVED;
    r.final_status = CILQR_RUNNING;
    r.ls_trials = 0;
    r.cost_evals = 0;
    r.trace_len = 0;
    res[b] = r;
}
// ... and the launch's error word ORed into the handle's latch, on the launch's own stream right behind the kernel
__global__ void k_latch_launch(const unsigned* __restrict__ ctl, unsigned* __restrict__ latch, int slot) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (__hip_atomic_load(ctl + SH_ERROR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        (void)__hip_atomic_fetch_or(latch, 1u << slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_add(latch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
static int mark_unsolved(cilqr_result* d_res, int B, hipStream_t s) {
    if (!d_res) return CILQR_OK;
    hipLaunchKernelGGL(k_mark_unsolved, dim3((B + 255) / 256), dim3(256), 0, s, d_res, B);
    HIP_TRY(hipGetLastError());
    return CILQR_OK;
}
static int latch_launch(cilqr_handle* h, hipStream_t s);
static int report_latch(cilqr_handle* h, bool clear, unsigned* mask_out);

static int order_after_slot(cilqr_handle* h, int k, hipStream_t s) {
    LaunchSlot& sl = h->slot[k];
    if (sl.launched && sl.last_stream != s) HIP_TRY(hipStreamWaitEvent(s, sl.done, 0));
    return CILQR_OK;
}
// ... and after EVERY launch of the handle in flight: called before any asynchronous work that touches handle-wide buffers
// (staging, timeline, multipliers) or the caller's arrays outside the solve entry points' own hazard tracking.
static int order_after_last_launch(cilqr_handle* h, hipStream_t s) {
    for (int k = 0; k < CILQR_MAX_IN_FLIGHT; ++k) {
        int rc = order_after_slot(h, k, s);
        if (rc) return rc;
    }
    return CILQR_OK;
}
// Host-side wait for the handle's own launches (before its arrays are replaced): the other handles of the process and
// their streams are left alone — a hipDeviceSynchronize() here stalled every batch in flight.
static int wait_slot(cilqr_handle* h, int k) {
    if (h->slot[k].launched) HIP_TRY(hipEventSynchronize(h->slot[k].done));
    return CILQR_OK;
}
static int wait_last_launch(cilqr_handle* h) {
    for (int k = 0; k < CILQR_MAX_IN_FLIGHT; ++k) {
        int rc = wait_slot(h, k);
        if (rc) return rc;
    }
    return CILQR_OK;
}
// `s` has just been given work that used slot k's buffers (or, slot 0, the handle-wide ones)
static int mark_slot(cilqr_handle* h, int k, hipStream_t s) {
    LaunchSlot& sl = h->slot[k];
    HIP_TRY(hipEventRecord(sl.done, s));
    sl.last_stream = s;
    sl.launched = true;
    return CILQR_OK;
}

static int grp_n(const cilqr_handle*) { return 2; }

static int check_ready(cilqr_handle* h) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
    if (h->params.empty()) return fail(CILQR_ERR_BAD_ARG, "cilqr_set_params has not been called");
    if (h->scenes.empty()) return fail(CILQR_ERR_BAD_ARG, "cilqr_set_scenarios has not been called");
    return CILQR_OK;
}

static int latch_launch(cilqr_handle* h, hipStream_t s) {
    if (!h->latch.p) {
        if (h->latch.ensure(sizeof(unsigned) * 4)) return fail(CILQR_ERR_DEVICE, "hipMalloc error latch");
        HIP_TRY(hipMemset(h->latch.p, 0, sizeof(unsigned) * 4));
    }
    hipLaunchKernelGGL(k_latch_launch, dim3(1), dim3(64), 0, s, static_cast<const unsigned*>(SL(h).sh_ctl.p),
                       static_cast<unsigned*>(h->latch.p), h->cur);
    HIP_TRY(hipGetLastError());
    h->latch_pending = true;
    return CILQR_OK;
}
// The latch as the host sees it once every launch of the handle has completed (callers wait first).  clear: the report is the
// one the caller acts on (cilqr_wait, the host-buffer entry points) — the next one starts from zero.
static int report_latch(cilqr_handle* h, bool clear, unsigned* mask_out) {
    *mask_out = 0;
    if (!h->latch.p || !h->latch_pending) return CILQR_OK;
    unsigned w[2] = {0, 0};
    HIP_TRY(hipMemcpy(w, h->latch.p, sizeof(w), hipMemcpyDeviceToHost));
    *mask_out = w[0];
    if (clear) {
        if (w[0]) HIP_TRY(hipMemset(h->latch.p, 0, sizeof(unsigned) * 4));
        h->latch_pending = false;
    }
    return CILQR_OK;
}
static int fail_if_latched(cilqr_handle* h) {
    unsigned mask = 0;
    int rc = report_latch(h, true, &mask);
    if (rc) return rc;
    if (mask) {
        char buf[256];
        std::snprintf(buf, sizeof(buf), "a bounded wait inside a launch expired (launch slots 0x%x: work sharing between blocks / trajectories "
                      "handed from wavefront to wavefront); a trajectory that was in transit has end_reason CILQR_END_NOT_SOLVED", mask);
        return fail(CILQR_ERR_DEVICE, buf);
    }
    return CILQR_OK;
}

extern "C" const char* cilqr_last_error(void) { return g_err.c_str(); }
#ifdef CILQR_COMPILER_VALIDATED
#define CILQR_VERSION_TAIL ")"
#else
#define CILQR_VERSION_TAIL "; compiler NOT the validated one: one trajectory per wavefront by default)"
#endif
#ifdef CILQR_DEV_BUILD
extern "C" const char* cilqr_version(void) { return "cilqr_amd 0.5-dev (gfx950, wave64, fp64; testing aids + cycle accounting" CILQR_VERSION_TAIL; }
#else
extern "C" const char* cilqr_version(void) { return "cilqr_amd 0.5 (gfx950, wave64, fp64" CILQR_VERSION_TAIL; }
#endif

extern "C" int cilqr_device_count(int32_t* n) {
    if (!n) return fail(CILQR_ERR_BAD_ARG, "null argument");
    int c = 0;
    const hipError_t e = hipGetDeviceCount(&c);
    *n = (e == hipSuccess && c > 0) ? c : 0;
    return CILQR_OK;
}

extern "C" int cilqr_create(int device, cilqr_handle** out) {
    if (!out) return fail(CILQR_ERR_BAD_ARG, "out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(CILQR_ERR_NO_DEVICE, "no HIP device visible (there is no CPU fallback)");
    if (device < 0 || device >= n) return fail(CILQR_ERR_BAD_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    cilqr_handle* h = new cilqr_handle();
    h->device = device;
#ifdef CILQR_DEV_BUILD
    // tuning experiments only (A/B runs on one box, development library): CILQR_TUNE="helper_max_batch=1536,share=0,..."
    // The production library does not read the environment: a variable must not change launch shapes in a deployed planner.
    if (const char* t = std::getenv("CILQR_TUNE")) {
        std::string sv(t);
        size_t pos = 0;
        while (pos < sv.size()) {
            size_t e = sv.find(',', pos);
            if (e == std::string::npos) e = sv.size();
            const std::string kv = sv.substr(pos, e - pos);
            const size_t eq = kv.find('=');
            bool known = false;
            if (eq != std::string::npos) {
                const std::string k = kv.substr(0, eq);
                const int v = std::atoi(kv.c_str() + eq + 1);
                known = true;
                if (k == "helper_max_batch") h->helper_max_batch = v;
                else if (k == "helper_max_batch_two_rows") h->helper_max_batch_two_rows = v;
                else if (k == "occ_floor_pct") h->occ_floor_pct = v;
                else if (k == "global_expansion") h->global_expansion = v;
                else if (k == "share") h->share = v;
                else if (k == "share_max_helpers") h->share_max_helpers = v;
                else if (k == "share_min_t0") h->share_min_t0 = v;
                else if (k == "share_backoff") h->share_backoff = v;
                else if (k == "persistent_blocks") h->persistent_blocks = v;
                else if (k == "resume_iters") h->resume_iters = v;
                else if (k == "group") h->group_mode = v;
                else if (k == "prof2") h->prof_two_per_simd = v;
                else if (k == "group_steal") h->group_steal = v;
                else if (k == "group_pair_costs") h->group_pair_costs = v;
                else if (k == "pair_sweep") h->group_pair_sweep = v;
                else if (k == "group_long") h->group_long = v;
                else if (k == "group_alm") h->group_alm = v;
                else if (k == "group_slice") h->group_slice = v;
                else if (k == "group_slice_long") h->group_slice_long = v;
                else if (k == "group_slice_window") h->group_slice_window_pct = v;
                else if (k == "group_loop") h->group_loop = v;
                else if (k == "poison") h->poison_scratch = v;
                else if (k == "grp_wait_spins") h->grp_wait_spins = v;
                else known = false;
            }
            if (!known && !kv.empty()) std::fprintf(stderr, "cilqr_amd: CILQR_TUNE: unknown setting '%s' ignored\n", kv.c_str());
            pos = e + 1;
        }
    }
#endif
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
    }
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto& sl : h->slot) {
        HIP_TRY(hipEventCreate(&sl.ev0));
        HIP_TRY(hipEventCreate(&sl.ev1));
        HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    *out = h;
    return CILQR_OK;
}

static void free_scenes(cilqr_handle* h) {
    for (void* p : h->scene_allocs) (void)hipFree(p);
    h->scene_allocs.clear();
    h->scenes.clear();
    h->scene_T.clear();
    h->scene_M.clear();
    h->scene_spacing.clear();
    h->scene_velo.clear();
}

extern "C" int cilqr_destroy(cilqr_handle* h) {
    if (!h) return CILQR_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)wait_last_launch(h);
    for (auto& sl : h->slot)
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    free_scenes(h);
    h->d_params.release();
    h->d_scenes.release();
    for (auto& sl : h->slot) {
        sl.scratch.release();
        sl.sh_ctl.release(); sl.sh_req.release(); sl.sh_hints.release();
        sl.park.release(); sl.rq.release();
        if (sl.ev0) (void)hipEventDestroy(sl.ev0);
        if (sl.ev1) (void)hipEventDestroy(sl.ev1);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    h->tl.release();
    h->latch.release();
    h->alm_mu.release();
    h->alm_mu_next.release();
    h->alm_rho.release();
    h->prof.release();
    for (auto& s : h->st) s.release();
    if (h->one.pinned) (void)hipHostFree(h->one.pinned);
    if (h->one.dev) (void)hipFree(h->one.dev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return CILQR_OK;
}

extern "C" int cilqr_set_timing(cilqr_handle* h, int32_t enabled) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
    h->timing = enabled != 0;
    return CILQR_OK;
}

extern "C" int cilqr_set_phase_profiling(cilqr_handle* h, int32_t enabled) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
#ifndef CILQR_DEV_BUILD
    if (enabled) return fail(CILQR_ERR_UNSUPPORTED, "cycle accounting is built into libcilqr_amd_dev.so only");
#endif
    h->profiling = enabled != 0;
    return CILQR_OK;
}

extern "C" int cilqr_set_block_timeline(cilqr_handle* h, int32_t enabled) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
    h->timeline = enabled != 0;
    return CILQR_OK;
}

extern "C" int cilqr_get_block_timeline(cilqr_handle* h, int64_t* out, int32_t B) {
    if (!h || !out) return fail(CILQR_ERR_BAD_ARG, "null argument");
    if (!h->tl.p || B < 1 || B > h->tl_B) return fail(CILQR_ERR_BAD_ARG, "no timeline recorded for that batch");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->tl.p, sizeof(long long) * 4 * (size_t)B, hipMemcpyDeviceToHost));
    return CILQR_OK;
}

static int ensure_alm(cilqr_handle* h, int B);

extern "C" int cilqr_set_alm_state(cilqr_handle* h, int32_t B, const double* mu, const double* rho) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->params[0].solve_type != 1 || B < 1) return fail(CILQR_ERR_BAD_ARG, "handle is not in alm mode");
    HIP_TRY(hipSetDevice(h->device));
    rc = ensure_alm(h, B);
    if (rc) return rc;
    const size_t nb = sizeof(double) * (size_t)B * h->params[0].N * h->alm_C;
    if (mu) HIP_TRY(hipMemcpy(h->alm_mu.p, mu, nb, hipMemcpyHostToDevice));
    if (rho) HIP_TRY(hipMemcpy(h->alm_rho.p, rho, sizeof(double) * B, hipMemcpyHostToDevice));
    return CILQR_OK;
}

extern "C" int cilqr_get_alm_state(cilqr_handle* h, int32_t B, double* mu, double* mu_next, double* rho,
                                   int32_t* cols) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->params[0].solve_type != 1 || B < 1 || B > h->alm_B) return fail(CILQR_ERR_BAD_ARG, "no alm state for that batch");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t nb = sizeof(double) * (size_t)B * h->params[0].N * h->alm_C;
    if (mu) HIP_TRY(hipMemcpy(mu, h->alm_mu.p, nb, hipMemcpyDeviceToHost));
    if (mu_next) HIP_TRY(hipMemcpy(mu_next, h->alm_mu_next.p, nb, hipMemcpyDeviceToHost));
    if (rho) HIP_TRY(hipMemcpy(rho, h->alm_rho.p, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (cols) *cols = h->alm_C;
    return CILQR_OK;
}

extern "C" int cilqr_set_debug_flags(cilqr_handle* h, int32_t flags) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
#ifndef CILQR_DEV_BUILD
    if (flags) return fail(CILQR_ERR_UNSUPPORTED, "the testing aids are built into libcilqr_amd_dev.so only");
#endif
    h->debug_flags = flags;
    return CILQR_OK;
}

extern "C" int cilqr_set_rollout_mode(cilqr_handle* h, int32_t mode) {
    if (!h || mode < -1 || mode > 1) return fail(CILQR_ERR_BAD_ARG, "mode must be -1, 0 or 1");
    h->rollout_mode = mode;
    return CILQR_OK;
}

extern "C" int cilqr_set_work_sharing(cilqr_handle* h, int32_t mode) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
    if (mode != 0 && mode != 1) return fail(CILQR_ERR_BAD_ARG, "work sharing mode must be 0 or 1");
    h->share = mode;
    return CILQR_OK;
}

extern "C" int cilqr_set_resume_iters(cilqr_handle* h, int32_t iters) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
    if (iters < -1 || iters > 100000) return fail(CILQR_ERR_BAD_ARG, "iterations per slice must be in [0, 100000] (or -1: automatic)");
    h->resume_iters = iters;
    return CILQR_OK;
}

extern "C" int cilqr_work_sharing_stats(cilqr_handle* h, uint32_t out[4]) {
    if (!h || !out) return fail(CILQR_ERR_BAD_ARG, "null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    h->last_parked = 0;
    // the control words are zeroed by persistent launches only: after any other launch they still hold an earlier
    // launch's counts, which are not this handle's last launch's — report zeros then (but for the latch)
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!SL(h).sh_ctl.p || !h->last_launch_reset_ctl) {
        unsigned mask0 = 0;
        int rc0 = report_latch(h, false, &mask0);
        if (rc0) return rc0;
        out[3] = mask0 ? 1u : 0u;
        return CILQR_OK;
    }
    unsigned w[SH_SLOT0];
    HIP_TRY(hipMemcpy(w, SL(h).sh_ctl.p, sizeof(w), hipMemcpyDeviceToHost));
    out[0] = w[SH_ANNOUNCED]; out[1] = w[SH_HELPED]; out[2] = w[SH_HELPERS]; out[3] = w[SH_ERROR];
    h->last_parked = w[SH_PARKED];
    unsigned mask = 0; // (... and what any other slot's launch, or an earlier launch of this slot, latched: not cleared here)
    int rc_l = report_latch(h, false, &mask);
    if (rc_l) return rc_l;
    if (mask) out[3] |= 1u;
    return CILQR_OK;
}

extern "C" int cilqr_resume_stats(cilqr_handle* h, uint32_t* parked) {
    if (!h || !parked) return fail(CILQR_ERR_BAD_ARG, "null argument");
    uint32_t tmp[4];
    int rc = cilqr_work_sharing_stats(h, tmp);
    if (rc) return rc;
    *parked = h->last_parked;
    return CILQR_OK;
}

extern "C" int cilqr_set_helper_mode(cilqr_handle* h, int32_t mode) {
    if (!h || mode < -1 || mode > 1) return fail(CILQR_ERR_BAD_ARG, "mode must be -1, 0 or 1");
    h->helper_mode = mode;
    return CILQR_OK;
}

extern "C" int cilqr_set_group_mode(cilqr_handle* h, int32_t mode) {
    if (!h || mode < -1 || mode > CILQR_GROUP_MAX) return fail(CILQR_ERR_BAD_ARG, "mode must be -1, 0, 1 or 2");
    h->group_mode = mode;
    update_window(h);
    return CILQR_OK;
}

// Batches in flight (include/cilqr_amd.h): k launch slots, each with a stream of its own
extern "C" int cilqr_set_batches_in_flight(cilqr_handle* h, int32_t k) {
    if (!h || k < 1 || k > CILQR_MAX_IN_FLIGHT) return fail(CILQR_ERR_BAD_ARG, "batches in flight must be in [1, 4]");
    HIP_TRY(hipSetDevice(h->device));
    int rc = wait_last_launch(h); // (slots change hands: nothing of this handle may be in flight)
    if (rc) return rc;
    // The slots' streams are created at the HIGHEST stream priority: the runtime keeps one pool of hardware queues per
    // priority level and spreads the streams of a level over its (four) queues, so these get queues of their own instead of
    // sharing one with the caller's stream.  Measured with three slots at normal priority: the third slot's stream landed on
    // the hardware queue of the caller's stream, the per-call event record the launches wait for (`ev_in`) queued up behind
    // that slot's 65 ms kernel, and every third launch waited for it — 9.8 ms per 8 192-trajectory batch against 8.9 with two
    // or four slots (profiles/r05_experiments/in_flight_stream_priority.txt).
    int prio_least = 0, prio_greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    for (int i = 0; i < k; ++i)
        if (k > 1 && !h->slot[i].stream)
            HIP_TRY(hipStreamCreateWithPriority(&h->slot[i].stream, hipStreamNonBlocking, prio_greatest));
    h->in_flight = k;
    h->next_slot = 0;
    h->cur = 0;
    return CILQR_OK;
}

extern "C" int cilqr_join_device(cilqr_handle* h, void* stream) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return order_after_last_launch(h, static_cast<hipStream_t>(stream));
}

extern "C" int cilqr_wait(cilqr_handle* h) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc = wait_last_launch(h);
    if (rc) return rc;
    return fail_if_latched(h); // (a hand-over that failed in ANY slot's launch since the last report)
}

extern "C" int cilqr_slot_kernel_ms(cilqr_handle* h, int32_t k, float* ms) {
    if (!h || !ms || k < 0 || k >= CILQR_MAX_IN_FLIGHT) return fail(CILQR_ERR_BAD_ARG, "bad argument");
    LaunchSlot& sl = h->slot[k];
    if (sl.timed_pending) {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipEventSynchronize(sl.ev1));
        HIP_TRY(hipEventElapsedTime(&sl.last_ms, sl.ev0, sl.ev1));
        sl.timed_pending = false;
    }
    *ms = sl.last_ms;
    return CILQR_OK;
}

extern "C" int cilqr_get_phase_cycles(cilqr_handle* h, int64_t* out, int32_t B) {
    if (!h || !out || B < 1 || B > h->prof_B || !h->prof.p) return fail(CILQR_ERR_BAD_ARG, "no phase profile available");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->prof.p, sizeof(long long) * CILQR_PROF_SLOTS * (size_t)B, hipMemcpyDeviceToHost));
    return CILQR_OK;
}

extern "C" int cilqr_last_kernel_ms(cilqr_handle* h, float* ms) {
    if (!h || !ms) return fail(CILQR_ERR_BAD_ARG, "null argument");
    if (SL(h).timed_pending) {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipEventSynchronize(SL(h).ev1));
        HIP_TRY(hipEventElapsedTime(&SL(h).last_ms, SL(h).ev0, SL(h).ev1));
        SL(h).timed_pending = false;
    }
    *ms = SL(h).last_ms;
    return CILQR_OK;
}

extern "C" int cilqr_last_launch_info(cilqr_handle* h, int32_t out[4]) {
    if (!h || !out) return fail(CILQR_ERR_BAD_ARG, "null argument");
    for (int i = 0; i < 4; ++i) out[i] = h->last_info[i];
    return CILQR_OK;
}

extern "C" int cilqr_set_params(cilqr_handle* h, const cilqr_params* params, int32_t n_params) {
    if (!h || !params || n_params < 1) return fail(CILQR_ERR_BAD_ARG, "bad params table");
    for (int i = 0; i < n_params; ++i) {
        const cilqr_params& p = params[i];
        if (p.N < 2 || p.N > CILQR_MAX_HORIZON) return fail(CILQR_ERR_BAD_ARG, "N must be in [2, 255]");
        if (p.N != params[0].N) return fail(CILQR_ERR_BAD_ARG, "all parameter sets of a handle must share N");
        if (p.solve_type != 0 && p.solve_type != 1) return fail(CILQR_ERR_BAD_ARG, "solve_type must be 0 (barrier) or 1 (alm)");
        if (p.solve_type != params[0].solve_type) return fail(CILQR_ERR_BAD_ARG, "all parameter sets of a handle must share solve_type");
        if (p.reference_point != 0 && p.reference_point != 1) return fail(CILQR_ERR_BAD_ARG, "reference_point must be 0 or 1");
        if (p.max_iter < 0) return fail(CILQR_ERR_BAD_ARG, "max_iter < 0");
    }
    HIP_TRY(hipSetDevice(h->device));
    {
        int rcw = wait_last_launch(h); // (launches in flight read the table that is about to be rewritten)
        if (rcw) return rcw;
    }
    if (!h->params.empty() && (h->params[0].N != params[0].N || h->params[0].solve_type != params[0].solve_type)) {
        // the multiplier arrays are [B][N][C]: another horizon (or leaving ALM mode) makes their contents
        // meaningless and their size wrong — the next ALM solve allocates and zeroes them afresh
        h->alm_B = 0; h->alm_C = 0; h->alm_N = 0;
    }
    h->params.assign(params, params + n_params);
    if (h->d_params.ensure(sizeof(cilqr_params) * n_params)) return fail(CILQR_ERR_DEVICE, "hipMalloc params");
    HIP_TRY(hipMemcpy(h->d_params.p, params, sizeof(cilqr_params) * n_params, hipMemcpyHostToDevice));
    update_window(h);
    return CILQR_OK;
}

// Constants of convex_interior() (cilqr_device.hpp) for one lane table, every rounding taken against the
// certificate: g = min_j (L_{j+2} - L_{j+1}).(M_{j+1} - M_j), h = max_j |(L_{j+2} - L_{j+1}) - (L_{j+1} - L_j)|,
// s_max = longest segment; rcap = min(1e3, (g - 1e-6) / h) (0 when the table cannot be certified).
static void lane_convexity_bounds(const double* x, const double* y, int L, double* rcap, double* smax) {
    *rcap = 0.0;
    *smax = 0.0;
    if (L < 3) return;
    double g = HUGE_VAL, hm = 0.0, sm = 0.0, cmax = 0.0;
    for (int j = 0; j < L; ++j) cmax = std::max(cmax, std::max(std::fabs(x[j]), std::fabs(y[j])));
    for (int j = 0; j + 1 < L; ++j) sm = std::max(sm, std::hypot(x[j + 1] - x[j], y[j + 1] - y[j]));
    for (int j = 0; j + 2 < L; ++j) {
        const double d0x = x[j + 1] - x[j], d0y = y[j + 1] - y[j];
        const double d1x = x[j + 2] - x[j + 1], d1y = y[j + 2] - y[j + 1];
        const double mx = 0.5 * (x[j + 2] - x[j]), my = 0.5 * (y[j + 2] - y[j]); // M_{j+1} - M_j
        g = std::min(g, d1x * mx + d1y * my);
        hm = std::max(hm, std::hypot(d1x - d0x, d1y - d0y));
    }
    if (!(cmax < 1e6) || !(sm < 1e3) || !(g == g) || !(hm == hm)) return; // non-finite or absurd table
    // coordinate differences carry an absolute error of a few ulp(cmax); products with segments <= sm
    const double slack = 64.0 * 2.220446049250313e-16 * (cmax + 1.0);
    g = g * (1.0 - 1e-9) - slack * (sm + 1.0);
    hm = hm * (1.0 + 1e-9) + slack;
    sm = sm * (1.0 + 1e-9) + slack;
    if (!(g > 1e-6)) return;
    double r = (g - 1e-6) / hm;
    r = std::min(r, 1e3) * (1.0 - 1e-8);
    *rcap = r;
    *smax = sm;
}

// Uploads a scenario table into NEW device arrays and swaps it in only when every step has succeeded: on any
// failure the handle keeps its previous tables (and their device arrays) untouched.
static int set_scenarios_impl(cilqr_handle* h, const cilqr_scenario_desc* scen, int32_t n_scen) {
    if (!h || !scen || n_scen < 1) return fail(CILQR_ERR_BAD_ARG, "bad scenario table");
    for (int i = 0; i < n_scen; ++i) {
        const cilqr_scenario_desc& s = scen[i];
        if (!s.lane_x || !s.lane_y || !s.lane_yaw || s.L < 1 || s.L > 65535)
            return fail(CILQR_ERR_BAD_ARG, "lane table missing or L outside [1, 65535]");
        if (s.M < 0 || (s.M > 0 && (!s.obs || s.T < 1))) return fail(CILQR_ERR_BAD_ARG, "bad obstacle block");
    }
    HIP_TRY(hipSetDevice(h->device));
    struct Pending { // everything allocated here is freed again unless commit() is reached
        std::vector<void*> keep, tmp;
        ~Pending() {
            for (void* p : tmp) (void)hipFree(p);
            for (void* p : keep) (void)hipFree(p);
        }
        int alloc(void** p, size_t bytes, bool temporary) {
            if (hipMalloc(p, bytes) != hipSuccess) return -1;
            (temporary ? tmp : keep).push_back(*p);
            return 0;
        }
    } pend;
    std::vector<DevScene> scenes;
    std::vector<int> scene_T, scene_M;
    std::vector<double> spacing, velo;
    for (int i = 0; i < n_scen; ++i) {
        const cilqr_scenario_desc& s = scen[i];
        std::vector<double> xy(2 * (size_t)s.L);
        for (int j = 0; j < s.L; ++j) {
            xy[2 * j] = s.lane_x[j];
            xy[2 * j + 1] = s.lane_y[j];
        }
        DevScene d;
        std::memset(&d, 0, sizeof(d));
        void *p_xy = nullptr, *p_aux = nullptr, *p_obs = nullptr, *p_tmp_yaw = nullptr, *p_tmp_obs = nullptr;
        const size_t MT = (size_t)s.M * (size_t)(s.M > 0 ? s.T : 0);
        if (pend.alloc(&p_xy, sizeof(double) * xy.size(), false) ||
            pend.alloc(&p_aux, sizeof(double) * CILQR_AUX_STRIDE * (size_t)s.L, false) ||
            pend.alloc(&p_tmp_yaw, sizeof(double) * s.L, true) ||
            (s.M > 0 && (pend.alloc(&p_obs, sizeof(double) * CILQR_OBS_STRIDE * MT, false) ||
                         pend.alloc(&p_tmp_obs, sizeof(double) * 3 * MT, true))))
            return fail(CILQR_ERR_DEVICE, "hipMalloc scenario tables");
        HIP_TRY(hipMemcpyAsync(p_xy, xy.data(), sizeof(double) * xy.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(p_tmp_yaw, s.lane_yaw, sizeof(double) * s.L, hipMemcpyHostToDevice, h->stream));
        if (s.M > 0) HIP_TRY(hipMemcpyAsync(p_tmp_obs, s.obs, sizeof(double) * 3 * MT, hipMemcpyHostToDevice, h->stream));
        {
            const int n = (int)((size_t)s.L > MT ? (size_t)s.L : MT);
            hipLaunchKernelGGL(k_prepare_tables, dim3((n + 255) / 256), dim3(256), 0, h->stream,
                               static_cast<const double*>(p_tmp_yaw), static_cast<double*>(p_aux), s.L,
                               static_cast<const double*>(p_tmp_obs), static_cast<double*>(p_obs), (int)MT);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(h->stream)); // xy (a local) and the caller's arrays are read by now
        }
        d.lane_xy = static_cast<const double*>(p_xy);
        d.lane_aux = static_cast<const double*>(p_aux);
        d.obs = static_cast<const double*>(p_obs);
        d.L = s.L; d.M = s.M; d.T = s.T;
        d.border_hi = s.road_borders[0];
        d.border_lo = s.road_borders[1];
        d.ref_velo = s.ref_velo;
        lane_convexity_bounds(s.lane_x, s.lane_y, s.L, &d.cert_rcap, &d.cert_smax);
        scenes.push_back(d);
        scene_T.push_back(s.T);
        scene_M.push_back(s.M);
        double span = 0.0;
        for (int j = 1; j < s.L; ++j) span += std::hypot(s.lane_x[j] - s.lane_x[j - 1], s.lane_y[j] - s.lane_y[j - 1]);
        spacing.push_back(s.L > 1 ? span / (s.L - 1) : 0.1);
        velo.push_back(std::fabs(s.ref_velo));
    }
    // the table of DevScene records: a new array as well, so that solves still queued on other streams keep
    // reading the old one until the device has drained
    DevBuf d_new;
    if (d_new.ensure(sizeof(DevScene) * n_scen)) return fail(CILQR_ERR_DEVICE, "hipMalloc scenes");
    {
        hipError_t e = hipMemcpy(d_new.p, scenes.data(), sizeof(DevScene) * n_scen, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipDeviceSynchronize(); // nothing in flight may still use the old tables
        if (e != hipSuccess) {
            d_new.release();
            return fail(CILQR_ERR_DEVICE, std::string("scenario upload: ") + hipGetErrorString(e));
        }
    }
    // commit
    free_scenes(h);
    h->d_scenes.release();
    h->d_scenes = d_new;
    h->scenes.swap(scenes);
    h->scene_T.swap(scene_T);
    h->scene_M.swap(scene_M);
    h->scene_spacing.swap(spacing);
    h->scene_velo.swap(velo);
    h->scene_allocs.swap(pend.keep); // free_scenes() left it empty: pend now owns nothing permanent
    update_window(h);
    return CILQR_OK;
}

extern "C" int cilqr_set_scenarios(cilqr_handle* h, const cilqr_scenario_desc* scen, int32_t n_scen) {
    if (h) h->one.valid = false; // the single-ego cache describes tables that are about to be replaced
    return set_scenarios_impl(h, scen, n_scen);
}

// host-side validation of the index arrays (the device trusts them)
static int validate_ids(cilqr_handle* h, int B, const int32_t* scenario_id, const int32_t* param_id,
                        const int32_t* tick) {
    const int N = h->params[0].N;
    const int ns = (int)h->scenes.size(), np = (int)h->params.size();
    for (int b = 0; b < B; ++b) {
        int sid = scenario_id ? scenario_id[b] : 0;
        int pid = param_id ? param_id[b] : 0;
        int tk = tick ? tick[b] : 0;
        if (sid < 0 || sid >= ns) return fail(CILQR_ERR_BAD_ARG, "scenario_id out of range");
        if (pid < 0 || pid >= np) return fail(CILQR_ERR_BAD_ARG, "param_id out of range");
        if (tk < 0) return fail(CILQR_ERR_BAD_ARG, "tick < 0");
        if (h->scene_M[sid] > 0 && tk + N + 1 > h->scene_T[sid])
            return fail(CILQR_ERR_OBSTACLE_HORIZON, "obstacle route shorter than tick + N + 1");
    }
    return CILQR_OK;
}

struct Staged {
    const int32_t* sid = nullptr;
    const int32_t* pid = nullptr;
    const int32_t* tick = nullptr;
};

// upload the optional id arrays into staging buffers 0..2
static int stage_ids(cilqr_handle* h, int B, const int32_t* scenario_id, const int32_t* param_id,
                     const int32_t* tick, Staged& out) {
    const int32_t* src[3] = {scenario_id, param_id, tick};
    const int32_t** dst[3] = {&out.sid, &out.pid, &out.tick};
    for (int i = 0; i < 3; ++i) {
        if (!src[i]) continue;
        if (h->st[i].ensure(sizeof(int32_t) * B)) return fail(CILQR_ERR_DEVICE, "hipMalloc ids");
        HIP_TRY(hipMemcpyAsync(h->st[i].p, src[i], sizeof(int32_t) * B, hipMemcpyHostToDevice, h->stream));
        *dst[i] = static_cast<const int32_t*>(h->st[i].p);
    }
    return CILQR_OK;
}

static int up(cilqr_handle* h, int slot, const void* src, size_t bytes, const double** out) {
    if (h->st[slot].ensure(bytes)) return fail(CILQR_ERR_DEVICE, "hipMalloc staging");
    HIP_TRY(hipMemcpyAsync(h->st[slot].p, src, bytes, hipMemcpyHostToDevice, h->stream));
    *out = static_cast<const double*>(h->st[slot].p);
    return CILQR_OK;
}

static int alloc_out(cilqr_handle* h, int slot, size_t bytes, void** out) {
    if (h->st[slot].ensure(bytes)) return fail(CILQR_ERR_DEVICE, "hipMalloc staging");
    *out = h->st[slot].p;
    return CILQR_OK;
}

// Which build of the solve kernel a batch gets.  Small batches: a helper wavefront per block.  Beyond that lone
// wavefronts in the register build that lets two share a SIMD, one trial per pass.  Two rows per lane (horizons
// above 63), barrier mode: those blocks help each other's line searches once they are done (k_solve's SHARE), which
// beats the helper wavefront from 1536 trajectories on (N = 64 ... 80: 17.4 vs 20.0 ms at N = 72, B = 2048; below
// that the helper wavefront is 10-20 % ahead) and from 512 on for N >= 96 (config 4's mix: 17.1 vs 28.2 ms at 1024,
// 26.1 vs 50.0 at 2048; the two tie below).  Of the augmented-Lagrangian builds only the one that keeps the cost
// expansion in global memory (two rows per lane, large batches) shares work; the helper range of the others grows with
// the horizon as measured before work sharing existed.
static bool two_rows(const cilqr_handle* h) { return !h->params.empty() && h->params[0].N + 1 > CILQR_WAVE; }
// horizons of 128 ... 255 (round 6): four rows per lane.  ONE family of builds takes them — the grouped kernel's long layout, both
// solve types, and the closed loop in one launch in barrier mode — so every launch of such a handle runs in pairs whatever the batch
// size and the group mode; what has no build at these horizons (closed loop under the augmented Lagrangian, piecewise entry points,
// testing aids, cycle accounting) says CILQR_ERR_UNSUPPORTED
static bool four_rows(const cilqr_handle* h) { return !h->params.empty() && h->params[0].N + 1 > 2 * CILQR_WAVE; }
static bool wants_helper(const cilqr_handle* h, int B) {
    if (four_rows(h)) return false;
    if (h->helper_mode >= 0) return h->helper_mode == 1;
    if (!two_rows(h)) return B <= h->helper_max_batch;
    if (h->helper_max_batch_two_rows >= 0) return B <= h->helper_max_batch_two_rows;
    const int N = h->params[0].N;
    if (h->params[0].solve_type == 1 || !h->share) return B <= 1536 + 80 * (N - 60);
    return B <= (N >= 96 ? 512 : 1536);
}
// lone wavefronts, two per SIMD (the builds with WPS = 2, NTP = 1)?
static bool lone_two_per_simd(const cilqr_handle* h, int B) {
    if (wants_helper(h, B)) return false;
    if (h->looping) return true; // (closed loop: helper wavefronts or lone wavefronts two per SIMD, nothing else)
    const bool alm = h->params[0].solve_type == 1;
    (void)alm;
    return true; // (everything beyond the helper range: the register build that lets two wavefronts share a SIMD)
}

// does the solve-kernel variant for this batch cost one trial per pass without a helper (one stage-cost slot)?
// Mirrors the dispatch in cilqr_solve_batch_device, which checks the two against each other.
static bool single_slot(const cilqr_handle* h, int B) {
    const bool alm = h->params[0].solve_type == 1;
    const bool prof2 = h->profiling && h->prof_two_per_simd && h->params[0].N == 50 && h->debug_flags == 0;
    if (!alm && !h->looping && (h->debug_flags != 0 || (h->profiling && !prof2))) return false;
    return lone_two_per_simd(h, B);
}

// does this batch run a build that keeps the cost expansion in global memory (k_solve's LG)?
static bool global_expansion(const cilqr_handle* h, int B) {
    if (!single_slot(h, B) || h->looping) return false;
    if (h->global_expansion == 0 && h->params[0].solve_type == 1) return false; // (the switch still means something under ALM only)
    const int N = h->params[0].N;
    const int alm = h->params[0].solve_type == 1 ? 1 : 0;
    if (N + 1 <= CILQR_WAVE) return false; // (builds exist for two rows per lane only; shorter horizons fit anyway)
    // worth it where the LDS block with the expansion inside keeps a CU from holding the 8 wavefronts its registers
    // allow (barrier mode: N >= 76; augmented Lagrangian, whose dense l_xx makes the block larger: every horizon
    // above 63): measured +30-40 % at N = 100, +6 % at N = 80, -5 % at N = 64 (barrier) where nothing is gained
    if (!alm) return true; // (round 5: the one lone two-row barrier build the library carries)
    return h->global_expansion == 1 || lds_bytes(N, 64, alm, 1, 0) * 8 > 163840;
}

// does this batch run the grouped build (k_solve_grp: CILQR_GROUP trajectories per wavefront, one rollout pass for all)?
// Barrier mode, one row per lane, persistent lone wavefronts two per SIMD, no closed loop, no testing aids.
static bool grouped(const cilqr_handle* h, int B) {
    if (four_rows(h)) return !(h->looping && h->params[0].solve_type == 1) && h->debug_flags == 0 && !h->profiling && h->persistent_blocks;
    if (h->group_mode == 0 || h->group_mode == 1) return false;
    if (h->debug_flags != 0) return false;
    if (h->params[0].solve_type == 1 && ((!h->group_alm && h->group_mode < 2) || h->looping || h->profiling)) return false; // (ALM in pairs: the long layout)
    // (the closed loop in one launch on the long layout — round 6, not yet run on a GPU — only when pairs are asked for explicitly;
    //  by default horizons of 64 ... 127 loop on k_solve's builds)
    if (two_rows(h) && (!h->group_long || (h->looping && h->group_mode < 2))) return false;
    if (h->looping && (!h->group_loop || h->profiling)) return false; // (closed loop in one launch: the LOOP builds of k_solve_grp)
    if (h->profiling && !(CILQR_GPROF && h->group_mode >= 2)) return false; // (cycle accounting: development library, when forced)
    if (!h->persistent_blocks) return false;
    if (h->group_mode >= 2) return true;
#ifndef CILQR_COMPILER_VALIDATED
    return false; // (built with a compiler other than the validated one, build.py: pairs only when asked for explicitly)
#endif
    return lone_two_per_simd(h, B);
}

static BatchArgs make_args(cilqr_handle* h, int B, const Staged& ids) {
    BatchArgs a;
    a.params = static_cast<const cilqr_params*>(h->d_params.p);
    a.scenes = static_cast<const DevScene*>(h->d_scenes.p);
    a.scenario_id = ids.sid;
    a.param_id = ids.pid;
    a.tick = ids.tick;
    a.scratch = static_cast<double*>(SL(h).scratch.p);
    a.prof = nullptr;
    a.B = B;
    a.N = h->params[0].N;
    a.n_params = (int)h->params.size();
    a.n_scenes = (int)h->scenes.size();
    a.flags = h->debug_flags;
    a.tier = h->rollout_mode;
    a.alm = h->params[0].solve_type == 1 ? 1 : 0;
    // the kernel variant built for two wavefronts per SIMD (see the dispatch in cilqr_solve_batch_device)
    const bool occ2 = lone_two_per_simd(h, B) && (a.alm || a.flags == 0) &&
                      (!h->profiling || (h->prof_two_per_simd && a.N == 50));
    a.W = occ2 ? (single_slot(h, B) ? (global_expansion(h, B) ? h->win_lg : h->win_occ) : h->win_occ2) : h->win;
    if (h->fused_call && grouped(h, B)) a.W = h->win_grp;
    a.alm_mu = static_cast<double*>(h->alm_mu.p);
    a.alm_mu_next = static_cast<double*>(h->alm_mu_next.p);
    a.alm_rho = static_cast<double*>(h->alm_rho.p);
    a.alm_C = h->alm_C;
    a.sh_ctl = nullptr;
    a.sh_req = nullptr;
    a.sh_hints = nullptr;
    a.sh_max_helpers = h->share_max_helpers;
    a.sh_min_t0 = h->share_min_t0 < 1 ? 1 : h->share_min_t0;
    a.sh_backoff = h->share_backoff < 0 ? 0 : (h->share_backoff > 8 ? 8 : h->share_backoff);
    a.timeline = nullptr;
    a.next = nullptr;
    a.park = nullptr;
    a.rq = nullptr;
    a.ctl = static_cast<unsigned*>(SL(h).sh_ctl.p);
    a.rq_cap = 0;
    a.res_iters = 0;
    a.res_window = 0;
    a.loop_ticks = 0;
    a.loop_x0 = nullptr;
    a.loop_tick = nullptr;
    a.loop_states = nullptr;
    a.loop_iters = nullptr;
    a.pair_costs = h->group_pair_costs;
    a.pair_sweep = h->group_pair_sweep;
    return a;
}

// ALM multipliers [B][N][C] + rho [B], kept by the handle across calls (hpp:106-112).  Re-laid out when the
// horizon or the column count (8 + 2 max M) changes — then zeroed: the old contents mean nothing — and grown
// when the batch grows: rows that exist keep their multipliers (a warm-started call continues from them, as the
// reference instance does), new rows start at zero / alm_rho_init.  Work still in flight on any stream may be
// using the old arrays, so the device is drained first.
static int ensure_alm(cilqr_handle* h, int B) {
    if (h->params[0].solve_type != 1) return CILQR_OK;
    const int N = h->params[0].N;
    int maxM = 0;
    for (int m : h->scene_M) maxM = m > maxM ? m : maxM;
    const int C = 8 + 2 * maxM;
    const bool same_layout = (C == h->alm_C && N == h->alm_N && h->alm_mu.p);
    if (same_layout && B <= h->alm_B) return CILQR_OK;
    HIP_TRY(hipDeviceSynchronize());
    const size_t row = sizeof(double) * (size_t)N * C;
    const size_t nb = row * (size_t)B;
    const int keep = same_layout ? h->alm_B : 0; // rows whose multipliers survive
    DevBuf mu, mun, rho;
    if (mu.ensure(nb) || mun.ensure(nb) || rho.ensure(sizeof(double) * B)) {
        mu.release(); mun.release(); rho.release();
        return fail(CILQR_ERR_DEVICE, "hipMalloc alm state");
    }
    std::vector<double> rho0((size_t)B, h->params[0].alm_rho_init);
    hipError_t e = hipMemset(mu.p, 0, nb);
    if (e == hipSuccess) e = hipMemset(mun.p, 0, nb);
    if (e == hipSuccess) e = hipMemcpy(rho.p, rho0.data(), sizeof(double) * B, hipMemcpyHostToDevice);
    if (e == hipSuccess && keep > 0) {
        e = hipMemcpy(mu.p, h->alm_mu.p, row * keep, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(mun.p, h->alm_mu_next.p, row * keep, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(rho.p, h->alm_rho.p, sizeof(double) * keep, hipMemcpyDeviceToDevice);
    }
    if (e != hipSuccess) {
        mu.release(); mun.release(); rho.release();
        return fail(CILQR_ERR_DEVICE, std::string("alm state: ") + hipGetErrorString(e));
    }
    h->alm_mu.release(); h->alm_mu_next.release(); h->alm_rho.release();
    h->alm_mu = mu; h->alm_mu_next = mun; h->alm_rho = rho;
    h->alm_B = B;
    h->alm_C = C;
    h->alm_N = N;
    return CILQR_OK;
}

// fused = for the fused solve: its large-batch launches run persistent blocks, which use one scratch area per
// resident block (at most 8 per CU) instead of one per trajectory
static int ensure_scratch(cilqr_handle* h, int B, bool fused = false) {
    const int N = h->params[0].N;
    {
        int rc_alm = ensure_alm(h, B);
        if (rc_alm) return rc_alm;
    }
    size_t areas = (size_t)B;
    if (fused && h->persistent_blocks && lone_two_per_simd(h, B) &&
        (h->params[0].solve_type == 1 || (h->debug_flags == 0 && (!h->profiling || (h->prof_two_per_simd && N == 50)))))
        areas = std::min<size_t>(areas, (size_t)CILQR_MAX_BLOCKS_PER_CU * (size_t)h->num_cus);
    size_t area = scratch_doubles(N);
    if (fused && grouped(h, B)) { // (persistent blocks: one area per resident block, CILQR_GROUP trajectories in it)
        area = std::max(area, (size_t)grp_n(h) * grp_scratch_doubles(N));
        areas = std::min<size_t>(areas, (size_t)CILQR_MAX_BLOCKS_PER_CU_G1 * (size_t)h->num_cus);
    }
    if (SL(h).scratch.ensure(sizeof(double) * area * areas))
        return fail(CILQR_ERR_DEVICE, "hipMalloc scratch");
    if (h->poison_scratch) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(SL(h).scratch.p, h->poison_scratch == 1 ? 0xFF : 0x00, SL(h).scratch.cap));
        HIP_TRY(hipDeviceSynchronize());
    }
    // the launch's control words: the persistent blocks' trajectory counter, the counters and slots of the work sharing
    if (!SL(h).sh_ctl.p) {
        if (SL(h).sh_ctl.ensure(sizeof(unsigned) * CILQR_SH_WORDS)) return fail(CILQR_ERR_DEVICE, "hipMalloc control words");
        HIP_TRY(hipMemset(SL(h).sh_ctl.p, 0, sizeof(unsigned) * CILQR_SH_WORDS));
    }
    // resumable solves: only the launches that can park — two rows per lane, barrier mode, persistent blocks (lone
    // wavefronts two per SIMD, no closed loop), more trajectories than resident blocks
    const bool can_park = fused && two_rows(h) && h->resume_iters != 0 && h->params[0].solve_type == 0 && !h->looping &&
                          h->persistent_blocks && lone_two_per_simd(h, B) && h->debug_flags == 0 && !h->profiling &&
                          (size_t)B > (size_t)h->num_cus; // (more than one round of resident blocks is possible)
    // (one buffer serves both kinds of parked state — k_solve's resumable solves and the grouped build's hand-overs — sized for
    //  the larger record: a handle may switch between the two kernels from call to call)
    const size_t park_need = sizeof(double) * std::max(park_doubles(N), grp_park_doubles(N)) * (size_t)B;
    if (can_park && (B > SL(h).park_B || N != SL(h).park_N || !SL(h).park.p || SL(h).park.cap < park_need)) {
        int rcw = wait_slot(h, h->cur); // (this handle's previous launch may be using the old arrays)
        if (rcw) return rcw;
        SL(h).park.release(); SL(h).rq.release();
        if (SL(h).park.ensure(park_need) || SL(h).rq.ensure(sizeof(unsigned long long) * (size_t)B))
            return fail(CILQR_ERR_DEVICE, "hipMalloc parked-solve state");
        SL(h).park_B = B;
        SL(h).park_N = N;
    }
    // the grouped build hands trajectories from wavefronts that hold two to wavefronts that have run dry (cilqr_group.hpp)
    if (fused && grouped(h, B) && h->group_steal) {
        const size_t need = park_need;
        const size_t rq_need = sizeof(unsigned long long) * (size_t)B * CILQR_GRP_Q_PER_TRAJECTORY; // (not reused within a launch)
        if (B > SL(h).park_B || N != SL(h).park_N || !SL(h).park.p || SL(h).park.cap < need || SL(h).rq.cap < rq_need) {
            int rcw = wait_slot(h, h->cur);
            if (rcw) return rcw;
            SL(h).park.release(); SL(h).rq.release();
            if (SL(h).park.ensure(need) || SL(h).rq.ensure(rq_need))
                return fail(CILQR_ERR_DEVICE, "hipMalloc parked-solve state");
            SL(h).park_B = B;
            SL(h).park_N = N;
        }
    }
    // work sharing between blocks (builds of horizons above 63): one request and one row of hints per trajectory
    if (two_rows(h) && (B > SL(h).sh_B || N != SL(h).sh_N || !SL(h).sh_req.p)) {
        int rcw = wait_slot(h, h->cur);
        if (rcw) return rcw;
        SL(h).sh_req.release(); SL(h).sh_hints.release();
        if (SL(h).sh_req.ensure(sizeof(ShareReq) * (size_t)B) || SL(h).sh_hints.ensure(sizeof(int) * (size_t)(N + 2) * (size_t)B))
            return fail(CILQR_ERR_DEVICE, "hipMalloc work-sharing state");
        HIP_TRY(hipMemset(SL(h).sh_req.p, 0xff, sizeof(ShareReq) * (size_t)B)); // every request closed (next = 255)
        SL(h).sh_B = B;
        SL(h).sh_N = N;
    }
    return CILQR_OK;
}

#define DL(slot, ptr, bytes) \
    HIP_TRY(hipMemcpyAsync((ptr), h->st[slot].p, (bytes), hipMemcpyDeviceToHost, h->stream))

// do two launches touch a common byte with a write on either side?
static bool ranges_conflict(const std::vector<MemRange>& a, const std::vector<MemRange>& b) {
    for (const auto& x : a)
        for (const auto& y : b)
            if ((x.w || y.w) && x.p < y.p + y.n && y.p < x.p + x.n) return true;
    return false;
}

struct SeqScope {
    cilqr_handle* h;
    explicit SeqScope(cilqr_handle* hh) : h(hh) { h->force_seq = true; h->cur = 0; }
    ~SeqScope() { h->force_seq = false; }
};

struct LoopArgs {
    int ticks = 0;
    double* x0 = nullptr;
    int32_t* tick = nullptr;
    double* states = nullptr;
    int32_t* iters = nullptr;
};

static int solve_batch_device_impl(cilqr_handle* h, int32_t B, const double* d_x0,
                                   const int32_t* d_scenario_id, const int32_t* d_param_id,
                                   const int32_t* d_tick, const double* d_last_u,
                                   double* d_u_out, double* d_x_out, cilqr_result* d_res_out,
                                   cilqr_trace_rec* d_trace_out, int32_t trace_cap, void* stream, const LoopArgs& loop) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (B < 1 || !d_x0 || !d_u_out || !d_x_out) return fail(CILQR_ERR_BAD_ARG, "bad batch arguments");
    if (trace_cap < 0) return fail(CILQR_ERR_BAD_ARG, "trace_cap < 0");
    HIP_TRY(hipSetDevice(h->device));
    struct LoopScope { // (the launch-shape helpers below look at the handle)
        cilqr_handle* h;
        LoopScope(cilqr_handle* hh, bool on) : h(hh) { h->looping = on; }
        ~LoopScope() { h->looping = false; }
    } loop_scope(h, loop.ticks >= 1);
    struct FusedScope {
        cilqr_handle* h;
        explicit FusedScope(cilqr_handle* hh) : h(hh) { h->fused_call = true; }
        ~FusedScope() { h->fused_call = false; }
    } fused_scope(h);
    if (loop.ticks >= 1 && (h->debug_flags != 0 || h->profiling))
        return fail(CILQR_ERR_UNSUPPORTED, "the closed loop has no testing-aid / cycle-accounting builds");
    if (four_rows(h) && !grouped(h, B))
        return fail(CILQR_ERR_UNSUPPORTED, "horizons above 127 run the grouped kernel's long layout only: no closed loop in one launch under "
                                           "the augmented Lagrangian, no testing aids / cycle accounting, persistent blocks on");
    // Which launch slot: slot 0 on the caller's stream, one launch of the handle at a time — or, after
    // cilqr_set_batches_in_flight(k > 1), the next of k slots round robin, each with a stream, scratch areas and control words
    // of its own.  Not for the augmented Lagrangian (its multipliers live in the handle, indexed by trajectory: two launches
    // would share them) and not with the development aids (one timeline / cycle-accounting buffer per handle).
    const bool slots_apply = h->in_flight > 1 && !h->force_seq && h->params[0].solve_type == 0 && !h->profiling && !h->timeline &&
                             !h->poison_scratch && h->debug_flags == 0;
    std::vector<MemRange> ranges;
    if (slots_apply) {
        h->cur = h->next_slot;
        h->next_slot = (h->next_slot + 1) % h->in_flight;
        const int N_ = h->params[0].N;
        const bool lp = loop.ticks >= 1;
        auto add = [&](const void* p, size_t n, bool w) { if (p && n) ranges.push_back({static_cast<const char*>(p), n, w}); };
        add(d_x0, sizeof(double) * 4 * (size_t)B, lp);
        add(d_scenario_id, sizeof(int32_t) * (size_t)B, false);
        add(d_param_id, sizeof(int32_t) * (size_t)B, false);
        add(d_tick, sizeof(int32_t) * (size_t)B, lp);
        add(d_last_u, sizeof(double) * 2 * N_ * (size_t)B, false);
        add(d_u_out, sizeof(double) * 2 * N_ * (size_t)B, true);
        add(d_x_out, sizeof(double) * 4 * (N_ + 1) * (size_t)B, true);
        add(d_res_out, sizeof(cilqr_result) * (size_t)B, true);
        add(d_trace_out, sizeof(cilqr_trace_rec) * (size_t)B * (size_t)trace_cap, true);
        add(loop.states, sizeof(double) * 4 * (size_t)B * (size_t)(lp ? loop.ticks : 0), true);
        add(loop.iters, sizeof(int32_t) * (size_t)B * (size_t)(lp ? loop.ticks : 0), true);
    } else {
        h->cur = 0;
    }
    rc = ensure_scratch(h, B, true);
    if (rc) return rc;
    Staged ids;
    ids.sid = d_scenario_id; ids.pid = d_param_id; ids.tick = d_tick;
    BatchArgs a = make_args(h, B, ids);
    a.loop_ticks = loop.ticks;
    a.loop_x0 = loop.x0;
    a.loop_tick = loop.tick;
    a.loop_states = loop.states;
    a.loop_iters = loop.iters;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (slots_apply) {
        // in-flight mode: the launch goes to the slot's own stream, behind (a) everything the caller's stream holds so far — the
        // launch's inputs —, (b) the slot's previous launch (stream order; an event if something else used its buffers since)
        // and (c) every launch in flight whose buffers overlap this one's with a write on either side
        // (ONE event, re-recorded by every call: a wait captures the record that precedes it in program order — later
        //  re-records do not move what an already enqueued hipStreamWaitEvent waits for)
        HIP_TRY(hipEventRecord(h->ev_in, s));
        s = SL(h).stream;
        HIP_TRY(hipStreamWaitEvent(s, h->ev_in, 0));
        rc = order_after_slot(h, h->cur, s);
        if (rc) return rc;
        for (int k = 0; k < CILQR_MAX_IN_FLIGHT; ++k)
            if (k != h->cur && h->slot[k].launched && ranges_conflict(h->slot[k].ranges, ranges)) {
                rc = order_after_slot(h, k, s);
                if (rc) return rc;
            }
        SL(h).ranges.swap(ranges);
    } else {
        rc = order_after_last_launch(h, s); // before the first asynchronous touch of a handle buffer (timeline, control words, ...)
        if (rc) return rc;
        SL(h).ranges.clear();
    }
    if (h->profiling) {
        if (h->prof.ensure(sizeof(long long) * CILQR_PROF_SLOTS * (size_t)B)) return fail(CILQR_ERR_DEVICE, "hipMalloc prof");
        a.prof = static_cast<long long*>(h->prof.p);
        h->prof_B = B;
    }
    if (h->timeline) {
        if (h->tl.ensure(sizeof(long long) * 4 * (size_t)B)) return fail(CILQR_ERR_DEVICE, "hipMalloc timeline");
        HIP_TRY(hipMemsetAsync(h->tl.p, 0, sizeof(long long) * 4 * (size_t)B, s));
        a.timeline = static_cast<long long*>(h->tl.p);
        h->tl_B = B;
    }
    if (h->timing) HIP_TRY(hipEventRecord(SL(h).ev0, s));
    if (grouped(h, B)) {
        // CILQR_GROUP trajectories per wavefront, one rollout pass for all of them (cilqr_group.hpp): persistent blocks
        const int G = (loop.ticks >= 1) ? 2 : grp_n(h);
        // compile-time horizons: BASELINE's 50 and the 30 of the reference's own YAMLs (config/scenario_*.yaml:5)
        auto kg = (a.N == 50) ? k_solve_grp<50, 2> : (a.N == 30 ? k_solve_grp<30, 2> : k_solve_grp<0, 2>);
        if (loop.ticks >= 1) kg = (a.N == 50) ? k_solve_grp<50, 2, true> : (a.N == 30 ? k_solve_grp<30, 2, true> : k_solve_grp<0, 2, true>);
        const bool longl = a.N + 1 > CILQR_WAVE || a.alm; // two rows per lane, or the augmented Lagrangian: the long layout
        if (longl) kg = (a.N == 100) ? k_solve_grp<100, 2, false, 2> : k_solve_grp<0, 2, false, 2>;
        if (a.alm) kg = (a.N + 1 > CILQR_WAVE) ? k_solve_grp<0, 2, false, 2, true, true> : k_solve_grp<0, 2, false, 1, true, true>;
        if (a.N + 1 > 2 * CILQR_WAVE) kg = a.alm ? k_solve_grp<0, 2, false, 4, true, true> : k_solve_grp<0, 2, false, 4>;
        if (longl && !a.alm && loop.ticks >= 1) kg = (a.N + 1 > 2 * CILQR_WAVE) ? k_solve_grp<0, 2, true, 4> : k_solve_grp<0, 2, true, 2>;
        const size_t shm = longl ? grpl_lds_bytes(a.N, a.W, G) : grp_lds_bytes(a.N, a.W, G);
        int per_cu = 0;
        rc = blocks_per_cu(h, reinterpret_cast<const void*>(kg), shm, &per_cu);
        if (rc) return rc;
        const int cap = per_cu * h->num_cus, want = (B + G - 1) / G;
        const int grid = want < cap ? want : cap;
        h->last_launch_shared = false;
        a.next = static_cast<unsigned*>(SL(h).sh_ctl.p) + SH_NEXT;
        HIP_TRY(hipMemsetAsync(SL(h).sh_ctl.p, 0, sizeof(unsigned) * CILQR_SH_WORDS, s));
        h->last_launch_reset_ctl = true;
        if (h->group_steal && !a.alm && SL(h).park.p && SL(h).park_B >= B && SL(h).park_N == a.N && // (ALM: multipliers are plain stores, no hand-overs)
            SL(h).rq.cap >= sizeof(unsigned long long) * (size_t)B * CILQR_GRP_Q_PER_TRAJECTORY) {
            a.park = static_cast<double*>(SL(h).park.p);
            a.rq = static_cast<unsigned long long*>(SL(h).rq.p);
            a.rq_cap = (int)std::min<size_t>(0x7fffffff, (size_t)B * CILQR_GRP_Q_PER_TRAJECTORY);
            HIP_TRY(hipMemsetAsync(SL(h).rq.p, 0, sizeof(unsigned long long) * (size_t)a.rq_cap, s));
            a.res_iters = (loop.ticks >= 1) ? 0 : (h->resume_iters >= 0 ? h->resume_iters : (longl ? h->group_slice_long : h->group_slice));
            a.res_window = (int)std::min<long long>(0x7fffffffLL, (long long)grid * G * h->group_slice_window_pct / 100);
            h->last_launch_shared = true; // (the host-buffer entry point then checks the launch's error word: a bounded wait that expired)
        }
        if (SL(h).scratch.cap < sizeof(double) * (size_t)G * grp_scratch_doubles(a.N) * (size_t)grid)
            return fail(CILQR_ERR_DEVICE, "internal: scratch areas / launch shape mismatch");
        if (a.park) { // trajectories change wavefronts in this launch: every result starts as "not solved"
            rc = mark_unsolved(d_res_out, B, s);
            if (rc) return rc;
#ifdef CILQR_DEV_BUILD
            // the test of a lost hand-over forces the wait's expiry — per handle (CILQR_TUNE=grp_wait_spins=n) or, read at every
            // launch so that ONE launch of several in flight can be made to fail, CILQR_GRP_WAIT_SPINS=n
            int spins = h->grp_wait_spins;
            if (const char* e = std::getenv("CILQR_GRP_WAIT_SPINS")) spins = std::atoi(e);
            if (spins > 0)
                HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(static_cast<unsigned*>(SL(h).sh_ctl.p) + SH_TEST_SPINS), spins, 1, s));
#endif
        }
        hipLaunchKernelGGL(kg, dim3(grid), dim3(CILQR_WAVE), shm, s, a, d_x0, d_last_u, d_u_out, d_x_out, d_res_out, d_trace_out,
                           d_trace_out ? trace_cap : 0);
        if (a.park) {
            HIP_TRY(hipGetLastError());
            rc = latch_launch(h, s);
            if (rc) return rc;
        }
        h->last_info[0] = G; h->last_info[1] = grid; h->last_info[2] = CILQR_WAVE; h->last_info[3] = a.W;
    } else {
        const bool two = (a.N + 1 > CILQR_WAVE);
        // helper wavefronts pay off while one wavefront per trajectory leaves SIMDs idle
        const bool help = wants_helper(h, B);
        // `one` = the variant costs one trial per pass without a helper: one stage-cost slot in LDS (k_solve's SLOTS)
        auto kern = k_solve<false, 1, false, true, false>;
        bool one = false, lg = false, persistent = false;
        h->last_launch_shared = false;
        h->last_launch_reset_ctl = false;
        if (loop.ticks >= 1) {
            // closed loop in one launch: the plain builds carry it
            if (a.alm) {
                if (help) kern = two ? k_solve<CILQR_ALM_DBG, 2, true, true, false, 1, CILQR_NT, 0, false, false, false, true>
                                     : k_solve<CILQR_ALM_DBG, 1, true, true, false, 1, CILQR_NT, 0, false, false, false, true>;
                else kern = two ? k_solve<CILQR_ALM_DBG, 2, true, false, false, 2, 1, 0, false, false, false, true>
                                : k_solve<CILQR_ALM_DBG, 1, true, false, false, 2, 1, 0, false, false, false, true>;
            } else {
                if (help) kern = two ? k_solve<false, 2, false, true, false, 1, CILQR_NT, 0, false, false, false, true>
                                     : k_solve<false, 1, false, true, false, 1, CILQR_NT, 0, false, false, false, true>;
                else kern = two ? k_solve<false, 2, false, false, false, 2, 1, 0, false, false, false, true>
                                : k_solve<false, 1, false, false, false, 2, 1, 0, false, false, false, true>;
            }
            one = !help;
            persistent = !help;
        } else if (a.alm) {
            if (help) kern = two ? k_solve<CILQR_ALM_DBG, 2, true, true, false> : k_solve<CILQR_ALM_DBG, 1, true, true, false>;
            else if (lone_two_per_simd(h, B)) {
                kern = two ? k_solve<CILQR_ALM_DBG, 2, true, false, false, 2, 1> : k_solve<CILQR_ALM_DBG, 1, true, false, false, 2, 1>;
                if (global_expansion(h, B)) { kern = k_solve<CILQR_ALM_DBG, 2, true, false, false, 2, 1, 0, true, true, false>; lg = true; }
                one = true;
                persistent = true;
                if (h->share && lg && SL(h).sh_req.p) { // (the ALM build with work sharing is the two-row one)
                    a.sh_ctl = static_cast<unsigned*>(SL(h).sh_ctl.p);
                    a.sh_req = static_cast<ShareReq*>(SL(h).sh_req.p);
                    a.sh_hints = static_cast<int*>(SL(h).sh_hints.p);
                    h->last_launch_shared = true;
                }
            }
            else return fail(CILQR_ERR_DEVICE, "internal: no build for this launch shape");
#ifdef CILQR_DEV_BUILD
        } else if (a.flags != 0) {
            kern = two ? k_solve<true, 2, false, false, false> : k_solve<true, 1, false, false, false>;
        } else if (a.prof) {
            kern = help ? (two ? k_solve<false, 2, false, true, true> : k_solve<false, 1, false, true, true>)
                        : (two ? k_solve<false, 2, false, false, true> : k_solve<false, 1, false, false, true>);
            if (!help && a.N == 50 && h->prof_two_per_simd && lone_two_per_simd(h, B)) {
                // the headline build itself with the accounting: persistent blocks, two wavefronts per SIMD
                kern = k_solve<false, 1, false, false, true, 2, 1, 50>;
                one = true;
                persistent = true;
            }
#endif
        } else if (help) {
            kern = two ? k_solve<false, 2, false, true, false> : k_solve<false, 1, false, true, false>;
            if (a.N == 50) kern = k_solve<false, 1, false, true, false, 1, CILQR_NT, 50>;
            if (a.N == 30) kern = k_solve<false, 1, false, true, false, 1, CILQR_NT, 30>; // (the reference's own horizon)
        } else if (lone_two_per_simd(h, B)) {
            // (two rows per lane: built with work sharing between blocks, which a.sh_ctl switches on.  Shorter horizons
            //  are not: a trial costs 5 us there, the hand-over of a search about 10, and the build costs the solve loop
            //  3 % in spilled registers — measured: config 5 -5 %, config 3 -33 % with it, config 4 +25 %)
            kern = k_solve<false, 1, false, false, false, 2, 1>;
            if (two) {
                if (!global_expansion(h, B)) return fail(CILQR_ERR_DEVICE, "internal: no build for this launch shape");
                kern = k_solve<false, 2, false, false, false, 2, 1, 0, true, true, true>;
                lg = true;
            }
            one = true;
            persistent = true;
            if (h->share && two && SL(h).sh_req.p) {
                a.sh_ctl = static_cast<unsigned*>(SL(h).sh_ctl.p);
                a.sh_req = static_cast<ShareReq*>(SL(h).sh_req.p);
                a.sh_hints = static_cast<int*>(SL(h).sh_hints.p);
                h->last_launch_shared = true;
            }
        } else {
            return fail(CILQR_ERR_DEVICE, "internal: no build for this launch shape");
        }
        const bool helped = help && (a.alm || a.flags == 0 || loop.ticks >= 1);
        if (one != single_slot(h, B)) return fail(CILQR_ERR_DEVICE, "internal: kernel variant / LDS layout mismatch");
        const size_t shm = lds_bytes(a.N, a.W, a.alm, one ? 1 : 2, lg ? 1 : 0);
        int grid = B;
        if (persistent && h->persistent_blocks) {
            // as many blocks as the chip holds at once; they pull trajectories from the counter
            int per_cu = 0;
            rc = blocks_per_cu(h, reinterpret_cast<const void*>(kern), shm, &per_cu);
            if (rc) return rc;
            const int cap = per_cu * h->num_cus;
            grid = B < cap ? B : cap;
            a.next = static_cast<unsigned*>(SL(h).sh_ctl.p) + SH_NEXT;
            HIP_TRY(hipMemsetAsync(SL(h).sh_ctl.p, 0, sizeof(unsigned) * CILQR_SH_WORDS, s));
            h->last_launch_reset_ctl = true;
            // resumable solves: the builds that carry them (two rows per lane, persistent), batches that take more than
            // one round of the resident blocks
            if (two && !a.alm && loop.ticks < 1 && h->resume_iters != 0 && B > grid && SL(h).park.p && SL(h).park_B >= B && SL(h).park_N == a.N) {
                a.park = static_cast<double*>(SL(h).park.p);
                a.rq = static_cast<unsigned long long*>(SL(h).rq.p);
                HIP_TRY(hipMemsetAsync(SL(h).rq.p, 0, sizeof(unsigned long long) * (size_t)SL(h).park_B, s));
                a.rq_cap = SL(h).park_B;
                a.res_iters = h->resume_iters < 0 ? 32 : h->resume_iters;
            }
        } else {
            a.sh_ctl = nullptr; // (the work sharing rides on the persistent blocks)
            a.sh_req = nullptr;
            a.sh_hints = nullptr;
            h->last_launch_shared = false;
        }
        if (SL(h).scratch.cap < sizeof(double) * scratch_doubles(a.N) * (size_t)(a.next ? grid : B))
            return fail(CILQR_ERR_DEVICE, "internal: scratch areas / launch shape mismatch");
        const bool hands_over = a.park != nullptr || a.sh_ctl != nullptr; // (resumable solves / work sharing between blocks)
        if (hands_over) {
            rc = mark_unsolved(d_res_out, B, s);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(helped ? 2 * CILQR_WAVE : CILQR_WAVE), shm, s, a, d_x0, d_last_u, d_u_out,
                           d_x_out, d_res_out, d_trace_out, d_trace_out ? trace_cap : 0);
        if (hands_over) {
            HIP_TRY(hipGetLastError());
            rc = latch_launch(h, s);
            if (rc) return rc;
        }
        h->last_info[0] = 1; h->last_info[1] = grid; h->last_info[2] = helped ? 2 * CILQR_WAVE : CILQR_WAVE; h->last_info[3] = a.W;
    }
    HIP_TRY(hipGetLastError());
    if (h->timing) {
        HIP_TRY(hipEventRecord(SL(h).ev1, s));
        SL(h).timed_pending = true;
    }
    return mark_slot(h, h->cur, s);
}

extern "C" int cilqr_solve_batch_device(cilqr_handle* h, int32_t B, const double* d_x0,
                                        const int32_t* d_scenario_id, const int32_t* d_param_id,
                                        const int32_t* d_tick, const double* d_last_u,
                                        double* d_u_out, double* d_x_out, cilqr_result* d_res_out,
                                        cilqr_trace_rec* d_trace_out, int32_t trace_cap, void* stream) {
    return solve_batch_device_impl(h, B, d_x0, d_scenario_id, d_param_id, d_tick, d_last_u, d_u_out, d_x_out, d_res_out,
                                   d_trace_out, trace_cap, stream, LoopArgs());
}

extern "C" int cilqr_closed_loop_batch_device(cilqr_handle* h, int32_t B, int32_t ticks, double* d_x0,
                                              const int32_t* d_scenario_id, const int32_t* d_param_id, int32_t* d_tick,
                                              const double* d_last_u, double* d_u_out, double* d_x_out,
                                              cilqr_result* d_res_out, double* d_states, int32_t* d_iters, void* stream) {
    if (ticks < 1) return fail(CILQR_ERR_BAD_ARG, "ticks < 1");
    if (!d_tick) return fail(CILQR_ERR_BAD_ARG, "the closed loop needs the tick array (it is advanced on the device)");
    LoopArgs loop;
    loop.ticks = ticks;
    loop.x0 = d_x0;
    loop.tick = d_tick;
    loop.states = d_states;
    loop.iters = d_iters;
    return solve_batch_device_impl(h, B, d_x0, d_scenario_id, d_param_id, d_tick, d_last_u, d_u_out, d_x_out, d_res_out,
                                   nullptr, 0, stream, loop);
}

extern "C" int cilqr_advance_batch_device(cilqr_handle* h, int32_t B, const double* d_x, double* d_x0, int32_t* d_tick,
                                          void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (B < 1 || !d_x || !d_x0) return fail(CILQR_ERR_BAD_ARG, "bad batch arguments");
    HIP_TRY(hipSetDevice(h->device));
    rc = order_after_last_launch(h, static_cast<hipStream_t>(stream)); // (x is the previous solve's output, x0 / tick the next one's input)
    if (rc) return rc;
    hipLaunchKernelGGL(k_advance, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), B, h->params[0].N, d_x,
                       d_x0, d_tick);
    HIP_TRY(hipGetLastError());
    // (x0 / tick are inputs of the handle's next solve wherever it runs: the launch counts as a use of every slot)
    for (int k = 0; k < h->in_flight; ++k) {
        rc = mark_slot(h, k, static_cast<hipStream_t>(stream));
        if (rc) return rc;
    }
    return CILQR_OK;
}

extern "C" int cilqr_solve_batch(cilqr_handle* h, int32_t B, const double* x0,
                                 const int32_t* scenario_id, const int32_t* param_id,
                                 const int32_t* tick, const double* last_u, double* u_out,
                                 double* x_out, cilqr_result* res_out, cilqr_trace_rec* trace_out,
                                 int32_t trace_cap) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (B < 1 || !x0 || !u_out || !x_out) return fail(CILQR_ERR_BAD_ARG, "bad batch arguments");
    if (trace_cap < 0) return fail(CILQR_ERR_BAD_ARG, "trace_cap < 0");
    rc = validate_ids(h, B, scenario_id, param_id, tick);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    SeqScope seq_scope(h); // (host buffers: the handle's staging buffers and stream; never a launch slot of its own)
    rc = order_after_last_launch(h, h->stream); // (the staging buffers are the handle's)
    if (rc) return rc;
    const int N = h->params[0].N;
    Staged ids;
    rc = stage_ids(h, B, scenario_id, param_id, tick, ids);
    if (rc) return rc;
    const double *d_x0 = nullptr, *d_last = nullptr;
    rc = up(h, 3, x0, sizeof(double) * 4 * B, &d_x0);
    if (rc) return rc;
    if (last_u) {
        rc = up(h, 4, last_u, sizeof(double) * 2 * N * (size_t)B, &d_last);
        if (rc) return rc;
    }
    void *d_u, *d_x, *d_res, *d_tr = nullptr;
    const size_t nb_u = sizeof(double) * 2 * N * (size_t)B, nb_x = sizeof(double) * 4 * (N + 1) * (size_t)B;
    rc = alloc_out(h, 5, nb_u, &d_u); if (rc) return rc;
    rc = alloc_out(h, 6, nb_x, &d_x); if (rc) return rc;
    rc = alloc_out(h, 7, sizeof(cilqr_result) * B, &d_res); if (rc) return rc;
    if (trace_out && trace_cap > 0) {
        rc = alloc_out(h, 8, sizeof(cilqr_trace_rec) * (size_t)B * trace_cap, &d_tr);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_tr, 0, sizeof(cilqr_trace_rec) * (size_t)B * trace_cap, h->stream));
    }
    rc = cilqr_solve_batch_device(h, B, d_x0, ids.sid, ids.pid, ids.tick, d_last,
                                  static_cast<double*>(d_u), static_cast<double*>(d_x),
                                  static_cast<cilqr_result*>(d_res),
                                  static_cast<cilqr_trace_rec*>(d_tr), trace_cap, h->stream);
    if (rc) return rc;
    DL(5, u_out, nb_u);
    DL(6, x_out, nb_x);
    if (res_out) DL(7, res_out, sizeof(cilqr_result) * B);
    if (d_tr) DL(8, trace_out, sizeof(cilqr_trace_rec) * (size_t)B * trace_cap);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return fail_if_latched(h); // (the outputs are in the caller's buffers either way: a trajectory lost in transit says CILQR_END_NOT_SOLVED)
}

// CILQRSolver::solve as main() calls it (hpp:37-41, mp:194-196): ONE ego, all arguments handed over on every
// call.  The tables stay in HBM between calls: they are uploaded again only when their contents differ bitwise
// from what is there — except that obstacle predictions which are the tail of the routes uploaded earlier (what
// utils::get_sub_routing_lines, ut:88-103, produces tick after tick) only move the tick offset.  The call's own
// buffers travel in one pinned block each way: in = x0[4] | last_u[N][2] | tick, out = u | x | result.
static bool same_doubles(const double* a, const std::vector<double>& b, size_t n) {
    return b.size() == n && (n == 0 || std::memcmp(a, b.data(), n * sizeof(double)) == 0);
}

extern "C" int cilqr_solve(cilqr_handle* h, const double* x0, const cilqr_scenario_desc* sc, const double* last_u,
                           double* u_out, double* x_out, cilqr_result* res_out) {
    if (!h || h->params.empty()) return fail(CILQR_ERR_BAD_ARG, "cilqr_set_params has not been called");
    if (!x0 || !sc || !u_out || !x_out) return fail(CILQR_ERR_BAD_ARG, "null argument");
    if (!sc->lane_x || !sc->lane_y || !sc->lane_yaw || sc->L < 1 || sc->M < 0 || (sc->M > 0 && (!sc->obs || sc->T < 1)))
        return fail(CILQR_ERR_BAD_ARG, "bad scenario");
    HIP_TRY(hipSetDevice(h->device));
    SeqScope seq_scope(h);
    auto& o = h->one;
    const int N = h->params[0].N;
    const size_t L = (size_t)sc->L;
    // upstream: RoutingLine::operator[] throws std::out_of_range (ut:52-58) — whether or not the routes happen to
    // match what an earlier call uploaded
    if (sc->M > 0 && sc->T < N + 1) return fail(CILQR_ERR_OBSTACLE_HORIZON, "obstacle route shorter than N + 1");
    int d = -1;
    if (o.valid && same_doubles(sc->lane_x, o.lane_x, L) && same_doubles(sc->lane_y, o.lane_y, L) &&
        same_doubles(sc->lane_yaw, o.lane_yaw, L) && sc->road_borders[0] == o.borders[0] &&
        sc->road_borders[1] == o.borders[1] && sc->ref_velo == o.ref_velo && sc->M == o.M) {
        if (sc->M == 0) {
            d = 0;
        } else {
            // candidate offsets: the tail of the uploaded routes, one tick after / the same as last time, none
            const int cand[4] = {o.T - sc->T, o.last_d + 1, o.last_d, 0};
            for (int ci = 0; ci < 4 && d < 0; ++ci) {
                const int dd = cand[ci];
                if (dd < 0 || dd + sc->T > o.T || dd + N + 1 > o.T) continue;
                bool eq = true;
                for (int j = 0; j < sc->M && eq; ++j)
                    eq = std::memcmp(sc->obs + (size_t)j * sc->T * 3, o.obs.data() + ((size_t)j * o.T + dd) * 3,
                                     sizeof(double) * 3 * (size_t)sc->T) == 0;
                if (eq) d = dd;
            }
        }
    }
    if (d < 0) {
        o.valid = false;
        int rc = set_scenarios_impl(h, sc, 1);
        if (rc) return rc;
        o.lane_x.assign(sc->lane_x, sc->lane_x + L);
        o.lane_y.assign(sc->lane_y, sc->lane_y + L);
        o.lane_yaw.assign(sc->lane_yaw, sc->lane_yaw + L);
        o.M = sc->M;
        o.T = sc->M > 0 ? sc->T : 0;
        o.obs.assign(sc->obs ? sc->obs : nullptr, sc->obs ? sc->obs + (size_t)o.M * o.T * 3 : nullptr);
        o.borders[0] = sc->road_borders[0]; o.borders[1] = sc->road_borders[1];
        o.ref_velo = sc->ref_velo;
        o.valid = true;
        o.uploads++;
        d = 0;
    } else {
        o.reuses++;
    }
    o.last_d = d;
    // staging: [x0 4][last_u 2N][tick (one double slot)] in, [u 2N][x 4(N+1)][result] out
    const size_t n_in = 4 + 2 * (size_t)N + 1, n_out = 2 * (size_t)N + 4 * (size_t)(N + 1);
    const size_t bytes_in = n_in * sizeof(double), bytes_out = n_out * sizeof(double) + sizeof(cilqr_result);
    const size_t need = bytes_in + bytes_out;
    if (need > o.cap) {
        if (o.pinned) (void)hipHostFree(o.pinned);
        if (o.dev) (void)hipFree(o.dev);
        o.pinned = o.dev = nullptr;
        o.cap = 0;
        HIP_TRY(hipHostMalloc(&o.pinned, need, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&o.dev, need));
        o.cap = need;
    }
    double* hin = static_cast<double*>(o.pinned);
    std::memcpy(hin, x0, 4 * sizeof(double));
    if (last_u) std::memcpy(hin + 4, last_u, 2 * (size_t)N * sizeof(double));
    int32_t tk = d;
    std::memcpy(hin + 4 + 2 * N, &tk, sizeof(tk));
    char* dbase = static_cast<char*>(o.dev);
    double* din = reinterpret_cast<double*>(dbase);
    double* dout = reinterpret_cast<double*>(dbase + bytes_in);
    HIP_TRY(hipMemcpyAsync(din, hin, bytes_in, hipMemcpyHostToDevice, h->stream));
    int rc = cilqr_solve_batch_device(h, 1, din, nullptr, nullptr, reinterpret_cast<const int32_t*>(din + 4 + 2 * N),
                                      last_u ? din + 4 : nullptr, dout, dout + 2 * N,
                                      reinterpret_cast<cilqr_result*>(dout + n_out), nullptr, 0, h->stream);
    if (rc) return rc;
    char* hout = static_cast<char*>(o.pinned) + bytes_in;
    HIP_TRY(hipMemcpyAsync(hout, dout, bytes_out, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::memcpy(u_out, hout, 2 * (size_t)N * sizeof(double));
    std::memcpy(x_out, hout + 2 * (size_t)N * sizeof(double), 4 * (size_t)(N + 1) * sizeof(double));
    cilqr_result r;
    std::memcpy(&r, hout + n_out * sizeof(double), sizeof(r));
    if (res_out) *res_out = r;
    if (r.end_reason == CILQR_END_BAD_INPUT) return fail(CILQR_ERR_OBSTACLE_HORIZON, "obstacle route shorter than tick + N + 1");
    return CILQR_OK;
}

extern "C" int cilqr_solve_cache_stats(cilqr_handle* h, int64_t* uploads, int64_t* reuses) {
    if (!h) return fail(CILQR_ERR_BAD_ARG, "null handle");
    if (uploads) *uploads = h->one.uploads;
    if (reuses) *reuses = h->one.reuses;
    return CILQR_OK;
}

// common prologue of the piecewise host-pointer entry points
struct Piece {
    Staged ids;
    BatchArgs a;
    int N = 0;
    size_t shm = 0;
};

static int piece_begin(cilqr_handle* h, int B, const int32_t* scenario_id, const int32_t* param_id,
                       const int32_t* tick, Piece& pc) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (B < 1) return fail(CILQR_ERR_BAD_ARG, "B < 1");
    if (four_rows(h)) return fail(CILQR_ERR_UNSUPPORTED, "the stage-by-stage entry points hold one or two rows per lane: horizons up to 127");
    rc = validate_ids(h, B, scenario_id, param_id, tick);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    h->cur = 0;
    rc = order_after_last_launch(h, h->stream); // (they share the handle's scratch areas and staging buffers)
    if (rc) return rc;
    rc = ensure_scratch(h, B);
    if (rc) return rc;
    rc = stage_ids(h, B, scenario_id, param_id, tick, pc.ids);
    if (rc) return rc;
    pc.a = make_args(h, B, pc.ids);
    pc.N = pc.a.N;
    pc.shm = lds_bytes(pc.N, pc.a.W, pc.a.alm, 1);
    return CILQR_OK;
}

extern "C" int cilqr_init_traj_batch(cilqr_handle* h, int32_t B, const double* x0,
                                     const int32_t* param_id, double* x_out) {
    Piece pc;
    int rc = piece_begin(h, B, nullptr, param_id, nullptr, pc);
    if (rc) return rc;
    if (!x0 || !x_out) return fail(CILQR_ERR_BAD_ARG, "null argument");
    const double* d_x0;
    rc = up(h, 3, x0, sizeof(double) * 4 * B, &d_x0); if (rc) return rc;
    void* d_x;
    const size_t nb_x = sizeof(double) * 4 * (pc.N + 1) * (size_t)B;
    rc = alloc_out(h, 6, nb_x, &d_x); if (rc) return rc;
    hipLaunchKernelGGL(k_init_traj, dim3(B), dim3(CILQR_WAVE), pc.shm, h->stream, pc.a, d_x0, static_cast<double*>(d_x));
    HIP_TRY(hipGetLastError());
    DL(6, x_out, nb_x);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CILQR_OK;
}

extern "C" int cilqr_ref_points_batch(cilqr_handle* h, int32_t B, const double* x,
                                      const int32_t* scenario_id, const int32_t* param_id,
                                      double* ref_out, int32_t* idx_out) {
    Piece pc;
    int rc = piece_begin(h, B, scenario_id, param_id, nullptr, pc);
    if (rc) return rc;
    if (!x || !ref_out) return fail(CILQR_ERR_BAD_ARG, "null argument");
    const int R = pc.N + 1;
    const double* d_x;
    rc = up(h, 3, x, sizeof(double) * 4 * R * (size_t)B, &d_x); if (rc) return rc;
    void *d_ref, *d_idx;
    rc = alloc_out(h, 5, sizeof(double) * 3 * R * (size_t)B, &d_ref); if (rc) return rc;
    rc = alloc_out(h, 6, sizeof(int32_t) * R * (size_t)B, &d_idx); if (rc) return rc;
    hipLaunchKernelGGL(k_ref_points, dim3(B), dim3(CILQR_WAVE), pc.shm, h->stream, pc.a, d_x,
                       static_cast<double*>(d_ref), static_cast<int32_t*>(d_idx));
    HIP_TRY(hipGetLastError());
    DL(5, ref_out, sizeof(double) * 3 * R * (size_t)B);
    if (idx_out) DL(6, idx_out, sizeof(int32_t) * R * (size_t)B);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CILQR_OK;
}

extern "C" int cilqr_total_cost_batch(cilqr_handle* h, int32_t B, const double* u, const double* x,
                                      const int32_t* scenario_id, const int32_t* param_id,
                                      const int32_t* tick, double* J_out) {
    Piece pc;
    int rc = piece_begin(h, B, scenario_id, param_id, tick, pc);
    if (rc) return rc;
    if (!u || !x || !J_out) return fail(CILQR_ERR_BAD_ARG, "null argument");
    const int N = pc.N, R = N + 1;
    const double *d_u, *d_x;
    rc = up(h, 3, u, sizeof(double) * 2 * N * (size_t)B, &d_u); if (rc) return rc;
    rc = up(h, 4, x, sizeof(double) * 4 * R * (size_t)B, &d_x); if (rc) return rc;
    void* d_J;
    rc = alloc_out(h, 5, sizeof(double) * B, &d_J); if (rc) return rc;
    hipLaunchKernelGGL((pc.a.alm ? k_total_cost<true> : k_total_cost<false>), dim3(B), dim3(CILQR_WAVE), pc.shm, h->stream, pc.a, d_u, d_x, static_cast<double*>(d_J));
    HIP_TRY(hipGetLastError());
    DL(5, J_out, sizeof(double) * B);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CILQR_OK;
}

extern "C" int cilqr_forward_pass_batch(cilqr_handle* h, int32_t B, const double* u, const double* x,
                                        const double* d, const double* K, const int32_t* scenario_id,
                                        const int32_t* param_id, const int32_t* tick, int32_t n_alpha,
                                        double* new_u, double* new_x, double* J_out) {
    Piece pc;
    int rc = piece_begin(h, B, scenario_id, param_id, tick, pc);
    if (rc) return rc;
    if (!u || !x || !d || !K || !new_u || !new_x) return fail(CILQR_ERR_BAD_ARG, "null argument");
    if (n_alpha < 1 || n_alpha > CILQR_MAX_ALPHA_TRIALS) return fail(CILQR_ERR_BAD_ARG, "n_alpha outside [1, 20]");
    const int N = pc.N, R = N + 1;
    const double *d_u, *d_x, *d_d, *d_K;
    rc = up(h, 3, u, sizeof(double) * 2 * N * (size_t)B, &d_u); if (rc) return rc;
    rc = up(h, 4, x, sizeof(double) * 4 * R * (size_t)B, &d_x); if (rc) return rc;
    rc = up(h, 5, d, sizeof(double) * 2 * N * (size_t)B, &d_d); if (rc) return rc;
    rc = up(h, 6, K, sizeof(double) * 8 * N * (size_t)B, &d_K); if (rc) return rc;
    void *d_nu, *d_nx, *d_J;
    const size_t nb_u = sizeof(double) * 2 * N * (size_t)B * n_alpha, nb_x = sizeof(double) * 4 * R * (size_t)B * n_alpha;
    rc = alloc_out(h, 7, nb_u, &d_nu); if (rc) return rc;
    rc = alloc_out(h, 8, nb_x, &d_nx); if (rc) return rc;
    rc = alloc_out(h, 9, sizeof(double) * (size_t)B * n_alpha, &d_J); if (rc) return rc;
    hipLaunchKernelGGL((pc.a.alm ? k_forward_pass<true> : k_forward_pass<false>), dim3(B), dim3(CILQR_WAVE), pc.shm, h->stream, pc.a, d_u, d_x, d_d, d_K,
                       n_alpha, static_cast<double*>(d_nu), static_cast<double*>(d_nx), static_cast<double*>(d_J));
    HIP_TRY(hipGetLastError());
    DL(7, new_u, nb_u);
    DL(8, new_x, nb_x);
    if (J_out) DL(9, J_out, sizeof(double) * (size_t)B * n_alpha);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CILQR_OK;
}

extern "C" int cilqr_cost_derivatives_batch(cilqr_handle* h, int32_t B, const double* u,
                                            const double* x, const int32_t* scenario_id,
                                            const int32_t* param_id, const int32_t* tick,
                                            double* l_x, double* l_u, double* l_xx, double* l_uu,
                                            double* A, double* Bm) {
    Piece pc;
    int rc = piece_begin(h, B, scenario_id, param_id, tick, pc);
    if (rc) return rc;
    if (!u || !x) return fail(CILQR_ERR_BAD_ARG, "null argument");
    const int N = pc.N, R = N + 1;
    const double *d_u, *d_x;
    rc = up(h, 3, u, sizeof(double) * 2 * N * (size_t)B, &d_u); if (rc) return rc;
    rc = up(h, 4, x, sizeof(double) * 4 * R * (size_t)B, &d_x); if (rc) return rc;
    double* host[6] = {l_x, l_u, l_xx, l_uu, A, Bm};
    const size_t nb[6] = {sizeof(double) * 4 * R * (size_t)B, sizeof(double) * 2 * N * (size_t)B,
                          sizeof(double) * 16 * R * (size_t)B, sizeof(double) * 4 * N * (size_t)B,
                          sizeof(double) * 16 * N * (size_t)B, sizeof(double) * 8 * N * (size_t)B};
    void* dev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 6; ++i) {
        if (!host[i]) continue;
        rc = alloc_out(h, 5 + i, nb[i], &dev[i]);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((pc.a.alm ? k_cost_derivatives<true> : k_cost_derivatives<false>), dim3(B), dim3(CILQR_WAVE), pc.shm, h->stream, pc.a, d_u, d_x,
                       static_cast<double*>(dev[0]), static_cast<double*>(dev[1]), static_cast<double*>(dev[2]),
                       static_cast<double*>(dev[3]), static_cast<double*>(dev[4]), static_cast<double*>(dev[5]));
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < 6; ++i)
        if (host[i]) DL(5 + i, host[i], nb[i]);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CILQR_OK;
}

extern "C" int cilqr_backward_pass_batch(cilqr_handle* h, int32_t B, const double* u, const double* x,
                                         const double* lamb, const int32_t* scenario_id,
                                         const int32_t* param_id, const int32_t* tick, double* d,
                                         double* K, double* dV, int32_t* status) {
    Piece pc;
    int rc = piece_begin(h, B, scenario_id, param_id, tick, pc);
    if (rc) return rc;
    if (!u || !x || !lamb || !d || !K || !dV || !status) return fail(CILQR_ERR_BAD_ARG, "null argument");
    const int N = pc.N, R = N + 1;
    const double *d_u, *d_x, *d_l;
    rc = up(h, 3, u, sizeof(double) * 2 * N * (size_t)B, &d_u); if (rc) return rc;
    rc = up(h, 4, x, sizeof(double) * 4 * R * (size_t)B, &d_x); if (rc) return rc;
    rc = up(h, 5, lamb, sizeof(double) * B, &d_l); if (rc) return rc;
    void *o_d, *o_K, *o_dV, *o_st;
    rc = alloc_out(h, 6, sizeof(double) * 2 * N * (size_t)B, &o_d); if (rc) return rc;
    rc = alloc_out(h, 7, sizeof(double) * 8 * N * (size_t)B, &o_K); if (rc) return rc;
    rc = alloc_out(h, 8, sizeof(double) * 2 * B, &o_dV); if (rc) return rc;
    rc = alloc_out(h, 9, sizeof(int32_t) * B, &o_st); if (rc) return rc;
    hipLaunchKernelGGL((pc.a.alm ? k_backward_pass<true> : k_backward_pass<false>), dim3(B), dim3(CILQR_WAVE), pc.shm, h->stream, pc.a, d_u, d_x, d_l,
                       static_cast<double*>(o_d), static_cast<double*>(o_K), static_cast<double*>(o_dV),
                       static_cast<int32_t*>(o_st));
    HIP_TRY(hipGetLastError());
    DL(6, d, sizeof(double) * 2 * N * (size_t)B);
    DL(7, K, sizeof(double) * 8 * N * (size_t)B);
    DL(8, dV, sizeof(double) * 2 * B);
    DL(9, status, sizeof(int32_t) * B);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CILQR_OK;
}

extern "C" int cilqr_detmath_eval(cilqr_handle* h, int32_t func, const double* x, const double* y,
                                  int32_t n, double* out) {
    if (!h || !x || !out || n < 1 || func < 0 || func > 10) return fail(CILQR_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const double *d_x, *d_y = nullptr;
    int rc = up(h, 3, x, sizeof(double) * n, &d_x); if (rc) return rc;
    if (y) { rc = up(h, 4, y, sizeof(double) * n, &d_y); if (rc) return rc; }
    void* d_o;
    rc = alloc_out(h, 5, sizeof(double) * n, &d_o); if (rc) return rc;
    hipLaunchKernelGGL(k_detmath, dim3((n + 255) / 256), dim3(256), 0, h->stream, func, d_x, d_y, static_cast<double*>(d_o), n);
    HIP_TRY(hipGetLastError());
    DL(5, out, sizeof(double) * n);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CILQR_OK;
}
