// cilqr_solver_shim.hpp — source-level drop-in for the reference's CILQRSolver on top of the C-ABI.
//
// Mirrors /root/reference/include/cilqr_solver.hpp:31-41:
//     explicit CILQRSolver(const GlobalConfig* config);
//     std::tuple<Eigen::MatrixX2d, Eigen::MatrixX4d> solve(const Eigen::Vector4d& x0,
//         const ReferenceLine& ref_waypoints, double ref_velo,
//         const std::vector<RoutingLine>& obs_preds, const Eigen::Vector2d& road_boaders);
// The Eigen-typed overload is compiled only when <Eigen/Core> is available (it is not in the build
// container: tests/eigen_standin/ holds a minimal stand-in with which tests/test_cabi.py compiles and links this
// overload against a caller shaped like the reference's main()); the plain-array overload below it is what the
// overload forwards to.  The config type
// is a template parameter: anything with `template<class T> T get_config(const std::string&) const`
// (the reference's GlobalConfig, include/global_config.hpp:30-36) works unchanged.
//
// State carried across calls exactly as upstream: is_first_solve / last_solve_u for
// use_last_solution (src/cilqr_solver.cpp:97-102,144).  One instance = one handle = one GPU stream;
// like the reference class it is not re-entrant.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "cilqr_amd.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define CILQR_SHIM_HAS_EIGEN 1
#endif
#endif

namespace cilqr_amd {

template <class Config>
inline cilqr_params params_from_config(const Config& cfg) {
    cilqr_params p{};
    p.N = cfg.template get_config<int>("lqr/N");
    p.max_iter = cfg.template get_config<int>("iteration/max_iter");
    p.solve_type = cfg.template get_config<std::string>("lqr/slove_type") == "alm" ? 1 : 0;
    p.reference_point = cfg.template get_config<std::string>("vehicle/reference_point") == "rear_center" ? 0 : 1;
    p.use_last_solution = cfg.template get_config<bool>("lqr/use_last_solution") ? 1 : 0;
    p.dt = cfg.template get_config<double>("delta_t");
    p.w_pos = cfg.template get_config<double>("lqr/w_pos");
    p.w_vel = cfg.template get_config<double>("lqr/w_vel");
    p.w_yaw = cfg.template get_config<double>("lqr/w_yaw");
    p.w_acc = cfg.template get_config<double>("lqr/w_acc");
    p.w_stl = cfg.template get_config<double>("lqr/w_stl");
    p.obstacle_exp_q1 = cfg.template get_config<double>("lqr/obstacle_exp_q1");
    p.obstacle_exp_q2 = cfg.template get_config<double>("lqr/obstacle_exp_q2");
    p.state_exp_q1 = cfg.template get_config<double>("lqr/state_exp_q1");
    p.state_exp_q2 = cfg.template get_config<double>("lqr/state_exp_q2");
    p.alm_rho_init = cfg.template get_config<double>("lqr/alm_rho_init");
    p.alm_gamma = cfg.template get_config<double>("lqr/alm_gamma");
    p.max_rho = cfg.template get_config<double>("lqr/max_rho");
    p.max_mu = cfg.template get_config<double>("lqr/max_mu");
    p.init_lamb = cfg.template get_config<double>("iteration/init_lamb");
    p.lamb_decay = cfg.template get_config<double>("iteration/lamb_decay");
    p.lamb_amplify = cfg.template get_config<double>("iteration/lamb_amplify");
    p.max_lamb = cfg.template get_config<double>("iteration/max_lamb");
    p.convergence_threshold = cfg.template get_config<double>("iteration/convergence_threshold");
    p.accept_step_threshold = cfg.template get_config<double>("iteration/accept_step_threshold");
    p.wheelbase = cfg.template get_config<double>("vehicle/wheelbase");
    p.width = cfg.template get_config<double>("vehicle/width");
    p.length = cfg.template get_config<double>("vehicle/length");
    p.velo_max = cfg.template get_config<double>("vehicle/velo_max");
    p.velo_min = cfg.template get_config<double>("vehicle/velo_min");
    p.yaw_lim = cfg.template get_config<double>("vehicle/yaw_lim");
    p.acc_max = cfg.template get_config<double>("vehicle/acc_max");
    p.acc_min = cfg.template get_config<double>("vehicle/acc_min");
    p.stl_lim = cfg.template get_config<double>("vehicle/stl_lim");
    p.d_safe = cfg.template get_config<double>("vehicle/d_safe");
    return p;
}

class CILQRSolver {
  public:
    CILQRSolver() = delete;
    template <class Config>
    explicit CILQRSolver(const Config* config, int device = 0) : CILQRSolver(params_from_config(*config), device) {}
    explicit CILQRSolver(const cilqr_params& p, int device = 0) : params_(p) {
        check(cilqr_create(device, &h_), "cilqr_create");
        check(cilqr_set_params(h_, &params_, 1), "cilqr_set_params");
    }
    ~CILQRSolver() { cilqr_destroy(h_); }
    CILQRSolver(const CILQRSolver&) = delete;
    CILQRSolver& operator=(const CILQRSolver&) = delete;

    // Plain-array form.  lane_x/y/yaw[L] = ref_waypoints.x/.y/.yaw; obs[M][T][3] = obs_preds[j][k]
    // from the current tick on (T >= N + 1); u_out[N][2], x_out[N+1][4].
    void solve(const double x0[4], const double* lane_x, const double* lane_y, const double* lane_yaw, int L,
               double ref_velo, const double* obs, int M, int T, const double road_boaders[2], double* u_out,
               double* x_out, cilqr_result* res = nullptr) {
        cilqr_scenario_desc sc{};
        sc.lane_x = lane_x; sc.lane_y = lane_y; sc.lane_yaw = lane_yaw; sc.L = L;
        sc.M = M; sc.obs = obs; sc.T = T;
        sc.road_borders[0] = road_boaders[0]; sc.road_borders[1] = road_boaders[1];
        sc.ref_velo = ref_velo;
        // the tables stay resident in HBM from tick to tick (cilqr_solve uploads only what changed)
        const bool warm = !is_first_solve_ && params_.use_last_solution;
        check(cilqr_solve(h_, x0, &sc, warm ? last_solve_u_.data() : nullptr, u_out, x_out, res), "cilqr_solve");
        is_first_solve_ = false;
        last_solve_u_.assign(u_out, u_out + 2 * params_.N);
    }

#ifdef CILQR_SHIM_HAS_EIGEN
    // Eigen form with the reference's signature; ReferenceLine / RoutingLine are duck-typed
    // (members x, y, yaw as std::vector<double>), so the reference's own classes fit.
    template <class ReferenceLineT, class RoutingLineT>
    std::tuple<Eigen::MatrixX2d, Eigen::MatrixX4d> solve(const Eigen::Vector4d& x0, const ReferenceLineT& ref_waypoints,
                                                         double ref_velo, const std::vector<RoutingLineT>& obs_preds,
                                                         const Eigen::Vector2d& road_boaders) {
        const int N = params_.N, M = static_cast<int>(obs_preds.size());
        int T = 0;
        for (const auto& r : obs_preds) {
            const int n = static_cast<int>(std::min(r.x.size(), std::min(r.y.size(), r.yaw.size())));
            T = (T == 0 || n < T) ? n : T;
        }
        std::vector<double> obs(static_cast<size_t>(M) * T * 3);
        for (int j = 0; j < M; ++j)
            for (int k = 0; k < T; ++k) {
                obs[(static_cast<size_t>(j) * T + k) * 3 + 0] = obs_preds[j].x[k];
                obs[(static_cast<size_t>(j) * T + k) * 3 + 1] = obs_preds[j].y[k];
                obs[(static_cast<size_t>(j) * T + k) * 3 + 2] = obs_preds[j].yaw[k];
            }
        std::vector<double> u(2 * N), x(4 * (N + 1));
        const double xs[4] = {x0[0], x0[1], x0[2], x0[3]};
        const double rb[2] = {road_boaders[0], road_boaders[1]};
        solve(xs, ref_waypoints.x.data(), ref_waypoints.y.data(), ref_waypoints.yaw.data(),
              static_cast<int>(ref_waypoints.x.size()), ref_velo, obs.data(), M, T, rb, u.data(), x.data());
        Eigen::MatrixX2d U(N, 2);
        Eigen::MatrixX4d X(N + 1, 4);
        for (int k = 0; k < N; ++k) { U(k, 0) = u[2 * k]; U(k, 1) = u[2 * k + 1]; }
        for (int k = 0; k <= N; ++k)
            for (int c = 0; c < 4; ++c) X(k, c) = x[4 * k + c];
        return std::make_tuple(U, X);
    }
#endif

    const cilqr_params& params() const { return params_; }

  private:
    static void check(int rc, const char* where) {
        if (rc != CILQR_OK) throw std::runtime_error(std::string(where) + ": " + cilqr_last_error());
    }
    cilqr_params params_;
    cilqr_handle* h_ = nullptr;
    bool is_first_solve_ = true;
    std::vector<double> last_solve_u_;
};

// ---------------------------------------------------------------------------------------------------------------------
// ShardedSolver — a batch over several GPUs from ONE host process, without torch (SURVEY.md 8(e) as a product feature; the
// benchmark's `torchrun bench.py --gpus N` is the other way to get there: one process per GPU).  The path shards by
// trajectory with no exchange step: device g of G solves the contiguous block [first_g, first_g + count_g) of the batch —
// ceil(B / G) trajectories each, the last device the remainder — on a handle of its own (cilqr_create(device)), the
// parameter and scenario tables replicated to every device; results come back into the caller's arrays in place; the only
// "collective" is the sum of eight statistics on the host.  One host thread per device drives its handle (a handle is not
// thread-safe; cilqr_last_error is per thread), so the devices run concurrently.  Results do not depend on G: a trajectory's
// solve is a function of its own inputs alone (tests: --devices 1 equals the plain call; shard arithmetic on the CPU).
struct ShardStats {
    long long trajectories = 0, iters = 0, ls_trials = 0, converged = 0, max_lamb = 0, max_iter = 0, bad_input = 0, not_solved = 0, nan_costs = 0;
    double sum_J_final = 0.0;
};

class ShardedSolver {
  public:
    // devices = 0: every visible device (cilqr_device_count).  share_devices: shard g runs on device g mod (visible devices) —
    // a REHEARSAL of G > 1 shards on a box with fewer GPUs (the shards' launches then share a GPU), not a way to go faster
    ShardedSolver(const cilqr_params* params, int n_params, const cilqr_scenario_desc* scen, int n_scen, int devices = 0,
                  bool share_devices = false) {
        int visible = 0;
        check(cilqr_device_count(&visible), "cilqr_device_count");
        const int G = devices > 0 ? devices : visible;
        if (G < 1 || visible < 1 || (G > visible && !share_devices))
            throw std::runtime_error("ShardedSolver: asked for " + std::to_string(G) + " devices, " + std::to_string(visible) + " visible");
        N_ = params[0].N;
        for (int g = 0; g < G; ++g) {
            cilqr_handle* h = nullptr;
            check(cilqr_create(g % visible, &h), "cilqr_create");
            handles_.emplace_back(h, &cilqr_destroy);
            check(cilqr_set_params(h, params, n_params), "cilqr_set_params");
            check(cilqr_set_scenarios(h, scen, n_scen), "cilqr_set_scenarios");
        }
    }
    int devices() const { return static_cast<int>(handles_.size()); }

    // trajectories [first, first + count) of a batch of B go to device g of G: contiguous blocks of ceil(B / G)
    static void shard_bounds(long long B, int G, int g, long long* first, long long* count) {
        const long long per = (B + G - 1) / G;
        const long long f = std::min<long long>(B, per * g);
        *first = f;
        *count = std::max<long long>(0, std::min<long long>(B, f + per) - f);
    }

    // cilqr_solve_batch for the whole batch (host arrays; any of scenario_id / param_id / tick / last_u / res may be null)
    ShardStats solve_batch(long long B, const double* x0, const int32_t* scenario_id, const int32_t* param_id, const int32_t* tick,
                           const double* last_u, double* u_out, double* x_out, cilqr_result* res_out) {
        const int G = devices(), N = N_;
        std::vector<cilqr_result> res_local;
        if (!res_out) { res_local.resize(static_cast<size_t>(B)); res_out = res_local.data(); }
        std::vector<std::string> errors(G);
        std::vector<std::thread> th;
        for (int g = 0; g < G; ++g) {
            long long first = 0, count = 0;
            shard_bounds(B, G, g, &first, &count);
            if (count <= 0) continue;
            th.emplace_back([=, &errors]() {
                const int rc = cilqr_solve_batch(handles_[g].get(), static_cast<int32_t>(count), x0 + 4 * first,
                                                 scenario_id ? scenario_id + first : nullptr, param_id ? param_id + first : nullptr,
                                                 tick ? tick + first : nullptr, last_u ? last_u + 2 * N * first : nullptr,
                                                 u_out + 2 * N * first, x_out + 4 * (N + 1) * first, res_out + first, nullptr, 0);
                if (rc != CILQR_OK) errors[g] = std::string("device ") + std::to_string(g) + ": " + cilqr_last_error();
            });
        }
        for (auto& t : th) t.join();
        for (const auto& e : errors)
            if (!e.empty()) throw std::runtime_error("ShardedSolver::solve_batch: " + e);
        ShardStats s;
        s.trajectories = B;
        for (long long b = 0; b < B; ++b) { // the sum a multi-process run all-reduces (stats.py): here on the host
            const cilqr_result& r = res_out[b];
            s.iters += r.iters; s.ls_trials += r.ls_trials;
            s.converged += r.end_reason == CILQR_END_CONVERGED; s.max_lamb += r.end_reason == CILQR_END_MAX_LAMB;
            s.max_iter += r.end_reason == CILQR_END_MAX_ITER; s.bad_input += r.end_reason == CILQR_END_BAD_INPUT;
            s.not_solved += r.end_reason == CILQR_END_NOT_SOLVED; // (a trajectory lost in transit: the call has thrown already, cilqr_amd.h)
            if (std::isnan(r.J_final)) s.nan_costs += 1; else s.sum_J_final += r.J_final;
        }
        return s;
    }

  private:
    static void check(int rc, const char* where) {
        if (rc != CILQR_OK) throw std::runtime_error(std::string(where) + ": " + cilqr_last_error());
    }
    std::vector<std::unique_ptr<cilqr_handle, int (*)(cilqr_handle*)>> handles_;
    int N_ = 0;
};

}  // namespace cilqr_amd
