// headless_planner.cpp — the reference's closed planning loop without the drawing:
// /root/reference/src/motion_planning.cpp:52-197 (load config, build centre lines and borders, fabricate the
// noise-free obstacle routes, then per tick: solve, ego_state = x.row(1)).  Host C++ on top of the
// C-ABI only (include/cilqr_amd.h via include/cilqr_solver_shim.hpp).
//
//   headless_planner <scenario.json> [ticks] [N]
// prints one line per tick: tick index, ego state, first control, iterations, J_final.
//
//   headless_planner <scenario.json> --batch B [--devices G] [--share] [--horizon N]
// the batch form of the first tick: B perturbed egos of the scenario (cilqr_perturbed_starts), one cold solve each, the batch
// sharded over G devices by cilqr_amd::ShardedSolver (contiguous blocks, tables replicated, statistics summed on the host:
// SURVEY.md 8(e) without torch; G = 0: every visible device; --share: shards share the visible devices, a rehearsal).
// Prints the summed statistics and a checksum over every output bit — the same for every G.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "cilqr_config.hpp"
#include "cilqr_solver_shim.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <scenario.json> [ticks] [N]\n", argv[0]);
        return 2;
    }
    using namespace cilqr_amd;
    FlatConfig cfg = FlatConfig::load(argv[1]);
    const double delta_t = cfg.get_config<double>("delta_t");
    const double max_simulation_time = cfg.get_config<double>("max_simulation_time");
    const double target_velocity = cfg.get_config<double>("vehicle/target_velocity");
    std::vector<double> rx = cfg.get_config<std::vector<double>>("laneline/reference/x");
    std::vector<double> ry = cfg.get_config<std::vector<double>>("laneline/reference/y");
    std::vector<double> border_widths = cfg.get_config<std::vector<double>>("laneline/border");
    std::vector<double> center_widths = cfg.get_config<std::vector<double>>("laneline/center_line");
    std::vector<std::vector<double>> init = cfg.get_config<std::vector<std::vector<double>>>("initial_condition");
    const int V = static_cast<int>(init.size());
    const int n = static_cast<int>(rx.size());

    // centre line 0 (mp:97-100)
    int L = 0;
    cilqr_reference_line_build(rx.data(), ry.data(), n, center_widths[0], 0.1, nullptr, nullptr, nullptr, nullptr, 0, &L);
    std::vector<double> lx(L), ly(L), lyaw(L), ls(L);
    cilqr_reference_line_build(rx.data(), ry.data(), n, center_widths[0], 0.1, lx.data(), ly.data(), lyaw.data(), ls.data(), L, &L);
    // road_borders = (max, min) (mp:101-103)
    std::sort(border_widths.begin(), border_widths.end(), std::greater<double>());
    const double road_borders[2] = {border_widths.front(), border_widths.back()};
    // routes (mp:121-173, noise disabled)
    std::vector<double> ic(static_cast<size_t>(V) * 4);
    for (int v = 0; v < V; ++v)
        for (int c = 0; c < 4; ++c) ic[v * 4 + c] = init[v][c];
    int T = 0;
    cilqr_build_routes(rx.data(), ry.data(), n, center_widths.data(), static_cast<int>(center_widths.size()), 0.1, ic.data(), V,
                       max_simulation_time, delta_t, nullptr, 0, &T, nullptr, nullptr);
    std::vector<double> routes(static_cast<size_t>(V) * T * 3);
    cilqr_build_routes(rx.data(), ry.data(), n, center_widths.data(), static_cast<int>(center_widths.size()), 0.1, ic.data(), V,
                       max_simulation_time, delta_t, routes.data(), T, &T, nullptr, nullptr);

    cilqr_params p = params_from_config(cfg);
    // ---- batch mode: B egos of this scenario, sharded over the devices ----
    long long batch = 0;
    int devices = 0, share = 0, horizon = 0;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--batch" && i + 1 < argc) batch = std::atoll(argv[++i]);
        else if (a == "--devices" && i + 1 < argc) devices = std::atoi(argv[++i]);
        else if (a == "--horizon" && i + 1 < argc) horizon = std::atoi(argv[++i]);
        else if (a == "--share") share = 1;
    }
    if (batch > 0) {
        if (horizon > 0) p.N = horizon;
        p.use_last_solution = 0;
        const int Nb = p.N, Mb = V - 1;
        cilqr_scenario_desc sc{};
        sc.lane_x = lx.data(); sc.lane_y = ly.data(); sc.lane_yaw = lyaw.data(); sc.L = L;
        sc.M = Mb; sc.obs = routes.data() + static_cast<size_t>(T) * 3; sc.T = T; // (vehicle 0 is the ego)
        sc.road_borders[0] = road_borders[0]; sc.road_borders[1] = road_borders[1];
        sc.ref_velo = target_velocity;
        std::vector<double> x0(static_cast<size_t>(batch) * 4), ub(static_cast<size_t>(batch) * 2 * Nb), xb(static_cast<size_t>(batch) * 4 * (Nb + 1));
        std::vector<cilqr_result> res(static_cast<size_t>(batch));
        const double base[4] = {init[0][0], init[0][1], init[0][2], init[0][3]};
        cilqr_perturbed_starts(base, static_cast<int32_t>(batch), 0xC11A0B5ULL, 0, x0.data());
        ShardedSolver sh(&p, 1, &sc, 1, devices, share != 0);
        const ShardStats st = sh.solve_batch(batch, x0.data(), nullptr, nullptr, nullptr, nullptr, ub.data(), xb.data(), res.data());
        unsigned long long hsh = 1469598103934665603ULL; // FNV-1a over every output bit
        auto mix = [&hsh](const void* ptr, size_t n) {
            const unsigned char* c = static_cast<const unsigned char*>(ptr);
            for (size_t i = 0; i < n; ++i) { hsh ^= c[i]; hsh *= 1099511628211ULL; }
        };
        mix(ub.data(), ub.size() * sizeof(double));
        mix(xb.data(), xb.size() * sizeof(double));
        for (const auto& r : res) { mix(&r.J_final, sizeof(double)); mix(&r.iters, sizeof(int)); mix(&r.ls_trials, sizeof(int)); mix(&r.end_reason, sizeof(int)); }
        std::printf("batch %lld horizon %d devices %d iters %lld ls_trials %lld converged %lld max_lamb %lld max_iter %lld bad_input %lld not_solved %lld nan %lld "
                    "sum_J_final %.17g checksum %016llx\n", batch, Nb, sh.devices(), st.iters, st.ls_trials, st.converged, st.max_lamb,
                    st.max_iter, st.bad_input, st.not_solved, st.nan_costs, st.sum_J_final, hsh);
        return 0;
    }
    if (argc > 3) p.N = std::atoi(argv[3]);
    CILQRSolver solver(p);
    const int N = p.N, M = V - 1;
    int ticks = (argc > 2) ? std::atoi(argv[2]) : 1000000;
    std::vector<double> u(2 * N), x(4 * (N + 1));
    double ego[4] = {init[0][0], init[0][1], init[0][2], init[0][3]};
    int done = 0;
    for (double t = 0.; t < max_simulation_time && done < ticks; t += delta_t, ++done) {
        const size_t index = static_cast<size_t>(t / delta_t);  // mp:181
        // utils::get_sub_routing_lines(obs_prediction, index) (utils.cpp:88-103): the tail of every route
        const int Tsub = T - static_cast<int>(index);
        std::vector<double> obs(static_cast<size_t>(M) * Tsub * 3);
        for (int j = 0; j < M; ++j)
            std::copy(routes.begin() + (static_cast<size_t>(j + 1) * T + index) * 3,
                      routes.begin() + (static_cast<size_t>(j + 1) * T + T) * 3, obs.begin() + static_cast<size_t>(j) * Tsub * 3);
        cilqr_result res{};
        solver.solve(ego, lx.data(), ly.data(), lyaw.data(), L, target_velocity, obs.data(), M, Tsub, road_borders, u.data(),
                     x.data(), &res);
        for (int c = 0; c < 4; ++c) ego[c] = x[4 + c];  // ego_state = new_x.row(1) (mp:197)
        std::printf("%zu %.17g %.17g %.17g %.17g %.17g %.17g %d %.17g\n", index, ego[0], ego[1], ego[2], ego[3], u[0], u[1],
                    res.iters, res.J_final);
    }
    return 0;
}
