// AddressSanitizer / UndefinedBehaviorSanitizer job for the host-side C++ of the product that needs no GPU:
// csrc/scenario.cpp (spline, lane sampling, route fabrication, benchmark start generator) and
// include/cilqr_config.hpp (flattened JSON and YAML readers).  TEST INFRASTRUCTURE, built by
// tests/test_sanitizers.py with -fsanitize=address,undefined -fno-sanitize-recover=all.
//   host_main <scenario.json> [scenario.yaml]
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "cilqr_amd.h"
#include "cilqr_config.hpp"
#include "cilqr_solver_shim.hpp"  // params_from_config (nothing of the GPU library is called here)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    using namespace cilqr_amd;
    FlatConfig cfg = FlatConfig::load(argv[1]);
    std::vector<double> rx = cfg.get_config<std::vector<double>>("laneline/reference/x");
    std::vector<double> ry = cfg.get_config<std::vector<double>>("laneline/reference/y");
    std::vector<double> centers = cfg.get_config<std::vector<double>>("laneline/center_line");
    std::vector<std::vector<double>> init = cfg.get_config<std::vector<std::vector<double>>>("initial_condition");
    const int n = static_cast<int>(rx.size()), V = static_cast<int>(init.size());
    int L = 0;
    if (cilqr_reference_line_build(rx.data(), ry.data(), n, centers[0], 0.1, nullptr, nullptr, nullptr, nullptr, 0, &L)) return 3;
    std::vector<double> x(L), y(L), yaw(L), s(L);
    if (cilqr_reference_line_build(rx.data(), ry.data(), n, centers[0], 0.1, x.data(), y.data(), yaw.data(), s.data(), L, &L)) return 3;
    // a capacity smaller than the sample count: only `cap` samples may be written
    std::vector<double> xs(10), ys(10), yaws(10), ss(10);
    int L2 = 0;
    cilqr_reference_line_build(rx.data(), ry.data(), n, centers[0], 0.1, xs.data(), ys.data(), yaws.data(), ss.data(), 10, &L2);
    double pos[3];
    if (cilqr_reference_line_position(rx.data(), ry.data(), n, centers[0], 0.5 * s.back(), pos)) return 4;
    (void)cilqr_reference_line_position(rx.data(), ry.data(), n, centers[0], s.back() + 100.0, pos);  // out of range: refused
    std::vector<double> ic(static_cast<size_t>(V) * 4);
    for (int v = 0; v < V; ++v)
        for (int c = 0; c < 4; ++c) ic[v * 4 + c] = init[v][c];
    int T = 0;
    const double tmax = cfg.get_config<double>("max_simulation_time"), dt = cfg.get_config<double>("delta_t");
    cilqr_build_routes(rx.data(), ry.data(), n, centers.data(), static_cast<int>(centers.size()), 0.1, ic.data(), V, tmax, dt,
                       nullptr, 0, &T, nullptr, nullptr);
    std::vector<double> routes(static_cast<size_t>(V) * T * 3);
    std::vector<int32_t> line_num(V);
    std::vector<double> start_s(V);
    if (cilqr_build_routes(rx.data(), ry.data(), n, centers.data(), static_cast<int>(centers.size()), 0.1, ic.data(), V, tmax, dt,
                           routes.data(), T, &T, line_num.data(), start_s.data()))
        return 5;
    cilqr_params p = params_from_config(cfg);
    std::vector<double> x0(4 * 1000);
    if (cilqr_perturbed_starts(ic.data(), 1000, 0xC11A0002ULL, 12345, x0.data())) return 6;
    double acc = 0.0;
    for (double v : x0) acc += v;
    for (double v : routes) acc += v;
    if (!(acc == acc)) return 7;
    int yaml_N = -1;
    if (argc > 2) {
        FlatConfig y2 = FlatConfig::load(argv[2]);
        yaml_N = y2.get_config<int>("lqr/N");
        if (yaml_N != p.N) return 8;
    }
    std::printf("SANITIZE-HOST-OK L=%d T=%d V=%d N=%d yaml_N=%d\n", L, T, V, p.N, yaml_N);
    return 0;
}
