// scenario.cpp — host-side construction of the solve path's inputs (no GPU involved).
//
// What the reference does before the first solve() call:
//   * natural cubic spline through the lane way-points, 1-D and arc-length-parametrised 2-D
//     (/root/reference/src/cubic_spline.cpp:17-39, 130-157);
//   * ReferenceLine: samples of the spline every `accuracy` metres, laterally offset by `width`
//     (/root/reference/src/utils.cpp:21-35), ReferenceLine::calc_position (:60-67);
//   * obstacle "predictions": each vehicle advances along its nearest centre line at constant
//     speed (/root/reference/src/motion_planning.cpp:121-173).
// Re-designed here: the spline system is tri-diagonal, so it is solved with the Thomas algorithm
// instead of a dense column-pivoting QR (cubic_spline.cpp:29); everything else keeps the reference's
// arithmetic (accumulating `s += accuracy`, `t += delta_t`) because the sample count and the sample
// positions depend on it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/cilqr_amd.h"

namespace {

struct Spline1D {
    std::vector<double> x, a, b, c, d, h;
    bool ok = false;

    Spline1D() = default;
    Spline1D(const std::vector<double>& xs, const std::vector<double>& ys) : x(xs), a(ys) {
        const int nx = static_cast<int>(x.size());
        if (nx < 2 || ys.size() != xs.size()) return;
        h.resize(nx - 1);
        for (int i = 0; i < nx - 1; ++i) {
            h[i] = x[i + 1] - x[i];
            if (h[i] < 0) return;  // "x coordinates must be sorted in ascending order"
        }
        // rows 0 and nx-1 are identity rows with rhs 0 (natural spline); interior row i:
        //   h[i-1] c[i-1] + 2 (h[i-1] + h[i]) c[i] + h[i] c[i+1] = 3 (a[i+1]-a[i])/h[i] - 3 (a[i]-a[i-1])/h[i-1]
        std::vector<double> lo(nx, 0.0), di(nx, 1.0), up(nx, 0.0), rhs(nx, 0.0);
        for (int i = 1; i < nx - 1; ++i) {
            lo[i] = h[i - 1];
            di[i] = 2.0 * (h[i - 1] + h[i]);
            up[i] = h[i];
            rhs[i] = 3.0 * (a[i + 1] - a[i]) / h[i] - 3.0 * (a[i] - a[i - 1]) / h[i - 1];
        }
        for (int i = 1; i < nx; ++i) {  // forward elimination
            const double m = lo[i] / di[i - 1];
            di[i] -= m * up[i - 1];
            rhs[i] -= m * rhs[i - 1];
        }
        c.assign(nx, 0.0);
        c[nx - 1] = rhs[nx - 1] / di[nx - 1];
        for (int i = nx - 2; i >= 0; --i) c[i] = (rhs[i] - up[i] * c[i + 1]) / di[i];
        b.resize(nx - 1);
        d.resize(nx - 1);
        for (int i = 0; i < nx - 1; ++i) {
            d[i] = (c[i + 1] - c[i]) / (3.0 * h[i]);
            b[i] = (a[i + 1] - a[i]) / h[i] - h[i] * (c[i + 1] + 2 * c[i]) / 3.0;
        }
        ok = true;
    }

    // segment index as std::upper_bound(...) - 1; the reference reads one past its coefficient
    // arrays when _x == x.back() (cubic_spline.cpp:75-77) — the last segment is used here instead.
    int segment(double v) const {
        auto it = std::upper_bound(x.begin(), x.end(), v);
        int idx = static_cast<int>(it - x.begin()) - 1;
        const int last = static_cast<int>(x.size()) - 2;
        if (idx > last) idx = last;
        if (idx < 0) idx = 0;
        return idx;
    }
    bool in_range(double v) const { return !(v < x.front() || v > x.back()); }
    double position(double v) const {
        const int i = segment(v);
        const double dx = v - x[i];
        return a[i] + b[i] * dx + c[i] * std::pow(dx, 2) + d[i] * std::pow(dx, 3);
    }
    double first_derivative(double v) const {
        const int i = segment(v);
        const double dx = v - x[i];
        return b[i] + 2.0 * c[i] * dx + 3.0 * d[i] * std::pow(dx, 2);
    }
};

struct Spline2D {
    std::vector<double> s;
    Spline1D sx, sy;
    bool ok = false;

    Spline2D(const double* wx, const double* wy, int n) {
        if (n < 2) return;
        std::vector<double> xs(wx, wx + n), ys(wy, wy + n);
        s.assign(1, 0.0);
        double acc = 0.0;
        for (int i = 0; i + 1 < n; ++i) {
            const double ds = std::hypot(xs[i + 1] - xs[i], ys[i + 1] - ys[i]);
            acc = (i == 0) ? ds : acc + ds;
            s.push_back(acc);
        }
        sx = Spline1D(s, xs);
        sy = Spline1D(s, ys);
        ok = sx.ok && sy.ok;
    }
    double yaw(double v) const { return std::atan2(sy.first_derivative(v), sx.first_derivative(v)); }
};

// ReferenceLine::calc_position (utils.cpp:60-67)
void line_position(const Spline2D& sp, double width, double cur_s, double out[3]) {
    const double px = sp.sx.position(cur_s);
    const double py = sp.sy.position(cur_s);
    const double lyaw = sp.yaw(cur_s);
    out[0] = px - width * std::sin(lyaw);
    out[1] = py + width * std::cos(lyaw);
    out[2] = lyaw;
}

struct Line {
    std::vector<double> x, y, yaw, longitude;
};

// ReferenceLine ctor (utils.cpp:21-35)
Line sample_line(const Spline2D& sp, double width, double accuracy) {
    Line ln;
    for (double s = 0.0; s <= sp.s.back(); s += accuracy) {
        double p[3];
        line_position(sp, width, s, p);
        ln.x.push_back(p[0]);
        ln.y.push_back(p[1]);
        ln.yaw.push_back(p[2]);
        ln.longitude.push_back(s);
    }
    return ln;
}

}  // namespace

extern "C" int cilqr_reference_line_build(const double* wx, const double* wy, int32_t n, double width,
                                          double accuracy, double* x, double* y, double* yaw,
                                          double* s, int32_t cap, int32_t* count) {
    if (!wx || !wy || n < 2 || !(accuracy > 0) || !count) return CILQR_ERR_BAD_ARG;
    Spline2D sp(wx, wy, n);
    if (!sp.ok) return CILQR_ERR_BAD_ARG;
    Line ln = sample_line(sp, width, accuracy);
    const int32_t total = static_cast<int32_t>(ln.x.size());
    *count = total;
    const int32_t m = std::min(total, cap);
    for (int32_t i = 0; i < m; ++i) {
        if (x) x[i] = ln.x[i];
        if (y) y[i] = ln.y[i];
        if (yaw) yaw[i] = ln.yaw[i];
        if (s) s[i] = ln.longitude[i];
    }
    return CILQR_OK;
}

extern "C" int cilqr_reference_line_position(const double* wx, const double* wy, int32_t n,
                                             double width, double cur_s, double out[3]) {
    if (!wx || !wy || n < 2 || !out) return CILQR_ERR_BAD_ARG;
    Spline2D sp(wx, wy, n);
    if (!sp.ok || !sp.sx.in_range(cur_s)) return CILQR_ERR_BAD_ARG;
    line_position(sp, width, cur_s, out);
    return CILQR_OK;
}

namespace {
// Where a vehicle stands on the road (mp:122-143): on every centre line the distance to the vehicle is followed sample by
// sample from the line's start until it first grows — the sample before that is the line's candidate (a line whose distance
// never grows has none) — and the vehicle is put on the line with the closest candidate; ties keep the earlier line.
struct Foothold {
    size_t lane;
    double s;
};

inline bool first_distance_minimum(const Line& ln, double px, double py, double* dist, double* s_at) {
    if (ln.x.empty()) return false;
    double before = std::hypot(ln.x[0] - px, ln.y[0] - py);
    for (size_t i = 1; i < ln.x.size(); ++i) {
        const double here = std::hypot(ln.x[i] - px, ln.y[i] - py);
        if (here > before) {
            *dist = before;
            *s_at = ln.longitude[i - 1];
            return true;
        }
        before = here;
    }
    return false;
}

inline Foothold nearest_foothold(const std::vector<Line>& lanes, double px, double py, double s_if_none) {
    Foothold best{0, s_if_none};
    bool have = false;
    double best_dist = 0.0;
    for (size_t l = 0; l < lanes.size(); ++l) {
        double dist, s_at;
        if (first_distance_minimum(lanes[l], px, py, &dist, &s_at) && (!have || dist < best_dist)) {
            have = true;
            best_dist = dist;
            best = Foothold{l, s_at};
        }
    }
    return best;
}
}  // namespace

extern "C" int cilqr_build_routes(const double* wx, const double* wy, int32_t n,
                                  const double* center_widths, int32_t n_center, double accuracy,
                                  const double* init_cond, int32_t V, double max_simulation_time,
                                  double delta_t, double* routes, int32_t T_cap, int32_t* T_out,
                                  int32_t* line_num_out, double* start_s_out) {
    if (!wx || !wy || n < 2 || !center_widths || n_center < 1 || !init_cond || V < 1 ||
        !(delta_t > 0) || !T_out)
        return CILQR_ERR_BAD_ARG;
    Spline2D sp(wx, wy, n);
    if (!sp.ok) return CILQR_ERR_BAD_ARG;
    std::vector<Line> center_lines;
    for (int l = 0; l < n_center; ++l) center_lines.push_back(sample_line(sp, center_widths[l], accuracy));

    // the sampling instants, accumulated exactly as upstream accumulates them (mp:146: `t += delta_t`)
    std::vector<double> instants;
    for (double t = 0.0; t < max_simulation_time + 10; t += delta_t) instants.push_back(t);
    const int32_t T = static_cast<int32_t>(instants.size());
    *T_out = T;
    if (!routes) return CILQR_OK;
    if (T_cap < T) return CILQR_ERR_BAD_ARG;

    for (int v = 0; v < V; ++v) {
        const double* start = init_cond + v * 4;  // (x, y, speed, yaw)
        const Foothold fh = nearest_foothold(center_lines, start[0], start[1], sp.s.back());
        if (line_num_out) line_num_out[v] = static_cast<int32_t>(fh.lane);
        if (start_s_out) start_s_out[v] = fh.s;
        const Line& lane = center_lines[fh.lane];
        // a lane has no driving direction upstream: it is read off the initial yaw (mp:150-158).  Travelling against the
        // lane = arc length running down, clamped at the lane's first sample, heading turned by pi
        const bool with_lane = start[3] <= M_PI_2;
        const double dir = with_lane ? 1.0 : -1.0;
        const double s_stop = with_lane ? lane.longitude.back() : lane.longitude.front();
        double* route = routes + static_cast<size_t>(v) * T_cap * 3;
        for (size_t k = 0; k < instants.size(); ++k, route += 3) {
            const double s_free = fh.s + dir * (instants[k] * start[2]);  // +-1 x p is exact: == start_s -+ t * speed
            const double s_at = with_lane ? std::min(s_free, s_stop) : std::max(s_free, s_stop);
            line_position(sp, center_widths[fh.lane], s_at, route);
            if (!with_lane) route[2] = std::fmod(route[2] + M_PI, 2 * M_PI);
        }
    }
    return CILQR_OK;
}

// ------------------------------------------------------------------------------------------------
// The benchmark workloads' initial states: counter-based generator — splitmix64 of
// (seed ^ splitmix64(counter)) -> uniform in (0, 1) -> Box-Muller — so that any implementation regenerates row b of
// a batch from (seed, b) alone.  Twin of toy-example-of-ilqr_amd/workloads.py::perturbed_starts (counter = 8 b + i,
// i = 0..4): x0_b = base + (U(-5, 5), +-U(0.05, 1.0), N(0, 0.5), N(0, 0.02)).  The uniform components are bit-identical
// to the Python twin; the normal ones go through log / cos / sqrt of the platform's libm and may differ from
// numpy's in the last place.
namespace {
inline uint64_t splitmix64(uint64_t v) {
    v += 0x9E3779B97F4A7C15ULL;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ULL;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBULL;
    return v ^ (v >> 31);
}
inline double uniform01(uint64_t seed, uint64_t counter) {
    const uint64_t r = splitmix64(seed ^ splitmix64(counter));
    return (static_cast<double>(r >> 11) + 0.5) / 9007199254740992.0;  // 2^53
}
inline double normal01(uint64_t seed, uint64_t counter) {
    const double u1 = uniform01(seed, 2 * counter), u2 = uniform01(seed, 2 * counter + 1);
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
}
}  // namespace

extern "C" int cilqr_perturbed_starts(const double base[4], int32_t B, uint64_t seed, int64_t first, double* x0_out) {
    if (!base || !x0_out || B < 1 || first < 0) return CILQR_ERR_BAD_ARG;
    for (int32_t i = 0; i < B; ++i) {
        const uint64_t b = static_cast<uint64_t>(first) + static_cast<uint64_t>(i);
        const double dx = -5.0 + 10.0 * uniform01(seed, 8 * b + 0);
        const double mag = 0.05 + 0.95 * uniform01(seed, 8 * b + 1);
        const double sgn = uniform01(seed, 8 * b + 2) < 0.5 ? 1.0 : -1.0;
        const double dv = 0.5 * normal01(seed, 8 * b + 3);
        const double dyaw = 0.02 * normal01(seed, 8 * b + 4);
        double* o = x0_out + static_cast<size_t>(i) * 4;
        o[0] = base[0] + dx;
        o[1] = base[1] + sgn * mag;
        o[2] = base[2] + dv;
        o[3] = base[3] + dyaw;
    }
    return CILQR_OK;
}
