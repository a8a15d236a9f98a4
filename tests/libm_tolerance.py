"""Diagnosis of the trajectories on which the HIP path (bit-identical to the oracle's detmath build) and the
oracle's glibc-libm build — the stand-in for what the reference binary links — end up more than 1e-5 apart.

TEST INFRASTRUCTURE (imports oracle/).  Used by tests/test_gpu_parity.py::test_libm_divergences_are_near_ties
and, as a script, to write profiles/r02_libm_tolerance.json:

    python tests/libm_tolerance.py [--configs 2 3 5] [--gpu] [--out profiles/r02_libm_tolerance.json]

The solve loop takes discrete decisions — line-search verdicts (cs:358-365), the Cholesky test of Q_uu
(cs:415-416), `cur < min_distance` in the lane scan (cs:300) — on floating-point numbers.  Two correct
implementations whose elementary functions differ in the last place can take a different branch where such a
comparison is a near-tie, and from there on they optimise along different paths.  For every trajectory outside
the 1e-5 band this module finds the first iteration at which the two decision traces part and reports
  (1) the decision margin there: the smallest relative distance to flipping among the comparisons evaluated in
      that iteration by either build (orc_margin_rec);
  (2) whether the libm build ITSELF changes its decision trace when x0 is moved by one unit in the last place
      (8 neighbours: each component +-1 ulp) — if it does, no implementation can be expected to reproduce the
      reference binary on that input, not even the reference built against another libm version.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

TOL = 1e-5
NEAR_TIE = 1e-9   # a decision whose relative margin is below this is a near-tie (typical margins: 1e-3 .. 1)
J_SPLIT = 1e-9    # relative difference of new_J from which two traces count as parted


def oracle_scenes(wl):
    from oracle import Scene
    return [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs, s.road_borders, s.ref_velo) for s in wl.scenes]


def outside_band(a, b, tol=TOL):
    """per trajectory: max |du|, |dx|, |dJ| and the mask of those outside the band (NaN counts as outside)"""
    nb = a["res"].shape[0]
    du = np.abs(a["u"] - b["u"]).reshape(nb, -1).max(axis=1)
    dx = np.abs(a["x"] - b["x"]).reshape(nb, -1).max(axis=1)
    dJ = np.abs(a["res"]["J_final"] - b["res"]["J_final"])
    bad = ~((du <= tol) & (dx <= tol) & (dJ <= tol))
    return du, dx, dJ, bad


def first_split(ta, tb):
    """index of the first trace record at which two decision traces part (len of the shorter if one is a prefix)"""
    n = min(len(ta), len(tb))
    for k in range(n):
        a, b = ta[k], tb[k]
        if (a["status"] != b["status"] or a["trials"] != b["trials"] or a["accepted"] != b["accepted"]
                or a["alpha_idx"] != b["alpha_idx"] or a["lamb"] != b["lamb"]):
            return k, "decision"
        ja, jb = float(a["new_J"]), float(b["new_J"])
        if not (abs(ja - jb) <= J_SPLIT * max(abs(ja), abs(jb), 1e-300)):
            return k, "cost"
    return n, ("length" if len(ta) != len(tb) else "none")


def same_trace(ta, tb):
    if len(ta) != len(tb):
        return False
    return all(ta[f].tolist() == tb[f].tolist() for f in ("status", "trials", "accepted", "alpha_idx", "lamb"))


def ulp_neighbours(x0):
    out = []
    for c in range(4):
        for direction in (np.inf, -np.inf):
            y = np.array(x0, dtype=np.float64)
            y[c] = np.nextafter(y[c], direction)
            out.append(y)
    return out


class Diagnoser:
    def __init__(self, wl):
        from oracle import Oracle
        self.wl = wl
        self.scenes = oracle_scenes(wl)
        self.orc = {"libm": Oracle("libm"), "det": Oracle("det")}
        self._solvers = {}

    def _solver(self, mode, pid):
        key = (mode, int(pid))
        if key not in self._solvers:
            self._solvers[key] = self.orc[mode].solver(self.wl.params[int(pid)])
        return self._solvers[key]

    def solve(self, mode, b, x0=None, margins=True):
        wl = self.wl
        s = self._solver(mode, wl.param_id[b])
        s.reset()
        return s.solve(wl.x0[b] if x0 is None else x0, self.scenes[int(wl.scenario_id[b])], tick=int(wl.tick[b]),
                       trace_cap=256, margins=margins)

    def one(self, b):
        """the record of one divergent trajectory"""
        lm, dt = self.solve("libm", b), self.solve("det", b)
        k, kind = first_split(lm["trace"], dt["trace"])
        kk = min(k, len(lm["trace"]) - 1, len(dt["trace"]) - 1)
        mg = {}
        for name in ("ls", "pd", "ref"):
            mg[name] = float(min(lm["margins"][name][kk], dt["margins"][name][kk]))
        upto = {name: float(min(lm["margins"][name][:kk + 1].min(), dt["margins"][name][:kk + 1].min()))
                for name in ("ls", "pd", "ref")}
        flips = 0
        for y in ulp_neighbours(self.wl.x0[b]):
            r = self.solve("libm", b, x0=y, margins=False)
            flips += 0 if same_trace(r["trace"], lm["trace"]) else 1
        which = min(mg, key=mg.get)
        return {"trajectory": int(b), "first_split_record": int(k), "split_kind": kind,
                "iterations_libm": int(lm["res"]["iters"]), "iterations_hip": int(dt["res"]["iters"]),
                "J_final_libm": float(lm["res"]["J_final"]), "J_final_hip": float(dt["res"]["J_final"]),
                "margin_at_split": mg, "smallest_margin_at_split": mg[which], "decision_kind": which,
                "smallest_margin_up_to_split": min(upto.values()),
                "libm_trace_changes_under_1ulp_x0": int(flips), "of_neighbours": 8}

    def control_flip_rate(self, rows):
        """how often a trajectory INSIDE the band changes its libm decision trace under the same 1-ulp moves"""
        flipped = 0
        for b in rows:
            base = self.solve("libm", b, margins=False)
            if any(not same_trace(self.solve("libm", b, x0=y, margins=False)["trace"], base["trace"])
                   for y in ulp_neighbours(self.wl.x0[b])):
                flipped += 1
        return flipped


def analyse(wl, hip_out, threads=8, control=64):
    """hip_out: dict(u, x, res) of the HIP path (or of the detmath oracle, its bit-identical CPU twin)."""
    dg = Diagnoser(wl)
    ref = dg.orc["libm"].solve_batch(wl.params, dg.scenes, wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=threads)
    du, dx, dJ, bad = outside_band(hip_out, ref)
    rows = np.nonzero(bad)[0]
    recs = [dg.one(int(b)) for b in rows]
    good = np.nonzero(~bad)[0]
    ctl_rows = good[:: max(1, len(good) // control)][:control] if len(good) else []
    ctl_flips = dg.control_flip_rate([int(b) for b in ctl_rows])
    inside = ~bad
    return {
        "workload": wl.name, "trajectories": int(wl.B), "tolerance": TOL,
        "within_1e-5": int(inside.sum()), "within_1e-5_frac": float(inside.mean()),
        "outside_1e-5": int(bad.sum()),
        "max_abs_du_inside": float(du[inside].max()) if inside.any() else None,
        "max_abs_dx_inside": float(dx[inside].max()) if inside.any() else None,
        "max_abs_dJ_inside": float(dJ[inside].max()) if inside.any() else None,
        "max_abs_dJ_outside": float(np.nanmax(dJ[bad])) if bad.any() else None,
        "near_tie_threshold": NEAR_TIE,
        "outside_with_near_tie_at_split": int(sum(r["smallest_margin_at_split"] < NEAR_TIE for r in recs)),
        "outside_where_libm_flips_under_1ulp_x0": int(sum(r["libm_trace_changes_under_1ulp_x0"] > 0 for r in recs)),
        "outside_explained": int(sum((r["smallest_margin_at_split"] < NEAR_TIE) or (r["libm_trace_changes_under_1ulp_x0"] > 0)
                                     for r in recs)),
        "decision_kinds": {k: int(sum(r["decision_kind"] == k for r in recs)) for k in ("ls", "pd", "ref")},
        "control": {"trajectories_inside_band_sampled": int(len(ctl_rows)),
                    "of_which_libm_flips_under_1ulp_x0": int(ctl_flips)},
        "records": recs,
    }


def make_workload(pkg, cfg):
    W = pkg.workloads
    return {2: W.config2, 3: W.config3, 5: W.config5}[cfg]()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="+", default=[2, 3, 5])
    ap.add_argument("--gpu", action="store_true", help="take the HIP results from the GPU (default: the detmath oracle, "
                                                       "which the GPU tests show to be bit-identical)")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import cilqr_amd as pkg
    from oracle import Oracle
    report = {"what": __doc__.split("\n\n")[0].replace("\n", " "),
              "hip_results_from": "MI355X (cilqr_solve_batch)" if args.gpu else "oracle detmath build (bit-identical twin of the HIP path)",
              "configs": []}
    for cfg in args.configs:
        wl = make_workload(pkg, cfg)
        if args.gpu:
            eng = pkg.BatchedCILQR(wl.params, wl.scenes)
            hip = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
            eng.close()
        else:
            hip = Oracle("det").solve_batch(wl.params, oracle_scenes(wl), wl.x0, wl.scenario_id, wl.param_id, wl.tick,
                                            n_threads=args.threads)
        rep = analyse(wl, hip, threads=args.threads)
        rep["baseline_config"] = cfg
        report["configs"].append(rep)
        brief = {k: v for k, v in rep.items() if k != "records"}
        print(json.dumps(brief), flush=True)
    if args.out:
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
