"""How much of the parity claim hangs on the one thing nobody can check here: the order in which Eigen adds the four terms of
its small inner products (cs:211-212, 400-436, 449-451).

TEST INFRASTRUCTURE (imports oracle/).  Used by tests/test_oracle_properties.py::test_sum_order_* and, as a script, to write
profiles/r06_sum_order_tolerance.json:

    python tests/sum_order_tolerance.py [--configs 2 3 5 4] [--rows 1024 2048 4096 1024] [--out profiles/r06_sum_order_tolerance.json]

The oracle (and with it the HIP kernels, which are bit-identical to its detmath build) adds every inner product in index order,
((p0 + p1) + p2) + p3 — what Eigen's coefficient-based lazy product computes without vectorisation.  Eigen is not vendored
upstream (CMakeLists.txt:39 takes /usr/include/eigen3), the products are written on `block(i, j, rows, cols)` expressions whose
sizes are dynamic at compile time, and the reference builds with -O3 and no -march (SSE2 packets of two doubles): depending on
version and expression, Eigen may evaluate the same sums as a balanced tree or lane-wise in packets.  None of that can be run
here.  What CAN be measured is the consequence: the oracle's libm flavour rebuilt with every four-term product associated
 * as adjacent pairs,    (p0 + p1) + (p2 + p3)   liboracle_tree.so  (-DORC_SUM4=1),
 * as interleaved pairs, (p0 + p2) + (p1 + p3)   liboracle_pkt.so   (-DORC_SUM4=2),
solved on the rows of the BASELINE configurations and compared with the default build: fraction of rows within 1e-5 (u, x,
J_final), rows with the same decision counters, and — reusing tests/libm_tolerance.py's yardstick — whether every row that
moves by more than 1e-5 is one on which the default build does not reproduce ITSELF to 1e-5 when x0 moves by one ulp."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import libm_tolerance as lt  # noqa: E402

TOL = 1e-5
MODES = {"tree": "(p0 + p1) + (p2 + p3)", "pkt": "(p0 + p2) + (p1 + p3)"}


def workload(pkg, cfg, rows):
    wl = pkg.workloads
    if cfg == "2":
        return wl.config2(B=rows, N=50)
    if cfg == "3":
        return wl.config3(B=rows, N=50)
    if cfg == "4":
        return wl.config4(B=rows, N=100)
    if cfg == "5":
        return wl.config5(B_base=rows // 16, N=50)
    raise ValueError(cfg)


def analyse(wl, threads=8, with_spread=True):
    from oracle import Oracle
    scenes = lt.oracle_scenes(wl)

    def run(mode, x0=None):
        return Oracle(mode).solve_batch(wl.params, scenes, wl.x0 if x0 is None else x0, wl.scenario_id, wl.param_id, wl.tick,
                                        n_threads=threads)

    base = run("libm")
    spread = None
    if with_spread:
        spread = np.zeros(wl.B)
        for c in range(4):
            for direction in (np.inf, -np.inf):
                x0 = wl.x0.copy()
                x0[:, c] = np.nextafter(x0[:, c], direction)
                spread = np.maximum(spread, lt.max_gap(run("libm", x0), base))
    out = {"workload": wl.name, "rows": int(wl.B), "horizon": int(wl.N), "tolerance": TOL,
           "default_build": "index order ((p0 + p1) + p2) + p3, glibc libm", "modes": {}}
    if spread is not None:
        out["rows_the_default_build_reproduces_under_1ulp_of_x0 (spread <= 1e-5)"] = int((spread <= TOL).sum())
    br = base["res"]
    for mode, what in MODES.items():
        r = run(mode)
        gap = lt.max_gap(r, base)
        bad = ~(gap <= TOL)
        rr = r["res"]
        same = ((rr["iters"] == br["iters"]) & (rr["ls_trials"] == br["ls_trials"]) & (rr["end_reason"] == br["end_reason"])
                & (rr["cost_evals"] == br["cost_evals"]) & (rr["final_status"] == br["final_status"]))
        ent = {"association": what, "within_1e-5": int((~bad).sum()), "within_1e-5_frac": float((~bad).mean()),
               "outside_1e-5": int(bad.sum()), "same_decision_counters": int(same.sum()),
               "same_decision_counters_frac": float(same.mean()),
               "outside_1e-5_among_same_decision_counters": int((bad & same).sum()),
               "bit_identical_rows": int(((r["u"] == base["u"]).reshape(wl.B, -1).all(axis=1)
                                          & (r["x"] == base["x"]).reshape(wl.B, -1).all(axis=1)).sum()),
               "gap_percentiles_50_90_99_max": [float(v) for v in np.percentile(gap[np.isfinite(gap)], [50, 90, 99, 100])]}
        if spread is not None:
            well = spread <= TOL
            ent["well_conditioned_rows_outside_1e-5"] = int((well & bad).sum())
            ent["max_gap_on_well_conditioned_rows"] = float(gap[well].max()) if well.any() else None
            ent["every_row_obeys gap <= max(1e-5, 2 x spread)"] = bool((gap <= np.maximum(TOL, 2.0 * spread)).all())
            ratio = gap[bad] / np.maximum(spread[bad], 1e-300)
            ent["max_gap_over_spread_outside"] = float(ratio.max()) if bad.any() else None
        out["modes"][mode] = ent
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="*", default=["2", "3", "5", "4"])
    ap.add_argument("--rows", nargs="*", type=int, default=[1024, 2048, 4096, 1024])
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import cilqr_amd as pkg
    rep = {"what": __doc__.split("\n\n")[0], "results": []}
    for cfg, rows in zip(a.configs, a.rows):
        r = analyse(workload(pkg, cfg, rows), a.threads)
        print(json.dumps(r), flush=True)
        rep["results"].append(r)
    if a.out:
        json.dump(rep, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
