#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: does "== oracle on the emulator" mean anything?  Hand-picked LOGIC bugs are planted into a copy of csrc/ —
one at a time: the regularisation's sign in the sweeps, two line-search rules, the lambda schedule, the obstacle ellipse's scale,
a row dropped when a trajectory changes wavefronts, the expected cost reduction, the serial first-local-minimum scan — the copy is built for
the wave64 emulator (tests/emu/build_emu.py --csrc) and a 40-trajectory batch in pairs per wavefront is compared with the
detmath oracle.  Every mutant must be caught (some output differs); the unmutated copy must pass.

    python scripts/emu_mutants.py [--out profiles/r06_emulator_mutants.json]"""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "toy-example-of-ilqr_amd", "csrc")
WORK = os.path.join(ROOT, "scratch", "mutants")

MUTANTS = [
    ("none (control)", None, None, None),
    ("pair sweep: Q_uu regularised with -lambda (cs:407-410)", "cilqr_group.hpp", "        if (diag) Q = Q + lamb;", "        if (diag) Q = Q - lamb;"),
    ("lone sweep: the same", "cilqr_device.hpp", "        if (diag) Q = Q + lamb;", "        if (diag) Q = Q - lamb;"),
    ("line search: accept on decay / approx >= threshold instead of > (cs:363-365)", "cilqr_device.hpp",
     "    if (decay > 0.0 && (approx < 0.0 || decay / approx > accept_thr)) return 2;",
     "    if (decay > 0.0 && (approx < 0.0 || decay / approx > 4.0 * accept_thr)) return 2;"),
    ("convergence tested at every step size, not only alpha = 1 (cs:358-361)", "cilqr_device.hpp",
     "    if (t == 0 && adecay < conv_thr) return 1;", "    if (adecay < conv_thr) return 1;"),
    ("lambda after a failure: lamb * amplify without the max() (cs:118-121)", "cilqr_kernels.hpp",
     "                        lamb = (c.k->lamb_amplify < la) ? la : c.k->lamb_amplify;", "                        lamb = la;"),
    ("obstacle ellipse: 1 x d_safe on the long axis (the Python variant's value) instead of 6 x (ut:389)", "cilqr_device.hpp",
     "        double a = 0.5 * p.length + p.d_safe * 6 + 0.5 * p.width;", "        double a = 0.5 * p.length + p.d_safe * 1 + 0.5 * p.width;"),
    ("hand-over: the last state row is not parked", "cilqr_group.hpp",
     "        for (int e = lane; e < 4 * (N + 1); e += CILQR_WAVE) park_st(pk + e, lx[e]);",
     "        for (int e = lane; e < 4 * N; e += CILQR_WAVE) park_st(pk + e, lx[e]);"),
    ("serial reference scan: stops one sample late (cs:298-306)", "cilqr_device.hpp",
     "        int adv = f0 ? (f1 ? (f2 ? (f3 ? 4 : 3) : 2) : 1) : 0;", "        int adv = f0 ? (f1 ? (f2 ? (f3 ? 4 : 3) : 2) : 1) : 1;"),
    ("expected cost reduction: alpha instead of alpha^2 on dV0 (cs:362)", "cilqr_device.hpp",
     "    const double approx = -(alpha * alpha * dV0 + alpha * dV1);", "    const double approx = -(alpha * dV0 + alpha * dV1);"),
]

CHECK = r"""
import json, os, sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import cilqr_amd as pkg
from oracle import Oracle, Scene
res = {}
for name, N, B, mode, flags in (("three_bend", 30, 40, 2, 0), ("two_straight", 50, 10, 2, 0), ("three_bend", 30, 6, 0, 0),
                                 ("three_bend", 30, 4, 0, pkg._lib.DBG_SERIAL_REF_SCAN)):   # (the last: the serial reference chain forced, a testing aid)
    cfg = pkg.GlobalConfig.get_instance(name); sc = pkg.build_scenario(cfg, name)
    p = pkg.params_from_config(cfg, N=N)
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), dev=True); eng.set_group_mode(mode)
    if flags: eng.set_debug_flags(flags)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0x5A0CE)
    try:
        out = eng.solve_batch(x0)
    except Exception as e:
        res[f"{name} N={N} mode {mode}" + (" serial scan" if flags else "")] = "error: " + str(e)[:80]; continue
    ref = Oracle("det").solve_batch(p, Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, sc.road_borders, sc.target_velocity), x0, n_threads=4)
    bad = [b for b in range(B) if not (np.array_equal(out["u"][b], ref["u"][b]) and np.array_equal(out["x"][b], ref["x"][b]) and out["res"]["iters"][b] == ref["res"]["iters"][b])]
    res[f"{name} N={N} mode {mode}" + (" serial scan" if flags else "")] = len(bad)
print("MUTANT " + json.dumps(res))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    rows = []
    for i, (what, fname, old, new) in enumerate(MUTANTS):
        d = os.path.join(WORK, f"m{i}")
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(CSRC, d)
        if fname:
            p = os.path.join(d, fname)
            t = open(p).read()
            assert t.count(old) >= 1, (what, "site not found")
            open(p, "w").write(t.replace(old, new))
        lib = build_emu.build(dev=True, csrc=d, out=os.path.join(WORK, f"libmut{i}.so"))
        env = dict(os.environ, CILQR_AMD_LIB=str(lib), CILQR_AMD_LIB_DEV=str(lib))
        r = subprocess.run([sys.executable, "-c", CHECK, ROOT], capture_output=True, text=True, timeout=1800, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("MUTANT ")]
        res = json.loads(line[-1][7:]) if line else {"crashed": r.stderr[-300:]}
        caught = any(v != 0 for v in res.values())
        rows.append({"mutant": what, "file": fname, "mismatching_trajectories": res, "caught": caught})
        print(json.dumps(rows[-1]), flush=True)
        shutil.rmtree(d, ignore_errors=True)
    ok = (not rows[0]["caught"]) and all(r["caught"] for r in rows[1:])
    rep = {"what": __doc__.split("\n\n")[0], "control_passes": not rows[0]["caught"], "mutants": len(rows) - 1,
           "caught": sum(r["caught"] for r in rows[1:]), "rows": rows}
    print(json.dumps({k: rep[k] for k in ("control_passes", "mutants", "caught")}))
    if a.out:
        json.dump(rep, open(a.out, "w"), indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
