"""The N > 1 path on CPU: world_size-2 gloo.  The solve path shards by trajectory with no data-path
collective; what must be right is (a) every rank regenerating exactly its own rows of the global
batch, (b) the statistics reduction bench.py uses, (c) shard-count invariance of the results.
The oracle stands in for the GPU kernels here (no GPU in this container)."""
import os
import pathlib
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

WORKER = textwrap.dedent('''
    import os, sys, json, importlib
    import numpy as np
    sys.path.insert(0, os.environ["CILQR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["CILQR_ROOT"], "oracle"))
    import torch, torch.distributed as dist
    import cilqr_amd as pkg
    from oracle import Oracle, Scene
    stats_mod = importlib.import_module("toy-example-of-ilqr_amd.stats")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = 12
    wl = pkg.workloads.config3(B=per, N=30, first=rank * per)         # what bench.py does per rank
    full = pkg.workloads.config3(B=per * world, N=30)
    assert np.array_equal(full.shard(rank, world).x0, wl.x0)          # (a)
    sc = wl.scenes[0]
    scene = Scene(sc.lane_x, sc.lane_y, sc.lane_yaw, sc.obs, sc.road_borders, sc.ref_velo)
    out = Oracle("det").solve_batch(wl.params, scene, wl.x0)
    vec = stats_mod.local_stats(out["res"], wl.N, wl.M_of)
    tot, tmax = stats_mod.reduce_stats(vec, 1.0 + rank, dist, None)   # (b)
    xs = [torch.zeros(per, wl.N + 1, 4, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(xs, torch.from_numpy(out["x"]))
    if rank == 0:
        ref = Oracle("det").solve_batch(full.params, scene, full.x0)
        ok_x = bool(np.array_equal(torch.cat(xs).numpy(), ref["x"]))  # (c)
        ref_vec = stats_mod.local_stats(ref["res"], full.N, full.M_of)
        print("RESULT " + json.dumps({"ok_x": ok_x, "tot": tot.tolist(), "ref": ref_vec.tolist(), "tmax": tmax}))
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_rank_gloo_sharding_and_stats(built, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, CILQR_ROOT=str(ROOT), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    import json
    r = json.loads(line[len("RESULT "):])
    assert r["ok_x"]
    np.testing.assert_allclose(r["tot"], r["ref"], rtol=1e-12)
    assert r["tmax"] == 2.0


def test_shards_partition_the_batch(pkg):
    for maker, kw in ((pkg.workloads.config2, dict(B=40, N=30)), (pkg.workloads.config4, dict(B=40, N=100))):
        full = maker(**kw)
        for world in (1, 2, 4, 8):
            parts = [full.shard(r, world) for r in range(world)]
            assert sum(p.B for p in parts) == full.B
            np.testing.assert_array_equal(np.concatenate([p.x0 for p in parts]), full.x0)
            np.testing.assert_array_equal(np.concatenate([p.scenario_id for p in parts]), full.scenario_id)
    # per-rank regeneration by global index == slicing the global batch
    a = pkg.workloads.config4(B=10, N=100, first=20)
    b = pkg.workloads.config4(B=40, N=100)
    np.testing.assert_array_equal(a.x0, b.x0[20:30])
    np.testing.assert_array_equal(a.scenario_id, b.scenario_id[20:30])
    # bench.py's default multi-GPU workload: rank r takes base starts r * B_base ... of the global sweep
    g5 = pkg.workloads.config5(B_base=16, N=30)
    r1 = pkg.workloads.config5(B_base=8, N=30, first=8)
    np.testing.assert_array_equal(r1.x0, g5.x0[128:256])
    np.testing.assert_array_equal(r1.param_id, g5.param_id[128:256])
    c5 = pkg.workloads.config5(B_base=3, N=30)
    assert c5.B == 48 and len(c5.params) == 16 and c5.param_id[:17].tolist() == list(range(16)) + [0]
    assert pkg.workloads.bytes_per_iteration(50, 3) == 8536 and pkg.workloads.bytes_per_iteration(100, 8) == 29056
