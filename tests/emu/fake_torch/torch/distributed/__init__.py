"""TEST INFRASTRUCTURE ONLY — the handful of torch.distributed calls bench.py makes (init, all_reduce SUM / MAX of a small vector,
all_gather_object, barrier), over FILES in $CILQR_FAKE_DIST_DIR: every collective has a sequence number, every rank writes its
contribution and waits for the others'.  For rehearsals of the N > 1 code path on the CPU emulator (scripts/emu_rehearse.py
--ranks N); nothing about it resembles RCCL but the call signatures."""
import os
import pickle
import time

import numpy as _np

_state = {"rank": 0, "world": 1, "seq": 0, "dir": None}


class ReduceOp:
    SUM, MAX = "sum", "max"


def init_process_group(backend=None, rank=0, world_size=1, device_id=None, **kw):
    _state.update(rank=int(rank), world=int(world_size), seq=0, dir=os.environ["CILQR_FAKE_DIST_DIR"])


def _exchange(obj):
    seq = _state["seq"]
    _state["seq"] += 1
    d, r, w = _state["dir"], _state["rank"], _state["world"]
    tmp = os.path.join(d, f".c{seq}_{r}.tmp")
    with open(tmp, "wb") as f:
        pickle.dump(obj, f)
    os.rename(tmp, os.path.join(d, f"c{seq}_{r}.pkl"))
    out = []
    t0 = time.time()
    for k in range(w):
        p = os.path.join(d, f"c{seq}_{k}.pkl")
        while not os.path.exists(p):
            if time.time() - t0 > 3600:
                raise TimeoutError(f"rank {r}: collective {seq} waits for rank {k}")
            time.sleep(0.01)
        with open(p, "rb") as f:
            out.append(pickle.load(f))
    return out


def all_reduce(t, op=ReduceOp.SUM):
    parts = _exchange(_np.array(t.a, copy=True))
    t.a[...] = _np.sum(parts, axis=0) if op == ReduceOp.SUM else _np.max(parts, axis=0)


def all_gather_object(out, obj):
    parts = _exchange(obj)
    for i, p in enumerate(parts):
        out[i] = p


def barrier():
    _exchange(None)


def destroy_process_group():
    pass
