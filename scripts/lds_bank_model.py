#!/usr/bin/env python3
"""Bank-conflict model of the backward sweep's per-step LDS reads.

Every lane of the 6 x 8 grid reads ten coefficients per step through its own running address (lane_map_M / make_lane_map in
csrc/cilqr_device.hpp): M[k][r'] (k = 0..3), M[k][c''] (k = 0..3), L[r'][c''] and l[r'].  All are ds_read_b64: two lane
groups of 32, bank = (byte address / 4) mod 64, identical addresses broadcast, each further distinct address on a busy bank
costs one more LDS cycle (/opt/skills/guides/MI355X_MICROARCH.md, LDS table).  This script rebuilds the addresses for a
layout (offsets in doubles of x, kd, lx, lu, lxx, luu, xch inside the trajectory's LDS block) and counts, per step, the extra
cycles each of the ten reads costs.

    python scripts/lds_bank_model.py [--N 50] [--layout single|grouped] [--pad-kd 0] [--json]
"""
import argparse
import json

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=50)
ap.add_argument("--layout", default="single")
ap.add_argument("--kd-stride", type=int, default=10, help="doubles per step of the Jacobian / gain array")
ap.add_argument("--lxx-stride", type=int, default=7)
ap.add_argument("--json", action="store_true")
a = ap.parse_args()
N, R = a.N, a.N + 1
KD, LXS = a.kd_stride, a.lxx_stride
off = {}
p = 0
if a.layout == "single":   # carve() of k_solve, one stage-cost slot, expansion in LDS
    off["x"] = p; p += 4 * R
    off["u"] = p; p += 2 * N
    off["kd"] = p; p += max(KD * N, 3 * R)
    off["lx"] = p; p += 4 * R
    off["lu"] = p; p += 2 * N
    off["lxx"] = p; p += LXS * R
    off["luu"] = p; p += 2 * N
    off["xch"] = p; p += 4
else:                      # carve_group(): per-trajectory block first (slot 0), then the shared area
    PG = 4 * R + 2 * N + (((3 * (N + 2) + 1) // 2 + 1) & ~1) + 32 + 20 + 12
    off["x"] = 0; off["u"] = 4 * R
    p = 2 * PG
    off["kd"] = p; p += max(KD * N, 3 * R)
    off["lx"] = p; p += 4 * R
    off["lu"] = p; p += 2 * N
    off["lxx"] = p; p += LXS * R
    off["luu"] = p; p += 2 * N
    off["xch"] = p; p += 4
ZERO, ONE, DT = off["xch"], off["xch"] + 1, off["xch"] + 2


def map_M(k, j):
    A5, B3 = off["kd"], off["kd"] + 5
    o, s = ZERO, 0
    if j < 4 and k == j: o = ONE
    if j == 2 and k == 0: o, s = A5 + 0, KD
    if j == 2 and k == 1: o, s = A5 + 2, KD
    if j == 2 and k == 3: o, s = A5 + 4, KD
    if j == 3 and k == 0: o, s = A5 + 1, KD
    if j == 3 and k == 1: o, s = A5 + 3, KD
    if j == 4 and k == 2: o = DT
    if j == 5 and k == 0: o, s = B3 + 0, KD
    if j == 5 and k == 1: o, s = B3 + 1, KD
    if j == 5 and k == 3: o, s = B3 + 2, KD
    return o, s


def lane_reads(lane, i):
    rp, cc = (lane >> 3) % 6, lane & 7
    ccm = min(cc, 5)
    reads = []
    for k in range(4):
        o, s = map_M(k, rp); reads.append(("m1[%d]" % k, o + s * i))
    for k in range(4):
        o, s = map_M(k, ccm); reads.append(("m2[%d]" % k, o + s * i))
    lq, slq = ZERO, 0
    if rp < 4 and cc < 4:
        lo, hi = min(rp, cc), max(rp, cc)
        e = {(0, 0): 0, (0, 1): 1, (0, 3): 2, (1, 1): 3, (1, 3): 4, (3, 3): 5, (2, 2): 6}.get((lo, hi), -1)
        if e >= 0: lq, slq = off["lxx"] + e, LXS
    elif rp >= 4 and cc == rp:
        lq, slq = off["luu"] + (rp - 4), 2
    reads.append(("Lq", lq + slq * i))
    if rp < 4: reads.append(("lv", off["lx"] + rp + 4 * i))
    else: reads.append(("lv", off["lu"] + (rp - 4) + 2 * i))
    return reads


def extra_cycles(addrs):
    """addrs: 64 double-offsets of one ds_read_b64 -> extra LDS cycles (two groups of 32 lanes, 64 banks of 4 bytes; a
    double covers two banks; same address = broadcast)"""
    extra = 0
    for g in (range(0, 32), range(32, 64)):
        per_bank = {}
        for ln in g:
            d = addrs[ln]
            for b in ((2 * d) % 64, (2 * d + 1) % 64):
                per_bank.setdefault(b, set()).add(d)
        extra += max(len(v) for v in per_bank.values()) - 1
    return extra


tot = {}
for i in range(N):
    per_lane = [lane_reads(l, i) for l in range(64)]
    for r in range(10):
        name = per_lane[0][r][0]
        tot[name] = tot.get(name, 0) + extra_cycles([per_lane[l][r][1] for l in range(64)])
rep = {"layout": a.layout, "N": N, "kd_stride": KD, "lxx_stride": LXS, "offsets_doubles": off,
       "extra_lds_cycles_per_step": {k: v / N for k, v in tot.items()},
       "extra_lds_cycles_per_step_total": sum(tot.values()) / N,
       "conflict_free_cycles_per_step": 10 * 2}
print(json.dumps(rep, indent=1) if a.json else rep)
