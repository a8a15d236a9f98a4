#!/usr/bin/env python3
"""Which device code does a library carry?  A manifest {function symbol: sha256 of its machine code} over every gfx950 code
object bundled into a libcilqr_amd*.so — kernels and the out-of-line device functions they call (the grouped kernel's phases).
Runs anywhere hipcc's LLVM tools are (no GPU).

  scripts/device_code_identity.py --out profiles/rNN_device_code.json          manifest of the shipped library
  scripts/device_code_identity.py --against profiles/rNN_device_code.json      exit 0 iff every symbol of that manifest is in
                                                                               the shipped library with the same bytes
  (--lib path  another library;  --only REGEX  compare only matching symbols, e.g. 'k_solve')

Why: the counter passes and kernel timings under profiles/ are stamped with a hash of the SOURCES (bench.py csrc_fingerprint).
A host-side edit (C-ABI plumbing, scenario construction) moves that stamp although no kernel changed; this file answers the
question that matters for such evidence — are the instructions the same — from the binaries.  bench.py prints the answer next
to the stamp (roofline.traffic_collected.device_code_of_dominant_kernel_unchanged)."""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_metadata import LLVM, ROOT, extract_code_objects  # noqa: E402


def functions_of(co):
    """(name, sha256[:16], bytes) of every FUNC symbol of a code object's .text"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "-s", "-W", co], check=True, text=True,
                         capture_output=True).stdout
    m = re.search(r"\]\s+\.text\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", txt)
    if not m:
        return []
    addr, off, size = (int(g, 16) for g in m.groups())
    blob = open(co, "rb").read()[off:off + size]
    out = []
    for line in txt.splitlines():
        f = line.split()
        # Num: Value Size Type Bind Vis Ndx Name
        if len(f) == 8 and f[3] == "FUNC" and f[6].isdigit():
            a, n = int(f[1], 16), int(f[2])
            if n and addr <= a and a + n <= addr + size:
                out.append((f[7], hashlib.sha256(blob[a - addr:a - addr + n]).hexdigest()[:16], n))
    return out


def manifest(lib):
    man = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in extract_code_objects(lib, tmp):
            for name, h, n in functions_of(co):
                # a function compiled into several code objects (one per compilation unit) appears once per copy
                man.setdefault(name, [])
                if [h, n] not in man[name]:
                    man[name].append([h, n])
    return {k: sorted(v) for k, v in sorted(man.items())}


def compare(ref, cur, only=None):
    """symbols of `ref` missing from / different in `cur` (symbols only in `cur` are additions: reported, not failures)"""
    pat = re.compile(only) if only else None
    missing, changed, renamed, same = [], [], [], 0
    # (a template that gained parameters changes the mangled NAME of every instantiation: a function whose bytes are all found
    #  in the current library under another name is the same machine code — counted under `renamed`)
    blobs = {tuple(x) for v in cur.values() for x in v}
    for name, v in ref.items():
        if pat and not pat.search(name):
            continue
        if name in cur and cur[name] == v:
            same += 1
        elif all(tuple(x) in blobs for x in v):
            renamed.append(name)
        elif name not in cur:
            missing.append(name)
        else:
            changed.append(name)
    ref_blobs = {tuple(x) for v in ref.values() for x in v}
    added = [n for n in cur if n not in ref and (not pat or pat.search(n)) and not all(tuple(x) in ref_blobs for x in cur[n])]
    return {"same": same + len(renamed), "renamed": renamed, "changed": changed, "missing": missing, "added": added}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "toy-example-of-ilqr_amd", "libcilqr_amd.so"))
    ap.add_argument("--out")
    ap.add_argument("--against")
    ap.add_argument("--only")
    a = ap.parse_args()
    man = manifest(a.lib)
    if a.out:
        json.dump({"lib": os.path.basename(a.lib), "functions": man}, open(a.out, "w"), indent=0, sort_keys=True)
        print("wrote", a.out, len(man), "functions,", sum(n for v in man.values() for _, n in v), "bytes of code")
    if a.against:
        ref = json.load(open(a.against))["functions"]
        ref = {k: sorted([list(x) for x in v]) for k, v in ref.items()}
        r = compare(ref, man, a.only)
        print(json.dumps({"same": r["same"], "of_which_under_another_name": len(r["renamed"]), "changed": len(r["changed"]),
                          "missing": len(r["missing"]), "added": len(r["added"])}))
        for k in ("changed", "missing", "added"):
            for n in r[k][:40]:
                print(" ", k, n)
        sys.exit(0 if not r["changed"] and not r["missing"] else 1)


if __name__ == "__main__":
    main()
