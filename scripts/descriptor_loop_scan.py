#!/usr/bin/env python3
"""Development aid / test: find buffer accesses that the compiler wrapped in a loop over the lanes' distinct descriptors or
scalar offsets ("waterfall": v_readfirstlane of the operand, compare, s_and_saveexec, the access, s_xor exec, s_cbranch_execnz).
A 16-byte buffer STORE in such a loop lost data on gfx950 with XNACK off when its data register was reused a few instructions
behind the loop (profiles/r04_experiments/tiled_slab_lost_rows.txt).   usage: scripts/descriptor_loop_scan.py FILE [--json]"""
import json, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_metadata as km


def scan(path):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in km.extract_code_objects(path, tmp):
            txt = subprocess.run([os.path.join(km.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            cur, lines = None, []
            for ln in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
                if m:
                    cur = m.group(1); continue
                ln = ln.strip()
                if ln and not ln.startswith("//"):
                    lines.append((cur, ln.split("//")[0].strip()))
            names = km.demangle(sorted({fn for fn, _ in lines if fn}))
            for i, (fn, ins) in enumerate(lines):
                if not re.match(r"buffer_(store|load|atomic)", ins):
                    continue
                before = [l[1] for l in lines[max(0, i - 3):i] if l[0] == fn]
                after = [l[1] for l in lines[i + 1:i + 4] if l[0] == fn]
                if any(b.startswith("s_and_saveexec") for b in before) and any(a.startswith("s_cbranch_execnz") for a in after):
                    out.append({"function": names.get(fn, fn or "?"), "access": ins})
    return out


if __name__ == "__main__":
    hits = scan(os.path.abspath(sys.argv[1]))
    if "--json" in sys.argv:
        print(json.dumps(hits))
    else:
        for h in hits:
            print(h["function"][:80], "|", h["access"])
        stores = sum(1 for h in hits if "store" in h["access"])
        print(f"{len(hits)} buffer accesses inside per-descriptor loops, {stores} of them stores")
