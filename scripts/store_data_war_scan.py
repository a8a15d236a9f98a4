#!/usr/bin/env python3
"""Development aid: scan gfx950 code for a 12- or 16-byte store whose DATA registers are overwritten by a vector instruction
a few instructions later (profiles/r04_experiments/tiled_slab_lost_rows.txt: such a store lost the first dword of lanes 12-15
to the next store's offset, written five scalar instructions behind it, inside the compiler's per-descriptor loop).
usage: scripts/store_data_war_scan.py FILE [window] [pattern]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_metadata as km


def vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


path = os.path.abspath(sys.argv[1]); window = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pat = sys.argv[3] if len(sys.argv) > 3 else ""
hits = total = 0
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
            if pat and pat not in names.get(fn, fn):
                continue
            m = re.match(r"(buffer|global|flat|scratch)_store_dwordx[34] (.*)", ins)
            if not m:
                continue
            total += 1
            ops = [t.strip() for t in m.group(2).split(",")]
            data = vregs(ops[0]) if m.group(1) == "buffer" else vregs(ops[1])
            for j in range(i + 1, min(i + 1 + window, len(lines))):
                if lines[j][0] != fn:
                    break
                nxt = lines[j][1]; op = nxt.split()[0]
                if op.startswith(("s_waitcnt", "s_barrier")) and ("vmcnt" in nxt or op == "s_barrier"):
                    break
                if not op.startswith(("v_", "ds_read", "buffer_load", "global_load", "flat_load")):
                    continue
                dst = vregs(nxt[len(op):].split(",")[0].strip())
                if dst & data:
                    hits += 1
                    valu_between = sum(1 for k in range(i + 1, j) if lines[k][1].startswith("v_"))
                    print(f"{names.get(fn, fn)[:56]}: {ins[:58]} -> +{j - i} ({valu_between} vector in between): {nxt[:60]}")
                    break
print(f"{path}: {total} stores of 12 / 16 bytes, {hits} with a data register rewritten within {window} instructions")
