#!/usr/bin/env python3
"""Development aid / test: find vector-memory accesses that the compiler wrapped in a "waterfall" loop — a loop over the
lanes' distinct values of an operand that must be scalar (a buffer descriptor, a scalar offset): v_readfirstlane of the
operand, compare, s_and_saveexec, the access, s_xor exec, s_cbranch_execnz back.
A 16-byte buffer STORE in such a loop lost data on gfx950 with XNACK off when its data register was reused a few instructions
behind the loop (profiles/r04_experiments/tiled_slab_lost_rows.txt; the mechanism was not established).

Round 5: the loop is recognised by its STRUCTURE, not by the distance of its pieces from the access — any
backward s_cbranch_execnz whose span holds a v_readfirstlane and an s_and_saveexec (however the scheduler has spread them)
— and every buffer / global / flat / scratch access inside is reported, stores and loads alike; the first version looked
for s_and_saveexec within three instructions before a buffer access and s_cbranch_execnz within three after.
(Ordinary divergent loops — a per-lane trip count, `for (k = lane; k <= N; k += 64) out[k] = ...` — also end in
s_cbranch_execnz; they have no v_readfirstlane / s_and_saveexec pair per trip and are not what lost data.)
   usage: scripts/descriptor_loop_scan.py FILE [--json]"""
import json, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_metadata as km

MEM = re.compile(r"(buffer|global|flat|scratch)_(store|load|atomic)")
MAX_SPAN = 64  # instructions: a waterfall loop is a handful; anything longer is an ordinary loop


def scan(path):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in km.extract_code_objects(path, tmp):
            txt = subprocess.run([os.path.join(km.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            funcs, cur, base = {}, None, 0
            for ln in txt.splitlines():
                m = re.match(r"^([0-9a-f]+) <(.*)>:$", ln)
                if m:
                    cur, base = m.group(2), int(m.group(1), 16)
                    funcs[cur] = []
                    continue
                ln = ln.strip()
                if not cur or not ln or ln.startswith("//"):
                    continue
                ins = ln.split("//")[0].strip()
                ma = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
                addr = int(ma.group(1), 16) - base if ma else None
                mt = re.search(r"<[^>+]+\+0x([0-9a-fA-F]+)>\s*$", ln)
                tgt = int(mt.group(1), 16) if mt else None
                funcs[cur].append((addr, ins, tgt))
            names = km.demangle(sorted(funcs))
            for fn, lines in funcs.items():
                at = {a: i for i, (a, _, _) in enumerate(lines) if a is not None}
                for i, (addr, ins, tgt) in enumerate(lines):
                    if not ins.startswith("s_cbranch_execnz") or tgt is None or addr is None or tgt > addr or tgt not in at:
                        continue
                    j = at[tgt]
                    if i - j > MAX_SPAN:
                        continue
                    span = [l[1] for l in lines[j:i]]
                    if not (any(x.startswith("v_readfirstlane") for x in span) and any(x.startswith("s_and_saveexec") for x in span)):
                        continue
                    for x in span:
                        if MEM.match(x):
                            out.append({"function": names.get(fn, fn or "?"), "access": x, "loop_instructions": i - j + 1})
    return out


if __name__ == "__main__":
    hits = scan(os.path.abspath(sys.argv[1]))
    if "--json" in sys.argv:
        print(json.dumps(hits))
    else:
        for h in hits:
            print(h["function"][:80], "|", h["access"], "| loop of", h["loop_instructions"])
        stores = sum(1 for h in hits if "store" in h["access"])
        print(f"{len(hits)} vector-memory accesses inside waterfall loops, {stores} of them stores")
