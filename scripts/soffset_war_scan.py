#!/usr/bin/env python3
"""Development aid: scan the gfx950 code of a library for a buffer store whose scalar-offset register is overwritten by one
of the next few instructions (a 16-byte buffer store was observed to pick up the NEW value).
usage: scripts/soffset_war_scan.py FILE [window]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_metadata as km

path = os.path.abspath(sys.argv[1]); window = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
        for i, (fn, ins) in enumerate(lines):
            m = re.match(r"buffer_(store|load)_\w+ .*, s\[\d+:\d+\], (s\d+)\b", ins)
            if not m:
                continue
            total += 1
            reg = m.group(2)
            for j in range(i + 1, min(i + 1 + window, len(lines))):
                nxt = lines[j][1]
                if lines[j][0] != fn:
                    break
                op = nxt.split()[0]
                dst = nxt[len(op):].split(",")[0].strip()
                if (op.startswith("s_") or op.startswith("v_readfirstlane") or op.startswith("v_readlane")) and dst == reg:
                    hits += 1
                    print(f"{km.demangle([fn]).get(fn, fn)[:70]}: {ins}  ->  +{j - i}: {nxt}")
                    break
print(f"{total} buffer accesses with a scalar offset register, {hits} followed within {window} instructions by a write of it")
