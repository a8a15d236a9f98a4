#!/usr/bin/env python3
"""Development aid: disassemble the gfx950 code in a library / object file and print, per function whose demangled name
contains PATTERN, its size and the innermost loops with scratch / lane-move counts.   scripts/isa_dump.py FILE PATTERN [outdir]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_metadata as km

path, pat = os.path.abspath(sys.argv[1]), sys.argv[2]
outdir = sys.argv[3] if len(sys.argv) > 3 else None
with tempfile.TemporaryDirectory() as tmp:
    for co in km.extract_code_objects(path, tmp):
        txt = subprocess.run([os.path.join(km.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur, body = None, {}
        for ln in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
            if m:
                cur = m.group(1); body[cur] = []
            elif cur and ln.strip():
                body[cur].append(ln.strip())
        names = km.demangle(list(body))
        for n, lines in body.items():
            d = names.get(n, n)
            if pat not in d:
                continue
            ops = [l.split()[0] for l in lines if not l.startswith("//")]
            print(f"{d[:110]}: {len(ops)} instructions, scratch {sum(o.startswith('scratch_') for o in ops)}, "
                  f"lane moves {sum(o.startswith('v_readlane') or o.startswith('v_writelane') for o in ops)}")
            if outdir:
                os.makedirs(outdir, exist_ok=True)
                open(os.path.join(outdir, re.sub(r'[^A-Za-z0-9_]', '_', d[:60]) + ".s"), "w").write("\n".join(lines))
