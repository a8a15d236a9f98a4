#!/usr/bin/env python3
"""Development aid: scan the gfx950 code of a library / object for a vector instruction that writes a scalar register pair
(the carry-out of v_mad_u64_u32 / v_mad_i64_i32 / v_add_co / v_sub_co / v_addc / v_subb, the scale flag of v_div_scale)
followed shortly by a SCALAR instruction that writes one of those registers again without reading it first.  On gfx950
the vector unit's scalar write was observed to land after the scalar unit's ("a lost scalar offset").
usage: scripts/sgpr_waw_scan.py FILE [window]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_metadata as km

VOP3B = ("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32",
         "v_subbrev_co_u32", "v_div_scale_f64", "v_div_scale_f32")


def regs(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(path, window):
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
                op = ins.split()[0]
                if not op.startswith(VOP3B):
                    continue
                ops = [t.strip() for t in ins[len(op):].split(",")]
                if len(ops) < 2:
                    continue
                sd = regs(ops[1])
                if not sd:
                    continue  # (vcc)
                total += 1
                live = set(sd)
                for j in range(i + 1, min(i + 1 + window, len(lines))):
                    if lines[j][0] != fn or not live:
                        break
                    nxt = lines[j][1]; nop = nxt.split()[0]
                    if nop.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_swappc")):
                        break
                    toks = [t.strip() for t in nxt[len(nop):].split(",")]
                    toks = [re.sub(r"^[-|]*|[|]*$", "", t.split()[0]) if t else t for t in toks]
                    dst = regs(toks[0]) if toks else set()
                    srcs = set().union(*[regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
                    if nop.startswith(("buffer_", "global_", "flat_", "ds_", "scratch_")):
                        srcs |= set().union(*[regs(t) for t in toks]); dst = set()
                    live -= srcs  # read first: the hardware orders a read behind the vector unit's write
                    is_scalar_write = nop.startswith("s_") and not nop.startswith(("s_nop", "s_waitcnt", "s_cmp", "s_bitcmp"))
                    if is_scalar_write and (dst & live):
                        hits += 1
                        print(f"{km.demangle([fn]).get(fn, fn)[:60]}: {ins}  ->  +{j - i}: {nxt}")
                        break
                    if nop.startswith("v_") and nop.startswith(VOP3B + ("v_cmp", "v_readfirstlane", "v_readlane")):
                        live -= dst  # the vector unit writes it again: in order
    print(f"{path}: {total} vector instructions with a scalar carry-out, {hits} overwritten by the scalar unit within {window} instructions unread")
    return hits


if __name__ == "__main__":
    sys.exit(1 if scan(os.path.abspath(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 6) else 0)
