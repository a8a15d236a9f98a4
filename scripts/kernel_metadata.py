#!/usr/bin/env python3
"""Register / spill / scratch evidence of the SHIPPED library, regenerated from the .so itself (runs anywhere hipcc's
LLVM tools are: no GPU needed).

  scripts/kernel_metadata.py [--lib path/to/libcilqr_amd.so] [--out profiles/rNN_kernel_metadata.json] [--loops]

1. unbundles the gfx950 code object (llvm-objdump --offloading on a temporary copy),
2. reads every kernel's descriptor metadata (llvm-readelf --notes): VGPR / SGPR counts, VGPR / SGPR spill counts,
   private segment (scratch) bytes, LDS bytes,
3. disassembles each k_solve variant and, for every INNERMOST loop of its body (a backward branch whose span holds
   no other backward branch), counts instructions, FP64 vector ops, scratch accesses and v_readlane / v_writelane
   (how the compiler reloads spilled scalars) and names the loop by what it contains: `backward_step` (ds_bpermute +
   v_rcp_f64: the Riccati step), `rollout_step` (buffer_store_dwordx4 into the slab), `cost_rows`, ...

The table in DESIGN.md section 4 ("Register allocation") is generated from this file's output:
  scripts/kernel_metadata.py --markdown
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    """c++filt when it is there; otherwise the template arguments of k_solve are decoded from the mangled name
    (Lb0E / Lb1E / Li<n>E) — enough for this file's purposes."""
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), text=True, capture_output=True,
                                 check=True).stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except Exception:
            pass
    res = {}
    for n in names:
        m = re.match(r"_Z\d+(k_\w+?)I((?:L[bi]\d+E)+)E", n)
        if m:
            args = re.findall(r"L([bi])(\d+)E", m.group(2))
            txt = ", ".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args)
            res[n] = f"void {m.group(1)}<{txt}>(...)"
        else:
            m2 = re.match(r"_Z\d+(k_[a-z_]+)", n)
            res[n] = (m2.group(1) if m2 else n) + "(...)"
    return res


def extract_code_objects(lib, tmp):
    """every gfx950 code object bundled into the library (one per compilation unit)"""
    dst = os.path.join(tmp, "lib.so")
    shutil.copy(lib, dst)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], check=True, stdout=subprocess.DEVNULL,
                   cwd=tmp)
    cos = sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)
    if not cos:
        raise RuntimeError("no gfx950 code object inside " + lib)
    return cos


def kernel_notes(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, text=True,
                         capture_output=True).stdout
    kernels, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:
            cur = {"agpr_count": int(m.group(1))}
            kernels.append(cur)
            continue
        if cur is None:
            continue
        m = re.match(r"\s+\.(name|symbol):\s+(\S+)", line)
        if m:
            cur[m.group(1)] = m.group(2)
            continue
        m = re.match(r"\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|"
                     r"group_segment_fixed_size|max_flat_workgroup_size|kernarg_segment_size):\s+(\d+)", line)
        if m:
            cur[m.group(1)] = int(m.group(2))
    return [k for k in kernels if "name" in k]


INSTR = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*(.*)$")


def disassemble(co, symbol):
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn",
                          "--disassemble-symbols=" + symbol, co], check=True, text=True, capture_output=True).stdout
    ins = []
    for line in txt.splitlines():
        m = INSTR.match(line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(4)))
    return ins


def device_functions(co, patterns=("grp_", "rollout_group", "ool_")):
    """the out-of-line device functions of the grouped build (no kernel descriptor: names from the symbol table)"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--symbols", "--wide", co], check=True, text=True,
                         capture_output=True).stdout
    syms = []
    for ln in txt.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC" and f[6] != "UND":
            syms.append(f[7])
    return sorted({s_ for s_ in syms if any(p_ in s_ for p_ in patterns) and not s_.endswith(".kd") and "k_solve" not in s_})


def is_fp64_valu(op):
    return op.startswith("v_") and "_f64" in op


def classify(ops):
    c = lambda pred: sum(1 for o in ops if pred(o))
    n_bperm = c(lambda o: o.startswith("ds_bpermute"))
    n_rcp = c(lambda o: o.startswith("v_rcp_f64"))
    n_bst = c(lambda o: o.startswith("buffer_store"))
    n_exp = c(lambda o: o.startswith("v_ldexp_f64"))
    n_dsr = c(lambda o: o.startswith("ds_read") or o.startswith("ds_load"))
    if n_bperm >= 8 and n_rcp >= 1:
        return "backward_step"
    if n_bst >= 3:
        return "rollout_step"
    if n_exp >= 4:
        return "cost_rows"
    if n_dsr >= 8 and c(is_fp64_valu) >= 8 and len(ops) < 200:
        return "ordered_sum"
    return "other"


def innermost_loops(ins):
    """loops = (target, branch) address pairs of backward branches; innermost = holds no other backward branch."""
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    base = ins[0][0] if ins else 0
    loops = []
    for i, (a, op, rest) in enumerate(ins):
        if not (op.startswith("s_cbranch") or op == "s_branch"):
            continue
        mm = re.search(r"<[^>+]+(?:\+0x([0-9A-Fa-f]+))?>", rest)
        if not mm:
            continue
        tgt = base + (int(mm.group(1), 16) if mm.group(1) else 0)
        if tgt <= a and tgt in addr_index:
            loops.append((addr_index[tgt], i))
    inner = []
    for (s, e) in loops:
        if not any((s2, e2) != (s, e) and s <= s2 and e2 <= e for (s2, e2) in loops):
            inner.append((s, e))
    return inner


def loop_report(ins):
    rep = []
    for (s, e) in innermost_loops(ins):
        ops = [op for (_, op, _) in ins[s:e + 1]]
        if len(ops) < 24:
            continue
        rep.append({
            "kind": classify(ops),
            "instructions": len(ops),
            "valu": sum(1 for o in ops if o.startswith("v_")),
            "fp64_valu": sum(1 for o in ops if is_fp64_valu(o)),
            "scratch": sum(1 for o in ops if o.startswith("scratch_")),
            "readlane_writelane": sum(1 for o in ops if o.startswith("v_readlane") or o.startswith("v_writelane")),
            "lds": sum(1 for o in ops if o.startswith("ds_")),
            "global_or_buffer": sum(1 for o in ops if o.startswith("global_") or o.startswith("buffer_")),
        })
    return rep


def variant_of(dem):
    m = re.search(r"k_solve<(.*?)>\(", dem)
    if m:
        return m.group(1)
    m = re.search(r"k_solve_grp<(.*?)>\(", dem)  # the grouped builds: <NC, G>
    return "grp:" + m.group(1) if m else None


TPL = ("DBG", "NCH", "ALM", "HELP", "PROF", "WPS", "NTP", "NC", "LG", "SHARE", "RES", "LOOP")


def describe(variant):
    if variant.startswith("grp:"):
        parts = [p.strip() for p in variant[4:].split(",")]
        nc, g = parts[0], parts[1]
        loop = len(parts) > 2 and parts[2] == "true"
        return (f"grouped: {g} trajectories per wavefront, waves/SIMD=2" + (f" N={nc}" if nc != "0" else "")
                + (" closed-loop" if loop else ""))
    parts = [p.strip() for p in variant.split(",")]
    kv = dict(zip(TPL, parts))
    tags = []
    if kv.get("DBG") == "true":
        tags.append("debug")
    if kv.get("PROF") == "true":
        tags.append("profiling")
    if kv.get("ALM") == "true":
        tags.append("alm")
    tags.append("helper" if kv.get("HELP") == "true" else "lone")
    tags.append("rows/lane=" + kv.get("NCH", "?"))
    tags.append("waves/SIMD=" + kv.get("WPS", "?"))
    if kv.get("NC", "0") != "0":
        tags.append("N=" + kv["NC"])
    if kv.get("LG") == "true":
        tags.append("global-expansion")
    if kv.get("SHARE") == "true":
        tags.append("share")
    if kv.get("RES") == "true":
        tags.append("resumable")
    if kv.get("LOOP") == "true":
        tags.append("closed-loop")
    return " ".join(tags)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "toy-example-of-ilqr_amd", "libcilqr_amd.so"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--loops", action="store_true", help="disassemble the k_solve variants and report their innermost loops")
    ap.add_argument("--markdown", action="store_true", help="print the DESIGN.md table instead of JSON")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        cos = extract_code_objects(a.lib, tmp)
        ks = []
        for co in cos:
            for k in kernel_notes(co):
                k["_co"] = co
                ks.append(k)
        dem = demangle([k["name"] for k in ks])
        out = {"library": os.path.relpath(a.lib, ROOT), "library_bytes": os.path.getsize(a.lib),
               "code_objects": len(cos), "code_object_bytes": sum(os.path.getsize(c) for c in cos), "kernels": []}
        for k in ks:
            d = dem[k["name"]]
            rec = {"kernel": d.split("(")[0], "vgpr": k.get("vgpr_count"), "sgpr": k.get("sgpr_count"),
                   "vgpr_spills": k.get("vgpr_spill_count"), "sgpr_spills": k.get("sgpr_spill_count"),
                   "scratch_bytes": k.get("private_segment_fixed_size"), "static_lds_bytes": k.get("group_segment_fixed_size"),
                   "max_threads": k.get("max_flat_workgroup_size")}
            v = variant_of(d)
            if v:
                rec["variant"] = describe(v)
                if a.loops or a.markdown:
                    ins = disassemble(k["_co"], k["name"])
                    ops = [op for (_, op, _) in ins]
                    rec["instructions"] = len(ins)
                    rec["static_counts"] = {
                        "fp64_valu": sum(1 for o in ops if is_fp64_valu(o)),
                        "scratch": sum(1 for o in ops if o.startswith("scratch_")),
                        "readlane_writelane": sum(1 for o in ops if o.startswith("v_readlane") or o.startswith("v_writelane")),
                        "ds_bpermute": sum(1 for o in ops if o.startswith("ds_bpermute")),
                        "mfma": sum(1 for o in ops if "mfma" in o),
                        "div_scale_f64": sum(1 for o in ops if o.startswith("v_div_scale_f64")),
                    }
                    rec["innermost_loops"] = loop_report(ins)
            out["kernels"].append(rec)
        # the grouped build's phases are functions of their own (cilqr_group.hpp): instructions, scratch accesses (their
        # prologue / epilogue save callee-saved registers) and what their innermost loops hold
        out["functions"] = []
        if a.loops or a.markdown:
            for co in cos:
                syms = device_functions(co)
                names = demangle(syms)
                for sy in syms:
                    ins = disassemble(co, sy)
                    ops = [op for (_, op, _) in ins]
                    if not ops:
                        continue
                    vmax = 0
                    for (_, _, rest) in ins:
                        pass
                    out["functions"].append({
                        "function": names[sy].split("(")[0], "instructions": len(ops),
                        "scratch": sum(1 for o in ops if o.startswith("scratch_")),
                        "readlane_writelane": sum(1 for o in ops if o.startswith("v_readlane") or o.startswith("v_writelane")),
                        "innermost_loops": loop_report(ins)})
    if a.markdown:
        print("| build of `k_solve` | VGPR | VGPR spills | SGPR spills | scratch B | instructions | backward step: instr / scratch / lane moves | rollout loops: count, scratch / lane moves inside |")
        print("|---|---|---|---|---|---|---|---|")
        for r in out["kernels"]:
            if "variant" not in r:
                continue
            bw = [l for l in r.get("innermost_loops", []) if l["kind"] == "backward_step"]
            ro = [l for l in r.get("innermost_loops", []) if l["kind"] == "rollout_step"]
            bcell = "; ".join(f'{l["instructions"]} / {l["scratch"]} / {l["readlane_writelane"]}' for l in bw[-1:]) or "-"
            rcell = f'{len(ro)}, {sum(l["scratch"] for l in ro)} / {sum(l["readlane_writelane"] for l in ro)}' if ro else "-"
            print(f'| {r["variant"]} | {r["vgpr"]} | {r["vgpr_spills"]} | {r["sgpr_spills"]} | {r["scratch_bytes"]} | '
                  f'{r.get("instructions", "")} | {bcell} | {rcell} |')
        return
    txt = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
        print("wrote", a.out, len(out["kernels"]), "kernels")
    else:
        print(txt)


if __name__ == "__main__":
    main()
