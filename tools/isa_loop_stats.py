#!/usr/bin/env python3
"""CPU only: the instruction mix of the ADMM iteration loop of one kernel instantiation, from the gfx950 assembly hipcc
emits -- the feedback loop for kernel work that needs no GPU (registers, scratch, and how many of the instructions one
wave-iteration issues are FP64 arithmetic).

    python tools/isa_loop_stats.py tile 20 8 10 2 1 [extra template args ...]     admm_tile_kernel<20,8,10,2,1,...>
    python tools/isa_loop_stats.py row 6 3 10 true false 2                        admm_solve_kernel<6,3,10,true,false,2>
    ... --fused        define TINYMPC_FUSED_NX / _NU as the Makefile does for the compiled-in units
    ... --keep FILE    keep the assembly

The iteration loop is taken to be the innermost loop (no backward branch inside it) that holds the most v_fmac_f64 / v_fma_f64
instructions.  Classes: fp64 (v_*_f64), dpp-fma (subset of fp64), mov (v_mov / v_accvgpr), perm (v_permlane / ds_bpermute /
v_readlane), lds (ds_*), vmem (global / buffer / scratch), salu, nop/wait.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tinympc_amd", "csrc")


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_") and "_f64" in op:
        return "fp64"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "mov"
    if op.startswith(("v_permlane", "ds_bpermute", "v_readlane", "v_readfirstlane", "v_writelane", "ds_swizzle")):
        return "perm"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op in ("s_nop", "s_waitcnt", "s_sleep") or op.startswith("s_wait"):
        return "nop/wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu-other"
    return "other"


def analyse(asm, want=None):
    text = open(asm).read()
    res = []
    for m in re.finditer(r"^(_ZN11tinympc_amd\w+):[^\n]*\n", text, re.M):
        sym = m.group(1)
        end = text.find(".Lfunc_end", m.end())
        body = text[m.end():end]
        if "s_endpgm" not in body:
            continue
        lines = []
        labels = {}
        for raw in body.splitlines():
            l = raw.split(";")[0].strip()
            if not l or l.startswith("."):
                if l.endswith(":"):
                    labels[l[:-1]] = len(lines)
                continue
            if l.endswith(":"):
                labels[l[:-1]] = len(lines)
                continue
            lines.append(l)
        loops = []
        for i, l in enumerate(lines):
            mm = re.match(r"s_c?branch\w*\s+(\S+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
                loops.append((labels[mm.group(1)], i))
        inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        best, best_n = None, -1
        for a, b in inner:
            n = sum(1 for l in lines[a:b + 1] if re.match(r"v_fma?c?_f64", l))
            if n > best_n:
                best, best_n = (a, b), n
        meta = {}
        mres = re.search(re.escape(sym) + r"\n.*?\.end_amdhsa_kernel", text[end:], re.S)
        blk = text[text.find(".amdhsa_kernel " + sym):]
        for key in ("next_free_vgpr", "accum_offset", "private_segment_fixed_size", "group_segment_fixed_size"):
            k = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", blk)
            if k:
                meta[key] = int(k.group(1))
        res.append((sym, lines, best, meta))
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    fused = "--fused" in sys.argv
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    if keep:
        args.remove(keep)
    kind, targs = args[0], args[1:]
    name = "admm_tile_kernel" if kind == "tile" else "admm_solve_kernel"
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "u.hip")
        with open(src, "w") as f:
            if fused:
                f.write("#define TINYMPC_FUSED_NX %s\n#define TINYMPC_FUSED_NU %s\n" % (targs[0], targs[1]))
            f.write('#include "%s/kernel_entry.hpp"\n#include "%s/tile_kernel.hip.h"\n' % (CSRC, CSRC))
            f.write("namespace tinympc_amd { template __global__ void %s<%s>(const SolveArgs); }\n" % (name, ", ".join(targs)))
        out = keep or os.path.join(tmp, "u.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out])
        for sym, lines, loop, meta in analyse(out):
            print("%s<%s>%s" % (name, ", ".join(targs), " [fused blocks]" if fused else ""))
            print("  registers: vgpr+agpr %s (accum_offset %s), scratch %s B/lane, LDS %s B" %
                  (meta.get("next_free_vgpr"), meta.get("accum_offset"), meta.get("private_segment_fixed_size"), meta.get("group_segment_fixed_size")))
            if not loop:
                print("  no loop found")
                continue
            body = lines[loop[0]:loop[1] + 1]
            c = collections.Counter(classify(l) for l in body)
            dpp = sum(1 for l in body if "_f64_dpp" in l)
            total = len(body)
            print("  iteration loop: %d instructions; fp64 %d (dpp-fma %d), valu-other %d, mov %d, perm %d, lds %d, vmem %d, scratch %d, salu %d, nop/wait %d" %
                  (total, c["fp64"], dpp, c["valu-other"], c["mov"], c["perm"], c["lds"], c["vmem"], c["scratch"], c["salu"], c["nop/wait"]))
            ops = collections.Counter(l.split()[0] for l in body)
            print("  top opcodes: " + ", ".join("%s %d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()
