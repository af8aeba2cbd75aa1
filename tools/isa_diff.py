#!/usr/bin/env python3
"""CPU only: does a kernel-header edit change the machine code of the compiled-in kernels?  Compiles one (nx, nu, N) translation
unit for gfx950 from the working tree and from a git revision, and compares the instruction streams variant by variant
(labels, comments and directives ignored).  An edit that is meant to touch only some variants can be proven not to touch the
headline kernel without a GPU -- the measured numbers of an unchanged instruction stream stay valid.
    python tools/isa_diff.py [rev = HEAD] [nx nu N = 12 4 10]"""
import collections
import difflib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tinympc_amd", "csrc")
HEADERS = ("admm_kernel.hip.h", "kernel_entry.hpp", "tile_kernel.hip.h")


def assemble(srcdir, dims, out):
    gen = os.path.join(srcdir, "_gen")
    os.makedirs(gen, exist_ok=True)
    name = "u_%d_%d_%d" % dims
    with open(os.path.join(gen, name + ".hip"), "w") as f:
        f.write("#define TINYMPC_FUSED_NX %d\n#define TINYMPC_FUSED_NU %d\n" % dims[:2])      # as csrc/Makefile writes the unit
        f.write('#include "../kernel_entry.hpp"\n')
        f.write("namespace tinympc_amd { extern const KernelEntry kentry_%d_%d_%d = KERNELS_FOR(%d, %d, %d); }\n" % (dims + dims))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", name + ".hip", "-o", out],
                          cwd=gen, stderr=subprocess.DEVNULL)


def kernels(path):
    text = open(path).read()
    out = {}
    for m in re.finditer(r"^(_ZN11tinympc_amd\w+):[^\n]*\n", text, re.M):
        sym = m.group(1)
        end = text.find(".Lfunc_end", m.end())
        if end < 0 or "s_endpgm" not in text[m.end():end]:
            continue                                            # a data symbol, not a kernel
        lines = [l.split(";")[0].strip() for l in text[m.end():end].splitlines()]
        out[sym] = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]
    return out


def label(sym):
    a = [int(v) for v in re.findall(r"L[ib](\d+)E", sym)]
    return "<%d,%d,%d soc%d dbg%d mode%d lin%d het%d kmax%d>" % tuple(a[:9]) if len(a) >= 9 else sym


def main():
    rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
    dims = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (12, 4, 10)
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(old)
        for h in HEADERS:
            with open(os.path.join(old, h), "w") as f:
                f.write(subprocess.check_output(["git", "show", "%s:tinympc_amd/csrc/%s" % (rev, h)], cwd=ROOT, text=True))
        new = os.path.join(tmp, "new")
        os.makedirs(new)
        for h in HEADERS:
            with open(os.path.join(new, h), "w") as f:
                f.write(open(os.path.join(CSRC, h)).read())
        assemble(old, dims, os.path.join(tmp, "old.s"))
        assemble(new, dims, os.path.join(tmp, "new.s"))
        a, b = kernels(os.path.join(tmp, "old.s")), kernels(os.path.join(tmp, "new.s"))
    changed = 0
    for sym in sorted(set(a) | set(b)):
        x, y = a.get(sym), b.get(sym)
        if x == y:
            print(f"identical  {label(sym)}  {len(x)} instructions")
            continue
        changed += 1
        if x is None or y is None:
            print(f"{'added' if x is None else 'removed'}    {label(sym)}")
            continue
        delta = collections.Counter(l.split()[0] for l in y)
        delta.subtract(collections.Counter(l.split()[0] for l in x))
        top = ", ".join(f"{k} {v:+d}" for k, v in sorted(delta.items(), key=lambda kv: -abs(kv[1]))[:6] if v)
        n = sum(1 for l in difflib.unified_diff(x, y, lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---"))
        print(f"CHANGED    {label(sym)}  {len(x)} -> {len(y)} instructions, {n} differing lines ({top})")
    print(f"{changed} of {len(set(a) | set(b))} variants of {dims} differ from {rev}")
    return 1 if changed else 0


if __name__ == "__main__":
    sys.exit(main())
