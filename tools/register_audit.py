#!/usr/bin/env python3
"""CPU only: register / scratch / occupancy report of every kernel instantiation compiled into the library
(hipcc -Rpass-analysis=kernel-resource-usage on tinympc_amd/csrc/_gen/{k,t}_*.hip).  Scratch > 0 = spilled to memory.
    python tools/register_audit.py [--all] > profiles/rNN_register_audit.md"""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tinympc_amd", "csrc", "_gen")


def usage(src):
    with tempfile.TemporaryDirectory() as tmp:
        p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                            "-c", src, "-o", os.path.join(tmp, "o.o")], capture_output=True, text=True, cwd=GEN)
    rows = []
    for b in p.stderr.split("Function Name:")[1:]:
        name = b.split()[0]
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        rows.append(dict(kind="tile" if "tile_kernel" in name else "one-row", args=[int(v) for _, v in re.findall(r"L([ib])(\d+)E", name)],
                         vgpr=g("VGPRs"), agpr=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"), occ=g(r"Occupancy \[waves/SIMD\]"),
                         lds=g(r"LDS Size \[bytes/block\]")))
    return rows


def main():
    srcs = sorted(glob.glob(os.path.join(GEN, "u_*.hip")))
    if not srcs:
        sys.exit("build the library first (tinympc_amd/csrc/_gen is empty)")
    with concurrent.futures.ThreadPoolExecutor(8) as ex:
        rows = [r for rs in ex.map(usage, srcs) for r in rs]
    show_all = "--all" in sys.argv
    print(f"{len(rows)} kernel instantiations, {sum(r['scratch'] > 0 for r in rows)} with scratch (memory spills)\n")
    print("one-row kernel `admm_solve_kernel<NX,NU,N,SOC,DBG,MODE,LIN,HET,KMAX,ADAPT,UB,HALF,PF>` -- default variants (box constraints, MODE 2; UB = the form launched when the box is the same at every knot), cone and adaptive-rho variants:\n")
    print("| (nx,nu,N) | variant | VGPR | AGPR | scratch B/lane | waves/SIMD | LDS B |")
    print("|---|---|---|---|---|---|---|")
    for r in sorted((r for r in rows if r["kind"] == "one-row"), key=lambda r: (r["args"][:3], r["args"][3:])):
        a = r["args"]
        # <NX,NU,N, SOC,DBG,MODE,LIN,HET, KMAX, ADAPT, UB>
        extra = tuple(a[9:11]) if len(a) >= 11 else (0, 0)
        tag = {((0, 0, 2, 0, 0), (0, 0)): "box", ((0, 0, 2, 0, 0), (0, 1)): "box, knot-invariant (UB)", ((1, 0, 2, 0, 0), (0, 0)): "cone",
               ((0, 0, 2, 0, 0), (1, 0)): "adaptive rho"}.get((tuple(a[3:8]), extra))
        if tag is None and not show_all:
            continue
        tag = tag or f"soc{a[3]} dbg{a[4]} mode{a[5]} lin{a[6]} het{a[7]} adapt{extra[0]} ub{extra[1]}"
        # (round 4: HALF rows, round 6: the PREFETCH form -- template arguments 12 and 13)
        if len(a) >= 12 and a[11]:
            tag += ", HALF rows"
        if len(a) >= 13 and a[12]:
            tag += ", PREFETCH form"
        print(f"| ({a[0]},{a[1]},{a[2]}) | {tag} | {r['vgpr']} | {r['agpr']} | {r['scratch']} | {r['occ']} | {r['lds']} |")
    print("\ntile kernel `admm_tile_kernel<NX,NU,N,W,R>`:\n")
    print("| (nx,nu,N) | W x R | form | VGPR | AGPR | scratch B/lane | waves/SIMD | LDS B |")
    print("|---|---|---|---|---|---|---|---|")
    for r in sorted((r for r in rows if r["kind"] == "tile"), key=lambda r: r["args"]):
        a = r["args"]
        form = "box in registers (UB)" if len(a) >= 9 and a[8] else "box table in LDS"
        print(f"| ({a[0]},{a[1]},{a[2]}) | {a[3]} x {a[4]} | {form} | {r['vgpr']} | {r['agpr']} | {r['scratch']} | {r['occ']} | {r['lds']} |")


if __name__ == "__main__":
    main()
