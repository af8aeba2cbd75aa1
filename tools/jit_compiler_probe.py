#!/usr/bin/env python3
"""GPU: what the process's hipRTC compiler does to a run-time instantiated kernel (round 6).  PyTorch's ROCm wheel brings its own
libhiprtc / libamd_comgr (ROCm 7.0 inside torch 2.10); once torch is imported they serve every later dlopen by soname -- this library's
too.  The per-instance-data form of (20,8,10) (bench.py's hetero_20_8_10 entry), 32 768 instances, one cold solve, ms:

    python tools/jit_compiler_probe.py            four child processes: torch first or not x prebuilt store on / off
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
if sys.argv[2] == "torch":
    import torch
    torch.zeros(1, device="cuda")
import bench_configs as bc
import tinympc_amd as tm
e, _ = bc.hetero_cell(20, 8, 10, B=32768)
import ctypes
ver = (ctypes.c_int * 2)()
try:
    h = ctypes.CDLL("libhiprtc.so"); h.hiprtcVersion(ctypes.byref(ver, 0), ctypes.byref(ver, 4))
except Exception:
    pass
maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "hiprtc" in ln or "comgr" in ln})
print("@@" + json.dumps(dict(ms=e["ms"], hiprtc=list(ver), libs=maps)))
'''


def main():
    print("| torch imported first | prebuilt store | hipRTC the process resolves | ms | compiler libraries mapped |")
    print("|---|---|---|---|---|")
    for first in ("none", "torch"):
        for pre in ("on", "0"):
            env = dict(os.environ)
            env.pop("TINYMPC_AMD_JIT_CACHE", None)
            if pre == "0":
                env["TINYMPC_AMD_JIT_PREBUILT"] = "0"
            p = subprocess.run([sys.executable, "-c", CHILD, ROOT, first], capture_output=True, text=True, env=env, timeout=600)
            r = [json.loads(ln[2:]) for ln in p.stdout.splitlines() if ln.startswith("@@")]
            if not r:
                print("| %s | %s | failed: %s |" % (first, pre, (p.stderr or "")[-200:].replace("\n", " ")))
                continue
            r = r[0]
            print("| %s | %s | %s | %.2f | %s |" % ("yes" if first == "torch" else "no", "on" if pre == "on" else "off", ".".join(map(str, r["hiprtc"])), r["ms"],
                                                  ", ".join(r["libs"])), flush=True)


if __name__ == "__main__":
    main()
