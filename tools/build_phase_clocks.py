#!/usr/bin/env python3
"""Experiment build: libtinympc_amd_clk.so = the product library with the (12,4,10) register kernel instrumented with
wall_clock64() phase counters (100 MHz) written into the four residual outputs: wave prologue, record load,
compute (all iterations of the tile, i.e. of its slowest row), write-back; read them with tools/phase_clocks.py.  Patches a COPY of
admm_kernel.hip.h; nothing in csrc/ is modified."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tinympc_amd", "csrc")
TMP = "/tmp/clkbuild"
shutil.rmtree(TMP, ignore_errors=True)
os.makedirs(TMP + "/_gen")
for f in os.listdir(SRC):
    if f.endswith((".h", ".hpp", ".hip")):
        shutil.copy(os.path.join(SRC, f), TMP)
shutil.copy(os.path.join(SRC, "_gen", "u_12_4_10.hip"), TMP + "/_gen")
p = TMP + "/admm_kernel.hip.h"
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("    const int lane = threadIdx.x & 63;\n    const int j = lane & 15;",
    "    const unsigned long long clk_entry = wall_clock64();\n    const int lane = threadIdx.x & 63;\n    const int j = lane & 15;")
rep("    const int ninst = P.index ? *P.count : P.batch;\n",
    "    __builtin_amdgcn_s_waitcnt(0);\n    const unsigned long long clk_pro = wall_clock64();\n    const int ninst = P.index ? *P.count : P.batch;\n")
rep("            const double* het = nullptr;\n", "            const unsigned long long c0 = wall_clock64();\n            const double* het = nullptr;\n")
rep("            int iter = 0, solved = 0, checked = 0;\n",
    "            __builtin_amdgcn_s_waitcnt(0);\n            const unsigned long long c1 = wall_clock64();\n            int iter = 0, solved = 0, checked = 0;\n")
rep("            // ---- write back (coalesced) -------------------------------------------------------\n",
    "            const unsigned long long c2 = wall_clock64();\n")
rep("            const double ps = grp_max16(is_state ? rp : 0.0), pi = grp_max16(is_input ? rp : 0.0);\n",
    "            __builtin_amdgcn_s_waitcnt(0);\n            const unsigned long long c3 = wall_clock64();\n"
    "            const double ps = grp_max16(is_state ? rp : 0.0), pi = grp_max16(is_input ? rp : 0.0);\n")
rep("                double4 rr = make_double4(ps, pi, ds, di);",
    "                double4 rr = make_double4((double)(clk_pro - clk_entry), (double)(c1 - c0), (double)(c2 - c1), (double)(c3 - c2));")
open(p, "w").write(s)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-but-set-variable -Wno-unused-variable".split()
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", TMP + "/_gen/u_12_4_10.hip", "-o", TMP + "/k_clk.o"])
objs = [os.path.join(SRC, "_gen", f) for f in os.listdir(SRC + "/_gen") if f.endswith(".o") and f != "u_12_4_10.o" and "_chk" not in f]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-Wl,-Bsymbolic", "-o",
                       os.path.join(ROOT, "tinympc_amd", "libtinympc_amd_clk.so"), *objs, TMP + "/k_clk.o", "-ldl"])
print("built libtinympc_amd_clk.so")
