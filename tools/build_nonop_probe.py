#!/usr/bin/env python3
"""Experiment build (UNSAFE, measurement only): libtinympc_amd_nonop.so = the product library with the leading `s_nop 1` of
every DPP block of the (12,4,10) kernel removed -- the upper bound of what fusing independent instructions into the heads of
the blocks could buy.  Results may be wrong (VALU write -> DPP read hazard); never ship.  Patches a COPY of the header."""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tinympc_amd", "csrc")
TMP = "/tmp/nonopbuild"
shutil.rmtree(TMP, ignore_errors=True)
os.makedirs(TMP + "/_gen")
for f in os.listdir(SRC):
    if f.endswith((".h", ".hpp", ".hip")):
        shutil.copy(os.path.join(SRC, f), TMP)
shutil.copy(os.path.join(SRC, "_gen", "u_12_4_10.hip"), TMP + "/_gen")
p = TMP + "/admm_kernel.hip.h"
s = open(p).read()
assert s.count('asm("s_nop 1\\n\\t" ') == 2
s = s.replace('asm("s_nop 1\\n\\t" ', 'asm("" ')
open(p, "w").write(s)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-but-set-variable -Wno-unused-variable".split()
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", TMP + "/_gen/u_12_4_10.hip", "-o", TMP + "/k.o"])
objs = [os.path.join(SRC, "_gen", f) for f in os.listdir(SRC + "/_gen") if f.endswith(".o") and f != "u_12_4_10.o" and "_chk" not in f]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-Wl,-Bsymbolic", "-o",
                       os.path.join(ROOT, "tinympc_amd", "libtinympc_amd_nonop.so"), *objs, TMP + "/k.o", "-ldl"])
print("built libtinympc_amd_nonop.so")
