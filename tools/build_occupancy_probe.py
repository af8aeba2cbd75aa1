#!/usr/bin/env python3
"""Experiment build: libtinympc_amd_w1.so = the product library with the (12,4,10) register kernel compiled for ONE wave per
SIMD (amdgpu_waves_per_eu(1,1)) instead of two.  If one wave already saturates the SIMD's FP64 issue port, throughput does
not drop -- and a third wave (which would need <= 168 VGPRs) cannot add anything either.  Patches a COPY of
admm_kernel.hip.h; nothing in csrc/ is modified.  Run the bench with TINYMPC_AMD_LIB=.../libtinympc_amd_w1.so."""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tinympc_amd", "csrc")
TMP = "/tmp/w1build"
shutil.rmtree(TMP, ignore_errors=True)
os.makedirs(TMP + "/_gen")
for f in os.listdir(SRC):
    if f.endswith((".h", ".hpp", ".hip")):
        shutil.copy(os.path.join(SRC, f), TMP)
shutil.copy(os.path.join(SRC, "_gen", "u_12_4_10.hip"), TMP + "/_gen")
p = TMP + "/admm_kernel.hip.h"
s = open(p).read()
a = "    return (n <= 10 || 2 * ((soc ? 8 : 6) * n + 2 * nz + 8) + 40 <= 256) ? 2 : 1;"
assert a in s
s = s.replace(a, "    return 1;")
open(p, "w").write(s)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-but-set-variable -Wno-unused-variable".split()
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", TMP + "/_gen/u_12_4_10.hip", "-o", TMP + "/k_w1.o"])
objs = [os.path.join(SRC, "_gen", f) for f in os.listdir(SRC + "/_gen") if f.endswith(".o") and f != "u_12_4_10.o" and "_chk" not in f]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-Wl,-Bsymbolic", "-o",
                       os.path.join(ROOT, "tinympc_amd", "libtinympc_amd_w1.so"), *objs, TMP + "/k_w1.o", "-ldl"])
print("built libtinympc_amd_w1.so")
