#!/usr/bin/env python3
"""Experiment builds (timing only, results are WRONG by construction): copies of the product library whose (6,3,10) kernels have parts
of the cone step removed, to see what each part costs per iteration (tools/soc_iter_cost.py under TINYMPC_AMD_LIB):
  abl1  no projection passes          abl2  abl1 + the third pass takes vc = x + gc without reading LDS
  abl3  abl2 + no LDS writes of x + gc in the sweep
Patches a COPY of admm_kernel.hip.h; nothing in csrc/ is modified."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tinympc_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-but-set-variable -Wno-unused-variable".split()

def build(tag, patches):
    tmp = "/tmp/socabl_" + tag
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp + "/_gen")
    for f in os.listdir(SRC):
        if f.endswith((".h", ".hpp", ".hip")):
            shutil.copy(os.path.join(SRC, f), tmp)
    shutil.copy(os.path.join(SRC, "_gen", "u_6_3_10.hip"), tmp + "/_gen")
    p = tmp + "/admm_kernel.hip.h"
    s = open(p).read()
    for a, b in patches:
        assert a in s, a
        s = s.replace(a, b)
    open(p, "w").write(s)
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-c", tmp + "/_gen/u_6_3_10.hip", "-o", tmp + "/k.o"])
    objs = [os.path.join(SRC, "_gen", f) for f in os.listdir(SRC + "/_gen") if f.endswith(".o") and f != "u_6_3_10.o" and "_chk" not in f]
    out = os.path.join(ROOT, "tinympc_amd", "libtinympc_amd_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-Wl,-Bsymbolic", "-o", out, *objs, tmp + "/k.o", "-ldl"])
    print("built", out)

P1 = [("if (p > 0 && p >= soc_passes) break;            // wave-uniform", "break;")]
P2 = P1 + [("const double pv = sT[(grp * N + s) * 16 + j];", "const double pv = tc;")]
P3 = P2 + [("if constexpr (SOC) sT[(grp * N + i) * 16 + j] = fma(xi, socmask, GC[i]);", ""),
           ("sT[(grp * N + s) * 16 + j] = tc;", "")]
# abl_vp: v|z not held (timing only: what the registers of the cone kernel's VP array are worth)
PVP = [("VP[s] = warm ? P.slack_prev[off] : 0.0;", ""), ("dmax = resid_max<(N > 12)>(dmax, VP[s] - vn);", "dmax = resid_max<(N > 12)>(dmax, VN[s] - vn);"),
       ("dmax = resid_max<(N > 12)>(dmax, VP[i] - vn);", "dmax = resid_max<(N > 12)>(dmax, VN[i] - vn);"),
       ("for (int s = 0; s < N; ++s) VP[s] = VN[s];                      // :445-446", "for (int s = 0; s < 1; ++s) {}"),
       ("P.slack_prev[off] = VP[s];", "P.slack_prev[off] = VN[s];"), ("double X[N], G[N], VN[N], VP[N], QX[N], Dn[N - 1];", "double X[N], G[N], VN[N], QX[N], Dn[N - 1];")]
VARIANTS = {"abl1": P1, "abl2": P2, "abl3": P3, "abl_vp": PVP}
if __name__ == "__main__":
    for t in (sys.argv[1:] or VARIANTS):
        build(t, VARIANTS[t])
