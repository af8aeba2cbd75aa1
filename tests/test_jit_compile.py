"""CPU: the kernel variants that are never compiled into the library (they are instantiated with hipRTC on first use,
csrc/jit.hip) must at least COMPILE for gfx950 -- hipRTC needs no GPU.  Catches a header edit that breaks a run-time-only
variant before any GPU box sees it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tinympc_amd", "csrc")

HARNESS = r'''
#include <hip/hiprtc.h>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
static std::string rd(const std::string& p) { std::ifstream f(p); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
int main(int argc, char** argv) {
    const std::string dir = argv[1];
    std::string hdr = rd(dir + "/admm_kernel.hip.h"), th = rd(dir + "/tile_kernel.hip.h");
    for (const char* inc : {"#include <hip/hip_runtime.h>", "#include <stdint.h>"}) { size_t p = hdr.find(inc); if (p != std::string::npos) hdr.replace(p, std::string(inc).size(), ""); }
    const char* hn[] = {"admm_kernel.hip.h", "tile_kernel.hip.h"};
    const char* hs[] = {hdr.c_str(), th.c_str()};
    int bad = 0;
    for (int i = 2; i < argc; ++i) {
        hiprtcProgram prog;
        hiprtcCreateProgram(&prog, "#include \"tile_kernel.hip.h\"\n", "jit.hip", 2, hs, hn);
        hiprtcAddNameExpression(prog, argv[i]);
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
        const hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
        if (r != HIPRTC_SUCCESS) {
            size_t n = 0; hiprtcGetProgramLogSize(prog, &n); std::string log(n, 0); hiprtcGetProgramLog(prog, &log[0]);
            std::printf("FAILED %s\n%s\n", argv[i], log.substr(0, 1500).c_str());
            ++bad;
        } else std::printf("ok %s\n", argv[i]);
        hiprtcDestroyProgram(&prog);
    }
    return bad;
}
'''

VARIANTS = [
    "tinympc_amd::admm_solve_kernel<5, 3, 7, false, false, 2, 0, false, 4>",       # an unseen shape
    "tinympc_amd::admm_solve_kernel<12, 4, 10, true, true, 2, 3, false, 4>",       # cone x debug x both half-space families
    "tinympc_amd::admm_solve_kernel<12, 4, 10, false, false, 2, 1, true, 8>",      # heterogeneous x half-spaces, 8 per knot
    "tinympc_amd::admm_tile_kernel<20, 4, 10, 2, 1, true>",                         # cones on a wide shape
    "tinympc_amd::admm_tile_kernel<6, 2, 60, 1, 2, false>",                         # a long horizon outside tile_dims.txt
    "tinympc_amd::admm_tile_kernel<20, 4, 10, 2, 1, true, 3, 8>",                   # cones + both half-space families, 8 per knot, wide
    "tinympc_amd::admm_tile_kernel<8, 2, 50, 1, 2, false, 1, 4>",                   # static half-spaces on a long horizon
]


def test_runtime_only_kernel_variants_compile(tmp_path):
    if not (os.path.exists("/opt/rocm/include/hip/hiprtc.h") and shutil.which("g++")):
        pytest.skip("hipRTC development files not installed")
    src = tmp_path / "rtc.cpp"
    src.write_text(HARNESS)
    exe = tmp_path / "rtc"
    subprocess.check_call(["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src), "-L/opt/rocm/lib", "-lhiprtc",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    p = subprocess.run([str(exe), CSRC] + VARIANTS, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
