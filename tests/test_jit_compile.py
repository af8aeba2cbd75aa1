"""CPU: the kernel variants that are never compiled into the library (they are instantiated with hipRTC on first use,
csrc/jit.hip) must at least COMPILE for gfx950 -- hipRTC needs no GPU.  Goes through the library's own tiny_jit_compile(), so
the embedded copy of the kernel headers is what gets compiled: catches a header edit that breaks a run-time-only variant
before any GPU box sees it.  Also covers the on-disk cache of the code objects (TINYMPC_AMD_JIT_CACHE)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    "tinympc_amd::admm_solve_kernel<5, 3, 7, false, false, 2, 0, false, 4>",       # an unseen shape
    "tinympc_amd::admm_solve_kernel<12, 4, 10, true, true, 2, 3, false, 4>",       # cone x debug x both half-space families
    "tinympc_amd::admm_solve_kernel<12, 4, 10, false, false, 2, 1, true, 8>",      # heterogeneous x half-spaces, 8 per knot
    "tinympc_amd::admm_tile_kernel<20, 4, 10, 2, 1, 3>",                            # cones on a wide shape
    "tinympc_amd::admm_tile_kernel<6, 2, 60, 1, 2, 0>",                            # a long horizon outside tile_dims.txt
    "tinympc_amd::admm_tile_kernel<20, 4, 10, 2, 1, 1, 3, 8>",                      # cones + both half-space families, 8 per knot, wide
    "tinympc_amd::admm_tile_kernel<8, 2, 50, 1, 2, 0, 1, 4>",                      # static half-spaces on a long horizon
]

CHILD = r'''
import json, sys, time
sys.path.insert(0, sys.argv[1])
import tinympc_amd as tm
out = []
for name in sys.argv[2:]:
    t = time.time()
    try:
        n, hit = tm.jit_compile(name)
        out.append({"name": name, "bytes": n, "hit": hit, "s": time.time() - t})
    except RuntimeError as e:
        out.append({"name": name, "error": str(e)[:1500]})
print(json.dumps({"results": out, "used": tm.jit_used()}))
'''


def _compile(names, cache_dir=None):
    env = dict(os.environ)
    env.pop("TINYMPC_AMD_JIT_CACHE", None)
    if cache_dir is not None:
        env["TINYMPC_AMD_JIT_CACHE"] = str(cache_dir)
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT] + list(names), capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    return json.loads(p.stdout.strip().splitlines()[-1])


def _need_hiprtc():
    if not any(os.path.exists(p) for p in ("/opt/rocm/lib/libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7")):
        pytest.skip("libhiprtc not installed")


def test_runtime_only_kernel_variants_compile():
    _need_hiprtc()
    r = _compile(VARIANTS)
    bad = [x for x in r["results"] if "error" in x]
    assert not bad, bad
    assert all(x["bytes"] > 1000 and not x["hit"] for x in r["results"])
    assert sorted(r["used"]) == sorted(VARIANTS)


def test_a_name_that_is_not_a_kernel_is_refused():
    r = _compile(["tinympc_amd::something_else<1>", "tinympc_amd::admm_solve_kernel<5, 3, 7, nonsense>"])
    assert all("error" in x for x in r["results"]), r
    assert r["used"] == []


def test_disk_cache_of_code_objects(tmp_path):
    _need_hiprtc()
    name = VARIANTS[0]
    first = _compile([name], tmp_path)["results"][0]
    files = sorted(os.listdir(tmp_path))
    assert not first["hit"] and len(files) == 1 and files[0].startswith("tinympc_amd_") and files[0].endswith(".co")
    second = _compile([name], tmp_path)["results"][0]                     # a new process: loaded, not compiled
    assert second["hit"] and second["bytes"] == first["bytes"] and second["s"] < 0.5 * first["s"] + 0.2
    # a damaged file is not trusted: it is recompiled and replaced
    path = os.path.join(tmp_path, files[0])
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0xFF
    open(path, "wb").write(bytes(blob))
    third = _compile([name], tmp_path)["results"][0]
    assert not third["hit"] and third["bytes"] == first["bytes"]
    assert _compile([name], tmp_path)["results"][0]["hit"]
    assert sorted(os.listdir(tmp_path)) == files                           # no temporary files left behind
    # another instantiation gets its own file
    _compile([VARIANTS[4]], tmp_path)
    assert len(os.listdir(tmp_path)) == 2
    # an unwritable / missing directory only costs the caching
    r = _compile([name], os.path.join(tmp_path, "does", "not", "exist"))["results"][0]
    assert "error" not in r and not r["hit"]


PREBUILD_CHILD = r'''
import json, sys
sys.path.insert(0, sys.argv[1])
import tinympc_amd as tm
out = []
for name in sys.argv[2:]:
    a = tm.jit_prebuild(name)
    b = tm.jit_prebuild(name)
    n, hit = tm.jit_compile(name)
    out.append({"first": a, "second": b, "bytes": n, "hit": hit})
print(json.dumps(out))
'''


def test_prebuilt_store_of_code_objects(tmp_path):
    """Round 6: code objects compiled at BUILD time (tinympc_amd.build() -> tiny_jit_prebuild for the names of csrc/jit_prebuilt.txt)
    are found by name before anything is compiled -- keyed by the kernel sources, not by the hipRTC version of the process, so that a
    process which has loaded another ROCm's compiler (PyTorch's wheel) runs the build's code."""
    _need_hiprtc()
    import json
    name = VARIANTS[0]
    env = dict(os.environ, TINYMPC_AMD_JIT_PREBUILT=str(tmp_path))
    env.pop("TINYMPC_AMD_JIT_CACHE", None)
    p = subprocess.run([sys.executable, "-c", PREBUILD_CHILD, ROOT, name], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])[0]
    files = os.listdir(tmp_path)
    assert r["first"][0] > 1000 and r["second"][0] == 0 and r["hit"] and r["bytes"] == r["first"][0]
    assert len(files) == 1 and files[0].startswith("tinympc_amd_pre_") and os.path.basename(r["first"][1]) == files[0]
    # switched off: the same name is compiled
    env["TINYMPC_AMD_JIT_PREBUILT"] = "0"
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, name], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and not json.loads(p.stdout.strip().splitlines()[-1])["results"][0]["hit"]
    # the library's own store (tinympc_amd/jit_prebuilt/, filled by build()): every listed name is there
    listed = [ln.strip() for ln in open(os.path.join(ROOT, "tinympc_amd", "csrc", "jit_prebuilt.txt")) if ln.strip() and not ln.startswith("#")]
    assert len(listed) >= 20
    r = _compile(listed[:4])
    assert all("error" not in x and x["hit"] for x in r["results"]), r["results"]
