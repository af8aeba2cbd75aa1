"""tinympc_amd -- MI355X-native batched TinyMPC ADMM solver (host-side mirror of the C ABI).

The product is ``libtinympc_amd.so`` (hand-written HIP for gfx950 + a C ABI, ``include/tinympc_amd.h``).
This package is a thin ctypes mirror of the reference's operator interface for the hot path
(``tiny_setup / tiny_set_bound_constraints / tiny_set_cone_constraints / tiny_update_settings /
tiny_set_x0 / tiny_set_x_ref / tiny_set_u_ref / tiny_solve``, reference
``src/tinympc/tiny_api.hpp:10-54``) with a leading batch axis.  There is no CPU fallback: if the
shared library is missing the import fails, and without a GPU ``TinyBatchSolver`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TINYMPC_AMD_LIB") or os.path.join(_HERE, "libtinympc_amd.so")   # override: A/B experiments
CSRC = os.path.join(_HERE, "csrc")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)

OK, ERR_DIM, ERR_NULL, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_HIP, ERR_ARG = 0, 1, -1, -2, -3, -4, -5
HOST, DEVICE, BROADCAST = 0, 1, 2

# TinyField (include/tinympc_amd.h)
FIELDS = ("x0", "Xref", "Uref", "x", "u", "vnew", "znew", "g", "y", "v", "z", "vcnew", "zcnew", "gc", "yc",
          "q", "r", "p", "d", "vlnew", "zlnew", "gl", "yl", "vlnew_tv", "zlnew_tv", "gl_tv", "yl_tv")
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}
STATE_FIELDS = {"Xref", "x", "vnew", "g", "v", "vcnew", "gc", "q", "p", "vlnew", "gl", "vlnew_tv", "gl_tv"}

PLAN_BYTES = 19 * 4 + 4 + 5 * 8 + 1024 * 4      # sizeof(TinyBatchPlan): 19 ints, padding, 5 doubles, the histogram
# every extern "C" symbol include/tinympc_amd.h declares (checked by tests/test_abi_symbols.py)
BATCH_SYMBOLS = (
    "tiny_batch_device_count", "tiny_batch_setup", "tiny_batch_setup_hetero", "tiny_batch_get_cache_instance",
    "tiny_batch_destroy", "tiny_batch_set_bound_constraints",
    "tiny_batch_set_cone_constraints", "tiny_batch_set_linear_constraints", "tiny_batch_set_tv_linear_constraints",
    "tiny_batch_update_settings", "tiny_batch_get_cache", "tiny_batch_set",
    "tiny_batch_get", "tiny_batch_reset", "tiny_batch_solve", "tiny_batch_solve_async", "tiny_batch_synchronize",
    "tiny_batch_get_status", "tiny_batch_reduce_stats", "tiny_batch_set_option", "tiny_batch_set_stream",
    "tiny_batch_phase", "tiny_batch_get_timing", "tiny_batch_get_step_log", "tiny_batch_set_reference_trajectory", "tiny_batch_last_error", "tiny_batch_supported_dims", "tiny_batch_algorithmic_bytes", "tiny_batch_kernel_path",
    "tiny_jit_compile", "tiny_jit_prebuild", "tiny_jit_used", "tiny_batch_allreduce_stats", "tiny_batch_stats_message",
    "tiny_rccl_unique_id", "tiny_rccl_comm_init_rank", "tiny_rccl_comm_destroy", "tiny_rccl_comm_count", "tiny_rccl_available", "tiny_reduce_stats_messages",
    "tiny_batch_get_option", "tiny_predict_split", "tiny_step_regroup_plan", "tiny_batch_get_plan", "tiny_batch_set_plan", "tiny_batch_set_cache", "tiny_batch_set_adaptive_rho", "tiny_batch_set_sensitivity", "tiny_batch_set_cache_state", "tiny_batch_get_cache_state")
GROUP_SYMBOLS = (
    "tiny_group_setup", "tiny_group_destroy", "tiny_group_shards", "tiny_group_shard", "tiny_group_shard_indices",
    "tiny_group_uses_rccl", "tiny_group_last_error", "tiny_group_set_bound_constraints", "tiny_group_set_cone_constraints",
    "tiny_group_set_linear_constraints", "tiny_group_set_tv_linear_constraints", "tiny_group_update_settings",
    "tiny_group_set_option", "tiny_group_set", "tiny_group_get", "tiny_group_reset", "tiny_group_solve",
    "tiny_group_solve_async", "tiny_group_synchronize", "tiny_group_get_status", "tiny_group_allreduce_stats")
REFERENCE_SYMBOLS = (
    "tiny_setup", "tiny_set_bound_constraints", "tiny_set_cone_constraints", "tiny_set_linear_constraints",
    "tiny_set_tv_linear_constraints", "tiny_precompute_and_set_cache",
    "tiny_solve", "solve", "tiny_update_settings", "tiny_set_default_settings", "tiny_set_x0", "tiny_set_x_ref",
    "tiny_set_u_ref", "tiny_solve_batch", "tiny_destroy", "tiny_initialize_sensitivity_matrices",
    "tiny_codegen", "tiny_codegen_with_sensitivity", "codegen_create_directories", "codegen_data_header", "codegen_data_source",
    "codegen_example",
    # the phase functions of admm.hpp:12-34
    "update_linear_cost", "backward_pass_grad", "forward_pass", "update_slack", "update_dual", "termination_condition",
    "project_soc", "project_hyperplane")
PHASES = {"update_linear_cost": 1, "backward_pass_grad": 2, "forward_pass": 3, "update_slack": 4, "update_dual": 5,
          "termination_condition": 6}


class TinyMPCError(RuntimeError):
    pass


def build(force: bool = False, report=None) -> str:
    """Compile libtinympc_amd.so for gfx950 with hipcc (cross-compiles without a GPU).  make decides by mtime which
    translation units are stale; what it did is reported -- every object either REBUILT by this call or REUSED (up to date) --
    on `report` (a callable taking one line; default: stderr) and kept in csrc/_gen/build_report.json, so that a build check can
    tell a fresh compile from a shipped library.  force (or TINYMPC_AMD_BUILD_FORCE=1): `make clean` first."""
    import glob
    import json
    import time
    say = report or (lambda line: print(line, file=sys.stderr, flush=True))
    force = force or bool(os.environ.get("TINYMPC_AMD_BUILD_FORCE"))
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    watched = lambda: {p: os.path.getmtime(p) for p in glob.glob(os.path.join(CSRC, "_gen", "*.o")) + ([LIB_PATH] if os.path.exists(LIB_PATH) else [])}
    before, t0 = watched(), time.time()
    subprocess.check_call(["make", "-C", CSRC, "-j", str(min(16, os.cpu_count() or 1))], stdout=subprocess.DEVNULL)
    after = watched()
    rebuilt = sorted(os.path.basename(p) for p, m in after.items() if before.get(p) != m)
    reused = sorted(os.path.basename(p) for p, m in after.items() if before.get(p) == m)
    mode = "forced full rebuild" if force else ("rebuilt" if len(reused) == 0 else ("incremental" if rebuilt else "reused (every object up to date)"))
    say("tinympc_amd.build: %s -- %d object(s) compiled now%s, %d reused, %.1f s" %
        (mode, len(rebuilt), (" (" + ", ".join(rebuilt[:8]) + (", ..." if len(rebuilt) > 8 else "") + ")") if rebuilt else "", len(reused), time.time() - t0))
    pre = prebuild_jit(say)
    try:
        with open(os.path.join(CSRC, "_gen", "build_report.json"), "w") as f:
            json.dump({"build_mode": mode, "rebuilt": rebuilt, "reused": reused, "seconds": time.time() - t0, "when": time.time(), "jit_prebuilt": pre}, f, indent=1)
    except OSError:
        pass
    return LIB_PATH


def prebuild_jit(say=None, jobs=None):
    """The prebuilt store of run-time instantiated kernels (csrc/jit_prebuilt.txt -> tinympc_amd/jit_prebuilt/*.co): every listed name
    that is not there yet is compiled by a process of its own (hipRTC of the toolchain this build runs on; needs no GPU), files the list
    no longer names are removed.  Returns {"compiled": n, "reused": m, "failed": [...]}."""
    import concurrent.futures
    say = say or (lambda line: print(line, file=sys.stderr, flush=True))
    listing = os.path.join(CSRC, "jit_prebuilt.txt")
    store = os.path.join(_HERE, "jit_prebuilt")
    if not os.path.exists(listing):
        return {"compiled": 0, "reused": 0, "failed": []}
    names = [ln.strip() for ln in open(listing) if ln.strip() and not ln.startswith("#")]
    os.makedirs(store, exist_ok=True)
    code = ("import sys, json; sys.path.insert(0, %r); import tinympc_amd as tm\n"
            "try:\n    print('@@' + json.dumps(tm.jit_prebuild(sys.argv[1])))\nexcept Exception as e:\n    print('@@' + json.dumps([-1, repr(e)]))" % os.path.dirname(_HERE))
    env = dict(os.environ)
    env.pop("TINYMPC_AMD_JIT_PREBUILT", None)

    def one(name):
        p = subprocess.run([sys.executable, "-c", code, name], capture_output=True, text=True, env=env)
        for ln in p.stdout.splitlines():
            if ln.startswith("@@"):
                import json
                return name, json.loads(ln[2:])
        return name, [-1, (p.stderr or "no output")[-300:]]
    t0 = __import__("time").time()
    keep, compiled, reused, failed = set(), 0, 0, []
    with concurrent.futures.ThreadPoolExecutor(jobs or min(8, os.cpu_count() or 1)) as ex:
        for name, (n, msg) in ex.map(one, names):
            if n < 0:
                failed.append((name, msg))
            else:
                keep.add(os.path.basename(msg))
                compiled += 1 if n > 0 else 0
                reused += 1 if n == 0 else 0
    if not failed:
        for f in os.listdir(store):
            if f.endswith(".co") and f not in keep:
                os.remove(os.path.join(store, f))
    say("tinympc_amd.build: run-time instantiated kernels prebuilt into tinympc_amd/jit_prebuilt/: %d compiled now, %d reused, %d failed, %.1f s"
        % (compiled, reused, len(failed), __import__("time").time() - t0))
    for name, msg in failed:
        say("  failed: %s: %s" % (name, msg))
    return {"compiled": compiled, "reused": reused, "failed": [n for n, _ in failed]}


_lib = None


def lib():
    """The loaded C ABI.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or `make -C tinympc_amd/csrc`). tinympc_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.tiny_batch_setup.argtypes = [C.POINTER(C.c_void_p), _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int]
        L.tiny_batch_setup_hetero.argtypes = [C.POINTER(C.c_void_p), _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int]
        L.tiny_batch_get_cache_instance.argtypes = [C.c_void_p, C.c_int, C.c_char_p, _dp, C.c_int]
        L.tiny_batch_destroy.argtypes = [C.c_void_p]
        L.tiny_batch_set_bound_constraints.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.tiny_batch_set_cone_constraints.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp]
        L.tiny_batch_set_linear_constraints.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.tiny_batch_set_tv_linear_constraints.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.tiny_batch_update_settings.argtypes = [C.c_void_p, C.c_double, C.c_double] + [C.c_int] * 10
        L.tiny_batch_get_cache.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
        L.tiny_batch_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.tiny_batch_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.tiny_batch_reset.argtypes = [C.c_void_p]
        L.tiny_batch_solve.argtypes = [C.c_void_p]
        L.tiny_batch_solve_async.argtypes = [C.c_void_p]
        L.tiny_batch_synchronize.argtypes = [C.c_void_p]
        L.tiny_batch_get_status.argtypes = [C.c_void_p, _ip, _ip, _ip, _dp]
        L.tiny_batch_reduce_stats.argtypes = [C.c_void_p, _dp, C.c_void_p]
        L.tiny_batch_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.tiny_batch_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.tiny_batch_get_timing.argtypes = [C.c_void_p, _fp, C.c_int]
        L.tiny_batch_set_reference_trajectory.argtypes = [C.c_void_p, _dp, C.c_int, _ip, C.c_int]
        L.tiny_batch_get_step_log.argtypes = [C.c_void_p, _ip, _dp, C.c_int]
        L.tiny_batch_last_error.argtypes = [C.c_void_p]
        L.tiny_batch_last_error.restype = C.c_char_p
        L.tiny_batch_supported_dims.argtypes = [_ip, C.c_int]
        L.tiny_batch_kernel_path.argtypes = [C.c_void_p]
        L.tiny_batch_algorithmic_bytes.argtypes = [C.c_void_p, C.c_int]
        L.tiny_batch_algorithmic_bytes.restype = C.c_long
        L.tiny_jit_compile.argtypes = [C.c_char_p, _ip, C.c_char_p, C.c_int]
        L.tiny_jit_compile.restype = C.c_long
        L.tiny_jit_prebuild.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.tiny_jit_prebuild.restype = C.c_long
        L.tiny_jit_used.argtypes = [C.c_char_p, C.c_int]
        L.tiny_batch_stats_message.argtypes = [C.c_void_p, C.c_void_p]
        L.tiny_batch_set_cache.argtypes = [C.c_void_p, C.c_char_p, _dp]
        L.tiny_predict_split.argtypes = [C.POINTER(C.c_uint), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.tiny_step_regroup_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_int]
        L.tiny_batch_get_plan.argtypes = [C.c_void_p, C.c_void_p]
        L.tiny_batch_set_plan.argtypes = [C.c_void_p, C.c_void_p]
        L.tiny_batch_get_option.argtypes = [C.c_void_p, C.c_char_p]
        L.tiny_batch_get_option.restype = C.c_long
        L.tiny_batch_set_adaptive_rho.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
        L.tiny_batch_set_sensitivity.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.tiny_batch_set_cache_state.argtypes = [C.c_void_p, C.c_char_p, _dp]
        L.tiny_batch_get_cache_state.argtypes = [C.c_void_p, C.c_char_p, _dp]
        L.tiny_rccl_unique_id.argtypes = [C.c_void_p]
        L.tiny_rccl_comm_init_rank.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.tiny_rccl_comm_destroy.argtypes = [C.c_void_p]
        L.tiny_rccl_comm_count.argtypes = [C.c_void_p]
        L.tiny_reduce_stats_messages.argtypes = [_dp, C.c_int, C.c_long, _dp]
        L.tiny_batch_allreduce_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, _dp]
        L.tiny_group_setup.argtypes = [C.POINTER(C.c_void_p), _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_int,
                                       C.c_int, _ip, C.c_int, C.c_int, C.c_int]
        for name in ("tiny_group_destroy", "tiny_group_shards", "tiny_group_uses_rccl", "tiny_group_reset", "tiny_group_solve",
                     "tiny_group_solve_async", "tiny_group_synchronize"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.tiny_group_shard.argtypes = [C.c_void_p, C.c_int]
        L.tiny_group_shard.restype = C.c_void_p
        L.tiny_group_shard_indices.argtypes = [C.c_void_p, C.c_int, _ip, C.c_int]
        L.tiny_group_last_error.argtypes = [C.c_void_p]
        L.tiny_group_last_error.restype = C.c_char_p
        L.tiny_group_set_bound_constraints.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.tiny_group_set_cone_constraints.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp]
        L.tiny_group_set_linear_constraints.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.tiny_group_set_tv_linear_constraints.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.tiny_group_update_settings.argtypes = [C.c_void_p, C.c_double, C.c_double] + [C.c_int] * 10
        L.tiny_group_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.tiny_group_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.tiny_group_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.tiny_group_get_status.argtypes = [C.c_void_p, _ip, _ip, _ip, _dp]
        L.tiny_group_allreduce_stats.argtypes = [C.c_void_p, _dp]
        _lib = L
    return _lib


def device_count() -> int:
    return int(lib().tiny_batch_device_count())


def supported_dims():
    buf = (C.c_int * (3 * 256))()
    n = lib().tiny_batch_supported_dims(buf, 256)
    return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n)]


def jit_compile(instantiation: str):
    """Compile one run-time instantiated kernel by its C++ name without loading it (needs no GPU); with the environment
    variable TINYMPC_AMD_JIT_CACHE=<directory> the code object is looked up / kept there.  Returns (bytes, from_disk)."""
    msg = C.create_string_buffer(2048)
    hit = C.c_int(0)
    n = lib().tiny_jit_compile(instantiation.encode(), C.byref(hit), msg, len(msg))
    if n <= 0:
        raise RuntimeError(msg.value.decode(errors="replace") or f"tiny_jit_compile failed ({n})")
    return int(n), bool(hit.value)


def jit_prebuild(instantiation: str, directory=None):
    """Build time: compile one run-time instantiated kernel with THIS process's hipRTC and keep it in the prebuilt store (default: next
    to the library, tinympc_amd/jit_prebuilt/).  Returns (code-object size -- 0 if it was already there --, its file)."""
    msg = C.create_string_buffer(2048)
    n = lib().tiny_jit_prebuild(instantiation.encode(), directory.encode() if directory else None, msg, len(msg))
    if n < 0:
        raise RuntimeError(msg.value.decode(errors="replace") or f"tiny_jit_prebuild failed ({n})")
    return int(n), msg.value.decode(errors="replace")          # (size -- 0: it was there already --, file)


def jit_used():
    """C++ names of the kernels this process instantiated at run time so far (what to feed jit_compile elsewhere)."""
    n = lib().tiny_jit_used(None, 0)
    buf = C.create_string_buffer(512 * max(n, 1))
    lib().tiny_jit_used(buf, len(buf))
    return [l for l in buf.value.decode().split("\n") if l]


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _colmajor(a, shape):
    """numpy (rows, cols) -> flat column-major doubles."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    a = np.broadcast_to(a, shape)
    return np.ascontiguousarray(a.T).ravel()


class TinyBatchSolver:
    """``batch`` independent MPC QPs sharing one (A, B, f, Q, R, rho): device-resident state, one
    HIP launch per ``solve()``.  Method names / argument meaning follow ``tiny_api.hpp``."""

    def __init__(self, A, B, f, Q, R, rho, nx, nu, N, batch, device=0, verbose=0):
        self._h = C.c_void_p()
        self.nx, self.nu, self.N, self.batch = int(nx), int(nu), int(N), int(batch)
        A = _colmajor(A, (nx, nx))
        Bm = _colmajor(B, (nx, nu))
        fv = _f64(np.zeros(nx) if f is None else f).ravel()
        Q = _f64(Q)
        R = _f64(R)
        Qd = (np.diag(Q) if Q.ndim == 2 else Q).copy()      # callers pass Q.asDiagonal() (dense) or the diagonal
        Rd = (np.diag(R) if R.ndim == 2 else R).copy()
        rc = lib().tiny_batch_setup(C.byref(self._h), A.ctypes.data_as(_dp), Bm.ctypes.data_as(_dp),
                                    fv.ctypes.data_as(_dp), Qd.ctypes.data_as(_dp), Rd.ctypes.data_as(_dp),
                                    float(rho), nx, nu, N, batch, device, verbose)
        if rc != OK:
            self._h = C.c_void_p()
            msg = {ERR_NO_DEVICE: "no MI355X / HIP device available (there is no CPU fallback)",
                   ERR_UNSUPPORTED: f"(nx,nu,N)=({nx},{nu},{N}) has no compiled kernel; have {supported_dims()}",
                   ERR_DIM: "bad dimensions"}.get(rc, f"error {rc}")
            raise TinyMPCError(f"tiny_batch_setup failed: {msg}")

    @classmethod
    def hetero(cls, A, B, f, Q, R, rho, N, device=0):
        """Heterogeneous batch: A [batch, nx, nx], B [batch, nx, nu], f [batch, nx] or None, Q [batch, nx], R [batch, nu]
        (user diagonals), rho [batch].  The Riccati recursion runs on the GPU for every instance."""
        A = np.asarray(A, dtype=np.float64)
        Bm = np.asarray(B, dtype=np.float64)
        batch, nx, nu = A.shape[0], A.shape[1], Bm.shape[2]
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.nx, self.nu, self.N, self.batch = nx, nu, int(N), batch
        Af = np.ascontiguousarray(A.transpose(0, 2, 1)).ravel()          # column-major per instance
        Bf = np.ascontiguousarray(Bm.transpose(0, 2, 1)).ravel()
        ff = None if f is None else _f64(f).ravel()
        Qf, Rf, rf = _f64(Q).ravel(), _f64(R).ravel(), _f64(np.broadcast_to(rho, (batch,))).ravel()
        rc = lib().tiny_batch_setup_hetero(C.byref(self._h), Af.ctypes.data_as(_dp), Bf.ctypes.data_as(_dp),
                                           None if ff is None else ff.ctypes.data_as(_dp), Qf.ctypes.data_as(_dp),
                                           Rf.ctypes.data_as(_dp), rf.ctypes.data_as(_dp), nx, nu, int(N), batch, device, 0)
        if rc != OK:
            self._h = C.c_void_p()
            raise TinyMPCError(f"tiny_batch_setup_hetero failed ({rc})")
        return self

    def cache_instance(self, instance, name):
        shapes = {"Kinf": (self.nu, self.nx), "Pinf": (self.nx, self.nx), "Quu_inv": (self.nu, self.nu),
                  "AmBKt": (self.nx, self.nx), "APf": (self.nx, 1), "BPf": (self.nu, 1), "Q": (self.nx, 1),
                  "R": (self.nu, 1), "riccati_iters": (1, 1)}
        r, c = shapes[name]
        out = np.zeros(r * c)
        n = lib().tiny_batch_get_cache_instance(self._h, int(instance), name.encode(), out.ctypes.data_as(_dp), r * c)
        assert n == r * c, (n, name)
        return out.reshape((r, c), order="F")

    @classmethod
    def from_problem(cls, prob, batch, device=0):
        return cls(prob["A"], prob["B"], prob.get("f"), prob["Q"], prob["R"], prob["rho"], prob["nx"], prob["nu"],
                   prob["N"], batch, device)

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None):
            lib().tiny_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc not in (OK,):
            raise TinyMPCError(f"{what} failed ({rc}): {lib().tiny_batch_last_error(self._h).decode()}")

    # ---- problem family (tiny_api.hpp:13-18, 36-43)
    def set_bound_constraints(self, x_min, x_max, u_min, u_max):
        nx, nu, N = self.nx, self.nu, self.N
        a = [_colmajor(x_min, (nx, N)), _colmajor(x_max, (nx, N)), _colmajor(u_min, (nu, N - 1)),
             _colmajor(u_max, (nu, N - 1))]
        self._check(lib().tiny_batch_set_bound_constraints(self._h, *[v.ctypes.data_as(_dp) for v in a]),
                    "set_bound_constraints")

    def set_cone_constraints(self, Acx, qcx, cx, Acu, qcu, cu):
        """STATE triple first (the positional order of the reference's definition, tiny_api.cpp:176-178)."""
        ia = [np.ascontiguousarray(np.asarray(v, dtype=np.int32).ravel()) for v in (Acx, qcx, Acu, qcu)]
        da = [_f64(v).ravel() for v in (cx, cu)]
        self._check(lib().tiny_batch_set_cone_constraints(
            self._h, len(ia[0]), ia[0].ctypes.data_as(_ip), ia[1].ctypes.data_as(_ip), da[0].ctypes.data_as(_dp),
            len(ia[2]), ia[2].ctypes.data_as(_ip), ia[3].ctypes.data_as(_ip), da[1].ctypes.data_as(_dp)),
            "set_cone_constraints")

    def set_linear_constraints(self, Alin_x, blin_x, Alin_u, blin_u):
        """Half-spaces a_k' z <= b_k (tiny_set_linear_constraints): Alin_x (n_s, nx), blin_x (n_s,), Alin_u (n_i, nu)."""
        Ax = np.asarray(Alin_x, dtype=np.float64).reshape(-1, self.nx)
        Au = np.asarray(Alin_u, dtype=np.float64).reshape(-1, self.nu)
        a = [np.ascontiguousarray(Ax.T).ravel(), _f64(blin_x).ravel(), np.ascontiguousarray(Au.T).ravel(), _f64(blin_u).ravel()]
        self._check(lib().tiny_batch_set_linear_constraints(self._h, Ax.shape[0], a[0].ctypes.data_as(_dp), a[1].ctypes.data_as(_dp),
                                                            Au.shape[0], a[2].ctypes.data_as(_dp), a[3].ctypes.data_as(_dp)),
                    "set_linear_constraints")

    def set_tv_linear_constraints(self, tv_Alin_x, tv_blin_x, tv_Alin_u, tv_blin_u):
        """tiny_set_tv_linear_constraints: tv_Alin_x (n_s*N, nx) [row n_s*i+k = constraint k at knot i], tv_blin_x (n_s, N),
        tv_Alin_u (n_i*(N-1), nu), tv_blin_u (n_i, N-1)."""
        Ax = np.asarray(tv_Alin_x, dtype=np.float64).reshape(-1, self.nx)
        Au = np.asarray(tv_Alin_u, dtype=np.float64).reshape(-1, self.nu)
        bx = np.asarray(tv_blin_x, dtype=np.float64).reshape(-1, self.N)
        bu = np.asarray(tv_blin_u, dtype=np.float64).reshape(-1, self.N - 1)
        a = [np.ascontiguousarray(Ax.T).ravel(), np.ascontiguousarray(bx.T).ravel(), np.ascontiguousarray(Au.T).ravel(),
             np.ascontiguousarray(bu.T).ravel()]
        self._check(lib().tiny_batch_set_tv_linear_constraints(self._h, bx.shape[0], a[0].ctypes.data_as(_dp), a[1].ctypes.data_as(_dp),
                                                               bu.shape[0], a[2].ctypes.data_as(_dp), a[3].ctypes.data_as(_dp)),
                    "set_tv_linear_constraints")

    def update_settings(self, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=1000, check_termination=1,
                        en_state_bound=1, en_input_bound=1, en_state_soc=0, en_input_soc=0, en_state_linear=0,
                        en_input_linear=0, en_tv_state_linear=0, en_tv_input_linear=0):
        self._check(lib().tiny_batch_update_settings(
            self._h, abs_pri_tol, abs_dua_tol, int(max_iter), int(check_termination), int(en_state_bound),
            int(en_input_bound), int(en_state_soc), int(en_input_soc), int(en_state_linear), int(en_input_linear),
            int(en_tv_state_linear), int(en_tv_input_linear)), "update_settings")

    def cache(self, name):
        shapes = {"Kinf": (self.nu, self.nx), "Pinf": (self.nx, self.nx), "Quu_inv": (self.nu, self.nu),
                  "AmBKt": (self.nx, self.nx), "APf": (self.nx, 1), "BPf": (self.nu, 1), "Q": (self.nx, 1),
                  "R": (self.nu, 1)}
        r, c = shapes[name]
        out = np.zeros(r * c)
        n = lib().tiny_batch_get_cache(self._h, name.encode(), out.ctypes.data_as(_dp), r * c)
        assert n == r * c
        return out.reshape((r, c), order="F")

    # ---- per-instance data
    def _shape(self, name):
        if name == "x0":
            return (self.nx, 1)
        return (self.nx, self.N) if name in STATE_FIELDS else (self.nu, self.N - 1)

    def set(self, name, value, broadcast=False):
        """value: [batch, rows, cols] (or [rows, cols] with broadcast=True); x0: [batch, nx]."""
        r, c = self._shape(name)
        a = np.asarray(value, dtype=np.float64)
        if broadcast:
            flat = np.ascontiguousarray(a.reshape(r, c).T).ravel()
        else:
            flat = np.ascontiguousarray(a.reshape(self.batch, r, c).transpose(0, 2, 1)).ravel()
        self._check(lib().tiny_batch_set(self._h, FIELD_ID[name], flat.ctypes.data_as(C.c_void_p),
                                         HOST | (BROADCAST if broadcast else 0)), f"set({name})")

    def get(self, name):
        r, c = self._shape(name)
        out = np.zeros((self.batch, c, r))
        self._check(lib().tiny_batch_get(self._h, FIELD_ID[name], out.ctypes.data_as(C.c_void_p), HOST), f"get({name})")
        out = out.transpose(0, 2, 1)
        return out[:, :, 0] if name == "x0" else out

    def set_device(self, name, ptr, broadcast=False):
        """ptr: device pointer (int) to [batch][cols][rows] doubles already resident in HBM."""
        self._check(lib().tiny_batch_set(self._h, FIELD_ID[name], C.c_void_p(ptr),
                                         DEVICE | (BROADCAST if broadcast else 0)), f"set_device({name})")

    def set_x0(self, x0, broadcast=False):          # tiny_set_x0
        self.set("x0", x0, broadcast)

    def set_x_ref(self, x_ref, broadcast=False):    # tiny_set_x_ref
        self.set("Xref", x_ref, broadcast)

    def set_u_ref(self, u_ref, broadcast=False):    # tiny_set_u_ref
        self.set("Uref", u_ref, broadcast)

    def reset(self):
        self._check(lib().tiny_batch_reset(self._h), "reset")

    # ---- hot path
    def solve(self) -> int:
        rc = lib().tiny_batch_solve(self._h)
        if rc not in (0, 1):
            self._check(rc, "solve")
        return rc

    def solve_async(self):
        self._check(lib().tiny_batch_solve_async(self._h), "solve_async")

    def phase(self, name):
        """ONE phase of the iteration over the batch (the reference's exported phase functions, admm.hpp:12-17), on
        the device records as they are.  'termination_condition' returns the per-instance booleans."""
        self._check(lib().tiny_batch_phase(self._h, PHASES[name]), name)
        if name == "termination_condition":
            return self.status()["solved"].astype(bool)

    def synchronize(self):
        self._check(lib().tiny_batch_synchronize(self._h), "synchronize")

    def status(self):
        B = self.batch
        it, so, st = (np.zeros(B, dtype=np.int32) for _ in range(3))
        res = np.zeros((B, 4))
        self._check(lib().tiny_batch_get_status(self._h, it.ctypes.data_as(_ip), so.ctypes.data_as(_ip),
                                                st.ctypes.data_as(_ip), res.ctypes.data_as(_dp)), "get_status")
        return dict(iter=it, solved=so, status=st, primal_residual_state=res[:, 0], primal_residual_input=res[:, 1],
                    dual_residual_state=res[:, 2], dual_residual_input=res[:, 3])

    def reduce_stats(self, device_out=None):
        """[sum_iter, sum_solved, batch, max residual x4, accumulated iters, accumulated solved, 0];
        device_out: optional device pointer (int) receiving the same 10 doubles (e.g. the buffer of an
        RCCL all-reduce)."""
        out = np.zeros(10)
        self._check(lib().tiny_batch_reduce_stats(self._h, out.ctypes.data_as(_dp),
                                                  C.c_void_p(device_out) if device_out else None), "reduce_stats")
        return out

    def reduce_stats_async(self, device_out):
        self._check(lib().tiny_batch_reduce_stats(self._h, None, C.c_void_p(device_out)), "reduce_stats")

    # ---- adaptive rho (types.hpp:75-79; admm.cpp:397-423; rho_benchmark.cpp)
    def set_adaptive_rho(self, enable=1, rho_min=1.0, rho_max=100.0, clip=1):
        self._check(lib().tiny_batch_set_adaptive_rho(self._h, int(enable), float(rho_min), float(rho_max), int(clip)), "set_adaptive_rho")

    def set_sensitivity(self, dKinf, dPinf, dC1=None, dC2=None):
        """(rows, cols) arrays: dKinf_drho (nu, nx), dPinf_drho (nx, nx), dC1_drho (nu, nu), dC2_drho (nx, nx)"""
        nx, nu = self.nx, self.nu
        a = [_colmajor(dKinf, (nu, nx)), _colmajor(dPinf, (nx, nx)), None if dC1 is None else _colmajor(dC1, (nu, nu)),
             None if dC2 is None else _colmajor(dC2, (nx, nx))]
        self._check(lib().tiny_batch_set_sensitivity(self._h, *[None if v is None else v.ctypes.data_as(_dp) for v in a]), "set_sensitivity")

    _CACHE_SHAPES = {"rho": lambda s: (1, 1), "Kinf": lambda s: (s.nu, s.nx), "Pinf": lambda s: (s.nx, s.nx),
                     "C1": lambda s: (s.nu, s.nu), "C2": lambda s: (s.nx, s.nx)}

    def set_cache_state(self, which, value):
        """per-instance cache state of an adaptive batch: value [batch] for 'rho', [batch, rows, cols] otherwise"""
        r, c = self._CACHE_SHAPES[which](self)
        a = np.asarray(value, dtype=np.float64).reshape(self.batch, r, c)
        flat = np.ascontiguousarray(a.transpose(0, 2, 1)).ravel()
        self._check(lib().tiny_batch_set_cache_state(self._h, which.encode(), flat.ctypes.data_as(_dp)), f"set_cache_state({which})")

    def get_cache_state(self, which):
        r, c = self._CACHE_SHAPES[which](self)
        out = np.zeros((self.batch, c, r))
        self._check(lib().tiny_batch_get_cache_state(self._h, which.encode(), out.ctypes.data_as(_dp)), f"get_cache_state({which})")
        out = out.transpose(0, 2, 1)
        return out[:, 0, 0] if which == "rho" else out

    def allreduce_stats(self, comm, n_ranks, rank, total_batch):
        """the path's one exchange on an RCCL communicator the caller owns (tiny_batch_allreduce_stats); returns the
        job-wide 10-entry statistics vector (the same on every rank)"""
        out = np.zeros(10)
        self._check(lib().tiny_batch_allreduce_stats(self._h, C.c_void_p(comm), int(n_ranks), int(rank), int(total_batch),
                                                     out.ctypes.data_as(_dp)), "allreduce_stats")
        return out

    def get_option(self, name) -> int:
        return int(lib().tiny_batch_get_option(self._h, name.encode()))

    def get_plan(self) -> bytes:
        """the settled launch form of this batch (TinyBatchPlan, include/tinympc_amd.h) as bytes: write them to a file, hand them
        to set_plan of another handle of the same (nx, nu, N) -- it takes the settled form on its first solve"""
        buf = C.create_string_buffer(PLAN_BYTES)
        self._check(lib().tiny_batch_get_plan(self._h, buf), "get_plan")
        return buf.raw

    def set_plan(self, plan: bytes):
        if len(plan) != PLAN_BYTES:
            raise TinyMPCError("set_plan: %d bytes, a TinyBatchPlan has %d" % (len(plan), PLAN_BYTES))
        self._check(lib().tiny_batch_set_plan(self._h, C.create_string_buffer(bytes(plan), PLAN_BYTES)), "set_plan")

    @staticmethod
    def plan_fields(plan: bytes) -> dict:
        """the scalar fields of a TinyBatchPlan (diagnostics, tests)"""
        import struct
        names = ("magic version bytes nx nu N batch max_iter check_termination open_questions auto_verdict auto_cap auto_cap_max_iter "
                 "auto_growth growth_verdict auto_probes tile_verdict regroup_verdict hist_valid").split()
        ints = struct.unpack_from("<%di" % len(names), plan, 0)
        off = (4 * len(names) + 7) // 8 * 8
        dbl = struct.unpack_from("<5d", plan, off)
        d = dict(zip(names, ints))
        d.update(zip(("auto_plain_rate", "auto_split_rate", "auto_gain", "tile_rate", "lockstep_ratio"), dbl))
        return d

    def stats_message_async(self, device_out):
        """the batch's 64-byte statistics message (8 doubles) -> device memory, on the batch's stream behind the solve"""
        self._check(lib().tiny_batch_stats_message(self._h, C.c_void_p(device_out)), "stats_message")

    def set_option(self, name, value):
        self._check(lib().tiny_batch_set_option(self._h, name.encode(), int(value)), f"set_option({name})")

    def set_stream(self, stream_ptr):
        self._check(lib().tiny_batch_set_stream(self._h, C.c_void_p(stream_ptr)), "set_stream")

    def timing_ms(self, capacity=4096):
        buf = np.zeros(capacity, dtype=np.float32)
        n = lib().tiny_batch_get_timing(self._h, buf.ctypes.data_as(_fp), capacity)
        return buf[:max(n, 0)].astype(np.float64)

    def set_reference_trajectory(self, xref_points, offsets=None):
        """xref_points: [n_points, nx] shared state-reference trajectory; the solve at MPC step k tracks the window
        k + offsets[b] ... + N - 1 (examples/quadrotor_tracking.cpp).  None removes it."""
        if xref_points is None:
            self._check(lib().tiny_batch_set_reference_trajectory(self._h, None, 0, None, HOST), "set_reference_trajectory")
            return
        t = _f64(xref_points).reshape(-1, self.nx)
        off = None if offsets is None else np.ascontiguousarray(np.asarray(offsets, dtype=np.int32).ravel())
        self._check(lib().tiny_batch_set_reference_trajectory(
            self._h, t.ctypes.data_as(_dp), t.shape[0], None if off is None else off.ctypes.data_as(_ip), HOST),
            "set_reference_trajectory")

    def step_log(self, steps):
        """(iters[steps, batch] (negative: hit max_iter), u0[steps, batch, nu]) of the last fused launch."""
        it = np.zeros((steps, self.batch), dtype=np.int32)
        u0 = np.zeros((steps, self.batch, self.nu))
        self._check(lib().tiny_batch_get_step_log(self._h, it.ctypes.data_as(_ip), u0.ctypes.data_as(_dp), steps),
                    "get_step_log")
        return it, u0

    def kernel_path(self) -> str:
        return {0: "regs", 1: "tile", 2: "cover", 3: "jit", 4: "tile-jit"}[lib().tiny_batch_kernel_path(self._h)]

    def algorithmic_bytes(self, cold=False) -> int:
        """cold: False / 0 bytes_warm, True / 1 bytes_cold (one_shot = 2), 2 the traffic of one_shot = 1."""
        return int(lib().tiny_batch_algorithmic_bytes(self._h, int(cold)))


class TinyGroupSolver:
    """ctypes mirror of TinyGroup (include/tinympc_amd.h section C): ONE host process, the batch sharded over several
    GPUs (contiguous blocks or round-robin), one RCCL all-gather of 64-byte statistics messages as the only exchange.
    Per-instance arrays carry the FULL batch axis in the caller's order."""

    def __init__(self, A, B, f, Q, R, rho, nx, nu, N, batch, devices=None, n_shards=0, interleaved=False, verbose=0):
        self._h = C.c_void_p()
        self.nx, self.nu, self.N, self.batch = int(nx), int(nu), int(N), int(batch)
        A = _colmajor(A, (nx, nx))
        Bm = _colmajor(B, (nx, nu))
        fv = _f64(np.zeros(nx) if f is None else f).ravel()
        Q, R = _f64(Q), _f64(R)
        Qd = (np.diag(Q) if Q.ndim == 2 else Q).copy()
        Rd = (np.diag(R) if R.ndim == 2 else R).copy()
        dev = None
        if devices is not None:
            dev = np.ascontiguousarray(devices, dtype=np.int32)
            n_shards = len(dev)
        rc = lib().tiny_group_setup(C.byref(self._h), A.ctypes.data_as(_dp), Bm.ctypes.data_as(_dp), fv.ctypes.data_as(_dp),
                                    Qd.ctypes.data_as(_dp), Rd.ctypes.data_as(_dp), float(rho), nx, nu, N, batch,
                                    None if dev is None else dev.ctypes.data_as(_ip), int(n_shards), int(bool(interleaved)), verbose)
        if rc != OK:
            self._h = C.c_void_p()
            raise TinyMPCError(f"tiny_group_setup failed ({rc}): {lib().tiny_group_last_error(None).decode()}")

    @classmethod
    def from_problem(cls, prob, batch, **kw):
        return cls(prob["A"], prob["B"], prob.get("f"), prob["Q"], prob["R"], prob["rho"], prob["nx"], prob["nu"], prob["N"], batch, **kw)

    def close(self):
        if getattr(self, "_h", None):
            lib().tiny_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != OK:
            raise TinyMPCError(f"{what} failed ({rc}): {lib().tiny_group_last_error(self._h).decode()}")

    @property
    def shards(self):
        return int(lib().tiny_group_shards(self._h))

    def uses_rccl(self):
        return bool(lib().tiny_group_uses_rccl(self._h))

    def shard_indices(self, k):
        n = lib().tiny_group_shard_indices(self._h, k, None, 0)
        idx = np.zeros(n, dtype=np.int32)
        lib().tiny_group_shard_indices(self._h, k, idx.ctypes.data_as(_ip), n)
        return idx

    def set_bound_constraints(self, x_min, x_max, u_min, u_max):
        nx, nu, N = self.nx, self.nu, self.N
        xm, xM = _colmajor(x_min, (nx, N)), _colmajor(x_max, (nx, N))
        um, uM = _colmajor(u_min, (nu, N - 1)), _colmajor(u_max, (nu, N - 1))
        self._check(lib().tiny_group_set_bound_constraints(self._h, xm.ctypes.data_as(_dp), xM.ctypes.data_as(_dp),
                                                           um.ctypes.data_as(_dp), uM.ctypes.data_as(_dp)), "set_bound_constraints")

    def set_cone_constraints(self, Acx, qcx, cx, Acu, qcu, cu):
        ai = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        Acx, qcx, Acu, qcu = ai(Acx), ai(qcx), ai(Acu), ai(qcu)
        cx, cu = _f64(cx).ravel(), _f64(cu).ravel()
        self._check(lib().tiny_group_set_cone_constraints(self._h, len(Acx), Acx.ctypes.data_as(_ip), qcx.ctypes.data_as(_ip),
                                                          cx.ctypes.data_as(_dp), len(Acu), Acu.ctypes.data_as(_ip),
                                                          qcu.ctypes.data_as(_ip), cu.ctypes.data_as(_dp)), "set_cone_constraints")

    def update_settings(self, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=1000, check_termination=1, en_state_bound=1,
                        en_input_bound=1, en_state_soc=0, en_input_soc=0, en_state_linear=0, en_input_linear=0,
                        en_tv_state_linear=0, en_tv_input_linear=0):
        self._check(lib().tiny_group_update_settings(self._h, abs_pri_tol, abs_dua_tol, max_iter, check_termination, en_state_bound,
                                                     en_input_bound, en_state_soc, en_input_soc, en_state_linear, en_input_linear,
                                                     en_tv_state_linear, en_tv_input_linear), "update_settings")

    def set_option(self, name, value):
        self._check(lib().tiny_group_set_option(self._h, name.encode(), int(value)), f"set_option({name})")

    def _shape(self, name):
        if name == "x0":
            return self.nx, 1
        return (self.nx, self.N) if name in STATE_FIELDS else (self.nu, self.N - 1)

    def set(self, name, value, broadcast=False):
        r, c = self._shape(name)
        a = np.asarray(value, dtype=np.float64)
        flat = (np.ascontiguousarray(a.reshape(r, c).T) if broadcast else
                np.ascontiguousarray(a.reshape(self.batch, r, c).transpose(0, 2, 1))).ravel()
        self._check(lib().tiny_group_set(self._h, FIELD_ID[name], flat.ctypes.data_as(C.c_void_p), HOST | (BROADCAST if broadcast else 0)), f"set({name})")

    def get(self, name):
        r, c = self._shape(name)
        out = np.zeros((self.batch, c, r))
        self._check(lib().tiny_group_get(self._h, FIELD_ID[name], out.ctypes.data_as(C.c_void_p)), f"get({name})")
        out = out.transpose(0, 2, 1)
        return out[:, :, 0] if name == "x0" else out

    def set_x0(self, x0, broadcast=False):
        self.set("x0", x0, broadcast)

    def set_x_ref(self, x_ref, broadcast=False):
        self.set("Xref", x_ref, broadcast)

    def set_u_ref(self, u_ref, broadcast=False):
        self.set("Uref", u_ref, broadcast)

    def reset(self):
        self._check(lib().tiny_group_reset(self._h), "reset")

    def solve(self) -> int:
        rc = lib().tiny_group_solve(self._h)
        if rc not in (0, 1):
            self._check(rc, "solve")
        return rc

    def solve_async(self):
        self._check(lib().tiny_group_solve_async(self._h), "solve_async")

    def synchronize(self):
        self._check(lib().tiny_group_synchronize(self._h), "synchronize")

    def allreduce_stats(self):
        out = np.zeros(10)
        self._check(lib().tiny_group_allreduce_stats(self._h, out.ctypes.data_as(_dp)), "allreduce_stats")
        return out

    def status(self):
        B = self.batch
        it, so, st = (np.zeros(B, dtype=np.int32) for _ in range(3))
        res = np.zeros((B, 4))
        self._check(lib().tiny_group_get_status(self._h, it.ctypes.data_as(_ip), so.ctypes.data_as(_ip), st.ctypes.data_as(_ip),
                                                res.ctypes.data_as(_dp)), "get_status")
        return dict(iter=it, solved=so, status=st, residuals=res)


def step_regroup_plan(steps, k=-1, known=False, half=0):
    """the stretches option "step_regroup" cuts a fused launch of `steps` MPC steps into (k <= 0: the automatic length).  Host
    arithmetic, no GPU."""
    out = np.zeros(max(int(steps), 1) + 1, dtype=np.int32)
    n = lib().tiny_step_regroup_plan(int(steps), int(k), int(bool(known)), int(half), out.ctypes.data_as(_ip), len(out))
    return [int(v) for v in out[:n]]


def predict_split(hist, nx, nu, N, max_iter, check_termination=1, num_cus=256):
    """(K, predicted time ratio) of the automatic split solve's cost model for an iteration-count histogram (hist[i] =
    instances needing i iterations); K = 0: a plain launch is predicted to be within 5 %.  Host arithmetic, no GPU."""
    h = np.zeros(1024, dtype=np.uint32)
    hist = np.asarray(hist)
    h[:min(len(hist), 1024)] = hist[:1024]
    r = C.c_double(1.0)
    k = lib().tiny_predict_split(h.ctypes.data_as(C.POINTER(C.c_uint)), nx, nu, N, max_iter, check_termination, num_cus, C.byref(r))
    return int(k), float(r.value)


def rccl_unique_id() -> bytes:
    """rank 0: a fresh 128-byte ncclUniqueId (tiny_rccl_unique_id)"""
    buf = C.create_string_buffer(128)
    rc = lib().tiny_rccl_unique_id(buf)
    if rc != OK:
        raise TinyMPCError(f"tiny_rccl_unique_id failed ({rc})")
    return buf.raw


def rccl_comm_init_rank(n_ranks, unique_id: bytes, rank, device) -> int:
    """join the communicator of `unique_id` as `rank` on GPU `device`; returns the ncclComm_t as an integer"""
    comm = C.c_void_p()
    rc = lib().tiny_rccl_comm_init_rank(C.byref(comm), int(n_ranks), C.c_char_p(unique_id), int(rank), int(device))
    if rc != OK:
        raise TinyMPCError(f"tiny_rccl_comm_init_rank failed ({rc})")
    return comm.value


def rccl_comm_destroy(comm):
    lib().tiny_rccl_comm_destroy(C.c_void_p(comm))


def rccl_comm_count(comm) -> int:
    """ranks of the communicator, asked of RCCL itself (ncclCommCount)"""
    n = lib().tiny_rccl_comm_count(C.c_void_p(comm))
    if n < 0:
        raise TinyMPCError(f"tiny_rccl_comm_count failed ({n})")
    return int(n)


def reduce_stats_messages(table, total_batch):
    """native host reduction of gathered 64-byte messages ([n_shards, 8]) -> the 10-entry statistics vector (no GPU needed)"""
    t = np.ascontiguousarray(np.asarray(table, dtype=np.float64).reshape(-1, 8))
    out = np.zeros(10)
    rc = lib().tiny_reduce_stats_messages(t.ctypes.data_as(_dp), t.shape[0], int(total_batch), out.ctypes.data_as(_dp))
    if rc != OK:
        raise TinyMPCError(f"tiny_reduce_stats_messages failed ({rc})")
    return out


def rccl_available() -> bool:
    """librccl loads next to this library's HIP runtime (nothing collective happens)"""
    return bool(lib().tiny_rccl_available())


def load_problem(name):
    """Problem families of the reference examples (numbers extracted by oracle/extract_problem_data.py)."""
    import json
    p = json.load(open(os.path.join(_HERE, "data", "problems.json")))[name]
    prob = dict(nx=p["nx"], nu=p["nu"], N=p["N"], rho=float(p["rho"]), A=np.array(p["A"], dtype=np.float64),
                B=np.array(p["B"], dtype=np.float64), f=np.array(p["f"], dtype=np.float64),
                Q=np.array(p["Q"], dtype=np.float64), R=np.array(p["R"], dtype=np.float64))
    extra = {k: v for k, v in p.items() if k not in prob and k != "source"}
    return prob, extra


def random_problem(nx, nu, N):
    """The synthetic problem family of BASELINE configs[4]: one shared (A, B, Q, R)
    per (nx, nu, N) cell, seeded; A is scaled to spectral radius 0.95."""
    rng = np.random.default_rng(1000 * nx + 10 * nu + N)
    M = rng.standard_normal((nx, nx))
    A = M * 0.95 / np.max(np.abs(np.linalg.eigvals(M)))
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = rng.uniform(1, 10, nx)
    R = rng.uniform(0.1, 1, nu)
    return dict(nx=nx, nu=nu, N=N, rho=1.0, A=A, B=B, f=np.zeros(nx), Q=Q, R=R), rng


def flops_per_iter(nx, nu, N):
    """FLOPs of one ADMM iteration with box constraints (footnote 1)."""
    S = nx * N + nu * (N - 1)
    return (4 * S + 2 * nx * nx + 3 * nx + (N - 1) * (2 * nx * nx + 4 * nx * nu + 2 * nu * nu + 2 * nu + 3 * nx)
            + (N - 1) * (2 * nx * nx + 4 * nx * nu + 2 * nx + 2 * nu) + 11 * S)
