"""TEST INFRASTRUCTURE -- ctypes drivers for the two CPU implementations.

* ``OracleSolver``  -> oracle/liboracle.so   (plain-C restatement, oracle/tinympc_oracle.c)
* ``RefSolver``     -> oracle/_ref/libtinympc_ref.so (the real TinyMPC reference behind
                      oracle/ref_shim.cpp; only exists after ``make -C oracle ref`` in a
                      container that has /root/reference)

Both expose the same surface (field names are the reference's, src/tinympc/types.hpp):
``s["vnew"]`` returns a live numpy view (column-major, shape (rows, cols)) of the
solver's own buffer; ``s.get("iter")`` / ``s.set("max_iter", 100)`` read/write scalars.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libtinympc_ref.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build_oracle(force: bool = False) -> str:
    """Compile the C restatement (gcc, <1 s)."""
    src = os.path.join(_HERE, "tinympc_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def build_ref() -> str | None:
    """Compile the real reference when /root/reference is present; else keep a prebuilt one."""
    if os.path.isdir("/root/reference/src/tinympc"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return REF_SO if os.path.exists(REF_SO) else None


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def _d(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))


def _i(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel())


class _CpuSolver:
    prefix = ""
    so = ""
    _libs: dict = {}

    @classmethod
    def lib(cls):
        if cls.so not in cls._libs:
            lib = C.CDLL(cls.so)
            p = cls.prefix
            f = getattr(lib, p + "setup")
            f.restype = C.c_void_p
            f.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_double]
            getattr(lib, p + "free").argtypes = [C.c_void_p]
            getattr(lib, p + "free").restype = None
            getattr(lib, p + "set_bounds").argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
            getattr(lib, p + "set_cones").argtypes = [C.c_void_p, C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp]
            getattr(lib, p + "set_linear").argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
            getattr(lib, p + "set_tv_linear").argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
            f = getattr(lib, p + "ptr")
            f.restype = _dp
            f.argtypes = [C.c_void_p, C.c_char_p, _ip, _ip]
            f = getattr(lib, p + "get")
            f.restype = C.c_double
            f.argtypes = [C.c_void_p, C.c_char_p]
            getattr(lib, p + "set").argtypes = [C.c_void_p, C.c_char_p, C.c_double]
            getattr(lib, p + "solve").argtypes = [C.c_void_p]
            getattr(lib, p + "phase").argtypes = [C.c_void_p, C.c_char_p]
            getattr(lib, p + "project_soc").argtypes = [_dp, C.c_int, C.c_double]
            getattr(lib, p + "project_soc").restype = None
            f = getattr(lib, p + "closed_loop")
            f.restype = C.c_long
            f.argtypes = [C.c_void_p, _dp, C.c_int, _ip, _dp]
            f = getattr(lib, p + "closed_loop_traj")
            f.restype = C.c_long
            f.argtypes = [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, _ip, _dp]
            cls._libs[cls.so] = lib
        return cls._libs[cls.so]

    # ---- adaptive rho (types.hpp:75-79, tiny_api.cpp:479-540)
    def set_adaptive_rho(self, enable=1, rho_min=1.0, rho_max=100.0, clip=1):
        for k, v in (("adaptive_rho", enable), ("adaptive_rho_min", rho_min), ("adaptive_rho_max", rho_max),
                     ("adaptive_rho_enable_clipping", clip)):
            self.set(k, v)

    def set_sensitivity(self, tables):
        """tables: dict name -> (rows, cols) array for dKinf_drho, dPinf_drho, dC1_drho, dC2_drho"""
        for k in ("dKinf_drho", "dPinf_drho", "dC1_drho", "dC2_drho"):
            self[k] = tables[k]

    CACHE_STATE = ("Kinf", "Pinf", "C1", "C2")

    def _f(self, name):
        return getattr(self.lib(), self.prefix + name)

    def __init__(self, nx, nu, N, A, B, f, Qdiag, Rdiag, rho):
        self.nx, self.nu, self.N = int(nx), int(nu), int(N)
        A, B, Q, R = _d(A), _d(B), _d(Qdiag), _d(Rdiag)
        fv = _d(np.zeros(nx) if f is None else f)
        self.h = self._f("setup")(nx, nu, N, A.ctypes.data_as(_dp), B.ctypes.data_as(_dp), fv.ctypes.data_as(_dp),
                                  Q.ctypes.data_as(_dp), R.ctypes.data_as(_dp), float(rho))
        if not self.h:
            raise RuntimeError(f"{self.prefix}setup failed")

    @classmethod
    def from_problem(cls, prob, N=None):
        """prob: dict with nx, nu, N, A (nx,nx), B (nx,nu), f (nx,), Q (nx,), R (nu,), rho."""
        return cls(prob["nx"], prob["nu"], N or prob["N"], prob["A"], prob["B"], prob.get("f"),
                   prob["Q"], prob["R"], prob["rho"])

    def close(self):
        if getattr(self, "h", None):
            self._f("free")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bounds(self, x_min, x_max, u_min, u_max):
        nx, nu, N = self.nx, self.nu, self.N
        arrs = [_d(np.broadcast_to(np.asarray(a, dtype=np.float64).reshape(-1, 1) if np.ndim(a) == 1 else a, s))
                for a, s in ((x_min, (nx, N)), (x_max, (nx, N)), (u_min, (nu, N - 1)), (u_max, (nu, N - 1)))]
        return self._f("set_bounds")(self.h, *[a.ctypes.data_as(_dp) for a in arrs])

    def set_cones(self, Acx, qcx, cx, Acu, qcu, cu):
        """State triple first (the positional order of the reference's definition)."""
        a = [_i(Acx), _i(qcx), _d(cx), _i(Acu), _i(qcu), _d(cu)]
        return self._f("set_cones")(self.h, len(a[0]), a[0].ctypes.data_as(_ip), a[1].ctypes.data_as(_ip),
                                    a[2].ctypes.data_as(_dp), len(a[3]), a[3].ctypes.data_as(_ip),
                                    a[4].ctypes.data_as(_ip), a[5].ctypes.data_as(_dp))

    def set_linear(self, Alin_x, blin_x, Alin_u, blin_u):
        """Static half-spaces a_k' z <= b_k: Alin_x (n_s, nx), blin_x (n_s,), Alin_u (n_i, nu), blin_u (n_i,)."""
        Ax, Au = np.asarray(Alin_x, dtype=np.float64).reshape(-1, self.nx), np.asarray(Alin_u, dtype=np.float64).reshape(-1, self.nu)
        a = [_d(Ax), _d(blin_x), _d(Au), _d(blin_u)]
        return self._f("set_linear")(self.h, Ax.shape[0], a[0].ctypes.data_as(_dp), a[1].ctypes.data_as(_dp),
                                     Au.shape[0], a[2].ctypes.data_as(_dp), a[3].ctypes.data_as(_dp))

    def set_tv_linear(self, tv_Alin_x, tv_blin_x, tv_Alin_u, tv_blin_u):
        """Time-varying half-spaces: tv_Alin_x (n_s*N, nx) [row n_s*i+k = constraint k at knot i], tv_blin_x (n_s, N),
        tv_Alin_u (n_i*(N-1), nu), tv_blin_u (n_i, N-1)."""
        Ax, Au = np.asarray(tv_Alin_x, dtype=np.float64).reshape(-1, self.nx), np.asarray(tv_Alin_u, dtype=np.float64).reshape(-1, self.nu)
        bx, bu = np.asarray(tv_blin_x, dtype=np.float64).reshape(-1, self.N), np.asarray(tv_blin_u, dtype=np.float64).reshape(-1, self.N - 1)
        a = [_d(Ax), _d(bx), _d(Au), _d(bu)]
        return self._f("set_tv_linear")(self.h, bx.shape[0], a[0].ctypes.data_as(_dp), a[1].ctypes.data_as(_dp),
                                        bu.shape[0], a[2].ctypes.data_as(_dp), a[3].ctypes.data_as(_dp))

    def __getitem__(self, name) -> np.ndarray:
        r, c = C.c_int(), C.c_int()
        p = self._f("ptr")(self.h, name.encode(), C.byref(r), C.byref(c))
        if not p:
            raise KeyError(name)
        flat = np.ctypeslib.as_array(p, shape=(r.value * c.value,))
        return flat.reshape((r.value, c.value), order="F")

    def __setitem__(self, name, value):
        self[name][...] = np.asarray(value, dtype=np.float64).reshape(self[name].shape)

    def get(self, name):
        return self._f("get")(self.h, name.encode())

    def set(self, name, value):
        if self._f("set")(self.h, name.encode(), float(value)):
            raise KeyError(name)

    def solve(self):
        return self._f("solve")(self.h)

    def phase(self, name):
        r = self._f("phase")(self.h, name.encode())
        if r < 0:
            raise KeyError(name)
        return r

    def project_soc(self, s, mu):
        s = _d(s).copy()
        self._f("project_soc")(s.ctypes.data_as(_dp), len(s), float(mu))
        return s

    def closed_loop(self, x0, steps):
        x0 = _d(x0).copy()
        iters = np.zeros(steps, dtype=np.int32)
        u0 = np.zeros((steps, self.nu))
        total = self._f("closed_loop")(self.h, x0.ctypes.data_as(_dp), steps, iters.ctypes.data_as(_ip),
                                       u0.ctypes.data_as(_dp))
        return int(total), iters, u0, x0

    def closed_loop_traj(self, x0, steps, traj, k0=0):
        """closed loop with the state reference of step k = points k0 + k ... k0 + k + N - 1 of traj [points, nx] (clamped to the
        last point); -> (total iterations, iters[steps] (negative: max_iter hit), u0[steps, nu], final x0)"""
        x0 = _d(x0).copy()
        t = np.ascontiguousarray(np.asarray(traj, dtype=np.float64).reshape(-1, self.nx))
        iters = np.zeros(steps, dtype=np.int32)
        u0 = np.zeros((steps, self.nu))
        total = self._f("closed_loop_traj")(self.h, x0.ctypes.data_as(_dp), steps, t.ctypes.data_as(_dp), t.shape[0], int(k0),
                                            iters.ctypes.data_as(_ip), u0.ctypes.data_as(_dp))
        return int(total), iters, u0, x0

    STATE_FIELDS = ("x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y",
                    "vc", "vcnew", "zc", "zcnew", "gc", "yc")
    LINEAR_FIELDS = ("vlnew", "zlnew", "gl", "yl", "vlnew_tv", "zlnew_tv", "gl_tv", "yl_tv")

    def snapshot(self, fields=None):
        return {k: self[k].copy() for k in (fields or self.STATE_FIELDS)}

    def restore(self, snap):
        for k, v in snap.items():
            self[k] = v


class OracleSolver(_CpuSolver):
    prefix = "oracle_"
    so = ORACLE_SO

    @classmethod
    def lib(cls):
        build_oracle()
        return super().lib()


class RefSolver(_CpuSolver):
    prefix = "ref_"
    so = REF_SO

    def init_sensitivity(self):
        """the reference's own tiny_initialize_sensitivity_matrices (quadrotor sizes only)"""
        if self._f("init_sensitivity")(self.h):
            raise RuntimeError("tiny_initialize_sensitivity_matrices needs nx = 12, nu = 4")

    def set_sensitivity(self, tables):
        if self.nx == 12 and self.nu == 4 and tables is None:
            return self.init_sensitivity()
        # other shapes: size the (empty) Eigen matrices through the shim, then fill them
        f = self._f("alloc_sensitivity")
        f.argtypes = [C.c_void_p]
        f(self.h)
        super().set_sensitivity(tables)

    @classmethod
    def lib(cls):
        lib = super().lib()
        lib.ref_mute_stdout.argtypes = [C.c_int]
        lib.ref_mute_stdout(1)
        lib.ref_init_sensitivity.argtypes = [C.c_void_p]
        lib.ref_stack_fill.argtypes = [C.c_int, C.c_int]
        lib.ref_stack_fill.restype = None
        lib.ref_solve_fill.argtypes = [C.c_void_p, C.c_int]
        return lib
