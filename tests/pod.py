"""ctypes mirrors of the plain-data structs of include/tinympc_amd.h (part B)."""
import ctypes as C

import numpy as np


class Mat(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_double)), ("rows", C.c_int64), ("cols", C.c_int64)]


class Vec(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_double)), ("rows", C.c_int64)]


class VecXi(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_int)), ("rows", C.c_int64)]


class TinySolution(C.Structure):
    _fields_ = [("iter", C.c_int), ("solved", C.c_int), ("x", Mat), ("u", Mat)]


class TinyCache(C.Structure):
    _fields_ = [("rho", C.c_double), ("Kinf", Mat), ("Pinf", Mat), ("Quu_inv", Mat), ("AmBKt", Mat), ("APf", Vec),
                ("BPf", Vec), ("C1", Mat), ("C2", Mat), ("dKinf_drho", Mat), ("dPinf_drho", Mat), ("dC1_drho", Mat),
                ("dC2_drho", Mat)]


class TinySettings(C.Structure):
    _fields_ = [("abs_pri_tol", C.c_double), ("abs_dua_tol", C.c_double), ("max_iter", C.c_int),
                ("check_termination", C.c_int), ("en_state_bound", C.c_int), ("en_input_bound", C.c_int),
                ("en_state_soc", C.c_int), ("en_input_soc", C.c_int), ("en_state_linear", C.c_int),
                ("en_input_linear", C.c_int), ("en_tv_state_linear", C.c_int), ("en_tv_input_linear", C.c_int),
                ("adaptive_rho", C.c_int), ("adaptive_rho_min", C.c_double), ("adaptive_rho_max", C.c_double),
                ("adaptive_rho_enable_clipping", C.c_int)]


_M = ("x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y", "x_min", "x_max", "u_min", "u_max")


class TinyWorkspace(C.Structure):
    _fields_ = ([("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int)] + [(n, Mat) for n in _M] +
                [("numStateCones", C.c_int), ("numInputCones", C.c_int), ("cx", Vec), ("cu", Vec), ("Acx", VecXi),
                 ("Acu", VecXi), ("qcx", VecXi), ("qcu", VecXi)] +
                [(n, Mat) for n in ("vc", "vcnew", "zc", "zcnew", "gc", "yc")] +
                [("numStateLinear", C.c_int), ("numInputLinear", C.c_int), ("Alin_x", Mat), ("blin_x", Vec),
                 ("Alin_u", Mat), ("blin_u", Vec)] + [(n, Mat) for n in ("vl", "vlnew", "zl", "zlnew", "gl", "yl")] +
                [("numtvStateLinear", C.c_int), ("numtvInputLinear", C.c_int)] +
                [(n, Mat) for n in ("tv_Alin_x", "tv_blin_x", "tv_Alin_u", "tv_blin_u", "vl_tv", "vlnew_tv", "zl_tv",
                                    "zlnew_tv", "gl_tv", "yl_tv")] +
                [("Q", Vec), ("R", Vec), ("Adyn", Mat), ("Bdyn", Mat), ("fdyn", Vec), ("Xref", Mat), ("Uref", Mat),
                 ("Qu", Vec), ("primal_residual_state", C.c_double), ("primal_residual_input", C.c_double),
                 ("dual_residual_state", C.c_double), ("dual_residual_input", C.c_double), ("status", C.c_int),
                 ("iter", C.c_int)])


class TinySolver(C.Structure):
    _fields_ = [("solution", C.POINTER(TinySolution)), ("settings", C.POINTER(TinySettings)),
                ("cache", C.POINTER(TinyCache)), ("work", C.POINTER(TinyWorkspace))]


def mat(a):
    """numpy (rows, cols) -> (Mat, keepalive)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    flat = np.ascontiguousarray(a.T).ravel()
    return Mat(flat.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], a.shape[1]), flat


def vec(a):
    flat = np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())
    return Vec(flat.ctypes.data_as(C.POINTER(C.c_double)), len(flat)), flat


def veci(a):
    flat = np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel())
    return VecXi(flat.ctypes.data_as(C.POINTER(C.c_int)), len(flat)), flat


def to_np(m):
    if isinstance(m, Mat):
        n = m.rows * m.cols
        return np.ctypeslib.as_array(m.data, shape=(n,)).reshape((m.rows, m.cols), order="F") if n else np.zeros((m.rows, m.cols))
    return np.ctypeslib.as_array(m.data, shape=(m.rows,)) if m.rows else np.zeros(0)
