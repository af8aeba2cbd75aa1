"""Run a parity suite (oracle/scenarios.py) through the product: libtinympc_amd.so via the C ABI."""
import os

import numpy as np

import tinympc_amd as tm

IN_FIELDS = ("Xref", "Uref", "vnew", "znew", "g", "y", "v", "z", "x", "u", "gc", "yc")
LIN_IN = ("gl", "yl")
TV_IN = ("gl_tv", "yl_tv")
OUT_FIELDS = ("x", "u", "vnew", "znew", "g", "y", "v", "z")
SOC_OUT = ("vcnew", "zcnew", "gc", "yc")
LIN_OUT = ("vlnew", "zlnew", "gl", "yl")
TV_OUT = ("vlnew_tv", "zlnew_tv", "gl_tv", "yl_tv")


def make_batch(suite, batch=None, replicate=1):
    prob, cfg = suite["problem"], suite["config"]
    B = (batch or suite["cases"]["x0"].shape[0]) * replicate
    s = tm.TinyBatchSolver(prob["A"], prob["B"], prob["f"], prob["Q"], prob["R"], prob["rho"], prob["nx"],
                           prob["nu"], prob["N"], B)
    s.set_bound_constraints(cfg["x_min"], cfg["x_max"], cfg["u_min"], cfg["u_max"])
    sc_, ic_ = cfg.get("state_cone"), cfg.get("input_cone")
    if sc_ is not None or ic_ is not None:
        sc_ = sc_ or ([], [], [])
        ic_ = ic_ or ([], [], [])
        s.set_cone_constraints(sc_[0], sc_[1], sc_[2], ic_[0], ic_[1], ic_[2])
    if cfg.get("linear") is not None:
        s.set_linear_constraints(*cfg["linear"])
    if cfg.get("tv_linear") is not None:
        s.set_tv_linear_constraints(*cfg["tv_linear"])
    s.update_settings(cfg["abs_pri_tol"], cfg["abs_dua_tol"], cfg["max_iter"], cfg["check_termination"],
                      cfg["en_state_bound"], cfg["en_input_bound"], cfg["en_state_soc"], cfg["en_input_soc"],
                      cfg.get("en_state_linear", 0), cfg.get("en_input_linear", 0), cfg.get("en_tv_state_linear", 0),
                      cfg.get("en_tv_input_linear", 0))
    if cfg.get("adaptive_rho"):                       # adaptive rho: sensitivity tables + settings (types.hpp:75-79)
        s.set_sensitivity(*[cfg["sensitivity." + k] for k in ("dKinf_drho", "dPinf_drho", "dC1_drho", "dC2_drho")])
        s.set_adaptive_rho(1, cfg.get("adaptive_rho_min", 1.0), cfg.get("adaptive_rho_max", 100.0), cfg.get("adaptive_rho_enable_clipping", 1))
    for kv in filter(None, os.environ.get("TINYMPC_TEST_OPTS", "").split(",")):     # rerun the suite under any option set
        k, v = kv.split("=")
        s.set_option(k, int(v))
    return s


def run_cases_hip(suite, replicate=1, debug=False, options=None):
    """One tiny_solve per case, all cases in ONE launch. replicate>1 tiles the cases (bigger grids)."""
    cases = suite["cases"]
    s = make_batch(suite, replicate=replicate)
    for k, v in (options or {}).items():
        s.set_option(k, v)
    if debug:
        s.set_option("debug", 1)
    rep = (lambda a: np.concatenate([a] * replicate, axis=0)) if replicate > 1 else (lambda a: a)
    s.set_x0(rep(cases["x0"]))
    cfg = suite["config"]
    lin = cfg.get("en_state_linear", 0) or cfg.get("en_input_linear", 0)
    tvl = cfg.get("en_tv_state_linear", 0) or cfg.get("en_tv_input_linear", 0)
    for f in IN_FIELDS + (LIN_IN if lin else ()) + (TV_IN if tvl else ()):
        if f in cases:
            s.set(f, rep(cases[f]))
    adaptive = bool(cfg.get("adaptive_rho"))
    if adaptive:                                      # the cache is per-instance state: each case's own, else tiny_setup's
        for k in ("rho", "Kinf", "Pinf", "C1", "C2"):
            if "cache_" + k in cases:
                s.set_cache_state(k, rep(cases["cache_" + k]))
    ret = s.solve()
    soc = cfg["en_state_soc"] or cfg["en_input_soc"]
    out = {f: s.get(f) for f in OUT_FIELDS + (SOC_OUT if soc else ()) + (LIN_OUT if lin else ()) + (TV_OUT if tvl else ())}
    if adaptive:
        for k in ("rho", "Kinf", "Pinf", "C1", "C2"):
            out[k] = s.get_cache_state(k)
    general = s.kernel_path() == "cover"          # the coverage kernel always leaves q|r and p|d behind
    if debug or general:
        for f in ("q", "r", "p", "d"):
            out[f] = s.get(f)
    st = s.status()
    out.update(iter=st["iter"].astype(float), sol_iter=st["iter"].astype(float), sol_solved=st["solved"].astype(float),
               status=st["status"].astype(float), ret=(1.0 - st["solved"]).astype(float),
               primal_residual_state=st["primal_residual_state"], primal_residual_input=st["primal_residual_input"],
               dual_residual_state=st["dual_residual_state"], dual_residual_input=st["dual_residual_input"])
    out["sol_x"], out["sol_u"] = out["vnew"], out["znew"]          # solution = slack (admm.cpp:436-437)
    out["batch_ret"] = ret
    out["tile_form"] = s.get_option("last_tile_form") if s.kernel_path() == "tile" else -1     # (W, R, LM) of the tile_dims.txt entry that ran
    out["half_rows"] = s.get_option("last_half_rows")         # 1: the one-row kernel's HALF form (two instances per DPP row) ran
    s.close()
    return out
