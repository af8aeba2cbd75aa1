"""TEST INFRASTRUCTURE -- seeded parity scenarios shared by oracle/gen_golden.py and tests/.

A *suite* is (problem, config, cases):
  problem : dict(nx, nu, N, A, B, f, Q, R, rho)               (user Q/R diagonals, no rho)
  config  : dict(max_iter, abs_pri_tol, abs_dua_tol, check_termination, en_state_bound,
                 en_input_bound, en_state_soc, en_input_soc, x_min, x_max, u_min, u_max,
                 state_cone=(A,q,c) | None, input_cone=(A,q,c) | None)
  cases   : dict of arrays with a leading batch axis B; matrices keep the reference's
            shapes (types.hpp:88-208): x0 [B,nx]; Xref, vnew, g, v, x, gc [B,nx,N];
            Uref, znew, y, z, u, yc [B,nu,N-1].

``run_cases(SolverClass, suite)`` executes every case on a CPU implementation (oracle or
real reference) as ONE tiny_solve from the given warm state and returns the resulting
workspace (same array shapes) + iter/solved/status/ret/residuals.
"""
from __future__ import annotations

import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PROBLEMS_JSON = os.path.join(_HERE, "..", "tinympc_amd", "data", "problems.json")

STATE_IN = ("vnew", "g", "v", "x", "gc", "gl", "gl_tv")       # nx x N
INPUT_IN = ("znew", "y", "z", "u", "yc", "yl", "yl_tv")       # nu x (N-1)
STATE_OUT = ("x", "vnew", "g", "v", "vcnew", "gc", "q", "p", "sol_x", "vlnew", "gl", "vlnew_tv", "gl_tv")
INPUT_OUT = ("u", "znew", "y", "z", "zcnew", "yc", "r", "d", "sol_u", "zlnew", "yl", "zlnew_tv", "yl_tv")
LINEAR_FLAGS = ("en_state_linear", "en_input_linear", "en_tv_state_linear", "en_tv_input_linear")
# adaptive rho (rho_benchmark.cpp): the cache is per-instance STATE then -- rho, Kinf, Pinf, C1, C2 move during a solve and
# persist into the next one.  cases may carry them as cache_rho [B], cache_Kinf [B,nu,nx], ...; outputs are rho, Kinf, ...
CACHE_STATE = ("Kinf", "Pinf", "C1", "C2")
SENS = ("dKinf_drho", "dPinf_drho", "dC1_drho", "dC2_drho")
SENS_JSON = os.path.join(_HERE, "..", "tinympc_amd", "data", "sensitivity_quadrotor.json")
SCALARS_OUT = ("iter", "status", "sol_iter", "sol_solved", "primal_residual_state", "primal_residual_input",
               "dual_residual_state", "dual_residual_input")


def load_problem(name):
    p = json.load(open(PROBLEMS_JSON))[name]
    out = dict(nx=p["nx"], nu=p["nu"], N=p["N"], rho=float(p["rho"]),
               A=np.array(p["A"], dtype=np.float64), B=np.array(p["B"], dtype=np.float64),
               f=np.array(p["f"], dtype=np.float64), Q=np.array(p["Q"], dtype=np.float64),
               R=np.array(p["R"], dtype=np.float64))
    extra = {k: v for k, v in p.items() if k not in out and k != "source"}
    return out, extra


def default_config(prob, **kw):
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    cfg = dict(max_iter=1000, abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=1,
               en_state_bound=1, en_input_bound=1, en_state_soc=0, en_input_soc=0,
               x_min=np.full((nx, N), -1e17), x_max=np.full((nx, N), 1e17),
               u_min=np.full((nu, N - 1), -1e17), u_max=np.full((nu, N - 1), 1e17),
               state_cone=None, input_cone=None,
               en_state_linear=0, en_input_linear=0, en_tv_state_linear=0, en_tv_input_linear=0,
               linear=None,        # (Alin_x (ns,nx), blin_x (ns,), Alin_u (ni,nu), blin_u (ni,))
               tv_linear=None)     # (tv_Alin_x (ns*N,nx), tv_blin_x (ns,N), tv_Alin_u (ni*(N-1),nu), tv_blin_u (ni,N-1))
    cfg.update(kw)
    for k, shp in (("x_min", (nx, N)), ("x_max", (nx, N)), ("u_min", (nu, N - 1)), ("u_max", (nu, N - 1))):
        a = np.asarray(cfg[k], dtype=np.float64)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        cfg[k] = np.ascontiguousarray(np.broadcast_to(a, shp))
    return cfg


def quadrotor_sensitivity():
    """the four d(.)/d(rho) tables tiny_initialize_sensitivity_matrices leaves in the cache (oracle/extract_sensitivity.py)"""
    t = json.load(open(SENS_JSON))
    return {k: np.array(t[k]["data"], dtype=np.float64).reshape(t[k]["cols"], t[k]["rows"]).T.copy() for k in SENS}


def zero_cases(prob, B):
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    c = dict(x0=np.zeros((B, nx)), Xref=np.zeros((B, nx, N)), Uref=np.zeros((B, nu, N - 1)))
    for k in STATE_IN:
        c[k] = np.zeros((B, nx, N))
    for k in INPUT_IN:
        c[k] = np.zeros((B, nu, N - 1))
    return c


def make_solver(cls, prob, cfg):
    s = cls(prob["nx"], prob["nu"], prob["N"], prob["A"], prob["B"], prob["f"], prob["Q"], prob["R"], prob["rho"])
    s.set_bounds(cfg["x_min"], cfg["x_max"], cfg["u_min"], cfg["u_max"])
    sc, ic = cfg.get("state_cone"), cfg.get("input_cone")
    if sc is not None or ic is not None:
        sc = sc or ([], [], [])
        ic = ic or ([], [], [])
        s.set_cones(sc[0], sc[1], sc[2], ic[0], ic[1], ic[2])
    if cfg.get("linear") is not None:
        s.set_linear(*cfg["linear"])
    if cfg.get("tv_linear") is not None:
        s.set_tv_linear(*cfg["tv_linear"])
    for k in ("max_iter", "abs_pri_tol", "abs_dua_tol", "check_termination", "en_state_bound", "en_input_bound",
              "en_state_soc", "en_input_soc") + LINEAR_FLAGS:
        s.set(k, cfg.get(k, 0))
    if cfg.get("adaptive_rho"):
        s.set_sensitivity({k: cfg["sensitivity." + k] for k in SENS})
        s.set_adaptive_rho(1, cfg.get("adaptive_rho_min", 1.0), cfg.get("adaptive_rho_max", 100.0),
                           cfg.get("adaptive_rho_enable_clipping", 1))
    return s


def load_case(s, cases, b):
    for k in STATE_IN + INPUT_IN:
        if k in cases:
            s[k] = cases[k][b]
    s["Xref"] = cases["Xref"][b]
    s["Uref"] = cases["Uref"][b]
    s["x"][:, 0] = cases["x0"][b]          # tiny_set_x0 (tiny_api.cpp:443-453)


def run_cases(cls, suite, fields=None):
    prob, cfg, cases = suite["problem"], suite["config"], suite["cases"]
    B = cases["x0"].shape[0]
    s = make_solver(cls, prob, cfg)
    out = {}
    names = [k for k in (STATE_OUT + INPUT_OUT) if fields is None or k in fields]
    for k in names:
        out[k] = np.zeros((B,) + s[k].shape)
    for k in SCALARS_OUT + ("ret",):
        out[k] = np.zeros(B)
    adaptive = bool(cfg.get("adaptive_rho"))
    if adaptive:                                   # the cache is per-case state: start every case from its own (or the fresh) cache
        fresh = {k: s[k].copy() for k in CACHE_STATE}
        fresh_rho = s.get("rho")
        out["rho"] = np.zeros(B)
        for k in CACHE_STATE:
            out[k] = np.zeros((B,) + s[k].shape)
    for b in range(B):
        load_case(s, cases, b)
        if adaptive:
            s.set("rho", cases["cache_rho"][b] if "cache_rho" in cases else fresh_rho)
            for k in CACHE_STATE:
                s[k] = cases["cache_" + k][b] if ("cache_" + k) in cases else fresh[k]
        out["ret"][b] = s.solve()
        for k in names:
            out[k][b] = s[k]
        for k in SCALARS_OUT:
            out[k][b] = s.get(k)
        if adaptive:
            out["rho"][b] = s.get("rho")
            for k in CACHE_STATE:
                out[k][b] = s[k]
    s.close()
    return out


# ----------------------------------------------------------------------------- suites

def _hover_cfg(prob, extra):
    h = extra["hover"]
    return default_config(prob, max_iter=h["max_iter"], x_min=np.full((prob["nx"], 1), h["x_min"]),
                          x_max=np.full((prob["nx"], 1), h["x_max"]), u_min=np.full((prob["nu"], 1), h["u_min"]),
                          u_max=np.full((prob["nu"], 1), h["u_max"]))


def hover_suite(cls, steps=(0, 1, 4, 5, 6, 7, 8, 12, 20, 30, 42, 60, 75, 99)):
    """BASELINE config 2: warm states captured along the 100-step quadrotor hover episode
    (examples/quadrotor_hovering.cpp).  Needs a CPU implementation to roll the loop."""
    prob, extra = load_problem("quadrotor_20hz")
    cfg = _hover_cfg(prob, extra)
    h = extra["hover"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    s = make_solver(cls, prob, cfg)
    s["Xref"] = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
    x0 = np.array(h["x0"], dtype=np.float64)
    cases = zero_cases(prob, len(steps))
    ep_iters, ep_u0, ep_x0 = [], [], []
    j = 0
    for k in range(h["steps"]):
        s["x"][:, 0] = x0
        if k in steps:
            cases["x0"][j] = x0
            cases["Xref"][j] = s["Xref"]
            cases["Uref"][j] = s["Uref"]
            for f in STATE_IN + INPUT_IN:
                cases[f][j] = s[f]
            j += 1
        ep_x0.append(x0.copy())
        s.solve()
        ep_iters.append(int(s.get("sol_iter")))
        ep_u0.append(s["u"][:, 0].copy())
        x0 = prob["A"] @ x0 + prob["B"] @ s["u"][:, 0] + prob["f"]     # examples/quadrotor_hovering.cpp:92
    s.close()
    episode = dict(iters=np.array(ep_iters, dtype=np.int32), u0=np.array(ep_u0), x0=np.array(ep_x0),
                   steps=np.array(steps, dtype=np.int32))
    return dict(problem=prob, config=cfg, cases=cases, episode=episode)


def adaptive_cfg(cfg, rho_min=1.0, rho_max=100.0, clip=1, sensitivity=None):
    """switch adaptive rho on in a config (settings types.hpp:75-79; tables tiny_api.cpp:479-540)"""
    sens = sensitivity or quadrotor_sensitivity()
    out = dict(cfg, adaptive_rho=1, adaptive_rho_min=float(rho_min), adaptive_rho_max=float(rho_max),
               adaptive_rho_enable_clipping=int(clip))
    for k in SENS:
        out["sensitivity." + k] = np.asarray(sens[k], dtype=np.float64)
    return out


def hover_adaptive_suite(cls, steps=(0, 1, 2, 5, 6, 7, 8, 20, 21, 60, 93, 94, 99), rho_min=1.0, rho_max=100.0, clip=1):
    """The hover episode with adaptive_rho = 1: warm states AND the per-solve cache state (rho, Kinf, Pinf, C1, C2, which
    persist from solve to solve) captured along the episode."""
    prob, extra = load_problem("quadrotor_20hz")
    cfg = adaptive_cfg(_hover_cfg(prob, extra), rho_min, rho_max, clip)
    h = extra["hover"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    s = make_solver(cls, prob, cfg)
    s["Xref"] = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
    x0 = np.array(h["x0"], dtype=np.float64)
    cases = zero_cases(prob, len(steps))
    cases["cache_rho"] = np.zeros(len(steps))
    for k in CACHE_STATE:
        cases["cache_" + k] = np.zeros((len(steps),) + s[k].shape)
    ep_iters, ep_rho = [], []
    j = 0
    for k in range(h["steps"]):
        s["x"][:, 0] = x0
        if k in steps:
            cases["x0"][j] = x0
            cases["Xref"][j] = s["Xref"]
            for f in STATE_IN + INPUT_IN:
                cases[f][j] = s[f]
            cases["cache_rho"][j] = s.get("rho")
            for c in CACHE_STATE:
                cases["cache_" + c][j] = s[c]
            j += 1
        s.solve()
        ep_iters.append(int(s.get("sol_iter")) * (1 if s.get("sol_solved") else -1))
        ep_rho.append(s.get("rho"))
        x0 = prob["A"] @ x0 + prob["B"] @ s["u"][:, 0] + prob["f"]
    s.close()
    episode = dict(iters=np.array(ep_iters, dtype=np.int32), rho=np.array(ep_rho), steps=np.array(steps, dtype=np.int32),
                   x_final=x0)
    return dict(problem=prob, config=cfg, cases=cases, episode=episode)


def tracking_adaptive_suite(B=16, seed=777, rho_min=0.5, rho_max=40.0, clip=1, max_iter=60):
    """config-3 style random tracking instances, one cold solve each with adaptive rho: every instance takes its own rho
    path (different clip range than the hover suite, so un-clipped values survive)."""
    base = tracking_random_suite(B=B, seed=seed)
    cfg = adaptive_cfg(dict(base["config"], max_iter=max_iter), rho_min, rho_max, clip)
    return dict(problem=base["problem"], config=cfg, cases=base["cases"])


def tracking_random_suite(B=24, seed=20260923):
    """BASELINE config 3: per-instance random references around the
    y-axis-line trajectory, duals zeroed (examples/quadrotor_tracking.cpp:89-93), one cold solve."""
    prob, extra = load_problem("quadrotor_20hz")
    cfg = _hover_cfg(prob, extra)
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    traj = np.array(extra["y_axis_line"], dtype=np.float64)      # [301, nx]
    cases = zero_cases(prob, B)
    for j in range(B):
        rng = np.random.default_rng(seed + j)
        k = int(rng.integers(0, 291))
        Xref = traj[k:k + N].T + rng.normal(0.0, 0.05, (nx, N))
        cases["Xref"][j] = Xref
        cases["Uref"][j] = rng.normal(0.0, 0.05, (nu, N - 1))
        x0 = Xref[:, 0].copy()
        x0[:3] += rng.normal(0.0, 0.1, 3)
        cases["x0"][j] = x0
    return dict(problem=prob, config=cfg, cases=cases)


def rocket_cfg(prob, extra, en_state_soc=0, en_input_soc=1):
    m = extra["mpc"]
    return default_config(prob, max_iter=m["max_iter"], abs_pri_tol=m["abs_pri_tol"],
                          x_min=np.array(m["x_min"]), x_max=np.array(m["x_max"]),
                          u_min=np.full((prob["nu"], 1), m["u_min"]), u_max=np.full((prob["nu"], 1), m["u_max"]),
                          en_state_soc=en_state_soc, en_input_soc=en_input_soc,
                          state_cone=(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"]),
                          input_cone=(m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"]))


def rocket_xref(extra, k, N):
    """examples/rocket_landing_mpc.cpp:111-113: Xref[:,i] = xinit + (xg - xinit) * (i+k)/(NTOTAL-1)."""
    m = extra["mpc"]
    xinit, xg = np.array(m["xinit"], dtype=np.float64), np.array(m["xg"], dtype=np.float64)
    return np.stack([xinit + (xg - xinit) * float(i + k) / (m["NTOTAL"] - 1) for i in range(N)], axis=1)


def rocket_random_suite(B=16, seed=20260923, en_state_soc=0, en_input_soc=1):
    """BASELINE config 4: rocket landing, second-order-cone thrust constraint on, perturbed
    initial states x0_j = 1.1*xinit*(1 + 0.05*U(-1,1)), one cold solve each."""
    prob, extra = load_problem("rocket_landing_20hz")
    cfg = rocket_cfg(prob, extra, en_state_soc, en_input_soc)
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    m = extra["mpc"]
    cases = zero_cases(prob, B)
    for j in range(B):
        rng = np.random.default_rng(seed + j)
        cases["x0"][j] = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, nx))
        cases["Xref"][j] = rocket_xref(extra, 0, N)
        cases["Uref"][j][2, :] = m["uref_z"]
    return dict(problem=prob, config=cfg, cases=cases)


def rocket_episode_suite(cls, en_state_soc=0, en_input_soc=1, steps=(0, 1, 2, 5, 10, 30, 60, 89)):
    """Warm states along examples/rocket_landing_mpc.cpp's 90-step loop (SOC flags explicit)."""
    prob, extra = load_problem("rocket_landing_20hz")
    cfg = rocket_cfg(prob, extra, en_state_soc, en_input_soc)
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    m = extra["mpc"]
    s = make_solver(cls, prob, cfg)
    x0 = 1.1 * np.array(m["xinit"], dtype=np.float64)
    s["Uref"][2, :] = m["uref_z"]
    s["Xref"] = rocket_xref(extra, 0, N)
    s["p"][:, N - 1] = -(s["Pinf"] @ s["Xref"][:, N - 1])          # rocket_landing_mpc.cpp:126 (overwritten anyway)
    cases = zero_cases(prob, len(steps))
    ep_iters, ep_solved, ep_u0 = [], [], []
    j = 0
    for k in range(m["NTOTAL"] - N):
        s["x"][:, 0] = x0
        s["Xref"] = rocket_xref(extra, k, N)
        s["Uref"][2, :] = m["uref_z"]
        if k in steps:
            cases["x0"][j] = x0
            cases["Xref"][j] = s["Xref"]
            cases["Uref"][j] = s["Uref"]
            for f in STATE_IN + INPUT_IN:
                cases[f][j] = s[f]
            j += 1
        s.solve()
        ep_iters.append(int(s.get("sol_iter")))
        ep_solved.append(int(s.get("sol_solved")))
        ep_u0.append(s["u"][:, 0].copy())
        x0 = prob["A"] @ x0 + prob["B"] @ s["u"][:, 0] + prob["f"]
    s.close()
    episode = dict(iters=np.array(ep_iters, dtype=np.int32), solved=np.array(ep_solved, dtype=np.int32),
                   u0=np.array(ep_u0), steps=np.array(steps, dtype=np.int32))
    return dict(problem=prob, config=cfg, cases=cases, episode=episode)


def cartpole_suite(cls, steps=(0, 1, 2, 3, 10, 50, 200, 389)):
    """BASELINE config 1: examples/cartpole_example.cpp, 390 closed-loop solves."""
    prob, extra = load_problem("cartpole")
    m = extra["mpc"]
    cfg = default_config(prob, max_iter=m["max_iter"])
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    s = make_solver(cls, prob, cfg)
    s["Xref"] = np.tile(np.array(m["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
    x0 = np.array(m["x0"], dtype=np.float64)
    cases = zero_cases(prob, len(steps))
    ep_iters, ep_u0, ep_err = [], [], []
    j = 0
    for k in range(m["steps"]):
        ep_err.append(float(np.linalg.norm(x0 - s["Xref"][:, 1])))
        s["x"][:, 0] = x0
        if k in steps:
            cases["x0"][j] = x0
            cases["Xref"][j] = s["Xref"]
            for f in STATE_IN + INPUT_IN:
                cases[f][j] = s[f]
            j += 1
        s.solve()
        ep_iters.append(int(s.get("sol_iter")))
        ep_u0.append(s["u"][:, 0].copy())
        x0 = prob["A"] @ x0 + prob["B"] @ s["u"][:, 0]
    s.close()
    episode = dict(iters=np.array(ep_iters, dtype=np.int32), u0=np.array(ep_u0), err=np.array(ep_err),
                   steps=np.array(steps, dtype=np.int32))
    return dict(problem=prob, config=cfg, cases=cases, episode=episode)


def random_problem(nx, nu, N):
    """BASELINE config 5 generator: one shared (A,B,Q,R) per cell."""
    rng = np.random.default_rng(1000 * nx + 10 * nu + N)
    M = rng.standard_normal((nx, nx))
    A = M * 0.95 / np.max(np.abs(np.linalg.eigvals(M)))
    Bm = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = rng.uniform(1, 10, nx)
    R = rng.uniform(0.1, 1, nu)
    return dict(nx=nx, nu=nu, N=N, rho=1.0, A=A, B=Bm, f=np.zeros(nx), Q=Q, R=R), rng


def sweep_suite(nx, nu, N, B=4, max_iter=500):
    prob, rng = random_problem(nx, nu, N)
    cfg = default_config(prob, max_iter=max_iter, u_min=np.full((nu, 1), -0.5), u_max=np.full((nu, 1), 0.5))
    cases = zero_cases(prob, B)
    for j in range(B):
        cases["x0"][j] = rng.uniform(-1, 1, nx)
        cases["Xref"][j] = np.tile(rng.uniform(-0.2, 0.2, (nx, 1)), (1, N))
    return dict(problem=prob, config=cfg, cases=cases)


def sweep_cone_suite(nx, nu, N, B=3, max_iter=120, state_cone=False, seed=31):
    """a config-5 sweep cell with a second-order cone on the first three inputs (and, state_cone, on three states) and a warm
    random slack / dual state: the tile kernel's cone variant on wide / long shapes (tile_dims.txt)"""
    suite = sweep_suite(nx, nu, N, B=B, max_iter=max_iter)
    rng = np.random.default_rng(seed)
    assert nu >= 3 and nx >= 4, "the reference's project_soc only handles dimension 3 (admm.cpp:53)"
    suite["config"].update(en_input_soc=1, input_cone=([0], [3], [0.6]))
    if state_cone:
        suite["config"].update(en_state_soc=1, state_cone=([1], [3], [0.8]))
    for k in ("vnew", "znew", "g", "y", "v", "z", "gc", "yc", "x", "u"):
        suite["cases"][k] = rng.normal(0.0, 0.2, suite["cases"][k].shape)
    return suite


def random_state_suite(name="quadrotor_20hz", B=8, seed=7, scale=0.3, soc=False):
    """Fully random warm state (every input array random) -- transpose-detecting: no
    symmetric / replicated structure anywhere (guide rule 16)."""
    prob, extra = load_problem(name)
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(seed)
    kw = dict(max_iter=37, x_min=rng.uniform(-1.0, -0.2, (nx, N)), x_max=rng.uniform(0.2, 1.0, (nx, N)),
              u_min=rng.uniform(-0.5, -0.1, (nu, N - 1)), u_max=rng.uniform(0.1, 0.5, (nu, N - 1)))
    if soc == "overlap":
        # cones that SHARE rows: the reference projects the cones of a column one after the other, overlapping or not
        # (admm.cpp:111-135), so the order matters -- rows 0..4 of the state are hit by three cones, rows 0..3 of the input by two
        kw.update(en_state_soc=1, en_input_soc=1, state_cone=([0, 2, 1], [3, 3, 3], [0.7, 0.5, 0.9]),
                  input_cone=([0, 1], [3, 3], [0.4, 0.6]) if nu >= 4 else ([0, 0], [3, 3], [0.4, 0.6]))
    elif soc:
        kw.update(en_state_soc=1, en_input_soc=1, state_cone=([1], [3], [0.7]), input_cone=([0], [3], [0.4]))
    cfg = default_config(prob, **kw)
    cases = zero_cases(prob, B)
    for k, v in cases.items():
        cases[k] = rng.normal(0.0, scale, v.shape)
    return dict(problem=prob, config=cfg, cases=cases)


def linear_example_cfg(prob, tv=False, k=0):
    """examples/quadrotor_linear_constraints.cpp:41-73 / quadrotor_tv_linear_constraints.cpp:41-103: altitude
    ceiling z <= 3 (time-varying: z <= z_lim(k+i)) and total thrust u1+u2+u3+u4 <= 6, boxes disabled."""
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    NTOTAL = 50
    ax = np.zeros((1, nx)); ax[0, 2] = 1.0
    au = np.ones((1, nu))
    if not tv:
        return default_config(prob, max_iter=100, en_state_bound=0, en_input_bound=0, en_state_linear=1,
                              en_input_linear=1, linear=(ax, np.array([3.0]), au, np.array([6.0])))
    zlim = np.array([1.1 + (3.0 - 1.1) * i / (NTOTAL - N - 1) for i in range(NTOTAL)])
    return default_config(prob, max_iter=100, en_state_bound=0, en_input_bound=0, en_tv_state_linear=1,
                          en_tv_input_linear=1,
                          tv_linear=(np.tile(ax, (N, 1)), zlim[k:k + N].reshape(1, N), np.tile(au, (N - 1, 1)),
                                     np.full((1, N - 1), 6.0)))


def linear_example_suite(cls, tv=False, steps=(0, 1, 5, 20, 39)):
    """Warm states along the 40-step loop of the (tv_)linear-constraint examples (16 / 17 of
    40 solves converge, converged-iteration sums 544 / 634).  For tv the bounds change every step, so each case
    carries its own config; suites returned as a list of single-step suites."""
    prob, _ = load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    NTOTAL = 50
    x0 = np.array([-2.0, -2.0, 1.0] + [0.0] * 9)
    xgoal = np.array([2.0, 2.0, 4.0] + [0.0] * 9)
    s = make_solver(cls, prob, linear_example_cfg(prob, tv, 0))
    x = x0.copy()
    out_suites, iters, solved = [], [], []
    for k in range(NTOTAL - N):
        cfg = linear_example_cfg(prob, tv, k)
        if tv:
            s.set_tv_linear(*cfg["tv_linear"])
        for i in range(N):
            alpha = float(k + i) / (NTOTAL - 1)
            s["Xref"][:, i] = (1 - alpha) * x0 + alpha * xgoal
        s["x"][:, 0] = x
        if k in steps:
            cases = zero_cases(prob, 1)
            cases["x0"][0] = x
            cases["Xref"][0] = s["Xref"]
            for f in STATE_IN + INPUT_IN:
                cases[f][0] = s[f]
            out_suites.append(dict(problem=prob, config=cfg, cases=cases))
        s.solve()
        iters.append(int(s.get("sol_iter")))
        solved.append(int(s.get("sol_solved")))
        x = prob["A"] @ x + prob["B"] @ s["u"][:, 0]
    s.close()
    return out_suites, np.array(iters), np.array(solved)


def random_linear_suite(name="quadrotor_20hz", B=6, seed=21, tv=True, static=True, box=True, soc=False):
    """Every slack family at once on a fully random workspace: box + (cone) + static + time-varying half-spaces,
    several constraints per knot (sequential projections, admm.cpp:148-211)."""
    prob, _ = load_problem(name)
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(seed)
    kw = dict(max_iter=29, en_state_bound=int(box), en_input_bound=int(box),
              x_min=rng.uniform(-1.0, -0.2, (nx, N)), x_max=rng.uniform(0.2, 1.0, (nx, N)),
              u_min=rng.uniform(-0.5, -0.1, (nu, N - 1)), u_max=rng.uniform(0.1, 0.5, (nu, N - 1)))
    if static:
        kw.update(en_state_linear=1, en_input_linear=1,
                  linear=(rng.normal(0, 1, (3, nx)), rng.normal(0, 0.3, 3), rng.normal(0, 1, (2, nu)), rng.normal(0, 0.3, 2)))
    if tv:
        kw.update(en_tv_state_linear=1, en_tv_input_linear=1,
                  tv_linear=(rng.normal(0, 1, (2 * N, nx)), rng.normal(0, 0.3, (2, N)),
                             rng.normal(0, 1, (1 * (N - 1), nu)), rng.normal(0, 0.3, (1, N - 1))))
    if soc == "overlap":
        # cones that SHARE rows: the reference projects the cones of a column one after the other, overlapping or not
        # (admm.cpp:111-135), so the order matters -- rows 0..4 of the state are hit by three cones, rows 0..3 of the input by two
        kw.update(en_state_soc=1, en_input_soc=1, state_cone=([0, 2, 1], [3, 3, 3], [0.7, 0.5, 0.9]),
                  input_cone=([0, 1], [3, 3], [0.4, 0.6]) if nu >= 4 else ([0, 0], [3, 3], [0.4, 0.6]))
    elif soc:
        kw.update(en_state_soc=1, en_input_soc=1, state_cone=([1], [3], [0.7]), input_cone=([0], [3], [0.4]))
    cfg = default_config(prob, **kw)
    cases = zero_cases(prob, B)
    for k, v in cases.items():
        cases[k] = rng.normal(0.0, 0.3, v.shape)
    return dict(problem=prob, config=cfg, cases=cases)


# ----------------------------------------------------------------------------- (de)serialisation

def save_suite(path, suite, outputs):
    flat = {}
    for k, v in suite["problem"].items():
        flat["problem." + k] = np.asarray(v)
    for k, v in suite["config"].items():
        if v is None:
            continue
        if k in ("linear", "tv_linear"):
            for part, arr in zip(("Ax", "bx", "Au", "bu"), v):
                flat[f"config.{k}.{part}"] = np.asarray(arr, dtype=np.float64)
        elif k.endswith("_cone"):
            flat[f"config.{k}.A"] = np.asarray(v[0], dtype=np.int32)
            flat[f"config.{k}.q"] = np.asarray(v[1], dtype=np.int32)
            flat[f"config.{k}.c"] = np.asarray(v[2], dtype=np.float64)
        else:
            flat["config." + k] = np.asarray(v)
    for k, v in suite["cases"].items():
        if np.any(v):
            flat["cases." + k] = v
    flat["cases.B"] = np.array(suite["cases"]["x0"].shape[0])
    for k, v in outputs.items():
        flat["out." + k] = v
    for k, v in suite.get("episode", {}).items():
        flat["episode." + k] = v
    np.savez_compressed(path, **flat)


def load_suite(path):
    z = np.load(path)
    prob, cfg, cases, out, episode = {}, {}, {}, {}, {}
    cones = {}
    for key in z.files:
        grp, name = key.split(".", 1)
        v = z[key]
        if grp == "problem":
            prob[name] = v.item() if v.ndim == 0 else v
        elif grp == "config":
            if "_cone." in name or name.startswith(("linear.", "tv_linear.")):
                cn, part = name.split(".")
                cones.setdefault(cn, {})[part] = v
            else:
                cfg[name] = v.item() if v.ndim == 0 else v
        elif grp == "cases":
            cases[name] = v
        elif grp == "out":
            out[name] = v
        elif grp == "episode":
            episode[name] = v
    for k in ("nx", "nu", "N"):
        prob[k] = int(prob[k])
    prob["rho"] = float(prob["rho"])
    for k in ("max_iter", "check_termination", "en_state_bound", "en_input_bound", "en_state_soc", "en_input_soc"):
        cfg[k] = int(cfg[k])
    for k in LINEAR_FLAGS:
        cfg[k] = int(cfg.get(k, 0))
    if "adaptive_rho" in cfg:
        cfg["adaptive_rho"] = int(cfg["adaptive_rho"])
        cfg["adaptive_rho_enable_clipping"] = int(cfg.get("adaptive_rho_enable_clipping", 1))
    for k in ("linear", "tv_linear"):
        cfg[k] = tuple(cones[k][p] for p in ("Ax", "bx", "Au", "bu")) if k in cones else None
    for cn in ("state_cone", "input_cone"):
        cfg[cn] = (cones[cn]["A"], cones[cn]["q"], cones[cn]["c"]) if cn in cones else None
    B = int(cases.pop("B"))
    full = zero_cases(prob, B)
    full.update(cases)
    return dict(problem=prob, config=cfg, cases=full, episode=episode), out
