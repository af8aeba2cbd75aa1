"""GPU: the native multi-GPU surface (TinyGroup, include/tinympc_amd.h section C) on the one GPU a test box has.

* one shard on device 0: the exchange is a real RCCL all-gather (ncclCommInitAll with one rank) -- the call path of
  the 8-GPU layout;
* two / three shards sharing device 0: the sharding logic (contiguous and round-robin scatter / gather of per-instance
  data, per-shard launches on their own streams, reduction of the 64-byte messages) against the unsharded batch, bit
  for bit.  RCCL refuses two ranks on one device, so these exchange through host memory with the same reduction."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scenarios as sc  # noqa: E402
import tinympc_amd as tm  # noqa: E402
from hip_runner import make_batch  # noqa: E402

pytestmark = pytest.mark.gpu
FIELDS = ("x", "u", "vnew", "znew", "g", "y", "v", "z")


def _unsharded(suite):
    cases = suite["cases"]
    s = make_batch(suite)
    s.set_x0(cases["x0"])
    s.set_x_ref(cases["Xref"])
    s.set_u_ref(cases["Uref"])
    ret = s.solve()
    out = {f: s.get(f) for f in FIELDS}
    st = s.status()
    stats = s.reduce_stats()
    s.close()
    return ret, out, st, stats


def _group(suite, **kw):
    prob, cfg, cases = suite["problem"], suite["config"], suite["cases"]
    B = cases["x0"].shape[0]
    g = tm.TinyGroupSolver.from_problem(prob, B, **kw)
    g.set_bound_constraints(cfg["x_min"], cfg["x_max"], cfg["u_min"], cfg["u_max"])
    g.update_settings(cfg["abs_pri_tol"], cfg["abs_dua_tol"], cfg["max_iter"], cfg["check_termination"],
                      cfg["en_state_bound"], cfg["en_input_bound"])
    g.set_x0(cases["x0"])
    g.set_x_ref(cases["Xref"])
    g.set_u_ref(cases["Uref"])
    return g


@pytest.mark.parametrize("kw", [dict(devices=[0]), dict(devices=[0, 0], interleaved=True), dict(devices=[0, 0], interleaved=False),
                                dict(devices=[0, 0, 0], interleaved=True)])
def test_group_equals_unsharded_batch(kw):
    B = 1021
    suite = sc.tracking_random_suite(B=B, seed=515)              # divergent iteration counts (config-3 recipe)
    ret, ref, st, stats = _unsharded(suite)
    g = _group(suite, **kw)
    assert g.shards == len(kw["devices"])
    assert g.uses_rccl() == (len(kw["devices"]) == 1)            # one rank per device -> RCCL; shared device -> host exchange
    seen = np.concatenate([g.shard_indices(k) for k in range(g.shards)])
    assert sorted(seen.tolist()) == list(range(B))
    if kw.get("interleaved") and g.shards > 1:
        assert g.shard_indices(1)[:3].tolist() == [1, 1 + g.shards, 1 + 2 * g.shards]
    assert g.solve() == ret
    for f in FIELDS:
        assert np.array_equal(g.get(f), ref[f]), f                # bit-identical, caller order
    gs = g.status()
    assert np.array_equal(gs["iter"], st["iter"]) and np.array_equal(gs["solved"], st["solved"]) and np.array_equal(gs["status"], st["status"])
    for c, k in enumerate(("primal_residual_state", "primal_residual_input", "dual_residual_state", "dual_residual_input")):
        assert np.array_equal(gs["residuals"][:, c], st[k])
    assert np.array_equal(g.allreduce_stats(), stats)
    # a second, warm solve: the exchange is repeatable and the accumulated counters keep adding up
    g.solve()
    st2 = g.allreduce_stats()
    assert st2[7] >= stats[7] and st2[2] == B
    g.close()


def test_group_closed_loop_options_reach_every_shard():
    prob, extra = tm.load_problem("quadrotor_20hz")
    h = extra["hover"]
    nx, nu, N, B = prob["nx"], prob["nu"], prob["N"], 64
    xref = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
    totals = []
    for kw in (dict(devices=[0]), dict(devices=[0, 0], interleaved=True)):
        g = tm.TinyGroupSolver.from_problem(prob, B, **kw)
        g.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]), np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
        g.update_settings(max_iter=h["max_iter"])
        g.set_option("advance_x0", 1)
        g.set_option("steps_per_launch", 20)
        g.set_x_ref(xref, broadcast=True)
        g.set_x0(np.array(h["x0"], dtype=np.float64), broadcast=True)
        for _ in range(5):
            g.solve_async()
        st = g.allreduce_stats()
        totals.append(st)
        assert st[7] == 882 * B and st[2] == B                   # the reference episode's iteration total on every instance
        g.close()
    assert np.array_equal(totals[0], totals[1])


@pytest.mark.parametrize("exe,args", [("multi_gpu_group", ["4096"]), ("multi_gpu_rccl", ["4096"])])
def test_c_examples_run(exe, args):
    """examples/multi_gpu_group.c (TinyGroup, RCCL inside the library) and examples/multi_gpu_rccl.c (a communicator the
    caller owns, linked against librccl) on however many GPUs the box has."""
    import subprocess
    path = os.path.join(ROOT, "examples", "_build", exe)
    if not os.path.exists(path):
        pytest.fail("examples/_build/%s is missing: __graft_entry__.build() produces it (the (e) row must not pass by skipping)" % exe)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([path] + args, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "solves converged" in p.stdout
    # EVERY visible device takes part (one today; on an 8-GPU node the same test runs 8 ranks / 8 shards): the line names the count
    gpus = tm.device_count()
    head = "%d %s" % (gpus, "GPU(s)" if exe == "multi_gpu_group" else "rank(s)")
    assert any(ln.startswith(head) for ln in p.stdout.splitlines()), p.stdout[:400]
    if exe == "multi_gpu_group":
        assert "RCCL" in p.stdout


def test_random_problems_sharded_equal_unsharded():
    """random shapes (one-row, tile and coverage kernels), settings, cones and warm states (tools/fuzz_parity.random_suite):
    a group of 2-3 shards on the one GPU must reproduce the single batch bit for bit, whatever kernel the shape selects"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    from hip_runner import run_cases_hip, IN_FIELDS, OUT_FIELDS
    done = 0
    for seed in range(300, 400):
        suite, kw = fuzz_parity.random_suite(seed)
        cfg, cases, prob = suite["config"], suite["cases"], suite["problem"]
        if any(cfg.get(k) for k in ("en_state_linear", "en_input_linear", "en_tv_state_linear", "en_tv_input_linear")):
            continue                                         # (the ctypes mirror of the group has no half-space setter)
        B = cases["x0"].shape[0]
        if B < 3:
            continue
        ref = run_cases_hip(suite)
        rng = np.random.default_rng(seed)
        n = int(rng.integers(2, 4))
        g = tm.TinyGroupSolver.from_problem(prob, B, devices=[0] * n, interleaved=bool(rng.integers(0, 2)))
        g.set_bound_constraints(cfg["x_min"], cfg["x_max"], cfg["u_min"], cfg["u_max"])
        sc_, ic_ = cfg.get("state_cone"), cfg.get("input_cone")
        if sc_ is not None or ic_ is not None:
            sc_ = sc_ or ([], [], [])
            ic_ = ic_ or ([], [], [])
            g.set_cone_constraints(sc_[0], sc_[1], sc_[2], ic_[0], ic_[1], ic_[2])
        g.update_settings(cfg["abs_pri_tol"], cfg["abs_dua_tol"], cfg["max_iter"], cfg["check_termination"], cfg["en_state_bound"],
                          cfg["en_input_bound"], cfg["en_state_soc"], cfg["en_input_soc"])
        g.set_x0(cases["x0"])
        for f in IN_FIELDS:
            if f in cases:
                g.set(f, cases[f])
        g.solve()
        st = g.status()
        assert np.array_equal(st["iter"], ref["iter"].astype(int)) and np.array_equal(st["solved"], ref["sol_solved"].astype(int)), seed
        for f in OUT_FIELDS:
            assert np.array_equal(g.get(f), ref[f]), (seed, f)
        g.close()
        done += 1
        if done >= 25:
            break
    assert done >= 15

