"""Option "step_regroup": a fused closed-loop launch of the one-row kernel cut into stretches of K MPC steps, each over the
instances ordered by the iteration count of their last solve (SolveArgs::perm; batch_dispatch.hip).  The four rows of a wave run in
lock step -- the order decides what a wave costs, never what an instance computes: every record, every per-step log entry and
every statistic must be bit-identical to the uncut launch (and, through tests/test_gpu_fused_variants.py, to single-step
launches and the oracle)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
sys.path.insert(0, HERE)

import scenarios as sc  # noqa: E402
import tinympc_amd as tm  # noqa: E402
from hip_runner import make_batch  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(HERE, "golden")


def rocket_batch(B, en_state_soc=0, en_input_soc=1, seed=5):
    """the rocket_landing_mpc loop of BASELINE configs[3] (tools/bench_configs.py config4) at a small batch, start states spread
    enough that the instances need different iteration counts"""
    prob, extra = tm.load_problem("rocket_landing_20hz")
    m = extra["mpc"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(seed)
    x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
    xinit, xg = np.array(m["xinit"], dtype=float), np.array(m["xg"], dtype=float)
    traj = np.stack([xinit + (xg - xinit) * float(i) / (m["NTOTAL"] - 1) for i in range(m["NTOTAL"])])
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
    s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"],
                           m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
    s.update_settings(abs_pri_tol=m["abs_pri_tol"], max_iter=m["max_iter"], en_state_soc=en_state_soc, en_input_soc=en_input_soc)
    uref = np.zeros((nu, N - 1)); uref[2, :] = m["uref_z"]
    s.set_u_ref(uref, broadcast=True)
    s.set_reference_trajectory(traj)
    s.set_x0(x0)
    s.set_option("advance_x0", 1)
    return s


FIELDS = ("x", "u", "vnew", "znew", "g", "y", "v", "z", "vcnew", "zcnew", "gc", "yc", "x0")


def episode(s, steps, regroup, launches=1, fields=FIELDS, streams=1):
    s.set_option("step_regroup_streams", streams)
    s.set_option("steps_per_launch", steps)
    s.set_option("step_log", 1)
    s.set_option("step_regroup", regroup)
    its, u0s, stretches = [], [], []
    for _ in range(launches):
        s.solve_async()
        it, u0 = s.step_log(steps)
        its.append(it); u0s.append(u0)
        stretches.append(s.get_option("step_regroup_stretches"))
    out = {k: s.get(k) for k in fields}
    out["it"], out["u0"] = np.concatenate(its), np.concatenate(u0s)
    st = s.status()
    out["iter"], out["solved"] = st["iter"], st["solved"]
    out["stats"] = s.reduce_stats()
    return out, stretches


@pytest.mark.parametrize("cones", [(0, 1), (1, 1)])
@pytest.mark.parametrize("K", [1, 4, 7])
@pytest.mark.parametrize("streams", [1, 2])
def test_regrouped_stretches_equal_the_uncut_launch(cones, K, streams):
    """streams = 2: the two halves of the batch on two streams, the second one half a stretch out of step (K = 1: no room for that)"""
    B, steps = 203, 24                                         # ragged last wave; 7 does not divide 24 (a short remainder joins)
    a, sa = episode(rocket_batch(B, *cones), steps, 0, launches=2)
    b, sb = episode(rocket_batch(B, *cones), steps, K, launches=2, streams=streams)
    assert sa == [1, 1]
    # first launch: nothing known yet -> a single step first, then stretches of K; second launch: sorted from its first step on
    n2 = len(range(0, steps, K)) if steps % K == 0 or steps % K >= (K + 1) // 2 else steps // K
    if streams == 1:
        assert sb[1] == n2 and sb[0] >= sb[1]
    else:
        assert sb[1] >= 2 * n2
    assert len(np.unique(np.abs(a["it"][3]))) > 2            # the instances do differ (or the order would not matter)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_regrouped_box_kernel_and_half_rows():
    """the plain box kernel skips the x|u store of the stretches that are not the last; (4,2,10) runs on the HALF form (eight
    rows per wave)"""
    for dims in ((12, 4, 10), (4, 2, 10)):
        suite = sc.sweep_suite(*dims, B=157, max_iter=60)
        outs = []
        for K in (0, 3):
            s = make_batch(suite)
            s.set_option("advance_x0", 1)
            s.set_x0(suite["cases"]["x0"]); s.set("Xref", suite["cases"]["Xref"]); s.set("Uref", suite["cases"]["Uref"])
            o, st = episode(s, 10, K, fields=("x", "u", "vnew", "znew", "g", "y", "v", "z", "x0"))
            assert s.kernel_path() == "regs"
            if dims == (4, 2, 10):
                assert s.get_option("last_half_rows") == 1
            assert st == ([1] if K == 0 else [4])             # 1 + 3 + 3 + 3
            outs.append(o)
            s.close()
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k]), (dims, k)


def test_automatic_regroup_follows_the_lock_step_estimate():
    """-1 (the default): the iteration totals of the previous fused launch decide -- rows that disagree by >= 5 % switch the
    stretches on for the launches after it; a batch of identical instances stays uncut"""
    B, steps = 4096, 90                                        # (the whole episode: the instances drift apart late)
    s = rocket_batch(B)
    a, sa = episode(s, steps, -1, launches=1)
    assert sa == [1]
    assert s.get_option("lockstep_permille") >= 1050 and s.get_option("step_regroup_verdict") == 1
    s.solve_async()
    assert s.get_option("step_regroup_stretches") > 1
    s.close()
    # identical instances: nothing to gain
    suite, _ = sc.load_suite(os.path.join(GOLDEN, "hover_warm.npz"))
    prob, extra = sc.load_problem("quadrotor_20hz")
    h = make_batch(suite, batch=B)
    h.set_x_ref(np.tile(np.array(extra["hover"]["xref"], dtype=float).reshape(-1, 1), (1, prob["N"])), broadcast=True)
    h.set_x0(np.array(extra["hover"]["x0"], dtype=float), broadcast=True)
    h.set_option("steps_per_launch", 20)
    h.solve_async()
    assert h.get_option("lockstep_permille") == 1000 and h.get_option("step_regroup_verdict") == -1
    h.solve_async()
    assert h.get_option("step_regroup_stretches") == 1
    h.close()


def test_shipped_plans_of_config_4_follow_the_cone_setting():
    """Round 6: tinympc_amd/data/plans.txt holds one entry per cone setting of BASELINE config 4 (`plan_soc <mask>`): the fused rocket
    episode runs in stretches of MPC steps with the thrust cone alone (rows of a wave disagree: lock-step estimate 1.13) and as ONE launch
    with the state cone on (1.02) -- on the FIRST episode of a fresh handle."""
    import tinympc_amd as tm
    prob, extra = tm.load_problem("rocket_landing_20hz")
    m = extra["mpc"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    B = 65536
    rng = np.random.default_rng(3)
    x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
    xinit, xg = np.array(m["xinit"], dtype=float), np.array(m["xg"], dtype=float)
    trj = np.stack([xinit + (xg - xinit) * float(i) / (m["NTOTAL"] - 1) for i in range(m["NTOTAL"])])
    got = {}
    for ss, si in ((0, 1), (1, 0)):
        s = tm.TinyBatchSolver.from_problem(prob, B)
        s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
        s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"], m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
        s.update_settings(abs_pri_tol=m["abs_pri_tol"], max_iter=m["max_iter"], en_state_soc=ss, en_input_soc=si)
        uref = np.zeros((nu, N - 1)); uref[2, :] = m["uref_z"]
        s.set_u_ref(uref, broadcast=True)
        s.set_reference_trajectory(trj)
        s.set_x0(x0)
        s.set_option("advance_x0", 1)
        s.set_option("steps_per_launch", 24)
        s.solve()
        got[(ss, si)] = (s.get_option("plan_shipped"), s.get_option("step_regroup_stretches"), s.get_option("step_regroup_verdict"))
        s.close()
    assert got[(0, 1)][0] == 1 and got[(0, 1)][2] == 1 and got[(0, 1)][1] > 1, got
    assert got[(1, 0)][0] == 1 and got[(1, 0)][2] == -1 and got[(1, 0)][1] == 1, got
