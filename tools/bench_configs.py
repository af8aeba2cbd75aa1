"""The other BASELINE configs as entries of bench.py's one JSON line: after the timed region of config 2, rank 0 of a 1-GPU run
times -- untimed by the headline, HIP events on the solver's stream --

  config3          quadrotor_tracking (12,4,10) x 262 144, per-instance random references (SURVEY.md 8(d) recipe), ONE cold solve
                   (reference workload: examples/quadrotor_tracking.cpp:77-106)
  config4          rocket_landing (6,3,10) x 65 536, input second-order cone on, the 90-step closed loop fused into one launch
                   (examples/rocket_landing_mpc.cpp:94-135)
  config4_state_cone / config4_both_cones   the same episode with en_state_soc = 1 / both switches on (the example passes both
                   cone sets, rocket_landing_mpc.cpp:94; admm.cpp:102-122)
  sweep_*          six cells of the config-5 sweep x 131 072, one cold solve each (tools/sweep_bench.py runs all 36)

Every entry carries what its roofline fraction is made of -- frac == iters * flops_per_iter / (ms * 1e-3) / (peak * 1e12), `ms` the
MEDIAN of its timed repetitions (`ms_min` beside it) -- and leaves a SAMPLE of its own input records + what the GPU computed for
them (SAMPLE instances, evenly spaced through the batch) for the checker that runs after the GPU legs (oracle/config_check.py, in
processes of its own): `parity_sample` (iteration counts and u[:,0] against the oracle) and `cpu_baseline` (the real reference timed
on the same records) are merged into the entry by bench.py.  This module itself never touches oracle/.
The inputs are the recipes of tools/config_bench.py / tools/sweep_bench.py (same seeds); the reference / initial-state
arrays are expanded along the horizon ON the device (torch) instead of being built as GB-sized host arrays.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinympc_amd as tm  # noqa: E402

FP64_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0
SWEEP_CELLS = ((4, 2, 10), (12, 4, 30), (4, 2, 50), (12, 8, 30), (20, 8, 10), (20, 8, 50))
SAMPLE = 256


def _traffic(name):
    """HBM bytes per launch of this entry's kernel from a committed rocprofv3 PMC run, when profiles/ holds one"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("configs", {}).get(name)
        return t
    except Exception:                                   # noqa: BLE001
        return None


def _entry(name, workload, ms_all, solves, iters, nx, nu, N, bytes_per_solve, kernel, **extra):
    fl = tm.flops_per_iter(nx, nu, N)
    ms = float(np.median(ms_all))
    t = ms * 1e-3
    tf = iters * fl / t / 1e12
    e = dict(workload=workload, kernel=kernel, ms=ms, ms_min=float(np.min(ms_all)), ms_max=float(np.max(ms_all)), timed_repetitions=len(ms_all),
             solves=int(solves), iters=int(iters), solves_per_s=solves / t, iters_per_s=iters / t,
             iters_per_solve=iters / solves,
             roofline=dict(bound="fp64-valu", frac=tf / FP64_PEAK_TFLOPS, achieved=tf, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                           flops_per_iter=fl, flops=iters * fl,
                           note="frac = iters x flops_per_iter / (ms x 1e-3) / (peak x 1e12), ms = the median repetition; flops_per_iter = SURVEY.md 8 footnote 1 (box iteration)"),
             hbm=dict(algorithmic_bytes_per_solve=int(bytes_per_solve), gbs=bytes_per_solve * solves / t / 1e9,
                      frac_formula=bytes_per_solve * solves / t / 1e9 / HBM_PEAK_GBS))
    tr = _traffic(name)
    if tr is not None:
        e["traffic"] = tr
    e.update(extra)
    return e


def _cold_solves(s, n):
    """n cold solves of the batch as it stands (reset -> solve), kernel time of each from HIP events on the solver's stream"""
    ms = []
    for _ in range(n):
        s.reset()
        s.set_option("timing", 1)
        s.solve_async()
        ms.append(float(np.sum(s.timing_ms())))
    return ms


_BUSY = {}


def _busy(device=0, ms=15.0):
    """keeps the GPU computing for ~`ms` on ANOTHER handle right before a first call is timed: a first call is then measured at the
    clocks a busy server runs at, not at those the GPU fell to while the host was building and uploading this entry's records"""
    s = _BUSY.get(device)
    if s is None:
        prob, _ = tm.load_problem("quadrotor_20hz")
        s = tm.TinyBatchSolver.from_problem(prob, 16384, device=device)
        s.update_settings(max_iter=100)
        s.set_option("advance_x0", 1)
        s.set_option("steps_per_launch", 20)
        s.set_x0(np.tile(np.array(prob.get("x0", np.zeros(prob["nx"]))).reshape(1, -1) + 0.3, (16384, 1)))
        _BUSY[device] = s
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        s.reset()
        s.solve_async()
        s.synchronize()


def _sample_idx(B):
    return np.unique(np.linspace(0, B - 1, SAMPLE).astype(np.int64))


def _prob_plain(prob):
    return {k: (np.asarray(v) if isinstance(v, (list, np.ndarray)) else v) for k, v in prob.items()}


def config3(B=262144, device=0):
    prob, extra = tm.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    traj = np.array(extra["y_axis_line"])
    rng = np.random.default_rng(20260923)
    k = rng.integers(0, 291, B)
    Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
    Uref = rng.normal(0, 0.05, (B, nu, N - 1))
    x0 = Xref[:, :, 0].copy()
    x0[:, :3] += rng.normal(0, 0.1, (B, 3))
    s = tm.TinyBatchSolver.from_problem(prob, B, device=device)
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=100)
    s.set_x_ref(Xref)
    s.set_u_ref(Uref)
    s.set_x0(x0)
    # a FRESH handle under the library's defaults first: its first call is what a caller's first solve costs (with the shipped plan of
    # tinympc_amd/data/plans.txt: the settled launch form at once; without: plain / split probes, stage schedule, tile alternative)
    _busy(device)
    auto = _cold_solves(s, 17)
    st2 = s.reduce_stats()
    shipped = int(s.get_option("plan_shipped"))
    plan = s.get_plan()
    ak, ag, av = s.get_option("auto_split_k"), s.get_option("auto_split_growth"), s.get_option("auto_split_verdict")
    s.set_option("repack_after", 0)                   # the plain launch
    plain = _cold_solves(s, 3)
    st = s.reduce_stats()
    assert st2[0] == st[0] and st2[1] == st[1], "the split solve must reproduce the plain one"
    settled = auto[6:]
    e = _entry("config3", "quadrotor_tracking (12,4,10) x %d, per-instance random Xref/Uref, one cold solve (BASELINE configs[2])" % B,
               settled, B, st[0], nx, nu, N, s.algorithmic_bytes(), s.kernel_path(), solves_launched=(3 + 17) * B,
               launch_form="the library's default dispatch (automatic split solve, clock-checked), repetitions 7-17 of 17",
               plain_launch_ms=float(np.median(plain)), plain_launch_ms_min=float(np.min(plain)),
               automatic_split_k=ak, automatic_split_growth=ag, automatic_split_verdict=av, solved_fraction=st[1] / B)
    # what a caller's FIRST solves of a batch cost (probes: plain / split / other stage schedule / tile alternative), and what the
    # first solve of a fresh handle costs when it imports the settled plan (tiny_batch_get_plan / tiny_batch_set_plan)
    e["unsettled_ms"] = [float(v) for v in auto[:6]]
    e["first_call_ms"] = float(auto[0])
    e["plan_shipped"] = shipped                        # 1: the handle took tinympc_amd/data/plans.txt's entry at its first solve
    try:
        s2 = tm.TinyBatchSolver.from_problem(prob, B, device=device)
        s2.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
        s2.update_settings(max_iter=100)
        s2.set_x_ref(Xref)
        s2.set_u_ref(Uref)
        s2.set_x0(x0)
        s2.set_plan(plan)
        _busy(device)
        planned = _cold_solves(s2, 3)
        st3 = s2.reduce_stats()
        assert st3[0] == st[0] and st3[1] == st[1], "a solve under an imported plan must reproduce the plain one"
        e["planned_first_call_ms"] = float(planned[0])
        e["planned_ms"] = [float(v) for v in planned]
        e["plan"] = {k: v for k, v in tm.TinyBatchSolver.plan_fields(plan).items() if k in ("open_questions", "auto_verdict", "auto_cap", "auto_growth", "tile_verdict")}
        s2.close()
    except Exception as ex:                            # noqa: BLE001
        e["planned_first_call_ms"] = None
        e["plan_error"] = repr(ex)
    idx = _sample_idx(B)
    stt = s.status()
    it = np.where(stt["solved"][idx] != 0, stt["iter"][idx], -stt["iter"][idx])
    spec = dict(name="config3", kind="single", problem=_prob_plain(prob),
                cfg_kw=dict(max_iter=100, x_min=np.full((nx, 1), -5.0), x_max=np.full((nx, 1), 5.0), u_min=np.full((nu, 1), -0.5), u_max=np.full((nu, 1), 0.5)),
                x0=x0[idx], Xref=Xref[idx], Uref=Uref[idx], gpu_iter=it.astype(np.int32), gpu_u0=s.get("u")[idx][:, :, 0])
    s.close()
    return e, spec


def config4(B=65536, device=0, en_state_soc=0, en_input_soc=1, name="config4"):
    prob, extra = tm.load_problem("rocket_landing_20hz")
    m = extra["mpc"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(20260923)
    x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
    xinit, xg = np.array(m["xinit"], dtype=float), np.array(m["xg"], dtype=float)
    traj = np.stack([xinit + (xg - xinit) * float(i) / (m["NTOTAL"] - 1) for i in range(m["NTOTAL"])])
    s = tm.TinyBatchSolver.from_problem(prob, B, device=device)
    s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
    s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"],
                           m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
    s.update_settings(abs_pri_tol=m["abs_pri_tol"], max_iter=m["max_iter"], en_state_soc=en_state_soc, en_input_soc=en_input_soc)
    uref = np.zeros((nu, N - 1)); uref[2, :] = m["uref_z"]
    steps = m["NTOTAL"] - N
    s.set_option("advance_x0", 1)
    s.set_option("steps_per_launch", steps)

    def episode(log):
        s.reset()
        s.set_u_ref(uref, broadcast=True)
        s.set_reference_trajectory(traj)              # examples/rocket_landing_mpc.cpp:111-113 (window k .. k+N-1)
        s.set_x0(x0)
        s.set_option("step_log", log)
        s.set_option("timing", 1)
        s.solve_async()
        return float(np.sum(s.timing_ms()))
    # the uncut launch first (option "step_regroup" = 0), then the library's default dispatch (-1): its first episode is still ONE
    # launch, whose iteration totals tell the library what lock step costs this batch; the episodes after it run in stretches if
    # that pays -- those are the timed ones
    # (round 6: the default dispatch FIRST, on the fresh handle -- with the shipped plan of tinympc_amd/data/plans.txt its first episode
    # already runs in stretches; the uncut launch is timed afterwards)
    _busy(device)
    unsettled = [episode(0) for _ in range(2)]         # (without a plan the first of them is still ONE launch: it is what tells the library what lock step costs)
    shipped = int(s.get_option("plan_shipped"))
    ms, st = [], None
    for _ in range(5):
        ms.append(episode(0))
        st = s.reduce_stats()
    regroup = dict(verdict=s.get_option("step_regroup_verdict"), launches_per_episode=s.get_option("step_regroup_stretches"),
                   lockstep_estimate=s.get_option("lockstep_permille") / 1000.0,
                   note="verdict 1: the 90-step launch runs as stretches of MPC steps over the instances ordered by their last iteration count "
                        "(batch_dispatch.hip step_regroup; bit-identical to the uncut launch); lockstep_estimate = rows x the largest "
                        "iteration total of every wave / the totals, from the uncut episode")
    s.set_option("step_regroup", 0)
    plain = [episode(0) for _ in range(2)]
    s.set_option("step_regroup", -1)
    S = nx * N + nu * (N - 1)
    cones = "input second-order cone on" if (en_input_soc and not en_state_soc) else ("state second-order cone on" if not en_input_soc else "state AND input second-order cones on")
    e = _entry(name, "rocket_landing (6,3,10) x %d, %s, %d-step closed loop fused into one launch (BASELINE configs[3])" % (B, cones, steps),
               ms, B * steps, st[7], nx, nu, N, s.algorithmic_bytes() + 8 * 3 * nu * (N - 1), s.kernel_path(),
               solved_fraction=st[8] / (B * steps), mpc_steps_per_launch=steps, en_state_soc=en_state_soc, en_input_soc=en_input_soc,
               solves_launched=9 * B * steps, plain_launch_ms=float(np.median(plain)), step_regroup=regroup,
               launch_form="the library's default dispatch (automatic step_regroup), episodes 3-7 of 9 on a fresh handle (8-9: the uncut launch, option off)",
               note="flops_per_iter counts the box iteration only (the cone projection's sqrt / divisions are extra work, not extra credit); "
                    "bytes: bytes_warm + the cone slack records, once per LAUNCH (S = %d)" % S)
    e["first_call_ms"] = float(unsettled[0])           # what the default dispatch's FIRST episode costs (before it knows the batch)
    e["plan_shipped"] = shipped
    e["unsettled_ms"] = [float(v) for v in unsettled]
    e["hbm"]["gbs"] /= steps                           # the records move once per launch, not once per fused step
    e["hbm"]["frac_formula"] /= steps
    episode(1)                                         # untimed: the same episode once more with the per-step log on, for the checker
    idx = _sample_idx(B)
    it, u0 = s.step_log(steps)
    spec = dict(name=name, kind="episode", problem=_prob_plain(prob), steps=steps, traj=traj,
                cfg_kw=dict(max_iter=m["max_iter"], abs_pri_tol=m["abs_pri_tol"], x_min=np.array(m["x_min"]), x_max=np.array(m["x_max"]),
                            u_min=np.full((nu, 1), m["u_min"]), u_max=np.full((nu, 1), m["u_max"]), en_state_soc=en_state_soc, en_input_soc=en_input_soc,
                            state_cone=(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"]),
                            input_cone=(m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])),
                x0=x0[idx], Xref=np.zeros((nx, N)), Uref=uref, gpu_iter=it[:, idx].astype(np.int32), gpu_u0=u0[:, idx, :])
    s.close()
    return e, spec


def sweep_cell(nx, nu, N, B=131072, device=0):
    import torch
    prob, rng = tm.random_problem(nx, nu, N)
    s = tm.TinyBatchSolver.from_problem(prob, B, device=device)
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    x0 = rng.uniform(-1, 1, (B, nx))                  # the same draws as tools/sweep_bench.py
    xr = rng.uniform(-0.2, 0.2, (B, nx, 1))
    s.set_x0(x0)
    # Xref = the instance's vector replicated along the horizon: expanded on the device into the [batch][cols][rows] layout
    dev = torch.device("cuda", device)
    xr_d = torch.from_numpy(np.ascontiguousarray(xr[:, :, 0])).to(dev)[:, None, :].expand(B, N, nx).contiguous()
    torch.cuda.synchronize(dev)
    s.set_device("Xref", xr_d.data_ptr())
    s.synchronize()
    del xr_d
    # (a shape the one-row kernel holds settles its launch form over the first solves: plain, split, the tile kernel's dynamic form)
    tile = s.kernel_path() == "tile"
    _busy(device)
    ms = _cold_solves(s, 4 if tile else 11)
    settled = ms[1:] if tile else ms[6:]
    st = s.reduce_stats()
    name = "sweep_%d_%d_%d" % (nx, nu, N)
    e = _entry(name, "random sweep cell (nx=%d, nu=%d, N=%d) x %d, one cold solve, max_iter 500 (BASELINE configs[4])" % (nx, nu, N, B),
               settled, B, st[0], nx, nu, N, s.algorithmic_bytes(), s.kernel_path(), solved_fraction=st[1] / B, solves_launched=len(ms) * B,
               automatic_split_k=s.get_option("auto_split_k"), tile_alt_verdict=s.get_option("tile_alt_verdict"))
    e["first_call_ms"] = float(ms[0])                  # a caller's first solve of the batch (one-row shapes: the first of the probe solves)
    e["plan_shipped"] = int(s.get_option("plan_shipped"))
    e["unsettled_ms"] = [float(v) for v in (ms[:1] if tile else ms[:6])]
    idx = _sample_idx(B)
    stt = s.status()
    it = np.where(stt["solved"][idx] != 0, stt["iter"][idx], -stt["iter"][idx])
    spec = dict(name=name, kind="single", problem=_prob_plain(prob),
                cfg_kw=dict(max_iter=500, u_min=np.full((nu, 1), -0.5), u_max=np.full((nu, 1), 0.5)),
                x0=x0[idx], Xref=np.tile(xr[idx], (1, 1, N)), Uref=np.zeros((nu, N - 1)), gpu_iter=it.astype(np.int32), gpu_u0=s.get("u")[idx][:, :, 0])
    s.close()
    return e, spec


def hetero_cell(nx=20, nu=8, N=10, B=32768, device=0):
    """per-instance problem data (tiny_api.cpp:307-381 once per INSTANCE: every instance its own A, B, rho and therefore its own cache,
    computed by the batched Riccati kernel) on a wide shape: the tile kernel's per-instance form.  The checker solves every sampled
    instance with ITS OWN oracle."""
    fam, rng = tm.random_problem(nx, nu, N)
    A0, B0 = np.asarray(fam["A"]), np.asarray(fam["B"])
    A = A0[None] * (1 + rng.normal(0, 1e-3, (B, 1, 1)))
    Bm = B0[None] * (1 + rng.normal(0, 0.05, (B, 1, 1)))
    rho = rng.uniform(0.8, 1.2, B) * fam["rho"]
    s = tm.TinyBatchSolver.hetero(A, Bm, None, np.tile(fam["Q"], (B, 1)), np.tile(fam["R"], (B, 1)), rho, N, device=device)
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    x0 = rng.uniform(-1, 1, (B, nx))
    xr = rng.uniform(-0.2, 0.2, (B, nx, 1))
    s.set_x0(x0)
    s.set_x_ref(np.repeat(xr, N, axis=2))
    _busy(device)
    ms = _cold_solves(s, 4)
    st = s.reduce_stats()
    name = "hetero_%d_%d_%d" % (nx, nu, N)
    e = _entry(name, "per-instance problem data (nx=%d, nu=%d, N=%d) x %d: every instance its own A, B, rho and cache; one cold solve, max_iter 500" % (nx, nu, N, B),
               ms[1:], B, st[0], nx, nu, N, s.algorithmic_bytes(), s.kernel_path(), solved_fraction=st[1] / B, solves_launched=len(ms) * B)
    e["first_call_ms"] = float(ms[0])
    idx = _sample_idx(B)
    stt = s.status()
    it = np.where(stt["solved"][idx] != 0, stt["iter"][idx], -stt["iter"][idx])
    base = _prob_plain(fam)
    spec = dict(name=name, kind="single", problem=base, problems=[dict(base, A=A[i], B=Bm[i], rho=float(rho[i])) for i in idx],
                cfg_kw=dict(max_iter=500, u_min=np.full((nu, 1), -0.5), u_max=np.full((nu, 1), 0.5)),
                x0=x0[idx], Xref=np.tile(xr[idx], (1, 1, N)), Uref=np.zeros((nu, N - 1)), gpu_iter=it.astype(np.int32), gpu_u0=s.get("u")[idx][:, :, 0])
    s.close()
    return e, spec


def tracking_cell(nx=12, nu=8, N=30, B=32768, T=10, device=0):
    """a closed-loop tracking episode as examples/quadrotor_tracking.cpp:77-106 runs it -- every MPC step: the reference window moves one
    knot, the duals are reset, solve, plant step -- fused into ONE launch on a long shape (the tile kernel's reference-window form)."""
    fam, rng = tm.random_problem(nx, nu, N)
    x0 = rng.uniform(-1, 1, (B, nx))
    traj = rng.normal(0, 0.2, (N + T + 2, nx))
    s = tm.TinyBatchSolver.from_problem(fam, B, device=device)
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=100)
    s.set_option("advance_x0", 1)
    s.set_option("steps_per_launch", T)
    s.set_reference_trajectory(traj)
    s.set_option("reset_duals", 1)

    def episode(log):
        s.reset()
        s.set_x0(x0)
        s.set_option("traj_step", 0)
        s.set_option("step_log", log)
        s.set_option("timing", 1)
        s.solve_async()
        return float(np.sum(s.timing_ms()))
    _busy(device)
    ms = [episode(0) for _ in range(4)]
    st = s.reduce_stats()
    name = "tracking_%d_%d_%d" % (nx, nu, N)
    e = _entry(name, "tracking episode (nx=%d, nu=%d, N=%d) x %d: %d closed-loop MPC steps fused into one launch, moving reference window + reset duals "
               "(examples/quadrotor_tracking.cpp:77-106)" % (nx, nu, N, B, T),
               ms[1:], B * T, st[7], nx, nu, N, s.algorithmic_bytes(), s.kernel_path(), solved_fraction=st[8] / (B * T), mpc_steps_per_launch=T,
               solves_launched=5 * B * T)
    e["first_call_ms"] = float(ms[0])
    e["hbm"]["gbs"] /= T
    e["hbm"]["frac_formula"] /= T
    episode(1)
    idx = _sample_idx(B)
    it, u0 = s.step_log(T)
    spec = dict(name=name, kind="tracking", problem=_prob_plain(fam), steps=T, traj=traj,
                cfg_kw=dict(max_iter=100, u_min=np.full((nu, 1), -0.5), u_max=np.full((nu, 1), 0.5)),
                x0=x0[idx], Xref=np.zeros((nx, N)), Uref=np.zeros((nu, N - 1)), gpu_iter=it[:, idx].astype(np.int32), gpu_u0=u0[:, idx, :])
    s.close()
    return e, spec


def run_all(device=0, budget_s=45.0, log=None, spec_out=None, only=None):
    """-> {"config3": ..., "config4": ..., "sweep_4_2_10": ...}; an entry that fails is reported as {"error": ...}, entries that
    would start after the budget is spent as {"skipped": ...} (the bench line must appear whatever happens here).  spec_out: where
    the samples for oracle/config_check.py are pickled."""
    import pickle
    out, specs, t0 = {}, [], time.perf_counter()
    jobs = [("config3", lambda: config3(device=device)), ("config4", lambda: config4(device=device)),
            ("config4_state_cone", lambda: config4(device=device, en_state_soc=1, en_input_soc=0, name="config4_state_cone")),
            ("config4_both_cones", lambda: config4(device=device, en_state_soc=1, en_input_soc=1, name="config4_both_cones"))]
    jobs += [("sweep_%d_%d_%d" % c, (lambda c=c: sweep_cell(*c, device=device))) for c in SWEEP_CELLS]
    # round 6: the round-5 launch forms at full batch -- per-instance problem data and a fused tracking episode on tile shapes
    jobs += [("hetero_20_8_10", lambda: hetero_cell(device=device)), ("tracking_12_8_30", lambda: tracking_cell(device=device))]
    for name, fn in jobs:
        if only and name not in only:
            continue
        if time.perf_counter() - t0 > budget_s:
            out[name] = {"skipped": "time budget of %.0f s spent" % budget_s}
            continue
        t1 = time.perf_counter()
        try:
            out[name], spec = fn()
            specs.append(spec)
            out[name]["wall_s_incl_setup"] = time.perf_counter() - t1
        except Exception as e:                         # noqa: BLE001
            out[name] = {"error": repr(e)}
        if log:
            log("configs: %s done in %.1f s" % (name, time.perf_counter() - t1))
    if spec_out and specs:
        with open(spec_out, "wb") as f:
            pickle.dump(specs, f)
    return out


if __name__ == "__main__":
    # python tools/bench_configs.py [entry ...]   (under rocprofv3 --pmc for tools/configs_traffic.py: one entry per run)
    print("@@CFG@@" + json.dumps(run_all(log=lambda m: print(m, file=sys.stderr), only=sys.argv[1:] or None, budget_s=600.0)))
