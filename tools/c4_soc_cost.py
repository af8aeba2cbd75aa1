import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np
import tinympc_amd as tm
import config_bench as cb
# config 4 with the cone switched off: the cost of an iteration without the projection
prob, extra = tm.load_problem("rocket_landing_20hz")
m = extra["mpc"]; nx, nu, N = prob["nx"], prob["nu"], prob["N"]; B = 65536
rng = np.random.default_rng(20260923)
x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
xinit, xg = np.array(m["xinit"], dtype=float), np.array(m["xg"], dtype=float)
traj = np.stack([xinit + (xg - xinit) * float(i) / (m["NTOTAL"] - 1) for i in range(m["NTOTAL"])])
for soc in (0, 1):
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
    s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"], m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
    s.update_settings(abs_pri_tol=m["abs_pri_tol"], max_iter=m["max_iter"], en_input_soc=soc)
    uref = np.zeros((nu, N - 1)); uref[2, :] = m["uref_z"]
    steps = m["NTOTAL"] - N
    s.reset(); s.set_u_ref(uref, broadcast=True); s.set_reference_trajectory(traj); s.set_x0(x0)
    s.set_option("advance_x0", 1); s.set_option("steps_per_launch", steps); s.synchronize()
    t0 = time.perf_counter(); s.solve_async(); s.synchronize(); dt = time.perf_counter() - t0
    st = s.reduce_stats()
    print(f"en_input_soc={soc}: {st[7] / dt:.3e} ADMM it/s, {st[7] / (B * steps):.1f} it/solve, {B * steps / dt:.3e} solves/s")
    s.close()
