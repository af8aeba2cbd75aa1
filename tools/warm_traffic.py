#!/usr/bin/env python3
"""HBM-side bytes of warm launches beyond the Infinity Cache: the hover workload at batch 262 144 (working set
~4x the 256 MiB L3), one launch per MPC step, per-instance reference records; steps 70-99 of the episode are the launches counted.
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` (separate passes, tools/gpu_stage.sh warm5), then
    python tools/warm_traffic.py --collect <dir>
turns the two counter files into bytes per solve next to the byte model bench.py uses (moved_bytes_per_solve)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = 262144


def run():
    import numpy as np
    import tinympc_amd as tm
    prob, extra = tm.load_problem("quadrotor_20hz")
    h = extra["hover"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]), np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
    s.update_settings(max_iter=h["max_iter"])
    s.set_option("advance_x0", 1)
    s.set_option("share_ref", 0)
    for kv in filter(None, os.environ.get("WARM_OPTS", "").split(",")):       # e.g. WARM_OPTS=prefetch=0
        k, v = kv.split("=")
        s.set_option(k, int(v))
    s.set_x_ref(np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N)), broadcast=True)
    s.set_x0(np.array(h["x0"], dtype=np.float64), broadcast=True)
    its = []
    for _ in range(100):
        s.solve_async()
        s.synchronize()
        its.append(int(s.status()["iter"][0]))
    print("@@ITERS@@" + json.dumps(its))
    s.close()


def collect(d):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(d, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
        rows = [r for r in csv.DictReader(open(f[0])) if "admm_solve_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c] if f else []
        per = {}
        for r in rows:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        vals = [per[k] for k in sorted(per, key=lambda x: int(x))]
        out[c] = vals[70:100]
    its = None
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(d, "warm_traffic_%s.out" % c)
        if os.path.exists(p):
            for ln in open(p):
                if ln.startswith("@@ITERS@@"):
                    its = json.loads(ln[9:])
    if not out.get("FETCH_SIZE") or not out.get("WRITE_SIZE") or its is None:
        print(json.dumps({"error": "counter files incomplete", "have": {k: len(v) for k, v in out.items()}}))
        return
    import bench
    S, nx, nu = 156, 12, 4
    bw = 8 * (nx + 8 * S) + 44
    model = sum(bench.moved_bytes_per_solve(bw, S, nx, nu, it, False, 1) for it in its[70:100]) / 30.0
    fetch = sum(out["FETCH_SIZE"]) / len(out["FETCH_SIZE"]) * 1024.0 / B
    write = sum(out["WRITE_SIZE"]) / len(out["WRITE_SIZE"]) * 1024.0 / B
    print(json.dumps({"batch": B, "launches_counted": "steps 70-99 of the hover episode, one launch per MPC step, per-instance reference records",
                      "fetch_bytes_per_solve_x2": 2 * fetch, "write_bytes_per_solve": write, "hbm_bytes_per_solve": 2 * fetch + write,
                      "byte_model_per_solve": model, "algorithmic_bytes_warm": bw, "ratio_measured_over_model": (2 * fetch + write) / model,
                      "note": "FETCH_SIZE x 2 as in profiles/traffic.json (8 B / lane loads); working set ~ batch x 3.7 KB x ... = ~1 GB, four times the Infinity Cache"}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--collect":
        collect(sys.argv[2])
    else:
        run()
