"""Helper of tests/test_gpu_parity.py::test_device_pointer_set_get_roundtrip (run as a script)."""
import os
import sys

import numpy as np
import torch                      # first: torch's bundled HIP runtime must be the one that initialises

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
torch.cuda.init()
import scenarios as sc            # noqa: E402
import tinympc_amd as tm          # noqa: E402
from hip_runner import make_batch, run_cases_hip   # noqa: E402

suite = sc.tracking_random_suite(B=50, seed=9)
ref = run_cases_hip(suite)
s = make_batch(suite)
dev = torch.device("cuda:0")
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 1))).to(dev)   # [B][cols][rows]
keep = {k: to_dev(suite["cases"][k]) for k in ("Xref", "Uref")}
x0 = torch.from_numpy(suite["cases"]["x0"].copy()).to(dev)
torch.cuda.synchronize()
s.set_device("x0", x0.data_ptr())
for k, t in keep.items():
    s.set_device(k, t.data_ptr())
s.synchronize()
assert s.solve() in (0, 1)
out_u = torch.zeros((50, suite["problem"]["N"] - 1, suite["problem"]["nu"]), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
assert tm.lib().tiny_batch_get(s._h, tm.FIELD_ID["u"], out_u.data_ptr(), tm.DEVICE) == 0
s.synchronize()
assert np.array_equal(out_u.cpu().numpy().transpose(0, 2, 1), ref["u"])
assert np.array_equal(s.status()["iter"], ref["iter"].astype(int))
s.close()
print("device pointer path ok")
