"""GPU: the multi-GPU split of SURVEY.md 8(e) on ONE device -- a divergent batch sharded round-robin
(shard_indices(interleaved=True)) over two handles must give, instance by instance, bit-identical results to the
unsharded batch, and the reduction of the shards' 64-byte statistics messages must equal the unsharded statistics."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scenarios as sc  # noqa: E402
from hip_runner import make_batch, IN_FIELDS, OUT_FIELDS  # noqa: E402
from tinympc_amd.distributed import WIRE_IDX, reduce_table, shard_indices  # noqa: E402

pytestmark = pytest.mark.gpu


def _solve(suite, idx):
    cases = {k: v[idx] for k, v in suite["cases"].items()}
    s = make_batch(dict(suite, cases=cases))
    s.set_x0(cases["x0"])
    for f in IN_FIELDS:
        if f in cases:
            s.set(f, cases[f])
    s.solve()
    out = {f: s.get(f) for f in OUT_FIELDS}
    st = s.status()
    out.update(iter=st["iter"], solved=st["solved"], status=st["status"],
               resid=np.stack([st[k] for k in ("primal_residual_state", "primal_residual_input", "dual_residual_state", "dual_residual_input")], axis=1))
    stats = s.reduce_stats()
    s.close()
    return out, stats


@pytest.mark.parametrize("world,interleaved", [(2, True), (2, False), (3, True)])
def test_sharded_batch_equals_unsharded(world, interleaved):
    B = 1021                                                     # not a multiple of the shard count or of a wave's 4 instances
    suite = sc.tracking_random_suite(B=B, seed=4242)             # config-3 recipe: iteration counts diverge
    full, full_stats = _solve(suite, np.arange(B))
    assert len(np.unique(full["iter"])) > 3
    table = []
    for r in range(world):
        idx = np.array(shard_indices(B, r, world, interleaved=interleaved))
        part, st = _solve(suite, idx)
        for k, v in part.items():
            assert np.array_equal(v, full[k][idx]), (k, r)       # bit-identical, instance by instance
        table.append(st[list(WIRE_IDX)])
    total = reduce_table(np.array(table), B).numpy()
    assert np.array_equal(total, full_stats), (total, full_stats)
