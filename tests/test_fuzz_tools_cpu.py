"""CPU: the closed-loop fuzzer's helper logic that needs no GPU -- a trial is a pure function of its seed, and the sensitivity
check classifies the one loop of round 1 that ended 5.8e-6 away from the oracle (seed 600878) as ill-conditioned: the oracle
itself moves further than that when its initial state is perturbed by 1e-14."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(HERE, ".."), os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..", "tools"), HERE):
    sys.path.insert(0, p)

from cpu_solvers import build_oracle  # noqa: E402


def test_trials_are_functions_of_the_seed_and_the_chaotic_loop_is_recognised():
    assert build_oracle()
    import fuzz_closed_loop as f
    a, b = f.draw(600878), f.draw(600878)
    assert (a["nx"], a["nu"], a["N"], a["T"], a["launches"]) == (8, 4, 30, 8, 2)
    assert np.array_equal(a["x0"], b["x0"]) and np.array_equal(a["fams"][0]["A"], b["fams"][0]["A"])
    ep = f.oracle_episode(a, 0)
    assert len(ep) == 16 and all(it == 29 for _, _, it in ep)              # every solve of that loop stops at max_iter
    amp = f.sensitivity(a, 0)
    assert amp > 1e-6, amp                                                 # 1e-14 in, more than the fuzzer's tolerance out
    calm = f.draw(700001)
    assert f.sensitivity(calm, 0) < 1e-9
