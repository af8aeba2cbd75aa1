"""CPU: the cost model behind the automatic split solve ("repack_after" = -1; tiny_predict_split, host arithmetic).
Measured on one MI355X (profiles/r02_sweep_auto_split.md, r02_configs_3_4.json): the config-3 distribution (mode 9, tail to
100) gains 16-19 % at K = 9 ... 16; a batch of identical instances, or one whose counts sit in a narrow band, gains nothing."""
import numpy as np

import tinympc_amd as tm


def _hist(counts):
    h = np.zeros(1024, dtype=np.int64)
    for it, c in counts.items():
        h[it] += c
    return h


def test_config3_distribution_gets_a_small_cap():
    # iteration histogram of BASELINE config 3 (262 144 cold tracking solves), rounded
    h = _hist({7: 9000, 8: 61000, 9: 98000, 10: 58000, 11: 14000, 12: 5000, 14: 3000, 18: 3000, 25: 3500, 40: 3500, 70: 3000, 100: 1144})
    k, ratio = tm.predict_split(h, 12, 4, 10, max_iter=100)
    assert 6 <= k <= 24 and 0.75 < ratio < 0.95, (k, ratio)


def test_uniform_and_narrow_distributions_do_not_split():
    assert tm.predict_split(_hist({9: 262144}), 12, 4, 10, 100)[0] == 0
    assert tm.predict_split(_hist({100: 65536}), 12, 4, 10, 100)[0] == 0
    k, ratio = tm.predict_split(_hist({104: 20000, 108: 60000, 110: 90000, 113: 60000, 118: 32144}), 12, 4, 10, 500)
    assert k == 0 or ratio > 0.9, (k, ratio)


def test_the_cap_respects_check_termination_and_small_batches_gain_less():
    h = _hist({7: 9000, 8: 61000, 9: 98000, 10: 58000, 11: 14000, 12: 5000, 14: 3000, 18: 3000, 25: 3500, 40: 3500, 70: 3000, 100: 1144})
    k, _ = tm.predict_split(h, 12, 4, 10, max_iter=100, check_termination=5)
    assert k % 5 == 0
    small = (h // 32).astype(np.int64)                        # 8 192 instances: the follow-up stages cannot fill the wave slots
    _, r_small = tm.predict_split(small, 12, 4, 10, 100)
    _, r_big = tm.predict_split(h, 12, 4, 10, 100)
    assert r_small > r_big


def test_the_cap_sits_one_check_interval_behind_the_mode():
    """Measured on BASELINE config 3 (profiles/r03_config3_split_schedule.txt): iteration counts 8 / 9 for 94 % of 262 144 instances, a
    thin tail to 100.  K = 10 ... 14 all cost the same, K = 9 costs 6 % more and K = 8 60 % -- the histogram is the LAST solve's, so the
    model's argmin (9) moves one check interval up when that is predicted to cost under 1 %."""
    h = np.zeros(1024, dtype=np.int64)
    h[7:21] = [277, 99663, 147420, 1360, 79, 95, 103, 143, 154, 197, 229, 271, 324, 397]
    h[21:100] = (11432 - 855) // 79
    h[100] = 855
    k, ratio = tm.predict_split(h, 12, 4, 10, max_iter=100)
    assert k == 10 and 0.6 < ratio < 0.9, (k, ratio)
    k5, _ = tm.predict_split(h, 12, 4, 10, max_iter=100, check_termination=5)
    assert k5 % 5 == 0 and k5 >= 10


def test_step_regroup_plan_covers_every_step_once():
    """CPU: the stretches "step_regroup" cuts a fused launch into (tiny_step_regroup_plan): every step exactly once, a single step first
    when nothing is known about the instances, no stretch shorter than half of K except that first one, the second half of a
    two-stream launch half a stretch out of step"""
    for steps in (1, 2, 7, 16, 20, 24, 89, 90, 100):
        for k in (-1, 1, 3, 8, 23, 45, 200):
            for known in (False, True):
                for half in (0, 1):
                    plan = tm.step_regroup_plan(steps, k, known, half)
                    assert sum(plan) == steps and all(n >= 1 for n in plan), (steps, k, known, half, plan)
                    K = k if k > 0 else max(8, (steps + 3) // 4)
                    if K >= steps:                               # not cut at all: one launch
                        assert plan == [steps]
                        continue
                    body = plan[1:] if not known else plan
                    if not known:
                        assert plan[0] == 1
                    assert max(body) < K + (K + 1) // 2 and len(body) <= steps // K + 2, (steps, k, known, half, plan)
                    if not half and len(body) > 1:               # (only the LAST stretch may fall short of K, and never below half of it)
                        assert all(n == K for n in body[:-1]) or body[-1] >= K, (steps, k, known, half, plan)
                        assert body[-1] >= (K + 1) // 2, (steps, k, known, half, plan)
    assert tm.step_regroup_plan(90, -1, False) == [1, 23, 23, 23, 20]                  # BASELINE config 4's episode
    assert tm.step_regroup_plan(90, -1, False, half=1) == [1, 12, 23, 23, 31]
    assert tm.step_regroup_plan(90, -1, True) == [23, 23, 23, 21]
