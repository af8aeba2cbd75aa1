"""CPU: bench.py's launch-shape arithmetic (the timed region must be EXACTLY --steps MPC steps in whole launches) and its
refusal to run without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_steps_per_launch_divides_the_timed_steps():
    import bench
    assert bench.steps_per_launch(100, 100) == 100          # the default: the whole reference episode in one launch
    assert bench.steps_per_launch(100, 10) == 100           # (the warm-up is cut into launches of its own)
    assert bench.steps_per_launch(20, 5) == 20              # the driver's flags: the 20 timed steps in ONE launch
    assert bench.steps_per_launch(7, 3) == 7
    assert bench.steps_per_launch(50, 0) == 50
    assert bench.steps_per_launch(1000, 1000) == 100        # never more than 100 steps per launch
    assert bench.steps_per_launch(300, 200) == 100
    assert bench.steps_per_launch(100, 100, 1) == 1
    assert bench.steps_per_launch(100, 10, 20) == 20
    assert bench.steps_per_launch(100, 10, 30) is None      # 30 does not divide the timed steps
    assert bench.steps_per_launch(5) == 5 and bench.steps_per_launch(101) == 1 and bench.steps_per_launch(202) == 2
    for steps in range(1, 230):
        T = bench.steps_per_launch(steps, 7)
        assert T and 1 <= T <= 100 and steps % T == 0


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)
    assert p.stdout.strip() == ""                             # stdout is reserved for the one result line
