"""The driver's contract for bench.py, checked on bench.py itself: `--gpus N` really starts N ranks (CPU, gloo), the like-for-like CPU
column times the window it says it times (tiny fixture, CPU), `check_line` -- the assertions bench.py runs on its own line before
printing -- holds on the committed lines of the last profile run, and (GPU) on a line produced live."""
from gcs_admm_amd import IPM_TOL
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run_bench(*argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, BENCH, *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_gpus_argument_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two fresh rank processes, each seeing WORLD_SIZE = 2 (the driver's
    command form; the fan-out this replaces is SolveInParallel at admm_solver_v3.py:490)"""
    rc, d, err = _run_bench("--gpus", "2", "--dry-launch")
    assert rc == 0, err[-2000:]
    assert d["dry_launch"] is True and d["n_gpus"] == 2 and d["gpus_argument"] == 2
    assert sorted(r["rank"] for r in d["ranks"]) == [0, 1] and all(r["world_size"] == 2 for r in d["ranks"])
    assert sorted(r["local_rank"] for r in d["ranks"]) == [0, 1]


def test_single_rank_dry_launch_stays_in_process():
    rc, d, err = _run_bench("--dry-launch")
    assert rc == 0 and d["n_gpus"] == 1 and d["ranks"] == [{"rank": 0, "world_size": 1, "local_rank": 0}], err[-2000:]


def test_more_ranks_than_devices_is_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    rc, d, err = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert rc == 2 and d is None and "device(s) visible" in err


def test_cpu_baseline_times_the_window_it_names(oracle_lib):
    """the like-for-like CPU column: advanced to the same first iteration, the same K iterations timed, and the continued run equals
    an uninterrupted one"""
    import numpy as np
    import bench
    from gcs_admm_amd.cases import load_fixture
    from oracle.oracle import Oracle
    _, g = load_fixture("benchmark1")
    first, warmup, steps = 6, 2, 5
    c = bench.cpu_baseline(g, "benchmark1", first, warmup, steps, seconds=0.5)
    assert c["window"] == {"first_iteration": first + warmup + 1, "last_iteration": first + warmup + steps}
    assert c["value"] > 0 and c["repetitions"] >= 1 and c["kind"] == "port" and c["cores"] >= 1 and str(c["cores"]) in c["thread_sweep"]
    # oracle_admm_run_from continues a run exactly
    a = Oracle(g, ipm_tol=IPM_TOL); ra = a.run(max_it=first + warmup + steps, eps_abs=0.0, eps_rel=0.0)
    b = Oracle(g, ipm_tol=IPM_TOL)
    _, _, rho, t1 = b.run_from(1, first + warmup, eps_abs=0.0, eps_rel=0.0)
    snap = b.snapshot()
    it, _, _, t2 = b.run_from(first + warmup + 1, first + warmup + steps, rho=rho, eps_abs=0.0, eps_rel=0.0)
    assert it == first + warmup + steps + 1
    assert np.array_equal(np.vstack([t1, t2]), ra["trace"])
    b.restore(snap)
    _, _, _, t3 = b.run_from(first + warmup + 1, first + warmup + steps, rho=rho, eps_abs=0.0, eps_rel=0.0)
    assert np.array_equal(t2, t3)


def test_window_is_centred_on_the_reference_run():
    import bench
    for w, k in ((5, 20), (20, 200), (0, 1)):
        first = bench.window_start("benchmark4", w, k)
        assert first >= 0 and abs((first + 1) + (first + w + k) - (1 + 465)) <= 1      # the window's midpoint is the run's


def _committed_lines():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04", "bench_default*.json")))
    return [json.loads(open(f).read().strip().splitlines()[-1]) for f in files]


def test_committed_default_lines_hold_the_contract():
    import bench
    lines = _committed_lines()
    if not lines:
        pytest.skip("no bench line of this round committed yet")
    for d in lines:
        assert bench.check_line(d)
        assert d["cpu_baseline"]["window"] == {k: d["config"]["window"][k] for k in ("first_iteration", "last_iteration")}
        conv = d["convergence"]
        assert conv["iterations_to_stop"] == conv["reference_iterations"] == 465 and conv["trace_within_reference_tolerance"] is True
        assert conv["inner_failures"] == 0 and d["value_to_stop"] > 0 and d["cpu_baseline"]["to_stop"]["iterations"] == 465


def test_check_line_rejects_a_broken_line():
    import bench
    base = {"metric": "admm_iterations_per_sec", "value": 1000.0, "unit": "iterations/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "x",
            "config": {"workload": "benchmark4", "window": {"first_iteration": 226, "last_iteration": 245}},
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achieved": 8.0, "frac": 1e-3, "traffic": None, "avg_launch_ms": 0.9},
            "cpu_baseline": {"kind": "port", "cores": 8, "value": 10.0, "unit": "iterations/s", "sample": "s", "window": {"first_iteration": 226, "last_iteration": 245}}}
    assert bench.check_line(base)
    for breaker in (lambda d: d["roofline"].update(avg_launch_ms=1.1), lambda d: d["cpu_baseline"]["window"].update(first_iteration=1),
                    lambda d: d.update(value=999.0), lambda d: d.pop("warmup"), lambda d: d.update(n_gpus=2)):
        bad = json.loads(json.dumps(base))
        breaker(bad)
        with pytest.raises((AssertionError, KeyError)):
            bench.check_line(bad)


@pytest.mark.gpu
def test_live_bench_line_holds_the_contract():
    """the driver's command, live (headline only): bench.py asserts check_line itself before printing; here once more from outside"""
    import bench
    rc, d, err = _run_bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-configs", timeout=900)
    assert rc == 0, err[-3000:]
    assert bench.check_line(d) and d["n_gpus"] == 1 and d["config"]["workload"] == "benchmark4"
    assert d["cpu_baseline"]["window"] == {k: d["config"]["window"][k] for k in ("first_iteration", "last_iteration")}
    assert d["convergence"]["iterations_to_stop"] == 465 and d["convergence"]["inner_failures"] == 0
    assert d["partitioned_s100k"]["rccl_ranks"] == 1
