import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_case_modules_follow_reference_contract():
    sys.path.insert(0, os.path.join(ROOT, "test_data"))
    import importlib
    mod = importlib.import_module("test1")
    assert set(mod.As) == {"s", "t", 0} and mod.n == 2 and mod.As[0].shape == (3, 2)
    assert np.allclose(mod.bs["s"], [0.1 + 1e-6, 0.1 + 1e-6, -0.1 + 1e-6, -0.1 + 1e-6])


def test_missing_case_exits_1():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "admm_solver_v3.py"), "--test_file", "nope", "--show_plot", "False"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and "not found" in r.stdout


def test_compute_cost_and_record(tmp_path):
    sys.path.insert(0, ROOT)
    from GCS_utils import compute_cost
    from utils import save_data
    z = {"a": np.array([0.0, 0.0, 3.0, 4.0]), "b": np.zeros(4)}
    assert abs(compute_cost(z, {("a", "b"): 0.5}) - (5.0 + 0.5e-4)) < 1e-15
    f = tmp_path / "admm_solver_v3_x.pkl"
    save_data(str(f), {}, {}, 1.0, 2.0, {}, {}, None, None, True, 3, np.ones(4), np.zeros(4), np.zeros(4))
    import pickle
    rec = pickle.load(open(f, "rb"))            # a file this test wrote itself
    assert list(rec) == ["As", "bs", "solve_time", "cost", "x_v_sol", "y_v_sol", "x_v_rounded", "y_v_rounded", "ADMM",
                         "iterations", "rho_seq", "pri_res_seq", "dual_res_seq"]


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "admm_solver_v3.py"), "--test_file", "benchmark1", "--show_plot", "False"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "BREAKING FOR OPT" in r.stdout and "Cost before rounding: 2.98" in r.stdout
    assert (tmp_path / "benchmark_data" / "admm_solver_v3_benchmark1.pkl").exists()
