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
@pytest.mark.parametrize("name", ["benchmark1", "benchmark2", "benchmark3", "benchmark4"])
def test_cli_end_to_end(tmp_path, name):
    """the command line of the reference (admm_solver_v3.py:32-35) on the HIP loop, all four benchmarks: the written record
    (utils.py:197-233) read back as data, stop iteration / cost / rounded length against the reference's own records
    (tests/golden, from benchmark_data/admm_solver_v3_<name>.pkl; GCS_utils.py:92-181 for the rounding)."""
    sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
    from pkl_reader import load_data
    from gcs_admm_amd.cases import load_fixture
    r = subprocess.run([sys.executable, os.path.join(ROOT, "admm_solver_v3.py"), "--test_file", name, "--show_plot", "False"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "BREAKING FOR OPT" in r.stdout and "Inner solver failures: 0" in r.stdout
    rec = load_data(str(tmp_path / "benchmark_data" / f"admm_solver_v3_{name}.pkl"))
    assert list(rec)[:13] == ["As", "bs", "solve_time", "cost", "x_v_sol", "y_v_sol", "x_v_rounded", "y_v_rounded", "ADMM",
                              "iterations", "rho_seq", "pri_res_seq", "dual_res_seq"]
    gold = load_fixture(name)[0]["golden_v3"]
    assert rec["iterations"] == gold["iterations"]
    assert abs(rec["cost"] - gold["cost"]) <= 2e-4 * gold["cost"]
    assert len(rec["pri_res_seq"]) == rec["iterations"] + 1 == len(rec["rho_seq"])
    # solve_time = device time of the vertex / edge kernels (the reference's meaning); the loop's wall time contains it
    assert 0.0 < rec["solve_time"] <= rec["loop_wall_time"] and rec["inner_failures"] == 0
    n = 2
    keys = list(rec["x_v_rounded"])
    assert keys == list(rec["As"]) and rec["y_v_rounded"]["s"] == 1 and rec["y_v_rounded"]["t"] == 1
    length = sum(float(np.linalg.norm(np.asarray(rec["x_v_rounded"][v])[:n] - np.asarray(rec["x_v_rounded"][v])[n:]))
                 for v in keys if rec["y_v_rounded"][v] == 1)
    gx, gy = np.array(gold["x_v_rounded"]), np.array(gold["y_v_rounded"])
    glen = sum(np.linalg.norm(gx[i][:n] - gx[i][n:]) for i in range(len(gy)) if gy[i] == 1)
    assert abs(length - glen) <= 1e-5 * glen


@pytest.mark.gpu
def test_cli_runs_a_lattice_case_built_on_the_device(tmp_path):
    """a case of 627 regions as a reference-style module (``As, bs, n``): the command line builds its graph with the device LPs
    (the reference's |V|^2 host LPs, utils.py:68-72, are the bottleneck it cannot scale past), runs the loop and rounds -- the
    walk (GCS_utils.py:109-146) as a loop, I_v_out in one pass over E.  The lattice's edges are known exactly (graph.lattice_boxes)."""
    sys.path.insert(0, ROOT)
    from gcs_admm_amd.graph import lattice_boxes
    g = lattice_boxes(25, 25, seed=3)
    lines = ["import numpy as np", "n = 2", "As = {}", "bs = {}"]
    for i, k in enumerate(g.keys):
        A = g.poly_A[g.poly_ptr[i]:g.poly_ptr[i + 1]]; b = g.poly_b[g.poly_ptr[i]:g.poly_ptr[i + 1]]
        lines.append(f"As[{k!r}] = np.array({A.tolist()!r}); bs[{k!r}] = np.array({b.tolist()!r})")
    case_dir = tmp_path / "cases"; case_dir.mkdir()
    (case_dir / "lattice25.py").write_text("\n".join(lines) + "\n")
    env = dict(os.environ, PYTHONPATH=str(case_dir) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "admm_solver_v3.py"), "--test_file", "lattice25", "--show_plot", "False"],
                       capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "BREAKING FOR OPT" in r.stdout and "Inner solver failures: 0" in r.stdout
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pkl_reader import load_data
    rec = load_data(str(tmp_path / "benchmark_data" / "admm_solver_v3_lattice25.pkl"))
    assert rec["y_v_rounded"]["s"] == 1 and rec["y_v_rounded"]["t"] == 1
    on_path = [v for v in rec["y_v_rounded"] if rec["y_v_rounded"][v] == 1]
    length = sum(float(np.linalg.norm(np.asarray(rec["x_v_rounded"][v])[:2] - np.asarray(rec["x_v_rounded"][v])[2:])) for v in on_path)
    s_pt, t_pt = g.interior[g.src], g.interior[g.dst]
    straight = float(np.linalg.norm(t_pt - s_pt))
    assert straight - 1e-6 <= length <= 1.3 * straight            # a path, and a sensible one (the lattice is nearly convex)
    assert rec["cost"] <= length + 1e-6                            # the relaxation bounds the rounded length from below


@pytest.mark.gpu
def test_cli_runs_a_graph_file(tmp_path):
    """a case stored as a graph file (graph.save_graph: sets and edges as CSR arrays in one .npz) runs through the
    same command line as a case module of that name would, with the same result as the module-built graph"""
    sys.path.insert(0, ROOT)
    from gcs_admm_amd.cases import load_fixture
    from gcs_admm_amd.graph import save_graph
    case, g = load_fixture("benchmark1")
    case_dir = tmp_path / "cases"; case_dir.mkdir()
    save_graph(g, str(case_dir / "stored1.npz"))
    env = dict(os.environ, PYTHONPATH=str(case_dir) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "admm_solver_v3.py"), "--test_file", "stored1", "--show_plot", "False"],
                       capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "BREAKING FOR OPT" in r.stdout
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pkl_reader import load_data
    rec = load_data(str(tmp_path / "benchmark_data" / "admm_solver_v3_stored1.pkl"))
    gold = case["golden_v3"]
    assert rec["iterations"] == gold["iterations"] == 39 and abs(rec["cost"] - gold["cost"]) <= 2e-4
    assert list(rec["As"]) == g.keys


@pytest.mark.gpu
def test_cli_runs_a_case_whose_terminals_are_regions(tmp_path):
    """the reference's case format lets 's' / 't' be any polytope (test_data/*.py build them with convert_pt_to_polytope, utils.py:12-28,
    but admm_solver_v3.py:415-464 constrains them like every set): three boxes in a row, the outer two are the terminals.  Loop, rounding
    and record through the command line; the rounded path leaves s and enters t at the facing sides, length 1 (point terminals at the
    box centres would give 2)."""
    lines = ["import numpy as np", "n = 2", "A = np.vstack([np.eye(2), -np.eye(2)])",
             "As = {'s': A, 't': A, 0: A}",
             "bs = {'s': np.array([1.0, 1.0, 0.0, 0.0]), 't': np.array([3.0, 1.0, -2.0, 0.0]), 0: np.array([2.2, 1.0, -0.8, 0.0])}"]
    case_dir = tmp_path / "cases"; case_dir.mkdir()
    (case_dir / "region_row.py").write_text("\n".join(lines) + "\n")
    env = dict(os.environ, PYTHONPATH=str(case_dir) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "admm_solver_v3.py"), "--test_file", "region_row", "--show_plot", "False"],
                       capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "BREAKING FOR OPT" in r.stdout and "Inner solver failures: 0" in r.stdout
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pkl_reader import load_data
    rec = load_data(str(tmp_path / "benchmark_data" / "admm_solver_v3_region_row.pkl"))
    assert abs(rec["cost"] - 1.0002) < 1e-2
    assert all(rec["y_v_rounded"][v] == 1 for v in ("s", 0, "t"))
    length = sum(float(np.linalg.norm(np.asarray(rec["x_v_rounded"][v])[:2] - np.asarray(rec["x_v_rounded"][v])[2:])) for v in ("s", 0, "t"))
    assert abs(length - 1.0) < 1e-5
