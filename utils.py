"""Root-level ``utils`` module so that reference-style case files
(``from utils import convert_pt_to_polytope, visualize_results`` --
test_data/test1.py:11-13 of the reference) import unchanged, without Drake.

Own implementations of the helpers the hot path is bracketed by:
``convert_pt_to_polytope`` / ``build_graph`` / ``delta`` (reference utils.py:12-98) live in
``gcs_admm_amd.graph``; ``save_data`` keeps the record layout of utils.py:212-229 so that the
reference's post-processing script can read the files this solver writes."""
import pickle

import numpy as np

from gcs_admm_amd.graph import build_graph, convert_pt_to_polytope, delta  # noqa: F401

RECORD_KEYS = ("As", "bs", "solve_time", "cost", "x_v_sol", "y_v_sol", "x_v_rounded", "y_v_rounded", "ADMM")
ADMM_KEYS = ("iterations", "rho_seq", "pri_res_seq", "dual_res_seq")


def save_data(data_file, As, bs, solve_time, cost, x_v_sol, y_v_sol, x_v_rounded, y_v_rounded, ADMM=True,
              iterations=None, rho_seq=None, pri_res_seq=None, dual_res_seq=None, **extra):
    """The reference's record (utils.py:197-233): same keys, same order.  ``extra`` appends fields the reference does
    not have (e.g. ``loop_wall_time``, ``inner_failures``); its post-processing script reads by key and ignores them."""
    record = dict(zip(RECORD_KEYS, (As, bs, solve_time, cost, x_v_sol, y_v_sol, x_v_rounded, y_v_rounded, ADMM)))
    if ADMM:
        record.update(zip(ADMM_KEYS, (iterations, rho_seq, pri_res_seq, dual_res_seq)))
    record.update(extra)
    with open(data_file, "wb") as f:
        pickle.dump(record, f)


def _polygon(A, b):
    """vertices of the bounded 2-D polygon A x <= b, counter-clockwise"""
    pts = []
    m = A.shape[0]
    for i in range(m):
        for j in range(i + 1, m):
            M = A[[i, j]]
            if abs(np.linalg.det(M)) < 1e-12:
                continue
            p = np.linalg.solve(M, b[[i, j]])
            if np.all(A @ p <= b + 1e-6):
                pts.append(p)
    if not pts:
        return np.zeros((0, 2))
    pts = np.array(pts)
    ctr = pts.mean(0)
    return pts[np.argsort(np.arctan2(pts[:, 1] - ctr[1], pts[:, 0] - ctr[0]))]


def visualize_results(As, bs, x_v, y_v, x_v_rounded=None, y_v_rounded=None, legend=False, save_to_file=None):
    """2-D picture of the regions and of the segments of the active vertices (y_v > 0.5)."""
    import matplotlib
    if save_to_file is not None:
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    panels = [("Original Data", x_v, y_v)]
    if x_v_rounded is not None and y_v_rounded is not None:
        panels.append(("Rounded Data", x_v_rounded, y_v_rounded))
    fig, axes = plt.subplots(1, len(panels), figsize=(8 * len(panels), 8), squeeze=False)
    cmap = plt.cm.tab10(np.linspace(0, 1, max(len(As), 1)))
    for ax, (title, xs, ys) in zip(axes[0], panels):
        for col, (key, A) in zip(cmap, As.items()):
            poly = _polygon(np.asarray(A, float), np.asarray(bs[key], float))
            if len(poly) and key not in ("s", "t"):
                ax.fill(poly[:, 0], poly[:, 1], alpha=0.3, color=col, label=f"Polytope {key}")
            if key in xs and key in ys and ys[key] > 0.5:
                seg = np.asarray(xs[key], float).reshape(2, 2)
                ax.plot(seg[:, 0], seg[:, 1], "o-", color=col)
        ax.set_aspect("equal", adjustable="datalim")
        ax.set_title(title)
        if legend:
            ax.legend()
    if save_to_file is not None:
        fig.savefig(save_to_file)
    else:
        plt.show()
