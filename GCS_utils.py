"""``compute_cost`` with the reference's meaning (GCS_utils.py:184-211): sum over vertices of
|z_v[:n] - z_v[n:]| plus 1e-4 per unit of edge activation.  (The device path computes the same
number with gcsadmm_cost; this dict version exists for callers that hold the reference's dicts.)
Rounding and the convex restriction (GCS_utils.py:17-181) live in gcs_admm_amd/rounding.py."""
import numpy as np

from gcs_admm_amd.rounding import rounding, solve_path_restriction  # noqa: F401


def compute_cost(z_v_sol, y_e_sol):
    length = sum(float(np.linalg.norm(z[: len(z) // 2] - z[len(z) // 2:])) for z in z_v_sol.values())
    return length + 1e-4 * float(sum(y_e_sol.values()))
