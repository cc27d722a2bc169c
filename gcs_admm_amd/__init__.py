"""MI355X-native ADMM loop for shortest paths in graphs of convex sets
(drop-in for the hot path of the reference's admm_solver_v3.py)."""
from .graph import (GcsGraph, build_graph, convert_pt_to_polytope, delta, graph_from_sets,  # noqa: F401
                    lattice_boxes, polytopes_overlap)
from .cases import load_fixture  # noqa: F401

__all__ = ["GcsGraph", "build_graph", "convert_pt_to_polytope", "delta", "graph_from_sets", "lattice_boxes",
           "polytopes_overlap", "load_fixture"]
