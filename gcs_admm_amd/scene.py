"""Graph construction at scale on the device.

The reference's ``build_graph`` (utils.py:31-82) decides every ordered pair of regions with one LP
feasibility solve through Drake/MOSEK (``check_overlap``, :49-65) -- |V|^2 host solves, which caps it at a
few hundred regions.  Here:

  1. Chebyshev centres and axis-aligned bounding boxes of all regions: batched tiny LPs on the MI355X
     (``gcsadmm_polytope_centers`` / ``gcsadmm_polytope_bounds``, csrc/polytope_lp.hip);
  2. broad phase on the host: sort-and-sweep over the first coordinate of the boxes (numpy), which leaves
     only pairs whose boxes touch;
  3. narrow phase on the device: one LP per candidate pair (``gcsadmm_polytope_overlaps``), the same
     decision as the reference's feasibility solve (closed sets, touching counts);
  4. edges in the reference's double-loop order, both directions of every intersecting pair.

It raises if the HIP library or a device is missing.  LP statuses are checked (build_graph_device): the only host LPs this
module ever runs are re-decisions of overlap LPs that hit their iteration limit.  The host functions of
``gcs_admm_amd.graph`` (scipy LPs) remain what the small reference cases are built with.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Hashable, List, Sequence, Tuple

import numpy as np

from .graph import GcsGraph, _finish_graph
from .solver import GcsAdmmError, load_library

__all__ = ["PolytopeScene", "build_graph_device", "graph_from_sets_device"]


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class PolytopeScene:
    """Regions ``A_p x <= b_p`` (p = 0..P-1) as one CSR, with the device LP entry points."""

    def __init__(self, polys: Sequence[Tuple[np.ndarray, np.ndarray]], device: int = 0):
        self.n = int(np.asarray(polys[0][0]).shape[1])
        self.P = len(polys)
        self.ptr = np.zeros(self.P + 1, np.int32)
        self.ptr[1:] = np.cumsum([np.asarray(A).shape[0] for A, _ in polys])
        self.A = np.ascontiguousarray(np.vstack([np.asarray(A, float).reshape(-1, self.n) for A, _ in polys]))
        self.b = np.ascontiguousarray(np.hstack([np.asarray(b, float).ravel() for _, b in polys]))
        self.device = int(device)
        self.lib = load_library()
        self.lib.gcsadmm_polytope_last_error.restype = C.c_char_p
        self._centers = None

    def _check(self, rc, what):
        if rc != 0:
            raise GcsAdmmError(f"{what}: {self.lib.gcsadmm_polytope_last_error().decode()} (status {rc})")

    def centers(self):
        """(centres [P, n], radii [P], LP status [P])."""
        cen = np.empty((self.P, self.n)); rad = np.empty(self.P); st = np.empty(self.P, np.int32)
        rc = self.lib.gcsadmm_polytope_centers(C.c_int(self.n), C.c_int(self.P), _ptr(self.ptr, C.c_int), _ptr(self.A, C.c_double),
                                               _ptr(self.b, C.c_double), C.c_int(self.device), _ptr(cen, C.c_double),
                                               _ptr(rad, C.c_double), _ptr(st, C.c_int))
        self._check(rc, "gcsadmm_polytope_centers")
        self._centers = cen
        return cen, rad, st

    def bounds(self, centers=None):
        """(lo [P, n], hi [P, n], LP status [P, 2n])."""
        cen = np.ascontiguousarray(centers if centers is not None else (self._centers if self._centers is not None else self.centers()[0]))
        lo = np.empty((self.P, self.n)); hi = np.empty((self.P, self.n)); st = np.empty((self.P, 2 * self.n), np.int32)
        rc = self.lib.gcsadmm_polytope_bounds(C.c_int(self.n), C.c_int(self.P), _ptr(self.ptr, C.c_int), _ptr(self.A, C.c_double),
                                              _ptr(self.b, C.c_double), _ptr(cen, C.c_double), C.c_int(self.device),
                                              _ptr(lo, C.c_double), _ptr(hi, C.c_double), _ptr(st, C.c_int))
        self._check(rc, "gcsadmm_polytope_bounds")
        return lo, hi, st

    def overlaps(self, pair_a, pair_b, tol: float = 1e-9, centers=None):
        """uint8 flags [num_pairs] and LP status: do regions pair_a[t], pair_b[t] intersect?"""
        pa = np.ascontiguousarray(pair_a, np.int32); pb = np.ascontiguousarray(pair_b, np.int32)
        out = np.zeros(len(pa), np.uint8); st = np.zeros(len(pa), np.int32)
        cen = centers if centers is not None else self._centers
        cen = np.ascontiguousarray(cen) if cen is not None else None
        rc = self.lib.gcsadmm_polytope_overlaps(C.c_int(self.n), C.c_int(self.P), _ptr(self.ptr, C.c_int), _ptr(self.A, C.c_double),
                                                _ptr(self.b, C.c_double), _ptr(cen, C.c_double) if cen is not None else None,
                                                C.c_long(len(pa)), _ptr(pa, C.c_int), _ptr(pb, C.c_int), C.c_double(tol),
                                                C.c_int(self.device), _ptr(out, C.c_ubyte), _ptr(st, C.c_int))
        self._check(rc, "gcsadmm_polytope_overlaps")
        return out, st


def candidate_pairs(lo: np.ndarray, hi: np.ndarray, pad: float = 1e-7):
    """Unordered pairs (i < j) whose padded boxes intersect: sort on the first coordinate, sweep with a
    vectorised window per box (numpy searchsorted), test the remaining coordinates on the candidates."""
    P = lo.shape[0]
    order = np.argsort(lo[:, 0], kind="stable")
    los = lo[order, 0]
    # box order[k] can meet later boxes order[k+1 .. end_k) only: those whose lo_0 <= hi_0 + pad
    end = np.searchsorted(los, hi[order, 0] + pad, side="right")
    cnt = np.maximum(end - np.arange(P) - 1, 0)
    tot = int(cnt.sum())
    if tot == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    first = np.repeat(np.arange(P), cnt)
    offs = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    second = first + 1 + offs
    i = order[first]; j = order[second]
    keep = np.all(lo[i, 1:] <= hi[j, 1:] + pad, axis=1) & np.all(lo[j, 1:] <= hi[i, 1:] + pad, axis=1)
    i, j = i[keep], j[keep]
    a = np.minimum(i, j).astype(np.int32); b = np.maximum(i, j).astype(np.int32)
    return a, b


def build_graph_device(As: Dict[Hashable, np.ndarray], bs: Dict[Hashable, np.ndarray], device: int = 0, tol: float = 1e-9,
                       scene=None, stats: dict | None = None):
    """``utils.build_graph`` (reference utils.py:31-82) with the LPs on the device.  Returns
    ``(vertices, edges, I_v_in, I_v_out, centres)``; ``edges`` in the reference's double-loop order.

    Every LP reports a status (0 converged, 1 / 2 decided early, -1 iteration limit) and none is ignored:
      * a centre LP that did not converge has no trustworthy interior point -> GcsAdmmError naming the regions;
      * a bounding-box LP that did not converge returns an INTERIOR iterate, i.e. a box that is too small, and the sweep
        would silently drop real neighbours: that side of the box is opened up (+-inf), which only adds candidates;
      * an overlap LP that did not converge is decided again by the host LP of ``graph.polytopes_overlap`` (one HiGHS
        solve per pair, the reference's own method) instead of from its unfinished iterate.
    ``stats`` (optional dict) receives the counts.  ``scene``: a prepared PolytopeScene (tests inject one)."""
    vertices = list(As.keys())
    if scene is None:
        scene = PolytopeScene([(As[v], bs[v]) for v in vertices], device)
    cen, rad, st_c = scene.centers()
    if np.any(st_c < 0):
        bad = [vertices[i] for i in np.nonzero(st_c < 0)[0][:8]]
        raise GcsAdmmError(f"centre LP did not converge for regions {bad} (of {int((st_c < 0).sum())})")
    if np.any(rad <= 0):
        bad = [vertices[i] for i in np.nonzero(rad <= 0)[0][:5]]
        raise ValueError(f"regions without interior: {bad}")
    lo, hi, st_b = scene.bounds(cen)
    n = lo.shape[1]
    # status layout of gcsadmm_polytope_bounds (polytope_lp.hip bounds_kernel): [P][n][2] = (min, max) per axis
    st_b = np.asarray(st_b).reshape(len(vertices), n, 2)
    fail_lo, fail_hi = st_b[:, :, 0] < 0, st_b[:, :, 1] < 0
    lo = np.where(fail_lo, -np.inf, lo); hi = np.where(fail_hi, np.inf, hi)
    pa, pb = candidate_pairs(lo, hi)
    flags, st_o = scene.overlaps(pa, pb, tol, cen)
    redo = np.nonzero(np.asarray(st_o) < 0)[0]
    if len(redo):
        from .graph import polytopes_overlap
        flags = np.array(flags, copy=True)
        for t in redo:
            u, w = vertices[pa[t]], vertices[pb[t]]
            flags[t] = 1 if polytopes_overlap(np.asarray(As[u], float), np.asarray(bs[u], float),
                                              np.asarray(As[w], float), np.asarray(bs[w], float)) else 0
    if stats is not None:
        stats.update(bounds_opened=int(fail_lo.sum() + fail_hi.sum()), overlaps_redone_on_host=int(len(redo)),
                     candidate_pairs=int(len(pa)))
    a = pa[flags != 0]; b = pb[flags != 0]
    tail = np.concatenate([a, b]); head = np.concatenate([b, a])
    o = np.lexsort((head, tail))                      # double-loop order: by tail, then head
    edges = [(vertices[t], vertices[h]) for t, h in zip(tail[o], head[o])]
    I_v_in = {v: [] for v in vertices}
    I_v_out = {v: [] for v in vertices}
    for e in edges:
        I_v_out[e[0]].append(e)
        I_v_in[e[1]].append(e)
    return vertices, edges, I_v_in, I_v_out, cen


def graph_from_sets_device(As, bs, n, device: int = 0) -> GcsGraph:
    """``graph_from_sets`` with edges and interior points from the device LPs."""
    keys = list(As.keys())
    if 's' not in As or 't' not in As:
        raise KeyError("case must define vertices 's' and 't'")
    _, edges, _, _, cen = build_graph_device(As, bs, device)
    index = {k: i for i, k in enumerate(keys)}
    tail = [index[u] for u, _ in edges]
    head = [index[w] for _, w in edges]
    polys = [(np.asarray(As[k], float), np.asarray(bs[k], float)) for k in keys]
    return _finish_graph(int(n), keys, tail, head, polys, cen, index['s'], index['t'])
