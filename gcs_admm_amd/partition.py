"""Vertex partition of a GCS for multi-GPU runs.

Each rank owns a set of vertices, solves their sub-problems, and holds every directed edge that
touches one of them.  An edge whose endpoints live on two ranks is a *cut edge*: after the vertex
step each side sends its copy of the edge's coupled words (c = 2n+1 words) to the other -- the halo
exchange -- and both then compute the identical average, their own dual update and, redundantly,
the other side's.  The five norms are summed with an ownership rule so that nothing is counted
twice (copy-indexed terms by the copy's owner, edge-indexed terms by the owner of the edge's tail)
and all-reduced; every rank then takes the same rho / stop decision from the same numbers.

Everything here is plain index bookkeeping on the host (numpy); the exchange itself is
``torch.distributed`` on whatever device the state lives on (RCCL on MI355X, gloo in the CPU tests).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List

import numpy as np

from .graph import GcsGraph


def strip_owner(g: GcsGraph, world: int) -> np.ndarray:
    """Contiguous ranges of the vertex order with (nearly) equal incidence counts.  For the lattice
    generators the order is row-major, so these are row strips; 's' and 't' go with their
    neighbours' rank (first / last strip)."""
    V = g.num_vertices
    deg = np.diff(g.inc_ptr).astype(np.int64) + 1
    others = np.array([v for v in range(V) if v not in (g.src, g.dst)], dtype=np.int64)
    cum = np.cumsum(deg[others])
    owner = np.zeros(V, dtype=np.int32)
    bounds = cum[-1] * (np.arange(1, world) / world) if len(cum) else np.zeros(0)
    owner[others] = np.searchsorted(bounds, cum - deg[others] / 2.0, side="right")
    for term in (g.src, g.dst):
        nb = [g.edge_head[e] for e in range(g.num_edges) if g.edge_tail[e] == term] or [others[0] if len(others) else 0]
        owner[term] = owner[nb[0]] if nb[0] not in (g.src, g.dst) else 0
    return owner


@dataclasses.dataclass
class LocalPartition:
    rank: int
    world: int
    graph: GcsGraph                 # owned vertices (global order) + every edge touching them, local ids
    num_incidences: int             # owned incidence columns + ghost columns
    inc_counted: np.ndarray         # uint8 [num_incidences]
    edge_counted: np.ndarray        # uint8 [E_local]
    nx_global: float
    nmu_global: float
    vertex_global: np.ndarray       # [V_local] global vertex id
    edge_global: np.ndarray         # [E_local] global edge id
    send_idx: Dict[int, np.ndarray]  # neighbour rank -> local incidence columns to send (canonical order)
    recv_idx: Dict[int, np.ndarray]  # neighbour rank -> ghost columns to fill (same canonical order)


def build_partition(g: GcsGraph, owner: np.ndarray, rank: int, world: int) -> LocalPartition:
    V, E, n = g.num_vertices, g.num_edges, g.n
    mine_v = np.nonzero(owner == rank)[0]
    vloc = -np.ones(V, dtype=np.int64); vloc[mine_v] = np.arange(len(mine_v))
    t_own, h_own = owner[g.edge_tail], owner[g.edge_head]
    mine_e = np.nonzero((t_own == rank) | (h_own == rank))[0]
    eloc = -np.ones(E, dtype=np.int64); eloc[mine_e] = np.arange(len(mine_e))
    # local CSR over owned vertices (same incidence order as the global graph)
    inc_ptr = np.zeros(len(mine_v) + 1, dtype=np.int64)
    inc_edge, inc_out = [], []
    col_of = {}                                      # (global edge, is_head) -> local column
    for lv, v in enumerate(mine_v):
        lo, hi = g.inc_ptr[v], g.inc_ptr[v + 1]
        for k in range(lo, hi):
            e = g.inc_edge[k]
            col_of[(int(e), int(not g.inc_out[k]))] = len(inc_edge)
            inc_edge.append(eloc[e]); inc_out.append(g.inc_out[k])
        inc_ptr[lv + 1] = len(inc_edge)
    ni_owned = len(inc_edge)
    # ghost columns: the remote endpoint's copy of each cut edge, ordered by (global edge, side)
    ghosts = []
    for e in mine_e:
        if t_own[e] != rank:
            ghosts.append((int(e), 0, int(t_own[e])))
        if h_own[e] != rank:
            ghosts.append((int(e), 1, int(h_own[e])))
    for i, (e, side, _r) in enumerate(ghosts):
        col_of[(e, side)] = ni_owned + i
    ni = ni_owned + len(ghosts)
    edge_inc_tail = np.array([col_of[(int(e), 0)] for e in mine_e], dtype=np.int32)
    edge_inc_head = np.array([col_of[(int(e), 1)] for e in mine_e], dtype=np.int32)
    inc_counted = np.zeros(ni, dtype=np.uint8); inc_counted[:ni_owned] = 1
    edge_counted = (t_own[mine_e] == rank).astype(np.uint8)
    # halo lists: with neighbour r, for every cut edge between us in (global edge, side) order, the side
    # that is MINE is sent and the side that is THEIRS is received
    send: Dict[int, List[int]] = {}
    recv: Dict[int, List[int]] = {}
    for e in mine_e:
        for side, own in ((0, t_own[e]), (1, h_own[e])):
            other = h_own[e] if side == 0 else t_own[e]
            if own == rank and other != rank:
                send.setdefault(int(other), []).append(col_of[(int(e), side)])
            elif own != rank:
                recv.setdefault(int(own), []).append(col_of[(int(e), side)])
    poly_ptr = np.zeros(len(mine_v) + 1, dtype=np.int64)
    As, bs = [], []
    for lv, v in enumerate(mine_v):
        As.append(g.poly_A[g.poly_ptr[v]:g.poly_ptr[v + 1]]); bs.append(g.poly_b[g.poly_ptr[v]:g.poly_ptr[v + 1]])
        poly_ptr[lv + 1] = poly_ptr[lv] + len(bs[-1])
    local = GcsGraph(
        n=n, keys=[g.keys[v] for v in mine_v],
        edge_tail=np.array([vloc[g.edge_tail[e]] for e in mine_e], dtype=np.int32),    # -1 = remote vertex
        edge_head=np.array([vloc[g.edge_head[e]] for e in mine_e], dtype=np.int32),
        inc_ptr=inc_ptr.astype(np.int32), inc_edge=np.array(inc_edge, dtype=np.int32),
        inc_out=np.array(inc_out, dtype=np.int32), edge_inc_tail=edge_inc_tail, edge_inc_head=edge_inc_head,
        poly_ptr=poly_ptr.astype(np.int32),
        poly_A=np.ascontiguousarray(np.vstack(As)) if As else np.zeros((0, n)),
        poly_b=np.ascontiguousarray(np.hstack(bs)) if bs else np.zeros(0),
        interior=np.ascontiguousarray(g.interior[mine_v]),
        src=int(vloc[g.src]), dst=int(vloc[g.dst]))
    return LocalPartition(rank=rank, world=world, graph=local, num_incidences=ni, inc_counted=inc_counted,
                          edge_counted=edge_counted, nx_global=float(g.nx), nmu_global=float(g.nmu),
                          vertex_global=mine_v, edge_global=mine_e,
                          send_idx={r: np.array(v, dtype=np.int64) for r, v in send.items()},
                          recv_idx={r: np.array(v, dtype=np.int64) for r, v in recv.items()})


class PartitionedLoop:
    """The ADMM loop over a vertex partition: backend-agnostic driver.

    ``backend`` is one rank's compute object with the DeviceSolver interface (``copy`` tensor
    [c, NI], ``vertex_step()``, ``edge_step() -> sums tensor[5]``, ``control(sums)``,
    ``read_control()``).  On MI355X that is ``gcs_admm_amd.solver.DeviceSolver`` and the group is
    RCCL; the CPU tests plug in an oracle-backed stand-in over gloo.  Per iteration: one halo
    exchange (point-to-point with each neighbour rank) and one 5-double all-reduce."""

    def __init__(self, part: LocalPartition, backend, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.part, self.backend, self.group = part, backend, group
        dev = backend.copy.device
        self._send = {r: torch.as_tensor(ix, device=dev) for r, ix in sorted(part.send_idx.items())}
        self._recv = {r: torch.as_tensor(ix, device=dev) for r, ix in sorted(part.recv_idx.items())}
        c = backend.copy.shape[0]
        self._sbuf = {r: torch.empty(c, len(ix), dtype=backend.copy.dtype, device=dev) for r, ix in self._send.items()}
        self._rbuf = {r: torch.empty(c, len(ix), dtype=backend.copy.dtype, device=dev) for r, ix in self._recv.items()}

    def halo_exchange(self):
        dist, copy = self.dist, self.backend.copy
        ops = []
        for r, ix in self._send.items():
            self.torch.index_select(copy, 1, ix, out=self._sbuf[r])
            ops.append(dist.P2POp(dist.isend, self._sbuf[r], r, group=self.group))
        for r in self._recv:
            ops.append(dist.P2POp(dist.irecv, self._rbuf[r], r, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for r, ix in self._recv.items():
            copy.index_copy_(1, ix, self._rbuf[r])

    def iterate(self, k: int = 1):
        for _ in range(k):
            self.backend.vertex_step()
            self.halo_exchange()
            sums = self.backend.edge_step()
            self.dist.all_reduce(sums, group=self.group)
            self.backend.control(sums)

    def solve(self, chunk: int = 25, max_it: int = 1000):
        done = 0
        while True:
            k = min(chunk, max_it - done)
            if k > 0:
                self.iterate(k)
                done += k
            cb = self.backend.read_control()
            if cb.status != -1 or done >= max_it:
                return cb
