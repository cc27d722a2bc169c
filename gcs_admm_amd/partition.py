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
from typing import Dict

import numpy as np

from .graph import GcsGraph


def strip_owner(g: GcsGraph, world: int) -> np.ndarray:
    """Contiguous ranges of the vertex order with (nearly) equal incidence counts.  For the lattice
    generators the order is row-major, so these are row strips; 's' and 't' go with their
    neighbours' rank (first / last strip)."""
    V = g.num_vertices
    deg = np.diff(g.inc_ptr).astype(np.int64) + 1
    mask = np.ones(V, bool); mask[[g.src, g.dst]] = False
    others = np.nonzero(mask)[0]
    cum = np.cumsum(deg[others])
    owner = np.zeros(V, dtype=np.int32)
    bounds = cum[-1] * (np.arange(1, world) / world) if len(cum) else np.zeros(0)
    owner[others] = np.searchsorted(bounds, cum - deg[others] / 2.0, side="right")
    for term in (g.src, g.dst):
        nb = g.edge_head[g.edge_tail == term]
        first = int(nb[0]) if len(nb) else (int(others[0]) if len(others) else 0)
        owner[term] = owner[first] if first not in (g.src, g.dst) else 0
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


def _ranges(starts: np.ndarray, counts: np.ndarray) -> np.ndarray:
    """concatenation of arange(starts[i], starts[i] + counts[i])"""
    tot = int(counts.sum())
    if tot == 0:
        return np.zeros(0, np.int64)
    rep = np.repeat(starts - np.concatenate([[0], np.cumsum(counts)[:-1]]), counts)
    return rep + np.arange(tot)


def build_partition(g: GcsGraph, owner: np.ndarray, rank: int, world: int) -> LocalPartition:
    """Rank ``rank``'s part of the graph: its vertices (global order), every edge touching them, one ghost copy column per
    cut edge, ownership masks and the halo index lists.  Pure index arithmetic (numpy, no Python loop over vertices or
    edges): 100k vertices / 400k edges take well under a second per rank."""
    V, E, n = g.num_vertices, g.num_edges, g.n
    owner = np.asarray(owner)
    mine_v = np.nonzero(owner == rank)[0]
    vloc = -np.ones(V, dtype=np.int64); vloc[mine_v] = np.arange(len(mine_v))
    t_own, h_own = owner[g.edge_tail], owner[g.edge_head]
    mine_e = np.nonzero((t_own == rank) | (h_own == rank))[0]
    eloc = -np.ones(E, dtype=np.int64); eloc[mine_e] = np.arange(len(mine_e))
    # local CSR over owned vertices (same incidence order as the global graph)
    gptr = g.inc_ptr.astype(np.int64)
    degs = gptr[mine_v + 1] - gptr[mine_v]
    inc_idx = _ranges(gptr[mine_v], degs)                      # global incidence slots of the owned vertices, in order
    inc_ptr = np.zeros(len(mine_v) + 1, dtype=np.int64); inc_ptr[1:] = np.cumsum(degs)
    inc_edge_g = g.inc_edge[inc_idx].astype(np.int64)
    inc_out = g.inc_out[inc_idx].astype(np.int32)
    ni_owned = len(inc_idx)
    # column of (global edge, side): side 0 = the tail's copy, 1 = the head's copy
    col = -np.ones(2 * E, dtype=np.int64)
    col[2 * inc_edge_g + (1 - inc_out)] = np.arange(ni_owned)
    # ghost columns: the remote endpoint's copy of each cut edge, ordered by (global edge, side)
    side_owner = np.stack([t_own[mine_e], h_own[mine_e]], 1).ravel()          # [(e, side)] in (edge, side) order
    side_key = (2 * mine_e[:, None] + np.arange(2)[None, :]).ravel()
    ghost = side_owner != rank
    col[side_key[ghost]] = ni_owned + np.arange(int(ghost.sum()))
    ni = ni_owned + int(ghost.sum())
    edge_inc_tail = col[2 * mine_e].astype(np.int32)
    edge_inc_head = col[2 * mine_e + 1].astype(np.int32)
    inc_counted = np.zeros(ni, dtype=np.uint8); inc_counted[:ni_owned] = 1
    edge_counted = (t_own[mine_e] == rank).astype(np.uint8)
    # halo lists: with neighbour r, for every cut edge between us in (global edge, side) order, the side
    # that is MINE is sent and the side that is THEIRS is received
    other_owner = np.stack([h_own[mine_e], t_own[mine_e]], 1).ravel()          # owner of the opposite endpoint
    send_mask = (side_owner == rank) & (other_owner != rank)
    send = {int(r): col[side_key[send_mask & (other_owner == r)]] for r in np.unique(other_owner[send_mask])}
    recv = {int(r): col[side_key[ghost & (side_owner == r)]] for r in np.unique(side_owner[ghost])}
    pptr = g.poly_ptr.astype(np.int64)
    ms = pptr[mine_v + 1] - pptr[mine_v]
    rows = _ranges(pptr[mine_v], ms)
    poly_ptr = np.zeros(len(mine_v) + 1, dtype=np.int64); poly_ptr[1:] = np.cumsum(ms)
    local = GcsGraph(
        n=n, keys=[g.keys[v] for v in mine_v],
        edge_tail=vloc[g.edge_tail[mine_e]].astype(np.int32),    # -1 = remote vertex
        edge_head=vloc[g.edge_head[mine_e]].astype(np.int32),
        inc_ptr=inc_ptr.astype(np.int32), inc_edge=eloc[inc_edge_g].astype(np.int32),
        inc_out=inc_out, edge_inc_tail=edge_inc_tail, edge_inc_head=edge_inc_head,
        poly_ptr=poly_ptr.astype(np.int32),
        poly_A=np.ascontiguousarray(g.poly_A[rows]).reshape(-1, n),
        poly_b=np.ascontiguousarray(g.poly_b[rows]),
        interior=np.ascontiguousarray(g.interior[mine_v]),
        src=int(vloc[g.src]), dst=int(vloc[g.dst]))
    return LocalPartition(rank=rank, world=world, graph=local, num_incidences=ni, inc_counted=inc_counted,
                          edge_counted=edge_counted, nx_global=float(g.nx), nmu_global=float(g.nmu),
                          vertex_global=mine_v, edge_global=mine_e,
                          send_idx={r: np.asarray(v, dtype=np.int64) for r, v in send.items()},
                          recv_idx={r: np.asarray(v, dtype=np.int64) for r, v in recv.items()})


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


def device_partition(g: GcsGraph, rank: int, world: int, state_dtype: str = "f32", device=None, group=None, owner=None, **kw):
    """One rank's DeviceSolver for a strip partition of ``g``, joined to an RCCL communicator through the C ABI
    (gcsadmm_attach_comm).  The 128-byte communicator id is created by rank 0 and distributed with ``torch.distributed``
    (which must be initialised when world > 1; any backend: it only carries the 128 bytes).  The loop itself then runs
    entirely behind the ABI: ``solver.enqueue_partitioned(k)`` / ``solver.solve_partitioned()``.  A single rank (world 1) joins a
    communicator of its own, so that the loop issues the same RCCL calls at every world size -- unless no RCCL can be loaded, in which
    case it runs without one (exchange and all-reduce are no-ops for one rank).
    Returns (LocalPartition, DeviceSolver)."""
    import torch
    from .solver import DeviceSolver
    err, part, dev, uid = None, None, None, None
    try:       # everything local first: a rank that fails here must not leave the others waiting in a collective
        owner = strip_owner(g, world) if owner is None else owner
        part = build_partition(g, owner, rank, world)
        dev = DeviceSolver(part.graph, state_dtype, device=device, num_incidences=part.num_incidences, inc_counted=part.inc_counted,
                           edge_counted=part.edge_counted, nx_global=part.nx_global, nmu_global=part.nmu_global, **kw)
        dev.check_halo(rank, world, part.send_idx, part.recv_idx)      # what attach_comm would reject, before anyone enters its collective
        if rank == 0:
            try:
                uid = dev.unique_id()
            except Exception:
                if world > 1:
                    raise
                uid = None      # a single rank needs no RCCL: no communicator, the loop's exchange and all-reduce are no-ops
    except Exception as exc:
        err = exc
    if world > 1:
        import torch.distributed as dist
        backend = dist.get_backend(group)
        tdev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device)) if backend == "nccl" else torch.device("cpu")
        flag = torch.tensor([0.0 if err is not None else 1.0], device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if err is None and float(flag.item()) == 0.0:
            err = RuntimeError("another rank failed to set up its partition")
        if err is not None:
            raise err
        t = torch.zeros(128, dtype=torch.uint8, device=tdev)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
        dist.broadcast(t, src=0, group=group)
        uid = bytes(t.cpu().numpy().tobytes())
    elif err is not None:
        raise err
    dev.attach_comm(rank, world, uid, part.send_idx, part.recv_idx)
    return part, dev
