"""Host side of the MI355X ADMM loop: thin ctypes binding of the C ABI
(include/gcsadmm.h, built from gcs_admm_amd/csrc by ``build.py``) plus the
driver that mirrors the reference's main loop (admm_solver_v3.py:621-733).

Device buffers are PyTorch-ROCm tensors; the library only ever sees their
``data_ptr()`` and the current HIP stream.  There is NO fallback: if the HIP
library is missing or no GPU is visible this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .graph import GcsGraph

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgcsadmm.so")

F64, F32 = 0, 1
RUNNING, CONVERGED, MAX_IT, DIVERGED = -1, 0, 1, 2
STATUS_NAME = {RUNNING: "running", CONVERGED: "converged", MAX_IT: "max_it", DIVERGED: "diverged"}

EXPORTS = ["gcsadmm_create", "gcsadmm_destroy", "gcsadmm_last_error", "gcsadmm_reset", "gcsadmm_vertex_step",
           "gcsadmm_edge_step", "gcsadmm_control", "gcsadmm_run", "gcsadmm_run_timed", "gcsadmm_read_control",
           "gcsadmm_cost", "gcsadmm_query", "gcsadmm_unit_iterations", "gcsadmm_vertex_prox",
           # vertex partitions across GPUs (RCCL)
           "gcsadmm_comm_unique_id", "gcsadmm_check_halo", "gcsadmm_attach_comm", "gcsadmm_run_partitioned", "gcsadmm_halo_pack", "gcsadmm_halo_unpack",
           "gcsadmm_halo_exchange", "gcsadmm_halo_buffers", "gcsadmm_run_partitioned_timed", "gcsadmm_comm_count", "gcsadmm_set_overlap",
           # graph construction at scale (gcs_admm_amd/scene.py)
           "gcsadmm_polytope_last_error", "gcsadmm_polytope_centers", "gcsadmm_polytope_bounds", "gcsadmm_polytope_overlaps"]


class GraphDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("num_vertices", C.c_int32), ("num_edges", C.c_int32), ("num_incidences", C.c_int32),
                ("inc_ptr", C.c_void_p), ("inc_edge", C.c_void_p), ("inc_out", C.c_void_p),
                ("edge_inc_tail", C.c_void_p), ("edge_inc_head", C.c_void_p),
                ("poly_ptr", C.c_void_p), ("poly_A", C.c_void_p), ("poly_b", C.c_void_p), ("center", C.c_void_p),
                ("src", C.c_int32), ("dst", C.c_int32), ("state_dtype", C.c_int32), ("device", C.c_int32),
                ("inc_counted", C.c_void_p), ("edge_counted", C.c_void_p),
                ("nx_global", C.c_double), ("nmu_global", C.c_double),
                # schedule of the vertex step (0 = automatic): see include/gcsadmm.h
                ("vertex_program", C.c_int32), ("wave_slots", C.c_int32), ("wave_align", C.c_int32),
                ("wave_store_dl", C.c_int32), ("wave_generic_rows", C.c_int32), ("edge_major_columns", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("rho", C.c_double), ("tau_incr", C.c_double), ("tau_decr", C.c_double), ("nu", C.c_double),
                ("it_rho_limit", C.c_int32), ("max_it", C.c_int32), ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("eps_edge", C.c_double), ("ipm_tol", C.c_double), ("ipm_max_iter", C.c_int32), ("cold_start", C.c_int32)]


class State(C.Structure):
    _fields_ = [("copy", C.c_void_p), ("mu", C.c_void_p), ("zedge", C.c_void_p),
                ("xv", C.c_void_p), ("zv", C.c_void_p), ("yv", C.c_void_p)]


class HaloDesc(C.Structure):
    _fields_ = [("num_peers", C.c_int32), ("peer_rank", C.c_void_p), ("send_ptr", C.c_void_p), ("send_cols", C.c_void_p),
                ("recv_ptr", C.c_void_p), ("recv_cols", C.c_void_p)]


def halo_arrays(send_idx, recv_idx):
    """The halo lists of a vertex partition (gcs_admm_amd.partition.LocalPartition.send_idx / recv_idx: neighbour rank ->
    columns in canonical (global edge, side) order) as the flat CSR arrays of ``gcsadmm_halo_desc``: peers in ascending
    rank order, send and receive lists of a peer equally long.  Returns (peer_rank, ptr, send_cols, recv_cols) int32."""
    peers = sorted(send_idx)
    if sorted(recv_idx) != peers:
        raise ValueError("a partition sends to and receives from the same neighbours")
    cnt = [len(send_idx[r]) for r in peers]
    if cnt != [len(recv_idx[r]) for r in peers]:
        raise ValueError("send and receive lists of a neighbour must be equally long")
    ptr = np.zeros(len(peers) + 1, np.int32); ptr[1:] = np.cumsum(cnt)
    cat = lambda d: (np.concatenate([np.asarray(d[r], np.int32) for r in peers]) if peers else np.zeros(0, np.int32))
    return np.asarray(peers, np.int32), ptr, np.ascontiguousarray(cat(send_idx)), np.ascontiguousarray(cat(recv_idx))


class ControlBlock(C.Structure):
    _fields_ = [("rho", C.c_double), ("mu_scale", C.c_double), ("sums", C.c_double * 5),
                ("pri", C.c_double), ("dual", C.c_double), ("eps_pri", C.c_double), ("eps_dual", C.c_double),
                ("it", C.c_int32), ("status", C.c_int32), ("inner_failures", C.c_int32), ("inner_iters", C.c_int32)]


_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m gcs_admm_amd.build` (hipcc, gfx950). "
                               "There is no CPU fallback.")
        # PyTorch-ROCm ships its own HIP runtime (same SONAME as /opt/rocm's): it must be the one already
        # loaded when libgcsadmm.so resolves libamdhip64, or the process ends up with two runtimes and
        # the tensors' device pointers mean nothing to the library.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        lib.gcsadmm_last_error.restype = C.c_char_p
        lib.gcsadmm_last_error.argtypes = [C.c_void_p]
        lib.gcsadmm_destroy.restype = None
        lib.gcsadmm_destroy.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


class GcsAdmmError(RuntimeError):
    pass


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceSolver:
    """One GCS instance (or one vertex partition of it) resident on one MI355X.

    ``graph`` is the integer/CSR description (gcs_admm_amd.graph.GcsGraph).  For
    a partition, ``num_incidences`` > ``inc_ptr[-1]`` adds ghost copy slots and
    ``inc_counted`` / ``edge_counted`` implement the ownership rule of the
    global norms.

    ``columns``: numbering of the state columns of ``copy`` / ``mu``.  "incidence" (default): column k = position k of
    the vertex CSR, the numbering ``graph`` and every per-column argument (``inc_counted``, halo lists) are written in.
    "edge": tail side of edge e in column e, head side in column E + e -- the edge step becomes a pure stream (the layout for
    large graphs, include/gcsadmm.h ``edge_major_columns``).  Arguments stay in incidence numbering either way; ``col_of``
    maps an incidence column to the state column, and ``copy[:, col_of]`` is the state in incidence order.
    """

    def __init__(self, graph: GcsGraph, state_dtype: str = "f64", device: Optional[int] = None,
                 num_incidences: Optional[int] = None, inc_counted=None, edge_counted=None,
                 nx_global: float = 0.0, nmu_global: float = 0.0, program: str = "auto", wave_slots: int = 0,
                 wave_align: int = 0, wave_store_dl: int = 0, wave_generic_rows: int = 0, columns: str = "incidence"):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the ADMM loop only runs on the GPU (no CPU fallback)")
        self.torch = torch
        self.lib = load_library()
        self.g = graph
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.dtype_code = {"f64": F64, "f32": F32}[state_dtype]
        self.tdtype = torch.float64 if state_dtype == "f64" else torch.float32
        g = graph
        ni_owned = int(g.inc_ptr[-1])
        self.NI = int(num_incidences) if num_incidences is not None else ni_owned
        if columns not in ("incidence", "edge"):
            raise ValueError("columns must be 'incidence' or 'edge'")
        self.edge_major = columns == "edge"
        tail, head = g.edge_inc_tail.astype(np.int32), g.edge_inc_head.astype(np.int32)
        self.col_of = np.arange(self.NI, dtype=np.int64)          # incidence column -> state column
        if self.edge_major:
            E_ = g.num_edges
            if self.NI != 2 * E_:
                raise ValueError("edge-major columns need exactly two columns per edge")
            self.col_of = np.empty(self.NI, dtype=np.int64)
            self.col_of[tail] = np.arange(E_); self.col_of[head] = E_ + np.arange(E_)
            tail, head = np.arange(E_, dtype=np.int32), (E_ + np.arange(E_)).astype(np.int32)
            if inc_counted is not None:
                ic_new = np.empty(self.NI, dtype=np.uint8)
                ic_new[self.col_of] = np.asarray(inc_counted, dtype=np.uint8)
                inc_counted = ic_new
        self._keep = [np.ascontiguousarray(a) for a in (
            g.inc_ptr.astype(np.int32), g.inc_edge.astype(np.int32), g.inc_out.astype(np.int32),
            tail, head, g.poly_ptr.astype(np.int32),
            g.poly_A.astype(np.float64), g.poly_b.astype(np.float64), g.interior.astype(np.float64))]
        k = self._keep
        ic = np.ascontiguousarray(inc_counted, dtype=np.uint8) if inc_counted is not None else None
        ec = np.ascontiguousarray(edge_counted, dtype=np.uint8) if edge_counted is not None else None
        self._keep += [ic, ec]
        desc = GraphDesc(g.n, g.num_vertices, g.num_edges, self.NI, _np_ptr(k[0]), _np_ptr(k[1]), _np_ptr(k[2]),
                         _np_ptr(k[3]), _np_ptr(k[4]), _np_ptr(k[5]), _np_ptr(k[6]), _np_ptr(k[7]), _np_ptr(k[8]),
                         g.src, g.dst, self.dtype_code, self.device_index,
                         _np_ptr(ic) if ic is not None else None, _np_ptr(ec) if ec is not None else None,
                         float(nx_global), float(nmu_global),
                         {"auto": 0, "wavefront": 1, "workgroup": 2, "workgroup256": 3}[program], int(wave_slots), int(wave_align),
                         int(wave_store_dl), int(wave_generic_rows), int(self.edge_major))
        h = C.c_void_p()
        st = self.lib.gcsadmm_create(C.byref(desc), C.byref(h))
        if st != 0:
            raise GcsAdmmError(f"gcsadmm_create failed ({st}): {self.lib.gcsadmm_last_error(None).decode()}")
        self.h = h
        c, E, V, n = g.c, g.num_edges, g.num_vertices, g.n
        z = lambda *s, dt=self.tdtype: torch.zeros(*s, dtype=dt, device=self.device)
        # (at least one column each: an empty tensor has a null data pointer, which the ABI rejects)
        self.copy, self.mu, self.zedge = z(c, max(self.NI, 1)), z(c, max(self.NI, 1)), z(c, max(E, 1))
        self.xv, self.zv, self.yv = z(V, 2 * n, dt=torch.float64), z(V, 2 * n, dt=torch.float64), z(V, dt=torch.float64)
        self.sums = z(5, dt=torch.float64)
        self._cost = z(1, dt=torch.float64)
        self.state = State(self.copy.data_ptr(), self.mu.data_ptr(), self.zedge.data_ptr(),
                           self.xv.data_ptr(), self.zv.data_ptr(), self.yv.data_ptr())
        self.params = None
        self.trace = None

    # ------------------------------------------------------------------
    def _check(self, st, what):
        if st != 0:
            raise GcsAdmmError(f"{what} failed ({st}): {self.lib.gcsadmm_last_error(self.h).decode()}")

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gcsadmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def reset(self, rho=1.0, tau_incr=2.0, tau_decr=2.0, nu=10.0, it_rho_limit=100, max_it=1000, eps_abs=1e-4,
              eps_rel=1e-3, eps_edge=1e-4, ipm_tol=None, ipm_max_iter=60, zero_state=True, cold_start=False):
        """Start a loop with the reference's literals as defaults (admm_solver_v3.py:621-651).  ``cold_start``: every vertex
        solve starts from the fixed interior point instead of the record its previous solve left (csrc/warm_start.h)."""
        if ipm_tol is None:
            from . import IPM_TOL as ipm_tol      # the package default (gcs_admm_amd/__init__.py)
        self.params = Params(rho, tau_incr, tau_decr, nu, it_rho_limit, max_it, eps_abs, eps_rel, eps_edge,
                             ipm_tol, ipm_max_iter, 1 if cold_start else 0)
        if zero_state:
            for t in (self.copy, self.mu, self.zedge, self.xv, self.zv, self.yv):
                t.zero_()
        self.trace = self.torch.zeros(max_it, 6, dtype=self.torch.float64, device=self.device)
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_reset(self.h, C.byref(self.params), self._stream()), "gcsadmm_reset")

    # (every call is made with the handle's device current: the stream handed over is torch's current stream OF THAT DEVICE, and a
    #  null stream handle is bound by HIP to whatever device is current)
    def vertex_step(self):
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_vertex_step(self.h, C.byref(self.state), self._stream()), "gcsadmm_vertex_step")

    def edge_step(self):
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_edge_step(self.h, C.byref(self.state), C.c_void_p(self.sums.data_ptr()),
                                                   self._stream()), "gcsadmm_edge_step")
        return self.sums

    def control(self, sums=None):
        s = self.sums if sums is None else sums
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_control(self.h, C.c_void_p(s.data_ptr()), C.c_void_p(self.trace.data_ptr()),
                                                 self._stream()), "gcsadmm_control")

    def enqueue(self, k: int):
        """k iterations back to back, no host synchronisation."""
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_run(self.h, C.byref(self.state), int(k), C.c_void_p(self.trace.data_ptr()),
                                             self._stream()), "gcsadmm_run")

    def enqueue_timed(self, k: int):
        vm, em = C.c_float(0), C.c_float(0)
        vl, el = C.c_int32(0), C.c_int32(0)
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_run_timed(self.h, C.byref(self.state), int(k), C.c_void_p(self.trace.data_ptr()),
                                                   self._stream(), C.byref(vm), C.byref(vl), C.byref(em), C.byref(el)),
                        "gcsadmm_run_timed")
        return dict(vertex_ms=vm.value, vertex_launches=vl.value, edge_ms=em.value, edge_launches=el.value)

    def vertex_prox(self, q, c, ipm_tol: float = 1e-10, ipm_max_iter: int = 60):
        """The x-update of the reference's vertex-edge splits (admm_solver_v1.py:334-383) for every vertex: q, c are
        [V, 4n+1] (weights and centres of the separable quadratic on (x_v, z_v, y_v)).  Returns (xv, zv, yv, failures)."""
        torch = self.torch
        V, n = self.g.num_vertices, self.g.n
        qd = torch.as_tensor(np.ascontiguousarray(q, dtype=np.float64), device=self.device)
        cd = torch.as_tensor(np.ascontiguousarray(c, dtype=np.float64), device=self.device)
        assert qd.shape == (V, 4 * n + 1) and cd.shape == qd.shape
        xv = torch.zeros(V, 2 * n, dtype=torch.float64, device=self.device); zv = torch.zeros_like(xv)
        yv = torch.zeros(V, dtype=torch.float64, device=self.device)
        fails = C.c_int32(0)
        self._check(self.lib.gcsadmm_vertex_prox(self.h, C.c_void_p(qd.data_ptr()), C.c_void_p(cd.data_ptr()), C.c_void_p(xv.data_ptr()),
                                                 C.c_void_p(zv.data_ptr()), C.c_void_p(yv.data_ptr()), C.c_double(ipm_tol),
                                                 C.c_int32(ipm_max_iter), C.byref(fails), self._stream()), "gcsadmm_vertex_prox")
        return xv, zv, yv, fails.value

    # ---- vertex partition across GPUs -------------------------------------------------
    def unique_id(self) -> bytes:
        """128 bytes naming a new RCCL communicator (rank 0 creates it, the host distributes it)"""
        buf = (C.c_ubyte * 128)()
        st = self.lib.gcsadmm_comm_unique_id(buf)
        if st != 0:
            raise GcsAdmmError(f"gcsadmm_comm_unique_id failed ({st}): {self.lib.gcsadmm_last_error(None).decode()}")
        return bytes(buf)

    def _halo_desc(self, send_idx, recv_idx):
        peers, ptr, sc, rc = halo_arrays(send_idx, recv_idx)
        if self.edge_major:      # the lists are written in incidence columns
            sc = np.ascontiguousarray(self.col_of[sc].astype(np.int32)); rc = np.ascontiguousarray(self.col_of[rc].astype(np.int32))
        self._halo_keep = (peers, ptr, sc, rc)
        return HaloDesc(len(peers), _np_ptr(peers), _np_ptr(ptr), _np_ptr(sc), _np_ptr(ptr), _np_ptr(rc))

    def check_halo(self, rank: int, world: int, send_idx, recv_idx):
        """The local checks of ``attach_comm`` alone (no collective): call on every rank and agree on the outcome first."""
        self._check(self.lib.gcsadmm_check_halo(self.h, int(rank), int(world), C.byref(self._halo_desc(send_idx, recv_idx))), "gcsadmm_check_halo")

    def attach_comm(self, rank: int, world: int, unique_id, send_idx, recv_idx):
        """Join the communicator (collective) and upload this partition's halo lists.  ``unique_id`` None: no communicator (the
        host moves the packed halo itself; ``enqueue_partitioned`` then works for world 1 only, without an all-reduce)."""
        hd = self._halo_desc(send_idx, recv_idx)
        idb = (C.c_ubyte * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_attach_comm(self.h, int(rank), int(world), idb, C.byref(hd)), "gcsadmm_attach_comm")
        self.has_comm = unique_id is not None

    def enqueue_partitioned(self, k: int):
        """k iterations of the partitioned loop back to back on the current stream (every rank enqueues the same k)"""
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_run_partitioned(self.h, C.byref(self.state), int(k), C.c_void_p(self.trace.data_ptr()),
                                                         self._stream()), "gcsadmm_run_partitioned")

    def enqueue_partitioned_timed(self, k: int):
        """k iterations of the partitioned loop with every stage bracketed by HIP events (collective); device ms per stage"""
        v, hl, e, r = (C.c_float(0) for _ in range(4))
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_run_partitioned_timed(self.h, C.byref(self.state), int(k), C.c_void_p(self.trace.data_ptr()),
                                                               self._stream(), C.byref(v), C.byref(hl), C.byref(e), C.byref(r)),
                        "gcsadmm_run_partitioned_timed")
        return dict(vertex_ms=v.value, halo_ms=hl.value, edge_ms=e.value, reduce_ms=r.value)

    def set_overlap(self, mode: int = 0) -> int:
        """Schedule of the partitioned loop (include/gcsadmm.h gcsadmm_set_overlap): 0 automatic (overlapped when the partition has
        neighbours), 1 overlapped even without neighbours (tests), 2 serial.  Returns the number of boundary wavefronts (0: serial)."""
        n = C.c_int32(0)
        with self.torch.cuda.device(self.device):
            self._check(self.lib.gcsadmm_set_overlap(self.h, int(mode), C.byref(n)), "gcsadmm_set_overlap")
        return n.value

    def comm_count(self) -> int:
        """ranks of the attached RCCL communicator as RCCL reports them (0: none attached)"""
        n = C.c_int32(0)
        self._check(self.lib.gcsadmm_comm_count(self.h, C.byref(n)), "gcsadmm_comm_count")
        return n.value

    def halo_pack(self):
        self._check(self.lib.gcsadmm_halo_pack(self.h, C.byref(self.state), self._stream()), "gcsadmm_halo_pack")

    def halo_unpack(self):
        self._check(self.lib.gcsadmm_halo_unpack(self.h, C.byref(self.state), self._stream()), "gcsadmm_halo_unpack")

    def halo_exchange(self):
        self._check(self.lib.gcsadmm_halo_exchange(self.h, C.byref(self.state), self._stream()), "gcsadmm_halo_exchange")

    def halo_buffers(self):
        """(send pointer, receive pointer, elements) of the packed halo buffers (device memory owned by the handle)"""
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_int64(0)
        self._check(self.lib.gcsadmm_halo_buffers(self.h, C.byref(a), C.byref(b), C.byref(n)), "gcsadmm_halo_buffers")
        return a.value, b.value, n.value

    def solve_partitioned(self, chunk: int = 25, **params):
        """The partitioned loop to its stop test: as ``solve`` but through gcsadmm_run_partitioned (collective)."""
        self.reset(**params)
        max_it = self.params.max_it
        done = 0
        while True:
            k = min(chunk, max_it - done)
            if k > 0:
                self.enqueue_partitioned(k)
                done += k
            cb = self.read_control()
            if cb.status != RUNNING or done >= max_it:
                return cb

    def read_control(self) -> ControlBlock:
        cb = ControlBlock()
        self._check(self.lib.gcsadmm_read_control(self.h, C.byref(cb), self._stream()), "gcsadmm_read_control")
        return cb

    def query(self):
        a, b, c, d, e = (C.c_int32(0) for _ in range(5))
        self._check(self.lib.gcsadmm_query(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)), "gcsadmm_query")
        return dict(num_waves=a.value, lds_bytes=b.value, num_special=c.value, num_workgroup_vertices=d.value,
                    workgroup_lds_bytes=e.value)

    def unit_iterations(self):
        """Newton iterations of the last vertex step per dispatch unit (empty for handles with fewer than 512 units)."""
        import numpy as np
        with self.torch.cuda.device(self.device):
            n = C.c_int32(0)
            self._check(self.lib.gcsadmm_unit_iterations(self.h, None, 0, C.byref(n), self._stream()), "gcsadmm_unit_iterations")
            out = np.zeros(n.value, np.int32)
            if n.value:
                self._check(self.lib.gcsadmm_unit_iterations(self.h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n), self._stream()),
                            "gcsadmm_unit_iterations")
        return out

    def cost(self) -> float:
        eps = self.params.eps_edge if self.params is not None else 1e-4
        self._check(self.lib.gcsadmm_cost(self.h, C.byref(self.state), C.c_double(eps), C.c_void_p(self._cost.data_ptr()),
                                          self._stream()), "gcsadmm_cost")
        return float(self._cost.item())

    # ------------------------------------------------------------------
    def solve(self, chunk: int = 25, timed: bool = False, **params):
        """Run the loop to its stop test (or max_it), polling the device control
        block every ``chunk`` iterations; returns the reference's record fields
        (utils.py:212-229 naming) plus solver statistics.

        ``timed``: bracket every kernel launch with HIP events (gcsadmm_run_timed) and report ``device_time_s``, the
        summed device time of the vertex-step and edge-step kernels -- the counterpart of the reference's
        ``solve_time``, which adds up SolveInParallel and the edge loop only (admm_solver_v3.py:489-491, 579-585,
        660, 677).  ``wall_time_s`` is the host wall time of the loop either way."""
        import time
        self.reset(**params)
        max_it = self.params.max_it
        done = 0
        dev_ms = 0.0
        self.torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        while True:
            k = min(chunk, max_it - done)
            if k > 0:
                if timed:
                    tm = self.enqueue_timed(k)
                    dev_ms += tm["vertex_ms"] + tm["edge_ms"]
                else:
                    self.enqueue(k)
                done += k
            cb = self.read_control()
            if cb.status != RUNNING or done >= max_it:
                break
        wall = time.perf_counter() - t0
        it = cb.it
        k = min(it, max_it)
        tr = self.trace[:k].cpu().numpy()
        return dict(iterations=int(it), status=STATUS_NAME[cb.status],
                    rho_seq=np.concatenate([[self.params.rho], tr[:, 0]]),
                    pri_res_seq=np.concatenate([[0.0], tr[:, 1]]),
                    dual_res_seq=np.concatenate([[0.0], tr[:, 2]]),
                    eps_pri_seq=tr[:, 3], eps_dual_seq=tr[:, 4],
                    inner_failures=int(tr[:, 5].sum()), cost=self.cost(),
                    wall_time_s=wall, device_time_s=(dev_ms * 1e-3 if timed else None))
