"""Vertex partition + halo exchange + all-reduce (gcs_admm_amd/partition.py) on CPU: the partitioned
loop must reproduce the single-partition loop.  The per-rank compute object is the oracle behind the
DeviceSolver interface (tests/oracle_backend.py); the exchange runs over a real 2-process gloo group."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.partition import PartitionedLoop, build_partition, strip_owner

HERE = os.path.dirname(os.path.abspath(__file__))


def _graph(name):
    return lattice_boxes(6, 8, seed=3) if name == "lattice" else load_fixture(name)[1]


@pytest.mark.parametrize("name,world", [("benchmark4", 2), ("benchmark4", 3), ("lattice", 2), ("lattice", 4)])
def test_partition_bookkeeping(name, world):
    g = _graph(name)
    owner = strip_owner(g, world)
    assert set(owner) <= set(range(world))
    parts = [build_partition(g, owner, r, world) for r in range(world)]
    # every vertex owned once; every edge counted once; every incidence (copy) counted once
    assert sum(len(p.vertex_global) for p in parts) == g.num_vertices
    assert sum(int(p.edge_counted.sum()) for p in parts) == g.num_edges
    assert sum(int(p.inc_counted.sum()) for p in parts) == 2 * g.num_edges
    for p in parts:
        lg = p.graph
        assert p.num_incidences == lg.inc_ptr[-1] + sum(len(v) for v in p.recv_idx.values())
        # both columns of every local edge exist; ghost columns are exactly the received ones
        cols = np.concatenate([lg.edge_inc_tail, lg.edge_inc_head])
        assert cols.min() >= 0 and cols.max() < p.num_incidences
        ghosts = np.concatenate(list(p.recv_idx.values())) if p.recv_idx else np.zeros(0, int)
        assert sorted(ghosts.tolist()) == list(range(lg.inc_ptr[-1], p.num_incidences))
        for r, ix in p.send_idx.items():
            assert len(ix) == len(parts[r].recv_idx[p.rank])       # matching message sizes
            assert ix.max() < lg.inc_ptr[-1]
    assert all(p.nx_global == g.nx and p.nmu_global == g.nmu for p in parts)


def _worker(rank, world, name, port, out):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    from oracle_backend import OracleBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = _graph(name)
    part = build_partition(g, strip_owner(g, world), rank, world)
    be = OracleBackend(part=part, max_it=400)
    loop = PartitionedLoop(part, be)
    cb = loop.solve(chunk=20, max_it=400)
    cost = torch.tensor([be.cost()], dtype=torch.float64)
    dist.all_reduce(cost)
    if rank == 0:
        np.savez(out, it=cb.it, status=cb.status, trace=be.trace[:cb.it], cost=cost.item())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["benchmark1", "lattice"])
def test_two_rank_gloo_matches_single_partition(tmp_path, name):
    from oracle_backend import OracleBackend
    g = _graph(name)
    ref = OracleBackend(graph=g, max_it=400)
    # single partition through the same driver interface (no group needed: world of one)
    while ref.state[3] == -1:
        ref.vertex_step(); s = ref.edge_step(); ref.control(s)
    out = str(tmp_path / "r.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, name, port, out), nprocs=2, join=True)
    got = np.load(out)
    it = int(ref.state[2])
    assert int(got["it"]) == it and int(got["status"]) == int(ref.state[3]) == 0
    # same arithmetic per vertex / per edge; only the order of the 5-term sums differs
    assert np.allclose(got["trace"][:, :5], ref.trace[:it, :5], rtol=1e-9, atol=1e-12)
    assert abs(float(got["cost"]) - ref.cost()) <= 1e-9 * abs(ref.cost())


def _halo_worker(rank, world, port, out):
    """the message layout of the C ABI's halo exchange (gcsadmm_halo_desc, csrc/gcsadmm.hip halo_pack_kernel /
    halo_unpack_kernel: per peer one block [c][columns of that peer], peers in ascending rank order), emulated with numpy
    over a real gloo group: after pack -> send/recv -> unpack every ghost column holds the remote owner's copy."""
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    from gcs_admm_amd.solver import halo_arrays
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = lattice_boxes(6, 9, seed=4)
    owner = strip_owner(g, world)
    part = build_partition(g, owner, rank, world)
    c, NI, ni_owned = g.c, part.num_incidences, int(part.graph.inc_ptr[-1])
    peers, ptr, scols, rcols = halo_arrays(part.send_idx, part.recv_idx)
    # owned column k of rank r carries the value 1000 r + global incidence id + w / 16 (w = word): recognisable anywhere
    gptr = g.inc_ptr.astype(np.int64)
    ginc = np.concatenate([np.arange(gptr[v], gptr[v + 1]) for v in part.vertex_global]) if len(part.vertex_global) else np.zeros(0, np.int64)
    copy = np.full((c, NI), np.nan)
    copy[:, :ni_owned] = 1000.0 * rank + ginc[None, :] + np.arange(c)[:, None] / 16.0
    # pack: block of peer p at ptr[p] * c, laid out [c][cnt_p]
    sendbuf = np.empty(c * len(scols)); recvbuf = np.empty(c * len(rcols))
    for p in range(len(peers)):
        lo, hi = int(ptr[p]), int(ptr[p + 1])
        sendbuf[lo * c:hi * c] = copy[:, scols[lo:hi]].ravel()
    ops = []
    ts, tr = torch.from_numpy(sendbuf), torch.from_numpy(recvbuf)
    for p, peer in enumerate(peers):
        lo, hi = int(ptr[p]) * c, int(ptr[p + 1]) * c
        ops.append(dist.P2POp(dist.isend, ts[lo:hi], int(peer)))
        ops.append(dist.P2POp(dist.irecv, tr[lo:hi], int(peer)))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for p in range(len(peers)):
        lo, hi = int(ptr[p]), int(ptr[p + 1])
        copy[:, rcols[lo:hi]] = recvbuf[lo * c:hi * c].reshape(c, hi - lo)
    # every ghost column now holds the owner's value for exactly that (edge, side): recompute it from the global graph
    ok = bool(np.isfinite(copy).all())
    lg = part.graph
    for e_loc, e_glob in enumerate(part.edge_global):
        for col, v_glob, is_head in ((lg.edge_inc_tail[e_loc], g.edge_tail[e_glob], 0), (lg.edge_inc_head[e_loc], g.edge_head[e_glob], 1)):
            gi = g.edge_inc_head[e_glob] if is_head else g.edge_inc_tail[e_glob]
            want = 1000.0 * owner[v_glob] + gi + np.arange(c) / 16.0
            ok = ok and bool(np.array_equal(copy[:, col], want))
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(float(t.item()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_message_layout_over_gloo(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29650 + world
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert res == 1.0
