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
