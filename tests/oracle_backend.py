"""CPU stand-in for DeviceSolver in the multi-rank tests: the oracle as the per-rank compute object
behind the same interface (copy tensor, vertex_step, edge_step, control, read_control).  Tests only."""
from gcs_admm_amd import IPM_TOL
import types

import numpy as np
import torch

from oracle import oracle as O


class OracleBackend:
    def __init__(self, part=None, graph=None, **params):
        if part is not None:
            self.o = O.Oracle(part.graph, ipm_tol=IPM_TOL, num_incidences=part.num_incidences, inc_counted=part.inc_counted,
                              edge_counted=part.edge_counted, nx_global=part.nx_global, nmu_global=part.nmu_global)
        else:
            self.o = O.Oracle(graph, ipm_tol=IPM_TOL)
        self.copy = torch.from_numpy(self.o.copy)          # shares memory with the oracle's array
        self.ap = O.admm_params(**params)
        self.state = np.array([self.ap.rho, 1.0, 1.0, -1.0])
        self.trace = np.zeros((self.ap.max_it, 6))
        self.fails = 0

    def vertex_step(self):
        if self.state[3] != -1:
            return
        self.fails = self.o.vertex_step(self.state[0], self.state[1])

    def edge_step(self):
        if self.state[3] != -1:
            return torch.zeros(5, dtype=torch.float64)
        return torch.from_numpy(self.o.edge_step(self.state[1]))

    def control(self, sums):
        if self.state[3] != -1:
            return
        it = int(self.state[2])
        self.o.control(self.ap, sums.numpy(), self.state, float(self.fails), self.trace[it - 1])

    def read_control(self):
        return types.SimpleNamespace(rho=self.state[0], mu_scale=self.state[1], it=int(self.state[2]), status=int(self.state[3]))

    def cost(self):
        return self.o.cost()
