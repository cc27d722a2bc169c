import sys, os
sys.path.insert(0, os.getcwd())
from gcs_admm_amd import IPM_TOL  # noqa: E402
import numpy as np
from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.solver import DeviceSolver
from oracle.oracle import Oracle
for name in ["benchmark1","benchmark2","benchmark3","benchmark4","test_autogen2"]:
    case,g=load_fixture(name)
    res=DeviceSolver(g,"f64",device=0).solve()
    ora=Oracle(g,ipm_tol=IPM_TOL).run(nthreads=16)
    k=min(len(res["pri_res_seq"]),len(ora["pri_res_seq"]))
    print(name,res["iterations"],ora["iterations"],"max rel pri %.2e dual %.2e"%(np.max(np.abs(res["pri_res_seq"][1:k]-ora["pri_res_seq"][1:k])/ora["pri_res_seq"][1:k]),np.max(np.abs(res["dual_res_seq"][1:k]-ora["dual_res_seq"][1:k])/ora["dual_res_seq"][1:k])),"cost diff %.2e"%abs(res["cost"]-ora["cost"]))
