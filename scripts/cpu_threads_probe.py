"""Probe (run on the GPU box): which OpenMP thread counts make the CPU oracle fastest on the headline workload."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from oracle import oracle
from bench import make_data
n, p = 100000, 10000
X, y = make_data(n, p, 0, torch.device("cuda", 0), torch.float64)
Xh = X.t().contiguous().cpu().numpy().T
glm = ad.glm.gaussian(y)
for nt, ntc in [(32, 4), (32, 8), (64, 8), (64, 16), (128, 8), (16, 16), (8, 8)]:
    os.environ["ORACLE_COL_THREADS"] = str(ntc)
    Xo = oracle.dense(Xh, n_threads=nt)
    t0 = time.perf_counter()
    st = ad.grpnet(Xo, glm, n_threads=nt, early_exit=False, exit_cond=lambda v: v.n_solutions >= 25)
    print(nt, ntc, "25 lambdas:", round(time.perf_counter() - t0, 2), "s", st.counters["n_updates"], flush=True)
