"""Prints the host-side wall-clock split of one headline path (debug scalars 910..917 of the result)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adelie_amd as ad
from adelie_amd import _abi
from bench import make_data
_abi.S.update({f"t_host_phase{i}": 910 + i for i in range(8)})
import adelie_amd.state as S
S._TIMERS = list(S._TIMERS) + [f"t_host_phase{i}" for i in range(8)]
n, p = 100000, 10000
X, y = make_data(n, p, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
glm = ad.glm.gaussian(y)
for rep in range(2):
    t0 = time.perf_counter()
    st = ad.grpnet(Xd, glm, early_exit=False, lmda_path_size=100)
    el = time.perf_counter() - t0
names = ["screen", "append", "gram+vars", "fit", "invariance", "kkt+solutions", "6", "7"]
print("wall %.1f ms  total_time %.1f ms" % (1e3 * el, 1e3 * st.total_time))
print({names[i]: round(st.timers[f"t_host_phase{i}"], 1) for i in range(8)})
print({k: round(v, 1) for k, v in st.timers.items() if not k.startswith("t_host_phase")})
