import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from adelie_amd import _abi
from bench import make_data
for k in range(8): _abi.S[f"dbg{k}"] = 900 + k
from adelie_amd import state as S
S._TIMERS += [f"dbg{k}" for k in range(8)]
n, p = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
X, y = make_data(n, p, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
glm = ad.glm.gaussian(y)
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
st = ad.grpnet(Xd, glm, early_exit=False, lmda_path_size=L)
st = ad.grpnet(Xd, glm, early_exit=False, lmda_path_size=L)
c = st.counters; t = st.timers
vis = c["n_cd_visits_screen"] + c["n_cd_visits_active"]
print("cd ms", t["t_cd_ms"], "updates", c["n_updates"], "visits", vis, "final active", st.active_set_size)
names = ["fetch-issue", "g-read+update", "barrier1", "col-wait+apply", "barrier2"]
tot = sum(t[f"dbg{k}"] for k in range(5))
for k, nm in enumerate(names):
    print(f"  {nm:16s} {t[f'dbg{k}']:14.0f} cycles  {100*t[f'dbg{k}']/tot:5.1f}%   per-visit {t[f'dbg{k}']/vis:8.1f}")
print("  total cycles", tot, "-> implied clock MHz if cd_ms is all kernel:", tot / (t["t_cd_ms"] * 1e-3) / 1e6)
