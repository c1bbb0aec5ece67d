import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from adelie_amd import _abi, state as S
for k in range(8): _abi.S[f"host{k}"] = 910 + k
S._TIMERS += [f"host{k}" for k in range(8)]
from bench import make_data
n, p = 100000, 10000
X, y = make_data(n, p, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X); glm = ad.glm.gaussian(y)
st = ad.grpnet(Xd, glm, early_exit=False)
t0 = time.perf_counter(); st = ad.grpnet(Xd, glm, early_exit=False); el = time.perf_counter() - t0
t = st.timers
print("python wall ms", el * 1e3, "solve total_time ms", st.total_time * 1e3)
for k, nm in enumerate(["screen(host logic)", "append_screen", "gram+vars(wait)", "fit", "invariance", "kkt+solutions"]):
    print(f"  {nm:20s} {t[f'host{k}']:8.1f} ms")
print("  device: sweep %.1f gram %.1f cd %.1f axpy %.1f" % (t["t_sweep_ms"], t["t_gram_ms"], t["t_cd_ms"], t["t_axpy_ms"]))
tf = (st.benchmark_fit_active + st.benchmark_fit_screen) * 1e3
for i in range(0, 100, 9): print(i, "active", st.active_sizes[i], "screen", st.screen_sizes[i], "fit ms %.3f" % tf[min(i + 1, len(tf) - 1)])
