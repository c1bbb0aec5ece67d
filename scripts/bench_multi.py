"""Multi-response (multigaussian) path on the GPU: 100-lambda grpnet, dense n x p design, K responses.
Prints wall time per path, the sweep / panel-step timers and, optionally, the CPU oracle on a bounded prefix of the path.
    python scripts/bench_multi.py [n p K]        (MULTI_CPU_LMDAS=10 adds the oracle leg)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad  # noqa: E402

n, p, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (100_000, 10_000, 4)
g = torch.Generator(device="cuda").manual_seed(0)
Xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64).T  # F-ordered (n, p)
rng = np.random.RandomState(0)
B = np.zeros((p, K))
nz = rng.choice(p, max(1, p // 20), replace=False)
B[nz] = rng.normal(size=(len(nz), K))
Bt = torch.from_numpy(B).cuda()
Y = (Xt @ Bt).cpu().numpy()
Y += np.sqrt(np.mean(Y ** 2)) * rng.normal(size=(n, K))
X = ad.matrix.dense(Xt)
glm = ad.glm.multigaussian(Y)
kw = dict(early_exit=False, lmda_path_size=100, min_ratio=1e-2, tol=1e-7)
ts = []
for it in range(3):
    t0 = time.perf_counter()
    s = ad.grpnet(X=X, glm=glm, **kw)
    ts.append(time.perf_counter() - t0)
    assert s.error == "", s.error
print(f"multigaussian n={n} p={p} K={K}: {len(s.lmdas)} lambdas, wall {min(ts):.3f} s/path (runs {['%.3f' % t for t in ts]}), "
      f"solver {s.total_time:.3f} s, dev {s.devs[-1]:.4f}, screen {s.screen_set.size} groups, active {s.active_set_size}")
c, t = s.counters, s.timers
print("counters:", {k: c[k] for k in ("n_sweeps", "n_panel_blocks", "n_panel_grams", "n_panel_cols", "n_cd_passes_screen", "n_cd_passes_active")})
print("timers(ms):", {k: round(t[k], 1) for k in ("t_sweep_ms", "t_gram_ms", "t_cd_ms")})
sw_bytes = n * p * 8
print(f"sweep: {t['t_sweep_ms'] / max(t['n_sweep_launches'], 1):.3f} ms per launch for {sw_bytes / 1e9:.2f} GB of X "
      f"-> {sw_bytes / (t['t_sweep_ms'] / max(t['n_sweep_launches'], 1) * 1e-3) / 1e12:.2f} TB/s (all {K} responses in one pass)")
L = int(os.environ.get("MULTI_CPU_LMDAS", "0"))
if L > 0:
    from oracle import oracle as orc
    nt = os.cpu_count() or 1
    Xh = np.asfortranarray(Xt.cpu().numpy())
    Xo = orc.dense(Xh, n_threads=min(nt, 16))
    t0 = time.perf_counter()
    so = ad.grpnet(X=Xo, glm=glm, n_threads=min(nt, 16), early_exit=False, lmda_path=s.lmdas[:L], tol=1e-7)
    dt = time.perf_counter() - t0
    d = np.abs(so.betas.toarray() - s.betas[:L].toarray()).max()
    print(f"cpu oracle ({min(nt, 16)} threads): first {L} lambdas in {dt:.1f} s; max|dbeta| vs HIP {d:.2e}")
