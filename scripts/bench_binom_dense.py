"""Binomial lasso on the dense headline design (IRLS through the panel engine): argv n p [L]."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
n, p = int(sys.argv[1]), int(sys.argv[2]); L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
X, y = make_data(n, p, 0, torch.device("cuda", 0), torch.float64)
yb = (y > np.median(y)).astype(np.float64)
Xd = ad.matrix.dense(X)
for rep in range(2):
    t0 = time.perf_counter()
    st = ad.grpnet(Xd, ad.glm.binomial(yb), early_exit=False, lmda_path_size=L)
    el = time.perf_counter() - t0
    print("path %.2fs nsol %d err %r dev %.4f active %d screen %d" % (el, len(st.lmdas), st.error, st.devs[-1], st.active_set_size, len(st.screen_set)), {k: round(v, 1) for k, v in st.timers.items() if k in ("t_sweep_ms", "t_gram_ms", "t_cd_ms")}, {k: st.counters[k] for k in ["n_irls_iters", "n_updates", "n_panel_blocks", "n_panel_grams"]}, flush=True)
