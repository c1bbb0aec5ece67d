import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X); glm = ad.glm.gaussian(y)
groups = np.arange(0, 10000, 10)
for nt in [1e-12, 1e-2]:
    for rep in range(2):
        st = ad.grpnet(Xd, glm, groups=groups, alpha=0.5, early_exit=False, newton_tol=nt)
    print("newton_tol", nt, "cd ms %.1f" % st.timers["t_cd_ms"], "blocks", st.counters["n_panel_blocks"], "updates", st.counters["n_updates"], "total %.1f" % (1e3 * st.total_time))
