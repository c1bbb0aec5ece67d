import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import adelie_amd as ad
from util import make_gaussian
for (n, p, gs) in ((2000, 400, 10), (3000, 1200, 10), (2000, 600, 3)):
    d = make_gaussian(n, p, seed=1, sparsity=0.95)
    kw = dict(groups=np.arange(0, p, gs), alpha=0.5, early_exit=False, progress_bar=False)
    out = []
    for m in ("1", "0"):
        os.environ["ADELIE_HIP_GROUP_NEXT_CORR"] = m
        st = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
        out.append(st)
    a, b = out
    print(n, p, gs, "max|dbeta| next_corr on vs off:", np.abs(a.betas.toarray() - b.betas.toarray()).max(), "blocks", a.counters["n_panel_blocks"], b.counters["n_panel_blocks"],
          "passes", a.counters["n_cd_passes_active"], b.counters["n_cd_passes_active"])
