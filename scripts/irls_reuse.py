"""Config 4 (binomial lasso on a 2-bit SNP design) with diagonal blocks reused across IRLS iterations (ADELIE_HIP_IRLS_REUSE=theta):
wall time, block builds and the solution against theta = 0.   python scripts/irls_reuse.py [n p [thetas...]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adelie_amd as ad
from bench import make_snp_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
thetas = [float(x) for x in sys.argv[3:]] or [0.0, 0.001, 0.01, 0.05, 0.2]
dev = torch.device("cuda", 0)
cd, imp, y = make_snp_data(n, p, 0, dev)
Xd = ad.matrix.snp_calldata(cd, imp, dtype=np.float64)
del cd
glm = ad.glm.binomial(y)
kw = dict(early_exit=False, lmda_path_size=100, progress_bar=False)
ref = None
for th in thetas:
    os.environ["ADELIE_HIP_IRLS_REUSE"] = str(th)
    os.environ["ADELIE_HIP_TRACE"] = "2"
    if ref is None:
        ad.grpnet(Xd, glm, **kw)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = ad.grpnet(Xd, glm, **kw)
    el = time.perf_counter() - t0
    assert st.error == "", st.error
    B = st.betas.toarray()
    c = st.counters
    line = dict(theta=th, seconds=round(el, 3), irls=c["n_irls_iters"], passes=c["n_cd_passes_screen"] + c["n_cd_passes_active"],
                blocks=c["n_panel_blocks"], built=c["n_panel_grams"], active=int(st.active_set_size), screen=len(st.screen_set))
    if ref is None:
        ref = (B, np.asarray(st.intercepts), np.asarray(st.devs))
    else:
        line.update(max_dbeta=float(np.abs(B - ref[0]).max()), max_dint=float(np.abs(st.intercepts - ref[1]).max()),
                    max_ddev=float(np.abs(np.asarray(st.devs) - ref[2]).max()), beta_scale=float(np.abs(ref[0]).max()))
    print(line, flush=True)
