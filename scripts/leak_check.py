"""Many solves in one process: device memory and host RSS must stay flat (streams, events, host-mapped reports, buffers)."""
import os, sys, numpy as np, torch, psutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = "1"
import adelie_amd as ad
from util import make_gaussian
d = make_gaussian(800, 300, seed=3, sparsity=0.7)
X = ad.matrix.dense(np.asfortranarray(d["X"]))
proc = psutil.Process()
for it in range(301):
    groups = np.arange(0, 300, 3) if it % 2 else np.arange(300)
    fam = ad.glm.gaussian(d["y"]) if it % 3 else ad.glm.binomial((d["y"] > 0).astype(float))
    st = ad.grpnet(X, fam, groups=groups, early_exit=False, lmda_path_size=12, min_ratio=0.1)
    assert st.error == ""
    if it % 100 == 0:
        free, total = torch.cuda.mem_get_info()
        print(it, "device used MB %.1f" % ((total - free) / 2**20), "host RSS MB %.1f" % (proc.memory_info().rss / 2**20), flush=True)
