import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import adelie_amd as ad
from util import make_gaussian
d = make_gaussian(400, 300, seed=11, sparsity=0.5, weights=True)
os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = sys.argv[1] if len(sys.argv) > 1 else "1"
st = ad.grpnet(ad.matrix.dense(np.asfortranarray(d["X"])), ad.glm.gaussian(d["y"], weights=d["weights"]), early_exit=False, lmda_path_size=5, min_ratio=0.5)
print(st.error, st.counters)
