import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import adelie_amd as ad
from test_constraint import _group_problem
from util import make_gaussian
n, p, G = 400, 120, 30
d = make_gaussian(n, p, seed=21, sparsity=0.7)
groups, sizes, spec, make = _group_problem(p, np.random.RandomState(9), G)
print("sizes", sizes, file=sys.stderr)
st = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), constraints=make(), groups=groups, alpha=0.7, early_exit=False,
               lmda_path_size=10, min_ratio=0.05, tol=1e-13, progress_bar=False)
print(st.error, len(st.lmdas))
