"""KKT diagnostics and sparse / concatenated designs at headline scale (100k x 10k): timings + consistency checks."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scipy.sparse as sp
import adelie_amd as ad
from adelie_amd import diagnostic as dg
from bench import make_data

n, p = 100000, 10000
X, y = make_data(n, p, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
st = ad.grpnet(Xd, ad.glm.gaussian(y), early_exit=False, progress_bar=False)
t0 = time.perf_counter(); d = dg.diagnostic(st); t1 = time.perf_counter()
L = len(st.lmdas)
print(f"diagnostic(state): L={L}  {t1 - t0:.2f} s")
t0 = time.perf_counter(); g = dg.gradients(Xd, d.residuals); t1 = time.perf_counter()
one = np.empty(p); ones = np.ones(n)
t2 = time.perf_counter()
for l in range(L):
    Xd.mul(d.residuals[l], ones, one)
t3 = time.perf_counter()
print(f"gradients: batched {t1 - t0:.3f} s   one X.mul per vector {t3 - t2:.3f} s   last row max diff {np.max(np.abs(g[-1] - one)):.2e}")
lm = np.asarray(st.lmdas)
print("max score / lmda over the path:", float(np.max(d.gradient_scores / lm[:, None])))
# sparse at scale: 1 % density
rng = np.random.RandomState(0)
M = sp.random(n, p, density=0.01, format="csc", random_state=rng, dtype=np.float64)
t0 = time.perf_counter(); Xs = ad.matrix.sparse(M); t1 = time.perf_counter()
v = rng.normal(size=n); out = np.empty(p)
Xs.mul(v, ones, out)
print(f"sparse 1%: create {t1 - t0:.2f} s  nnz {M.nnz}  mul max diff {np.max(np.abs(out - M.T @ v)):.2e}")
t0 = time.perf_counter(); Xc = ad.matrix.concatenate([Xd, Xs], axis=1); t1 = time.perf_counter()
out2 = np.empty(2 * p); Xc.mul(v, ones, out2)
ref = np.empty(p); Xd.mul(v, ones, ref)
print(f"concatenate 2 x (100k x 10k): {t1 - t0:.2f} s  consistent {np.array_equal(out2[:p], ref) and np.array_equal(out2[p:], out)}")
