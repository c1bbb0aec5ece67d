"""Panel engine at odd shapes (tiny n, p >> n, f32) against the CPU oracle."""
import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = sys.argv[1] if len(sys.argv) > 1 else "256"
import adelie_amd as ad
from oracle import oracle
from util import make_gaussian, assert_same_path
for (n, p, dtype, mr) in [(50, 1000, np.float64, 1e-2), (130, 900, np.float64, 1e-3), (67, 2000, np.float64, 5e-2), (300, 1500, np.float32, 1e-2), (1, 400, np.float64, 0.5), (2, 600, np.float64, 0.1)]:
    d = make_gaussian(n, p, seed=n + p, sparsity=0.9)
    X = np.asarray(d["X"], dtype=dtype, order="F")
    kw = dict(early_exit=False, lmda_path_size=30, min_ratio=mr, tol=1e-7 if dtype == np.float32 else 1e-13)
    a = ad.grpnet(ad.matrix.dense(X), ad.glm.gaussian(d["y"], dtype=dtype), **kw)
    b = ad.grpnet(oracle.dense(X), ad.glm.gaussian(d["y"], dtype=dtype), **kw)
    try:
        assert_same_path(a, b, 5e-3 if dtype == np.float32 else 1e-6)
        ok = "OK"
    except AssertionError as e:
        ok = "MISMATCH " + str(e)[:200]
    print(n, p, dtype.__name__, "err", repr(a.error), repr(b.error), "nsol", len(a.lmdas), len(b.lmdas), "screen", len(a.screen_set), "active", a.active_set_size, "panel blocks", a.counters["n_panel_blocks"], ok, flush=True)
