"""8-fold Gaussian cv_grpnet on a 2-bit SNP design (calldata generated on the host), shared sweeps on / off.  argv: n p"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad

n, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 10000)
rng = np.random.default_rng(0)
calls = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.6, 0.25, 0.1, 0.05])
X = ad.matrix.snp_calldata(calls)
beta = rng.standard_normal(p) * (rng.random(p) < 0.02)
y = np.where(calls < 0, 0, calls).astype(np.float32) @ beta.astype(np.float32) + rng.standard_normal(n)
glm = ad.glm.gaussian(y=y.astype(np.float64))
res = {}
for mode in ("1", "0", "1", "0"):
    os.environ["ADELIE_HIP_SWEEP_BATCH"] = mode
    t0 = time.perf_counter()
    cv = ad.cv_grpnet(X, glm, n_folds=8, seed=0, n_concurrent=8, progress_bar=False)
    print(f"shared sweeps {mode}: {time.perf_counter() - t0:.2f} s  best_idx {cv.best_idx}", flush=True)
    res[mode] = cv.losses
print("max rel diff", float(np.max(np.abs(res["1"] - res["0"]) / np.abs(res["0"]))))
