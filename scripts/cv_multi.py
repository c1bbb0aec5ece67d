"""cv_grpnet over a multi-response family: per-lambda losses on the device (adelie_hip_design_multi_path_losses) vs host
predictions + glm.loss.  usage: cv_multi.py [n p K family]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad

n, p, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (100000, 1000, 4)
family = sys.argv[4] if len(sys.argv) > 4 else "multigaussian"
rng = np.random.RandomState(0)
X = np.asfortranarray(rng.normal(size=(n, p)))
S = X[:, :10] @ rng.normal(size=(10, K)) + rng.normal(size=(n, K))
Y = np.eye(K)[np.argmax(S, axis=1)] if family == "multinomial" else S
glm = getattr(ad.glm, family)(y=Y)
Xd = ad.matrix.dense(X)
res = {}
for mode in ("device", "host", "device"):
    saved = dict(ad.cv._MULTI_KINDS)
    if mode == "host":
        ad.cv._MULTI_KINDS.clear()
    t0 = time.perf_counter()
    cv = ad.cv_grpnet(Xd, glm, n_folds=5, seed=0, lmda_path_size=50, progress_bar=False)
    el = time.perf_counter() - t0
    ad.cv._MULTI_KINDS.update(saved)
    print(f"{mode:7s} {el:7.2f} s  best_idx {cv.best_idx}", flush=True)
    res[mode] = cv.losses
print("max |device - host| / |host|:", float(np.max(np.abs(res["device"] - res["host"]) / np.abs(res["host"]))))
