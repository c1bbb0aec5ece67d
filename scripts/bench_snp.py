"""Config-4-shaped run (binomial lasso on a 2-bit SNP design), scaled by argv: n p [L]."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
n, p = int(sys.argv[1]), int(sys.argv[2])
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rng = np.random.default_rng(0)
t0 = time.perf_counter()
# calldata: P(1)=0.25, P(2)=0.05, P(missing)=0.10 (adelie.data.snp_unphased defaults), generated column-block-wise
cd = np.empty((n, p), dtype=np.int8, order="F")
for j0 in range(0, p, 1000):
    u = rng.random((n, min(1000, p - j0)), dtype=np.float32)
    blk = np.zeros(u.shape, dtype=np.int8)
    blk[u < 0.25] = 1
    blk[(u >= 0.25) & (u < 0.30)] = 2
    blk[u >= 0.90] = -9
    cd[:, j0:j0 + blk.shape[1]] = blk
imp = ad.matrix.compute_impute(cd)
beta = rng.standard_normal(p) * (rng.random(p) < 0.05 * min(1, 2000 / p))
eta = np.zeros(n)
for j in np.flatnonzero(beta):
    c = cd[:, j].astype(np.float64); c[c < 0] = imp[j]
    eta += c * beta[j]
eta = (eta - eta.mean()) / max(eta.std(), 1e-9)
y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
print("gen %.1fs" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
X = ad.matrix.snp_calldata(cd, imp)
print("upload+pack %.2fs" % (time.perf_counter() - t0), flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    st = ad.grpnet(X, ad.glm.binomial(y), early_exit=False, lmda_path_size=L)
    el = time.perf_counter() - t0
    print("path %.2fs  nsol %d err %r dev %.3f active %d screen %d" % (el, len(st.lmdas), st.error, st.devs[-1] if len(st.devs) else -1, st.active_set_size, len(st.screen_set)))
    print("  ", {k: round(v, 1) for k, v in st.timers.items()}, st.counters["n_irls_iters"], st.counters["n_updates"])
