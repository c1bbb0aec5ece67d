import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import adelie_amd as ad
from oracle import oracle
case, seed0 = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed0 * 1000 + case)
n = int(rng.choice([37, 150, 513, 1200, 2049])); p = int(rng.choice([40, 130, 300, 777]))
fam = rng.choice(["gaussian", "gaussian", "binomial"]); kind = rng.choice(["dense", "dense", "snp"])
dtype = np.float64 if rng.uniform() < 0.75 else np.float32
grouped = rng.uniform() < 0.5
assert not grouped and kind == "dense"
groups = np.arange(p)
alpha = float(rng.choice([1.0, 0.7, 0.3])); intercept = bool(rng.uniform() < 0.8); use_w = rng.uniform() < 0.5
min_nv = int(rng.choice([1, 1, 64, 256])); bsz = rng.choice(["", "32", "64", "128"])
Xh = np.asfortranarray(rng.normal(size=(n, p)).astype(dtype))
beta = rng.normal(size=p) * (rng.uniform(size=p) < 0.15)
eta = Xh @ beta; eta = eta / max(eta.std(), 1e-9)
w = rng.uniform(0.2, 1.8, size=n) if use_w else None
y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
L = int(rng.choice([8, 15])); mr = float(rng.choice([0.3, 0.05]))
kw = dict(groups=groups, alpha=alpha, intercept=intercept, early_exit=False, lmda_path_size=L, min_ratio=mr, tol=1e-13, irls_tol=1e-10)
print("n", n, "p", p, "alpha", alpha, "L", L, "mr", mr, "min_nv", min_nv, "bsz", bsz)
for mn, bz in [(min_nv, bsz), (100000, "")]:
    os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = str(mn)
    if bz: os.environ["ADELIE_HIP_PANEL_BSZ"] = bz
    else: os.environ.pop("ADELIE_HIP_PANEL_BSZ", None)
    a = ad.grpnet(ad.matrix.dense(Xh), ad.glm.binomial(y, weights=w), **kw)
    b = ad.grpnet(oracle.dense(Xh), ad.glm.binomial(y, weights=w), **kw)
    A, B = a.betas.toarray(), b.betas.toarray()
    print("min_nv", mn, "err", repr(a.error), repr(b.error), "nsol", len(a.lmdas), len(b.lmdas))
    for l in range(min(len(a.lmdas), len(b.lmdas))):
        print("  l", l, "lmda %.5g" % a.lmdas[l], "max|db| %.3e" % np.abs(A[l] - B[l]).max(), "nnz", (A[l] != 0).sum(), (B[l] != 0).sum(), "dev %.6f %.6f" % (a.devs[l], b.devs[l]))
    print("  active", a.active_set_size, b.active_set_size, "screen", len(a.screen_set), len(b.screen_set), "irls", a.counters["n_irls_iters"], b.counters["n_irls_iters"], "blocks", a.counters["n_panel_blocks"])
