"""Re-runs one case of scripts/fuzz_parity.py (same RNG stream): python scripts/fuzz_case.py <case> <seed> [hip|oracle|both]"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import adelie_amd as ad
from oracle import oracle
case, seed0 = int(sys.argv[1]), int(sys.argv[2]); which = sys.argv[3] if len(sys.argv) > 3 else "both"
rng = np.random.RandomState(seed0 * 1000 + case)
n = int(rng.choice([37, 150, 513, 1200, 2049])); p = int(rng.choice([40, 130, 300, 777]))
BIG = bool(os.environ.get("FUZZ_BIG"))
if BIG:
    n = int(rng.choice([900, 2500, 4100])); p = int(rng.choice([1200, 2000]))
fam = rng.choice(["gaussian", "gaussian", "binomial"]); kind = rng.choice(["dense", "dense", "snp"])
dtype = np.float64 if rng.uniform() < 0.75 else np.float32
grouped = rng.uniform() < 0.5
if grouped:
    sizes = []
    while sum(sizes) < p:
        sizes.append(int(rng.choice([1, 1, 2, 3, 7, 16, 33])))
    sizes[-1] -= sum(sizes) - p
    if sizes[-1] <= 0:
        sizes.pop(); sizes[-1] += p - sum(sizes)
    groups = np.concatenate([[0], np.cumsum(sizes)[:-1]])
else:
    groups = np.arange(p)
alpha = float(rng.choice([1.0, 0.7, 0.3])); intercept = bool(rng.uniform() < 0.8); use_w = rng.uniform() < 0.5
min_nv = int(rng.choice([1, 1, 64, 256])); bsz = rng.choice(["", "32", "64", "128"])
if BIG:
    min_nv, bsz = 256, ""
os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = str(min_nv)
if bsz: os.environ["ADELIE_HIP_PANEL_BSZ"] = bsz
if kind == "dense":
    Xh = np.asfortranarray(rng.normal(size=(n, p)).astype(dtype)); mk = lambda mod: mod.dense(Xh); Xd = Xh.astype(np.float64)
else:
    cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.6, 0.22, 0.1, 0.08]); imp = ad.matrix.compute_impute(cd)
    mk = lambda mod: mod.snp_calldata(cd, imp, dtype=dtype); Xd = np.where(cd < 0, imp[None, :], cd).astype(np.float64)
beta = rng.normal(size=p) * (rng.uniform(size=p) < 0.15); eta = Xd @ beta; eta = eta / max(eta.std(), 1e-9)
w = rng.uniform(0.2, 1.8, size=n) if use_w else None
if fam == "gaussian":
    y = eta + rng.normal(size=n); glm = lambda: ad.glm.gaussian(y, weights=w, dtype=dtype)
else:
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64); glm = lambda: ad.glm.binomial(y, weights=w, dtype=dtype)
f32 = dtype == np.float32
kw = dict(groups=groups, alpha=alpha, intercept=intercept, early_exit=False, lmda_path_size=int(rng.choice([8, 15])),
          min_ratio=float(rng.choice([0.3, 0.05])) if not BIG else 0.02, tol=1e-7 if f32 else 1e-13)
kw["max_iters"] = 4000
if fam == "binomial": kw["irls_tol"] = 1e-6 if f32 else 1e-10
kw.update(eval(os.environ.get("FUZZ_KW", "{}")))
print(n, p, fam, kind, dtype.__name__, "grouped", grouped, "alpha", alpha, "min_nv", min_nv, "bsz", bsz, kw["lmda_path_size"], kw["min_ratio"], flush=True)
for name, mod in [("hip", ad.matrix), ("oracle", oracle)]:
    if which not in (name, "both"): continue
    t0 = time.time()
    st = ad.grpnet(mk(mod), glm(), **kw)
    print(name, "%.2fs" % (time.time() - t0), "err", repr(st.error), "nsol", len(st.lmdas), "active", st.active_set_size, {k: st.counters[k] for k in ["n_cd_passes_screen", "n_cd_passes_active", "n_updates", "n_panel_blocks"]}, flush=True)
if which == "cmp":
    a = ad.grpnet(mk(ad.matrix), glm(), **kw); b = ad.grpnet(mk(oracle), glm(), **kw)
    A, B = a.betas.toarray(), b.betas.toarray()
    ww = (w / w.sum()) if w is not None else np.full(n, 1.0 / n)
    pen = np.sqrt(np.diff(np.append(groups, p)).astype(float))
    def obj(bv, b0, lm):
        r = y - Xd @ bv - b0
        val = 0.5 * np.sum(ww * r * r)
        for g0, g1, pk in zip(groups, np.append(groups[1:], p), pen):
            nb = np.linalg.norm(bv[g0:g1]); val += lm * pk * (alpha * nb + 0.5 * (1 - alpha) * nb * nb)
        return val
    for l in range(min(len(a.lmdas), len(b.lmdas))):
        oa, ob = obj(A[l], a.intercepts[l], a.lmdas[l]), obj(B[l], b.intercepts[l], b.lmdas[l])
        print("l", l, "lmda %.4g" % a.lmdas[l], "max|db| %.2e" % np.abs(A[l] - B[l]).max(), "nnz", (A[l] != 0).sum(), (B[l] != 0).sum(), "obj hip-oracle %.3e" % (oa - ob), "(obj %.6g)" % ob)
    print("passes", a.counters["n_cd_passes_active"], b.counters["n_cd_passes_active"], a.counters["n_cd_passes_screen"], b.counters["n_cd_passes_screen"], "updates", a.counters["n_updates"], b.counters["n_updates"])
