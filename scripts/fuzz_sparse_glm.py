"""Randomised parity of the round-5 engines for GLMs on sparse-resident designs and elastic nets on standardized views against
the oracle on the dense (standardized) copy:  python scripts/fuzz_sparse_glm.py [n_cases] [seed]
  * IRLS on a design kept sparse, groups of one: the panel engine over compressed columns (forced from the first screen value on
    with ADELIE_HIP_CD_BLOCK_MIN_NV=1 in half of the cases), f64 / f32, binomial / poisson, weights, offsets, alpha, penalty,
    with and without an intercept, empty columns and rows;
  * the same on the standardized view of the sparse design and of a dense / 2-bit design (lasso / elastic net with an intercept:
    penalty_l2 route on the base design; otherwise the view's own full-Gram engines with the kept Gram)."""
import os, sys, time, warnings
import numpy as np
import scipy.sparse as sp
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import adelie_amd as ad
from oracle import oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ONLY = [int(c) for c in os.environ["FUZZ_ONLY"].split(",")] if os.environ.get("FUZZ_ONLY") else None
warnings.simplefilter("ignore")
bad = 0
t0 = time.time()
for case in range(N):
    if ONLY is not None and case not in ONLY:
        continue
    rng = np.random.RandomState(seed0 * 1000 + case)
    dtype = np.float64 if rng.uniform() < 0.75 else np.float32
    n = int(rng.choice([60, 200, 500, 1200]))
    p = int(rng.choice([20, 90, 300]))
    base_kind = rng.choice(["sparse", "sparse", "dense", "snp"])
    view = bool(rng.uniform() < 0.5) or base_kind != "sparse"
    if base_kind == "snp":
        calls = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.55, 0.3, 0.1, 0.05]).astype(np.int8)
        imp = ad.matrix.compute_impute(calls)
        D = np.where(calls < 0, imp[None], calls).astype(np.float64)
        base = ad.matrix.snp_calldata(calls, imp, dtype=dtype)
    else:
        dens = float(rng.choice([0.03, 0.1, 0.3])) if base_kind == "sparse" else 1.0
        D = rng.normal(size=(n, p)) * (rng.uniform(size=(n, p)) < dens)
        if base_kind == "sparse" and rng.uniform() < 0.3:
            D[:, rng.randint(p)] = 0   # an empty column
            D[rng.randint(n)] = 0      # an empty row
        D = D.astype(dtype).astype(np.float64)
        base = ad.matrix.sparse(sp.csc_matrix(D.astype(dtype)), resident="csc") if base_kind == "sparse" else ad.matrix.dense(np.asfortranarray(D.astype(dtype)))
    Xd = D
    X = base
    if view:
        sd = D.std(axis=0)
        if np.any(sd == 0):
            continue
        X = ad.matrix.standardize(base, lazy=True) if base_kind != "sparse" else ad.matrix.standardize(base)
        Xd = (D - D.mean(axis=0)) / sd
    k = min(p, 8)
    beta = np.zeros(p); beta[rng.choice(p, k, replace=False)] = rng.normal(size=k)
    eta = Xd @ beta
    eta = eta / max(eta.std(), 1e-12)
    fam = rng.choice(["binomial", "poisson"])
    w = rng.uniform(0.2, 1.0, n) if rng.uniform() < 0.5 else None
    if w is not None:
        w = w / w.sum()
    if fam == "binomial":
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(float)
        glm = ad.glm.binomial(y, weights=w, dtype=dtype)
    else:
        y = rng.poisson(np.exp(0.5 * eta)).astype(float)
        glm = ad.glm.poisson(y, weights=w, dtype=dtype)
    kw = dict(early_exit=False, lmda_path_size=int(rng.choice([6, 12])), min_ratio=float(rng.choice([0.2, 0.05])), progress_bar=False,
              alpha=float(rng.choice([1.0, 1.0, 0.5, 0.2])), intercept=bool(rng.uniform() < 0.75))
    G = p
    if rng.uniform() < 0.3:   # groups of 2 or 5 coefficients (every p in the draw is a multiple of 10)
        gsz = int(rng.choice([2, 5]))
        kw["groups"] = np.arange(0, p, gsz)
        G = p // gsz
    if rng.uniform() < 0.4:
        kw["penalty"] = rng.uniform(0.5, 2.0, G).astype(dtype)
    if rng.uniform() < 0.3:
        kw["offsets"] = (0.1 * rng.normal(size=n)).astype(dtype)
    if dtype == np.float64:
        kw.update(tol=1e-12, irls_tol=1e-12)
    else:
        kw.update(newton_tol=1e-5)   # (the default 1e-12 is unreachable in single precision: the root find of a group would end in its error)
    force = rng.uniform() < 0.5
    if force:
        os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = "1"
    else:
        os.environ.pop("ADELIE_HIP_CD_BLOCK_MIN_NV", None)
    try:
        a = ad.grpnet(X, glm, **kw)
        o = ad.grpnet(orc.dense(np.asfortranarray(Xd.astype(dtype))), glm, **kw)
    except Exception as e:  # noqa: BLE001
        print(f"case {case}: EXCEPTION {e!r}  ({base_kind} view={view} {fam} {dtype.__name__} n={n} p={p} {kw.get('alpha')} icpt={kw['intercept']} force={force})")
        bad += 1
        continue
    tol = 2e-6 if dtype == np.float64 else 5e-3
    if a.error != o.error:
        print(f"case {case}: errors differ: {a.error!r} vs {o.error!r}  ({base_kind} view={view} {fam} {dtype.__name__} n={n} p={p} alpha={kw.get('alpha')} "
              f"icpt={kw['intercept']} groups={'groups' in kw} force={force} L={a.betas.shape[0]}/{o.betas.shape[0]})")
        bad += 1
        continue
    L = min(a.betas.shape[0], o.betas.shape[0])
    db = np.abs(a.betas.toarray()[:L] - o.betas.toarray()[:L]).max() if L else 0.0
    di = np.abs(a.intercepts[:L] - o.intercepts[:L]).max() if L else 0.0
    ok = a.betas.shape[0] == o.betas.shape[0] and db < tol and di < tol
    if not ok:
        bad += 1
    print(f"case {case}: {'ok ' if ok else 'BAD'} {base_kind:6s} view={int(view)} {fam:8s} {dtype.__name__} n={n} p={p} alpha={kw['alpha']} icpt={int(kw['intercept'])} groups={len(kw['groups']) if 'groups' in kw else 0} "
          f"force={int(force)} blocks={a.counters['n_panel_blocks']} L={a.betas.shape[0]}/{o.betas.shape[0]} dbeta={db:.1e} dicpt={di:.1e}", flush=True)
print(f"{bad} bad of {N} in {time.time() - t0:.0f} s")
