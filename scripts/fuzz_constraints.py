"""Randomised parity sweep of constrained paths: HIP against the CPU oracle on random shapes, families, group layouts and
mixes of box / lower / upper / one-sided constraints on the groups of one coefficient (FUZZ_N cases, default 40).
FUZZ_GROUPS=1: also objects on the groups of several coefficients (solved on the device by kernels_cons.hip), with bounds that
are zero, tiny, large or absent per coefficient; the run then also checks that no such group was visited on the host."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import adelie_amd as ad
from adelie_amd import constraint
from oracle import oracle

N = int(os.environ.get("FUZZ_N", "40"))
bad = 0
ONLY = [int(x) for x in os.environ["FUZZ_ONLY"].split(",")] if os.environ.get("FUZZ_ONLY") else None
for case in (ONLY if ONLY is not None else range(N)):
    rng = np.random.RandomState(1000 + case)
    n = int(rng.choice([120, 400, 900]))
    p = int(rng.choice([40, 150, 420]))
    dtype = np.float64 if rng.uniform() < 0.8 else np.float32
    X = np.asfortranarray(rng.normal(size=(n, p)) + 0.4 * rng.normal(size=(n, 1))).astype(dtype)
    beta = rng.normal(size=p) * (rng.uniform(size=p) < 0.15)
    eta = X.astype(np.float64) @ beta
    fam = rng.choice(["gaussian", "binomial", "poisson"])
    if fam == "gaussian":
        y = eta + rng.normal(size=n)
        glm = lambda: ad.glm.gaussian(y.astype(dtype), dtype=dtype)
    elif fam == "binomial":
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta / (1 + np.abs(eta).max() / 4)))).astype(dtype)
        glm = lambda: ad.glm.binomial(y, dtype=dtype)
    else:
        y = rng.poisson(np.exp(np.clip(eta / (1 + np.abs(eta).max() / 2), -3, 3))).astype(dtype)
        glm = lambda: ad.glm.poisson(y, dtype=dtype)
    grouped = rng.uniform() < 0.4
    if grouped:
        groups, j = [], 0
        while j < p:
            groups.append(j)
            j += int(rng.choice([1, 1, 2, 5]))
        groups = np.array(groups)
    else:
        groups = np.arange(p)
    sizes = np.diff(np.append(groups, p))

    def make():
        r2 = np.random.RandomState(77 + case)
        out = []
        for q in sizes:
            k = r2.randint(0, 6) if q == 1 else 0
            if q > 1 and os.environ.get("FUZZ_GROUPS") and r2.uniform() < 0.6:
                bnd = lambda: r2.choice([0.0, 0.02, 0.3, 1e100], size=q, p=[0.25, 0.3, 0.3, 0.15]).astype(dtype)  # noqa: E731
                if r2.uniform() < 0.5:
                    out.append(constraint.box(-bnd(), bnd()))
                else:
                    out.append(constraint.one_sided(r2.choice([-1.0, 1.0], size=q).astype(dtype), bnd()))
                continue
            if k <= 1:
                out.append(None)
            elif k == 2:
                out.append(constraint.lower(np.array([-r2.choice([0.0, 0.05, 0.3])], dtype=dtype)))
            elif k == 3:
                out.append(constraint.upper(np.array([r2.choice([0.0, 0.05, 0.3])], dtype=dtype)))
            elif k == 4:
                out.append(constraint.box(np.array([-r2.uniform(0, 0.2)], dtype=dtype), np.array([r2.uniform(0, 0.2)], dtype=dtype)))
            else:
                out.append(constraint.one_sided(np.array([r2.choice([-1.0, 1.0])], dtype=dtype), np.array([r2.uniform(0, 0.2)], dtype=dtype)))
        return out

    tol = float(os.environ.get("FUZZ_TOL", "1e-12")) if dtype == np.float64 else 1e-7
    kw = dict(groups=groups, alpha=float(rng.choice([1.0, 0.7])), lmda_path_size=int(rng.choice([6, 15])),
              min_ratio=float(rng.choice([0.05, 0.2])), early_exit=False, tol=tol, intercept=bool(rng.uniform() < 0.8),
              progress_bar=False)
    if fam != "gaussian":
        kw["irls_tol"] = float(os.environ.get("FUZZ_IRLS_TOL", "1e-10")) if dtype == np.float64 else 1e-5
    os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = str(rng.choice([1, 256]))
    try:
        ref = ad.grpnet(oracle.dense(X), glm(), constraints=make(), **kw)
        st = ad.grpnet(ad.matrix.dense(X), glm(), constraints=make(), **kw)
        lim = 1e-6 if dtype == np.float64 else 5e-3
        db = np.abs(st.betas.toarray() - ref.betas.toarray()).max() if len(st.lmdas) == len(ref.lmdas) else np.inf
        dd = np.abs((st.duals - ref.duals)).max() if st.duals.shape == ref.duals.shape and st.duals.shape[1] else 0.0
        ok = st.error == ref.error and db < lim and dd < 10 * lim
        if os.environ.get("FUZZ_GROUPS"):
            ok = ok and st.counters["n_host_cons_visits"] == 0
        print(f"{'ok  ' if ok else 'FAIL'} case {case}: n={n} p={p} {fam} {np.dtype(dtype).name} grouped={grouped} G={len(groups)} "
              f"ncons={sum(c is not None for c in make())} dev_visits={st.counters['n_dev_cons_visits']} | max|dbeta| {db:.2e} max|ddual| {dd:.2e} "
              f"err '{st.error[:40]}' / '{ref.error[:40]}'")
        bad += not ok
    except Exception as e:  # noqa: BLE001
        print(f"FAIL case {case}: exception {type(e).__name__}: {e}")
        bad += 1
print(f"{(len(ONLY) if ONLY else N) - bad}/{len(ONLY) if ONLY else N} ok")
sys.exit(1 if bad else 0)
