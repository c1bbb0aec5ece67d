"""Randomised parity sweep of the multi-response path: HIP vs the CPU oracle over random shapes, class counts, feature
groupings, weights, intercepts, penalties and both families.
usage: python scripts/fuzz_multi.py [n_cases] [seed]"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
import adelie_amd as ad
from oracle import oracle
bad = 0
worst = 0.0
t00 = time.time()
ONLY = os.environ.get("FUZZ_ONLY")  # "case[,case...]": re-run single cases; FUZZ_IRLS_TOL / FUZZ_TOL override the tolerances
for case in range(N):
    if ONLY and str(case) not in ONLY.split(","):
        continue
    rng = np.random.RandomState(seed0 * 1000 + case)
    n = int(rng.choice([23, 150, 513, 1200, 2049]))
    p = int(rng.choice([5, 40, 130, 300]))
    K = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16]))
    fam = rng.choice(["multigaussian", "multigaussian", "multinomial"])
    if fam == "multinomial" and K < 2:
        K = 2
    order = "F" if rng.uniform() < 0.7 else "C"
    X = np.array(rng.normal(size=(n, p)), order=order)
    if rng.uniform() < 0.5:
        sizes = []
        while sum(sizes) < p:
            sizes.append(int(rng.choice([1, 1, 2, 3, 7])))
        sizes[-1] -= sum(sizes) - p
        if sizes[-1] <= 0:
            sizes.pop(); sizes[-1] += p - sum(sizes)
        groups = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(int)
    else:
        groups = np.arange(p)
    gs = np.diff(np.concatenate([groups, [p]]))
    if gs.max() * K > 128:
        groups = np.arange(p); gs = np.ones(p, dtype=int)
    alpha = float(rng.choice([1.0, 0.7, 0.3]))
    intercept = bool(rng.uniform() < 0.7)
    w = rng.uniform(0.2, 1.8, size=n) if rng.uniform() < 0.5 else None
    penalty = None
    if rng.uniform() < 0.3:
        penalty = np.sqrt(gs * K) * rng.uniform(0.5, 2.0, size=len(gs))
        penalty[rng.choice(len(gs), max(1, len(gs) // 10), replace=False)] = 0
    B = rng.normal(size=(p, K)) * (rng.uniform(size=(p, 1)) < 0.15)
    eta = X @ B
    eta = eta / max(eta.std(), 1e-9)
    if fam == "multigaussian":
        Y = eta + rng.normal(size=(n, K)) + rng.normal(size=K)
        glm = lambda: ad.glm.multigaussian(Y, weights=w)
    else:
        mu = np.exp(eta); mu /= mu.sum(1)[:, None]
        Y = np.array([rng.multinomial(1, m) for m in mu]).astype(np.float64)
        glm = lambda: ad.glm.multinomial(Y, weights=w)
    kw = dict(groups=groups, alpha=alpha, intercept=intercept, penalty=penalty, early_exit=False,
              lmda_path_size=int(rng.choice([8, 15])), min_ratio=float(rng.choice([0.3, 0.05])), tol=1e-13, max_iters=4000)
    if fam == "multinomial":
        kw["irls_tol"] = float(os.environ.get("FUZZ_IRLS_TOL", "1e-12"))
    if os.environ.get("FUZZ_TOL"):
        kw["tol"] = float(os.environ["FUZZ_TOL"])
    desc = f"case {case}: n={n} p={p} K={K} {fam} order={order} G={len(groups)} alpha={alpha} icpt={intercept} w={w is not None} pen={penalty is not None}"
    if os.environ.get("FUZZ_VERBOSE"):
        print("start " + desc, flush=True)
    t0 = time.time()
    try:
        a = ad.grpnet(ad.matrix.dense(X), glm(), **kw)
        t_hip = time.time() - t0
        b = ad.grpnet(oracle.dense(np.asfortranarray(X)), glm(), **kw)
        if a.error or b.error:
            ok = a.error == b.error
            msg = f"errors {a.error!r} / {b.error!r}"
        else:
            d = max(np.abs(a.betas.toarray() - b.betas.toarray()).max() if a.betas.shape[1] else 0.0,
                    np.abs(a.intercepts - b.intercepts).max())
            if len(a.lmdas) == len(b.lmdas):  # (a path at lmda_max == 0 -- every group unpenalised -- has lmdas 0 on both sides)
                la_, lb_ = np.asarray(a.lmdas, dtype=float), np.asarray(b.lmdas, dtype=float)
                dl = np.abs(la_ - lb_).max() / max(np.abs(lb_).max(), 1e-300) if len(la_) else 0.0
            else:
                dl = np.inf
            worst = max(worst, d)
            ok = len(a.lmdas) == len(b.lmdas) and d < 1e-6 and dl < 1e-9
            msg = f"max|dbeta|={d:.2e} dlmda={dl:.1e} hip {t_hip:.2f}s"
    except Exception as e:  # noqa: BLE001
        ok, msg = False, f"EXC {type(e).__name__}: {e}"
    if not ok:
        bad += 1
        print("FAIL " + desc + " :: " + msg, flush=True)
    elif os.environ.get("FUZZ_VERBOSE"):
        print("ok   " + msg, flush=True)
print(f"{N - bad}/{N} cases agree, worst max|dbeta| {worst:.2e}, {time.time() - t00:.0f} s")
