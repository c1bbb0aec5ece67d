"""Randomised checks of the derived design constructors (sparse / concatenate / subset / standardize compositions) against
numpy, and of the KKT diagnostics on random fits.  usage: python scripts/fuzz_designs.py [n_cases] [seed]"""
import os, sys, time, warnings
import numpy as np
import scipy.sparse as sp
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import adelie_amd as ad
from adelie_amd import diagnostic as dg
from matrix_checks import run_naive

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ONLY = [int(c) for c in os.environ["FUZZ_ONLY"].split(",")] if os.environ.get("FUZZ_ONLY") else None
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
warnings.simplefilter("ignore")
bad = 0
t0 = time.time()
for case in range(N):
    if ONLY is not None and case not in ONLY:
        continue
    rng = np.random.RandomState(seed0 * 1000 + case)
    dtype = np.float64 if rng.uniform() < 0.7 else np.float32
    n = int(rng.choice([5, 33, 64, 257, 1000]))
    pieces, dense = [], []
    for _ in range(int(rng.randint(1, 4))):
        p = int(rng.choice([1, 3, 17, 40]))
        kind = rng.choice(["dense", "snp", "sparse", "ndarray"])
        if kind == "snp":
            calls = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.55, 0.3, 0.1, 0.05]).astype(np.int8)
            imp = ad.matrix.compute_impute(calls)
            D = np.where(calls < 0, imp[None], calls).astype(dtype)
            M = ad.matrix.snp_calldata(calls, dtype=dtype)
        elif kind == "sparse":
            D = (rng.normal(size=(n, p)) * (rng.uniform(size=(n, p)) < 0.3)).astype(dtype)
            M = ad.matrix.sparse(sp.csc_matrix(D) if rng.uniform() < 0.5 else sp.csr_matrix(D),
                                 resident=str(rng.choice(["csc", "dense", "auto"])))  # kept sparse / expanded
        else:
            D = np.asfortranarray(rng.normal(size=(n, p)), dtype=dtype)
            M = ad.matrix.dense(D) if kind == "dense" else D
        pieces.append(M); dense.append(D)
    try:
        X = ad.matrix.concatenate(pieces, axis=1) if len(pieces) > 1 or isinstance(pieces[0], np.ndarray) else pieces[0]
        D = np.concatenate(dense, axis=1)
        if rng.uniform() < 0.5 and n > 8:
            rows = np.sort(rng.choice(n, size=int(rng.randint(2, n)), replace=False))
            X = X[rows]; D = D[rows]
        if rng.uniform() < 0.5 and D.shape[1] > 2:
            cols = np.sort(rng.choice(D.shape[1], size=int(rng.randint(1, D.shape[1])), replace=False))
            X = X[:, cols]; D = D[:, cols]
        if rng.uniform() < 0.4:
            c = rng.normal(size=D.shape[1]); s_ = rng.uniform(0.5, 2, D.shape[1])
            X = ad.matrix.standardize(X, centers=c, scales=s_, lazy=[True, False, "auto"][int(rng.randint(3))])  # view / copy
            D = ((D.astype(np.float64) - c[None]) / s_[None]).astype(dtype)
        if ONLY is not None:
            w = np.random.RandomState(42).uniform(0, 1, D.shape[0]).astype(dtype)
            m = np.empty(D.shape[1], dtype=dtype); X.mean(w, m)
            sq = np.empty(D.shape[1], dtype=dtype); X.sq_mul(w, sq)
            var = np.empty(D.shape[1], dtype=dtype); X.var(m.copy(), w, var)
            D64 = D.astype(np.float64)
            print("max|X|", np.abs(D).max(), "mean err", np.abs(m - w @ D64).max(), "sq err", np.abs(sq - w @ D64 ** 2).max(),
                  "var err", np.abs(var - w @ (D64 - m[None]) ** 2).max(), "var max", np.abs(var).max(), "sq max", np.abs(sq).max())
        run_naive(X, np.asfortranarray(D), dtype)
        if dtype == np.float64 and D.shape[0] >= 33 and D.shape[1] >= 3:
            y = D[:, 0] * 1.5 + rng.normal(size=D.shape[0])
            st = ad.grpnet(X, ad.glm.gaussian(y=y), tol=1e-13, progress_bar=False)
            d = dg.diagnostic(st)
            lm = np.asarray(st.lmdas)
            ratio = d.gradient_scores / lm[:, None]
            worst = float(np.max(ratio)) if len(lm) else 0.0
            # the stopping rule bounds the squared coefficient change by tol: gradients are exact to ~ sqrt(tol)
            if float(np.max(d.gradient_scores - lm[:, None])) > 2e-6:
                l, g = np.unravel_index(np.argmax(ratio), ratio.shape)
                raise AssertionError(f"score/lmda {worst} at lmda index {l}/{len(lm)} group {g} beta {st.betas[l, g]} "
                                     f"dev {st.devs[l]} lmda {lm[l]}")
            assert np.allclose(d.gradients, d.residuals @ D.astype(np.float64), atol=1e-10)
    except Exception as e:  # noqa: BLE001
        bad += 1
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print(f"case {case}: FAILED {type(e).__name__}: {str(e)[:300]} at {tb.filename.split('/')[-1]}:{tb.lineno} `{tb.line}` "
              f"n={D.shape[0]} p={D.shape[1]} dtype={np.dtype(dtype).name}", flush=True)
print(f"{N} cases, {bad} failures, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
