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
                pz = 0.0 if os.environ.get("FUZZ_NO_ZERO_BOUNDS") else 0.25   # (a bound of exactly 0 pins a coefficient: free multiplier)
                bnd = lambda: r2.choice([0.0, 0.02, 0.3, 1e100], size=q, p=[0.0, 0.4, 0.4, 0.2] if pz == 0 else [0.25, 0.3, 0.3, 0.15]).astype(dtype)  # noqa: E731
                cfgs = dict(tol=1e-15, pinball_tol=1e-14) if os.environ.get("FUZZ_CONS_TIGHT") else None
                if r2.uniform() < 0.5:
                    out.append(constraint.box(-bnd(), bnd(), configs=cfgs))
                else:
                    out.append(constraint.one_sided(r2.choice([-1.0, 1.0], size=q).astype(dtype), bnd(), configs=cfgs))
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
        dd = np.abs((st.duals - ref.duals)).max() if st.duals.shape == ref.duals.shape and st.duals.shape[1] and st.duals.shape[0] else 0.0
        ok = st.error == ref.error and db < lim and dd < 10 * lim
        if os.environ.get("FUZZ_GROUPS") and not os.environ.get("ADELIE_HIP_CONS_HOST"):
            ok = ok and st.counters["n_host_cons_visits"] == 0
        if st.error and ref.error and dtype == np.float32:
            ok = True  # (single precision at the default newton_tol: both sides stop with one of the reference's errors)
        print(f"{'ok  ' if ok else 'FAIL'} case {case}: n={n} p={p} {fam} {np.dtype(dtype).name} grouped={grouped} G={len(groups)} "
              f"ncons={sum(c is not None for c in make())} dev_visits={st.counters['n_dev_cons_visits']} | max|dbeta| {db:.2e} max|ddual| {dd:.2e} "
              f"err '{st.error[:40]}' / '{ref.error[:40]}'")
        bad += not ok
        if os.environ.get("FUZZ_DUMP") and not ok:
            D, R = st.duals.toarray(), ref.duals.toarray()
            dg = np.asarray(st.dual_groups)
            cons_list = make()
            l = int(np.argmax(np.abs(D - R).max(axis=1)))
            print(f"  worst lambda index {l} of {len(st.lmdas)}; lmda {st.lmdas[l]:.6g}; screen sets equal: "
                  f"{np.array_equal(np.sort(st.screen_set), np.sort(ref.screen_set))}; |abs_grad diff| {np.abs(st.abs_grad - ref.abs_grad).max():.3e}")
            for li in range(len(st.lmdas)):
                print(f"   lambda {li} {st.lmdas[li]:.5g}: max|dbeta| {np.abs(st.betas[li].toarray() - ref.betas[li].toarray()).max():.2e} "
                      f"max|ddual| {np.abs(D[li] - R[li]).max():.2e} nnz beta {st.betas[li].nnz}/{ref.betas[li].nnz} nnz dual {np.count_nonzero(D[li])}/{np.count_nonzero(R[li])}")
            print("   hip counters", {k: st.counters[k] for k in ("n_basil_iters", "n_cd_passes_screen", "n_cd_passes_active")}, "oracle",
                  {k: ref.counters.get(k) for k in ("n_basil_iters", "n_cd_passes_screen", "n_cd_passes_active")})
            print("   screen hip", np.sort(st.screen_set)[:40], "\n   screen ora", np.sort(ref.screen_set)[:40])
            l = next((li for li in range(len(st.lmdas)) if np.abs(D[li] - R[li]).max() > 1e-9), l)  # first divergent lambda
            print(f"   first divergent lambda index {l}")
            cols = np.argsort(-np.abs(D[l] - R[l]))[:5]
            B, Br = st.betas.toarray(), ref.betas.toarray()
            for c in cols:
                gi = int(np.searchsorted(dg, c, side="right") - 1)
                q = int(sizes[gi]); k0 = int(groups[gi])
                co = cons_list[gi]
                kind = type(co).__name__
                lo = getattr(co, "_lower", getattr(co, "_D", None)); up = getattr(co, "_upper", getattr(co, "_b", None))
                print(f"   dual col {c} (group {gi}, size {q}, {kind}, comp {c - dg[gi]}): hip {D[l, c]:.6g} oracle {R[l, c]:.6g} | "
                      f"in screen hip/oracle {gi in set(st.screen_set)}/{gi in set(ref.screen_set)}\n"
                      f"      a={np.asarray(lo)} b={np.asarray(up)}\n      beta hip {B[l, k0:k0 + q]}\n      beta ora {Br[l, k0:k0 + q]}")
    except Exception as e:  # noqa: BLE001
        print(f"FAIL case {case}: exception {type(e).__name__}: {e}")
        bad += 1
print(f"{(len(ONLY) if ONLY else N) - bad}/{len(ONLY) if ONLY else N} ok")
sys.exit(1 if bad else 0)
