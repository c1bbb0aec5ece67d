"""HIP design operations through the C ABI against numpy: the reference's run_naive check list
(tests/test_matrix.py:251-409,529-557,711-752) on dense f32/f64, C/F order, ragged sizes, adopted device tensors and
the 2-bit SNP design."""
import warnings

import numpy as np
import pytest

import adelie_amd as ad
from matrix_checks import run_naive

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,order", [(np.float64, "F"), (np.float64, "C"), (np.float32, "F"), (np.float32, "C")])
@pytest.mark.parametrize("n,p", [(2, 2), (100, 20), (20, 100), (257, 37), (1025, 3), (5000, 70)])
def test_dense(hip, n, p, dtype, order):
    rng = np.random.RandomState(0)
    X = np.asarray(rng.normal(size=(n, p)), dtype=dtype, order=order)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run_naive(ad.matrix.dense(X), X, dtype)


@pytest.mark.parametrize("n", [64, 257, 4099])  # 257 / 4099: columns not 16-byte aligned -> scalar-load variants
def test_adopted_device_tensor(hip, n):
    import torch

    p = 33
    t = torch.randn((p, n), dtype=torch.float64, device="cuda").t()  # (n, p) column-major view
    X = t.cpu().numpy()
    run_naive(ad.matrix.dense(t), np.asfortranarray(X), np.float64)
    tc = torch.randn((n, p), dtype=torch.float32, device="cuda")  # row-major: transposed copy on device
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run_naive(ad.matrix.dense(tc), tc.cpu().numpy(), np.float32)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,p", [(300, 17), (1023, 40), (8, 3)])
def test_snp_calldata(hip, n, p, dtype):
    rng = np.random.RandomState(1)
    calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8)
    imp = ad.matrix.compute_impute(calldata)
    Xd = np.asfortranarray(np.where(calldata < 0, imp[None], calldata).astype(dtype))
    run_naive(ad.matrix.snp_calldata(calldata, dtype=dtype), Xd, dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,p", [(4099, 2050), (9000, 2600), (4096, 2048)])
def test_snp_wide_sweep_through_nibble_tables(hip, n, p, dtype):
    """Sizes that take the nibble-table form of the 2-bit sweep (sweep_snp_lut_kernel: >= 512 columns, >= 4096 rows): rows
    that do not fill the last 256-row tile, columns that do not fill the last 256-column workgroup, several row splits; an
    all-missing and an all-zero column; full products and a column range."""
    rng = np.random.RandomState(n + p)
    calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.55, 0.3, 0.1, 0.05]).astype(np.int8)
    calldata[:, 5] = -9
    calldata[:, 6] = 0
    imp = ad.matrix.compute_impute(calldata)
    Xd = np.where(calldata < 0, imp[None], calldata).astype(np.float64)
    X = ad.matrix.snp_calldata(calldata, dtype=dtype)
    v = rng.normal(size=n).astype(dtype)
    w = rng.uniform(0.5, 1.5, n).astype(dtype)
    tol = 1e-11 if dtype == np.float64 else 2e-4
    out = np.empty(p, dtype=dtype)
    X.mul(v, w, out)
    ref = Xd.T @ (v.astype(np.float64) * w)
    assert np.abs(out - ref).max() <= tol * np.abs(ref).max()
    j, q = 37, p - 137
    outb = np.empty(q, dtype=dtype)
    X.bmul(j, q, v, w, outb)
    assert np.abs(outb - ref[j:j + q]).max() <= tol * np.abs(ref).max()
    sq = np.empty(p, dtype=dtype)                                     # (the squared design: weighted column variances)
    X.sq_mul(w, sq)
    ref2 = (Xd ** 2).T @ w.astype(np.float64)
    assert np.abs(sq - ref2).max() <= tol * np.abs(ref2).max()
    short = np.empty(100, dtype=dtype)                                # (the decode form, for comparison in one design)
    X.bmul(j, 100, v, w, short)
    assert np.abs(short - outb[:100]).max() <= tol * np.abs(ref).max()


def test_snp_from_snpdat_file(hip, tmp_path):
    rng = np.random.RandomState(2)
    n, p = 700, 21
    calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8)
    calldata[:, 3] = 0  # an all-zero column
    calldata[:, 4] = -9  # an all-missing column
    h = ad.io.snp_unphased(str(tmp_path / "x.snpdat"))
    h.write(calldata)
    h.read()
    X = ad.matrix.snp_unphased(h)
    imp = h.impute
    Xd = np.asfortranarray(np.where(calldata < 0, imp[None], calldata).astype(np.float64))
    run_naive(X, Xd, np.float64)


def test_inconsistent_inputs_raise(hip):
    X = ad.matrix.dense(np.asfortranarray(np.zeros((10, 4))))
    with pytest.raises(RuntimeError, match="cmul"):
        X.cmul(4, np.zeros(10), np.zeros(10))
    with pytest.raises(RuntimeError, match="bmul"):
        X.bmul(2, 3, np.zeros(10), np.zeros(10), np.zeros(3))
    with pytest.raises(RuntimeError, match="mul"):
        X.mul(np.zeros(9), np.zeros(9), np.zeros(4))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,p", [(5, 2), (257, 9), (4099, 33)])
def test_snp_bed_matches_calldata(hip, tmp_path, n, p, dtype):
    """A design built from a PLINK .bed image (transcoded and mean-imputed on the device) is the same matrix as the one
    built from the decoded calldata: impute vector, X^T v, X v (bit-identical: same 2-bit codes, same kernels)."""
    rng = np.random.RandomState(n)
    cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.55, 0.25, 0.1, 0.1])
    cd[:, 0] = -9 if n > 300 else cd[:, 0]          # an all-missing SNP: impute 0 (mean over zero valid calls)
    path = str(tmp_path / "g.bed")
    ad.io.write_bed(path, cd)
    A = ad.matrix.snp_bed(path, n, dtype=dtype)
    imp = ad.matrix.compute_impute(cd)
    B = ad.matrix.snp_calldata(cd, imp, dtype=dtype)
    assert A.shape == (n, p)
    np.testing.assert_allclose(A.impute(), imp, rtol=1e-6 if dtype == np.float32 else 1e-15, atol=0)
    v = rng.normal(size=n).astype(dtype)
    u = rng.normal(size=p).astype(dtype)
    if dtype == np.float64:
        np.testing.assert_array_equal(A.T @ v, B.T @ v)
        np.testing.assert_array_equal(A @ u, B @ u)
    else:
        np.testing.assert_allclose(A.T @ v, B.T @ v, rtol=1e-4, atol=1e-3)
    with pytest.raises(RuntimeError, match="bad magic"):
        ad.matrix.snp_bed(np.zeros(3 + ((n + 3) // 4) * p, dtype=np.uint8), n)
    with pytest.raises(RuntimeError, match="truncated"):
        ad.matrix.snp_bed(np.fromfile(path, dtype=np.uint8)[:-1], n, p)


@pytest.mark.parametrize("family", ["gaussian", "binomial"])
@pytest.mark.parametrize("kind", ["dense", "snp"])
def test_glm_path_losses_match_host(hip, family, kind):
    """cv_grpnet's per-lambda losses computed on the device (adelie_hip_design_glm_path_losses) equal predict() +
    glm.loss() on the host (reference cv.py:296-312): two weight vectors, intercepts, offsets, empty rows."""
    import scipy.sparse as sp

    rng = np.random.RandomState(4)
    n, p, L = 1237, 61, 9
    if kind == "dense":
        Xh = np.asfortranarray(rng.normal(size=(n, p)))
        X = ad.matrix.dense(Xh)
    else:
        cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.6, 0.2, 0.1, 0.1])
        imp = ad.matrix.compute_impute(cd)
        Xh = np.where(cd < 0, imp[None, :], cd).astype(np.float64)
        X = ad.matrix.snp_calldata(cd, imp)
    B = sp.random(L, p, density=0.2, random_state=rng, format="csr", dtype=np.float64)
    B = sp.vstack([B[:3], sp.csr_matrix((1, p)), B[3:]]).tocsr()   # an all-zero row
    L = B.shape[0]
    b0 = rng.normal(size=L)
    off = 0.1 * rng.normal(size=n)
    y = rng.normal(size=n) if family == "gaussian" else (rng.uniform(size=n) < 0.4).astype(np.float64)
    wa = rng.uniform(0.5, 1.5, size=n); wa /= wa.sum()
    wb = wa.copy(); wb[rng.choice(n, n // 5, replace=False)] = 0; wb /= wb.sum()
    ga = ad.glm.gaussian(y, weights=wa) if family == "gaussian" else ad.glm.binomial(y, weights=wa)
    gb = ga.reweight(wb)
    la, lb = X.glm_path_losses(ga.core_kind, B, b0, off, y, ga.weights, gb.weights)
    etas = ad.diagnostic.predict(X=X, betas=B, intercepts=b0, offsets=off)
    np.testing.assert_allclose(etas, (B @ Xh.T) + b0[:, None] + off, rtol=0, atol=1e-11)
    np.testing.assert_allclose(la, [ga.loss(e) for e in etas], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(lb, [gb.loss(e) for e in etas], rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
def test_snp_plink_fileset(hip, tmp_path):
    """matrix.snp_plink: dimensions from .fam / .bim, records transcoded on the device; same design as the calldata route."""
    rng = np.random.RandomState(5)
    n, p = 203, 17
    cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.55, 0.25, 0.1, 0.1])
    prefix = str(tmp_path / "toy")
    ad.io.write_plink(prefix, cd)
    A = ad.matrix.snp_plink(prefix)
    B = ad.matrix.snp_calldata(cd)
    assert A.shape == (n, p)
    v = rng.normal(size=n)
    w = rng.uniform(size=n)
    oa, ob = np.empty(p), np.empty(p)
    A.mul(v, w, oa)
    B.mul(v, w, ob)
    np.testing.assert_allclose(oa, ob, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(A.impute(), B.impute(), rtol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "snp"])
def test_standardize_and_subset_derived_designs(hip, oracle, kind):
    """matrix.standardize / matrix.subset (reference matrix_naive_standardize.ipp, matrix_naive_subset.ipp) as designs
    materialised on the device: all twelve MatrixNaiveBase operations against numpy, and a grpnet path on the standardized
    design against the oracle on the numpy-standardized matrix."""
    rng = np.random.RandomState(17)
    n, p = 211, 23
    if kind == "dense":
        Z = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.5, 3, p) + rng.normal(size=p))
        M = ad.matrix.dense(Z)
    else:
        cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.5, 0.3, 0.1, 0.1])
        imp = ad.matrix.compute_impute(cd)
        Z = np.asfortranarray(np.where(cd < 0, imp[None], cd).astype(float))
        M = ad.matrix.snp_calldata(cd, imp)
    S = ad.matrix.standardize(M, ddof=1)
    c = Z.mean(0)
    s = np.sqrt(((Z - c) ** 2).sum(0) / (n - 1))
    np.testing.assert_allclose(S._centers, c, atol=1e-12)
    np.testing.assert_allclose(S._scales, s, rtol=1e-12)
    Xs = np.asfortranarray((Z - c) / s)
    run_naive(S, Xs, np.float64)
    np.testing.assert_allclose(ad.matrix.standardize(Z, ddof=1), Xs, atol=1e-12)
    rows = rng.choice(n, 57, replace=False)
    cols = np.array([5, 0, 22, 7, 7])
    run_naive(ad.matrix.subset(M, rows, axis=0), np.asfortranarray(Z[rows]), np.float64)
    run_naive(ad.matrix.subset(M, cols, axis=1), np.asfortranarray(Z[:, cols]), np.float64)
    run_naive(M[rows, cols], np.asfortranarray(Z[rows][:, cols]), np.float64)
    if kind == "snp":  # rows / columns of a 2-bit design are re-packed as a 2-bit design (2 bits per call, not a dense copy)
        sub = M[rows, cols]
        assert sub._kind == "snp" and ad.matrix.subset(M, cols, axis=1)._kind == "snp"
        np.testing.assert_allclose(sub.impute(), imp[cols], rtol=1e-14)
        big = ad.matrix.subset(M, np.arange(n)[::-1], axis=0)     # every row, reversed: ragged last byte (n = 211)
        run_naive(big, np.asfortranarray(Z[::-1]), np.float64)
    assert M[:, 3:9].shape == (n, 6) and M[::2].shape == ((n + 1) // 2, p)
    with pytest.raises(RuntimeError):
        ad.matrix.subset(M, [p], axis=1)
    y = Xs[:, :3] @ np.array([1.0, -2.0, 0.5]) + 0.3 * rng.normal(size=n)
    a = ad.grpnet(S, ad.glm.gaussian(y), tol=1e-12, early_exit=False, lmda_path_size=15)
    b = ad.grpnet(oracle.dense(Xs), ad.glm.gaussian(y), tol=1e-12, early_exit=False, lmda_path_size=15)
    assert a.error == "" and np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-9


@pytest.mark.parametrize("kind,dtype", [("dense", np.float64), ("snp", np.float64), ("dense", np.float32), ("snp", np.float32)])
def test_lazy_standardized_view(hip, oracle, kind, dtype):
    """matrix.standardize(lazy=True) (reference MatrixNaiveStandardize, matrix_naive_standardize.ipp: a wrapper, nothing
    materialised): the view shares the resident dense / 2-bit matrix; its operations are the base design's plus rank-one
    corrections, grpnet composes the base kernels with the same corrections on its full-Gram engines.  Checked against the
    numpy-standardized matrix (the reference's check list) and, for paths, against the oracle on that matrix and against the
    materialised copy (lazy=False: the panel engines)."""
    rng = np.random.RandomState(23)
    n, p = 420, 260
    if kind == "dense":
        Z = np.asfortranarray((rng.normal(size=(n, p)) * rng.uniform(0.5, 3, p) + rng.normal(size=p)).astype(dtype))
        M = ad.matrix.dense(Z)
    else:
        cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.5, 0.3, 0.1, 0.1])
        imp = ad.matrix.compute_impute(cd)
        Z = np.asfortranarray(np.where(cd < 0, imp[None], cd).astype(dtype))
        M = ad.matrix.snp_calldata(cd, imp, dtype=dtype)
    S = ad.matrix.standardize(M, lazy=True)
    assert S._kind == "std" and (S.rows(), S.cols()) == (n, p)
    assert ad.matrix.standardize(M)._kind == ("std" if kind == "snp" else "dense")   # lazy="auto"
    Z64 = Z.astype(np.float64)
    c, s = Z64.mean(0), Z64.std(0)
    np.testing.assert_allclose(S._centers, c, atol=1e-12 if dtype == np.float64 else 1e-5)
    Xs = np.asfortranarray(((Z64 - c) / s).astype(dtype))
    run_naive(S, Xs, dtype)
    run_naive(S.alias(), Xs, dtype)
    c2, s2 = rng.normal(size=p), rng.uniform(0.5, 2.0, p)
    run_naive(ad.matrix.standardize(M, centers=c2, scales=s2, lazy=True), np.asfortranarray(((Z64 - c2) / s2).astype(dtype)), dtype)
    # derived designs of a view start from its materialised form
    cols = rng.choice(p, 17, replace=False)
    run_naive(ad.matrix.subset(S, cols, axis=1), np.asfortranarray(Xs[:, cols]), dtype)
    with pytest.raises(RuntimeError, match="non-zero"):
        ad.matrix.standardize(M, centers=np.zeros(p), scales=np.zeros(p), lazy=True)
    if dtype == np.float32:
        y = (Xs[:, :3] @ np.array([1.0, -2.0, 0.5]) + 0.3 * rng.normal(size=n)).astype(dtype)
        kw = dict(tol=1e-7, early_exit=False, lmda_path_size=12, min_ratio=0.05, progress_bar=False)
        a = ad.grpnet(S, ad.glm.gaussian(y, dtype=dtype), **kw)
        b = ad.grpnet(ad.matrix.standardize(M, lazy=False), ad.glm.gaussian(y, dtype=dtype), **kw)
        assert a.error == "" and np.abs(a.betas.toarray() - b.betas.toarray()).max() < 2e-3
        return
    beta = np.zeros(p)
    beta[rng.choice(p, 30, replace=False)] = rng.normal(size=30)
    y = Xs @ beta + 0.5 * rng.normal(size=n)
    w = rng.uniform(0.2, 1.0, n)
    for glm, kw in [
        (ad.glm.gaussian(y), dict(tol=1e-12, min_ratio=1e-3, lmda_path_size=30)),     # screen sets past 128 values
        (ad.glm.gaussian(y, weights=w / w.sum()), dict(tol=1e-12, groups=np.arange(0, p, 4), alpha=0.7)),
        (ad.glm.gaussian(y), dict(tol=1e-12, intercept=False)),
        (ad.glm.binomial((y > np.median(y)).astype(float)), dict(tol=1e-10, irls_tol=1e-10, lmda_path_size=10, min_ratio=0.1)),
    ]:
        kw = dict(dict(early_exit=False, lmda_path_size=15, min_ratio=2e-2, progress_bar=False), **kw)
        a = ad.grpnet(S, glm, **kw)
        b = ad.grpnet(oracle.dense(Xs), glm, **kw)
        assert a.error == "" and b.error == ""
        tol = 1e-8 if "irls_tol" not in kw else 1e-6
        assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < tol
        assert np.abs(a.intercepts - b.intercepts).max() < tol
    assert max(a.screen_sizes) >= 0
    # A lasso with an intercept on the view runs on the base design's own columns with the penalty factors times |s| (the
    # panel engines; solver._lasso_in_raw_coordinates) and comes back in standardized coordinates: every field of the state
    # against the fit on the materialised copy, and a warm start from it (which takes the view's own engines).
    Sm = ad.matrix.standardize(M, lazy=False)
    # An elastic net (0 < alpha < 1) does the same with separate factors for the quadratic part (penalty * s^2, ABI 8).
    pen = rng.uniform(0.5, 2.0, p)
    ybin = (y > np.median(y)).astype(float)
    for glm, extra in [(ad.glm.gaussian(y), {}), (ad.glm.binomial(ybin), dict(irls_tol=1e-10)),
                       (ad.glm.gaussian(y), dict(alpha=0.6)), (ad.glm.gaussian(y, weights=w / w.sum()), dict(alpha=0.3, penalty=pen)),
                       (ad.glm.binomial(ybin), dict(irls_tol=1e-10, alpha=0.5))]:
        is_glm = "irls_tol" in extra
        kw = dict(tol=1e-11, early_exit=False, lmda_path_size=14, min_ratio=5e-2 if is_glm else 2e-3, progress_bar=False, **extra)
        a = ad.grpnet(S, glm, **kw)
        b = ad.grpnet(Sm, glm, **kw)
        assert a.error == "" and b.error == "" and a._X is S
        assert is_glm or a.counters["n_panel_blocks"] > 0   # (Gaussian case: screen sets past 128 values, the panel engines ran)
        assert getattr(a, "_penalty_l2", None) is None and np.array_equal(a.penalty, b.penalty)
        if "alpha" in extra:   # ... and against the checker on the standardized matrix itself (on the part of the path where
            # the problem is well conditioned: 260 columns on 420 rows, all of them active below that)
            kwo = dict(kw, min_ratio=5e-2, tol=1e-13)
            a2 = ad.grpnet(S, glm, **kwo)
            o = ad.grpnet(oracle.dense(Xs), glm, **kwo)
            assert a2.error == "" and o.error == ""
            # (1e-6, the stated tolerance: one borderline variable enters the checker's screen set at one lambda of the alpha = 0.6
            # path and not the device's -- on the materialised copy just the same, scripts/dbg_enet_view.py -- a coefficient of 1e-7
            # at tol 1e-11, 2e-8 at tol 1e-13: the stopping rule's resolution; the other paths agree to 1e-14)
            assert np.abs(a2.betas.toarray() - o.betas.toarray()).max() < 1e-6
        assert np.allclose(a.lmdas, b.lmdas, rtol=1e-10) and abs(a.lmda_max - b.lmda_max) <= 1e-10 * b.lmda_max
        assert np.array_equal(a.screen_set, b.screen_set)
        for name, tol in [("intercepts", 1e-7), ("devs", 1e-8), ("screen_beta", 1e-7), ("grad", 1e-7), ("abs_grad", 1e-7),
                          ("screen_X_means", 1e-9), ("screen_vars", 1e-9), ("resid", 1e-7)]:
            assert np.abs(np.asarray(getattr(a, name)) - np.asarray(getattr(b, name))).max() < tol, name
        assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < (5e-7 if is_glm else 1e-7)   # (IRLS: the stopping rule's resolution)
        half = a.lmdas[:7]
        first = ad.grpnet(S, glm, **dict(kw, lmda_path=half))
        rest = ad.grpnet(S, glm, warm_start=first, **dict(kw, lmda_path=a.lmdas[7:]))
        # (a continued path differs from the uninterrupted one by the stopping rule's resolution, 1e-5 here on the
        # materialised copy as well: the warm start from the re-expressed state must do no worse)
        assert rest.error == "" and np.abs(rest.betas.toarray() - a.betas.toarray()[7:]).max() < 5e-5
        assert np.abs(rest.intercepts - a.intercepts[7:]).max() < 5e-5
    kwc = dict(n_folds=3, seed=1, lmda_path_size=8, min_ratio=0.1, progress_bar=False)
    ca = ad.cv_grpnet(S, ad.glm.gaussian(y), **kwc)
    cb = ad.cv_grpnet(ad.matrix.dense(Xs), ad.glm.gaussian(y), **kwc)
    assert np.abs(ca.losses - cb.losses).max() < 1e-7 * max(1.0, np.abs(cb.losses).max())
    cons = [None] * p
    cons[3] = ad.constraint.lower(np.array([-0.5]))
    kwk = dict(tol=1e-12, early_exit=False, lmda_path_size=8, min_ratio=0.2, progress_bar=False)
    import warnings
    with warnings.catch_warnings():   # constrained fits run ON THE VIEW (its panel form; round 6): no materialised copy, no warning
        warnings.simplefilter("error", RuntimeWarning)
        ka = ad.grpnet(S, ad.glm.gaussian(y), constraints=cons, **kwk)
    cons_b = [None] * p
    cons_b[3] = ad.constraint.lower(np.array([-0.5]))
    kb = ad.grpnet(ad.matrix.dense(Xs), ad.glm.gaussian(y), constraints=cons_b, **kwk)
    assert ka.error == "" and np.abs(ka.betas.toarray() - kb.betas.toarray()).max() < 1e-9
    Y = np.stack([y, -y], axis=1)   # multi-response fits: the view is materialised first
    kwm = dict(tol=1e-12, early_exit=False, lmda_path_size=8, min_ratio=0.2, progress_bar=False)
    ma = ad.grpnet(S, ad.glm.multigaussian(Y), **kwm)
    mb = ad.grpnet(ad.matrix.dense(Xs), ad.glm.multigaussian(Y), **kwm)
    assert ma.error == "" and np.abs(ma.betas.toarray() - mb.betas.toarray()).max() < 1e-9


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("fmt", ["csc", "csr"])
def test_sparse_design_runs_the_reference_check_list(hip, dtype, fmt):
    """matrix.sparse (reference matrix.py:1301-1385): the CSC entries are expanded on the device; every MatrixNaiveBase
    operation then has to agree with the dense matrix, as the reference's tests/test_matrix.py::test_naive_sparse asks."""
    import scipy.sparse as sp
    import warnings

    rng = np.random.RandomState(4)
    n, p = 211, 37
    D = (rng.normal(size=(n, p)) * (rng.uniform(size=(n, p)) < 0.15)).astype(dtype)
    D[:, 5] = 0  # an empty column
    M = sp.csc_matrix(D) if fmt == "csc" else sp.csr_matrix(D)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = ad.matrix.sparse(M)
    assert (X.rows(), X.cols()) == (n, p)
    run_naive(X, np.asfortranarray(D), dtype)
    with pytest.raises(TypeError, match="scipy.sparse"):
        ad.matrix.sparse(D)


def test_sparse_design_adds_duplicate_entries_and_solves_like_dense(hip):
    import scipy.sparse as sp

    rng = np.random.RandomState(5)
    n, p = 300, 60
    D = rng.normal(size=(n, p)) * (rng.uniform(size=(n, p)) < 0.2)
    y = D[:, :3] @ np.array([1.0, -2.0, 0.5]) + 0.1 * rng.normal(size=n)
    s1 = ad.grpnet(ad.matrix.sparse(sp.csc_matrix(D)), ad.glm.gaussian(y=y), progress_bar=False)
    s2 = ad.grpnet(ad.matrix.dense(np.asfortranarray(D)), ad.glm.gaussian(y=y), progress_bar=False)
    assert np.array_equal(s1.lmdas, s2.lmdas)
    assert np.array_equal(s1.betas.toarray(), s2.betas.toarray())   # same resident values, same kernels: identical
    # a non-canonical CSC (the same cell stored twice): the entries add up, as in the reference's sparse dot products
    M = sp.csc_matrix((np.array([1.0, 2.0, 5.0]), np.array([0, 0, 2]), np.array([0, 2, 3])), shape=(3, 2))
    out = np.empty(2)
    ad.matrix.sparse(M).mul(np.ones(3), np.array([1.0, 10.0, 100.0]), out)
    assert np.array_equal(out, [3.0, 500.0])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_concatenate_resident_designs(hip, dtype):
    """matrix.concatenate (reference matrix.py:214-310; tests/test_matrix.py::test_naive_cconcatenate / _rconcatenate):
    the pieces are copied into one dense design on the device and must then pass the reference's check list."""
    import scipy.sparse as sp

    rng = np.random.RandomState(6)
    n = 203
    A = np.asfortranarray(rng.normal(size=(n, 9)), dtype=dtype)
    calls = rng.choice([0, 1, 2, -9], size=(n, 7), p=[0.6, 0.25, 0.1, 0.05]).astype(np.int8)
    imp = ad.matrix.compute_impute(calls)
    Bd = np.where(calls < 0, imp[None], calls).astype(dtype)
    C = (rng.normal(size=(n, 5)) * (rng.uniform(size=(n, 5)) < 0.3)).astype(dtype)
    X = ad.matrix.concatenate(
        [ad.matrix.dense(A), ad.matrix.snp_calldata(calls, dtype=dtype), ad.matrix.sparse(sp.csc_matrix(C))], axis=1)
    assert (X.rows(), X.cols()) == (n, 21)
    run_naive(X, np.asfortranarray(np.concatenate([A, Bd, C], axis=1)), dtype)
    A2 = np.asfortranarray(rng.normal(size=(58, 9)), dtype=dtype)
    Xr = ad.matrix.concatenate([A, A2], axis=0)          # ndarrays are uploaded first
    assert (Xr.rows(), Xr.cols()) == (n + 58, 9)
    run_naive(Xr, np.asfortranarray(np.concatenate([A, A2], axis=0)), dtype)
    calls2 = rng.choice([0, 1, 2, -9], size=(n, 4), p=[0.5, 0.3, 0.1, 0.1]).astype(np.int8)
    imp2 = ad.matrix.compute_impute(calls2)
    S2 = ad.matrix.concatenate([ad.matrix.snp_calldata(calls, dtype=dtype), ad.matrix.snp_calldata(calls2, dtype=dtype)], axis=1)
    assert S2._kind == "snp"   # 2-bit designs side by side stay 2 bits per call
    run_naive(S2, np.asfortranarray(np.concatenate([Bd, np.where(calls2 < 0, imp2[None], calls2).astype(dtype)], axis=1)), dtype)
    with pytest.raises(RuntimeError, match="same number of rows"):
        ad.matrix.concatenate([A, A2], axis=1)
    with pytest.raises(RuntimeError, match="same number of columns"):
        ad.matrix.concatenate([A, C], axis=0)


def test_concatenated_design_solves_like_the_dense_matrix(hip):
    """reference tests/test_solver.py:659-754 (test_solve_gaussian_concatenate): a path on the concatenation of designs
    equals the path on the concatenated array."""
    rng = np.random.RandomState(7)
    n, ps = 150, [20, 35, 10]
    Xs = [np.asfortranarray(rng.normal(size=(n, p))) for p in ps]
    X = np.asfortranarray(np.concatenate(Xs, axis=1))
    y = X[:, [1, 25, 60]] @ np.array([1.0, -1.0, 2.0]) + 0.2 * rng.normal(size=n)
    groups = np.arange(0, X.shape[1], 5)
    s1 = ad.grpnet(ad.matrix.concatenate([ad.matrix.dense(x) for x in Xs], axis=1), ad.glm.gaussian(y=y), groups=groups,
                   progress_bar=False)
    s2 = ad.grpnet(ad.matrix.dense(X), ad.glm.gaussian(y=y), groups=groups, progress_bar=False)
    assert np.array_equal(s1.lmdas, s2.lmdas) and np.array_equal(s1.betas.toarray(), s2.betas.toarray())


def test_new_constructors_reject_bad_inputs(hip):
    import scipy.sparse as sp

    rng = np.random.RandomState(8)
    X = ad.matrix.dense(np.asfortranarray(rng.normal(size=(20, 4))))
    with pytest.raises(RuntimeError, match="mul_batch\\(\\) is given inconsistent inputs"):
        X.mul_batch(np.zeros((3, 19)))
    assert X.mul_batch(np.zeros((0, 20))).shape == (0, 4)
    view = ad.matrix.kronecker_eye(X, 2)
    with pytest.raises(NotImplementedError, match="resident designs"):
        ad.matrix.concatenate([X, view], axis=1)
    with pytest.raises(ValueError, match="axis must be 0 or 1"):
        ad.matrix.concatenate([X, X], axis=2)
    with pytest.raises(RuntimeError, match="non-empty"):
        ad.matrix.concatenate([], axis=1)
    with pytest.raises(RuntimeError, match="same dtype"):
        ad.matrix.concatenate([X, ad.matrix.dense(np.asfortranarray(rng.normal(size=(20, 4)), dtype=np.float32))], axis=1)
    with pytest.raises(RuntimeError, match="float32 or float64"):
        ad.matrix.sparse(sp.csc_matrix(np.eye(3, dtype=np.int64)))
    with pytest.raises(ValueError, match="'naive' or 'cov'"):
        ad.matrix.sparse(sp.csc_matrix(np.eye(3)), method="lazy")
    with pytest.raises(RuntimeError, match=r"\(p, p\)"):
        ad.matrix.sparse(sp.csc_matrix(np.ones((3, 4))), method="cov")
    with pytest.raises(RuntimeError, match="multi_path_losses\\(\\) is given inconsistent inputs"):
        X.multi_path_losses(0, 2, sp.csr_matrix(np.zeros((1, 7))), np.zeros((1, 2)), np.zeros((20, 2)), np.zeros((20, 2)),
                            np.ones(20) / 20, np.ones(20) / 20)


def test_snpdat_decoder_rejects_corrupt_images(hip, tmp_path):
    """A truncated or corrupted ``.snpdat`` image must raise, never read outside the buffer: every byte of the header and of the
    first column's offset table / chunk headers is damaged in turn (the offsets, counts and chunk sizes the decoder follows)."""
    rng = np.random.RandomState(0)
    n, p = 700, 5
    cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.5, 0.3, 0.1, 0.1])
    io = ad.io.snp_unphased(str(tmp_path / "a.snpdat"))
    io.write(np.asfortranarray(cd))
    io.read()
    good = np.array(io._buffer, copy=True)
    b = hip
    h = _abi_handle = __import__("ctypes").c_void_p()

    def attempt(buf):
        rc = b.fn("design_create_snp_unphased")(buf.ctypes.data, buf.size, 1, 0, h)
        if rc == 0:
            b.fn("design_destroy")(h)
        return rc

    assert attempt(good) == 0
    hdr = 17 + 24 * p + 8 * (p + 1)
    n_rejected = 0
    for cut in (0, 5, 16, hdr - 1, hdr + 10, good.size - 1):
        assert attempt(good[:cut].copy()) != 0
    # byte 4 is left alone: it turns n = 700 into ~4e9 rows, which is still below the decoder's plausibility cap (2^32) and
    # would only make the test allocate gigabytes for a "valid" all-zero tail
    for pos in [q for q in range(1, 17) if q != 4] + list(range(17 + 24 * p, hdr + 40)):
        for val in (0xFF, 0x00, 0x7F):
            bad = good.copy()
            if bad[pos] == val:
                continue
            bad[pos] = val
            n_rejected += attempt(bad) != 0
    assert n_rejected > 50
    # a row count beyond the cap is refused before anything is sized from it
    bad = good.copy()
    bad[1:9] = np.frombuffer(np.uint64(1 << 45).tobytes(), dtype=np.uint8)
    assert attempt(bad) != 0 and b"row / column counts" in b.fn("last_error")()


@pytest.mark.parametrize("kind", ["dense", "snp"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_contiguous_subsets_share_the_resident_matrix(hip, oracle, kind, dtype):
    """matrix.subset over a contiguous index range (reference matrix_naive_subset.ipp: a lazy wrapper): a slice of the resident
    matrix, nothing copied (adelie_hip_design_create_slice) -- every matrix operation against numpy on the sliced array, a path
    on the slice against the oracle; other index sets (and row ranges off a 16-byte boundary, rows of a 2-bit design) copy."""
    rng = np.random.RandomState(3)
    n, p = 333, 57
    if kind == "dense":
        Z = np.asfortranarray(rng.normal(size=(n, p)).astype(dtype))
        M = ad.matrix.dense(Z)
    else:
        cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.5, 0.3, 0.1, 0.1])
        imp = ad.matrix.compute_impute(cd)
        Z = np.asfortranarray(np.where(cd < 0, imp[None], cd).astype(dtype))
        M = ad.matrix.snp_calldata(cd, imp, dtype=dtype)
    c = ad.matrix.subset(M, np.arange(11, 40), axis=1)
    assert getattr(c, "_slice_of", None) is M and c._kind == kind and c.shape == (n, 29)
    run_naive(c, np.asfortranarray(Z[:, 11:40]), dtype)
    run_naive(ad.matrix.subset(c, np.arange(5, 9), axis=1), np.asfortranarray(Z[:, 16:20]), dtype)   # a slice of a slice
    one = ad.matrix.subset(M, [p - 1], axis=1)
    assert getattr(one, "_slice_of", None) is M
    run_naive(one, np.asfortranarray(Z[:, p - 1:]), dtype)
    r = ad.matrix.subset(M, np.arange(64, 300), axis=0)
    if kind == "dense":
        assert getattr(r, "_slice_of", None) is M
        assert getattr(ad.matrix.subset(M, np.arange(1, 300), axis=0), "_slice_of", None) is None   # off the boundary: a copy
    else:
        assert getattr(r, "_slice_of", None) is None                                                # 2-bit rows: re-packed copy
    run_naive(r, np.asfortranarray(Z[64:300]), dtype)
    run_naive(ad.matrix.subset(M, np.arange(1, 300), axis=0), np.asfortranarray(Z[1:300]), dtype)
    assert getattr(ad.matrix.subset(M, [3, 5, 6], axis=1), "_slice_of", None) is None
    if dtype == np.float64:
        Zs = Z[64:300, 11:40] if kind == "dense" else Z[:, 11:40]
        S = ad.matrix.subset(r, np.arange(11, 40), axis=1) if kind == "dense" else c
        y = Zs[:, :3] @ np.array([1.0, -2.0, 0.5]) + 0.3 * rng.normal(size=Zs.shape[0])
        kw = dict(tol=1e-12, early_exit=False, lmda_path_size=12, min_ratio=0.02, progress_bar=False)
        a = ad.grpnet(S, ad.glm.gaussian(y), **kw)
        b = ad.grpnet(oracle.dense(np.asfortranarray(Zs)), ad.glm.gaussian(y), **kw)
        assert a.error == "" and np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-9
        cvs = ad.cv_grpnet(S, ad.glm.gaussian(y), n_folds=3, seed=0, lmda_path_size=6, min_ratio=0.2, progress_bar=False)
        cvd = ad.cv_grpnet(ad.matrix.dense(np.asfortranarray(Zs)), ad.glm.gaussian(y), n_folds=3, seed=0, lmda_path_size=6,
                           min_ratio=0.2, progress_bar=False)
        assert np.abs(cvs.losses - cvd.losses).max() < 1e-9 * max(1.0, np.abs(cvd.losses).max())


@pytest.mark.parametrize("kind", ["dense", "snp"])
def test_standardized_view_on_the_panel_engines(hip, oracle, monkeypatch, kind):
    """Grouped and no-intercept fits on a lazily standardized view (what solver._lasso_in_raw_coordinates cannot re-express on
    the base design) run the panel engines on the BASE design's columns with the view's corrections around every step: the
    changes over the scales and kappa off every row before it, the block's gradient and the diagonal blocks corrected from the
    raw sums behind it.  Forced from the first screen value on; against the oracle on the standardized matrix and against the
    view's own full-Gram engines (ADELIE_HIP_STD_PANEL=0)."""
    rng = np.random.RandomState(29)
    n, p = 500, 240
    if kind == "dense":
        Z = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.5, 3, p) + rng.normal(size=p))
        M = ad.matrix.dense(Z)
    else:
        cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.5, 0.3, 0.1, 0.1])
        imp = ad.matrix.compute_impute(cd)
        Z = np.asfortranarray(np.where(cd < 0, imp[None], cd).astype(np.float64))
        M = ad.matrix.snp_calldata(cd, imp)
    S = ad.matrix.standardize(M, lazy=True)
    Xs = np.asfortranarray((Z - Z.mean(0)) / Z.std(0))
    beta = np.zeros(p)
    beta[rng.choice(p, 25, replace=False)] = rng.normal(size=25)
    y = Xs @ beta + 0.5 * rng.normal(size=n)
    yb = (y > np.median(y)).astype(float)
    w = rng.uniform(0.2, 1.0, n)
    w /= w.sum()
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    for glm, kw in [
        (ad.glm.gaussian(y, weights=w), dict(groups=np.arange(0, p, 4), alpha=0.7)),
        (ad.glm.gaussian(y), dict(intercept=False)),
        (ad.glm.gaussian(y), dict(intercept=False, alpha=0.4, penalty=rng.uniform(0.5, 2.0, p))),
        (ad.glm.binomial(yb), dict(intercept=False, alpha=0.5, irls_tol=1e-11, min_ratio=0.1)),
        (ad.glm.binomial(yb, weights=w), dict(groups=np.arange(0, p, 3), irls_tol=1e-11, min_ratio=0.1)),
    ]:
        kw = dict(dict(tol=1e-12, early_exit=False, lmda_path_size=12, min_ratio=3e-2, progress_bar=False), **kw)
        a = ad.grpnet(S, glm, **kw)
        monkeypatch.setenv("ADELIE_HIP_STD_PANEL", "0")
        b = ad.grpnet(S, glm, **kw)
        monkeypatch.delenv("ADELIE_HIP_STD_PANEL")
        o = ad.grpnet(oracle.dense(Xs), glm, **kw)
        assert a.error == "" and b.error == "" and o.error == ""
        assert a.counters["n_panel_blocks"] > 0 and b.counters["n_panel_blocks"] == 0
        tol = 1e-6 if "irls_tol" in kw else 1e-8
        assert np.abs(a.betas.toarray() - o.betas.toarray()).max() < tol, kw
        assert np.abs(a.intercepts - o.intercepts).max() < tol
        assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < tol
        # the invariants handed back are the view's: residual / gradient of the last solution
        for name in ("grad", "resid") if "irls_tol" not in kw else ("grad", "eta"):
            assert np.abs(np.asarray(getattr(a, name)) - np.asarray(getattr(o, name))).max() < 1e-6, name
