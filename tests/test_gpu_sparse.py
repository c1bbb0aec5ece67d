"""A sparse design KEPT SPARSE in HBM (``matrix.sparse(..., resident="csc")``, ``adelie_hip_design_create_csc``,
``kernels_sparse.hip``) — reference ``MatrixNaiveSparse`` (``matrix_naive_sparse.ipp``, ``adelie/matrix.py:1301-1385``).

The reference's own check list for the class (``tests/test_matrix.py::test_naive_sparse`` -> ``run_naive``) runs against numpy
on the densified matrix; paths solved on the sparse-resident design are compared with the CPU oracle on the dense copy of the
same data (the oracle has no sparse class: it checks the numbers, not the storage)."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import adelie_amd as ad
from matrix_checks import run_naive

pytestmark = pytest.mark.gpu


def _rand_sparse(rng, n, p, density, dtype=np.float64):
    D = (rng.normal(size=(n, p)) * (rng.uniform(size=(n, p)) < density)).astype(dtype)
    return D


def _csc(M, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ad.matrix.sparse(M, resident="csc", **kw)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("n,p,density", [(211, 37, 0.15), (1030, 90, 0.03), (70, 300, 0.4), (5, 3, 1.0)])
def test_kept_sparse_design_runs_the_reference_check_list(hip, dtype, fmt, n, p, density):
    rng = np.random.RandomState(4)
    D = _rand_sparse(rng, n, p, density, dtype)
    if p > 6:
        D[:, 5] = 0  # an empty column
    if n > 9:
        D[7] = 0     # an empty row
    M = sp.csc_matrix(D) if fmt == "csc" else sp.csr_matrix(D)
    X = _csc(M)
    assert X._kind == "sparse" and (X.rows(), X.cols()) == (n, p)
    run_naive(X, np.asfortranarray(D), dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kept_sparse_design_with_row_blocks(hip, oracle, dtype):
    """More rows than one row block of the sweeps (about 1 MB of an n-vector: 131072 f64 / 262144 f32 rows): the sweeps run
    per (column, row block) with a fixed-order sum over the blocks."""
    rng = np.random.RandomState(5)
    n, p = (140_001 if dtype == np.float64 else 300_017), 24
    D = _rand_sparse(rng, n, p, 0.002, dtype)
    D[:, 3] = 0
    D[n - 1, 4] = 2.5  # an entry in the last row of the last block
    X = _csc(sp.csc_matrix(D))
    run_naive(X, np.asfortranarray(D), dtype)
    if dtype == np.float64:
        y = D[:, :3] @ np.array([1.0, -2.0, 0.5]) + 0.05 * rng.normal(size=n)
        kw = dict(tol=1e-12, early_exit=False, lmda_path_size=12, min_ratio=0.05, progress_bar=False)
        a = ad.grpnet(X, ad.glm.gaussian(y), **kw)
        b = ad.grpnet(oracle.dense(np.asfortranarray(D)), ad.glm.gaussian(y), **kw)
        assert a.error == "" and np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-9


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kept_sparse_design_with_row_tiles(hip, oracle, dtype):
    """Enough stored entries for the tile-major copy (full sweeps from tiles of v in LDS, csc_tile_sweep_kernel): 16384 f64 /
    32768 f32 rows per tile, a ragged last tile, an empty column, an empty tile inside a column."""
    rng = np.random.RandomState(6)
    n, p = 70_001, 60
    D = _rand_sparse(rng, n, p, 0.05, dtype)
    D[:, 3] = 0
    D[16384:32768, 4] = 0      # column 4 has nothing in its second f64 tile
    D[n - 1, 5] = 1.5
    X = _csc(sp.csc_matrix(D))
    run_naive(X, np.asfortranarray(D), dtype)
    D64 = D.astype(np.float64)
    c, s = D64.mean(axis=0), D64.std(axis=0)
    s[3] = 1.0                 # (the empty column has no spread)
    Z = ad.matrix.standardize(X, centers=c, scales=s)
    run_naive(Z, np.asfortranarray(((D64 - c) / s).astype(dtype)), dtype)
    if dtype == np.float64:
        y = D[:, :3] @ np.array([1.0, -2.0, 0.5]) + 0.05 * rng.normal(size=n)
        kw = dict(tol=1e-12, early_exit=False, lmda_path_size=12, min_ratio=0.05, progress_bar=False)
        a = ad.grpnet(X, ad.glm.gaussian(y), **kw)
        b = ad.grpnet(oracle.dense(np.asfortranarray(D)), ad.glm.gaussian(y), **kw)
        assert a.error == "" and np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-9


def test_kept_sparse_design_sums_duplicates_and_validates(hip):
    # the same cell stored twice: the entries add up, as in the reference's sparse dot products
    M = sp.csc_matrix((np.array([1.0, 2.0, 5.0]), np.array([0, 0, 2]), np.array([0, 2, 3])), shape=(3, 2))
    out = np.empty(2)
    _csc(M).mul(np.ones(3), np.array([1.0, 10.0, 100.0]), out)
    assert np.array_equal(out, [3.0, 500.0])
    with pytest.raises(TypeError, match="scipy.sparse"):
        ad.matrix.sparse(np.eye(3), resident="csc")
    with pytest.raises(ValueError, match="resident"):
        ad.matrix.sparse(sp.csc_matrix(np.eye(3)), resident="hbm")
    # a C-ABI caller's row form that does not hold the column form's entries is refused
    from adelie_amd import _abi
    b = _abi.hip_backend()
    Mc = sp.csc_matrix(np.array([[1.0, 0.0], [0.0, 2.0], [3.0, 0.0]]))
    Mr = sp.csr_matrix(np.array([[1.0, 0.0], [0.0, 2.5], [3.0, 0.0]]))  # one value differs
    h = _abi.C.c_void_p()
    ip, ix, vv = Mc.indptr.astype(np.int64), Mc.indices.astype(np.int32), Mc.data.astype(np.float64)
    rp, rx, rv = Mr.indptr.astype(np.int64), Mr.indices.astype(np.int32), Mr.data.astype(np.float64)
    with pytest.raises(RuntimeError, match="same entries"):
        b.check(b.fn("design_create_csc")(ip.ctypes.data, ix.ctypes.data, vv.ctypes.data, rp.ctypes.data, rx.ctypes.data,
                                          rv.ctypes.data, 3, 2, _abi.dtype_code(np.float64), 0, h))
    # the default keeps very sparse matrices sparse and expands the others
    rng = np.random.RandomState(0)
    assert ad.matrix.sparse(sp.csc_matrix(_rand_sparse(rng, 400, 50, 0.005)))._kind == "sparse"
    assert ad.matrix.sparse(sp.csc_matrix(_rand_sparse(rng, 400, 50, 0.3)))._kind == "dense"


def _problem(rng, n, p, density, k=6):
    D = _rand_sparse(rng, n, p, density)
    beta = np.zeros(p)
    beta[rng.choice(p, k, replace=False)] = rng.normal(size=k) * 2
    y = D @ beta + 0.3 * rng.normal(size=n)
    return D, y


@pytest.mark.parametrize("case", ["lasso", "enet_groups", "weights_no_intercept", "wide", "wide_groups"])
def test_gaussian_paths_on_a_kept_sparse_design_match_the_oracle(hip, oracle, case):
    """StateGaussianNaive on MatrixNaiveSparse (reference tests/test_solver.py solves on every matrix class): screen sets
    below and above the 128 values where the multi-CU Gram engine takes over from the single-workgroup kernel."""
    rng = np.random.RandomState(11)
    kw = dict(tol=1e-12, early_exit=False, lmda_path_size=25, min_ratio=1e-2, progress_bar=False)
    if case == "lasso":
        D, y = _problem(rng, 400, 60, 0.1)
        glm = ad.glm.gaussian(y)
    elif case == "enet_groups":
        D, y = _problem(rng, 500, 90, 0.08)
        glm = ad.glm.gaussian(y)
        kw.update(groups=np.arange(0, 90, 3), alpha=0.6)
    elif case == "weights_no_intercept":
        D, y = _problem(rng, 350, 80, 0.1)
        w = rng.uniform(0.2, 1.0, 350)
        glm = ad.glm.gaussian(y, weights=w / w.sum())
        kw.update(intercept=False, penalty=rng.uniform(0.5, 2.0, 80))
    else:  # p > n, screen sets of several hundred values
        D, y = _problem(rng, 300, 900, 0.05, k=40)
        glm = ad.glm.gaussian(y)
        kw.update(lmda_path_size=40, min_ratio=2e-3)
        if case == "wide_groups":
            kw.update(groups=np.arange(0, 900, 3), alpha=0.8)
    a = ad.grpnet(_csc(sp.csc_matrix(D)), glm, **kw)
    b = ad.grpnet(oracle.dense(np.asfortranarray(D)), glm, **kw)
    assert a.error == "" and b.error == ""
    assert np.allclose(a.lmdas, b.lmdas, rtol=1e-12)
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-8
    assert np.abs(a.intercepts - b.intercepts).max() < 1e-8
    assert np.abs(np.asarray(a.devs) - np.asarray(b.devs)).max() < 1e-9
    if case.startswith("wide"):
        assert max(a.screen_sizes) * (3 if case == "wide_groups" else 1) > 128  # the multi-CU Gram block passes ran


@pytest.mark.parametrize("family", ["binomial", "poisson"])
def test_glm_paths_on_a_kept_sparse_design_match_the_oracle(hip, oracle, family):
    rng = np.random.RandomState(12)
    n, p = 600, 70
    D = _rand_sparse(rng, n, p, 0.1)
    eta = D[:, :4] @ np.array([1.0, -1.5, 0.7, 0.5])
    if family == "binomial":
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(float)
        glm = ad.glm.binomial(y)
    else:
        y = rng.poisson(np.exp(0.3 * eta)).astype(float)
        glm = ad.glm.poisson(y)
    kw = dict(tol=1e-10, irls_tol=1e-10, early_exit=False, lmda_path_size=15, min_ratio=5e-2, progress_bar=False)
    a = ad.grpnet(_csc(sp.csc_matrix(D)), glm, **kw)
    b = ad.grpnet(oracle.dense(np.asfortranarray(D)), glm, **kw)
    assert a.error == "" and b.error == ""
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-6
    assert np.abs(a.intercepts - b.intercepts).max() < 1e-6


def test_kept_sparse_equals_expanded_on_the_device_and_f32(hip):
    """The same entries kept sparse and expanded to a dense design (other engines, other summation order): the paths agree
    to the solver tolerance in both precisions."""
    rng = np.random.RandomState(13)
    D, y = _problem(rng, 800, 150, 0.05)
    for dtype, tol in [(np.float64, 1e-8), (np.float32, 2e-3)]:
        M = sp.csc_matrix(D.astype(dtype))
        kw = dict(tol=1e-12 if dtype == np.float64 else 1e-7, early_exit=False, lmda_path_size=20, min_ratio=2e-2,
                  progress_bar=False)
        a = ad.grpnet(_csc(M), ad.glm.gaussian(y.astype(dtype), dtype=dtype), **kw)
        b = ad.grpnet(ad.matrix.sparse(M, resident="dense"), ad.glm.gaussian(y.astype(dtype), dtype=dtype), **kw)
        assert a.error == "" and b.error == ""
        assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < tol


def test_derived_designs_of_a_kept_sparse_design(hip):
    """subset / concatenate of sparse pieces stay sparse (composed on the host); standardize and mixed concatenations expand
    (reference matrix.py:1414-1640, 214-310)."""
    rng = np.random.RandomState(14)
    n, p = 230, 41
    D = _rand_sparse(rng, n, p, 0.1)
    X = _csc(sp.csc_matrix(D))
    rows = rng.choice(n, 77, replace=False)
    cols = rng.choice(p, 13, replace=False)
    Sr, Sc = ad.matrix.subset(X, rows, axis=0), ad.matrix.subset(X, cols, axis=1)
    assert Sr._kind == "sparse" and Sc._kind == "sparse"
    run_naive(Sr, np.asfortranarray(D[rows]), np.float64)
    run_naive(Sc, np.asfortranarray(D[:, cols]), np.float64)
    run_naive(X[rows, cols], np.asfortranarray(D[rows][:, cols]), np.float64)
    with pytest.raises(RuntimeError):
        ad.matrix.subset(X, [p], axis=1)
    D2 = _rand_sparse(rng, n, 9, 0.2)
    H = ad.matrix.concatenate([X, _csc(sp.csc_matrix(D2))], axis=1)
    assert H._kind == "sparse"
    run_naive(H, np.asfortranarray(np.concatenate([D, D2], axis=1)), np.float64)
    A = np.asfortranarray(rng.normal(size=(n, 4)))
    Hm = ad.matrix.concatenate([X, A], axis=1)  # a dense piece: everything is expanded
    run_naive(Hm, np.asfortranarray(np.concatenate([D, A], axis=1)), np.float64)
    Hz = ad.matrix.concatenate([ad.matrix.standardize(X), _csc(sp.csc_matrix(D2))], axis=1)  # a standardized piece: expanded
    run_naive(Hz, np.asfortranarray(np.concatenate([(D - D.mean(axis=0)) / D.std(axis=0), D2], axis=1)), np.float64)
    # an alias (what cv_grpnet hands every fold) shares the resident arrays
    run_naive(X.alias(), np.asfortranarray(D), np.float64)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_standardized_view_of_a_kept_sparse_design(hip, oracle, dtype):
    """matrix.standardize over a sparse design (reference matrix.py:1414-1533, matrix_naive_standardize.ipp): the entries stay
    compressed, centring and scaling are rank-one corrections in every operation's epilogue.  The reference's check list for
    the class (tests/test_matrix.py::test_naive_standardize) against the dense standardized matrix, with default and with
    given centres / scales, over one and over several row blocks; paths against the oracle on the dense standardized copy."""
    rng = np.random.RandomState(21)
    for n, p, dens in [(333, 29, 0.12), (140_003, 11, 0.004)]:
        if dtype == np.float32 and n > 1000:
            continue
        D = _rand_sparse(rng, n, p, dens, dtype)
        X = _csc(sp.csc_matrix(D))
        Z = ad.matrix.standardize(X)
        assert Z._kind == "sparse" and Z._std is not None
        D64 = D.astype(np.float64)
        c, s = D64.mean(axis=0), D64.std(axis=0)
        assert np.abs(Z._centers - c).max() < (1e-12 if dtype == np.float64 else 1e-6)
        assert np.abs(Z._scales - s).max() < (1e-12 if dtype == np.float64 else 1e-5)
        run_naive(Z, np.asfortranarray(((D64 - c) / s).astype(dtype)), dtype)
        run_naive(Z.alias(), np.asfortranarray(((D64 - c) / s).astype(dtype)), dtype)
        c2, s2 = rng.normal(size=p), rng.uniform(0.5, 2.0, p)
        Z2 = ad.matrix.standardize(X, centers=c2, scales=s2)
        run_naive(Z2, np.asfortranarray(((D64 - c2) / s2).astype(dtype)), dtype)
    with pytest.raises(RuntimeError, match="non-zero"):
        ad.matrix.standardize(X, centers=np.zeros(p), scales=np.zeros(p))
    if dtype == np.float32:
        return
    n, p = 500, 80
    D, y = _problem(rng, n, p, 0.1)
    Zd = (D - D.mean(axis=0)) / D.std(axis=0)
    Z = ad.matrix.standardize(_csc(sp.csc_matrix(D)))
    w = rng.uniform(0.2, 1.0, n)
    for glm, kw in [
        (ad.glm.gaussian(y), dict(tol=1e-12)),
        (ad.glm.gaussian(y, weights=w / w.sum()), dict(tol=1e-12, groups=np.arange(0, p, 4), alpha=0.7)),
        (ad.glm.gaussian(y), dict(tol=1e-12, intercept=False)),
        (ad.glm.binomial((y > np.median(y)).astype(float)), dict(tol=1e-10, irls_tol=1e-10)),
        # elastic nets with an intercept: solved on the sparse base's own columns, penalty |s| and penalty s^2 (penalty_l2, ABI 8)
        (ad.glm.gaussian(y, weights=w / w.sum()), dict(tol=1e-12, alpha=0.4, penalty=rng.uniform(0.5, 2.0, p))),
        (ad.glm.binomial((y > np.median(y)).astype(float)), dict(tol=1e-10, irls_tol=1e-10, alpha=0.5)),
    ]:
        kw.update(early_exit=False, lmda_path_size=15, min_ratio=2e-2, progress_bar=False)
        a = ad.grpnet(Z, glm, **kw)
        b = ad.grpnet(oracle.dense(np.asfortranarray(Zd)), glm, **kw)
        assert a.error == "" and b.error == ""
        tol = 1e-8 if "irls_tol" not in kw else 1e-6
        assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < tol
        assert np.abs(a.intercepts - b.intercepts).max() < tol
    kwc = dict(n_folds=3, seed=1, lmda_path_size=8, min_ratio=0.1, progress_bar=False)
    ca = ad.cv_grpnet(Z, ad.glm.gaussian(y), **kwc)
    cb = ad.cv_grpnet(ad.matrix.dense(np.asfortranarray(Zd)), ad.glm.gaussian(y), **kwc)
    assert np.abs(ca.losses - cb.losses).max() < 1e-7 * max(1.0, np.abs(cb.losses).max())


def test_kept_sparse_design_serves_constrained_and_multi_response_fits(hip, oracle):
    rng = np.random.RandomState(15)
    D, y = _problem(rng, 300, 40, 0.15)
    X = _csc(sp.csc_matrix(D))
    cons = [None] * 40
    cons[3] = ad.constraint.lower(np.array([-0.5]))
    # constraints live in the panel engines: the fit runs on the expanded copy (with a warning), as the reference composes any
    # matrix with any constraint
    kwc = dict(tol=1e-12, early_exit=False, lmda_path_size=10, min_ratio=0.1, progress_bar=False)
    with pytest.warns(RuntimeWarning, match="materialised dense copy"):
        ca = ad.grpnet(X, ad.glm.gaussian(y), constraints=cons, **kwc)
    cons2 = [None] * 40
    cons2[3] = ad.constraint.lower(np.array([-0.5]))
    cb = ad.grpnet(ad.matrix.dense(np.asfortranarray(D)), ad.glm.gaussian(y), constraints=cons2, **kwc)
    assert ca.error == "" and np.abs(ca.betas.toarray() - cb.betas.toarray()).max() < 1e-9
    # ... under IRLS the design has a panel form of its own (compressed columns): constrained fits run on it, no copy, no warning,
    # one-coefficient closed forms and a box object on a group alike; against the oracle on the dense matrix
    import warnings
    yb = (y > np.median(y)).astype(float)
    grp = np.concatenate([np.arange(0, 30), [30, 35]])

    def cons_glm():
        c = [None] * len(grp)
        c[3] = ad.constraint.lower(np.array([-0.2]))
        c[7] = ad.constraint.box(np.array([-0.1]), np.array([0.15]))
        c[30] = ad.constraint.box(np.full(5, -0.05), np.full(5, 0.1))
        return c

    kwb = dict(groups=grp, tol=1e-12, irls_tol=1e-11, early_exit=False, lmda_path_size=8, min_ratio=0.1, progress_bar=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        ga = ad.grpnet(X, ad.glm.binomial(yb), constraints=cons_glm(), **kwb)
    gb = ad.grpnet(oracle.dense(np.asfortranarray(D)), ad.glm.binomial(yb), constraints=cons_glm(), **kwb)
    assert ga.error == "" and gb.error == ""
    assert np.abs(ga.betas.toarray() - gb.betas.toarray()).max() < 1e-7
    assert np.abs(ga.intercepts - gb.intercepts).max() < 1e-7
    # multi-response fits need dense column slices: the view is built over the expanded copy
    Y = np.stack([y, -y + 0.1 * rng.normal(size=300)], axis=1)
    kw = dict(tol=1e-12, early_exit=False, lmda_path_size=10, min_ratio=0.1, progress_bar=False)
    a = ad.grpnet(X, ad.glm.multigaussian(Y), **kw)
    b = ad.grpnet(ad.matrix.dense(np.asfortranarray(D)), ad.glm.multigaussian(Y), **kw)
    assert a.error == "" and np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-9


def test_cv_grpnet_on_a_kept_sparse_design(hip):
    """cv_grpnet (reference adelie/cv.py:239-314) on aliases of the sparse-resident design, per-lambda losses through the
    device's sp_tmul: the same table as on the expanded copy."""
    rng = np.random.RandomState(16)
    D, y = _problem(rng, 500, 60, 0.1)
    M = sp.csc_matrix(D)
    kw = dict(n_folds=4, seed=3, lmda_path_size=12, min_ratio=0.05, progress_bar=False)
    a = ad.cv_grpnet(_csc(M), ad.glm.gaussian(y), **kw)
    b = ad.cv_grpnet(ad.matrix.sparse(M, resident="dense"), ad.glm.gaussian(y), **kw)
    assert np.allclose(a.lmdas, b.lmdas, rtol=1e-10)
    if not np.abs(a.losses - b.losses).max() < 1e-7 * max(1.0, np.abs(b.losses).max()):
        c = ad.cv_grpnet(_csc(M), ad.glm.gaussian(y), n_concurrent=1, **kw)
        d = np.abs(a.losses - b.losses)
        raise AssertionError(f"where {np.argwhere(d > 1e-6).tolist()} csc {a.losses[d > 1e-6]} dense {b.losses[d > 1e-6]} "
                             f"csc sequential {c.losses[d > 1e-6]}")
    assert a.best_idx == b.best_idx


@pytest.mark.parametrize("engine", ["gram_sparse", "gram_view", "panel_view", "panel_sparse", "panel_sparse_groups"])
def test_irls_on_a_sparse_design_reuses_its_blocks_under_small_weight_drift(hip, oracle, monkeypatch, engine):
    """IRLS on a design kept sparse.  Groups of one on the plain matrix: the panel engine over compressed columns (step over
    the stored entries, 64-visit diagonal blocks by row-list merges; ADELIE_HIP_SPARSE_PANEL=0 keeps the full-Gram engines).
    Full-Gram engines (that hook, or a standardized view with no intercept): the screen set's Gram is kept while the weights
    drift by at most 10 % (the rule of the panel blocks) and every pass starts from the exact gradient of the current residual,
    so the fixed point is the exact one.  Against the oracle, against always-rebuild (ADELIE_HIP_IRLS_REUSE=0), and the saved
    work by counter."""
    rng = np.random.RandomState(5)
    n, p = 900, 400
    D, y = _problem(rng, n, p, 0.08, k=25)
    yb = (y > np.median(y)).astype(float)
    X = _csc(sp.csc_matrix(D))
    Xd = D
    kw = dict(tol=1e-12, irls_tol=1e-12, early_exit=False, lmda_path_size=20, min_ratio=5e-2, progress_bar=False)
    if engine in ("gram_view", "panel_view"):   # elastic net without an intercept on the standardized view of the sparse design: the
        # panel engine over compressed columns with the view's corrections around every step, or (hook) the view's full-Gram engines
        X = ad.matrix.standardize(X)
        Xd = (D - D.mean(axis=0)) / D.std(axis=0)
        kw.update(alpha=0.7, intercept=False)
        if engine == "gram_view":
            monkeypatch.setenv("ADELIE_HIP_STD_PANEL", "0")
    if engine == "gram_sparse":
        monkeypatch.setenv("ADELIE_HIP_SPARSE_PANEL", "0")
    if engine == "panel_sparse_groups":   # the group panel engine on compressed columns (blocks of whole groups, eigenbases per IRLS iteration)
        kw.update(groups=np.arange(0, p, 4), alpha=0.8)
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")   # the multi-CU engines from the first screen value on
    a = ad.grpnet(X, ad.glm.binomial(yb), **kw)
    monkeypatch.setenv("ADELIE_HIP_IRLS_REUSE", "0")
    b = ad.grpnet(X, ad.glm.binomial(yb), **kw)
    monkeypatch.delenv("ADELIE_HIP_IRLS_REUSE")
    o = ad.grpnet(oracle.dense(np.asfortranarray(Xd)), ad.glm.binomial(yb), **kw)
    assert a.error == "" and b.error == "" and o.error == ""
    assert a.counters["n_irls_iters"] > 30
    if engine == "panel_sparse_groups":
        assert a.counters["n_panel_blocks"] > 0   # (blocks of groups are rebuilt with their eigenbases per IRLS iteration)
    elif engine in ("panel_sparse", "panel_view"):
        assert a.counters["n_panel_blocks"] > 0 and a.counters["n_panel_grams"] < 0.7 * b.counters["n_panel_grams"]
    else:
        assert a.counters["n_panel_blocks"] == 0 and a.timers["gram_flops"] < 0.7 * b.timers["gram_flops"]   # Grams were kept
    # (a kept Gram couples the coordinates of a pass approximately: the passes contract more slowly and stop, by the same rule,
    # a little further from the common fixed point: 2e-7 here)
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-6
    assert np.abs(a.betas.toarray() - o.betas.toarray()).max() < 1e-6 and np.abs(a.intercepts - o.intercepts).max() < 1e-6
