"""Parity of the HIP path with the CPU oracle through the C ABI (`adelie_hip_grpnet_solve`), written like the reference's
own solver tests (tests/test_solver.py:596-649,899-975): same shapes, weighted, random penalties with zeros, warm restart.

Tolerance.  Both implementations run the same iterate sequence in exact arithmetic (same visiting order, same
predicates); they differ by floating-point summation order (wave-shuffle / MFMA reductions vs sequential sums).
f64: max-abs 1e-6 on beta and intercept (the reference's own threshold, tests/test_solver.py:444-445) — observed
differences are ~1e-13; support / screen sets must be identical.  f32: 1e-3 (tests/test_solver.py:745-746)."""
import numpy as np
import pytest

import adelie_amd as ad
from util import assert_same_path, kkt_gaussian, make_gaussian

pytestmark = pytest.mark.gpu


def _both(oracle, d, glm_f, dtype=np.float64, **kw):
    X = np.asarray(d["X"], dtype=dtype, order="F")
    a = ad.grpnet(ad.matrix.dense(X), glm_f(), **kw)
    b = ad.grpnet(oracle.dense(X), glm_f(), **kw)
    return a, b


@pytest.mark.parametrize("alpha", [1.0, 0.5])
@pytest.mark.parametrize("intercept", [True, False])
@pytest.mark.parametrize("n,p,G", [(10, 4, 2), (10, 100, 10), (10, 100, 20), (100, 23, 4), (100, 100, 50), (300, 120, 120)])
def test_gaussian_groups_weighted(hip, oracle, n, p, G, intercept, alpha):
    d = make_gaussian(n, p, G=G, seed=0, sparsity=0.95, weights=True, zero_pen=0.05)
    kw = dict(groups=d["groups"], penalty=d["penalty"], alpha=alpha, intercept=intercept, tol=1e-10, min_ratio=1e-1,
              lmda_path_size=30, early_exit=False)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"], weights=d["weights"]), **kw)
    assert_same_path(a, b, 1e-6)
    assert np.abs(a.grad - b.grad).max() < 1e-8 and np.abs(a.resid - b.resid).max() < 1e-8
    assert abs(a.rsq - b.rsq) < 1e-8 and abs(a.resid_sum - b.resid_sum) < 1e-10
    assert np.allclose(a.screen_vars, b.screen_vars, atol=1e-9)
    v = kkt_gaussian(d["X"], d["y"], a.weights, d["groups"], d["group_sizes"], d["penalty"], alpha, intercept,
                     a.betas, a.intercepts, a.lmdas, 1e-5)
    assert v < 3e-5 * max(1.0, float(np.max(np.abs(a.grad))))


def test_config1_lasso_default_settings(hip, oracle):
    """BASELINE.json configs[0]: dense 1000x200, ungrouped lasso, 100 lambdas, default tolerances."""
    d = make_gaussian(1000, 200, seed=0, sparsity=0.95)
    for ee in (True, False):
        a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"]), early_exit=ee)
        assert_same_path(a, b, 1e-6)
        # same algorithm, same visiting order: the work counters agree up to the rare coordinate whose
        # "changed" predicate (exact ak_old == ak compare, pin_naive:97) flips on a last-bit difference
        assert abs(a.counters["n_updates"] - b.counters["n_updates"]) <= 0.01 * b.counters["n_updates"]
        assert a.counters["n_cd_visits_screen"] == b.counters["n_cd_visits_screen"]
        assert a.counters["n_basil_iters"] == b.counters["n_basil_iters"]


def test_config3_shape_group10_enet(hip, oracle):
    d = make_gaussian(2000, 400, seed=1, sparsity=0.95)
    kw = dict(groups=np.arange(0, 400, 10), alpha=0.5, early_exit=False)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"]), **kw)
    assert_same_path(a, b, 1e-6)
    assert len(a.screen_transforms) == len(a.screen_set)


def test_ragged_group_sizes_incl_large(hip, oracle):
    rng = np.random.RandomState(5)
    n, p = 400, 151
    d = make_gaussian(n, p, seed=5)
    groups = np.array([0, 1, 2, 70, 71, 100, 150])  # sizes 1,1,68,1,29,50,1 (one group wider than a wavefront)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"]), groups=groups, alpha=0.7, early_exit=False,
                 lmda_path_size=20, min_ratio=1e-1, tol=1e-10)
    assert_same_path(a, b, 1e-6)


def test_f32(hip, oracle):
    d = make_gaussian(500, 100, seed=2)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"].astype(np.float32)), dtype=np.float32, early_exit=False,
                 lmda_path_size=30, min_ratio=5e-2)
    assert a.betas.dtype == np.float32
    assert len(a.lmdas) == len(b.lmdas) == 30
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-3
    assert np.abs(a.intercepts - b.intercepts).max() < 1e-3


def test_strong_rule_and_user_path_above_lmda_max(hip, oracle):
    d = make_gaussian(200, 80, seed=3)
    ref = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), lmda_path_size=0)
    # (no lambda exactly equal to lmda_max: at that point the KKT test `abs_grad > lmda*alpha*pen` is an exact tie
    #  and a last-bit difference in the recomputed lmda_max legitimately changes the screening order)
    path = ref.lmda_max * np.array([2.0, 1.5, 0.999, 0.7, 0.4, 0.2, 0.1])
    for rule in ("strong", "pivot"):
        a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"]), lmda_path=path, screen_rule=rule, early_exit=False,
                     tol=1e-10)
        assert len(a.lmdas) == 7
        assert_same_path(a, b, 1e-6)
        assert a.betas[:2].nnz == 0  # lambdas above lmda_max: null model, solutions saved (solver_base.hpp:553-591)


def test_lmda_path_size_zero_returns_lmda_max_only(hip, oracle):
    d = make_gaussian(120, 50, seed=4, weights=True)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"], weights=d["weights"]), lmda_path_size=0)
    assert len(a.lmdas) == 0 and a.error == ""
    assert np.isclose(a.lmda_max, b.lmda_max, rtol=1e-12)


def test_warm_start_continuation(hip, oracle):
    """tests/test_solver.py:633-649: continue from a solved state down a longer path."""
    d = make_gaussian(150, 90, G=30, seed=6, weights=True)
    glm = lambda: ad.glm.gaussian(d["y"], weights=d["weights"])
    Xg, Xo = ad.matrix.dense(d["X"]), oracle.dense(d["X"])
    kw = dict(groups=d["groups"], tol=1e-10, early_exit=False)
    s1g = ad.grpnet(Xg, glm(), lmda_path_size=12, min_ratio=0.3, **kw)
    s1o = ad.grpnet(Xo, glm(), lmda_path_size=12, min_ratio=0.3, **kw)
    nxt = [s1o.lmdas[-1] * 0.8, s1o.lmdas[-1] * 0.6]
    s2g = ad.grpnet(Xg, glm(), lmda_path=nxt, warm_start=s1g, **kw)
    s2o = ad.grpnet(Xo, glm(), lmda_path=nxt, warm_start=s1o, **kw)
    assert_same_path(s2g, s2o, 1e-6)


def test_errors_and_partial_paths(hip, oracle):
    d = make_gaussian(60, 40, seed=1)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"]), max_iters=1, early_exit=False)
    assert a.error == b.error and a.error.startswith("adelie_core solver: max coordinate descents")
    assert len(a.lmdas) == len(b.lmdas)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"]), max_screen_size=3, early_exit=False)
    assert a.error == b.error == "adelie_core solver: maximum screen set size reached."
    assert len(a.lmdas) == len(b.lmdas) > 0
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-6
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"]), max_active_size=2, early_exit=False)
    assert a.error == b.error == "adelie_core solver: Maximum number of active groups reached."
    with pytest.raises(RuntimeError, match="alpha must be in"):
        ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), alpha=-0.1)


def test_exit_cond_callback(hip):
    d = make_gaussian(100, 40, seed=8)
    st = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), early_exit=False,
                   exit_cond=lambda s: s.n_solutions >= 7)
    assert len(st.lmdas) == 7


@pytest.mark.parametrize("n,p,G", [(200, 30, 30), (150, 60, 12)])
def test_binomial_irls(hip, oracle, n, p, G):
    rng = np.random.RandomState(4)
    d = make_gaussian(n, p, G=G, seed=9)
    eta = d["X"][:, :3] @ np.array([1.0, -2.0, 0.5])
    y = rng.binomial(1, 1 / (1 + np.exp(-eta))).astype(float)
    kw = dict(groups=d["groups"], tol=1e-10, irls_tol=1e-10, early_exit=False, lmda_path_size=15, min_ratio=5e-2)
    a, b = _both(oracle, d, lambda: ad.glm.binomial(y), **kw)
    assert_same_path(a, b, 1e-6)
    assert np.abs(a.eta - b.eta).max() < 1e-6 and abs(a.beta0 - b.beta0) < 1e-6
    assert a.counters["n_irls_iters"] == b.counters["n_irls_iters"]
    assert np.isclose(a.loss_null, b.loss_null, rtol=1e-10)


def test_gaussian_forced_through_irls(hip, oracle):
    d = make_gaussian(80, 40, G=10, seed=2)
    kw = dict(groups=d["groups"], tol=1e-12, irls_tol=1e-12, early_exit=False, lmda_path_size=15, min_ratio=1e-1)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"], opt=False), **kw)
    assert_same_path(a, b, 1e-6)
    c = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
    assert np.abs(a.betas.toarray() - c.betas.toarray()).max() < 1e-5


def test_snp_design_equals_densified(hip, oracle):
    """tests/test_solver.py:749-818: the SNP matrix and its densified copy give the same path (atol 1e-3 there)."""
    rng = np.random.RandomState(3)
    n, p = 600, 80
    calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8)
    imp = ad.matrix.compute_impute(calldata)
    Xd = np.asfortranarray(np.where(calldata < 0, imp[None], calldata).astype(np.float64))
    beta = rng.normal(size=p) * (rng.uniform(size=p) < 0.1)
    y = Xd @ beta + rng.normal(size=n)
    kw = dict(early_exit=False, tol=1e-10, lmda_path_size=30)
    s_snp = ad.grpnet(ad.matrix.snp_calldata(calldata), ad.glm.gaussian(y), **kw)
    s_den = ad.grpnet(ad.matrix.dense(Xd), ad.glm.gaussian(y), **kw)
    s_orc = ad.grpnet(oracle.snp_calldata(calldata), ad.glm.gaussian(y), **kw)
    assert_same_path(s_snp, s_den, 1e-6)
    assert_same_path(s_snp, s_orc, 1e-6)
    yb = rng.binomial(1, 1 / (1 + np.exp(-(Xd @ beta)))).astype(float)
    kb = dict(early_exit=False, tol=1e-10, irls_tol=1e-10, lmda_path_size=12, min_ratio=1e-1)
    b_snp = ad.grpnet(ad.matrix.snp_calldata(calldata), ad.glm.binomial(yb), **kb)
    b_orc = ad.grpnet(oracle.snp_calldata(calldata), ad.glm.binomial(yb), **kb)
    assert_same_path(b_snp, b_orc, 1e-6)


def test_caller_state_untouched_and_result_surface(hip):
    """state.solve() returns a NEW state (py_state.cpp:1218-1226); the attribute surface of SURVEY.md 8b exists."""
    d = make_gaussian(100, 30, seed=1)
    st = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), early_exit=False, lmda_path_size=10)
    for name in ["betas", "intercepts", "devs", "lmdas", "duals", "lmda_max", "lmda_path", "lmda", "screen_set",
                 "screen_begins", "screen_beta", "screen_is_active", "active_set_size", "active_set", "grad", "abs_grad",
                 "resid", "resid_sum", "rsq", "X_means", "y_mean", "y_var", "loss_null", "loss_full", "screen_X_means",
                 "screen_transforms", "screen_vars", "weights", "X", "benchmark_screen", "benchmark_fit_screen",
                 "benchmark_fit_active", "benchmark_kkt", "benchmark_invariance", "n_valid_solutions", "active_sizes",
                 "screen_sizes", "error", "total_time", "_offsets", "_glm", "_X"]:
        assert hasattr(st, name), name
    assert st.betas.shape == (10, 30) and st.betas.indices.dtype in (np.int32, np.int64)
    assert st.screen_is_active.dtype == bool and st.active_set.shape == (30,)


@pytest.mark.parametrize("engine", ["panel", "panel-seq"])
@pytest.mark.parametrize("n,p,alpha", [(400, 300, 1.0), (1500, 700, 0.6), (1027, 520, 1.0)])
def test_block_cd_passes_match_oracle(hip, oracle, monkeypatch, n, p, alpha, engine):
    """Forces the multi-CU block Gauss-Seidel passes at sizes the oracle checks in seconds: several 128-visit blocks per
    pass, ragged last block, active-set growth inside screen passes, n not a multiple of the row slice.
    engine "panel": residual-based blocks with cached diagonal Gram blocks and the look-ahead launches (solve of block j
    fused with the step for block j+1, cross-block correction; kernels_cd_panel.hip, the default);
    engine "panel-seq": the same without look-ahead (step -> reduce -> solve strictly in sequence, the form IRLS and constrained
    problems run).  (The full-Gram block engines of kernels_cd_block*.hip are the covariance method's pin solver: tests/test_cov.py.)"""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    monkeypatch.setenv("ADELIE_HIP_LOOKAHEAD", "0" if engine == "panel-seq" else "1")
    d = make_gaussian(n, p, seed=11, sparsity=0.5, weights=True)
    # beta is resolved to ~sqrt(tol) by the stopping rule; 1e-14 makes two trajectories that differ in the order of
    # the first activation (the lambda_0 == lmda_max tie) agree to 1e-6
    kw = dict(alpha=alpha, tol=1e-14, early_exit=False, lmda_path_size=25, min_ratio=1e-2)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"], weights=d["weights"]), **kw)
    assert_same_path(a, b, 1e-6)
    assert a.active_set_size > 128
    assert abs(a.counters["n_updates"] - b.counters["n_updates"]) <= 0.01 * b.counters["n_updates"]
    # (the oracle also counts the empty active pass of fits with an empty screen set; the device skips those fits)
    assert abs(a.counters["n_cd_passes_active"] - b.counters["n_cd_passes_active"]) <= 8
    # errors keep working through the block path
    e1 = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), max_active_size=5, early_exit=False)
    assert e1.error == "adelie_core solver: Maximum number of active groups reached."
    e2 = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), max_iters=3, early_exit=False)
    assert e2.error.startswith("adelie_core solver: max coordinate descents")


@pytest.mark.parametrize("engine", ["panel", "panel-seq"])
@pytest.mark.parametrize("alpha", [1.0, 0.5])
def test_group_block_cd_passes_match_oracle(hip, oracle, monkeypatch, alpha, engine):
    """Forces the multi-CU block passes for grouped problems (kernels_cd_block_group.hip): mixed group sizes
    (1..40), several blocks per pass, groups activated inside screen passes.  engine "panel": residual-based blocks with
    cached diagonal Gram blocks, eigenbases from those blocks (the default); "panel-seq": without look-ahead."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    monkeypatch.setenv("ADELIE_HIP_LOOKAHEAD", "0" if engine == "panel-seq" else "1")
    rng = np.random.RandomState(7)
    n, p = 1200, 640
    d = make_gaussian(n, p, seed=13, sparsity=0.6, weights=True)
    sizes = []
    while sum(sizes) < p:
        sizes.append(int(rng.choice([1, 1, 2, 5, 10, 40])))
    sizes[-1] -= sum(sizes) - p
    if sizes[-1] <= 0:
        sizes.pop()
        sizes[-1] += p - sum(sizes)
    groups = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    kw = dict(groups=groups, alpha=alpha, tol=1e-14, early_exit=False, lmda_path_size=20, min_ratio=2e-2)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"], weights=d["weights"]), **kw)
    assert_same_path(a, b, 1e-6)
    assert a.active_set_size > 30
    assert abs(a.counters["n_updates"] - b.counters["n_updates"]) <= 0.02 * b.counters["n_updates"] + 5
    assert a.counters["n_panel_blocks"] > 0


@pytest.mark.parametrize("layout", ["bucket_edges", "short_block_ends"])
def test_group_update_buckets_match_oracle(hip, oracle, monkeypatch, layout):
    """Group sizes on both sides of every bucket of the rotated group solve's gradient update (round 6: the block entries of a
    chunk are fetched in one batch of q rounded up to 4 / 8 / 12 / 16 columns, the surplus columns meeting zero changes; q > 16
    keeps the four-column loop) on the look-ahead panel engine, several blocks per pass, against the oracle."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    n, p = 1500, 760
    d = make_gaussian(n, p, seed=29, sparsity=0.5, weights=True)
    sizes = []
    if layout == "bucket_edges":
        cyc = [2, 3, 4, 5, 8, 9, 12, 13, 16, 17, 1, 24]
        while sum(sizes) + cyc[len(sizes) % len(cyc)] <= p:
            sizes.append(cyc[len(sizes) % len(cyc)])
        if sum(sizes) < p:
            sizes.append(p - sum(sizes))
    else:
        # every 128-value block ends in a group whose rounded-up size reaches past the values that follow it (2 -> 4 columns
        # with ONE value behind it, 9 -> 12 with two, 13 -> 16 with one): the surplus columns must stay inside the columns the
        # solve has written (the constraint fuzzer found them in uninitialised LDS, ROUNDS.md round 6)
        for tail in ([2, 1], [9, 2], [13, 1], [5, 1, 1], [2, 1]):
            body = 128 - sum(tail)
            sizes += [5] * (body // 5) + ([body % 5] if body % 5 else []) + tail
        sizes.append(p - sum(sizes))
    groups = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    for alpha in (1.0, 0.4):
        kw = dict(groups=groups, alpha=alpha, tol=1e-14, early_exit=False, lmda_path_size=18, min_ratio=3e-2)
        a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"], weights=d["weights"]), **kw)
        assert_same_path(a, b, 1e-6)
        assert a.active_set_size > 40 and a.counters["n_panel_blocks"] > 10
        act = set(np.asarray(sizes)[np.unique(np.searchsorted(groups, np.flatnonzero(a.betas.toarray()[-1]), side="right") - 1)].tolist())
        if layout == "bucket_edges":
            assert {4, 5, 8, 9, 12, 13, 16, 17} <= act   # (every bucket edge was a changed group at the end of the path)
        else:
            assert {2, 9, 13} <= act


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_panel_engine_binomial_snp(hip, oracle, monkeypatch, dtype):
    """IRLS on a 2-bit SNP design through the panel engine: the diagonal blocks are rebuilt per IRLS iteration (weights
    change), the residual of the working response is updated block by block (solver_glm_naive.hpp:234-459)."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    rng = np.random.default_rng(5)
    n, p = 1501, 400
    u = rng.random((n, p))
    cd = np.zeros((n, p), dtype=np.int8, order="F")
    cd[u < 0.25] = 1
    cd[(u >= 0.25) & (u < 0.30)] = 2
    cd[u >= 0.92] = -9
    imp = ad.matrix.compute_impute(cd)
    Xd = np.where(cd < 0, imp[None, :], cd).astype(np.float64)
    beta = rng.standard_normal(p) * (rng.random(p) < 0.1)
    eta = Xd @ beta
    eta = (eta - eta.mean()) / eta.std()
    y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    kw = dict(early_exit=False, lmda_path_size=12, min_ratio=0.05, tol=1e-12 if dtype == np.float64 else 1e-7,
              irls_tol=1e-10 if dtype == np.float64 else 1e-6)
    a = ad.grpnet(ad.matrix.snp_calldata(cd, imp, dtype=dtype), ad.glm.binomial(y, dtype=dtype), **kw)
    b = ad.grpnet(oracle.snp_calldata(cd, imp, dtype=dtype), ad.glm.binomial(y, dtype=dtype), **kw)
    assert a.error == "" and b.error == ""
    assert a.counters["n_panel_blocks"] > 0 and a.counters["n_panel_grams"] > 0
    assert_same_path(a, b, 1e-6 if dtype == np.float64 else 5e-3)


@pytest.mark.parametrize("family", ["gaussian", "binomial"])
def test_snp_paths_with_the_nibble_table_sweeps(hip, oracle, family):
    """A 2-bit design large enough that the solver's own sweeps (full design, column lists of the KKT check, the squared design
    of the IRLS variances) take the nibble-table form (>= 4096 rows, >= 512 columns) at the default engine thresholds."""
    rng = np.random.default_rng(17)
    n, p = 4500, 1500
    u = rng.random((n, p))
    cd = np.zeros((n, p), dtype=np.int8, order="F")
    cd[u < 0.25] = 1
    cd[(u >= 0.25) & (u < 0.30)] = 2
    cd[u >= 0.93] = -9
    imp = ad.matrix.compute_impute(cd)
    Xd = np.where(cd < 0, imp[None, :], cd).astype(np.float64)
    beta = rng.standard_normal(p) * (rng.random(p) < 0.05)
    eta = Xd @ beta
    eta = (eta - eta.mean()) / eta.std()
    if family == "gaussian":
        yg = eta + 0.5 * rng.standard_normal(n)
        glm = lambda: ad.glm.gaussian(yg)
    else:
        y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
        glm = lambda: ad.glm.binomial(y)
    kw = dict(early_exit=False, lmda_path_size=15, min_ratio=0.05, tol=1e-12, irls_tol=1e-10, progress_bar=False)
    a = ad.grpnet(ad.matrix.snp_calldata(cd, imp), glm(), **kw)
    b = ad.grpnet(oracle.snp_calldata(cd, imp), glm(), **kw)
    assert a.error == "" and b.error == ""
    assert (a.betas[-1].toarray() != 0).sum() > 50
    assert_same_path(a, b, 1e-6)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("bsz", [32, 64])
def test_panel_engine_small_blocks(hip, oracle, monkeypatch, bsz, dtype):
    """Panel blocks of 32 / 64 visits (the IRLS default is 64): diagonal blocks come from the lower-triangle MFMA kernel
    (syrk64), ragged last block, n not a multiple of the K tile."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    monkeypatch.setenv("ADELIE_HIP_PANEL_BSZ", str(bsz))
    d = make_gaussian(1203, 450, seed=17, sparsity=0.6, weights=True)
    f32 = dtype == np.float32
    kw = dict(alpha=0.8, tol=1e-7 if f32 else 1e-14, early_exit=False, lmda_path_size=20, min_ratio=2e-2)
    a, b = _both(oracle, d, lambda: ad.glm.gaussian(d["y"], weights=d["weights"], dtype=dtype), dtype=dtype, **kw)
    assert a.counters["n_panel_blocks"] > 0
    assert_same_path(a, b, 5e-3 if f32 else 1e-6)


def test_group_panel_binomial(hip, oracle, monkeypatch):
    """Grouped binomial path (IRLS) through the group panel engine: diagonal blocks and eigenbases are rebuilt per IRLS
    iteration from the IRLS weights (solver_glm_naive.hpp:361-385)."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    rng = np.random.RandomState(5)
    n, p, gs = 900, 240, 6
    X = np.asfortranarray(rng.normal(size=(n, p)))
    beta = rng.normal(size=p) * np.repeat(rng.uniform(size=p // gs) < 0.3, gs)
    eta = X @ beta
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta / eta.std()))).astype(np.float64)
    kw = dict(groups=np.arange(0, p, gs), alpha=0.7, early_exit=False, lmda_path_size=12, min_ratio=0.1, tol=1e-12,
              irls_tol=1e-10)
    a = ad.grpnet(ad.matrix.dense(X), ad.glm.binomial(y), **kw)
    b = ad.grpnet(oracle.dense(X), ad.glm.binomial(y), **kw)
    assert a.counters["n_panel_blocks"] > 0
    assert_same_path(a, b, 1e-6)


@pytest.mark.parametrize("grouped", [False, True])
@pytest.mark.parametrize("family", ["gaussian", "binomial"])
def test_warm_start_on_panel_engine(hip, oracle, monkeypatch, grouped, family):
    """Warm restart (tests/test_solver.py:633-649) with the panel engine active from the first fit: the solver starts from
    a non-empty screen / active set, so every screen-derived quantity and diagonal block is built for pre-existing groups."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    d = make_gaussian(700, 360, seed=31, sparsity=0.8, weights=True)
    groups = np.arange(0, 360, 4) if grouped else np.arange(360)
    if family == "gaussian":
        glm = lambda: ad.glm.gaussian(d["y"], weights=d["weights"])
    else:
        yb = (d["y"] > np.median(d["y"])).astype(np.float64)
        glm = lambda: ad.glm.binomial(yb, weights=d["weights"])
    Xg, Xo = ad.matrix.dense(d["X"]), oracle.dense(d["X"])
    kw = dict(groups=groups, tol=1e-13, early_exit=False)
    if family == "binomial":
        kw["irls_tol"] = 1e-10
    s1g = ad.grpnet(Xg, glm(), lmda_path_size=10, min_ratio=0.4, **kw)
    s1o = ad.grpnet(Xo, glm(), lmda_path_size=10, min_ratio=0.4, **kw)
    assert s1g.active_set_size > 0
    nxt = [s1o.lmdas[-1] * 0.8, s1o.lmdas[-1] * 0.6, s1o.lmdas[-1] * 0.45]
    s2g = ad.grpnet(Xg, glm(), lmda_path=nxt, warm_start=s1g, **kw)
    s2o = ad.grpnet(Xo, glm(), lmda_path=nxt, warm_start=s1o, **kw)
    assert s2g.counters["n_panel_blocks"] > 0
    assert_same_path(s2g, s2o, 1e-6)


@pytest.mark.parametrize("grouped", [False, True])
def test_panel_engine_error_paths_match_oracle(hip, oracle, monkeypatch, grouped):
    """max_active_size / max_iters / max_screen_size through the panel engine (scalar and group blocks): same error strings,
    same partial paths, and the pre-fit state is restored (residual undone) exactly like the reference does
    (solver_gaussian_naive.hpp:286-290,326-329)."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    d = make_gaussian(600, 280, seed=41, sparsity=0.6, weights=True)
    groups = np.arange(0, 280, 5) if grouped else np.arange(280)
    glm = lambda: ad.glm.gaussian(d["y"], weights=d["weights"])
    Xg, Xo = ad.matrix.dense(d["X"]), oracle.dense(d["X"])
    base = dict(groups=groups, early_exit=False, lmda_path_size=20, min_ratio=0.05, tol=1e-12)
    for extra in [dict(max_active_size=7), dict(max_iters=6), dict(max_screen_size=9)]:
        a = ad.grpnet(Xg, glm(), **base, **extra)
        b = ad.grpnet(Xo, glm(), **base, **extra)
        assert a.error == b.error and a.error != "", (extra, a.error, b.error)
        assert len(a.lmdas) == len(b.lmdas), extra
        if len(a.lmdas):
            assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-6
        # invariants of the returned (rolled back) state
        np.testing.assert_allclose(a.resid, b.resid, atol=1e-8)
        np.testing.assert_allclose(a.screen_beta[:len(b.screen_beta)], b.screen_beta, atol=1e-8)
        assert abs(a.rsq - b.rsq) < 1e-8 and a.active_set_size == b.active_set_size


def test_exit_cond_on_panel_engine(hip, monkeypatch):
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    d = make_gaussian(500, 200, seed=43, sparsity=0.7)
    st = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), early_exit=False, lmda_path_size=30,
                   exit_cond=lambda s: s.n_solutions >= 9)
    assert st.error == "" and len(st.lmdas) == 9 and st.counters["n_panel_blocks"] > 0


@pytest.mark.parametrize("hook,values", [("ADELIE_HIP_SPECULATE", ["0", "1"]), ("ADELIE_HIP_LOOKAHEAD", ["0", "1"]),
                                         ("ADELIE_HIP_IRLS_REUSE", ["0", "0.01", "0.2"])])
def test_engine_hooks_do_not_change_results(hip, oracle, monkeypatch, hook, values):
    """The hooks that survive (common.hpp::Hooks) are scheduling changes (speculative pass, look-ahead) or a bounded change of the
    iterate sequence (IRLS block reuse): every setting gives the oracle's path on a lasso, a grouped and a binomial
    problem whose screen sets span several blocks of different tile classes."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    rng = np.random.RandomState(11)
    n, p = 700, 330
    d = make_gaussian(n, p, seed=11, sparsity=0.5)
    groups = np.concatenate([[0, 1, 2], np.arange(3, 300, 9), [300, 301, 320]])   # sizes 1,1,1, 9 x 33, 1, 19, 10
    yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-d["X"][:, :5] @ np.ones(5)))).astype(float)
    cases = [
        (lambda: ad.glm.gaussian(d["y"]), dict(early_exit=False, lmda_path_size=12, min_ratio=0.02, tol=1e-12)),
        (lambda: ad.glm.gaussian(d["y"]), dict(groups=groups, alpha=0.6, early_exit=False, lmda_path_size=12, min_ratio=0.02,
                                              tol=1e-12)),
        (lambda: ad.glm.binomial(yb), dict(early_exit=False, lmda_path_size=10, min_ratio=0.05, tol=1e-12, irls_tol=1e-11)),
    ]
    for mk, kw in cases:
        ref = ad.grpnet(oracle.dense(d["X"]), mk(), progress_bar=False, **kw)
        for v in values:
            monkeypatch.setenv(hook, v)
            st = ad.grpnet(ad.matrix.dense(d["X"]), mk(), progress_bar=False, **kw)
            assert st.error == "" and len(st.lmdas) == len(ref.lmdas)
            assert st.counters["n_panel_blocks"] > 0
            # the design is strongly collinear (p / n ~ 0.5, path down to 2 % of lambda_max): two correct runs may stop one
            # pass apart, which is worth ~1e-7 in beta at this tol; a wrong block shows up at 1e-3 and above
            assert np.abs(st.betas.toarray() - ref.betas.toarray()).max() < 1e-6, (hook, v)
            assert np.abs(st.intercepts - ref.intercepts).max() < 1e-6


@pytest.mark.gpu
def test_speculative_first_active_pass_is_bit_identical(hip, oracle, monkeypatch):
    """The first active-set pass of the next lambda, enqueued behind the invariance sweep while the host screens
    (Solver::spec_*), performs the same operations in the same order as the unspeculated path: identical bits, with and
    without KKT failures (forced here by a tiny strong-rule screen budget), early exit and a live-state read in between."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    d = make_gaussian(1500, 1200, seed=21, sparsity=0.5)
    X = ad.matrix.dense(d["X"])
    runs = {}
    for kw_name, kw in [("plain", dict(early_exit=False, lmda_path_size=30, min_ratio=0.02, tol=1e-10)),
                        ("kkt_failures", dict(early_exit=False, lmda_path_size=30, min_ratio=0.02, tol=1e-10,
                                              screen_rule="strong", max_screen_size=1200)),
                        ("early_exit", dict(early_exit=True, lmda_path_size=40, min_ratio=0.001, tol=1e-10)),
                        ("error_mid_path", dict(early_exit=False, lmda_path_size=30, min_ratio=0.02, tol=1e-10,
                                                max_screen_size=500))]:
        for spec in ("0", "1"):
            monkeypatch.setenv("ADELIE_HIP_SPECULATE", spec)
            st = ad.grpnet(X, ad.glm.gaussian(d["y"]), progress_bar=False, **kw)
            assert (st.error == "") == (kw_name != "error_mid_path"), st.error
            runs[kw_name, spec] = st
        a, b = runs[kw_name, "0"], runs[kw_name, "1"]
        assert a.error == b.error and len(a.lmdas) == len(b.lmdas)
        assert a.counters["n_speculated"] == 0 and b.counters["n_speculated"] > 5
        assert np.array_equal(a.betas.toarray(), b.betas.toarray())
        assert np.array_equal(a.intercepts, b.intercepts) and np.array_equal(a.devs, b.devs)
        assert np.array_equal(a.resid, b.resid) and np.array_equal(a.screen_beta, b.screen_beta)
        assert np.array_equal(a.grad, b.grad) and a.rsq == b.rsq
        for k in ("n_cd_passes_active", "n_cd_passes_screen", "n_updates", "n_panel_blocks", "n_panel_cols", "n_basil_iters"):
            assert a.counters[k] == b.counters[k], k
    assert runs["early_exit", "1"].counters["n_spec_rollbacks"] >= 1  # the pass enqueued before the exit was taken back
    # a live-state read of the device-resident residual in the middle of the path takes the pass back and the path goes on
    monkeypatch.setenv("ADELIE_HIP_SPECULATE", "1")
    seen = []

    def peek(s):
        if s.n_solutions in (12, 20):
            seen.append(np.array(s.resid, copy=True))
        return False

    c = ad.grpnet(X, ad.glm.gaussian(d["y"]), progress_bar=False, exit_cond=peek, early_exit=False, lmda_path_size=30,
                  min_ratio=0.02, tol=1e-10)
    assert len(seen) == 2 and c.counters["n_spec_rollbacks"] >= 2
    assert np.array_equal(c.betas.toarray(), runs["plain", "0"].betas.toarray())
    ref = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), progress_bar=False, early_exit=False, lmda_path_size=30,
                    min_ratio=0.02, tol=1e-10)
    assert np.abs(c.betas.toarray() - ref.betas.toarray()).max() < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["groups", "multigaussian", "short_active_sets"])
def test_speculative_pass_group_engine_and_short_passes(hip, monkeypatch, kind):
    """The same on the group panel engine (single- and multi-response) and for active sets shorter than the look-ahead's
    minimum (plain passes)."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    d = make_gaussian(1200, 900, seed=23, sparsity=0.5)
    X = ad.matrix.dense(d["X"])
    kw = dict(early_exit=False, lmda_path_size=25, min_ratio=0.03, tol=1e-10, progress_bar=False)
    if kind == "groups":
        mk = lambda: ad.glm.gaussian(d["y"])
        kw.update(groups=np.arange(0, 900, 6), alpha=0.7)
    elif kind == "multigaussian":
        rng = np.random.RandomState(5)
        Y = np.stack([d["y"], d["y"][::-1] + rng.normal(size=1200), rng.normal(size=1200)], axis=1)
        mk = lambda: ad.glm.multigaussian(Y)
    else:
        mk = lambda: ad.glm.gaussian(d["y"])
        kw.update(lmda_path_size=12, min_ratio=0.5)  # the active set stays below three blocks
    out = {}
    for spec in ("0", "1"):
        monkeypatch.setenv("ADELIE_HIP_SPECULATE", spec)
        out[spec] = ad.grpnet(X, mk(), **kw)
        assert out[spec].error == ""
    a, b = out["0"], out["1"]
    assert a.counters["n_speculated"] == 0 and b.counters["n_speculated"] > 3
    assert np.array_equal(a.betas.toarray(), b.betas.toarray()) and np.array_equal(a.intercepts, b.intercepts)
    assert np.array_equal(a.resid, b.resid) and np.array_equal(a.devs, b.devs)
    for k in ("n_cd_passes_active", "n_cd_passes_screen", "n_updates", "n_panel_blocks", "n_panel_cols"):
        assert a.counters[k] == b.counters[k], k


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["binomial", "poisson"])
def test_irls_block_reuse_against_rebuild_every_iteration(hip, oracle, monkeypatch, family):
    """DESIGN 4, third deliberate deviation: with ADELIE_HIP_IRLS_REUSE=theta a diagonal block built under earlier IRLS weights
    stays in use while every weight moved by at most theta; the fixed point of the passes is unchanged, the iterates inside a
    pass move by O(theta).  theta = 0 (rebuild per iteration, the reference's arithmetic) against the default on a problem with
    near-saturated working weights (large linear predictor: binomial weights down to 1e-9, poisson means up to e^9)."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    rng = np.random.RandomState(17)
    n, p = 900, 300
    X = np.asfortranarray(rng.normal(size=(n, p)))
    beta = np.zeros(p)
    beta[:6] = [4.0, -3.0, 2.5, 2.0, -2.0, 1.5] if family == "binomial" else [0.9, -0.8, 0.7, 0.6, -0.5, 0.4]
    eta = X @ beta
    if family == "binomial":
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(float)
        mk = lambda: ad.glm.binomial(y)
    else:
        y = rng.poisson(np.exp(np.clip(eta, None, 9.0))).astype(float)
        mk = lambda: ad.glm.poisson(y)
    kw = dict(early_exit=False, lmda_path_size=15, min_ratio=0.02, tol=1e-12, irls_tol=1e-11, progress_bar=False)
    out = {}
    for theta in ("0", None):          # None: the library's DEFAULT threshold (common.hpp: 10 % accumulated weight drift)
        if theta is None:
            monkeypatch.delenv("ADELIE_HIP_IRLS_REUSE", raising=False)
        else:
            monkeypatch.setenv("ADELIE_HIP_IRLS_REUSE", theta)
        out[theta] = ad.grpnet(ad.matrix.dense(X), mk(), **kw)
    a, b = out["0"], out[None]
    assert a.error == "" and b.error == "" and len(a.lmdas) == len(b.lmdas) == 15
    assert a.counters["n_panel_blocks"] > 0
    assert b.timers["n_gram_launches"] < a.timers["n_gram_launches"]   # (blocks really were kept under the default)
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-7
    assert np.abs(a.intercepts - b.intercepts).max() < 1e-7
    assert sorted(a.screen_set.tolist()) == sorted(b.screen_set.tolist())
    # ... and both against the oracle, whose variances and blocks are always those of the current weights
    o = ad.grpnet(oracle.dense(X), mk(), **kw)
    assert o.error == ""
    for s_ in (a, b):
        assert np.abs(s_.betas.toarray() - o.betas.toarray()).max() < 1e-6
        assert np.abs(s_.intercepts - o.intercepts).max() < 1e-6


def test_penalty_l2_is_the_quadratic_parts_own_factor(hip, oracle):
    """adelie_hip_grpnet_args::penalty_l2 (ABI 8; adelie_amd.solver passes it for elastic nets on standardized views): separate
    factors for the quadratic part of the penalty.  Equal to `penalty` it changes nothing (bit for bit); a constant multiple c of
    `penalty` is the elastic net with  lmda' = lmda (alpha + (1 - alpha) c),  alpha' = alpha / (alpha + (1 - alpha) c)  -- checked
    against the ORACLE on that re-parametrised problem (which knows no penalty_l2), Gaussian and binomial, single-workgroup and
    panel engines; groups of several coefficients refuse it."""
    rng = np.random.RandomState(77)
    n, p = 400, 300
    X = np.asfortranarray(rng.normal(size=(n, p)))
    beta = np.zeros(p)
    beta[rng.choice(p, 20, replace=False)] = rng.normal(size=20)
    y = X @ beta + 0.5 * rng.normal(size=n)
    pen = rng.uniform(0.5, 2.0, p)
    Xd = ad.matrix.dense(X)
    alpha, c = 0.6, 2.5
    lmdas = 0.5 * np.logspace(0, -2, 12)
    for glm, extra in [(ad.glm.gaussian(y), dict(tol=1e-13)), (ad.glm.binomial((y > 0).astype(float)), dict(tol=1e-12, irls_tol=1e-12))]:
        kw = dict(early_exit=False, progress_bar=False, penalty=pen, **extra)
        a0 = ad.grpnet(Xd, glm, alpha=alpha, lmda_path=lmdas, **kw)
        a1 = ad.grpnet(Xd, glm, alpha=alpha, lmda_path=lmdas, _penalty_l2=pen.copy(), **kw)
        assert a0.error == "" and a1.error == "" and np.array_equal(a0.betas.toarray(), a1.betas.toarray())
        a = ad.grpnet(Xd, glm, alpha=alpha, lmda_path=lmdas, _penalty_l2=c * pen, **kw)
        scale = alpha + (1 - alpha) * c
        o = ad.grpnet(oracle.dense(X), glm, alpha=alpha / scale, lmda_path=lmdas * scale, **kw)
        assert a.error == "" and o.error == ""
        assert np.abs(a.betas.toarray() - o.betas.toarray()).max() < (1e-8 if "irls_tol" not in extra else 1e-6)
        assert np.abs(a.intercepts - o.intercepts).max() < 1e-6
        assert max(a.screen_sizes) > 128   # (the panel engines took over along the path)
    with pytest.raises(RuntimeError, match="penalty_l2 needs groups of one coefficient"):
        ad.grpnet(Xd, ad.glm.gaussian(y), groups=np.arange(0, p, 2), alpha=0.5, _penalty_l2=np.ones(p // 2), progress_bar=False)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["gaussian_plain_128", "gaussian_default", "binomial", "binomial_weights_offsets"])
def test_snp_word_per_lane_step_shapes(hip, oracle, monkeypatch, case):
    """The 2-bit panel step of round 6 (one 32-bit word per lane, pair tables in the residual update, the last workgroups summing
    the partials; n >= 16384) on shapes the configuration-sized tests do not reach: a row count that ends inside a word and
    inside a slice (20 037 = 9 x 2048 + 1605 = ... + 100 words + 5 calls), blocks of 128 visits in the sequential form (two
    pair-table batches; ADELIE_HIP_LOOKAHEAD=0 puts every Gaussian pass there), the look-ahead passes' two-block opening step
    (which stays on the generic body), blocks of 64 under IRLS, observation weights and offsets.  Against the oracle at tolerances
    that resolve the fixed point.  (No case without an intercept: on uncentred 2-bit columns the reference's coordinate descent
    itself needs 20 minutes for this problem.)"""
    rng = np.random.RandomState(41)
    n, p = 20037, 700
    cd = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8)
    cd = np.asfortranarray(cd)
    imp = np.array([cd[:, j][cd[:, j] >= 0].mean() for j in range(p)])
    Xv = np.where(cd < 0, imp[None, :], cd).astype(float)
    beta = np.zeros(p)
    beta[rng.choice(p, 250, replace=False)] = rng.normal(size=250) * 0.5
    eta = Xv @ beta
    eta = (eta - eta.mean()) / eta.std()
    kw = dict(early_exit=False, lmda_path_size=14, progress_bar=False)
    if case.startswith("gaussian"):
        y = eta + 0.3 * rng.normal(size=n)
        mk = lambda: ad.glm.gaussian(y)
        kw.update(tol=1e-13, min_ratio=0.01)
        if case == "gaussian_plain_128":
            monkeypatch.setenv("ADELIE_HIP_LOOKAHEAD", "0")
    else:
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-2 * eta))).astype(float)
        kw.update(tol=1e-12, irls_tol=1e-11, min_ratio=0.03)
        if case == "binomial":
            mk = lambda: ad.glm.binomial(y)
        else:
            w = rng.uniform(0.2, 1.8, n)
            mk = lambda: ad.glm.binomial(y, weights=w / w.sum())
            kw.update(offsets=0.3 * rng.normal(size=n))
    st = ad.grpnet(ad.matrix.snp_calldata(cd, imp), mk(), **kw)
    ref = ad.grpnet(oracle.snp_calldata(cd, imp), mk(), **kw)
    assert st.error == "" and ref.error == "" and len(st.lmdas) == len(ref.lmdas) == 14
    assert st.counters["n_panel_blocks"] > 20 and max(st.screen_sizes) > 256
    assert np.abs(st.betas.toarray() - ref.betas.toarray()).max() < 1e-7
    assert np.abs(st.intercepts - ref.intercepts).max() < 1e-7
    assert np.array_equal(st.betas.toarray() != 0, ref.betas.toarray() != 0)
    # the screen sets agree up to where the pivot rule puts its knee on the last lambda's scores (a handful of near-tied
    # groups: the weighted case lands six apart; every ACTIVE group is screened on both sides)
    a, b = set(st.screen_set.tolist()), set(ref.screen_set.tolist())
    assert len(a ^ b) <= 8 and set(np.flatnonzero(st.betas.toarray()[-1])).issubset(a & b)
