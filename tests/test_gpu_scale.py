"""Size-independent properties at (a tenth of) BASELINE.json's headline size: too big for the oracle to be the checker
in a unit test, so the path is certified from first principles on the device design itself — KKT of sampled
solutions, the state invariants (resid / grad / rsq), monotone deviance, sorted lambdas."""
import numpy as np
import pytest

import adelie_amd as ad

pytestmark = pytest.mark.gpu


# the last two are BASELINE.json configs 2 and 3 at full size (8 GB design generated on the device)
@pytest.mark.parametrize("n,p,gs,alpha", [(20000, 4000, 1, 1.0), (20000, 2000, 10, 0.5),
                                          (100000, 10000, 1, 1.0), (100000, 10000, 10, 0.5)])
def test_full_size_properties(hip, n, p, gs, alpha):
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    Xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
    X = Xt.t()
    rng = np.random.RandomState(0)
    beta = rng.normal(size=p) * (rng.uniform(size=p) < 0.05)
    y = (X @ torch.from_numpy(beta).cuda()).cpu().numpy() + np.sqrt(beta @ beta) * rng.normal(size=n)
    Xd = ad.matrix.dense(X)
    groups = np.arange(0, p, gs)
    st = ad.grpnet(Xd, ad.glm.gaussian(y), groups=groups, alpha=alpha, early_exit=False)
    assert st.error == "" and len(st.lmdas) == 100
    assert np.all(np.diff(st.lmdas) < 0) and np.all(np.diff(st.devs) >= -1e-9)
    w = np.full(n, 1 / n)
    pen = np.sqrt(np.full(len(groups), gs, dtype=float))
    yc = y - y.mean()
    for l in [5, 50, 99]:
        b = st.betas[l].toarray().ravel()
        r = yc - (Xd @ b) - (st.intercepts[l] - y.mean())
        assert abs(np.sum(w * r)) < 1e-10
        grad = Xd.T @ (w * r)
        lm = st.lmdas[l]
        gn = np.linalg.norm(grad.reshape(-1, gs), axis=1)
        bn = np.linalg.norm(b.reshape(-1, gs), axis=1)
        zero = bn == 0
        # stationarity holds to the CD stopping rule max_k A_kk dbeta_k^2 < tol*y_var (pin_base:100-122,
        # gaussian_naive.hpp:312): each coordinate's gradient is within ~sqrt(tol*y_var*A_kk) of its target,
        # plus the drift from the other coordinates' last moves; allow 20x that (and 1 % of it on the zero groups, whose
        # gradient was inside the bound when visited and only drifts with the later moves of the pass)
        slack = 20 * np.sqrt(1e-7 * st.y_var * gs)
        assert np.all(gn[zero] <= lm * alpha * pen[zero] * (1 + 1e-6) + 0.01 * slack)
        G_, B_ = grad.reshape(-1, gs)[~zero], b.reshape(-1, gs)[~zero]
        target = lm * pen[~zero, None] * (alpha * B_ / bn[~zero, None] + (1 - alpha) * B_)
        assert np.abs(G_ - target).max() < slack
    # state invariants at the end of the path (adelie/state.py:1563-1674)
    b = st.betas[-1].toarray().ravel()
    r = yc - (Xd @ b)
    assert np.abs(st.resid - r).max() < 1e-8
    assert np.abs(st.grad - (Xd.T @ (w * r) - np.sum(w * r) * st.X_means)).max() < 1e-9
    assert abs(st.rsq - (np.sum(w * yc ** 2) - np.sum(w * (r - np.sum(w * r)) ** 2))) < 1e-7


@pytest.mark.parametrize("n,p", [(20000, 3000), (100000, 10000)])
def test_binomial_snp_properties(hip, n, p):
    """Config 4 shape (binomial lasso on a 2-bit SNP design; the second case is a fifth of BASELINE's rows and columns, the
    size the panel engine's IRLS path was tuned on): the path is certified from the KKT conditions of the binomial
    deviance on the device design itself, plus monotone deviance and the intercept's stationarity."""
    import torch

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    cdt = torch.empty((p, n), dtype=torch.int8, device=dev)
    for j0 in range(0, p, 2048):
        j1 = min(p, j0 + 2048)
        u = torch.rand((j1 - j0, n), generator=g, device=dev)
        blk = torch.zeros_like(u, dtype=torch.int8)
        blk[u < 0.25] = 1
        blk[(u >= 0.25) & (u < 0.30)] = 2
        blk[u >= 0.92] = -9
        cdt[j0:j1] = blk
    cd = cdt.t()
    Xd = ad.matrix.snp_calldata(cd)
    rng = np.random.RandomState(3)
    beta = rng.normal(size=p) * (rng.uniform(size=p) < 200.0 / p)
    eta = Xd @ beta
    eta = (eta - eta.mean()) / eta.std()
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    L = 40
    st = ad.grpnet(Xd, ad.glm.binomial(y), early_exit=False, lmda_path_size=L, min_ratio=0.05, tol=1e-9, irls_tol=1e-9)
    assert st.error == "" and len(st.lmdas) == L
    assert st.counters["n_panel_blocks"] > 0
    assert np.all(np.diff(st.lmdas) < 0) and np.all(np.diff(st.devs) >= -1e-9)
    w = np.full(n, 1 / n)
    for l in [3, L // 2, L - 1]:
        b = st.betas[l].toarray().ravel()
        e = (Xd @ b) + st.intercepts[l]
        mu = 1 / (1 + np.exp(-e))
        resid = w * (y - mu)
        assert abs(resid.sum()) < 1e-8                      # intercept stationarity
        grad = Xd.T @ resid
        lm = st.lmdas[l]
        zero = b == 0
        assert np.all(np.abs(grad[zero]) <= lm * (1 + 1e-5) + 1e-9)
        # active coordinates: |grad_j - lm*sign(b_j)| small relative to lm (IRLS + CD stopping rules)
        assert np.abs(grad[~zero] - lm * np.sign(b[~zero])).max() < 2e-3 * lm + 1e-7
    assert st.active_set_size > 128
