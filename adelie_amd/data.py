"""Synthetic data generators — self-contained restatement of the recipes in reference ``adelie/data.py``:
``dense`` (``:84-219``) and ``snp_unphased`` (``:222-359``), single-response gaussian / binomial only.
Used by the tests and by ``bench.py``."""
import numpy as np

from . import glm as _glm


def _sample_y(glm, eta, beta, rho, snr):
    """Reference ``data.py:13-82`` (gaussian and binomial branches)."""
    n = eta.shape[0]
    signal_scale = np.sqrt(rho * np.sum(beta) ** 2 + (1 - rho) * np.sum(beta ** 2))
    noise_scale = signal_scale / np.sqrt(snr)
    if glm == "gaussian":
        y = eta + noise_scale * np.random.normal(0, 1, eta.shape)
        return _glm.gaussian(y=y.ravel())
    if glm == "binomial":
        eta = eta.ravel()
        mu = 1 / (1 + np.exp(-eta / noise_scale))
        y = np.random.binomial(1, mu).astype(eta.dtype)
        return _glm.binomial(y=y)
    raise NotImplementedError(glm)


def _group_layout(p, G, equal_groups):
    """Group start columns and widths.  Unequal groups: G - 1 distinct cut points drawn from 1..p-1 (one ``choice`` draw)."""
    if equal_groups:
        starts = np.arange(G) * (p // G)
    else:
        cuts = np.random.choice(np.arange(1, p), size=G - 1, replace=False)
        starts = np.sort(np.concatenate([[0], cuts])).astype(int)
    widths = np.diff(np.append(starts, p)).astype(int)
    return starts, widths


def _penalty_factors(widths, p, zero_penalty):
    """sqrt(width) per group, a random ``zero_penalty`` share of them zeroed, rescaled to squared norm p."""
    G = widths.shape[0]
    w = np.sqrt(widths)
    unpenalised = np.random.choice(G, int(zero_penalty * G), replace=False)
    w[unpenalised] = 0
    return w / (np.linalg.norm(w) / np.sqrt(p))


def _equicorrelated_normal(n, p, rho):
    """n x p standard normal columns sharing one factor so that every pair has correlation rho (column-major)."""
    noise = np.random.normal(0, 1, (n, p))
    factor = np.random.normal(0, 1, n)
    return np.asfortranarray(np.sqrt(rho) * factor[:, None] + np.sqrt(1 - rho) * noise)


def dense(n: int, p: int, G: int, *, K: int = 1, glm: str = "gaussian", equal_groups: bool = False, rho: float = 0,
          sparsity: float = 0.95, zero_penalty: float = 0, snr: float = 1, seed: int = 0):
    """Dense Gaussian design with ``G`` groups: the recipe of reference ``adelie.data.dense`` (``data.py:84-219``).

    The legacy global stream is consumed in the reference's order — group cuts, unpenalised groups, the design, its shared
    factor, the coefficients, the zeroed coefficients, the response noise — so that a seed gives the reference's data
    (the notebook replays of ``tests/test_reference_known_answers.py`` depend on it)."""
    if not (n >= 1 and p >= 1 and G >= 1 and snr > 0 and seed >= 0):
        raise ValueError("n, p, G >= 1, snr > 0 and seed >= 0 are required")
    if K != 1:
        raise NotImplementedError("single-response families only")
    np.random.seed(seed)
    groups, group_sizes = _group_layout(p, G, equal_groups)
    penalty = _penalty_factors(group_sizes, p, zero_penalty)
    X = _equicorrelated_normal(n, p, rho)

    coef = np.random.normal(0, 1, (p, K))
    in_model = np.ones(p, dtype=bool)
    in_model[np.random.choice(p, int(sparsity * p), replace=False)] = False
    support = np.flatnonzero(in_model)
    signal = coef[support]
    response = _sample_y(glm, X[:, support] @ signal, signal, rho, snr)
    return {"X": X, "glm": response, "groups": groups, "group_sizes": group_sizes, "penalty": penalty}


def snp_unphased(n: int, p: int, G: int = None, *, glm: str = "gaussian", sparsity: float = 0.95,
                 one_ratio: float = 0.25, two_ratio: float = 0.05, missing_ratio: float = 0.1, snr: float = 1,
                 seed: int = 0):
    """Unphased SNP calldata (reference ``adelie.data.snp_unphased``, ``data.py:222-359``): int8 entries in
    {0,1,2} with -9 marking missing calls; the response is generated from the mean-imputed matrix."""
    assert n >= 1 and p >= 1 and seed >= 0
    G = p if G is None else G
    np.random.seed(seed)
    nnz_ratio = one_ratio + two_ratio
    X = np.zeros((n, p), dtype=np.int8)
    nnz = int(nnz_ratio * n * p)
    idx = np.random.choice(n * p, nnz, replace=False)
    n_two = int(two_ratio / max(nnz_ratio, 1e-300) * nnz)
    flat = X.ravel()
    flat[idx[:n_two]] = 2
    flat[idx[n_two:]] = 1
    miss = np.random.choice(n * p, int(missing_ratio * n * p), replace=False)
    flat[miss] = -9
    X = np.asfortranarray(flat.reshape(n, p))

    groups = np.concatenate([[0], np.random.choice(np.arange(1, p), size=G - 1, replace=False)]) if G < p \
        else np.arange(p)
    groups = np.sort(groups).astype(int)
    group_sizes = np.concatenate([groups, [p]], dtype=int)
    group_sizes = group_sizes[1:] - group_sizes[:-1]

    valid = X >= 0
    impute = np.where(valid, X, 0).sum(axis=0) / np.maximum(valid.sum(axis=0), 1)
    Xd = np.where(valid, X, impute[None]).astype(np.float64)
    beta = np.random.normal(0, 1, p)
    beta[np.random.choice(p, int(sparsity * p), replace=False)] = 0
    eta = Xd @ beta
    glm_o = _sample_y(glm, eta[:, None] if glm == "gaussian" else eta, beta[beta != 0], 0, snr)
    return {"X": X, "glm": glm_o, "groups": groups, "group_sizes": group_sizes, "impute": impute}
