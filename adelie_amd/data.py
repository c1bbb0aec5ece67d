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


def dense(n: int, p: int, G: int, *, K: int = 1, glm: str = "gaussian", equal_groups: bool = False, rho: float = 0,
          sparsity: float = 0.95, zero_penalty: float = 0, snr: float = 1, seed: int = 0):
    """Dense Gaussian design with ``G`` groups (reference ``adelie.data.dense``, ``data.py:84-219``)."""
    assert n >= 1 and p >= 1 and G >= 1 and snr > 0 and seed >= 0 and K == 1
    np.random.seed(seed)
    if equal_groups:
        groups = (p // G) * np.arange(G)
    else:
        groups = np.concatenate([[0], np.random.choice(np.arange(1, p), size=G - 1, replace=False)])
        groups = np.sort(groups).astype(int)
    group_sizes = np.concatenate([groups, [p]], dtype=int)
    group_sizes = group_sizes[1:] - group_sizes[:-1]
    penalty = np.sqrt(group_sizes)
    penalty[np.random.choice(G, int(zero_penalty * G), replace=False)] = 0
    penalty /= np.linalg.norm(penalty) / np.sqrt(p)

    X = np.random.normal(0, 1, (n, p))
    Z = np.random.normal(0, 1, n)
    X = np.sqrt(rho) * Z[:, None] + np.sqrt(1 - rho) * X
    X = np.asfortranarray(X)

    beta = np.random.normal(0, 1, (p, K))
    beta_zero_indices = np.random.choice(p, int(sparsity * p), replace=False)
    beta_nnz_indices = np.array(sorted(set(np.arange(p)) - set(beta_zero_indices)), dtype=int)
    X_sub = X[:, beta_nnz_indices]
    beta_sub = beta[beta_nnz_indices]
    eta = X_sub @ beta_sub
    glm_o = _sample_y(glm, eta, beta_sub, rho, snr)
    return {"X": X, "glm": glm_o, "groups": groups, "group_sizes": group_sizes, "penalty": penalty}


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
