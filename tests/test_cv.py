"""cv_grpnet: fold partition rule, the induced-validation-loss identity, and the fold sharding over ranks
(world_size 2, gloo, CPU) — reference adelie/cv.py:130-325 (untested upstream; pinned here as SURVEY.md 8a/C1 says)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import adelie_amd as ad
from adelie_amd.cv import fold_ranges
from util import make_gaussian

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fold_ranges_match_reference_rule():
    for n, k in [(10, 3), (100, 8), (7, 7), (23, 5)]:
        fr = fold_ranges(n, k)
        sizes = [e - b for b, e in fr]
        assert sum(sizes) == n and fr[0][0] == 0 and fr[-1][1] == n
        assert all(fr[i][1] == fr[i + 1][0] for i in range(k - 1))
        assert sizes == [n // k + (i < n % k) for i in range(k)]  # cv.py:241-245


def test_cv_loss_identity(oracle):
    """cv_losses[k, i] == sum_{i in fold k} w_i l_i(eta_i) / sum_{i in fold k} w_i with beta from the training fit."""
    d = make_gaussian(120, 20, seed=3)
    X, y = d["X"], d["y"]
    Xo = oracle.dense(X)
    glm = ad.glm.gaussian(y)
    res = ad.cv_grpnet(Xo, glm, n_folds=4, seed=7, lmda_path_size=12, tol=1e-12)
    assert res.losses.shape == (4, 12) and res.lmdas.shape == (12,)
    assert res.best_idx == int(np.argmin(res.avg_losses)) and np.allclose(res.avg_losses, res.losses.mean(0))
    np.random.seed(7)
    order = np.random.choice(120, 120, replace=False)
    k = 2
    b, e = fold_ranges(120, 4)[k]
    w = glm.weights.copy()
    w[order[b:e]] = 0
    w /= w.sum()
    st = ad.grpnet(Xo, glm.reweight(w), lmda_path=res.lmdas, early_exit=False, tol=1e-12)
    val = order[b:e]
    for i in [0, 5, 11]:
        eta = X @ st.betas[i].toarray().ravel() + st.intercepts[i]
        li = 0.5 * eta[val] ** 2 - y[val] * eta[val]
        assert np.isclose(res.losses[k, i], li.mean(), rtol=1e-5, atol=1e-8)


def test_cv_fold_sharding_world_size_2_gloo(oracle, tmp_path):
    """Two CPU ranks over gloo: folds k%2==rank solved per rank, one all_gather, identical result on both ranks and
    equal to the single-process result."""
    script = tmp_path / "run.py"
    script.write_text(f"""
import os, sys, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import torch.distributed as dist
import adelie_amd as ad
from oracle import oracle
from util import make_gaussian
dist.init_process_group(backend="gloo")
d = make_gaussian(90, 15, seed=1)
res = ad.cv_grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), n_folds=5, seed=3, lmda_path_size=8, tol=1e-12,
                   process_group=True)
np.save({str(tmp_path)!r} + f"/loss{{dist.get_rank()}}.npy", res.losses)
np.random.seed(50 + dist.get_rank())   # unseeded call, different streams: rank 0's permutation is broadcast
res = ad.cv_grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), n_folds=5, lmda_path_size=8, tol=1e-12,
                   process_group=True)
np.save({str(tmp_path)!r} + f"/uloss{{dist.get_rank()}}.npy", res.losses)
dist.destroy_process_group()
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)], env=env, timeout=600)
    l0, l1 = np.load(tmp_path / "loss0.npy"), np.load(tmp_path / "loss1.npy")
    assert np.array_equal(l0, l1)
    d = make_gaussian(90, 15, seed=1)
    single = ad.cv_grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), n_folds=5, seed=3, lmda_path_size=8, tol=1e-12)
    assert np.allclose(l0, single.losses, rtol=1e-12, atol=0)
    u0, u1 = np.load(tmp_path / "uloss0.npy"), np.load(tmp_path / "uloss1.npy")
    assert np.array_equal(u0, u1)
    np.random.seed(50)
    usingle = ad.cv_grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), n_folds=5, lmda_path_size=8, tol=1e-12)
    assert np.allclose(u0, usingle.losses, rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_cv_on_device_matches_oracle(hip, oracle):
    d = make_gaussian(400, 60, seed=2)
    a = ad.cv_grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), n_folds=4, seed=1, lmda_path_size=15, tol=1e-10)
    b = ad.cv_grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), n_folds=4, seed=1, lmda_path_size=15, tol=1e-10)
    assert np.allclose(a.lmdas, b.lmdas, rtol=1e-10)
    assert np.abs(a.losses - b.losses).max() < 1e-7
    assert a.best_idx == b.best_idx


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "snp"])
def test_cv_concurrent_folds_on_alias_handles(hip, kind, monkeypatch):
    """Folds solved concurrently from host threads on alias handles of the resident design (own HIP stream each) give
    bit-identical loss tables; an alias is the same matrix."""
    rng = np.random.RandomState(2)
    n, p = 900, 120
    if kind == "dense":
        X = ad.matrix.dense(np.asfortranarray(rng.normal(size=(n, p))))
    else:
        cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.6, 0.2, 0.1, 0.1])
        X = ad.matrix.snp_calldata(cd)
    Xa = X.alias()
    v = rng.normal(size=n)
    np.testing.assert_array_equal(X.T @ v, Xa.T @ v)
    y = (X @ (rng.normal(size=p) * (rng.uniform(size=p) < 0.1))) + rng.normal(size=n)
    glm = ad.glm.gaussian(y)
    kw = dict(n_folds=6, seed=5, lmda_path_size=20, tol=1e-10)
    a = ad.cv_grpnet(X, glm, n_concurrent=1, **kw)
    monkeypatch.setenv("ADELIE_HIP_SWEEP_BATCH", "0")
    b = ad.cv_grpnet(X, glm, n_concurrent=3, **kw)
    c = ad.cv_grpnet(X, glm, **kw)  # default concurrency
    np.testing.assert_array_equal(a.losses, b.losses)
    np.testing.assert_array_equal(a.losses, c.losses)
    assert a.best_idx == b.best_idx
    # with the folds' sweeps batched into shared passes over X (the default; dense designs only) the gradients are summed in
    # another order: same table to rounding, and the same whoever happens to share a batch
    monkeypatch.delenv("ADELIE_HIP_SWEEP_BATCH")
    d = ad.cv_grpnet(X, glm, n_concurrent=3, **kw)
    e = ad.cv_grpnet(X, glm, n_concurrent=4, **kw)
    np.testing.assert_allclose(d.losses, a.losses, rtol=1e-7, atol=1e-9)
    np.testing.assert_array_equal(d.losses, e.losses)
    assert d.best_idx == a.best_idx
    del Xa


@pytest.mark.gpu
def test_cv_fold_sharding_two_ranks_on_device(hip, tmp_path):
    """The multi-GPU path of cv_grpnet with DEVICE designs: two ranks (both on cuda:0 — the test box has one GPU — rendezvous
    over gloo; the driver's multi-GPU runs use RCCL) shard the folds, draw their permutation WITHOUT a seed (rank 0's is
    broadcast), all_gather the loss rows, and agree with each other and with the single-process result for that permutation."""
    script = tmp_path / "run.py"
    script.write_text(f"""
import os, sys, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import torch, torch.distributed as dist
import adelie_amd as ad
from util import make_gaussian
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
d = make_gaussian(600, 80, seed=1)
X = ad.matrix.dense(d["X"])
np.random.seed(100 + rank)          # different streams per rank: the fold order must still be rank 0's
res = ad.cv_grpnet(X, ad.glm.gaussian(d["y"]), n_folds=5, lmda_path_size=12, tol=1e-12, process_group=True)
np.save({str(tmp_path)!r} + f"/loss{{rank}}.npy", res.losses)
assert len(res.fold_stats) == len(range(rank, 5, 2))
dist.destroy_process_group()
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29618", str(script)], env=env, timeout=900)
    l0, l1 = np.load(tmp_path / "loss0.npy"), np.load(tmp_path / "loss1.npy")
    assert np.array_equal(l0, l1)
    d = make_gaussian(600, 80, seed=1)
    np.random.seed(100)                 # rank 0's stream
    single = ad.cv_grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), n_folds=5, lmda_path_size=12, tol=1e-12)
    assert np.allclose(l0, single.losses, rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_solve_many_entry_point(hip, monkeypatch):
    """``adelie_hip_grpnet_solve_many`` (ABI 10): independent states solved by ONE library call, one host thread per solve below
    the ABI.  Same numbers as the solves one after another (bit for bit: separate sweeps), a handle may not appear twice, and
    ``cv_grpnet`` gives the same table through it as through the round-5 loop of Python threads."""
    from adelie_amd.state import solve_many

    rng = np.random.RandomState(8)
    n, p = 1500, 200
    X = ad.matrix.dense(np.asfortranarray(rng.normal(size=(n, p))))
    hs = [X, X.alias(), X.alias()]
    ys = [X @ (rng.normal(size=p) * (rng.uniform(size=p) < 0.1)) + rng.normal(size=n) for _ in hs]
    kw = dict(lmda_path_size=25, min_ratio=0.05, early_exit=False, tol=1e-10, progress_bar=False)
    fams = [ad.glm.gaussian(ys[0]), ad.glm.binomial((ys[1] > 0).astype(float)), ad.glm.gaussian(ys[2])]
    extra = [dict(), dict(), dict(groups=np.arange(0, p, 5), alpha=0.6)]
    one_by_one = [ad.grpnet(h, f, **kw, **e) for h, f, e in zip(hs, fams, extra)]
    states = [ad.grpnet(h, f, _prepare_only=True, **kw, **e) for h, f, e in zip(hs, fams, extra)]
    together = solve_many(states)
    for a, b in zip(one_by_one, together):
        assert b.error == "" and len(b.lmdas) == 25
        np.testing.assert_array_equal(a.betas.toarray(), b.betas.toarray())
        np.testing.assert_array_equal(a.intercepts, b.intercepts)
        np.testing.assert_array_equal(a.devs, b.devs)
    # the same handle twice: refused (concurrent solves need a stream and scratch each)
    twice = [ad.grpnet(X, fams[0], _prepare_only=True, **kw), ad.grpnet(X, fams[2], _prepare_only=True, **kw)]
    with pytest.raises(RuntimeError, match="appears twice"):
        solve_many(twice)
    # cv_grpnet: the path solves through the entry point against the loop of Python threads
    glm = ad.glm.gaussian(ys[0])
    kwc = dict(n_folds=5, seed=3, lmda_path_size=20, tol=1e-10)
    monkeypatch.setenv("ADELIE_HIP_SWEEP_BATCH", "0")
    new = ad.cv_grpnet(X, glm, n_concurrent=3, **kwc)
    monkeypatch.setenv("ADELIE_HIP_CV_SOLVE_MANY", "0")
    old = ad.cv_grpnet(X, glm, n_concurrent=3, **kwc)
    np.testing.assert_array_equal(new.losses, old.losses)
    assert len(new.fold_stats) == len(old.fold_stats) == 5
