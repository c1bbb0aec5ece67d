"""``cv_grpnet`` — mirrors ``adelie.cv.cv_grpnet`` (reference ``adelie/cv.py:130-325``), with the fold loop
sharded over the GPUs of one node.

Folds differ only by their weight vector (validation rows get weight 0, reference ``cv.py:248-252``), so every
rank keeps a full replica of X in its own HBM and solves the folds ``k`` with ``k % world_size == rank``.
Nothing inside the path solver communicates; the only collective is one ``all_gather`` of the ``(L,)`` loss
rows at the end (RCCL over xGMI when the process group backend is ``nccl``, ``gloo`` in CPU tests).
"""
from dataclasses import dataclass
import logging
import os
import threading

import numpy as np
import scipy.sparse

from . import matrix
from .diagnostic import coefficient, coefficients, predict
from .solver import grpnet

logger = logging.getLogger("adelie_amd")

# concurrent folds share their sweeps while at least one cv_grpnet call with folds in flight is running in this process
_batch_lock = threading.Lock()
_batch_users = 0


def _sweep_batch(backend, on):
    """Reference-counted switch of the library-wide "sweep_batch" config: two cv_grpnet calls running in one process must not
    turn it off under each other."""
    global _batch_users
    with _batch_lock:
        if on:
            if _batch_users == 0 and backend.fn("set_config")(b"sweep_batch", 1.0) != 0:
                return False  # nothing was counted: the caller must not pair this with a disable
            _batch_users += 1
            return True
        _batch_users = max(_batch_users - 1, 0)
        if _batch_users == 0 and backend.fn("set_config")(b"sweep_batch", 0.0) != 0:
            logger.warning("adelie_amd: could not switch the shared CV sweeps off again (set_config failed).")
            return False
    return True


# families whose per-lambda CV losses adelie_hip_design_multi_path_losses computes (glm_kind it expects)
_MULTI_KINDS = {"multigaussian": 0, "multinomial": 3}


@dataclass
class CVGrpnetResult:
    """Result of K-fold CV group elastic net (reference ``cv.py:25-45``)."""

    lmdas: np.ndarray
    losses: np.ndarray
    avg_losses: np.ndarray
    best_idx: int
    # not in the reference: counters / device timers / wall time of the solves behind each fold THIS rank ran (bench.py)
    fold_stats: list = None

    def fit(self, X, glm, *, lmda_path_size: int = 100, **grpnet_params):
        """Fits the full data down to the best CV lambda (reference ``cv.py:95-127``)."""
        logger_level = logger.level
        logger.setLevel(logging.ERROR)
        state = grpnet(X=X, glm=glm, lmda_path_size=0, progress_bar=False)
        logger.setLevel(logger_level)
        lmda_star = self.lmdas[self.best_idx]
        full_lmdas = state.lmda_max * np.logspace(
            0, np.log10(lmda_star / state.lmda_max), lmda_path_size
        )
        return grpnet(X=X, glm=glm, lmda_path=full_lmdas, early_exit=False, **grpnet_params)


def fold_ranges(n: int, n_folds: int):
    """Contiguous fold blocks with the remainder spread over the first folds (reference ``cv.py:222-245``)."""
    fold_size = n // n_folds
    remaining = n % n_folds
    out = []
    for fold in range(n_folds):
        begin = (fold_size + 1) * min(fold, remaining) + max(fold - remaining, 0) * fold_size
        out.append((begin, begin + fold_size + (fold < remaining)))
    return out


def _fold_loss(X, glm, fold_idx, full_lmdas, *, n_threads, early_exit, min_ratio, lmda_path_size, grpnet_params, stats=None,
               phase=None, carry=None):
    """Body of the reference's fold loop (``cv.py:247-314``).

    ``phase``: the loop in two halves around the path solve, for folds whose paths are solved together by ONE library call
    (``state.solve_many``): ``"prepare"`` returns the fold's carry -- the reweighted family, the bootstrap state and the
    UNSOLVED path state --, ``"evaluate"`` takes the carry with the solved state in it and returns the fold's loss row."""
    if phase == "evaluate":   # (the solved state is read out of its result handle here, on the fold's own thread)
        weights, weights_sum, glm_c, state0, unsolved, ctx = carry
        state = unsolved._finish_solve(ctx)
    else:
        weights = glm.weights.copy()
        weights[fold_idx] = 0
        weights_sum = np.sum(weights)
        weights /= weights_sum
        glm_c = glm.reweight(weights)

        if phase == "prepare":
            # ONE solve per fold: it computes the fold's lmda_max and joins the fold's own grid above full_lmdas[0] to the
            # full-data grid itself (adelie_hip_grpnet_args::lmda_aug_ratios, ABI 10) -- the numbers the two calls below produce,
            # without the first call's three sweeps over X and a state's worth of host work.  The argument struct is
            # marshalled here, on the fold's own thread.
            state = grpnet(X=X, glm=glm_c, ddev_tol=0, n_threads=n_threads, early_exit=early_exit, lmda_path=full_lmdas,
                           _lmda_aug=(np.logspace(0, np.log10(min_ratio), lmda_path_size), float(full_lmdas[0])),
                           _prepare_only=True, **grpnet_params)
            return [weights, weights_sum, glm_c, None, state, state._begin_solve(False, None)]
        state0 = state = grpnet(X=X, glm=glm_c, n_threads=n_threads, lmda_path_size=0, progress_bar=False)
        curr_lmdas = state.lmda_max * np.logspace(0, np.log10(min_ratio), lmda_path_size)
        curr_lmdas = curr_lmdas[curr_lmdas > full_lmdas[0]]
        aug_lmdas = np.sort(np.concatenate([full_lmdas, curr_lmdas]))[::-1]

        state = grpnet(X=X, glm=glm_c, ddev_tol=0, n_threads=n_threads, early_exit=early_exit, lmda_path=aug_lmdas,
                       **grpnet_params)

    weights_sum_val = np.sum(glm.weights[fold_idx])
    betas, intercepts, lmdas = state.betas, state.intercepts, state.lmdas
    if len(lmdas) >= 2 and np.asarray(intercepts).ndim == 1:
        # every full-data lambda at once (diagnostic.coefficients: the numbers of one coefficient() call per lambda)
        full_betas, full_intercepts = coefficients(lmdas_new=full_lmdas, betas=betas, intercepts=intercepts, lmdas=lmdas)
    else:
        beta_ints = [coefficient(lmda=lmda, betas=betas, intercepts=intercepts, lmdas=lmdas) for lmda in full_lmdas]
        full_betas = scipy.sparse.vstack([x[0] for x in beta_ints]).tocsr()
        full_intercepts = np.array([x[1] for x in beta_ints])
    if (hasattr(X, "glm_path_losses") and X._backend.has("design_glm_path_losses") and hasattr(glm, "core_kind")
            and not getattr(glm, "is_multi", False)):
        # predictions and both losses per lambda on the device; only 2 L scalars come back
        full_data_losses, train = X.glm_path_losses(glm.core_kind, full_betas, full_intercepts, state._offsets, glm.y,
                                                    glm.weights, glm_c.weights)
        train_losses = weights_sum * train
    elif (getattr(glm, "is_multi", False) and hasattr(X, "multi_path_losses") and not isinstance(X, matrix._MultiView)
          and X._backend.has("design_multi_path_losses") and getattr(glm, "name", None) in _MULTI_KINDS):
        # multi-response fits: the (L, n, K) predictions stay on the device as well
        full_data_losses, train = X.multi_path_losses(_MULTI_KINDS[glm.name], glm.y.shape[1], full_betas,
                                                      full_intercepts, state._offsets, glm.y, glm.weights, glm_c.weights)
        train_losses = weights_sum * train
    else:
        etas = predict(X=X, betas=full_betas, intercepts=full_intercepts, offsets=state._offsets, n_threads=n_threads)
        full_data_losses = np.array([glm.loss(eta) for eta in etas])
        train_losses = weights_sum * np.array([glm_c.loss(eta) for eta in etas])
    if stats is not None and state0 is None:  # (one solve per fold, see "prepare")
        stats.append({"counters": dict(state.counters), "timers": dict(state.timers), "total_time": state.total_time})
    elif stats is not None:  # both solves of the fold: the lmda_max bootstrap and the path
        stats.append({
            "counters": {k: state0.counters[k] + v for k, v in state.counters.items()},
            "timers": {k: state0.timers[k] + v for k, v in state.timers.items()},
            "total_time": state0.total_time + state.total_time,
        })
    return (full_data_losses - train_losses) / weights_sum_val if weights_sum_val > 0 else np.zeros(len(full_lmdas))


def cv_grpnet(X, glm, *, n_threads: int = 1, early_exit: bool = False, min_ratio: float = 1e-1,
              lmda_path_size: int = 100, n_folds: int = 5, seed: int = None, process_group=None, n_concurrent: int = None,
              _share=None, **grpnet_params):
    """Cross-validated group elastic net (reference ``adelie.cv.cv_grpnet``; same arguments and defaults).

    ``process_group``: optional ``torch.distributed`` group (or ``True`` for the default group).  When given,
    fold ``k`` is solved by rank ``k % world_size`` and the per-fold loss rows are all-gathered; every rank
    returns the same ``CVGrpnetResult``.  Every rank must pass the same data and seed.

    ``n_concurrent``: folds solved at the same time on this rank's GPU, each from its own host thread on its own HIP stream
    (an alias handle of the resident design).  One path keeps most of the chip idle while its sequential block solves run,
    so two or three folds interleave well (default 3 for device designs, 1 otherwise); single-response folds on a dense design
    also share their full-gradient sweeps — those that reach a sweep within a short window are answered by one pass over X — so
    there the default is 8.  The result does not depend on it beyond the summation order of the shared sweeps (last bits).
    """
    rank, world = 0, 1
    dist = group = None
    if process_group is not None:
        import torch.distributed as dist  # noqa: F811
        group = None if process_group is True else process_group
        rank, world = dist.get_rank(group), dist.get_world_size(group)

    if isinstance(X, np.ndarray) and world > 1 and dist.get_backend(group) == "nccl":
        # one replica per rank, on the rank's own GPU (the device torch.distributed was initialised with)
        import torch
        X = matrix.dense(X, method="naive", n_threads=n_threads, device=torch.cuda.current_device())
    X = matrix.as_design(X, n_threads=n_threads)
    n = X.rows()

    if seed is not None:
        np.random.seed(seed)
    order = np.random.choice(n, n, replace=False)
    if world > 1:
        # the folds must be the same partition on every rank: rank 0's permutation is the one used (an unseeded call would
        # otherwise draw a different one per process and the gathered loss table would mix folds of different partitions)
        box = [order]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        order = np.asarray(box[0])

    logger_level = logger.level
    logger.setLevel(logging.ERROR)
    try:
        state = grpnet(X=X, glm=glm, n_threads=n_threads, lmda_path_size=0, progress_bar=False)
        full_lmdas = state.lmda_max * np.logspace(0, np.log10(min_ratio), lmda_path_size)

        cv_losses = np.zeros((n_folds, full_lmdas.shape[0]))
        fold_stats = [] if hasattr(state, "counters") else None
        ranges = fold_ranges(n, n_folds)
        my_folds = [fold for fold in range(n_folds) if fold % world == rank]
        if _share is not None:  # (measurement aid, scripts/cv_phases.py: the share of rank r of w, solved here without a gather)
            my_folds = [fold for fold in range(n_folds) if fold % int(_share[1]) == int(_share[0])]
        can_alias = hasattr(X, "alias") and hasattr(X, "_backend") and X._backend.has("design_alias")
        # dense Gaussian folds share their sweeps (SweepBatcher), so all of them may as well be in flight; otherwise three
        # interleave well and more only contend
        shares = getattr(X, "_kind", None) == "dense" and not getattr(glm, "is_multi", False)
        nc = n_concurrent if n_concurrent is not None else ((8 if shares else 3) if can_alias else 1)
        nc = max(1, min(int(nc), len(my_folds))) if can_alias else 1
        cons = grpnet_params.get("constraints")
        if cons is not None and any(c is not None for c in cons):
            nc = 1  # the constraint objects carry their multipliers between calls: folds sharing them run one at a time

        def one(Xa, fold):
            b, e = ranges[fold]
            return _fold_loss(Xa, glm, order[b:e], full_lmdas, n_threads=n_threads, early_exit=early_exit,
                              min_ratio=min_ratio, lmda_path_size=lmda_path_size, grpnet_params=grpnet_params,
                              stats=fold_stats)

        if nc <= 1:
            for fold in my_folds:
                cv_losses[fold] = one(X, fold)
        else:
            import queue
            from concurrent.futures import ThreadPoolExecutor

            handles = queue.Queue()
            for h in [X] + [X.alias() for _ in range(nc - 1)]:
                handles.put(h)

            def run(fold):
                Xa = handles.get()
                try:
                    return fold, one(Xa, fold)
                finally:
                    handles.put(Xa)

            # folds in flight share their full-gradient sweeps: the ones that reach a sweep within a short window are answered
            # by one pass over X (solver.hip::SweepBatcher)
            batch = X._backend.has("set_config") and os.environ.get("ADELIE_HIP_SWEEP_BATCH", "1") != "0"
            if batch and not _sweep_batch(X._backend, True):
                logger.warning("adelie_amd: shared CV sweeps could not be enabled; the folds keep separate sweeps.")
                batch = False
            # several folds in flight would interleave their progress bars: off unless asked for
            grpnet_params.setdefault("progress_bar", False)
            # The folds' path solves go out together through ONE library call (adelie_hip_grpnet_solve_many, ABI 10: one host thread
            # per solve below the ABI -- the fold loop a C++ caller of the library would write) when the design is a plain
            # resident one and nothing needs Python between a fold's two solves; the bootstrap solves, the preambles and the
            # evaluation at the full-data grid stay on this side's threads.  ADELIE_HIP_CV_SOLVE_MANY=0: the round-5 loop (every
            # fold's whole body on a Python thread).
            many = (X._backend.has("grpnet_solve_many") and os.environ.get("ADELIE_HIP_CV_SOLVE_MANY", "1") != "0"
                    and getattr(X, "_kind", None) in ("dense", "snp") and not isinstance(X, (matrix._StdView, matrix._MultiView))
                    and not getattr(glm, "is_multi", False) and grpnet_params.get("exit_cond") is None
                    and grpnet_params.get("warm_start") is None and not grpnet_params.get("check_state", False))
            try:
                with ThreadPoolExecutor(max_workers=nc) as pool:
                    if not many:
                        for fold, row in pool.map(run, my_folds):
                            cv_losses[fold] = row
                    else:
                        from .state import run_many
                        for i0 in range(0, len(my_folds), nc):   # (at most `nc` folds in flight: one handle each)
                            batch_folds = my_folds[i0:i0 + nc]
                            hs = [handles.get() for _ in batch_folds]
                            try:
                                def prep(a):
                                    b, e = ranges[a[1]]
                                    return _fold_loss(a[0], glm, order[b:e], full_lmdas, n_threads=n_threads, early_exit=early_exit,
                                                      min_ratio=min_ratio, lmda_path_size=lmda_path_size,
                                                      grpnet_params=grpnet_params, phase="prepare")
                                carries = list(pool.map(prep, zip(hs, batch_folds)))
                                def evalf(k):   # (called from solve k's own thread the moment it returns: the others still run)
                                    b, e = ranges[batch_folds[k]]
                                    cv_losses[batch_folds[k]] = _fold_loss(
                                        hs[k], glm, order[b:e], full_lmdas, n_threads=n_threads, early_exit=early_exit,
                                        min_ratio=min_ratio, lmda_path_size=lmda_path_size, grpnet_params=grpnet_params,
                                        stats=fold_stats, phase="evaluate", carry=carries[k])

                                run_many([c[4] for c in carries], [c[5] for c in carries], on_done=evalf)
                            finally:
                                for h in hs:
                                    handles.put(h)
            finally:
                if batch:
                    _sweep_batch(X._backend, False)
    finally:
        logger.setLevel(logger_level)

    if world > 1:
        cv_losses = _gather_fold_rows(dist, None if process_group is True else process_group, cv_losses, n_folds,
                                      rank, world)

    avg_losses = np.mean(cv_losses, axis=0)
    best_idx = int(np.argmin(avg_losses))
    return CVGrpnetResult(lmdas=full_lmdas, losses=cv_losses, avg_losses=avg_losses, best_idx=best_idx,
                          fold_stats=fold_stats)


def _gather_fold_rows(dist, group, local, n_folds, rank, world):
    """One all_gather of the (n_folds, L) loss table; rows a rank does not own are zero, so the gathered
    tables are merged by ownership (fold k lives on rank k % world)."""
    import torch

    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    merged = np.empty_like(local)
    for k in range(n_folds):
        merged[k] = outs[k % world][k].cpu().numpy()
    return merged
