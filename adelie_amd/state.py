"""State objects — mirrors ``adelie.state`` for ``gaussian_naive`` and ``glm_naive``.

Reference: ``adelie/state.py:118-176`` (``base.create_from_core`` / ``base.solve``), ``:1007-1045``
(sentinel rendering), ``:1421-1675`` (``gaussian_naive_base.check``), ``:1677-2024`` (``gaussian_naive``),
``:2407-2753`` (``glm_naive``); core constructors ``py_state.cpp:1068-1154`` / ``:1556-1720``.

In the reference a state is a Python object that *inherits* from a pybind C++ state; ``solve()`` passes
the C++ state by value into the solver and re-wraps the solved copy.  Here a state is a plain Python
object that owns the constructor arguments; ``solve()`` marshals them into ``adelie_hip_grpnet_args``
(include/adelie_hip.h), runs the path on the GPU, and returns a *new* state whose attributes are read
back through the result accessors — the caller's state is left untouched, as in the reference.
"""
import logging

import numpy as np
from scipy.sparse import csr_matrix

from . import _abi
from . import constraint as _constraint
from . import glm as _glm
from . import matrix as _matrix

logger = logging.getLogger("adelie_amd")


def _render_inputs(*, groups, lmda_max, lmda_path, lmda_path_size, max_screen_size, max_active_size, dtype):
    """Reference ``state.py:1007-1045`` (``_render_gaussian_inputs``)."""
    if max_screen_size is None:
        max_screen_size = len(groups)
    if max_active_size is None:
        max_active_size = len(groups)
    max_screen_size = int(np.minimum(max_screen_size, len(groups)))
    max_active_size = int(np.minimum(max_active_size, len(groups)))
    lmda_path_size = lmda_path_size if lmda_path is None else len(lmda_path)
    setup_lmda_max = lmda_max is None
    setup_lmda_path = lmda_path is None
    if setup_lmda_max:
        lmda_max = -1
    if setup_lmda_path:
        lmda_path = np.empty(0, dtype=dtype)
    return max_screen_size, max_active_size, int(lmda_path_size), setup_lmda_max, setup_lmda_path, lmda_max, lmda_path


_SCREEN_RULES = {"strong": _abi.SCREEN_STRONG, "pivot": _abi.SCREEN_PIVOT}

_RESULT_VALUE_VECS = ["intercepts", "devs", "lmdas", "lmda_path", "screen_beta", "grad", "abs_grad", "resid",
                      "screen_X_means", "screen_vars",
                      "benchmark_screen", "benchmark_fit_screen", "benchmark_fit_active", "benchmark_kkt",
                      "benchmark_invariance"]
_RESULT_INDEX_VECS = ["screen_set", "screen_begins", "screen_is_active", "active_set", "n_valid_solutions",
                      "active_sizes", "screen_sizes"]
_COUNTERS = ["n_basil_iters", "n_sweeps", "n_cd_visits_screen", "n_cd_visits_active", "n_updates", "n_irls_iters",
             "n_new_screen_cols", "n_cd_passes_screen", "n_cd_passes_active", "n_gram_col_reads",
             "n_resid_col_reads", "n_panel_blocks", "n_panel_grams", "n_panel_cols", "n_irls_screen_cols",
             "n_speculated", "n_spec_rollbacks", "n_sweeps_shared", "n_update_cols", "n_device_screens",
             "n_host_screens", "n_host_cons_visits", "n_dev_cons_visits"]
_TIMERS = ["gram_flops", "t_sweep_ms", "t_gram_ms", "t_cd_ms", "t_axpy_ms", "n_sweep_launches", "n_gram_launches",
           "t_host_screen_ms", "t_panel_step_ms", "n_panel_step_launches", "t_host_screen_wait_ms"]


class _ProgressBar:
    """The reference's progress bar (a tqdm-style line, ``util/tqdm.hpp``) with its suffix ``[dev:xx.x%]``
    (``solver_base.hpp:225-239``), rendered on stderr once per saved lambda (at most every 50 ms)."""

    def __init__(self, total, width=10):
        import sys
        import time

        self._time = time
        self._out = sys.stderr
        self.total, self.width = total, width
        self.t0 = self.t_last = time.perf_counter()
        self.n, self.dev, self.label = 0, 0.0, "dev"

    @staticmethod
    def _clock(sec):
        sec = int(sec)
        return "%02d:%02d:%02d" % (sec // 3600, (sec // 60) % 60, sec % 60)

    def _render(self):
        el = self._time.perf_counter() - self.t0
        frac = self.n / self.total
        rate = self.n / el if el > 0 else 0.0
        left = (self.total - self.n) / rate if rate > 0 else 0.0
        fill = int(round(frac * self.width))
        self._out.write("\r%3d%%|%s%s| %d/%d [%s<%s, %.2fit/s] [%s:%.1f%%]" % (
            int(100 * frac), "\u2588" * fill, " " * (self.width - fill), self.n, self.total, self._clock(el),
            self._clock(left), rate, self.label, 100 * self.dev))
        self._out.flush()

    def update(self, n, dev):
        self.n, self.dev, self.stale = n, dev, True
        now = self._time.perf_counter()
        if now - self.t_last >= 0.05 or n >= self.total:
            self.t_last = now
            self._render()
            self.stale = False

    def close(self):
        if getattr(self, "stale", True):
            self._render()
        self._out.write("\n")
        self._out.flush()


class live_state:
    """What ``exit_cond`` receives: the state *while it is being solved* (the reference hands the callback its live C++
    state, ``py_state.cpp:62-91``).  Attributes are read from the solver on access through the result accessors of
    include/adelie_hip.h on the ``live`` handle; constructor arguments (``alpha``, ``penalty``, ``groups`` ...) come from the
    state ``solve()`` was called on.  Only valid inside the callback."""

    _DEVICE_RESIDENT = ("grad", "resid", "eta", "screen_beta", "screen_X_means", "screen_vars")
    _SCALARS = {"lmda_max": None, "lmda": None, "rsq": None, "resid_sum": None, "beta0": None, "loss_null": None,
                "loss_full": None, "active_set_size": int}

    def __init__(self, owner, backend, handle):
        d = self.__dict__
        d["_owner"], d["_backend"], d["_handle"], d["_synced"] = owner, backend, handle, False

    def __setattr__(self, name, value):
        raise AttributeError("the live state is read-only")

    @property
    def n_solutions(self):
        return int(self._backend.fn("result_size")(self._handle, _abi.V["lmdas"]))

    def _path(self):
        b, r, o = self._backend, self._handle, self._owner
        indptr = b.result_vec(r, _abi.I["betas_indptr"], index=True)
        indices = b.result_vec(r, _abi.I["betas_indices"], index=True)
        values = b.result_vec(r, _abi.V["betas_values"]).astype(o.dtype)
        betas = csr_matrix((values, indices, indptr), shape=(len(indptr) - 1, o._X.cols()))
        return o._tidy_path(betas, b.result_vec(r, _abi.V["intercepts"]).astype(o.dtype))

    def __getattr__(self, name):
        b, r, o = self._backend, self._handle, self._owner
        if name == "betas":
            return self._path()[0]
        if name == "intercepts":
            return self._path()[1]
        if name in self._SCALARS:
            v = b.fn("result_scalar")(r, _abi.S[name])
            return int(v) if self._SCALARS[name] is int else o.dtype(v)
        if name in self._DEVICE_RESIDENT:
            if not self._synced:
                b.check(b.fn("result_sync")(r))
                self.__dict__["_synced"] = True
            return b.result_vec(r, _abi.V[name]).astype(o.dtype)
        if name in _abi.V:
            v = b.result_vec(r, _abi.V[name])
            return v if name.startswith("benchmark") else v.astype(o.dtype)
        if name in _abi.I:
            v = b.result_vec(r, _abi.I[name], index=True)
            return v.astype(bool) if name == "screen_is_active" else v
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(o, name)


def run_many(states, ctxs, on_done=None):
    """The library call of :func:`solve_many` alone: ``ctxs[k] = states[k]._begin_solve(...)`` (prepared by the caller, e.g. on
    the folds' own threads); every context ends up holding its result handle (``None`` where that solve failed) for
    ``states[k]._finish_solve(ctxs[k])``.  ``on_done(k)``, when given, is called from solve k's own thread as soon as that
    solve has returned -- its handle is in place -- while the others still run (``cv_grpnet`` evaluates the fold there); what it
    raises is re-raised here.  Raises after all solves have returned if one failed."""
    K = len(states)
    if K == 0:
        return
    backend = states[0]._X._backend
    if not backend.has("grpnet_solve_many") or any(s._solve_entry != "grpnet_solve" for s in states):
        raise RuntimeError("solve_many needs naive-method states on a backend that exports adelie_hip_grpnet_solve_many.")
    X = (_abi.C.c_void_p * K)(*[s._X._handle for s in states])
    A = (_abi.C.POINTER(_abi.GrpnetArgs) * K)(*[_abi.C.pointer(c["args"]) for c in ctxs])
    H = (_abi.C.c_void_p * K)()
    pending = {}

    def _take(k):
        c = ctxs[k]
        if c["bar"] is not None:
            c["bar"].close()
        c["handle"] = _abi.C.c_void_p(H[k]) if H[k] else None
        c["taken"] = True

    def _done(k, rc_k, user):
        try:
            _take(k)
            if rc_k == 0 and on_done is not None:
                on_done(int(k))
        except BaseException as e:  # noqa: BLE001 - ctypes would swallow it
            pending.setdefault("exc", e)

    cb = _abi.DONE_FN(_done)
    rc = backend.fn("grpnet_solve_many")(X, A, K, H, cb, None)
    for k, c in enumerate(ctxs):
        if not c.get("taken"):
            _take(k)
    if rc != 0 or "exc" in pending:
        for c in ctxs:  # nobody will finish what is left: free it
            if c.get("handle") is not None and not c.get("finished"):
                backend.fn("result_destroy")(c["handle"])
                c["handle"] = None
            c["keep"] = None
        if "exc" in pending:
            raise pending["exc"]
        backend.check(rc)


def solve_many(states, progress_bar: bool = False):
    """Solves several independent states concurrently through ONE library call (``adelie_hip_grpnet_solve_many``, ABI 10: one
    host thread per solve below the ABI) -- the folds of ``cv_grpnet``: every state on its own handle of the resident design
    (an alias per fold on one device, or replicas on several).  Returns the solved states in order; a failed solve raises
    after all of them have returned.  Needs a backend that exports the entry point (the HIP library)."""
    states = list(states)
    ctxs = [s._begin_solve(progress_bar and len(states) == 1, None) for s in states]
    run_many(states, ctxs)
    return [s._finish_solve(c) for s, c in zip(states, ctxs)]


class base:
    """Common machinery of the two naive states."""

    # names of the constructor arguments that are carried over unchanged into a solved state
    _static = ()

    def _marshal(self):  # pragma: no cover - abstract
        raise NotImplementedError

    def solve(self, progress_bar: bool = True, exit_cond=None):
        """Runs the path solver; returns a new solved state (reference ``state.py:157-176``; this state is left untouched).

        ``exit_cond`` is called after every saved lambda with the live state (reference ``py_state.cpp:62-91``,
        ``solver_base.hpp:581,679``): a :class:`live_state` view whose attributes (``lmdas``, ``devs``, ``betas``,
        ``intercepts``, ``lmda``, ``screen_set``, ``active_set_size``, ... and on request the device-resident ``grad`` /
        ``resid`` / ``screen_beta``) are read from the solver at that moment.  An exception raised by it stops the path and is
        re-raised here.  ``progress_bar`` renders the reference's bar with its ``[dev:xx.x%]`` suffix
        (``solver_base.hpp:225-239``) on stderr.
        """
        ctx = self._begin_solve(progress_bar, exit_cond)
        backend, args, handle = ctx["backend"], ctx["args"], ctx["handle"]
        try:
            backend.check(backend.fn(self._solve_entry)(self._X._handle, _abi.C.byref(args), handle))
        finally:
            if ctx["bar"] is not None:
                ctx["bar"].close()
        return self._finish_solve(ctx)

    def _begin_solve(self, progress_bar, exit_cond):
        """Everything of a solve up to the library call: the marshalled argument struct (kept alive in the returned context)
        and the poll callback that drives the progress bar / ``exit_cond``."""
        backend = self._X._backend
        self._glm_cb_pending = {}  # one per solve, shared by every callback factory of _marshal (_callback_errors)
        args, keep = self._marshal()
        pending = {}
        total = int(self.lmda_path_size) if self.setup_lmda_path else len(self.lmda_path)
        bar = _ProgressBar(total) if progress_bar and total > 0 else None

        def _poll(user, final, n_solutions, live):
            try:
                if not final:
                    return 0
                view = live_state(self, backend, live) if (bar is not None or exit_cond is not None) else None
                if bar is not None:
                    bar.label, shown = self._progress(view.devs)
                    bar.update(int(n_solutions), shown)
                if exit_cond is not None:
                    return 1 if exit_cond(view) else 0
            except BaseException as e:  # noqa: BLE001 - ctypes would swallow it: stop the path, re-raise after the solve
                pending["exc"] = e
                return 1
            return 0

        cb = _abi.POLL_FN(_poll)
        args.poll = cb
        args.poll_user = None
        return dict(backend=backend, args=args, keep=keep, cb=cb, pending=pending, bar=bar, handle=_abi.C.c_void_p())

    def _finish_solve(self, ctx):
        """The solved state from the result handle the library filled in (and the exceptions callbacks parked meanwhile)."""
        backend, pending, handle = ctx["backend"], ctx["pending"], ctx["handle"]
        ctx["keep"] = None
        ctx["finished"] = True  # (the handle is destroyed below, whatever happens)
        if "exc" not in pending and "exc" in getattr(self, "_glm_cb_pending", {}):
            pending["exc"] = self._glm_cb_pending.pop("exc")
        if "exc" in pending:
            backend.fn("result_destroy")(handle)
            raise pending["exc"]
        try:
            out = self._from_result(backend, handle)
        finally:
            backend.fn("result_destroy")(handle)
        if out.error != "":
            if out.error.startswith("adelie_core solver: "):
                logger.error(RuntimeError(out.error))
            else:
                logger.warning(RuntimeError(out.error))
        return out

    def _tidy_path(self, betas, intercepts):
        """Hook of the multi-response states: splits the per-class intercepts off the coefficient rows."""
        return betas, intercepts


    def _constraint_list(self):
        """The (G,) list of constraint objects, or None when there is none (reference ``state.py:24-45`` render_constraints:
        a shorter list is left-padded with None)."""
        cons = getattr(self, "constraints", None)
        if cons is None or not any(c is not None for c in cons):
            return None
        cons = list(cons)
        G = len(self.groups)
        return [None] * (G - len(cons)) + cons if len(cons) < G else cons

    _solve_entry = "grpnet_solve"

    @staticmethod
    def _progress(devs):
        """What the progress bar shows next to the count: the fraction of deviance explained (``solver_base.hpp:225-239``)."""
        return "dev", (float(devs[-1]) if len(devs) else 0.0)

    def _from_result(self, backend, r):
        new = object.__new__(type(self))
        new.__dict__.update({k: v for k, v in self.__dict__.items()})
        dtype = self.dtype
        for name in _RESULT_VALUE_VECS:
            v = backend.result_vec(r, _abi.V[name])
            setattr(new, name, v.astype(dtype) if not name.startswith("benchmark") else v)
        for name in _RESULT_INDEX_VECS:
            v = backend.result_vec(r, _abi.I[name], index=True)
            setattr(new, name, v)
        new.screen_is_active = new.screen_is_active.astype(bool)
        new.n_valid_solutions = new.n_valid_solutions.astype(np.int32)
        new.active_sizes = new.active_sizes.astype(np.int32)
        new.screen_sizes = new.screen_sizes.astype(np.int32)
        indptr = backend.result_vec(r, _abi.I["betas_indptr"], index=True)
        indices = backend.result_vec(r, _abi.I["betas_indices"], index=True)
        values = backend.result_vec(r, _abi.V["betas_values"]).astype(dtype)
        p = self._X.cols()
        new.betas = csr_matrix((values, indices, indptr), shape=(len(indptr) - 1, p))
        new.betas, new.intercepts = self._tidy_path(new.betas, new.intercepts)
        cons = self._constraint_list()
        if cons is None:
            new.duals = csr_matrix((len(indptr) - 1, 0), dtype=dtype)
        else:  # state_base.hpp `duals`: row l = the non-zero multipliers at lmdas[l], column dual_groups[i] for group i
            dp = backend.result_vec(r, _abi.I["duals_indptr"], index=True)
            di = backend.result_vec(r, _abi.I["duals_indices"], index=True)
            dv = backend.result_vec(r, _abi.V["duals_values"]).astype(dtype)
            new.dual_groups = _constraint.render_dual_groups(cons)
            n_duals = int(sum(c.dual_size for c in cons if c is not None))
            new.duals = csr_matrix((dv, di, dp), shape=(len(dp) - 1, n_duals))
            mu = backend.result_vec(r, _abi.V["constraint_mu"])
            for c, m in zip(cons, mu):  # the objects are left holding the multipliers of the last fit, as the reference's are
                if c is not None and c._abi()[0] != _constraint.KIND_HOST:  # (host objects were the live objects all along)
                    c._mu[0] = m
            vmu = backend.result_vec(r, _abi.V["constraint_vmu"])
            if len(vmu):
                # box / one-sided objects on several coefficients that the DEVICE solved: the same courtesy.  Which groups those
                # were is the library's report (ABI 9), not a rule restated here.  Such an object is not live during the solve
                # (an exit_cond callback sees the multipliers it entered with); it holds the last fit's on return.
                for i in backend.result_vec(r, _abi.I["constraint_dev_groups"], index=True):
                    c = cons[int(i)]
                    g0 = int(self.groups[int(i)])
                    c._mu[...] = vmu[g0:g0 + c.primal_size]
        sc = lambda nm: backend.fn("result_scalar")(r, _abi.S[nm])
        new.lmda_max = dtype(sc("lmda_max"))
        new.lmda = dtype(sc("lmda"))
        new.active_set_size = int(sc("active_set_size"))
        new.total_time = sc("total_time")
        new.counters = {nm: int(v) if v == v else 0 for nm, v in ((nm, sc(nm)) for nm in _COUNTERS)}
        new.timers = {nm: (v if v == v else 0.0) for nm, v in ((nm, sc(nm)) for nm in _TIMERS)}
        err = backend.fn("result_error")(r)
        new.error = err.decode() if err else ""
        # screen_transforms: list of (q,q) F-ordered blocks in screen order
        flat = backend.result_vec(r, _abi.V["screen_transforms"]).astype(dtype)
        tr, off = [], 0
        qs = np.asarray(self.group_sizes)[new.screen_set]
        if len(flat) == int(np.sum(np.square(qs))):
            if len(qs) and np.all(qs == qs[0]):
                # groups of one size (lasso: (1,1) blocks): all blocks at once, each a column-major (q,q) view, no Python loop
                q = int(qs[0])
                tr = list(flat.reshape(len(qs), q, q).transpose(0, 2, 1))
            else:
                for q in qs.tolist():
                    tr.append(flat[off:off + q * q].reshape(q, q, order="F"))
                    off += q * q
        new.screen_transforms = tr
        self._from_result_extra(new, backend, r, sc)
        return new

    def _from_result_extra(self, new, backend, r, sc):
        pass

    def _common_args(self, a, keep):
        dtype = self.dtype

        def val(x):
            arr = np.ascontiguousarray(x, dtype=dtype)
            keep.append(arr)
            return arr.ctypes.data

        def idx(x):
            arr = np.ascontiguousarray(x, dtype=np.int64)
            keep.append(arr)
            return arr.ctypes.data

        a.G = len(self.groups)
        a.groups = idx(self.groups)
        a.group_sizes = idx(self.group_sizes)
        a.alpha = float(self.alpha)
        a.penalty = val(self.penalty)
        aug = getattr(self, "_lmda_aug", None)     # (ABI 10; set by cv_grpnet: the fold's own grid joins lmda_path inside the solve)
        if aug is not None:
            ratios = np.ascontiguousarray(aug[0], dtype=np.float64)
            keep.append(ratios)
            a.lmda_aug_ratios = ratios.ctypes.data_as(_abi.C.POINTER(_abi.C.c_double))
            a.n_lmda_aug = len(ratios)
            a.lmda_aug_min = float(aug[1])
        pl2 = getattr(self, "_penalty_l2", None)   # (ABI 8; set by solver.grpnet for an elastic net on a standardized view)
        if pl2 is not None:
            if len(pl2) != len(self.groups):
                raise RuntimeError("adelie_amd: _penalty_l2 must be (G,) where groups is (G,).")
            a.penalty_l2 = val(pl2)
        a.resid = val(self.resid)
        a.grad = val(self.grad)
        lp = np.ascontiguousarray(self.lmda_path, dtype=dtype)
        keep.append(lp)
        a.lmda_path = lp.ctypes.data if lp.size else None
        a.n_lmda_path = lp.size
        a.lmda_max = float(self.lmda_max)
        a.min_ratio = float(self.min_ratio)
        a.lmda_path_size = int(self.lmda_path_size)
        a.max_screen_size = int(self.max_screen_size)
        a.max_active_size = int(self.max_active_size)
        a.pivot_subset_ratio = float(self.pivot_subset_ratio)
        a.pivot_subset_min = int(self.pivot_subset_min)
        a.pivot_slack_ratio = float(self.pivot_slack_ratio)
        if self.screen_rule not in _SCREEN_RULES:
            raise RuntimeError("adelie_core: Invalid screen rule type: " + str(self.screen_rule))
        a.screen_rule = _SCREEN_RULES[self.screen_rule]
        a.early_exit = int(bool(self.early_exit))
        a.max_iters = int(self.max_iters)
        a.tol = float(self.tol)
        a.adev_tol = float(self.adev_tol)
        a.ddev_tol = float(self.ddev_tol)
        a.newton_tol = float(self.newton_tol)
        a.newton_max_iters = int(self.newton_max_iters)
        a.setup_lmda_max = int(self.setup_lmda_max)
        a.setup_lmda_path = int(self.setup_lmda_path)
        a.intercept = int(bool(self.intercept))
        a.n_threads = int(self.n_threads)
        ss = np.ascontiguousarray(self.screen_set, dtype=np.int64)
        keep.append(ss)
        a.screen_set_size = ss.size
        a.screen_set = ss.ctypes.data
        sb = np.ascontiguousarray(self.screen_beta, dtype=dtype)
        keep.append(sb)
        a.screen_beta_size = sb.size
        a.screen_beta = sb.ctypes.data
        sa = np.ascontiguousarray(self.screen_is_active, dtype=np.int8)
        keep.append(sa)
        if sa.size != ss.size:
            raise RuntimeError("adelie_core: screen_is_active must be (s,) where screen_set is (s,).")
        a.screen_is_active = sa.ctypes.data
        a.active_set_size = int(self.active_set_size)
        act = np.ascontiguousarray(self.active_set, dtype=np.int64)
        if act.size != a.G:
            raise RuntimeError("adelie_core: active_set must be (G,) where groups is (G,).")
        keep.append(act)
        a.active_set = act.ctypes.data
        a.lmda = float(self.lmda)
        cons = self._constraint_list()
        if cons is not None:
            # One-coefficient box / one-sided constraints: (kind, a, b) per group + the multipliers held on entry, solved on the
            # device.  Every other object is reached through callbacks (kind 3); for the built-in box / one-sided classes on
            # several coefficients the arguments also carry what the CPU checker needs to run its own restatement.
            G, p = a.G, self._X.cols()
            kind = np.zeros(G, dtype=np.int32)
            ca, cb, mu = np.zeros(G, dtype=dtype), np.zeros(G, dtype=dtype), np.zeros(G, dtype=dtype)
            ndual = np.zeros(G, dtype=np.int64)
            native = np.zeros(G, dtype=np.int32)
            va, vb, vmu = np.zeros(p, dtype=dtype), np.zeros(p, dtype=dtype), np.zeros(p, dtype=dtype)
            cfg = np.zeros((G, 5), dtype=np.float64)
            lin = (_abi.C.c_void_p * G)()
            any_host = any_lin = False
            for i, c in enumerate(cons):
                if c is None:
                    continue
                kind[i], ca[i], cb[i] = c._abi()
                ndual[i] = c.duals()
                if kind[i] == _constraint.KIND_HOST:
                    any_host = True
                    nat = c._native()
                    if nat is not None:
                        g0, q = int(self.groups[i]), int(self.group_sizes[i])
                        native[i] = nat[0]
                        if nat[1] is not None:
                            va[g0:g0 + q], vb[g0:g0 + q] = nat[1], nat[2]
                            vmu[g0:g0 + q] = c._mu   # (ABI 7: box / one-sided objects are solved on the device; what they hold on entry)
                        cfg[i] = nat[3]
                        if nat[0] == _constraint.NATIVE_LINEAR:  # (m, d) objects do not fit per-coefficient arrays
                            desc, arrays = c._linear_descriptor()
                            keep += [desc, arrays]
                            lin[i] = _abi.C.addressof(desc)
                            any_lin = True
                else:
                    mu[i] = c._mu[0]
            keep += [kind, ca, cb, mu, ndual, native, va, vb, cfg, vmu]
            a.constraint_kind = kind.ctypes.data
            a.constraint_a = ca.ctypes.data
            a.constraint_b = cb.ctypes.data
            a.constraint_mu = mu.ctypes.data
            a.constraint_duals = ndual.ctypes.data
            if any_host:
                a.constraint_native = native.ctypes.data
                a.constraint_va = va.ctypes.data
                a.constraint_vb = vb.ctypes.data
                a.constraint_cfg = cfg.ctypes.data
                a.constraint_vmu = vmu.ctypes.data
                if any_lin:
                    keep.append(lin)
                    a.constraint_lin = _abi.C.addressof(lin)
                cbs = self._constraint_callbacks(cons)
                keep.append(cbs)
                a.constraint_cb = _abi.C.pointer(cbs)

    def _callback_errors(self):
        """The one dict per marshalled solve in which every host callback (GLM methods, constraint methods) parks the first
        exception it raised; ``solve()`` re-raises it.  ``_common_args`` runs before ``_glm_callbacks``: both must see the
        SAME dict, so neither creates its own."""
        d = getattr(self, "_glm_cb_pending", None)
        if d is None:
            d = self._glm_cb_pending = {}
        return d

    def _constraint_callbacks(self, cons):
        """``adelie_hip_constraint_callbacks`` over the constraint objects that are not device closed forms — the role of the
        reference's trampoline ``PyConstraintBase``: the solver hands host vectors in double precision, the methods work in
        place.  An exception raised by a method aborts the solve and is re-raised by ``solve()``."""
        pending = self._callback_errors()
        as_arr = np.ctypeslib.as_array
        # the reference's trampoline (py_constraint.cpp:19-36,77-90) hands every solve / solve_zero a uint64 scratch vector
        # of the object's own buffer_size() as the last positional argument: a class written against that API has a
        # mandatory `buffer` parameter
        scratch = {}

        def buf(g):
            if g not in scratch:
                scratch[g] = np.zeros(int(cons[g].buffer_size()), dtype=np.uint64)
            return scratch[g]

        def guarded(f):
            def call(*args):
                try:
                    f(*args)
                    return 0
                except BaseException as e:  # noqa: BLE001
                    pending.setdefault("exc", e)
                    return 1
            return call

        @guarded
        def solve(user, g, d, x, quad, linear, l1, l2, Q):
            xv = as_arr(x, (d,))
            work = xv.copy()
            cons[g].solve(work, as_arr(quad, (d,)).copy(), as_arr(linear, (d,)).copy(), float(l1), float(l2),
                          as_arr(Q, (d * d,)).reshape(d, d, order="F").copy(), buf(g))
            xv[...] = work

        @guarded
        def gradient(user, g, d, x, out):
            res = np.zeros(d)
            cons[g].gradient(as_arr(x, (d,)).copy(), res)
            as_arr(out, (d,))[...] = res

        @guarded
        def solve_zero(user, g, d, v, norm):
            norm[0] = float(cons[g].solve_zero(as_arr(v, (d,)).copy(), buf(g)))

        @guarded
        def dual(user, g, m, mu_out):
            idx, val = np.zeros(m, dtype=int), np.zeros(m)
            nnz = cons[g].duals_nnz()
            cons[g].dual(idx, val)
            dense = np.zeros(m)
            dense[idx[:nnz]] = val[:nnz]
            as_arr(mu_out, (m,))[...] = dense

        return _abi.ConstraintCallbacks(None, _abi.CONS_SOLVE_FN(solve), _abi.CONS_GRADIENT_FN(gradient),
                                        _abi.CONS_SOLVE_ZERO_FN(solve_zero), _abi.CONS_DUAL_FN(dual))

    def _check_constraints(self, G):
        """The constructor checks of the reference on ``constraints`` (``state_base.ipp:39-49`` and the object sizes)."""
        cons = self._constraint_list()
        if cons is None:
            return
        if len(cons) != G:
            raise RuntimeError("adelie_core: constraints must be (G,) where groups is (G,).")
        seen = set()
        for c, q in zip(cons, np.asarray(self.group_sizes)):
            if c is None:
                continue
            if not isinstance(c, _constraint.ConstraintBase):
                raise RuntimeError("adelie_core: constraints must be instances of adelie_amd.constraint.ConstraintBase "
                                   "(box / lower / upper / one_sided, or a subclass that provides solve / gradient / "
                                   "solve_zero).")
            if id(c) in seen:
                raise RuntimeError("adelie_core: constraints must contain distinct objects or nullptr.")
            seen.add(id(c))
            if c.primal_size != q:
                raise RuntimeError("adelie_core: constraint of a group must have the group's size.")

    def _check_shapes(self):
        G = len(self.groups)
        if len(self.group_sizes) != G:
            raise RuntimeError("adelie_core: group_sizes must be (G,) where groups is (G,).")
        if len(self.penalty) != G:
            raise RuntimeError("adelie_core: penalty must be (G,) where groups is (G,).")
        self._check_constraints(G)
        n, p = self._X.rows(), self._X.cols()
        if len(self.resid) != n:
            raise RuntimeError("adelie_core: resid must be (n,) where X is (n, p).")
        if len(self.grad) != p:
            raise RuntimeError("adelie_core: grad must be (p,) where X is (n, p).")


class gaussian_naive_base(base):
    """Gaussian naive state (reference ``state.py:1421-1675``)."""

    def _marshal(self):
        keep = []
        a = _abi.GrpnetArgs()
        self._common_args(a, keep)
        dtype = self.dtype
        w = np.ascontiguousarray(self.weights, dtype=dtype)
        xm = np.ascontiguousarray(self.X_means, dtype=dtype)
        keep += [w, xm]
        a.glm_kind = _abi.GLM_GAUSSIAN
        a.weights = w.ctypes.data
        a.X_means = xm.ctypes.data
        a.y_mean = float(self.y_mean)
        a.y_var = float(self.y_var)
        a.resid_sum = float(self.resid_sum)
        a.rsq = float(self.rsq)
        return a, keep

    def _from_result_extra(self, new, backend, r, sc):
        dtype = self.dtype
        new.rsq = dtype(sc("rsq"))
        new.resid_sum = dtype(sc("resid_sum"))
        new.loss_null = dtype(sc("loss_null"))
        new.loss_full = dtype(sc("loss_full"))

    def check(self, method: str = None, logger=logger):
        """Invariance checks of reference ``state.py:1563-1674`` on the host copy of the state
        (X accessed through the matrix handle).  ``method="assert"`` asserts each check."""
        X = self._X
        dtype = self.dtype
        n, p = X.rows(), X.cols()
        w = np.asarray(self.weights, dtype=dtype)
        ok = True

        def _c(cond, msg):
            nonlocal ok
            ok = ok and bool(cond)
            logger.log(logging.INFO if cond else logging.ERROR, f"check {msg}: {'pass' if cond else 'FAIL'}")
            if method == "assert":
                assert cond, msg

        _c(np.allclose(np.sum(w), 1), "weights sum to 1")
        beta = np.zeros(p, dtype=dtype)
        for ss, g in enumerate(self.screen_set):
            b = self.screen_begins[ss] if hasattr(self, "screen_begins") else int(np.sum(self.group_sizes[self.screen_set[:ss]]))
            q = self.group_sizes[g]
            beta[self.groups[g]:self.groups[g] + q] = self.screen_beta[b:b + q]
        Xb = X @ beta
        yc = self._yc
        resid = yc - Xb
        _c(np.allclose(self.resid, resid, atol=1e-6), "resid = y_c - X beta")
        _c(np.allclose(self.resid_sum, np.sum(w * resid), atol=1e-6), "resid_sum")
        grad = X.T @ (w * resid)
        if self.intercept:
            grad = grad - np.sum(w * resid) * np.asarray(self.X_means)
        _c(np.allclose(self.grad, grad, atol=1e-6), "grad = X_c^T W resid")
        return ok


class glm_naive_base(base):
    """GLM naive state (reference ``state.py:2407-2753``)."""

    def _marshal(self):
        keep = []
        a = _abi.GrpnetArgs()
        self._common_args(a, keep)
        dtype = self.dtype
        g = self._glm
        y = np.ascontiguousarray(g.y, dtype=dtype)
        w = np.ascontiguousarray(g.weights, dtype=dtype)
        off = np.ascontiguousarray(self._offsets, dtype=dtype)
        eta = np.ascontiguousarray(self.eta, dtype=dtype)
        keep += [y, w, off, eta]
        kind = getattr(g, "core_kind", _abi.GLM_CALLBACK)
        a.glm_kind = int(kind) if kind != _abi.GLM_GAUSSIAN else _abi.GLM_GAUSSIAN_IRLS
        if kind == _abi.GLM_CALLBACK:
            cbs = self._glm_callbacks(g, len(y))
            keep.append(cbs)
            a.glm_cb = _abi.C.pointer(cbs)
        a.glm_y = y.ctypes.data
        a.glm_weights = w.ctypes.data
        a.offsets = off.ctypes.data
        a.eta = eta.ctypes.data
        a.beta0 = float(self.beta0)
        a.loss_null = float(self.loss_null)
        a.loss_full = float(self.loss_full)
        a.irls_max_iters = int(self.irls_max_iters)
        a.irls_tol = float(self.irls_tol)
        a.setup_loss_null = int(self.setup_loss_null)
        return a, keep

    def _glm_callbacks(self, glm, n):
        """``adelie_hip_glm_callbacks`` over a Python subclass of ``glm.GlmBase64/32`` — the role of the reference's trampoline
        ``PyGlmBase`` (``py_glm.cpp:8-92``): the solver hands host n-vectors, the methods write their outputs in place.  An
        exception raised by a method aborts the solve and is re-raised by ``solve()``."""
        ctype = _abi.C.c_double if np.dtype(self.dtype) == np.float64 else _abi.C.c_float
        pending = self._callback_errors()

        def vec(address):
            return np.ctypeslib.as_array((ctype * n).from_address(address))

        def guarded(f):
            def call(*args):
                try:
                    f(*args)
                    return 0
                except BaseException as e:  # noqa: BLE001
                    pending.setdefault("exc", e)
                    return 1
            return call

        @guarded
        def gradient(user, eta, grad):
            glm.gradient(vec(eta), vec(grad))

        @guarded
        def hessian(user, eta, grad, hess, inv_hess_grad):
            e, g, h = vec(eta), vec(grad), vec(hess)
            glm.hessian(e, g, h)
            glm.inv_hessian_gradient(e, g, h, vec(inv_hess_grad))

        @guarded
        def loss(user, eta, out):
            out[0] = float(glm.loss(vec(eta)))

        return _abi.GlmCallbacks(None, _abi.GLM_GRADIENT_FN(gradient), _abi.GLM_HESSIAN_FN(hessian),
                                 _abi.GLM_LOSS_FN(loss))

    def _from_result_extra(self, new, backend, r, sc):
        dtype = self.dtype
        new.beta0 = dtype(sc("beta0"))
        new.loss_null = dtype(sc("loss_null"))
        new.loss_full = dtype(sc("loss_full"))
        new.eta = backend.result_vec(r, _abi.V["eta"]).astype(dtype)


def _matrix_of(X, n_threads):
    return _matrix.as_design(X, n_threads=n_threads)


def gaussian_naive(
    *, X, y, X_means, y_mean, y_var, resid, resid_sum, constraints, groups, group_sizes, alpha, penalty, weights,
    offsets, screen_set, screen_beta, screen_is_active, active_set_size, active_set, rsq, lmda, grad,
    lmda_path=None, lmda_max=None, max_iters=int(1e5), tol=1e-7, adev_tol=0.9, ddev_tol=0, newton_tol=1e-12,
    newton_max_iters=1000, n_threads=1, early_exit=True, intercept=True, screen_rule="pivot", min_ratio=1e-2,
    lmda_path_size=100, max_screen_size=None, max_active_size=None, pivot_subset_ratio=0.1, pivot_subset_min=1,
    pivot_slack_ratio=1.25,
):
    """Creates a Gaussian, naive method state object (reference ``adelie.state.gaussian_naive``,
    ``state.py:1677-2024``; argument meaning identical)."""
    X = _matrix_of(X, n_threads)
    dtype = X.dtype
    (max_screen_size, max_active_size, lmda_path_size, setup_lmda_max, setup_lmda_path, lmda_max, lmda_path) = \
        _render_inputs(groups=groups, lmda_max=lmda_max, lmda_path=lmda_path, lmda_path_size=lmda_path_size,
                       max_screen_size=max_screen_size, max_active_size=max_active_size, dtype=dtype)
    s = gaussian_naive_base()
    s.dtype = dtype
    s._X = X
    s.X = X
    s._glm = _glm.gaussian(y=np.asarray(y), weights=np.asarray(weights), dtype=dtype)
    s.weights = s._glm.weights
    s._offsets = np.array(offsets, copy=True, dtype=dtype)
    s.X_means = np.array(X_means, copy=True, dtype=dtype)
    s.y_mean = y_mean
    s.y_var = y_var
    s._yc = (np.asarray(y, dtype=dtype) - s._offsets) - (y_mean if intercept else 0)
    s.resid = np.array(resid, copy=True, dtype=dtype)
    s.resid_sum = resid_sum
    s.constraints = constraints
    s.groups = np.array(groups, copy=True, dtype=int)
    s.group_sizes = np.array(group_sizes, copy=True, dtype=int)
    s.alpha = alpha
    s.penalty = np.array(penalty, copy=True, dtype=dtype)
    s.lmda_path = np.asarray(lmda_path, dtype=dtype)
    s.lmda_max = lmda_max
    s.min_ratio = min_ratio
    s.lmda_path_size = lmda_path_size
    s.max_screen_size = max_screen_size
    s.max_active_size = max_active_size
    s.pivot_subset_ratio = pivot_subset_ratio
    s.pivot_subset_min = pivot_subset_min
    s.pivot_slack_ratio = pivot_slack_ratio
    s.screen_rule = screen_rule
    s.max_iters = max_iters
    s.tol = tol
    s.adev_tol = adev_tol
    s.ddev_tol = ddev_tol
    s.newton_tol = newton_tol
    s.newton_max_iters = newton_max_iters
    s.early_exit = early_exit
    s.setup_lmda_max = setup_lmda_max
    s.setup_lmda_path = setup_lmda_path
    s.intercept = intercept
    s.n_threads = n_threads
    s.screen_set = np.asarray(screen_set, dtype=int)
    s.screen_beta = np.asarray(screen_beta, dtype=dtype)
    s.screen_is_active = np.asarray(screen_is_active, dtype=bool)
    s.active_set_size = active_set_size
    s.active_set = np.asarray(active_set, dtype=int)
    s.rsq = rsq
    s.lmda = lmda
    s.grad = np.asarray(grad, dtype=dtype)
    s.error = ""
    s._check_shapes()
    if len(s.weights) != X.rows():
        raise RuntimeError("adelie_core: weights must be (n,) where X is (n, p).")
    if len(s.X_means) != X.cols():
        raise RuntimeError("adelie_core: X_means must be (p,) where X is (n, p).")
    return s


def _split_class_intercepts(state, betas):
    """The first ``K`` view columns of a multi-response fit are the per-class intercepts: every solution is split into
    ``(L, K)`` intercepts and coefficients over ``X (x) I_K`` (``solver_multigaussian_naive.hpp:31-44``,
    ``py_state.cpp:1352-1366``)."""
    K, L = state.n_classes, betas.shape[0]
    if not state.multi_intercept:
        return betas, np.zeros((L, K), dtype=state.dtype)
    B = betas.tocsc()
    return csr_matrix(B[:, K:].tocsr()), np.asarray(B[:, :K].todense(), dtype=state.dtype).reshape(L, K)


class multigaussian_naive_base(gaussian_naive_base):
    """MultiGaussian naive state (reference ``state.py:2027-2391``): the Gaussian naive solver with the global intercept
    off on the expanded design ``[1 (x) I_K, X (x) I_K]``; the per-response intercepts are its first ``K`` (unpenalised)
    coefficients and are split off every solution (``solver_multigaussian_naive.hpp:31-44``, ``py_state.cpp:1352-1366``)."""


    def _tidy_path(self, betas, intercepts):
        return _split_class_intercepts(self, betas)

    def check(self, method: str = None, logger=logger):
        raise NotImplementedError("adelie_amd: check() of the multi-response state.")


def multigaussian_naive(
    *, X, y, X_means, y_var, resid, resid_sum, constraints, groups, group_sizes, alpha, penalty, weights,
    offsets, screen_set, screen_beta, screen_is_active, active_set_size, active_set, rsq, lmda, grad,
    lmda_path=None, lmda_max=None, max_iters=int(1e5), tol=1e-7, adev_tol=0.9, ddev_tol=0, newton_tol=1e-12,
    newton_max_iters=1000, n_threads=1, early_exit=True, intercept=True, screen_rule="pivot", min_ratio=1e-2,
    lmda_path_size=100, max_screen_size=None, max_active_size=None, pivot_subset_ratio=0.1, pivot_subset_min=1,
    pivot_slack_ratio=1.25,
):
    """Creates a MultiGaussian, naive method state object (reference ``adelie.state.multigaussian_naive``,
    ``state.py:2027-2391``; argument meaning identical: ``X`` is the raw ``(n, p)`` design, ``y`` and ``offsets`` are
    ``(n, K)``, ``groups`` / ``grad`` / ``X_means`` are in the expanded coordinates)."""
    X_raw = _matrix_of(X, n_threads)
    dtype = X_raw.dtype
    y = np.asarray(y, dtype=dtype)
    offsets = np.array(offsets, order="C", copy=True, dtype=dtype)
    n, n_classes = offsets.shape
    # state.py:1100-1125 (_render_multi_inputs)
    X_exp = _matrix.kronecker_eye(X_raw, n_classes, n_threads=n_threads)
    if intercept:
        X_exp = _matrix.concatenate(
            [_matrix.kronecker_eye(np.ones((n, 1), dtype=dtype), n_classes, n_threads=n_threads), X_exp],
            axis=1, n_threads=n_threads)
    (max_screen_size, max_active_size, lmda_path_size, setup_lmda_max, setup_lmda_path, lmda_max, lmda_path) = \
        _render_inputs(groups=groups, lmda_max=lmda_max, lmda_path=lmda_path, lmda_path_size=lmda_path_size,
                       max_screen_size=max_screen_size, max_active_size=max_active_size, dtype=dtype)
    if len(X_means) != X_exp.cols():
        raise RuntimeError("X_means must have the same length as the number of columns of X after reshaping.")
    s = multigaussian_naive_base()
    s.dtype = dtype
    s._glm = _glm.multigaussian(y=y, weights=np.asarray(weights), dtype=dtype)
    s.X = X_raw
    s._X_raw = X_raw
    s._X = X_exp            # what solve() runs on (the reference's `_X_expanded`)
    s._X_expanded = X_exp
    s.n_classes = int(n_classes)
    s.multi_intercept = bool(intercept)
    s.weights = np.repeat(s._glm.weights, n_classes) / n_classes  # state.py:2316
    s._offsets = offsets
    s.X_means = np.array(X_means, copy=True, dtype=dtype)
    # state.py:2333-2337: not the mean of y; only enters loss_null / loss_full
    s.y_mean = np.linalg.norm(np.sum(s._glm.weights[:, None] * (y - offsets), axis=-1) / n_classes)
    s.y_var = y_var
    s.resid = np.array(resid, copy=True, dtype=dtype).ravel()
    s.resid_sum = resid_sum
    s.constraints = constraints
    s.groups = np.array(groups, copy=True, dtype=int)
    s.group_sizes = np.array(group_sizes, copy=True, dtype=int)
    s.alpha = alpha
    s.penalty = np.array(penalty, copy=True, dtype=dtype)
    s.lmda_path = np.asarray(lmda_path, dtype=dtype)
    s.lmda_max = lmda_max
    s.min_ratio = min_ratio
    s.lmda_path_size = lmda_path_size
    s.max_screen_size = max_screen_size
    s.max_active_size = max_active_size
    s.pivot_subset_ratio = pivot_subset_ratio
    s.pivot_subset_min = pivot_subset_min
    s.pivot_slack_ratio = pivot_slack_ratio
    s.screen_rule = screen_rule
    s.max_iters = max_iters
    s.tol = tol
    s.adev_tol = adev_tol
    s.ddev_tol = ddev_tol
    s.newton_tol = newton_tol
    s.newton_max_iters = newton_max_iters
    s.early_exit = early_exit
    s.setup_lmda_max = setup_lmda_max
    s.setup_lmda_path = setup_lmda_path
    s.intercept = False     # the core state runs with the global intercept off (state.py:2364)
    s.n_threads = n_threads
    s.screen_set = np.asarray(screen_set, dtype=int)
    s.screen_beta = np.asarray(screen_beta, dtype=dtype)
    s.screen_is_active = np.asarray(screen_is_active, dtype=bool)
    s.active_set_size = active_set_size
    s.active_set = np.asarray(active_set, dtype=int)
    s.rsq = rsq
    s.lmda = lmda
    s.grad = np.asarray(grad, dtype=dtype)
    s.error = ""
    s._check_shapes()
    return s


class multiglm_naive_base(glm_naive_base):
    """Multi-response GLM naive state (reference ``state.py:2756-3121``): ``StateGlmNaive`` with the global intercept off on
    ``[1 (x) I_K, X (x) I_K]``; null model and tidy step of ``solver_multiglm_naive.hpp:99-246``."""

    def _marshal(self):
        a, keep = glm_naive_base._marshal(self)
        y = np.ascontiguousarray(self._glm.y, dtype=self.dtype)  # (n, K) row-major
        w = np.ascontiguousarray(self._glm.weights, dtype=self.dtype)  # (n,)
        keep += [y, w]
        a.glm_y = y.ctypes.data
        a.glm_weights = w.ctypes.data
        return a, keep


    def _tidy_path(self, betas, intercepts):
        return _split_class_intercepts(self, betas)


def multiglm_naive(
    *, X, glm, constraints, groups, group_sizes, alpha, penalty, offsets, screen_set, screen_beta, screen_is_active,
    active_set_size, active_set, lmda, grad, eta, resid, loss_full, loss_null=None, lmda_path=None,
    lmda_max=None, irls_max_iters=int(1e4), irls_tol=1e-7, max_iters=int(1e5), tol=1e-7, adev_tol=0.9, ddev_tol=0,
    newton_tol=1e-12, newton_max_iters=1000, n_threads=1, early_exit=True, intercept=True, screen_rule="pivot",
    min_ratio=1e-2, lmda_path_size=100, max_screen_size=None, max_active_size=None, pivot_subset_ratio=0.1,
    pivot_subset_min=1, pivot_slack_ratio=1.25,
):
    """Creates a multi-response GLM, naive method state object (reference ``adelie.state.multiglm_naive``,
    ``state.py:2756-3121``).  ``X`` is the raw ``(n, p)`` design; ``offsets`` is ``(n, K)``; ``eta`` / ``resid`` are the
    flattened ``(n, K)`` arrays; ``groups`` / ``grad`` are in the expanded coordinates."""
    X_raw = _matrix_of(X, n_threads)
    dtype = X_raw.dtype
    if getattr(glm, "core_kind", None) != _abi.GLM_MULTINOMIAL:
        raise RuntimeError("adelie_amd: of the multi-response IRLS families only adelie_amd.glm.multinomial runs on device.")
    offsets = np.array(offsets, order="C", copy=True, dtype=dtype)
    n, n_classes = offsets.shape
    X_exp = _matrix.kronecker_eye(X_raw, n_classes, n_threads=n_threads)
    if intercept:
        X_exp = _matrix.concatenate(
            [_matrix.kronecker_eye(np.ones((n, 1), dtype=dtype), n_classes, n_threads=n_threads), X_exp],
            axis=1, n_threads=n_threads)
    s = glm_naive(
        X=X_exp, glm=glm, constraints=constraints, groups=groups, group_sizes=group_sizes, alpha=alpha, penalty=penalty,
        offsets=offsets.ravel(), screen_set=screen_set, screen_beta=screen_beta, screen_is_active=screen_is_active,
        active_set_size=active_set_size, active_set=active_set, beta0=0, lmda=lmda, grad=grad,
        eta=np.asarray(eta, dtype=dtype).ravel(), resid=np.asarray(resid, dtype=dtype).ravel(), loss_full=loss_full,
        loss_null=loss_null, lmda_path=lmda_path, lmda_max=lmda_max, irls_max_iters=irls_max_iters, irls_tol=irls_tol,
        max_iters=max_iters, tol=tol, adev_tol=adev_tol, ddev_tol=ddev_tol, newton_tol=newton_tol,
        newton_max_iters=newton_max_iters, n_threads=n_threads, early_exit=early_exit, intercept=False,
        screen_rule=screen_rule, min_ratio=min_ratio, lmda_path_size=lmda_path_size, max_screen_size=max_screen_size,
        max_active_size=max_active_size, pivot_subset_ratio=pivot_subset_ratio, pivot_subset_min=pivot_subset_min,
        pivot_slack_ratio=pivot_slack_ratio,
    )
    s.__class__ = multiglm_naive_base
    s._offsets = offsets  # (n, K), as the reference keeps it (state.py:3046); the core gets the same memory flattened
    s.offsets = offsets
    s.X = X_raw
    s._X_raw = X_raw
    s._X_expanded = X_exp
    s.n_classes = int(n_classes)
    s.multi_intercept = bool(intercept)
    return s


def glm_naive(
    *, X, glm, constraints, groups, group_sizes, alpha, penalty, offsets, screen_set, screen_beta, screen_is_active,
    active_set_size, active_set, beta0, lmda, grad, eta, resid, loss_full, loss_null=None, lmda_path=None,
    lmda_max=None, irls_max_iters=int(1e4), irls_tol=1e-7, max_iters=int(1e5), tol=1e-7, adev_tol=0.9, ddev_tol=0,
    newton_tol=1e-12, newton_max_iters=1000, n_threads=1, early_exit=True, intercept=True, screen_rule="pivot",
    min_ratio=1e-2, lmda_path_size=100, max_screen_size=None, max_active_size=None, pivot_subset_ratio=0.1,
    pivot_subset_min=1, pivot_slack_ratio=1.25,
):
    """Creates a GLM, naive method state object (reference ``adelie.state.glm_naive``, ``state.py:2407-2753``)."""
    X = _matrix_of(X, n_threads)
    dtype = X.dtype
    if not hasattr(glm, "core_kind"):  # a user-defined family: evaluated through host callbacks (glm_naive_base._glm_callbacks)
        if not isinstance(glm, (_glm.GlmBase64, _glm.GlmBase32)):
            raise RuntimeError("glm must be an adelie_amd.glm family or a subclass of adelie_amd.glm.GlmBase64 / GlmBase32.")
        if getattr(glm, "is_multi", False):
            raise RuntimeError("adelie_amd: of the multi-response families glm.multigaussian and glm.multinomial run on device; "
                               "user-defined multi-response GLMs are not supported.")
        if (np.float64 if isinstance(glm, _glm.GlmBase64) else np.float32) != np.dtype(dtype).type:
            raise RuntimeError("glm and X must have the same underlying value type (GlmBase64 with a float64 design).")
    (max_screen_size, max_active_size, lmda_path_size, setup_lmda_max, setup_lmda_path, lmda_max, lmda_path) = \
        _render_inputs(groups=groups, lmda_max=lmda_max, lmda_path=lmda_path, lmda_path_size=lmda_path_size,
                       max_screen_size=max_screen_size, max_active_size=max_active_size, dtype=dtype)
    setup_loss_null = loss_null is None  # state.py:2401-2403
    if setup_loss_null:
        loss_null = np.inf
    s = glm_naive_base()
    s.dtype = dtype
    s._X = X
    s.X = X
    s._glm = glm
    s._offsets = np.array(offsets, copy=True, dtype=dtype)
    s.offsets = s._offsets
    s.constraints = constraints
    s.groups = np.array(groups, copy=True, dtype=int)
    s.group_sizes = np.array(group_sizes, copy=True, dtype=int)
    s.alpha = alpha
    s.penalty = np.array(penalty, copy=True, dtype=dtype)
    s.lmda_path = np.asarray(lmda_path, dtype=dtype)
    s.lmda_max = lmda_max
    s.min_ratio = min_ratio
    s.lmda_path_size = lmda_path_size
    s.max_screen_size = max_screen_size
    s.max_active_size = max_active_size
    s.pivot_subset_ratio = pivot_subset_ratio
    s.pivot_subset_min = pivot_subset_min
    s.pivot_slack_ratio = pivot_slack_ratio
    s.screen_rule = screen_rule
    s.irls_max_iters = irls_max_iters
    s.irls_tol = irls_tol
    s.max_iters = max_iters
    s.tol = tol
    s.adev_tol = adev_tol
    s.ddev_tol = ddev_tol
    s.newton_tol = newton_tol
    s.newton_max_iters = newton_max_iters
    s.early_exit = early_exit
    s.setup_lmda_max = setup_lmda_max
    s.setup_lmda_path = setup_lmda_path
    s.setup_loss_null = setup_loss_null
    s.intercept = intercept
    s.n_threads = n_threads
    s.screen_set = np.asarray(screen_set, dtype=int)
    s.screen_beta = np.asarray(screen_beta, dtype=dtype)
    s.screen_is_active = np.asarray(screen_is_active, dtype=bool)
    s.active_set_size = active_set_size
    s.active_set = np.asarray(active_set, dtype=int)
    s.beta0 = beta0
    s.lmda = lmda
    s.grad = np.asarray(grad, dtype=dtype)
    s.eta = np.asarray(eta, dtype=dtype)
    s.resid = np.asarray(resid, dtype=dtype)
    s.loss_null = loss_null
    s.loss_full = loss_full
    s.error = ""
    s._check_shapes()
    n = X.rows()
    if len(s._offsets) != n:
        raise RuntimeError("adelie_core: offsets must be (n,) where X is (n, p).")
    if len(s.eta) != n:
        raise RuntimeError("adelie_core: eta must be (n,) where X is (n, p).")
    if irls_tol <= 0:
        raise RuntimeError("adelie_core: irls_tol must be > 0.")
    return s


class gaussian_cov_base(base):
    """Gaussian, covariance method state (reference ``state.py:1128-1418``; core ``state_gaussian_cov.hpp:40-145``)."""

    _solve_entry = "gaussian_cov_solve"

    @staticmethod
    def _progress(devs):
        """Relative change of the deviance explained (``solver_gaussian_cov.hpp:165-181``): the saturated loss is unknown here."""
        if len(devs) < 2 or devs[-1] == 0:
            return "rdev", 0.0
        return "rdev", float((devs[-1] - devs[-2]) / devs[-1])

    def _marshal(self):
        keep = []
        a = _abi.GrpnetArgs()
        self.resid = np.empty(0, dtype=self.dtype)  # no residual in the covariance method; the common marshaller wants the name
        self._common_args(a, keep)
        del self.resid
        v = np.ascontiguousarray(self.v, dtype=self.dtype)
        keep.append(v)
        a.glm_kind = _abi.GLM_GAUSSIAN
        a.cov_v = v.ctypes.data
        a.rsq = float(self.rsq)
        a.rdev_tol = float(self.rdev_tol)
        return a, keep

    def _from_result_extra(self, new, backend, r, sc):
        new.rsq = self.dtype(sc("rsq"))
        del new.resid  # (the accessor returns an empty vector for this state)

    def check(self, method: str = None, logger=logger):
        """Invariants of reference ``state.py:1260-1327`` that do not need the solver: ``grad = v - A beta`` on the screen set."""
        A, p = self._X, self._X.cols()
        beta = np.zeros(p, dtype=self.dtype)
        begins = np.concatenate([[0], np.cumsum(self.group_sizes[self.screen_set])])
        for ss, g in enumerate(self.screen_set):
            q = self.group_sizes[g]
            beta[self.groups[g]:self.groups[g] + q] = self.screen_beta[begins[ss]:begins[ss] + q]
        nz = np.flatnonzero(beta)
        Ab = np.empty(p, dtype=self.dtype)
        A.mul(nz, beta[nz], Ab)
        ok = bool(np.allclose(self.grad, np.asarray(self.v) - Ab, atol=1e-6))
        logger.log(logging.INFO if ok else logging.ERROR, f"check grad = v - A beta: {'pass' if ok else 'FAIL'}")
        if method == "assert":
            assert ok, "grad = v - A beta"
        return ok


def gaussian_cov(
    *, A, v, constraints, groups, group_sizes, alpha, penalty, screen_set, screen_beta, screen_is_active, active_set_size,
    active_set, rsq, lmda, grad, lmda_path=None, lmda_max=None, max_iters=int(1e5), tol=1e-7, rdev_tol=1e-3, newton_tol=1e-12,
    newton_max_iters=1000, n_threads=1, early_exit=True, screen_rule="pivot", min_ratio=1e-2, lmda_path_size=100,
    max_screen_size=None, max_active_size=None, pivot_subset_ratio=0.1, pivot_subset_min=1, pivot_slack_ratio=1.25,
):
    """Creates a Gaussian, covariance method state object (reference ``adelie.state.gaussian_cov``, ``state.py:1128-1418``;
    argument meaning identical).  ``A`` is an ndarray or an ``adelie_amd.matrix.dense(method="cov")`` handle."""
    if isinstance(A, np.ndarray):
        A = _matrix.dense(A, method="cov", n_threads=n_threads)
    if not isinstance(A, (_matrix.MatrixCovBase64, _matrix.MatrixCovBase32)) or not hasattr(A, "_backend"):
        raise ValueError("A must be an instance of MatrixCovBase32, MatrixCovBase64, or np.ndarray.")
    dtype = A.dtype
    (max_screen_size, max_active_size, lmda_path_size, setup_lmda_max, setup_lmda_path, lmda_max, lmda_path) = \
        _render_inputs(groups=groups, lmda_max=lmda_max, lmda_path=lmda_path, lmda_path_size=lmda_path_size,
                       max_screen_size=max_screen_size, max_active_size=max_active_size, dtype=dtype)
    s = gaussian_cov_base()
    s.dtype = dtype
    s._X = s.A = s._A = A
    s.v = np.array(v, copy=True, dtype=dtype)
    s.constraints = constraints
    s.groups = np.array(groups, copy=True, dtype=int)
    s.group_sizes = np.array(group_sizes, copy=True, dtype=int)
    s.alpha = alpha
    s.penalty = np.array(penalty, copy=True, dtype=dtype)
    s.lmda_path = np.asarray(lmda_path, dtype=dtype)
    s.lmda_max = lmda_max
    s.min_ratio = min_ratio
    s.lmda_path_size = lmda_path_size
    s.max_screen_size = max_screen_size
    s.max_active_size = max_active_size
    s.pivot_subset_ratio = pivot_subset_ratio
    s.pivot_subset_min = pivot_subset_min
    s.pivot_slack_ratio = pivot_slack_ratio
    s.screen_rule = screen_rule
    s.max_iters = max_iters
    s.tol = tol
    s.rdev_tol = rdev_tol
    s.adev_tol = 0          # the covariance state passes 0 for both to the base state (state_gaussian_cov.hpp:118)
    s.ddev_tol = 0
    s.newton_tol = newton_tol
    s.newton_max_iters = newton_max_iters
    s.early_exit = early_exit
    s.setup_lmda_max = setup_lmda_max
    s.setup_lmda_path = setup_lmda_path
    s.intercept = False
    s.n_threads = n_threads
    s.screen_set = np.asarray(screen_set, dtype=int)
    s.screen_beta = np.asarray(screen_beta, dtype=dtype)
    s.screen_is_active = np.asarray(screen_is_active, dtype=bool)
    s.active_set_size = active_set_size
    s.active_set = np.asarray(active_set, dtype=int)
    s.rsq = rsq
    s.lmda = lmda
    s.grad = np.asarray(grad, dtype=dtype)
    s.error = ""
    G, p = len(s.groups), A.cols()
    if len(s.group_sizes) != G:
        raise RuntimeError("adelie_core: group_sizes must be (G,) where groups is (G,).")
    if len(s.penalty) != G:
        raise RuntimeError("adelie_core: penalty must be (G,) where groups is (G,).")
    s._check_constraints(G)
    if len(s.v) != p:
        raise RuntimeError("adelie_core: v must be (p,) where A is (p, p).")
    if len(s.grad) != p:
        raise RuntimeError("adelie_core: grad must be (p,) where A is (p, p).")
    return s
