#!/usr/bin/env python
"""bench.py — the grpnet hot path on MI355X, one JSON line per run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}]

``--config`` selects the BASELINE.json workload (default 2, the one the headline metric is quoted on):

  2  Gaussian lasso, dense 100k x 10k f64, 100-lambda path, ``early_exit=False``                    step = one path
  3  the same design, groups of 10, alpha = 0.5 (group elastic net)                                  step = one path
  4  binomial lasso (IRLS) on a 2-bit SNP design 500k x 50k (25 % ones, 5 % twos, 10 % missing)      step = one path
  5  ``cv_grpnet`` with 8 folds on the config-2 data                                                 step = one CV (8 fold paths)

The design is resident in HBM when the timed region starts.  Every line carries ``roofline`` (the dominant kernel, timed live
with HIP events on the stream it runs on, inside the timed steps), ``roofline_path`` (algorithmic matrix bytes
of the whole path / wall time), the solver's counters, and — rank 0, N = 1 — ``cpu_baseline``: the CPU oracle (``oracle/``, a
restatement of the reference algorithm with the reference's OpenMP flags; kind "port") on a bounded sample of the same
workload, on the box's host cores.

Multi-GPU (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``, one rank per GPU over RCCL):
  * configs 2-4: every rank holds a replica of the design and solves its own paths; rank r trains on the complement of CV
    fold r % 8 — the fold shards of ``cv_grpnet`` — no data-path collective, one all_gather of the per-lambda rows at the
    end.  Per-GPU work is fixed: ``"scaling": "weak"``; ``value`` = paths of all ranks / max-over-ranks wall.
  * config 5: ONE 8-fold CV whose folds are sharded over the ranks (fold k on rank k % N, one all_gather of the loss
    table): total work is fixed, ``"scaling": "strong"``.
  * the default line (config 2) additionally carries the legs ``"cfg3"``, ``"f32"`` (config 2 in single precision) and
    ``"cfg4"`` — the same measurement as ``--config 3 / 4`` with fewer steps, each with its own roofline objects — unless
    ``--no-extra-legs`` is given (profiling runs); at N = 1 also ``"standardized_snp_view"`` (a Gaussian path on config 4's
    2-bit design under the lazy standardized view: 6.25 GB resident instead of a 200 GB copy) and ``"sparse_resident"`` (a
    1M x 100k sparse design with 1e8 stored entries kept sparse: 745 GiB as dense f64), ROUNDS.md 9.9, and
  * ``"cv_config5"``: the same sharded CV timed right after the headline
    steps, so that a ``--gpus 1/2/4/8`` series holds BASELINE.json's second target (8-fold CV at 1/2/4/8 GPUs) as well.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X spec (guides/MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
HBM_COPY_GBS = 6290.0
MFMA_F64_PEAK_TFLOPS = 78.6  # dense f64 matrix-core peak (same guide)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32 matrix-core peak (same guide)

METRIC = "lambda-paths/sec (100-lambda grpnet); HBM GB/s vs peak"


# -------------------------------------------------------------------------------------------------------------------------
# synthetic data (mirrors adelie/data.py:84-235), generated on the device
# -------------------------------------------------------------------------------------------------------------------------
def make_data(n, p, seed, device, dtype):
    """Dense Gaussian design (column-major) and a 95 %-sparse linear response at snr = 1."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    Xt = torch.randn((p, n), generator=g, device=device, dtype=dtype)   # (p, n) row-major == (n, p) column-major
    X = Xt.t()
    gb = torch.Generator(device="cpu")
    gb.manual_seed(seed + 1)
    beta = torch.randn(p, generator=gb, dtype=torch.float64)
    mask = torch.rand(p, generator=gb) < 0.05
    beta = beta * mask
    eta = (X @ beta.to(device=device, dtype=dtype)).to(torch.float64).cpu().numpy()
    rng = np.random.default_rng(seed + 2)
    y = eta + float(np.sqrt(np.sum(beta.numpy() ** 2))) * rng.standard_normal(n)
    return X, y


def make_snp_data(n, p, seed, device):
    """int8 calldata (n, p) column-major on the device with adelie.data.snp_unphased's default rates (25 % ones, 5 % twos,
    10 % missing, data.py:222-235), its mean-impute vector, and a Bernoulli response from a sparse logistic model."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cdt = torch.empty((p, n), dtype=torch.int8, device=device)
    for j0 in range(0, p, 2048):
        j1 = min(p, j0 + 2048)
        u = torch.rand((j1 - j0, n), generator=g, device=device)
        blk = torch.zeros_like(u, dtype=torch.int8)
        blk[u < 0.25] = 1
        blk[(u >= 0.25) & (u < 0.30)] = 2
        blk[u >= 0.90] = -9
        cdt[j0:j1] = blk
        del u, blk
    cd = cdt.t()
    imp = torch.empty(p, dtype=torch.float64, device=device)
    for j0 in range(0, p, 2048):  # (column chunks: a whole-matrix mask / product / f64 reduction input would be 10x the calldata)
        blk = cdt[j0:j0 + 2048]
        valid = blk >= 0
        imp[j0:j0 + 2048] = (blk * valid).sum(dim=1, dtype=torch.float64) / valid.sum(dim=1).clamp(min=1)
        del blk, valid
    rng = np.random.default_rng(seed)
    beta = rng.standard_normal(p) * (rng.random(p) < min(0.05, 500 / p))
    eta = torch.zeros(n, dtype=torch.float64, device=device)
    for j in np.flatnonzero(beta):
        c = cd[:, j].to(torch.float64)
        eta += torch.where(c < 0, imp[j], c) * beta[j]
    eta = ((eta - eta.mean()) / eta.std()).cpu().numpy()
    y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    return cd, imp.cpu().numpy(), y


# -------------------------------------------------------------------------------------------------------------------------
# algorithmic bytes of a path: matrix bytes only; vectors are cache resident
# -------------------------------------------------------------------------------------------------------------------------
def path_bytes(counters, n, p, col_bytes_per_row, group_size, shared_launches=0):
    """B_path = s*n*[p*N_sweep + sum_visits q_g + sum_new q_g] (+ IRLS: s*n*2*sum_irls |S|), N_sweep = 2 + n_basil_iters.
    Visits are counted in groups by the group engines and in columns by the lasso engines: both are `group_size` columns.
    Concurrent CV folds answer several solvers' sweeps with ONE pass over X: those sweeps (`n_sweeps_shared`, summed over the
    solvers) are replaced by the number of shared launches, so that bytes which were never moved are not counted."""
    c = counters
    n_sweep = 2 + c["n_basil_iters"] - c.get("n_sweeps_shared", 0) + shared_launches
    visit_cols = (c["n_cd_visits_screen"] + c["n_cd_visits_active"]) * group_size
    cols = p * n_sweep + visit_cols + c["n_new_screen_cols"] + 2 * c.get("n_irls_screen_cols", 0)
    return float(n) * col_bytes_per_row * cols, {"n_sweep": int(n_sweep), "visit_cols": int(visit_cols),
                                                 "new_screen_cols": int(c["n_new_screen_cols"]),
                                                 "irls_screen_cols": int(c.get("n_irls_screen_cols", 0))}


def measured_traffic(key):
    """HBM bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE doubled as the gfx950 note in
    guides/MI355X_MICROARCH.md prescribes, plus WRITE_SIZE); None when this kernel / shape was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(key)
    except Exception:
        return None


def tag_traffic_source(obj):
    """`traffic` in a roofline object is not measured in this run: it is the per-launch HBM byte count of the PMC passes kept
    under profiles/ (a constant of the tree).  Say so next to every one of them."""
    if isinstance(obj, dict):
        if obj.get("traffic") is not None and "bound" in obj:
            obj["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run; not measured in this run)"
        for v in obj.values():
            tag_traffic_source(v)
    elif isinstance(obj, list):
        for v in obj:
            tag_traffic_source(v)


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


# -------------------------------------------------------------------------------------------------------------------------
class Ctx:
    """Process-wide plumbing shared by every leg of a run: rank / world, the torch device and the (optional) process group."""

    def __init__(self):
        import torch

        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        # test hooks (one-GPU boxes): BENCH_BACKEND=gloo + BENCH_DEVICE=0 run several ranks on one device to exercise the
        # multi-rank logic; the driver's multi-GPU runs use the defaults (RCCL, one device per rank)
        self.backend = os.environ.get("BENCH_BACKEND", "nccl")
        dev_index = int(os.environ.get("BENCH_DEVICE", self.local_rank))
        torch.cuda.set_device(dev_index)
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
            else:
                dist.init_process_group(backend=self.backend)
            self.dist = dist
        self.device = torch.device("cuda", dev_index)
        self.comm_device = self.device if self.backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, fn, steps):
        """EXACTLY `steps` calls bracketed by barrier + synchronize on both sides; the MAX over ranks."""
        self.barrier()
        t0 = time.perf_counter()
        outs = [fn() for _ in range(steps)]
        self.barrier()
        el = time.perf_counter() - t0
        if self.dist is not None:
            t = self.torch.tensor([el], device=self.comm_device, dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            el = float(t.item())
        return outs, el


def gathered_fold_counts(ctx, n_folds):
    """A multi-rank run must prove that the configured backend (RCCL unless BENCH_BACKEND says otherwise) carried the fold
    gather and that every rank contributed rows: the fold counts of the ranks, collected by an all_gather of their own."""
    torch, dist = ctx.torch, ctx.dist
    if ctx.backend == "nccl":
        assert dist.get_backend() == "nccl", dist.get_backend()
    mine = torch.tensor([len(range(ctx.rank, n_folds, ctx.world))], device=ctx.comm_device, dtype=torch.int64)
    per_rank = [torch.zeros_like(mine) for _ in range(ctx.world)]
    dist.all_gather(per_rank, mine)
    per_rank = [int(t.item()) for t in per_rank]
    assert sum(per_rank) == n_folds and all(c > 0 for c in per_rank[:min(ctx.world, n_folds)]), per_rank
    return per_rank


def measure(ctx, cfg, *, n, p, gs, alpha, dtype, L, steps, warmup, n_folds=8, cv_leg=False, data=None):
    """One workload: builds (or re-uses) the resident design, runs `warmup` untimed and `steps` timed steps, and returns
    (line, keep): `line` = the JSON object of this workload (value, rooflines, counters), `keep` = what the CPU baseline and a
    following leg on the same data need (design, response, last result)."""
    import adelie_amd as ad

    torch, dist, world, rank = ctx.torch, ctx.dist, ctx.world, ctx.rank
    tdtype = torch.float64 if dtype == "f64" else torch.float32
    npdtype = np.float64 if dtype == "f64" else np.float32
    s_val = np.dtype(npdtype).itemsize

    # ---- the workload -------------------------------------------------------------------------------------------------
    keep = dict(data or {})
    if cfg == 4:
        if "Xd" not in keep:
            cd, imp, y = make_snp_data(n, p, 0, ctx.device)
            keep.update(cd=cd, imp=imp, y=y, Xd=ad.matrix.snp_calldata(cd, imp, dtype=npdtype))  # packed to 2 bits on the device
        col_bytes_per_row = 0.25
        family = "binomial"
    else:
        if "Xd" not in keep:
            X, y = make_data(n, p, seed=0, device=ctx.device, dtype=tdtype)
            keep.update(X=X, y=y, Xd=ad.matrix.dense(X))        # adopts the resident tensor in place (no copy)
        col_bytes_per_row = s_val
        family = "gaussian"
    Xd = keep["Xd"]
    y = keep["y"].astype(npdtype)

    # fold weights for the weak-scaling replicas (the fold shards of cv_grpnet); full-data weights at N = 1
    weights = None
    if world > 1 and cfg != 5:
        order = np.random.RandomState(0).permutation(n)
        b, e = ad.cv.fold_ranges(n, 8)[rank % 8]
        weights = np.full(n, 1.0, dtype=npdtype)
        weights[order[b:e]] = 0
        weights /= weights.sum()
    glm = (ad.glm.binomial if family == "binomial" else ad.glm.gaussian)(y, weights=weights, dtype=npdtype)
    groups = None if gs == 1 else np.arange(0, p, gs)
    kw = dict(early_exit=False, lmda_path_size=L, groups=groups, alpha=alpha, progress_bar=False)
    cv_kw = dict(n_folds=n_folds, seed=0, lmda_path_size=L, process_group=(True if world > 1 else None))
    keep.update(glm=glm, kw=kw, cv_kw=cv_kw, npdtype=npdtype)

    def step():
        if cfg == 5:
            return ad.cv_grpnet(Xd, glm, **cv_kw)
        return ad.grpnet(Xd, glm, **kw)

    for _ in range(warmup):
        step()
    bs0 = Xd.batch_stats() if cfg == 5 else None
    outs, elapsed = ctx.timed(step, steps)
    bs1 = Xd.batch_stats() if cfg == 5 else None
    last = outs[-1]
    keep["last"] = last

    if cfg != 5:
        assert last.error == "", last.error
        assert len(last.lmdas) == L, len(last.lmdas)
        if dist is not None:  # the fold gather of cv_grpnet: one small all_gather of the per-lambda rows
            row = torch.from_numpy(np.asarray(last.devs, dtype=np.float64)).to(ctx.comm_device)
            rows = [torch.empty_like(row) for _ in range(world)]
            dist.all_gather(rows, row)
        units = world * steps                           # paths
        stats = [dict(counters=o.counters, timers=o.timers, total_time=o.total_time) for o in outs]
    else:
        assert last.losses.shape == (n_folds, L) and np.all(np.isfinite(last.losses))
        units = n_folds * steps                         # fold paths of ONE sharded CV per step
        stats = [f for o in outs for f in o.fold_stats]
        if dist is not None:
            keep["folds_per_rank_gathered"] = gathered_fold_counts(ctx, n_folds)

    # ---- the secondary CV leg of the default line ---------------------------------------------------------------------
    cv_obj = None
    if cv_leg:
        glm_full = ad.glm.gaussian(y, dtype=npdtype)
        cv_fn = lambda: ad.cv_grpnet(Xd, glm_full, **cv_kw)  # noqa: E731
        cv_fn()
        (cv_res,), cv_el = ctx.timed(cv_fn, 1)
        cv_obj = {
            "workload": f"cv_grpnet(n_folds={n_folds}, seed=0, min_ratio=0.1, {L} lambdas) on the same design; fold k on "
                        f"rank k % {world}, one all_gather of the ({n_folds}, {L}) loss table",
            "scaling": "strong", "n_gpus": world, "cv_wall_s": cv_el, "folds_per_s": n_folds / cv_el,
            "folds_per_rank": [len(range(r, n_folds, world)) for r in range(world)],
            "folds_per_rank_gathered": (gathered_fold_counts(ctx, n_folds) if dist is not None else None),
            "comm_backend": (dist.get_backend() if dist is not None else None),
            "best_idx": int(cv_res.best_idx), "min_avg_loss": float(cv_res.avg_losses.min()),
        }

    # ---- one extra, untimed step with per-launch events on the panel step kernel (second HBM-bound kernel of the path) --
    panel = None
    if rank == 0 and cfg in (2, 3, 4):
        os.environ["ADELIE_HIP_TIME_PANEL"] = "1"
        stp = ad.grpnet(Xd, glm, **kw)
        del os.environ["ADELIE_HIP_TIME_PANEL"]
        if stp.timers["n_panel_step_launches"] > 0:
            # gradient columns + residual-update columns (n_update_cols: columns, not groups)
            cols = stp.counters["n_panel_cols"] + stp.counters["n_update_cols"]
            bytes_ = float(cols) * n * col_bytes_per_row
            ms = stp.timers["t_panel_step_ms"]
            panel = {
                "kernel": ("panel_step_snp16_kernel (sequential form under IRLS, blocks of 64 visits: r -= X_B dbeta_B of the previous "
                           "block, partial gradients of the next block; 2-bit columns, one 32-bit word = 16 calls per lane; its last eight "
                           "workgroups sum the slice partials; the one-workgroup solve is a launch of its own between two steps)" if cfg == 4 else
                           ("panel_fused_grp_kernel" if gs > 1 else "panel_fused_kernel") +
                           " / panel_step_kernel (r -= X_B dbeta_B of the previous block; partial gradients of the "
                           "next block; the fused launch also carries the one-workgroup solve of the current block)"),
                "bound": "hbm", "achieved": bytes_ / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": measured_traffic(f"panel_step_kernel:{n}x{p}:{dtype}:g{gs}"),
                "launches": int(stp.timers["n_panel_step_launches"]), "avg_launch_ms": ms / stp.timers["n_panel_step_launches"],
                "algorithmic_bytes_per_launch": bytes_ / stp.timers["n_panel_step_launches"],
            }

    if rank != 0:
        return None, keep

    tot = lambda key, kind: sum(st[kind][key] for st in stats)  # noqa: E731
    # sweep kernel, timed live with HIP events on the solver's stream inside the timed steps
    sweep_launches = tot("n_sweep_launches", "timers")
    sweep_ms = tot("t_sweep_ms", "timers")
    sweep_bytes = float(n) * p * col_bytes_per_row      # one launch reads the design once
    sweep_avg = sweep_ms / max(sweep_launches, 1)
    sweep_roof = None
    if sweep_launches:
        ach = sweep_bytes / (sweep_avg * 1e-3) / 1e9
        kname = "sweep_kernel" if cfg != 4 else "sweep_snp_lut_kernel<Snp2bit, nibble tables>"
        sweep_roof = {
            "kernel": kname + " (grad = X^T (w*r) - rsum*xbar, full design)", "bound": "hbm", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": measured_traffic(f"sweep_kernel:{n}x{p}:{dtype}" + (":snp" if cfg == 4 else "")),
            "launches": int(sweep_launches), "avg_launch_ms": sweep_avg, "algorithmic_bytes_per_launch": sweep_bytes,
        }
    gram_ms, gram_flops = tot("t_gram_ms", "timers"), tot("gram_flops", "timers")
    gram_roof = None
    if gram_ms > 0:
        tf = gram_flops / (gram_ms * 1e-3) / 1e12
        peak = MFMA_F64_PEAK_TFLOPS if dtype == "f64" else MFMA_F32_PEAK_TFLOPS
        strips = cfg != 4   # Gaussian dense designs: every block build is a strip build (kernels_strip.hip)
        gram_bytes = float(tot("n_gram_col_reads", "counters")) * n * col_bytes_per_row
        gram_roof = {
            "kernel": (f"strip_lt_kernel / strip_kernel (the new rows of a panel block's diagonal and cross block, {dtype} MFMA "
                       "16x16x4 fed from HBM: bound by the column reads, see hbm_gbs)" if strips else
                       f"syrk_batch_kernel / gram_batch_kernel (diagonal and cross blocks X_b^T W X_b' of the panel engine, "
                       f"{dtype} MFMA 16x16x4)"),
            "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
            "frac": tf / peak, "traffic": measured_traffic(f"syrk_kernel:{n}x{p}:{dtype}"),
            "launches": int(tot("n_gram_launches", "timers")),
            "avg_launch_ms": gram_ms / max(tot("n_gram_launches", "timers"), 1),
            "algorithmic_flops_per_launch": gram_flops / max(tot("n_gram_launches", "timers"), 1),
            # the same launches against the HBM roofline: columns read (rows of the strip + the columns they meet) * n * s
            "hbm_gbs": gram_bytes / (gram_ms * 1e-3) / 1e9, "hbm_frac": gram_bytes / (gram_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": gram_bytes / max(tot("n_gram_launches", "timers"), 1),
        }
    # dominant kernel: by device time per path, among the three timed live (sweeps, block builds, panel steps).  Config 2: the
    # sweeps (118 against 103 ms of fused launches); config 3: the fused group launch; config 4 since round 6: the 2-bit panel
    # step (a latency chain: its HBM fraction is what the contract asks for); config 5: the shared sweep (below).  The other
    # two objects stand beside it (`roofline_sweep`, `roofline_gram_mfma`, `roofline_panel_step`)
    per_path = {"sweep": (sweep_ms / max(steps, 1), sweep_roof), "gram": (gram_ms / max(steps, 1), gram_roof)}
    if panel is not None:
        per_path["panel"] = (panel["avg_launch_ms"] * panel["launches"], panel)
    roofline = max((v for v in per_path.values() if v[1] is not None), key=lambda v: v[0], default=(0, sweep_roof))[1]
    if cfg == 2 and sweep_roof is not None:  # (118 against ~100 ms: pinned, so that run-to-run noise cannot swap the object)
        roofline = sweep_roof
    shared_launches = 0
    if cfg == 5 and bs1["launches"] > bs0["launches"]:
        # the folds in flight share their sweeps: the dominant kernel is the K-wide sweep on the batcher's stream, one
        # pass over X per launch for all the folds that reached their invariance sweep together
        nl, nvec, ms = (bs1[k] - bs0[k] for k in ("launches", "vectors", "ms"))
        shared_launches = int(nl)
        ach = nl * sweep_bytes / (ms * 1e-3) / 1e9
        roofline = {
            "kernel": "multi_sweep_kernel (shared sweep of the folds in flight: grad_k = X^T v_k for the K folds that "
                      "arrived together, X streamed once)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": measured_traffic(f"multi_sweep_kernel:{n}x{p}:{dtype}"), "launches": int(nl),
            "avg_launch_ms": ms / nl, "algorithmic_bytes_per_launch": sweep_bytes,
            "vectors_per_launch": nvec / nl,
        }

    # whole-path fraction: algorithmic bytes of everything the timed steps solved / their wall time
    counters = {k: sum(st["counters"][k] for st in stats) for k in stats[0]["counters"]}
    B, parts = path_bytes(counters, n, p, col_bytes_per_row, gs, shared_launches=shared_launches)
    wall = elapsed  # rank 0's steps; at N > 1 every rank does the same amount (cfg 5: rank 0's folds)
    path_gbs = B / wall / 1e9
    roofline_path = {"bound": "hbm", "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": path_gbs / HBM_PEAK_GBS,
                     # the same against the float4-copy ceiling measured on this part (guides/MI355X_MICROARCH.md: 6.29 TB/s)
                     "peak_measured_copy": HBM_COPY_GBS, "frac_of_measured_copy": path_gbs / HBM_COPY_GBS,
                     "algorithmic_bytes": B, "wall_s": wall,
                     "terms_in_columns": parts,
                     "note": "rank 0's paths over the timed steps; matrix bytes only (SURVEY.md 8d)" + (
                         "; sweeps that several folds shared are counted ONCE (one pass over X answers up to 8 folds), the "
                         "sweeps a fold made alone and the two preamble sweeps of every grpnet call once each" if cfg == 5 else "")}

    last_stat = stats[-1]
    tm = last_stat["timers"]
    workloads = {
        2: f"Gaussian GLM, dense X {n}x{p} {dtype} column-major resident in HBM, group size {gs}, alpha={alpha}, "
           f"{L}-lambda path, early_exit=False",
        3: f"Gaussian GLM, dense X {n}x{p} {dtype} column-major resident in HBM, group size {gs} ({p // gs} groups), "
           f"alpha={alpha}, {L}-lambda path, early_exit=False",
        4: f"Binomial GLM (IRLS), matrix.snp_unphased 2-bit packed {n}x{p} (25% ones, 5% twos, 10% missing, mean-imputed) "
           f"resident in HBM, lasso, {L}-lambda path, early_exit=False",
        5: f"cv_grpnet {n_folds}-fold, Gaussian GLM, dense X {n}x{p} {dtype} resident in HBM, lasso, "
           f"min_ratio=0.1, {L} lambdas, folds sharded fold k -> rank k % {world}",
    }
    out = {
        "metric": METRIC,
        "value": units / elapsed,
        "unit": "paths/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True,
        "scaling": "strong" if cfg == 5 else "weak",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": workloads[cfg] + ("" if world == 1 or cfg == 5 else
                                                  "; rank r trains on the complement of CV fold r%8 (weak scaling)"),
                   "baseline_config": cfg, "n": n, "p": p, "lmda_path_size": L, "group_size": gs, "alpha": alpha},
        "roofline": roofline,
        "roofline_path": roofline_path,
        "roofline_sweep": sweep_roof if roofline is not sweep_roof else None,
        "roofline_gram_mfma": gram_roof if roofline is not gram_roof else None,
        "roofline_panel_step": panel if roofline is not panel else None,
        "breakdown_ms_last_path": {
            "sweep": tm["t_sweep_ms"], "gram_mfma": tm["t_gram_ms"], "cd": tm["t_cd_ms"], "resid_axpy": tm["t_axpy_ms"],
            # host time between the KKT check of one lambda and the fit of the next (screening rule, appends, launches of the
            # new groups' variances), split into computing and waiting for the device (the in-stream speculative pass)
            "host_screen_compute": tm["t_host_screen_ms"] - tm["t_host_screen_wait_ms"],
            "host_screen_wait": tm["t_host_screen_wait_ms"],
            "total": 1e3 * last_stat["total_time"],
        },
        "counters": counters,
    }
    if cfg == 5:
        out["cv"] = {"cv_wall_s": elapsed / steps, "folds_per_s": units / elapsed, "n_folds": n_folds,
                     "folds_per_rank": [len(range(r, n_folds, world)) for r in range(world)],
                     "folds_per_rank_gathered": keep.get("folds_per_rank_gathered"),
                     "comm_backend": (dist.get_backend() if dist is not None else None),
                     "best_idx": int(last.best_idx), "min_avg_loss": float(last.avg_losses.min()),
                     "note": "counters / rooflines aggregate rank 0's folds only"}
    else:
        out["final_active"] = int(last.active_set_size)
        out["final_screen"] = int(len(last.screen_set))
    if cv_obj is not None:
        out["cv_config5"] = cv_obj
    return out, keep


LEG_KEYS = ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline", "roofline_path", "roofline_sweep",
            "roofline_gram_mfma", "roofline_panel_step", "breakdown_ms_last_path", "final_active", "final_screen")


def launch_ranks_if_needed(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the N ranks ourselves — re-exec under
    ``torch.distributed.run`` (one process per GPU, rendezvous on 127.0.0.1, RCCL) with the same arguments; rank 0 of the
    children prints the line.  Under a launcher (WORLD_SIZE set) the two numbers must agree.  Fails loudly when the box has
    fewer than N devices — except in the one-GPU rehearsal (BENCH_DEVICE set: every rank on that device, BENCH_BACKEND=gloo)."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks")
        return
    if args.gpus <= 1:
        return
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    if "BENCH_DEVICE" not in os.environ and have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, this box has {have} "
                         "(one-GPU rehearsal of the multi-rank logic: BENCH_BACKEND=gloo BENCH_DEVICE=0)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--p", type=int, default=None)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--lmda-path-size", type=int, default=100)
    ap.add_argument("--group-size", type=int, default=None, help="override the config's group size (configs 2/3)")
    ap.add_argument("--alpha", type=float, default=None, help="override the config's alpha (configs 2/3)")
    ap.add_argument("--n-folds", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cv-leg", action="store_true", help="skip the cv_config5 object of the default line")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the cfg3 / f32 / cfg4 objects of the default line (N = 1 only)")
    ap.add_argument("--cpu-budget-s", type=float, default=None)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-1thread-budget-s", type=float, default=12.0, help="config 2: extra CPU leg with one thread (0: skip)")
    ap.add_argument("--tight-tol-budget-s", type=float, default=30.0,
                    help="config 3: CPU budget of the tol = 1e-10 comparison of a lambda prefix (0: skip)")
    ap.add_argument("--cfg4-parity-budget-s", type=float, default=200.0,
                    help="config 4: CPU budget of the coarse-path parity sample with non-zero coefficients (0: skip)")
    args = ap.parse_args()
    launch_ranks_if_needed(args)
    cfg = args.config
    if args.steps is None:
        args.steps = {2: 3, 3: 2, 4: 1, 5: 3}[cfg]
    n = args.n or (500_000 if cfg == 4 else 100_000)
    p = args.p or (50_000 if cfg == 4 else 10_000)
    gs = args.group_size or (10 if cfg == 3 else 1)
    alpha = args.alpha if args.alpha is not None else (0.5 if cfg == 3 else 1.0)
    L = args.lmda_path_size

    ctx = Ctx()
    default_line = (cfg == 2 and gs == 1 and alpha == 1.0 and args.dtype == "f64" and args.n is None and args.p is None)
    out, keep = measure(ctx, cfg, n=n, p=p, gs=gs, alpha=alpha, dtype=args.dtype, L=L, steps=args.steps, warmup=args.warmup,
                        n_folds=args.n_folds, cv_leg=(default_line and not args.no_cv_leg))

    if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args, keep, keep["y"].astype(keep["npdtype"]), keep["glm"], keep["kw"],
                                           keep["cv_kw"], keep["npdtype"], keep["last"], keep["Xd"], n, p)

    # ---- the other BASELINE.json configurations, as objects of the default line (N = 1: they are single-GPU workloads) -----
    if default_line and ctx.world == 1 and not args.no_extra_legs:
        def leg(line):
            return {k: line[k] for k in LEG_KEYS if k in line}

        # bounded CPU legs of the other configurations (the headline's own is above): ~20 s of oracle time each
        leg_args = argparse.Namespace(**vars(args))
        leg_args.cpu_budget_s, leg_args.tight_tol_budget_s, leg_args.cfg4_parity_budget_s = 20.0, 15.0, 90.0
        leg_args.cpu_1thread_budget_s = 0.0

        def cpu_leg(c, kp, n_, p_):
            if args.no_cpu_baseline:
                return None
            try:
                return cpu_baseline(c, leg_args, kp, kp["y"].astype(kp["npdtype"]), kp["glm"], kp["kw"], kp["cv_kw"],
                                    kp["npdtype"], kp["last"], kp["Xd"], n_, p_)
            except Exception as e:  # noqa: BLE001  (a failed CPU leg is recorded, it does not cost the line)
                return {"error": repr(e)}

        if "cv_config5" in out:
            out["cv_config5"]["cpu_baseline"] = cpu_leg(5, keep, n, p)
        # config 3 re-uses the resident design of the headline (same X, grouped penalty)
        line3, keep3 = measure(ctx, 3, n=n, p=p, gs=10, alpha=0.5, dtype="f64", L=L, steps=2, warmup=1,
                               data={k: keep[k] for k in ("X", "y", "Xd", "Xh") if k in keep})
        out["cfg3"] = leg(line3)
        out["cfg3"]["cpu_baseline"] = cpu_leg(3, keep3, n, p)
        del keep3
        # box / one-sided constraint objects on 200 of config 3's 1000 groups: every visit of such a group is a device launch
        # (kernels_cons.hip); the same path with the visits on the host objects beside it (constraints)
        out["constrained_groups"] = constrained_leg(keep["Xd"], keep["y"], n, p)
        # throughput with independent headline paths IN FLIGHT on one GPU (alias handles of the resident design, one host thread
        # and stream each, as cv_grpnet's folds): `value` above is one path at a time -- a single path is a latency chain that
        # leaves part of the chip idle; this is what a server answering several fits gets.  Not the headline metric.
        out["concurrent_paths"] = concurrent_leg(keep["Xd"], keep["glm"], keep["kw"])
        del keep, line3
        import gc
        gc.collect()
        ctx.torch.cuda.empty_cache()
        # the headline workload in single precision (SURVEY.md 8d asks for both arithmetic types)
        line32, k32 = measure(ctx, 2, n=n, p=p, gs=1, alpha=1.0, dtype="f32", L=L, steps=3, warmup=1)
        out["f32"] = leg(line32)
        del k32, line32
        gc.collect()
        ctx.torch.cuda.empty_cache()
        line4, k4 = measure(ctx, 4, n=500_000, p=50_000, gs=1, alpha=1.0, dtype="f64", L=L, steps=2, warmup=1)
        out["cfg4"] = leg(line4)
        out["cfg4"]["cpu_baseline"] = cpu_leg(4, k4, 500_000, 50_000)
        # two designs whose dense f64 form does not fit (or barely fits) in HBM, as further objects of the default line
        # (ROUNDS.md 9.9): config 4's 2-bit design under the lazy standardized view, and a sparse design kept sparse
        out["standardized_snp_view"] = lazy_views_leg(ad_design=k4["Xd"], L=L, y_binomial=k4["y"])
        del k4, line4
        gc.collect()
        ctx.torch.cuda.empty_cache()
        out["sparse_resident"] = sparse_leg(L, cpu_budget_s=(0.0 if args.no_cpu_baseline else 15.0))

    if ctx.rank == 0:
        tag_traffic_source(out)
        print(json.dumps(out), flush=True)

    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


# -------------------------------------------------------------------------------------------------------------------------
# designs that do not fit dense (extra objects of the default line; a failure is recorded, it does not cost the headline)
# -------------------------------------------------------------------------------------------------------------------------
def lazy_views_leg(ad_design, L, y_binomial=None):
    """Gaussian lasso path, and config 4's own binomial path, on matrix.standardize(<config 4's 500k x 50k 2-bit design>) as a
    view (6.25 GB resident; the materialised copy would be 200 GB).  A lasso with an intercept on the view runs on the 2-bit
    matrix's own columns with rescaled penalty factors (adelie_amd/solver.py::_lasso_in_raw_coordinates): the panel engines."""
    import adelie_amd as ad

    try:
        X = ad_design
        n, p = X.shape
        rng = np.random.default_rng(7)
        beta = np.zeros(p)
        beta[rng.choice(p, 50, replace=False)] = rng.standard_normal(50)
        eta = np.zeros(n)
        X.btmul(0, p, beta, eta)
        y = eta + np.std(eta) * rng.standard_normal(n)
        Z = ad.matrix.standardize(X, lazy=True)
        kw = dict(lmda_path_size=L, min_ratio=2e-2, early_exit=False, progress_bar=False)
        ad.grpnet(Z, ad.glm.gaussian(y), **dict(kw, lmda_path_size=5, min_ratio=0.5))
        t0 = time.perf_counter()
        st = ad.grpnet(Z, ad.glm.gaussian(y), **kw)
        el = time.perf_counter() - t0
        out = {"workload": f"Gaussian lasso, {L} lambdas, standardize(snp_unphased {n}x{p}) as a view sharing the 2-bit matrix",
               "value": 1.0 / el, "unit": "paths/s", "ms_per_step": el * 1e3, "resident_bytes": int(n * p / 4),
               "dense_copy_bytes": int(n * p * 8), "lambdas": len(st.lmdas), "final_active": int(st.active_set_size),
               "error": st.error}
        if y_binomial is not None:
            t0 = time.perf_counter()
            sb = ad.grpnet(Z, ad.glm.binomial(np.asarray(y_binomial, dtype=np.float64)), lmda_path_size=L, early_exit=False,
                           progress_bar=False)
            el = time.perf_counter() - t0
            out["binomial"] = {"workload": "config 4's binomial lasso on the same standardized view", "value": 1.0 / el,
                               "unit": "paths/s", "ms_per_step": el * 1e3, "lambdas": len(sb.lmdas),
                               "final_active": int(sb.active_set_size), "n_irls_iters": int(sb.counters["n_irls_iters"]),
                               "error": sb.error}
            # the same as an elastic net: penalty * |s| and penalty * s^2 as separate factors (adelie_hip_grpnet_args::penalty_l2);
            # the view's own engines would rebuild the screen set's full Gram (12 GB at 39 k screened columns) per IRLS iteration
            t0 = time.perf_counter()
            se = ad.grpnet(Z, ad.glm.binomial(np.asarray(y_binomial, dtype=np.float64)), lmda_path_size=L, early_exit=False,
                           progress_bar=False, alpha=0.5)
            el = time.perf_counter() - t0
            out["binomial_elastic_net"] = {"workload": "the same with alpha = 0.5 (elastic net)", "value": 1.0 / el, "unit": "paths/s",
                                           "ms_per_step": el * 1e3, "lambdas": len(se.lmdas),
                                           "final_active": int(se.active_set_size),
                                           "n_irls_iters": int(se.counters["n_irls_iters"]),
                                           "n_panel_blocks": int(se.counters["n_panel_blocks"]), "error": se.error}
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def sparse_leg(L, n=1_000_000, p=100_000, density=1e-3, cpu_budget_s=0.0):
    """Gaussian lasso path on a sparse design kept sparse in HBM (matrix.sparse(resident="csc")): 745 GiB as dense f64."""
    import scipy.sparse as sp

    import adelie_amd as ad

    try:
        rng = np.random.default_rng(0)
        nnz = int(n * p * density)
        M = sp.csc_matrix((rng.standard_normal(nnz), (rng.integers(0, n, size=nnz), rng.integers(0, p, size=nnz))), shape=(n, p))
        M.sum_duplicates()
        M.sort_indices()
        beta = np.zeros(p)
        beta[rng.choice(p, 50, replace=False)] = rng.standard_normal(50) * 3
        y = M @ beta + rng.standard_normal(n)
        X = ad.matrix.sparse(M, resident="csc")
        kw = dict(lmda_path_size=L, min_ratio=1e-2, early_exit=False, progress_bar=False)
        ad.grpnet(X, ad.glm.gaussian(y), **dict(kw, lmda_path_size=5, min_ratio=0.5))
        t0 = time.perf_counter()
        st = ad.grpnet(X, ad.glm.gaussian(y), **kw)
        el = time.perf_counter() - t0
        out = {"workload": f"Gaussian lasso, {L} lambdas, sparse design {n}x{p} with {M.nnz} stored entries kept sparse (CSC + CSR)",
               "value": 1.0 / el, "unit": "paths/s", "ms_per_step": el * 1e3, "resident_bytes": int(M.nnz * 24 + (n + p + 2) * 8),
               "dense_copy_bytes": int(n * p * 8), "lambdas": len(st.lmdas), "final_active": int(st.active_set_size),
               "n_sweeps": int(st.counters["n_sweeps"]), "error": st.error}
        # dominant kernel: the full sweep over the tile-major copy (csc_tile_sweep_kernel: 2-byte row + value per stored entry,
        # the tile of v through LDS), timed live with HIP events on the solver's stream like the dense sweep
        nl, ms = st.timers["n_sweep_launches"], st.timers["t_sweep_ms"]
        if nl > 0:
            sweep_bytes = float(M.nnz) * 10 + 8.0 * n + 8.0 * p
            ach = sweep_bytes / (ms / nl * 1e-3) / 1e9
            out["roofline"] = {"kernel": "csc_tile_sweep_kernel (+ csc_sweep_reduce_kernel): grad = X^T v over the tile-major copy, "
                                         "full design", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": measured_traffic(f"csc_tile_sweep_kernel:{n}x{p}:f64"),
                               "launches": int(nl), "avg_launch_ms": ms / nl, "algorithmic_bytes_per_launch": sweep_bytes}
        if cpu_budget_s > 0:
            out["cpu_baseline"] = sparse_cpu_baseline(M, y, st, cpu_budget_s)
        # IRLS on the same design: the panel engine over compressed columns (step over the stored entries, 64-visit diagonal
        # blocks by hash joins of the row lists) instead of a Gram of the whole screen set per IRLS iteration
        yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-(y - y.mean()) / y.std()))).astype(np.float64)
        kwb = dict(lmda_path_size=50, min_ratio=5e-2, early_exit=False, progress_bar=False)
        t0 = time.perf_counter()
        sb = ad.grpnet(X, ad.glm.binomial(yb), **kwb)
        el = time.perf_counter() - t0
        out["binomial"] = {"workload": "binomial lasso (IRLS), 50 lambdas, min_ratio 0.05, on the same sparse design", "value": 1.0 / el,
                           "unit": "paths/s", "ms_per_step": el * 1e3, "lambdas": len(sb.lmdas), "final_active": int(sb.active_set_size),
                           "n_irls_iters": int(sb.counters["n_irls_iters"]), "n_panel_blocks": int(sb.counters["n_panel_blocks"]),
                           "n_block_builds": int(sb.counters["n_panel_grams"]), "error": sb.error}
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def concurrent_leg(Xd, glm, kw, rounds=2):
    """Aggregate paths/s of the headline workload with 2 and 4 identical paths in flight (threads on alias handles)."""
    from concurrent.futures import ThreadPoolExecutor

    import adelie_amd as ad

    try:
        out = {"workload": "the headline path, k independent copies in flight on one GPU (alias handles, one stream each)"}
        for k in (2, 4, 8):
            handles = [Xd] + [Xd.alias() for _ in range(k - 1)]

            def one(h):
                st = ad.grpnet(h, glm, **kw)
                assert st.error == "", st.error
                return len(st.lmdas)

            with ThreadPoolExecutor(max_workers=k) as pool:
                list(pool.map(one, handles))                     # warm-up (alias handles allocate their scratch)
                t0 = time.perf_counter()
                for _ in range(rounds):
                    list(pool.map(one, handles))
                el = time.perf_counter() - t0
            out[f"in_flight_{k}"] = {"value": k * rounds / el, "unit": "paths/s", "ms_per_path_wall": el / (k * rounds) * 1e3}
            # the same with the full-gradient sweeps of the paths in flight answered by ONE K-wide pass over X (what cv_grpnet's
            # folds do, solver.hip::SweepBatcher: every path sees its own result; summation order differs in the last bits)
            from adelie_amd.cv import _sweep_batch
            if _sweep_batch(Xd._backend, True):
                try:
                    with ThreadPoolExecutor(max_workers=k) as pool:
                        list(pool.map(one, handles))
                        t0 = time.perf_counter()
                        for _ in range(rounds):
                            list(pool.map(one, handles))
                        el = time.perf_counter() - t0
                    out[f"in_flight_{k}"]["shared_sweeps"] = {"value": k * rounds / el, "unit": "paths/s",
                                                              "ms_per_path_wall": el / (k * rounds) * 1e3}
                finally:
                    _sweep_batch(Xd._backend, False)
            del handles
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def constrained_leg(Xd, y, n, p, gs=10, ncons=200, L=50):
    """Config 3's design and grouping with constraint objects on `ncons` groups (half boxes, half non-negativity), `L` lambdas:
    device visits against host visits (ADELIE_HIP_CONS_HOST=1), same iterates."""
    import adelie_amd as ad

    try:
        G = p // gs
        groups = np.arange(0, p, gs)
        which = np.sort(np.random.default_rng(3).choice(G, ncons, replace=False))

        def make():
            cons = [None] * G
            for k, g in enumerate(which):
                cons[g] = (ad.constraint.box(np.full(gs, -0.02), np.full(gs, 0.05)) if k % 2 == 0 else ad.constraint.lower(np.zeros(gs)))
            return cons

        kw = dict(groups=groups, alpha=0.5, early_exit=False, lmda_path_size=L, progress_bar=False)
        out = {"workload": f"Gaussian group elastic net {n}x{p} f64, groups of {gs}, alpha 0.5, {L} lambdas, {ncons} of {G} groups "
                           f"carry a box / one-sided constraint object (ConstraintBox / ConstraintOneSided, proximal Newton)"}
        B = {}
        for arm, env in (("device_visits", None), ("host_visits", "1")):
            if env:
                os.environ["ADELIE_HIP_CONS_HOST"] = env
            try:
                ad.grpnet(Xd, ad.glm.gaussian(y), constraints=make(), **dict(kw, lmda_path_size=5, min_ratio=0.5))
                t0 = time.perf_counter()
                st = ad.grpnet(Xd, ad.glm.gaussian(y), constraints=make(), **kw)
                el = time.perf_counter() - t0
            finally:
                os.environ.pop("ADELIE_HIP_CONS_HOST", None)
            B[arm] = st.betas
            out[arm] = {"value": 1.0 / el, "unit": "paths/s", "ms_per_step": el * 1e3, "lambdas": len(st.lmdas), "error": st.error,
                        "n_dev_cons_visits": int(st.counters["n_dev_cons_visits"]),
                        "n_host_cons_visits": int(st.counters["n_host_cons_visits"]), "final_active": int(st.active_set_size),
                        "duals_nnz_last": int(st.duals[-1].nnz)}
        out["max_abs_dbeta_device_vs_host_visits"] = float(np.abs(B["device_visits"] - B["host_visits"]).max())
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def sparse_cpu_baseline(M, y, gpu_state, budget_s, ns=200_000, ps=20_000):
    """The oracle restates the reference's dense and SNP matrices only, so the CPU figure next to the sparse-resident leg is an
    INDEPENDENT solver: scikit-learn's coordinate-descent Lasso (same objective: adelie's Gaussian loss with weights 1/n is
    sklearn's (1/2n)||y - Xb - b0||^2, alpha = lmda) — on the top-left `ns` x `ps` SUB-BLOCK of the same matrix (on the full
    1M x 100k matrix its set-up alone takes 40 s), warm-started down the lambda grid of the GPU's path on that sub-block (kept
    sparse in HBM the same way) until the budget is spent; its coefficients double as a parity sample for the sparse kernels."""
    from sklearn.linear_model import Lasso

    import adelie_amd as ad

    Ms = M[:ns, :ps].tocsc()
    Ms.sort_indices()
    ys = np.ascontiguousarray(y[:ns])
    n, p = Ms.shape
    Xs = ad.matrix.sparse(Ms, resident="csc")
    kw = dict(lmda_path_size=len(gpu_state.lmdas), min_ratio=1e-2, early_exit=False, progress_bar=False, tol=1e-12)
    ad.grpnet(Xs, ad.glm.gaussian(ys), **dict(kw, lmda_path_size=5, min_ratio=0.5))
    t0 = time.perf_counter()
    gs = ad.grpnet(Xs, ad.glm.gaussian(ys), **kw)
    g_el = time.perf_counter() - t0
    lm = np.asarray(gs.lmdas)
    mdl = Lasso(alpha=float(lm[0]), fit_intercept=True, warm_start=True, tol=1e-12, max_iter=100000, selection="cyclic")
    t0 = time.perf_counter()
    k, db, nnz_cmp = 0, 0.0, 0
    for i, a in enumerate(lm):
        mdl.set_params(alpha=float(a))
        mdl.fit(Ms, ys)
        k = i + 1
        bg = np.asarray(gs.betas[i].toarray()).reshape(-1)
        db = max(db, float(np.abs(mdl.coef_ - bg).max()))
        nnz_cmp += int(np.count_nonzero(mdl.coef_))
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    L = len(lm)
    return {"value": (k / L) / el, "unit": "paths/s", "cores": 1, "kind": "independent",
            "sample": (f"scikit-learn Lasso (cyclic coordinate descent, tol 1e-12, warm starts) on the top-left {n}x{p} sub-block "
                       f"({Ms.nnz} stored entries) of the same scipy CSC matrix and response, first {k} of {L} lambdas of the GPU path's "
                       f"grid on that sub-block in a {budget_s:.0f} s budget, 1 thread; value = (solved fraction)/time on the SUB-BLOCK "
                       f"problem, an upper bound on its paths/s; the GPU solves that sub-block's full path (kept sparse, tol 1e-12) in "
                       f"gpu_same_sample_s; the oracle has no sparse design"),
            "seconds": el, "lambdas_solved": k, "max_abs_dbeta_vs_gpu": db, "nonzero_coefficients_compared": nnz_cmp,
            "gpu_same_sample_s": g_el, "gpu_same_sample_final_active": int(gs.active_set_size)}


# -------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (reference algorithm, OpenMP) on the same data, bounded
# -------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg, args, keep, y, glm, kw, cv_kw, npdtype, gpu_last, Xd, n, p):
    from oracle import oracle

    import adelie_amd as ad

    # Thread count: the reference documents that more threads are not faster for this solver (parallelism.ipynb cells
    # 10-18: its OpenMP regions are per column visit).  Probed on the 256-core GPU-box host (scripts/cpu_threads_probe.py,
    # first 25 lambdas): 8 thr 2.27 s, 16 thr 1.31 s, 32 thr 1.65 s, 64 thr 3.2 s, 128 thr 6.3 s -> 16.
    cores = min(host_cores(), args.cpu_threads)
    os.environ["ORACLE_COL_THREADS"] = str(cores)
    os.environ.setdefault("OMP_PROC_BIND", "TRUE")  # reference adelie/__init__.py:8-19
    budget = args.cpu_budget_s if args.cpu_budget_s is not None else {2: 30.0, 3: 30.0, 4: 30.0, 5: 30.0}[cfg]
    L = args.lmda_path_size
    base = {"unit": "paths/s", "cores": cores, "kind": "port", "host_cores_available": host_cores()}

    parity = {}

    def bounded_path(Xo, glm_, kw_, gpu_state, budget_s=None, X_eval=None, tag=None):
        """Runs the oracle until the budget is spent; returns (solved lambdas, seconds, max|dbeta| vs the GPU path) and
        records, under `parity[tag]`, what the reference's own fall-back criterion compares (tests/test_solver.py:446-466):
        the objective of both solutions at every solved lambda, evaluated by ONE evaluator (diagnostic.objective on the
        resident design), plus how many coefficients the comparison covered."""
        b = budget if budget_s is None else budget_s
        t0 = time.perf_counter()
        st = ad.grpnet(Xo, glm_, n_threads=cores, exit_cond=lambda s: (time.perf_counter() - t0) > b, **kw_)
        el = time.perf_counter() - t0
        k = len(st.lmdas)
        db = None
        if k:
            Bc, Bg = st.betas.toarray(), gpu_state.betas[:k].toarray()
            db = float(np.abs(Bc - Bg).max())
            info = {"lambdas_compared": k, "coefficients_compared": int(Bc.size),
                    "nonzero_cpu": int(np.count_nonzero(Bc)), "nonzero_gpu": int(np.count_nonzero(Bg)),
                    "lambdas_with_nonzero_coefficients": int(np.count_nonzero(np.any(Bc != 0, axis=1))),
                    "max_abs_dbeta": db,
                    "max_abs_dintercept": float(np.abs(np.asarray(st.intercepts) - np.asarray(gpu_state.intercepts)[:k]).max())}
            if X_eval is not None:
                okw = dict(lmdas=np.asarray(st.lmdas), groups=kw_.get("groups"), alpha=kw_.get("alpha", 1))
                o_c = ad.diagnostic.objective(X_eval, glm_, st.betas, np.asarray(st.intercepts), **okw)
                o_g = ad.diagnostic.objective(X_eval, glm_, gpu_state.betas[:k], np.asarray(gpu_state.intercepts)[:k], **okw)
                rel = (o_g - o_c) / np.maximum(np.abs(o_c), np.finfo(np.float64).tiny)
                # which of the reference's two criteria this comparison meets (tests/test_solver.py:444-445 coefficients to 1e-6;
                # :446-466 else the objective): a path that stops on `tol` is resolved to the stopping rule, not to 1e-6 in beta
                info["criterion"] = ("coefficients" if db <= 1e-6 else "objective")
                info.update(max_rel_objective_gap_gpu_minus_cpu=float(rel.max()),
                            min_rel_objective_gap_gpu_minus_cpu=float(rel.min()),
                            reference_criterion_obj_gpu_le_obj_cpu_x_1p1e8=bool(np.all(o_g <= o_c * (1 + 1e-8) + 1e-300)
                                                                                or np.all(np.abs(rel) <= 1e-8)))
            parity[tag or "default_tol"] = info
        return k, el, db

    def host_copy():
        if "Xh" not in keep:
            keep["Xh"] = keep["X"].t().contiguous().cpu().numpy().T  # (n, p) F-ordered host copy of the same matrix
        return keep["Xh"]

    if cfg in (2, 3):
        Xh = host_copy()
        Xo = oracle.dense(Xh, n_threads=cores)
        k, el, db = bounded_path(Xo, glm, kw, gpu_last, X_eval=Xd)
        if k == L:
            value, sample = 1.0 / el, f"full {L}-lambda path on the same {n}x{p} data (host copy), {cores} OpenMP threads"
        else:
            value = (k / L) / el
            sample = (f"first {k} of {L} lambdas of the same path on the same data (time budget {budget:.0f} s), {cores} OpenMP "
                      f"threads; value = (solved fraction)/time, an UPPER bound on the CPU paths/s: later lambdas cost more")
        out = dict(base, value=value, sample=sample, seconds=el, lambdas_solved=k, max_abs_dbeta_vs_gpu=db)
        if cfg == 3 and args.tight_tol_budget_s > 0:
            # At the default tol = 1e-7 two correct group-elastic-net runs differ by the stopping rule's resolution (DESIGN.md,
            # numerics); the same path at tol = 1e-10 on both sides (a lambda prefix, own budget) separates that from a defect
            kw_t = dict(kw, tol=1e-10)
            g_t = ad.grpnet(Xd, glm, **kw_t)
            bounded_path(Xo, glm, kw_t, g_t, budget_s=args.tight_tol_budget_s, X_eval=Xd, tag="tol_1e-10")
        out["parity"] = parity
        if cfg == 2 and args.cpu_1thread_budget_s > 0:
            # the reference documents that one thread is often the fastest setting for this solver (parallelism.ipynb cell 18):
            # the same path with n_threads = 1, on its own (shorter) budget
            os.environ["ORACLE_COL_THREADS"] = "1"
            b1 = args.cpu_1thread_budget_s
            t0 = time.perf_counter()
            st1 = ad.grpnet(oracle.dense(Xh, n_threads=1), glm, n_threads=1,
                            exit_cond=lambda s: (time.perf_counter() - t0) > b1, **kw)
            e1 = time.perf_counter() - t0
            k1 = len(st1.lmdas)
            out["one_thread"] = {"value": (k1 / L) / e1, "unit": "paths/s", "cores": 1, "seconds": e1, "lambdas_solved": k1,
                                 "sample": f"first {k1} of {L} lambdas in a {b1:.0f} s budget, 1 thread; (solved fraction)/time, "
                                           f"an upper bound on the 1-thread paths/s"}
        return out

    if cfg == 4:
        # FULL-SIZE leg: the oracle on the same n x p int8 calldata and response for a lambda prefix, with
        # the GPU path's time over the same prefix beside it.  The oracle keeps the calldata as int8 on the host (25 GB at
        # 500k x 50k, plus its own copy): when the host cannot hold that, the leg falls back to the top-left sub-block and says so.
        import psutil

        need = 3 * n * p  # bytes: host copy + the oracle's own copy + slack
        avail = psutil.virtual_memory().available
        full = avail >= need
        ns, ps = (n, p) if full else (min(n, 100_000), min(p, 10_000))
        cd_s = keep["cd"][:ns, :ps]
        cd_h = cd_s.t().contiguous().cpu().numpy().T  # (ns, ps) F-ordered, no extra host copy
        imp_s = keep["imp"][:ps]
        y_s = np.ascontiguousarray(y[:ns])
        glm_s = ad.glm.binomial(y_s, dtype=npdtype)
        if full:
            g_state, g_el = gpu_last, None  # the timed GPU path of this run IS the same problem
        else:
            Xg = ad.matrix.snp_calldata(cd_h, imp_s, dtype=npdtype)
            ad.grpnet(Xg, glm_s, **kw)
            t0 = time.perf_counter()
            g_state = ad.grpnet(Xg, glm_s, **kw)
            g_el = time.perf_counter() - t0
        Xo4 = oracle.snp_calldata(cd_h, imp_s, dtype=npdtype, n_threads=cores)
        k, el, db = bounded_path(Xo4, glm_s, kw, g_state, X_eval=(Xd if full else Xg))
        if args.cfg4_parity_budget_s > 0:
            # The path's first lambdas have nothing active, and the oracle solves only a few of them in the throughput budget:
            # a COARSE path over the same data (12 lambdas down to 0.5 lmda_max: non-zero coefficients from the second one on)
            # solved by both sides is the full-size parity sample
            kw_c = dict(kw, lmda_path_size=12, min_ratio=0.5)
            g_c = ad.grpnet(Xd if full else Xg, glm_s, **kw_c)
            bounded_path(Xo4, glm_s, kw_c, g_c, budget_s=args.cfg4_parity_budget_s, X_eval=(Xd if full else Xg),
                         tag="coarse_path_12_lambdas_min_ratio_0.5")
        g_prefix = float(np.sum(g_state.benchmark_fit_screen[:k]) + np.sum(g_state.benchmark_fit_active[:k])) if k else None
        out = dict(base, value=(k / L) / el, seconds=el, lambdas_solved=k, max_abs_dbeta_vs_gpu=db, full_size=bool(full),
                   gpu_fit_seconds_same_prefix=g_prefix, parity=parity)
        if full:
            out["sample"] = (f"the SAME {n}x{p} calldata and response (host int8 copy), first {k} of {L} lambdas in a {budget:.0f} s "
                             f"budget, {cores} OpenMP threads; value = (solved fraction)/time, an UPPER bound on the CPU paths/s "
                             f"(later lambdas cost more); gpu_fit_seconds_same_prefix = the GPU path's fit time over those lambdas")
        else:
            out["sample"] = (f"host RAM available {avail / 2**30:.0f} GiB < {need / 2**30:.0f} GiB needed for the full calldata: "
                             f"sub-block {ns}x{ps} (top-left) of the same calldata and response, first {k} of {L} lambdas in a "
                             f"{budget:.0f} s budget, {cores} OpenMP threads; value = (solved fraction)/time on the SUB-BLOCK "
                             f"problem; the GPU path solves that sub-block's full path in gpu_same_sample_s")
            out.update(gpu_same_sample_s=g_el, gpu_same_sample_paths_per_s=1.0 / g_el)
        return out

    # cfg 5: one fold of the same CV (the full-data lmda_max call + the fold's two grpnet calls) with the oracle
    Xh = host_copy()
    Xo = oracle.dense(Xh, n_threads=cores)
    np.random.seed(0)
    order = np.random.choice(n, n, replace=False)
    b, e = ad.cv.fold_ranges(n, args.n_folds)[0]
    w = np.full(n, 1.0 / n)
    w[order[b:e]] = 0
    w /= w.sum()
    glm_c = ad.glm.gaussian(y, weights=w, dtype=npdtype)
    t0 = time.perf_counter()
    full = ad.grpnet(Xo, glm, n_threads=cores, lmda_path_size=0, progress_bar=False)
    lm = full.lmda_max * np.logspace(0, -1, L)
    st = ad.grpnet(Xo, glm_c, n_threads=cores, lmda_path=lm, early_exit=False, progress_bar=False,
                   exit_cond=lambda s: (time.perf_counter() - t0) > budget)
    el = time.perf_counter() - t0
    k = len(st.lmdas)
    return dict(base, value=(k / L) / el, seconds=el, lambdas_solved=k,
                sample=(f"fold 0 of the same {args.n_folds}-fold CV (training weights of fold 0, the CV's lambda grid down to "
                        f"min_ratio 0.1), first {k} of {L} lambdas in a {budget:.0f} s budget, {cores} OpenMP threads; value = "
                        f"fold paths/s as (solved fraction)/time (upper bound); the reference runs its folds one after another"))


if __name__ == "__main__":
    main()
