#!/usr/bin/env python
"""bench.py — headline benchmark of the grpnet hot path on MI355X.

A *step* is one full 100-lambda ``grpnet`` path (Gaussian GLM, ungrouped lasso, ``early_exit=False``) on a dense
synthetic Gaussian design that is already resident in HBM when the timed region starts.  At N=1 the workload is
BASELINE.json ``configs[1]`` (100k x 10k, f64).  At N>1 every rank keeps a full replica of X (same seed) and solves
its own paths — rank r trains on the complement of CV fold r%8, i.e. the fold shards of ``cv_grpnet`` — with no
data-path collective; one RCCL all_gather of the per-lambda deviance rows at the end (the fold gather).  Per-GPU
work is fixed as N grows: "scaling": "weak".  ``value`` = paths solved by all ranks / max-over-ranks wall time.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — dominant HBM-bound kernel (the full gradient sweep grad = X^T(w*r) - rsum*xbar), timed live
                 with HIP events on the stream it is launched on, inside the timed steps;
  cpu_baseline — the CPU oracle (oracle/, kind "port": a restatement of the reference algorithm, OpenMP) on the
                 same data copied back to the host, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (guides/MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def make_data(n, p, seed, device, dtype):
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    # column-major (n, p): generate (p, n) row-major and view transposed
    Xt = torch.randn((p, n), generator=g, device=device, dtype=dtype)
    X = Xt.t()
    gb = torch.Generator(device="cpu")
    gb.manual_seed(seed + 1)
    beta = torch.randn(p, generator=gb, dtype=torch.float64)
    mask = torch.rand(p, generator=gb) < 0.05  # 95% sparse truth (adelie.data.dense defaults)
    beta = beta * mask
    eta = (X @ beta.to(device=device, dtype=dtype)).to(torch.float64).cpu().numpy()
    rng = np.random.default_rng(seed + 2)
    noise_scale = float(np.sqrt(np.sum(beta.numpy() ** 2)))  # snr = 1
    y = eta + noise_scale * rng.standard_normal(n)
    return X, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--p", type=int, default=10_000)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--lmda-path-size", type=int, default=100)
    ap.add_argument("--group-size", type=int, default=1)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=90.0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    import torch

    import adelie_amd as ad

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # test hooks (one-GPU boxes): BENCH_BACKEND=gloo + BENCH_DEVICE=0 run several ranks on one device to exercise the
    # multi-rank logic; the driver's multi-GPU runs use the defaults (RCCL, one device per rank)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("BENCH_DEVICE", local_rank))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm_device = device if backend == "nccl" else torch.device("cpu")
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    npdtype = np.float64 if args.dtype == "f64" else np.float32
    n, p = args.n, args.p

    X, y = make_data(n, p, seed=0, device=device, dtype=tdtype)
    Xd = ad.matrix.dense(X)  # adopts the resident tensor in place (no copy)
    y = y.astype(npdtype)

    # fold weights for the multi-GPU (CV-shard) workload; full-data weights at N=1
    weights = None
    if world > 1:
        order = np.random.RandomState(0).permutation(n)
        from adelie_amd.cv import fold_ranges

        b, e = fold_ranges(n, 8)[rank % 8]
        weights = np.full(n, 1.0, dtype=npdtype)
        weights[order[b:e]] = 0
        weights /= weights.sum()
    glm = ad.glm.gaussian(y, weights=weights, dtype=npdtype)

    groups = None if args.group_size == 1 else np.arange(0, p, args.group_size)
    kw = dict(early_exit=False, lmda_path_size=args.lmda_path_size, groups=groups, alpha=args.alpha)

    def step():
        return ad.grpnet(Xd, glm, **kw)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        state = step()
    barrier()
    t0 = time.perf_counter()
    sweep_ms = 0.0
    sweep_launches = 0
    states = []
    for _ in range(args.steps):
        state = step()
        sweep_ms += state.timers["t_sweep_ms"]
        sweep_launches += state.timers["n_sweep_launches"]
        states.append(state)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=comm_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the fold gather of cv_grpnet: one small all_gather of the per-lambda rows
        row = torch.from_numpy(np.asarray(state.devs, dtype=np.float64)).to(comm_device)
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
    assert state.error == "", state.error
    assert len(state.lmdas) == args.lmda_path_size

    # one extra, untimed step with per-launch events on the panel step kernel (second HBM-bound kernel of the path)
    panel = None
    if rank == 0:
        os.environ["ADELIE_HIP_TIME_PANEL"] = "1"
        stp = step()
        del os.environ["ADELIE_HIP_TIME_PANEL"]
        if stp.timers["n_panel_step_launches"] > 0:
            s_ = np.dtype(npdtype).itemsize
            cols = stp.counters["n_panel_cols"] + stp.counters["n_updates"]  # gradient columns + residual-update columns
            bytes_ = float(cols) * n * s_
            ms = stp.timers["t_panel_step_ms"]
            panel = {
                "kernel": "panel_fused_kernel / panel_step_kernel (r -= X_B dbeta_B of the previous block; partial gradients of the "
                          "next block; the fused launch also carries the one-workgroup solve of the current block)",
                "bound": "hbm", "achieved": bytes_ / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": measured_traffic(n, p, args.dtype, "panel_step_kernel"),
                "launches": int(stp.timers["n_panel_step_launches"]), "avg_launch_ms": ms / stp.timers["n_panel_step_launches"],
                "algorithmic_bytes_per_launch": bytes_ / stp.timers["n_panel_step_launches"],
            }

    if rank == 0:
        s = np.dtype(npdtype).itemsize
        sweep_bytes = float(n) * p * s  # algorithmic bytes of one launch: X read once (vectors are cache resident)
        avg_ms = sweep_ms / max(sweep_launches, 1)
        achieved = sweep_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        c = state.counters
        tm = state.timers
        out = {
            "metric": "lambda-paths/sec (100-lambda grpnet, dense Gaussian)",
            "value": world * args.steps / elapsed,
            "unit": "paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"Gaussian GLM, dense X {n}x{p} {args.dtype} column-major resident in HBM, "
                            f"group size {args.group_size}, alpha={args.alpha}, {args.lmda_path_size}-lambda path, "
                            f"early_exit=False"
                            + ("" if world == 1 else f"; rank r trains on the complement of CV fold r%8 (weak scaling)"),
                "n": n, "p": p, "lmda_path_size": args.lmda_path_size,
            },
            "roofline": {
                "kernel": "sweep_kernel (grad = X^T (w*r) - rsum*xbar, full design)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(n, p, args.dtype),
                "launches": int(sweep_launches),
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": sweep_bytes,
            },
            "roofline_panel_step": panel,
            "breakdown_ms_last_step": {
                "sweep": tm["t_sweep_ms"], "gram_mfma": tm["t_gram_ms"], "cd": tm["t_cd_ms"], "resid_axpy": tm["t_axpy_ms"],
                "host_screen": tm["t_host_screen_ms"], "total": 1e3 * state.total_time,
                "gram_tflops": (tm["gram_flops"] / (tm["t_gram_ms"] * 1e-3) / 1e12) if tm["t_gram_ms"] > 0 else None,
            },
            "counters": c,
            "final_active": int(state.active_set_size),
            "final_screen": int(len(state.screen_set)),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X, y, glm, kw, args, npdtype, state)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(n, p, dtype, kernel="sweep_kernel"):
    """HBM bytes per sweep launch from the PMC passes kept under profiles/ (FETCH_SIZE doubled as the gfx950 note in
    guides/MI355X_MICROARCH.md prescribes, plus WRITE_SIZE); None when this shape was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(f"{kernel}:{n}x{p}:{dtype}")
    except Exception:
        return None


def cpu_baseline(X, y, glm, kw, args, npdtype, gpu_state):
    """Times the CPU oracle (a port of the reference algorithm, same OpenMP flags) on the same data."""
    from oracle import oracle

    import adelie_amd as ad

    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # Thread count: the reference documents that more threads are not faster for this solver (parallelism.ipynb cells
    # 10-18: its OpenMP regions are per column visit).  Probed on the 256-core GPU-box host (scripts/cpu_threads_probe.py,
    # first 25 lambdas): 8 thr 2.27 s, 16 thr 1.31 s, 32 thr 1.65 s, 64 thr 3.2 s, 128 thr 6.3 s -> use 16.
    cores = min(avail, args.cpu_threads)
    os.environ["ORACLE_COL_THREADS"] = str(cores)
    os.environ.setdefault("OMP_PROC_BIND", "TRUE")  # reference adelie/__init__.py:8-19
    Xh = X.t().contiguous().cpu().numpy().T  # (n, p) F-ordered host copy of the same matrix
    Xo = oracle.dense(Xh, n_threads=cores)
    budget = args.cpu_budget_s
    t0 = time.perf_counter()
    done = {"n": 0}

    def exit_cond(view):
        done["n"] = view.n_solutions
        return (time.perf_counter() - t0) > budget

    st = ad.grpnet(Xo, glm, n_threads=cores, exit_cond=exit_cond, **kw)
    el = time.perf_counter() - t0
    n_sol = len(st.lmdas)
    L = args.lmda_path_size
    full = n_sol == L
    db = float(np.abs(st.betas.toarray() - gpu_state.betas[:n_sol].toarray()).max()) if n_sol else None
    if full:
        value = 1.0 / el
        sample = f"full {L}-lambda path on the same {args.n}x{args.p} data (host copy), {cores} OpenMP threads"
    else:
        # time budget hit: report the rate on the solved prefix, scaled by the GPU path's own share of work on it
        value = (n_sol / L) / el
        sample = (f"first {n_sol} of {L} lambdas of the same path (time budget {budget:.0f}s), value scaled as "
                  f"(solved fraction)/time: an UPPER bound on the CPU paths/s since later lambdas cost more")
    return {"value": value, "unit": "paths/s", "cores": cores, "kind": "port", "sample": sample,
            "seconds": el, "max_abs_dbeta_vs_gpu": db}


if __name__ == "__main__":
    main()
