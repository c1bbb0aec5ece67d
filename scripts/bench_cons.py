"""Constrained group paths: box / one-sided objects on groups of several coefficients, visited on the device (kernels_cons.hip)
against the same path with the visits forced onto the host objects (ADELIE_HIP_CONS_HOST=1: numpy classes through callbacks).

    python scripts/bench_cons.py [n p group_size n_constrained L]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import adelie_amd as ad  # noqa: E402
from bench import make_data  # noqa: E402


def run(n=100_000, p=10_000, gs=10, ncons=200, L=100):
    dev = torch.device("cuda", 0)
    X, y = make_data(n, p, seed=0, device=dev, dtype=torch.float64)
    Xd = ad.matrix.dense(X)
    G = p // gs
    groups = np.arange(0, p, gs)
    rng = np.random.default_rng(3)
    which = np.sort(rng.choice(G, ncons, replace=False))

    def make():
        cons = [None] * G
        for k, g in enumerate(which):  # half boxes, half one-sided (non-negativity of the group's coefficients)
            cons[g] = (ad.constraint.box(np.full(gs, -0.02), np.full(gs, 0.05)) if k % 2 == 0
                       else ad.constraint.lower(np.zeros(gs)))
        return cons

    kw = dict(groups=groups, alpha=0.5, early_exit=False, lmda_path_size=L, progress_bar=False)
    out = {"workload": f"Gaussian group elastic net {n}x{p}, groups of {gs}, alpha 0.5, {L} lambdas, {ncons} of {G} groups with a box / "
                       f"one-sided constraint object"}
    arms = (("device", None),) if os.environ.get("BENCH_CONS_DEVICE_ONLY") else (("device", None), ("host", "1"))
    for arm, env in arms:
        if env:
            os.environ["ADELIE_HIP_CONS_HOST"] = env
        try:
            ad.grpnet(Xd, ad.glm.gaussian(y), constraints=make(), **dict(kw, lmda_path_size=5, min_ratio=0.5))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = ad.grpnet(Xd, ad.glm.gaussian(y), constraints=make(), **kw)
            el = time.perf_counter() - t0
        finally:
            os.environ.pop("ADELIE_HIP_CONS_HOST", None)
        out[arm] = {"seconds": el, "paths_per_s": 1.0 / el, "lambdas": len(st.lmdas), "error": st.error,
                    "n_dev_cons_visits": st.counters["n_dev_cons_visits"], "n_host_cons_visits": st.counters["n_host_cons_visits"],
                    "final_active": int(st.active_set_size), "duals_nnz": int(st.duals[-1].nnz)}
        out[arm + "_betas"] = st.betas
    if "host_betas" in out:
        d = np.abs((out.pop("device_betas") - out.pop("host_betas"))).max()
        out["max_abs_dbeta_device_vs_host_objects"] = float(d)
    else:
        out.pop("device_betas")
    return out


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    print(json.dumps(run(*a)))
