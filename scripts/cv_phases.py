"""Wall-clock phases of an 8-fold cv_grpnet with all folds in flight: per fold, when its two grpnet calls and its
post-processing start and end relative to the CV's start (ms)."""
import os, sys, time, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from adelie_amd import cv as cvm
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
glm = ad.glm.gaussian(y)
ad.cv_grpnet(Xd, glm, n_folds=8, seed=0)
ev = []
T0 = [0.0]
orig = cvm.grpnet
def timed(*a, **k):
    t0 = time.perf_counter()
    r = orig(*a, **k)
    ev.append((threading.get_ident() % 1000, "path" if k.get("lmda_path") is not None else "boot", (t0 - T0[0]) * 1e3, (time.perf_counter() - T0[0]) * 1e3,
               r.total_time * 1e3 if hasattr(r, "total_time") else 0))
    return r
cvm.grpnet = timed
origl = cvm._fold_loss
def timedl(*a, **k):
    t0 = time.perf_counter()
    r = origl(*a, **k)
    ev.append((threading.get_ident() % 1000, "fold", (t0 - T0[0]) * 1e3, (time.perf_counter() - T0[0]) * 1e3, 0))
    return r
cvm._fold_loss = timedl
for rep in range(2):
    ev.clear()
    T0[0] = time.perf_counter()
    ad.cv_grpnet(Xd, glm, n_folds=8, seed=0)
    tot = (time.perf_counter() - T0[0]) * 1e3
    print("cv wall %.1f ms" % tot)
    for e in sorted(ev, key=lambda e: (e[0], e[2])):
        print("  thr %3d %-5s start %7.1f end %7.1f  (%.1f ms, native total_time %.1f)" % (e[0], e[1], e[2], e[3], e[3] - e[2], e[4]))

# one rank's share of the same CV on this GPU alone (fold k on rank k % N: 8 / N folds), as the first rank of an N-GPU run would
# see it: the expected strong-scaling table of DESIGN.md 9 (no gather, no contention from other ranks: every rank has its own GPU)
cvm.grpnet, cvm._fold_loss = orig, origl
for N in (1, 2, 4, 8):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        ad.cv_grpnet(Xd, glm, n_folds=8, seed=0, _share=(0, N))
        ts.append((time.perf_counter() - t0) * 1e3)
    print("share of rank 0 of %d (%d folds): %s ms" % (N, 8 // N, " ".join("%.1f" % t for t in ts)))
