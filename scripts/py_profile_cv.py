"""Where the host time of an 8-fold cv_grpnet goes: cProfile of the folds run one after another (n_concurrent = 1: every Python
frame is on the profiled thread), next to the wall time of the default concurrent run."""
import os, sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
glm = ad.glm.gaussian(y)
ad.cv_grpnet(Xd, glm, n_folds=8, seed=0)
for _ in range(2):
    t0 = time.perf_counter()
    ad.cv_grpnet(Xd, glm, n_folds=8, seed=0)
    print("cv (8 folds in flight) %.3f s" % (time.perf_counter() - t0))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = ad.cv_grpnet(Xd, glm, n_folds=8, seed=0, n_concurrent=1)
pr.disable()
print("cv (one fold at a time) %.3f s" % (time.perf_counter() - t0))
ps = pstats.Stats(pr).sort_stats("tottime")
ps.print_stats(22)
ps.sort_stats("cumulative").print_stats(30)
