import os, sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
glm = ad.glm.gaussian(y)
ad.grpnet(Xd, glm, early_exit=False, lmda_path_size=10)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = ad.cv_grpnet(Xd, glm, n_folds=8, seed=0)
pr.disable()
print("cv %.2f s" % (time.perf_counter() - t0))
ps = pstats.Stats(pr).sort_stats("cumulative")
ps.print_stats(25)
