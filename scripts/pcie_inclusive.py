"""Host-resident design: time of the one-time upload (adelie_hip_design_create_dense) next to one headline path."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
Xh = X.t().contiguous().cpu().numpy().T           # (n, p) F-ordered pageable host copy
del X; torch.cuda.empty_cache()
for rep in range(2):
    t0 = time.perf_counter(); Xd = ad.matrix.dense(Xh); t_up = time.perf_counter() - t0
    glm = ad.glm.gaussian(y)
    t0 = time.perf_counter(); st = ad.grpnet(Xd, glm, early_exit=False); t_path = time.perf_counter() - t0
    print("upload %.3f s (%.1f GB/s)  path %.3f s  -> PCIe-inclusive %.2f paths/s (first path), %.2f paths/s resident" % (t_up, 8.0 / t_up, t_path, 1 / (t_up + t_path), 1 / t_path), flush=True)
    del Xd
