"""The full 2-bit sweep of config 4's design on an otherwise idle chip (X.mul: two 4 MB host vectors in, 50k values out, per
call; the copies are ~0.5 ms of it) next to its duration inside a path (bench.py --config 4, roofline_sweep)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_snp_data

n, p = (int(a) for a in sys.argv[1:3]) if len(sys.argv) >= 3 else (500000, 50000)
dev = torch.device("cuda", 0)
cd, imp, y = make_snp_data(n, p, 0, dev)
X = ad.matrix.snp_calldata(cd, imp)
del cd
rng = np.random.default_rng(0)
v = rng.standard_normal(n); w = rng.uniform(0.5, 1.5, n); out = np.empty(p)
for _ in range(3): X.mul(v, w, out)
torch.cuda.synchronize(); t = time.perf_counter()
K = 20
for _ in range(K): X.mul(v, w, out)
torch.cuda.synchronize(); el = (time.perf_counter() - t) / K
print("X.mul on %dx%d 2-bit: %.3f ms per call (incl. the host copies) = %.2f TB/s of calls" % (n, p, el * 1e3, n * p / 4 / el / 1e12))
