"""Multi-response (multigaussian) path over a 2-bit SNP base next to the same path over its densified copy:
python scripts/bench_multi_snp.py [n p K].  Prints wall, device memory of the two designs, max|dbeta|."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from adelie_amd import matrix

n, p, K = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (100000, 4000, 4)))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
u = torch.rand((p, n), device=dev, generator=g)
calls = ((u > 0.5).to(torch.int8) + (u > 0.8).to(torch.int8) + (u > 0.95).to(torch.int8)).t()   # (n, p) column-major view
calls = torch.where(calls == 3, torch.tensor(-9, dtype=torch.int8, device=dev), calls)
del u
Xs = matrix.snp_calldata(calls)
ok = calls >= 0   # column mean of the non-missing calls (io/utils.hpp:10-31)
imp = (torch.where(ok, calls, torch.zeros_like(calls)).to(torch.float64).sum(0) / ok.sum(0).clamp(min=1).to(torch.float64))
D = torch.where(calls < 0, imp[None, :].to(torch.float64), calls.to(torch.float64))
B = torch.zeros((p, K), dtype=torch.float64, device=dev)
idx = torch.randperm(p, device=dev, generator=g)[:30]
B[idx] = torch.randn((30, K), dtype=torch.float64, device=dev, generator=g)
Y = ((D - D.mean(0)) @ B + torch.randn((n, K), dtype=torch.float64, device=dev, generator=g)).cpu().numpy()
kw = dict(early_exit=False, lmda_path_size=50, min_ratio=0.05, progress_bar=False)
def run(X):
    ad.grpnet(X, ad.glm.multigaussian(y=Y), **dict(kw, lmda_path_size=3))
    torch.cuda.synchronize(); t = time.perf_counter()
    st = ad.grpnet(X, ad.glm.multigaussian(y=Y), **kw)
    torch.cuda.synchronize(); return st, time.perf_counter() - t
s1, t1 = run(Xs)
Dd = D.t().contiguous().t()          # column-major (n, p)
Xd = matrix.dense(Dd)
s2, t2 = run(Xd)
print("n=%d p=%d K=%d | 2-bit base %.3f s (%.1f MB) | dense base %.3f s (%.1f MB) | max|dbeta| %.2e | nnz last %d" % (
    n, p, K, t1, n * p / 4 / 1e6, t2, n * p * 8 / 1e6, np.abs(s1.betas.toarray() - s2.betas.toarray()).max(),
    (s1.betas[-1].toarray() != 0).sum()))
