import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
n, p = 100000, 10000
g = torch.Generator(device="cuda"); g.manual_seed(0)
Xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64); X = Xt.t()
rng = np.random.RandomState(0)
beta = rng.normal(size=p) * (rng.uniform(size=p) < 0.05)
y = (X @ torch.from_numpy(beta).cuda()).cpu().numpy() + np.sqrt(beta @ beta) * rng.normal(size=n)
Xd = ad.matrix.dense(X)
for eng in ["panel", "gram"]:
    os.environ["ADELIE_HIP_CD_ENGINE"] = eng
    st = ad.grpnet(Xd, ad.glm.gaussian(y), early_exit=False)
    w = np.full(n, 1 / n); yc = y - y.mean()
    for l in [5, 20, 50, 99]:
        b = st.betas[l].toarray().ravel()
        r = yc - (Xd @ b) - (st.intercepts[l] - y.mean())
        grad = Xd.T @ (w * r)
        lm = st.lmdas[l]
        zero = b == 0
        viol = (np.abs(grad[zero]) - lm).max()
        act = np.abs(np.abs(grad[~zero]) - lm).max() if (~zero).any() else 0
        print(eng, l, "lm %.5g nnz %d max zero-viol %.3e (rel %.2e) active dev %.3e  sum wr %.1e y_var %.3g" % (lm, (~zero).sum(), viol, viol / lm, act, np.sum(w * r), st.y_var))
