import sys, hashlib, numpy as np
sys.path.insert(0, "/root/repo")
import adelie_amd as ad
rng = np.random.RandomState(5)
n, p = 20037, 600
cd = np.asfortranarray(rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8))
imp = np.array([cd[:, j][cd[:, j] >= 0].mean() for j in range(p)])
Xv = np.where(cd < 0, imp[None, :], cd).astype(float)
beta = np.zeros(p); beta[rng.choice(p, 200, replace=False)] = rng.normal(size=200) * 0.5
eta = Xv @ beta; eta = (eta - eta.mean()) / eta.std()
y = (rng.uniform(size=n) < 1 / (1 + np.exp(-2 * eta))).astype(float)
yg = eta + 0.3 * rng.normal(size=n)
X = ad.matrix.snp_calldata(cd, imp)
for name, glm, kw in [("binomial", ad.glm.binomial(y), dict(min_ratio=0.03, lmda_path_size=14)),
                      ("gaussian", ad.glm.gaussian(yg), dict(min_ratio=0.01, lmda_path_size=20))]:
    st = ad.grpnet(X, glm, early_exit=False, progress_bar=False, **kw)
    h = hashlib.sha1(st.betas.toarray().tobytes() + st.intercepts.tobytes()).hexdigest()[:16]
    print(name, h, st.counters["n_panel_blocks"], st.error)
Xd = ad.matrix.dense(np.asfortranarray(rng.normal(size=(3000, 900))))
yd = rng.normal(size=3000)
st = ad.grpnet(Xd, ad.glm.gaussian(yd), early_exit=False, progress_bar=False, min_ratio=0.05, lmda_path_size=15)
print("dense", hashlib.sha1(st.betas.toarray().tobytes()).hexdigest()[:16], st.counters["n_panel_blocks"])
Xg = ad.matrix.dense(np.asfortranarray(rng.normal(size=(4000, 1200))))
yg2 = rng.normal(size=4000)
st = ad.grpnet(Xg, ad.glm.gaussian(yg2), groups=np.arange(0, 1200, 10), alpha=0.5, early_exit=False, progress_bar=False, min_ratio=0.05,
               lmda_path_size=15)
print("groups10", hashlib.sha1(st.betas.toarray().tobytes()).hexdigest()[:16], st.counters["n_panel_blocks"])
Ym = rng.normal(size=(4000, 3)) + np.asarray(Xg.cmul(0, np.ones(4000)) if False else 0)
st = ad.grpnet(Xg, ad.glm.multigaussian(Ym), early_exit=False, progress_bar=False, min_ratio=0.2, lmda_path_size=12)
print("multi3", hashlib.sha1(st.betas.toarray().tobytes()).hexdigest()[:16], st.counters["n_panel_blocks"])
