import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
n, p = 100000, 10000
X, y = make_data(n, p, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
glm = ad.glm.gaussian(y)
st = ad.grpnet(Xd, glm, early_exit=False)
st = ad.grpnet(Xd, glm, early_exit=False)
t = (st.benchmark_fit_active + st.benchmark_fit_screen) * 1e3
print("total cd ms", t.sum(), "n_fits", len(t), st.counters)
for i in range(0, 100, 6):
    print(i, "active", st.active_sizes[i], "screen", st.screen_sizes[i], "fit ms %.3f" % t[min(i+1, len(t)-1)])
