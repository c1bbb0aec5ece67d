"""Cycle profile of the group solve on a multi-response view (needs a build with AHIP_EXTRA_FLAGS=-DAHIP_GRP_PROFILE)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (needs a build with AHIP_EXTRA_FLAGS=-DAHIP_GRP_PROFILE: the counters are compiled in, no environment variable)
import adelie_amd as ad
from adelie_amd import _abi
import adelie_amd.state as S_
_abi.S.update({f"dbg{i}": 900 + i for i in range(8)})
S_._TIMERS = list(S_._TIMERS) + [f"dbg{i}" for i in range(8)]
n, p, K = 100000, 4000, int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator(device="cuda").manual_seed(0)
Xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64).T
rng = np.random.RandomState(0)
B = np.zeros((p, K)); nz = rng.choice(p, p // 20, replace=False); B[nz] = rng.normal(size=(len(nz), K))
Y = (Xt @ torch.from_numpy(B).cuda()).cpu().numpy(); Y += np.sqrt(np.mean(Y ** 2)) * rng.normal(size=(n, K))
st = ad.grpnet(X=ad.matrix.dense(Xt), glm=ad.glm.multigaussian(Y), early_exit=False)
names = ["prologue*", "rotation", "norm+newton", "changed+backrot", "mark+gupdate", "epilogue", "-", "blocks"]
tot = sum(st.timers[f"dbg{i}"] for i in range(6)); nb = st.timers["dbg7"]
print("K", K, "blocks", nb, "updates", st.counters["n_updates"], "cd ms", st.timers["t_cd_ms"])
for i in range(6):
    c = st.timers[f"dbg{i}"]
    print("%-18s %6.1f%%  %8.0f cycles/block  %.2f us/block" % (names[i], 100 * c / tot, c / nb, c / nb / 2400))
