import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (needs a build with AHIP_EXTRA_FLAGS=-DAHIP_GRP_PROFILE: the counters are compiled in, no environment variable)
import adelie_amd as ad
from adelie_amd import _abi
import adelie_amd.state as S_
_abi.S.update({f"dbg{i}": 900 + i for i in range(8)})
S_._TIMERS = list(S_._TIMERS) + [f"dbg{i}" for i in range(8)]
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X); glm = ad.glm.gaussian(y)
st = ad.grpnet(Xd, glm, groups=np.arange(0, 10000, 10), alpha=0.5, early_exit=False)
names = ["prologue", "loads+norm", "newton", "change test", "mark+gupdate", "epilogue", "-", "blocks"]  # grp_solve_body_rot's marks (ADELIE_HIP_GROUP_ROT=0: prologue*, rotation, norm+newton, changed+backrot, mark+gupdate, epilogue)
tot = sum(st.timers[f"dbg{i}"] for i in range(6))
nb = st.timers["dbg7"]
print("blocks", nb, "updates", st.counters["n_updates"], "cd ms", st.timers["t_cd_ms"])
for i in range(6):
    c = st.timers[f"dbg{i}"]
    print("%-18s %6.1f%%  %8.0f cycles/block  %.2f us/block" % (names[i], 100 * c / tot, c / nb, c / nb / 2400))
