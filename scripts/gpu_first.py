import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import adelie_amd as ad
from oracle import oracle
np.random.seed(1)
n,p=257,37
X=np.asfortranarray(np.random.randn(n,p)); v=np.random.randn(n); w=np.random.rand(n)
Xg=ad.matrix.dense(X)
out=np.empty(p); Xg.mul(v,w,out); print("mul err", np.abs(out-X.T@(v*w)).max())
print("cmul err", abs(Xg.cmul(5,v,w)-X[:,5]@(v*w)))
o=np.zeros(n); Xg.btmul(3,4,np.arange(4.)+1,o); print("btmul err", np.abs(o-X[:,3:7]@(np.arange(4.)+1)).max())
C=np.empty((6,6)); Xg.cov(10,6,np.sqrt(w),C); print("cov err", np.abs(C-(X[:,10:16].T*w)@X[:,10:16]).max())
C=np.empty((37,37)); Xg.cov(0,37,np.sqrt(w),C); print("cov37 err", np.abs(C-(X.T*w)@X).max())
import __graft_entry__ as g
g.smoke()
