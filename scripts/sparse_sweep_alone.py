"""Full sweep of a sparse-resident design through adelie_hip_bench_sweep (HIP events around `reps` launches, no host copies).
    python scripts/sparse_sweep_alone.py [n p density]"""
import ctypes as C
import json
import sys

import numpy as np
import scipy.sparse as sp

import adelie_amd as ad
from adelie_amd import _abi

n, p, dens = (int(float(sys.argv[1])), int(float(sys.argv[2])), float(sys.argv[3])) if len(sys.argv) > 3 else (1_000_000, 100_000, 1e-3)
rng = np.random.default_rng(0)
nnz = int(n * p * dens)
M = sp.csc_matrix((rng.normal(size=nnz), (rng.integers(0, n, size=nnz), rng.integers(0, p, size=nnz))), shape=(n, p))
M.sum_duplicates()
M.sort_indices()
b = _abi.hip_backend()
X = ad.matrix.sparse(M, resident="csc")
ms = C.c_double()
b.check(b.fn("bench_sweep")(X._handle, 20, C.byref(ms)))
print(json.dumps({"sweep_ms": ms.value, "nnz": int(M.nnz), "algorithmic_GBps": (M.nnz * 12 + n * 8) / ms.value / 1e6}))
