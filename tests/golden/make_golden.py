"""Generates the golden vectors in this directory.

The reference itself cannot run in this environment (its extension needs Eigen), so the vectors are
produced by the CPU oracle (oracle/grpnet_oracle.cpp) AFTER it passed tests/test_oracle_solver.py (known answers of
the reference's notebooks, scikit-learn, KKT).  They pin both the oracle (regression) and the HIP path (parity).
Inputs are regenerated from the recorded seeds by `golden_cases.make_case`; only outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import adelie_amd as ad  # noqa: E402
from golden_cases import CASES, make_case, solve_case  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    for name in CASES:
        case = make_case(name)
        X = oracle.snp_calldata(case["calldata"]) if "calldata" in case else oracle.dense(case["X"])
        st = solve_case(ad, X, case)
        assert st.error == "", (name, st.error)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            betas_data=st.betas.data, betas_indices=st.betas.indices, betas_indptr=st.betas.indptr,
            lmdas=st.lmdas, devs=st.devs, intercepts=st.intercepts, screen_sizes=st.screen_sizes,
            active_sizes=st.active_sizes, lmda_max=st.lmda_max,
        )
        print(name, "solutions", len(st.lmdas), "dev", float(st.devs[-1]), "active", int(st.active_sizes[-1]))


if __name__ == "__main__":
    main()
