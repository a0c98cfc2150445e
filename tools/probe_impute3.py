"""GPU-box probe: the masked small-shape kernel (psmf_impute_kernel3) against the CPU oracle by number of columns, 1 pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import impute
from oracle.impute_oracle import impute_filter

rng = np.random.default_rng(4)
d, r = 19, int(os.environ.get("R", 10))
for n in (2, 3, 5, 10, 40, 400):
    Yorig = np.cumsum(0.3 * rng.standard_normal((d, n)), axis=1)
    M = (rng.random((d, n)) > 0.4).astype(int)
    Mmiss = ((1 - M)).astype(float)
    C0, X0 = rng.random((d, r)), rng.random((r, n))
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    for robust in (False, True):
        ep, ef, ib, st = impute_filter(Yorig * M, C0, X0.copy(), M, Mmiss, V, Q, 10.0, P, 2, 1, Yorig, 0.0, robust=robust, lambda0=1.8, return_state=True)
        try:
            res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2, 1, robust=robust, lambda0=1.8, want_bands=True)
            ex = np.abs(res["X"][0] - st["X"]).max(axis=0)
            print(n, robust, "X err by column:", np.array2string(ex[:8], precision=2), "max", ex.max(), "C", np.abs(res["C"][0] - st["C"]).max())
        except Exception as e:
            print(n, robust, "FAILED", e)
