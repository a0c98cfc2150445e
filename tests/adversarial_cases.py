"""Adversarial series / initial states for the carried Newton-Schulz starts of the blocked engine (test data only).

Every parity series elsewhere in tests/ is the smooth synthetic generator; the r x r inversions of the blocked engine start
each timestep from the previous step's inverse, corrected by a rank-2 + scalar predictor.  These cases break the
assumptions behind that start -- an abrupt change of the data, a handful of huge innovations, a process noise so small that
the two inverses nearly coincide, a nearly singular prior covariance -- so that either the start still converges or the
fallback (direct symmetric sweep) has to fire; in both cases the result must match the oracle to the stated tolerance
(pypsmf/psmf/psmf.py:85-102,140-165 is what is being reproduced).
"""

import numpy as np

from oracle import psmf_oracle as O

CASES = ("level_shift", "outlier_block", "tiny_Q", "tiny_P0", "outliers_then_quiet")


def make_case(name, d, r, T, robust, seed=4711):
    """-> dict(Y (T, d) float32, C0, V0, P0, Q, checkpoints)."""
    Y = O.synthetic_series(d, r, T, seed + (7 if robust else 0), noise="t" if robust else "normal", dtype=np.float64)
    rng = np.random.default_rng(seed + 99)
    C0 = (0.1 * rng.standard_normal((d, r))).astype(np.float32).astype(np.float64)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    sigma = float(np.std(Y[: min(T, 200)]))
    half = T // 2
    cps = [half - 1, half + 1, half + 40, T]
    if name == "level_shift":
        # from k = T/2 on every series jumps by its own offset of a few standard deviations
        Y[half:] += 4.0 * sigma * rng.standard_normal(d)[None, :]
    elif name == "outlier_block":
        # three consecutive timesteps in which a tenth of the rows read 1000 sigma
        rows = rng.choice(d, size=d // 10, replace=False)
        Y[half:half + 3, rows] += 1e3 * sigma
    elif name == "outliers_then_quiet":
        # one timestep of 1000 sigma on ALL rows, then the series goes flat (constant): innovations collapse
        Y[half] += 1e3 * sigma
        Y[half + 1:] = Y[half - 1][None, :]
    elif name == "tiny_Q":
        Q = 1e-8 * np.eye(r)
        cps = [40, 300, half, T]
    elif name == "tiny_P0":
        P0 = 1e-6 * np.eye(r)
        cps = [1, 40, half, T]
    else:
        raise KeyError(name)
    return dict(Y=Y.astype(np.float32), C0=C0, V0=V0, P0=P0, Q=Q, checkpoints=tuple(cps), event=half)
