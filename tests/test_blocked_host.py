"""Exact time-blocking (host model of the blocked device engine) == step-by-step CPU oracle.  CPU only."""

import numpy as np
import pytest

from conftest import relerr
from oracle import psmf_oracle as O
from host_models import blocked_epoch_host, blocked_pipelined_epoch_host


@pytest.mark.parametrize("robust", [False, True])
@pytest.mark.parametrize("B", [1, 7, 32])
def test_blocked_equals_stepwise(robust, B):
    rng = np.random.default_rng(2)
    d, r, T = 300, 6, 75       # T not a multiple of B: ragged last block
    Y = O.synthetic_series(d, r, T, 11, noise="t" if robust else "normal", dtype=np.float64)
    C0 = 0.1 * rng.standard_normal((d, r))
    st = O.State(C=C0, V=0.1 * np.eye(r), mu=np.zeros(r), P=np.eye(r), Q=0.1 * np.eye(r), rho=1.0, lam=1.8)
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=robust), O.RandomWalkDyn())
    C, V, P, mu, rho, lam, Yp2 = blocked_epoch_host(C0, Y, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), 1.0,
                                                    B=B, robust=robust, lambda0=1.8)
    for a, b in ((C, st.C), (V, st.V), (P, st.P), (mu, st.mu), (Yp2, Yp)):
        assert relerr(a, b) < 1e-9
    if robust:
        assert relerr(rho, st.rho) < 1e-10 and lam == st.lam


@pytest.mark.parametrize("robust", [False, True])
@pytest.mark.parametrize("B", [5, 32])
def test_pipelined_assembly_equals_stepwise(robust, B):
    """K of block b + 1 assembled from the cross-Gram and the tracked Gram (what the pipelined device engine does) instead
    of recomputed from C: the same recursion to round-off."""
    rng = np.random.default_rng(3)
    d, r, T = 257, 6, 83
    Y = O.synthetic_series(d, r, T, 12, noise="t" if robust else "normal", dtype=np.float64)
    C0 = 0.1 * rng.standard_normal((d, r))
    st = O.State(C=C0, V=0.1 * np.eye(r), mu=np.zeros(r), P=np.eye(r), Q=0.1 * np.eye(r), rho=1.0, lam=1.8)
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=robust), O.RandomWalkDyn())
    C, V, P, mu, rho, lam, Yp2 = blocked_pipelined_epoch_host(C0, Y, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), 1.0,
                                                              B=B, robust=robust, lambda0=1.8)
    for a, b in ((C, st.C), (V, st.V), (P, st.P), (mu, st.mu), (Yp2, Yp)):
        assert relerr(a, b) < 1e-9
