"""The per-step engine on configurations drawn at random (seeded): rank 1 ... 64 (every solve path: one 16 x 16 tile, 2 x 2, 3 x 3,
4 x 4 tiles; LDS sweeps when the switches say so), PSMF / rPSMF, Q = q I (the two inversions side by side) or a general Q (one after
the other), uniform or non-uniform diagonal R, float64 or float32 storage, a run cut at a random step (the second part starts on the
carried state) -- against the float64 oracle.  The round that built these paths changed the solve block, the serial stage and the
row sweep's geometry for r > 32 within a day; this is the net under them.  GPU only: `pytest -m gpu`.
Reference: pypsmf/psmf/psmf.py:85-165, rpsmf.py:116-171."""

import numpy as np
import pytest

from conftest import relerr
from oracle import psmf_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

N_CASES = 40


def _case(i):
    rng = np.random.default_rng(9000 + i)
    r = int(rng.choice([1, 2, 3, 5, 8, 9, 15, 16, 17, 24, 31, 32, 33, 37, 40, 47, 48, 49, 56, 63, 64]))
    d = int(rng.integers(max(r + 3, 20), 2500))
    T = int(rng.integers(12, 36))
    return dict(i=i, r=r, d=d, T=T, cut=int(rng.integers(1, T)), robust=bool(rng.integers(0, 2)), general_Q=bool(rng.random() < 0.3),
                nonuniform=bool(rng.random() < 0.25), storage="f32" if rng.random() < 0.3 else "f64", seed=int(rng.integers(1 << 30)))


@pytest.mark.parametrize("i", range(N_CASES))
def test_step_engine_random_configuration(i):
    from rpsmf_amd import _capi as c

    cs = _case(i)
    r, d, T, robust = cs["r"], cs["d"], cs["T"], cs["robust"]
    rng = np.random.default_rng(cs["seed"])
    Ct = rng.standard_normal((d, r))
    x = rng.standard_normal(r)
    Y = np.empty((T, d))
    for t in range(T):
        x = x + 0.1 * rng.standard_normal(r)
        Y[t] = Ct @ x + 0.3 * (rng.standard_t(3.0, d) if robust else rng.standard_normal(d))
    C0 = 0.1 * rng.standard_normal((d, r))
    if cs["storage"] == "f32":          # the device stores C and y in float32: start both sides from representable values
        Y = Y.astype(np.float32).astype(np.float64)
        C0 = C0.astype(np.float32).astype(np.float64)
    V0, P0 = 0.1 * np.eye(r), np.eye(r)
    A = rng.standard_normal((r, r)) / np.sqrt(r)
    Q = 0.1 * np.eye(r) + (0.05 * (A @ A.T) if cs["general_Q"] else 0.0)
    rho = 0.3 + 2.0 * rng.random(d) if cs["nonuniform"] else 1.0
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=rho, lam=1.8)
    f = c.DeviceFilter(d, r, storage=cs["storage"], robust=robust, engine="step", nonuniform_R=cs["nonuniform"])
    if cs["nonuniform"]:
        f.set_row_noise(rho)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
    tol = 1e-9 if cs["storage"] == "f64" else 1e-5
    for a, b in ((0, cs["cut"]), (cs["cut"], T)):
        st, Yp, _ = O.run_epoch(st, Y[a:b], O.Mode(robust=robust), O.RandomWalkDyn(), k0=a, want_grad=False)
        f.run(a, b)
        s = f.get_state()
        for n in ("C", "V", "mu", "P"):
            assert relerr(s[n], getattr(st, n)) < tol, (cs, n, b, relerr(s[n], getattr(st, n)))
        assert relerr(f.y_pred(a, b - a), Yp) < tol, (cs, "y_pred", b)
    f.close()
