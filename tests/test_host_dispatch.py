"""Which kind of device handle the drop-in classes ask for -- decided on the host before any device exists, so checkable without a
GPU: the closed-form dynamics run inside the device loop of either engine (round 5: the per-step engine evaluates them too), a
Q[k] schedule of matrices is uploaded as such, everything else is host-stepped.  Reference: pypsmf/psmf/psmf.py:104-115
(f and Q[k] are read every step), nonlinearities.py:42-150."""

import numpy as np

import rpsmf_amd as psmf
from rpsmf_amd import _capi
from rpsmf_amd import nonlinearities as NL


def _iter(nl, r, d=30, T=6, Qs=None, Rs=None, **kw):
    rng = np.random.default_rng(0)
    n = getattr(nl, "n_params", 0)
    Qs = Qs if Qs is not None else {k: 0.1 * np.eye(r) for k in range(T + 1)}
    Rs = Rs if Rs is not None else {k: 1.0 for k in range(T + 1)}
    return psmf.PSMFIter(0.1 * rng.random((n, 1)), rng.standard_normal((d, r)), 0.1 * np.eye(r), np.zeros((r, 1)), np.eye(r), Qs, Rs, nl, **kw)


def test_closed_form_kinds_are_device_kinds_at_any_rank_and_engine():
    for nl, r, kw in ((NL.Sinusoid(40), 40, {}), (NL.FourierBasis(64, N=1), 64, {}), (NL.ScaledWalk(12, bias=True), 12, dict(engine="step")),
                      (NL.Sinusoid(8, scaled=False), 8, dict(engine="step"))):
        f = _iter(nl, r, **kw)
        assert not f._host_stepped()
        k = f._device_kwargs()
        assert k["dyn_kind"] == nl.device_kind and k["dyn_kind"] != _capi.DYN_HOST
    f = _iter(NL.Sinusoid(6), 6, Rs={k: np.linspace(0.5, 2.0, 30) for k in range(7)})       # non-uniform diagonal R: per-step engine
    k = f._device_kwargs()
    assert k["nonuniform_R"] and k["engine"] == "step" and k["dyn_kind"] == _capi.DYN_SINUSOID


def test_what_stays_host_stepped():
    f = _iter(NL.FourierBasis(3, N=5), 3)                   # more than 4 + 4 terms
    assert f._host_stepped() and f._device_kwargs()["dyn_kind"] == _capi.DYN_HOST
    g = _iter(lambda theta, x, t: np.tanh(x), 3, recognise=False)        # an arbitrary callable
    assert g._host_stepped()


def test_q_schedule_classification_and_matrices():
    r, T = 4, 6
    # scalar multiples of Q[1]: a scalar schedule
    f = _iter(psmf.RandomWalk(), r, T=T, Qs={k: (0.1 + 0.01 * k) * np.eye(r) for k in range(T + 1)})
    rho, Q1, rho_s, q_s = f._device_rho_q(T)
    assert rho_s is None and np.allclose(q_s[1:], [(0.1 + 0.01 * k) / 0.11 for k in range(1, T + 1)])
    # not multiples: matrices, entry k = Q[k], entry 0 = Q[1] (unused by the device)
    Qs = {k: 0.1 * np.eye(r) + 0.01 * (k % 3) * np.ones((r, r)) for k in range(T + 1)}
    g = _iter(psmf.RandomWalk(), r, T=T, Qs=Qs)
    assert g._device_rho_q(T)[3] == "host"
    Qm = g._q_matrices(T)
    assert Qm.shape == (T + 1, r, r) and all(np.array_equal(Qm[k], Qs[k]) for k in range(1, T + 1)) and np.array_equal(Qm[0], Qs[1])
    g._q_matrix_sched = True
    assert g._device_kwargs()["engine"] == "step" and not g._host_stepped()
