"""GPU-box probe: the regime of the cos-phase full filter (eigenvalues of P, G, V; kappa) after a few passes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench
np.set_printoptions(precision=3, linewidth=200)
d, T, r = 20000, 2000, 20
ser = bench.Series(d, r, T, 4711, 0, d, False)
st0 = bench.init_state(d, r, 4711)
chunks = list(ser.chunks())
for kind, truth in ((_capi.DYN_COS_PHASE, False), (_capi.DYN_COS_PHASE, True), (_capi.DYN_RANDOM_WALK, False)):
    f = _capi.DeviceFilter(d, r, storage="f32", dyn_kind=kind)
    for a, Yc in chunks:
        f.upload_series(Yc, t0=a, T_total=T)
    theta = None
    if kind != _capi.DYN_RANDOM_WALK:
        theta = 1e-3 * np.arange(1, r + 1) if truth else 0.05 + 0.1 * np.random.default_rng(3).random(r)
    f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"], theta=theta)
    for i in range(2):
        f.run(0, T)
    s = f.get_state()
    G = s["C"].T @ s["C"]
    print("kind", kind, "truth", truth, f.geometry()["filter_kernel"])
    print("  eig P", np.linalg.eigvalsh(s["P"]))
    print("  eig G", np.linalg.eigvalsh(G))
    print("  eig V", np.linalg.eigvalsh(s["V"]))
    print("  s, eta, N", s["s"], s["eta"], s["N"], " |mu|", np.abs(s["mu"]).max())
    f.close()
