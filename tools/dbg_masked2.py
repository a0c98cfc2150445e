import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi as c
lib = c.load_library()
lib.psmf_debug_read.restype = C.c_int
lib.psmf_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
d, r, T = int(sys.argv[1]), int(sys.argv[2]), 1
rng = np.random.default_rng(3)
Y = rng.standard_normal((3, d)); M = (rng.random((3, d)) > 0.4).astype(np.uint8); Y = Y * M
C0 = 0.1 * rng.standard_normal((d, r))
f = c.DeviceFilter(d, r, storage="f64", engine="step", masked=True)
f.upload_series(Y); f.upload_mask(M)
f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
print(f.geometry()["filter_kernel"])
f.run(0, 1)
s = f.get_state()
G = np.zeros(r * r)
lib.psmf_debug_read(f._h, G.ctypes.data_as(C.POINTER(C.c_double)), r * r)
G = G.reshape(r, r)
Cn = s["C"]
Gref = (Cn * M[1][:, None].astype(float)).T @ Cn
err = np.abs(G - Gref)
print("max err", err.max(), "max ref", np.abs(Gref).max())
np.set_printoptions(linewidth=250, precision=1)
blk = lambda a, b: err[16 * a:16 * a + 16, 16 * b:16 * b + 16].max() if r > 16 * max(a, b) else -1
print("block errs", [[float("%.1e" % blk(a, b)) for b in range(2)] for a in range(2)])
bad = np.argwhere(err > 1e-9 * np.abs(Gref).max())
print("bad entries", len(bad), bad[:20].tolist())
if len(bad):
    i, j = bad[0]
    print("G", G[i, j], "ref", Gref[i, j], "Gref^T?", Gref[j, i])
    # is G[i,j] equal to some other ref entry?
    k = np.argwhere(np.abs(Gref - G[i, j]) < 1e-9)
    print("value found at", k.tolist()[:5])
f.close()
