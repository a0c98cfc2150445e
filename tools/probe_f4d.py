"""GPU-box probe: per-step Newton-Schulz residuals of the last block of a filter4 run.  Needs the debug library:
   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DF4_DEBUG -I include -o tools/bin/libpsmf_dbg.so rpsmf_amd/csrc/psmf_capi.hip -L/opt/rocm/lib -lrccl"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rpsmf_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libpsmf_dbg.so")
import bench
d, T, r = 20000, 440, 20
q = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
ser = bench.Series(d, r, T, 4711, 0, d, False)
st0 = bench.init_state(d, r, 4711)
f = _capi.DeviceFilter(d, r, storage="f32", dyn_kind=_capi.DYN_COS_PHASE)
for a, Yc in ser.chunks():
    f.upload_series(Yc, t0=a, T_total=T)
theta = 0.05 + 0.1 * np.random.default_rng(3).random(r)
f.set_state(st0["C"], st0["V"], st0["P"], q * np.eye(r), st0["mu"], rho=st0["rho"], lambda0=st0["lam"], theta=theta)
for i in range(5):
    f.counters(reset=True)
    f.run(0, T)
    print(f.counters())
out = np.zeros(48 * 8)
lib = _capi.load_library()
lib.psmf_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
lib.psmf_debug_read(f._h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size)
print("jb: |R_Y|^2 try_y ydone yit | |R_X|^2 try_x xdone xit")
for jb in range(44):
    v = out[jb * 8:jb * 8 + 8]
    print(f"{jb:2d}: {v[0]:.3e} {int(v[1])} {int(v[2])} {int(v[3])} | {v[4]:.3e} {int(v[5])} {int(v[6])} {int(v[7])}")
