"""Per-step engine at r = 40 / 64 (and masked), few steps: for rocprofv3 --kernel-trace --stats."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rpsmf_amd import _capi as c
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
for (d, r, masked) in [(20000, 40, False), (100000, 32, False), (100000, 32, True), (100000, 16, False), (100000, 48, False)]:
    st = "f64" if r > 32 else "f32"
    Y = rng.standard_normal((T, d)).astype(np.float32)
    f = c.DeviceFilter(d, r, storage=st, masked=masked, engine="step", use_graph=False)
    f.upload_series(Y)
    if masked:
        f.upload_mask((rng.random((T, d)) > 0.4).astype(np.uint8))
    f.set_state(0.1 * rng.standard_normal((d, r)), 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
    f.run(0, T)
    t0 = time.perf_counter(); f.run(0, T); dt = (time.perf_counter() - t0) / T
    print(f"d={d} r={r} masked={masked}: {1e6*dt:.1f} us/step", flush=True)
    f.close()
    if not masked:
        f = c.DeviceFilter(d, r, storage=st, engine="step", use_graph=False, coef_update=False)
        f.upload_series(Y)
        f.set_state(0.1 * rng.standard_normal((d, r)), 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
        f.run(0, T)
        t0 = time.perf_counter(); f.run(0, T); dt = (time.perf_counter() - t0) / T
        print(f"d={d} r={r} NO coefficient update (no solve block): {1e6*dt:.1f} us/step", flush=True)
        f.close()
