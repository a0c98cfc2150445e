"""Per-step time of the masked per-step engine (cfg.masked) and of the plain per-step engine at a few shapes; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.  usage: python tools/probe_masked.py [steps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rpsmf_amd import _capi as c

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(0)
for (d, r, masked, storage) in [(20000, 10, True, "f64"), (100000, 32, True, "f32"), (100000, 32, False, "f32"), (20000, 40, False, "f64"), (20000, 40, True, "f64")]:
    Y = rng.standard_normal((T, d)).astype(np.float32)
    M = (rng.random((T, d)) > 0.4).astype(np.uint8)
    f = c.DeviceFilter(d, r, storage=storage, masked=masked, engine="step")
    f.upload_series(Y)
    if masked:
        f.upload_mask(M)
    f.set_state(0.1 * rng.standard_normal((d, r)), 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
    f.run(0, T)
    t0 = time.perf_counter()
    for _ in range(3):
        f.run(0, T, sync=False)
    f.sync()
    dt = (time.perf_counter() - t0) / (3 * T)
    print(f"d={d} r={r} masked={masked} {storage}: {1e6 * dt:.1f} us per timestep, {1 / dt:.0f} timesteps/s", flush=True)
    f.close()
