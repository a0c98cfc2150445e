"""Masked per-step engine, few steps, no graph: for rocprofv3 --kernel-trace --stats."""
import sys
import numpy as np
sys.path.insert(0, ".")
from rpsmf_amd import _capi as c
T = 96
rng = np.random.default_rng(0)
for (d, r, st) in [(100000, 32, "f32"), (20000, 10, "f64")]:
    Y = rng.standard_normal((T, d)).astype(np.float32)
    f = c.DeviceFilter(d, r, storage=st, masked=True, engine="step", use_graph=False)
    f.upload_series(Y)
    f.upload_mask((rng.random((T, d)) > 0.4).astype(np.uint8))
    f.set_state(0.1 * rng.standard_normal((d, r)), 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
    f.run(0, T)
    f.run(0, T)
    f.close()
