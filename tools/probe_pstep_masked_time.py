"""GPU-box probe: us per masked timestep of the persistent per-step kernel at one shape (per-phase stamps with a -DPSTEP_PROF build and
PSMF_PSTEP_PROF=1).   python tools/probe_pstep_masked_time.py d r T [storage]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi as c

d, r, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
storage = sys.argv[4] if len(sys.argv) > 4 else "f64"
rng = np.random.default_rng(0)
Y = rng.standard_normal((T, d), dtype=np.float32)
M = (rng.random((T, d)) > 0.4).astype(np.uint8)
f = c.DeviceFilter(d, r, storage=storage, engine="step", masked=True)
f.upload_series(Y)
f.upload_mask(M)
f.set_state(0.1 * rng.standard_normal((d, r)), 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
f.run(0, min(T, 200))
ms = min(f.run_timed(0, T) for _ in range(3))
print(json.dumps(dict(d=d, r=r, T=T, storage=storage, masked=True, us_per_step=1e3 * ms / T, kern=f.geometry()["filter_kernel"])), flush=True)
f.close()
