"""Duration of the per-step engine's solve block alone: d = 512 (one row-sweep workgroup), so psmf_sweep_solve ~ its block 0.
Run under rocprofv3 --kernel-trace; PSMF_STEP_WAVE_SOLVE=0 selects the LDS sweeps."""
import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from rpsmf_amd import _capi as c
T = 64
rng = np.random.default_rng(0)
for r in (32, 20, 16):
    for Qiso in (True, False):
        d = 512
        Y = rng.standard_normal((T, d)).astype(np.float32)
        f = c.DeviceFilter(d, r, storage="f32", engine="step", use_graph=False)
        f.upload_series(Y)
        B = rng.standard_normal((r, r))
        Q = 0.1 * np.eye(r) + (0.0 if Qiso else 0.01 * B @ B.T)
        f.set_state(0.1 * rng.standard_normal((d, r)), 0.1 * np.eye(r), np.eye(r), Q, np.zeros(r), rho=1.0, lambda0=1.8)
        f.run(0, T)
        t0 = time.perf_counter(); f.run(0, T); dt = (time.perf_counter() - t0) / T
        print(f"r={r} Q=qI:{Qiso} wave_solve={os.environ.get('PSMF_STEP_WAVE_SOLVE','1')}: {1e6*dt:.1f} us/step", flush=True)
        f.close()
