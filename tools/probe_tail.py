"""GPU-box probe: the launched per-step engine with and without the sweep's tail reduction (PSMF_TAIL_REDUCE), graph replay."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpsmf_amd import _capi as c
T = 512
rng = np.random.default_rng(0)
os.environ["PSMF_STEP_PERSISTENT"] = "0"
for (d, r, kw) in [(20000, 40, {}), (20000, 64, {}), (100000, 32, {}), (100000, 48, {}), (100000, 32, dict(nonuniform_R=True)), (5000, 40, {})]:
    st = "f64" if r > 32 else "f32"
    Y = rng.standard_normal((T, d)).astype(np.float32)
    C0 = 0.1 * rng.standard_normal((d, r))
    out = []
    for tail in ("1", "0"):
        os.environ["PSMF_TAIL_REDUCE"] = tail
        f = c.DeviceFilter(d, r, storage=st, engine="step", **kw)
        if kw:
            f.set_row_noise(0.5 + rng.random(d))
        f.upload_series(Y)
        f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
        f.run(0, T)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); f.run(0, T); best = min(best, (time.perf_counter() - t0) / T)
        out.append(1e6 * best)
        f.close()
    print(f"d={d} r={r} {kw}: tail reduce {out[0]:.2f} us/step, serial-stage reduce {out[1]:.2f} us/step", flush=True)
