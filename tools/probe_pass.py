"""GPU-box probe: per-pass time of the headline workload (event-timed psmf_run_timed) and inversion counters."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench

d, r, T = 100000, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ser = bench.Series(d, r, T, 35853, 0, d, False)
st0 = bench.init_state(d, r, 35853)
f = _capi.DeviceFilter(d, r, storage="f32", store_y_pred=("noyp" not in sys.argv))
for a, Yc in ser.chunks():
    f.upload_series(Yc, t0=a, T_total=T)
f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
for i in range(5):
    f.counters(reset=True)
    t0 = time.perf_counter()
    try:
        ms = f.run_timed(0, T)
    except Exception as e:
        ms = float('nan'); print('   (', type(e).__name__, ')')
    wall = time.perf_counter() - t0
    c = f.counters()
    print(f"pass {i}: {1e3 * ms / T:.3f} us/step (event) {1e6 * wall / T:.3f} (wall)  per block of 32: {32e3 * ms / T:.1f} us  NS its/step {c['ns_iterations'] / max(1, c['ns_steps']):.2f} sweeps {c['sweep_steps']} | in-situ filter kernel {c['filter_us_mean']:.1f} us, gap {c['filter_gap_us_mean']:.1f} us ({c['filter_launches']} launches)", flush=True)
t0 = time.perf_counter()
try:
    for i in range(3):
        f.run(0, T, sync=False)
    f.sync()
except Exception:
    pass
print(f"3 passes enqueued back to back: {1e6 * (time.perf_counter() - t0) / (3 * T):.3f} us/step", flush=True)
f.close()
