"""GPU-box probe: where the cold pass of config E (one pass of T steps from the initial state) differs from a carried pass --
Newton-Schulz iterations, failed starts and direct sweeps per segment of 500 timesteps, and the segments' wall time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench

d, r, T = 100_000, 32, 10_000
ser = bench.Series(d, r, T, 35853, 0, d, False, global_noise=True)
st0 = bench.init_state(d, r, 35853)
f = _capi.DeviceFilter(d, r, storage="f32")
for a, Yc in ser.chunks():
    f.upload_series(Yc, t0=a, T_total=T)
def reset():
    f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
reset(); f.run(0, T); f.run(0, T)
seg = int(os.environ.get("SEG", "480"))
for label in ("cold", "carried"):
    if label == "cold":
        reset()
    f.sync()
    rows = []
    for a in range(0, T, seg):
        b = min(T, a + seg)
        f.counters(reset=True)
        t0 = time.perf_counter(); f.run(a, b); dt = time.perf_counter() - t0
        c = f.counters()
        rows.append((a, b, 1e6 * dt / (b - a), c["ns_iterations"] / max(1, c["ns_steps"]), c["ns_failed"], c["sweep_steps"], c["filter_us_mean"], c["filter_gap_us_mean"]))
    print(label)
    for rw in rows:
        print("  steps %5d-%5d  %.2f us/step  iter/step %.3f  failed %d  sweeps %d  block %.1f us gap %.2f" % rw, flush=True)
for label in ("cold", "carried"):
    if label == "cold":
        reset()
    f.sync(); f.counters(reset=True)
    t0 = time.perf_counter(); f.run(0, T); dt = time.perf_counter() - t0
    c = f.counters()
    print(label, "whole pass %.2f ms" % (1e3 * dt), c)
