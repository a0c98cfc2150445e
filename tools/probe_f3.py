"""GPU-box probe: filter3 vs filter2 (PSMF_FILTER3=0/1 in child processes): parity against each other and timing."""
import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def child():
    from rpsmf_amd import _capi
    import bench
    out = {}
    for (d, r, T, rob, storage) in ((20000, 32, 1500, 0, "f64"), (20000, 24, 1500, 1, "f64"), (100000, 32, 3000, 0, "f32"), (10000, 20, 3000, 1, "f32")):
        seed = 35833 if rob else 35853
        ser = bench.Series(d, r, T, seed, 0, d, bool(rob))
        st0 = bench.init_state(d, r, seed)
        f = _capi.DeviceFilter(d, r, robust=bool(rob), storage=storage)
        for a, Yc in ser.chunks():
            f.upload_series(Yc, t0=a, T_total=T)
        f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
        f.counters(reset=True)
        ms1 = f.run_timed(0, T)
        c1 = f.counters(reset=True)
        s = f.get_state()
        ms2 = f.run_timed(0, T)
        c2 = f.counters(reset=True)
        tk = f.time_kernel(0, 20)
        key = f"{d}_{r}_{rob}_{storage}"
        np.savez(os.path.join(os.environ["OUTDIR"], f"state_{os.environ.get('PSMF_FILTER3','1')}_{key}.npz"), **{k: np.asarray(v) for k, v in s.items() if v is not None})
        out[key] = dict(us_epoch1=1e3 * ms1 / T, us_epoch2=1e3 * ms2 / T, filter_us=tk, c1=c1, c2=c2)
        f.close()
    print(json.dumps(out), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "f3")
        os.makedirs(outdir, exist_ok=True)
        res = {}
        for v in ("0", "1"):
            pr = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PSMF_FILTER3=v, OUTDIR=outdir), capture_output=True, text=True, timeout=500)
            print("FILTER3=" + v, pr.stdout[-3000:], pr.stderr[-2000:], flush=True)
        import glob
        for f0 in sorted(glob.glob(os.path.join(outdir, "state_0_*.npz"))):
            f1 = f0.replace("state_0_", "state_1_")
            if not os.path.exists(f1): continue
            a, b = np.load(f0), np.load(f1)
            rel = {k: float(np.max(np.abs(a[k] - b[k])) / (np.max(np.abs(a[k])) + 1e-300)) for k in a.files if a[k].dtype.kind == "f" and a[k].size}
            print(os.path.basename(f0), json.dumps(rel), flush=True)
