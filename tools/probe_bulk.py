"""GPU-box probe: streaming bulk kernels (PSMF_BULK2=1) vs the previous ones (=0): agreement of the final state and timings."""
import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

CASES = ((100000, 32, 1280, 0), (10000, 20, 1100, 1), (20004, 24, 500, 0), (4100, 16, 300, 0), (65536, 32, 640, 1))

def child():
    from rpsmf_amd import _capi
    import bench
    out = {}
    for (d, r, T, rob) in CASES:
        seed = 35833 if rob else 35853
        ser = bench.Series(d, r, T, seed, 0, d, bool(rob))
        st0 = bench.init_state(d, r, seed)
        f = _capi.DeviceFilter(d, r, robust=bool(rob), storage="f32")
        for a, Yc in ser.chunks():
            f.upload_series(Yc, t0=a, T_total=T)
        f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
        ms1 = f.run_timed(0, T)
        s = f.get_state()
        yp = f.y_pred(T - 40, 40)
        ms2 = f.run_timed(0, T)
        tk = [f.time_kernel(i, 20) for i in range(3)]
        key = f"{d}_{r}_{rob}"
        np.savez(os.path.join(os.environ["OUTDIR"], f"bulk_{os.environ.get('PSMF_BULK2','1')}_{key}.npz"), C=s["C"][:4096], V=s["V"], P=s["P"], mu=s["mu"], yp=yp[:, :4096],
                 Csum=np.array([np.abs(s["C"]).sum()]), ypsum=np.array([np.abs(yp).sum()]))
        out[key] = dict(us_epoch1=1e3 * ms1 / T, us_epoch2=1e3 * ms2 / T, filter_us=tk[0], xgram_us=tk[1], apply_us=tk[2])
        f.close()
    print(json.dumps(out), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        outdir = "/tmp/bulkprobe"
        os.makedirs(outdir, exist_ok=True)
        for v in ("0", "1"):
            pr = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PSMF_BULK2=v, OUTDIR=outdir), capture_output=True, text=True, timeout=500)
            print("BULK2=" + v, pr.stdout[-3000:], pr.stderr[-2000:], flush=True)
        import glob
        for f0 in sorted(glob.glob(os.path.join(outdir, "bulk_0_*.npz"))):
            f1 = f0.replace("bulk_0_", "bulk_1_")
            if not os.path.exists(f1): continue
            a, b = np.load(f0), np.load(f1)
            rel = {k: float(np.max(np.abs(a[k] - b[k])) / (np.max(np.abs(a[k])) + 1e-300)) for k in a.files}
            print(os.path.basename(f0), json.dumps(rel), flush=True)
