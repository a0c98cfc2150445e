import json, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("pm", "/root/repo/tools/probe_pstep_masked.py")
pm = importlib.util.module_from_spec(spec); spec.loader.exec_module(pm)
from rpsmf_amd import _capi as c

def dbg(d, r, T):
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((T, d)); M = (rng.random((T, d)) > 0.4).astype(np.uint8); Y = Y * M
    C0 = 0.1 * rng.standard_normal((d, r))
    cuts = tuple((k, k + 1) for k in range(T))
    try:
        kp, po = pm.run(True, d, r, Y, M, C0, False, "f64", cuts)
    except Exception as e:
        print(d, r, "persistent per-step launches failed:", str(e)[:80]); po = None
    kt, to = pm.run(False, d, r, Y, M, C0, False, "f64", cuts)
    if po:
        for i in range(T):
            print(d, r, "single-step launches, step", i + 1, {n: float("%.1e" % pm.relerr(po[i][n], to[i][n])) for n in ("C", "V", "mu", "P", "sc")})
    try:
        kp, po = pm.run(True, d, r, Y, M, C0, False, "f64", ((0, T),))
        print(d, r, "one launch:", {n: float("%.1e" % pm.relerr(po[0][n], to[-1][n])) for n in ("C", "V", "mu", "P")})
    except Exception as e:
        print(d, r, "one launch failed:", str(e)[:80])

dbg(4096, 12, 5)
dbg(1000, 20, 5)
dbg(4096, 32, 5)
dbg(300, 32, 5)
