import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from tools.probe_perf import run
    for (d, r) in ((100000, 32), (10000, 20)):
        for wg in (128, 256, 512):
            for coef in (False, True):
                o = run(d, r, 1500, "f32", coef, wg); o["NT"] = os.environ.get("PSMF_SWEEP_THREADS", "512")
                print(json.dumps(o), flush=True)
else:
    for nt in ("256", "512"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PSMF_SWEEP_THREADS=nt))
