import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from tools.probe_perf import run
    for wg in (256, 512, 1024, 2048):
        for coef in (False, True):
            o = run(100000, 32, 1500, "f32", coef, wg); o["U"] = os.environ.get("PSMF_UNROLL", "4")
            print(json.dumps(o), flush=True)
else:
    for u in ("4", "8"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PSMF_UNROLL=u))
