import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_perf import run
T = 2000
for (d, r, wg) in ((100000, 32, 512), (100000, 32, 256), (10000, 20, 512), (100000, 64, 512), (100000, 16, 512), (100000, 8, 512)):
    for coef in (True, False):
        print(json.dumps(run(d, r, T, "f32", coef, wg)), flush=True)
