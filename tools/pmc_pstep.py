"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `tools/probe_pstep_time.py d r T storage` -> HBM-side bytes per timestep of the
persistent per-step kernel (gfx950 correction: FETCH_SIZE x 2 for wide coalesced reads, MI355X_MICROARCH.md section HBM; WRITE_SIZE exact).

    python tools/pmc_pstep.py out.json <fetch_dir> <write_dir> d r T storage_bytes
"""
import json, sys
import collections, csv, glob, os


def summarise(d, counter):
    """per kernel: launches, mean / min / max of the counter (KB).  (pmc_summary.summarise cuts kernel names at the first "(", which
    is inside "(anonymous namespace)" for this kernel: own parser.)"""
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: dict(launches=len(v), mean_KB=sum(v) / len(v), min_KB=min(v), max_KB=max(v)) for k, v in acc.items()}


out, fdir, wdir = sys.argv[1:4]
d, r, T, es = (int(x) for x in sys.argv[4:8])
F, W = summarise(fdir, "FETCH_SIZE"), summarise(wdir, "WRITE_SIZE")
name = [n for n in F if "psmf_pstep_k" in n][0]
# the probe launches the kernel once for 300 timesteps (warm-up), then for T timesteps per timed run: take the largest launches
f_max, w_max = F[name].get("max_KB", F[name]["mean_KB"]) * 1024, W[name].get("max_KB", W[name]["mean_KB"]) * 1024
doc = dict(
    note=f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 tools/probe_pstep_time.py {d} {r} {T} ...`: "
         "ONE launch of psmf_pstep_k per run.  Bytes of the largest launch (T timesteps) / T; FETCH_SIZE as counted and doubled (gfx950 tallies "
         "128-byte requests at 64 bytes for wide coalesced reads; the kernel's y loads are 4-byte-per-lane rows, its polls 8-byte sc1 loads).",
    kernel=name, launches=F[name]["launches"], timesteps_per_launch=T,
    fetch_bytes_per_timestep_as_counted=f_max / T, fetch_bytes_per_timestep_doubled=2.0 * f_max / T, write_bytes_per_timestep=w_max / T,
    algorithmic_y_plus_yhat_bytes_per_timestep=2.0 * es * d,
    step_at_a_time_bytes_per_timestep=2.0 * es * d * (r + 1),
    FETCH_SIZE=F[name], WRITE_SIZE=W[name])
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k not in ("note", "FETCH_SIZE", "WRITE_SIZE")}, indent=1))
