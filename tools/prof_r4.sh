# GPU-box recipe of the round-4 evidence under profiles/ (profiles/README.md).  usage: bash tools/prof_r4.sh <part>   (part = a | b)
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"
O=gpurun_out/r4p
mkdir -p $O
if [ "$1" = "a" ]; then
  # the bench line of the final build
  python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  # kernel trace + stats of the default workload (whole-pass launches of the chained filter kernel)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-extras > $O/bench_under_rocprof.json 2> $O/kt.err
  python tools/trace_summary.py $O/kt "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-extras; whole-pass launches = 2 pre-warm + 2 cold + 1 warm-up + 5 timed (T = 10 000 each); the short launches are psmf_time_kernel stand-alone blocks" > $O/kernel_trace_chain_summary.json
  find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_chain.csv \;
  find $O -name "*kernel_trace.csv" -size +10M -delete || true
  # per-step and masked engines under the profiler (few steps: rocprofv3's CSV writer crashes on long traces on this image)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 tools/probe_step_r40.py 64 > $O/step_engine.txt 2> $O/step.err
  find $O/step -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_step_engine.csv \;
  python tools/probe_masked.py 512 > $O/masked_engine.txt 2>&1
  python tools/probe_modes.py > $O/modes.txt 2>&1
else
  # HBM traffic of the blocked engine's kernels (separate --pmc passes; FETCH_SIZE x 2 on gfx950)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 --T 1000 --no-extras > $O/bench_pmc_f.json 2> $O/pmc_f.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 --T 1000 --no-extras > $O/bench_pmc_w.json 2> $O/pmc_w.err
  (cd tools && python pmc_block_engine.py ../$O/pmc.json ../$O/pmc_f ../$O/pmc_w > ../$O/pmc_summary.txt)
  # instruction mix of the filter kernel
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 --T 1000 --no-extras > $O/bench_pmc_sq.json 2> $O/pmc_sq.err
  python - <<'PY' > $O/pmc_filter3_sq.json
import csv, glob, json, collections
f = glob.glob("gpurun_out/r4p/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "psmf_blk_filter3" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
out = {}
for k in acc:
    out[k[:60]] = {"launches (one per block of 32 timesteps under the profiler)": n[k], **{c: v / max(1, n[k]) / 32.0 for c, v in acc[k].items()}, "unit": "instructions per timestep (whole workgroup, 8 waves)"}
print(json.dumps(out, indent=1))
PY
  find $O -name "*kernel_trace.csv" -size +10M -delete || true
  find $O -name "*counter_collection.csv" -size +10M -delete || true
fi
du -sh $O
