# GPU-box recipe of the round-5 evidence under profiles/ (profiles/README.md).  usage: bash tools/prof_r5.sh <part>   (part = a | b)
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"
O=gpurun_out/r5p
mkdir -p $O
if [ "$1" = "a" ]; then
  # the bench line of the final build
  python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  # kernel trace + stats of the default workload (whole-pass launches of the chained filter kernel)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-extras > $O/bench_under_rocprof.json 2> $O/kt.err
  python tools/trace_summary.py $O/kt "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-extras; whole-pass launches = 2 pre-warm + 2 cold + 1 warm-up + 5 timed (T = 10 000 each); the short launches are psmf_time_kernel stand-alone blocks" > $O/kernel_trace_chain_summary.json
  find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_chain.csv \;
  find $O -name "*kernel_trace.csv" -size +10M -delete || true
  # the persistent per-step kernel under the profiler: config E's shape, unmasked and masked
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ps -- python3 tools/probe_pstep_time.py 100000 32 2000 f32 > $O/pstep_under_rocprof.txt 2> $O/ps.err
  find $O/ps -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_pstep.csv \;
  python tools/probe_pstep.py > $O/pstep_parity_and_timing.txt 2>&1
  python tools/probe_pstep_masked.py > $O/pstep_masked_parity_and_timing.txt 2>&1
  python tools/probe_masked.py 512 > $O/masked_engine.txt 2>&1
  python tools/probe_modes.py > $O/modes.txt 2>&1
else
  # HBM traffic of the persistent per-step kernel (separate --pmc passes; FETCH_SIZE x 2 on gfx950)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 tools/probe_pstep_time.py 100000 32 2000 f32 > $O/pstep_pmc_f.txt 2> $O/pmc_f.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 tools/probe_pstep_time.py 100000 32 2000 f32 > $O/pstep_pmc_w.txt 2> $O/pmc_w.err
  (cd tools && python pmc_pstep.py ../$O/pmc_pstep.json ../$O/pmc_f ../$O/pmc_w 100000 32 2000 4 > ../$O/pmc_pstep_summary.txt)
  find $O -name "*kernel_trace.csv" -size +10M -delete || true
  find $O -name "*counter_collection.csv" -size +10M -delete || true
fi
du -sh $O
