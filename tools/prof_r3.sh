# GPU-box recipe of the round-3 rocprofv3 evidence (profiles/README.md).  usage: bash tools/prof_r3.sh
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"
mkdir -p gpurun_out/r3p
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3p/pmc_f -- python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 --T 1000 --no-extras > gpurun_out/r3p/bench_pmc_f.json 2> gpurun_out/r3p/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3p/pmc_w -- python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 --T 1000 --no-extras > gpurun_out/r3p/bench_pmc_w.json 2> gpurun_out/r3p/pmc_w.err
cd tools && python pmc_block_engine.py ../gpurun_out/r3p/pmc.json ../gpurun_out/r3p/pmc_f ../gpurun_out/r3p/pmc_w > ../gpurun_out/r3p/pmc_summary.txt
cd ..
# kernel trace + stats of the default workload (whole-pass launches of the chained filter kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3p/kt -- python3 bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-extras > gpurun_out/r3p/bench_under_rocprof.json 2> gpurun_out/r3p/kt.err
python tools/trace_summary.py gpurun_out/r3p/kt "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-extras; whole-pass launches = 2 pre-warm + 1 cold + 1 warm-up + 5 timed (T = 10 000 each); the short launches are psmf_time_kernel stand-alone blocks" > gpurun_out/r3p/kernel_trace_chain_summary.json
find gpurun_out/r3p -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3p/kernel_stats_chain.csv \;
find gpurun_out/r3p -name "*kernel_trace.csv" -size +10M -delete || true
du -sh gpurun_out/r3p
# the general-dynamics kernels under the profiler: per-kernel durations of the cos-phase full filter (psmf_blk_filter4) and FourierBasis (psmf_blk_filter)
# (rocprofv3 segfaults in its CSV writer on the many short launches of tools/probe_modes.py on this image: no kernel stats of the modes)
find gpurun_out/r3p -name "*kernel_trace.csv" -size +10M -delete || true
