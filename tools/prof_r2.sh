# GPU-box recipe of the round-2 rocprofv3 evidence (profiles/README.md).  usage: bash tools/prof_r2.sh
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2p/pmc_f -- python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 --T 1000 --no-extras > gpurun_out/r2p/bench_pmc_f.json 2> gpurun_out/r2p/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2p/pmc_w -- python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 --T 1000 --no-extras > gpurun_out/r2p/bench_pmc_w.json 2> gpurun_out/r2p/pmc_w.err
cd tools && python pmc_block_engine.py ../gpurun_out/r2p/pmc.json ../gpurun_out/r2p/pmc_f ../gpurun_out/r2p/pmc_w > ../gpurun_out/r2p/pmc_summary.txt
cd .. && find gpurun_out/r2p -name "*kernel_trace.csv" -size +10M -delete || true
du -sh gpurun_out/r2p
