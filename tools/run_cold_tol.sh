mkdir -p gpurun_out/r5g
for tol in 3e-4 6e-4 1e-3; do
  echo "== PSMF_NS_TOL=$tol" >> gpurun_out/r5g/cold_tol.txt
  PSMF_NS_TOL=$tol SEG=2000 timeout -k 10 200 python tools/probe_cold.py 2>&1 | grep -E "whole pass|steps" >> gpurun_out/r5g/cold_tol.txt
  echo "== PSMF_NS_TOL=$tol parity" >> gpurun_out/r5g/cold_tol.txt
  PSMF_NS_TOL=$tol timeout -k 10 300 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -s -k "config_B_C or single_gpu" 2>&1 | grep -E "worst|passed|failed|Error" | cut -c1-600 >> gpurun_out/r5g/cold_tol.txt
  PSMF_NS_TOL=$tol timeout -k 10 300 python -m pytest tests/test_hip_adversarial.py -m gpu -q -s -k "not filter4" 2>&1 | grep -E "worst|passed|failed|Error" | cut -c1-200 >> gpurun_out/r5g/cold_tol.txt
done
cat gpurun_out/r5g/cold_tol.txt
