"""Rehearsal of `bench.py --gpus 2` on ONE GPU: two processes under torch.distributed.run, rows of C / y sharded over them, the
exchange carried by the host-mediated communicator over gloo (RCCL refuses two ranks on one device), both ranks on device 0.
Everything of the N > 1 bench path except the RCCL transport runs: rendezvous, per-shard series and state, the sharded blocked
engine, barriers, max-over-ranks timing, the one JSON line -- plus a sharded parity check against the unsharded CPU oracle.
GPU only: `pytest -m gpu`."""

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.mark.parametrize("engine,d,r", [("block", 8000, 32), ("step", 3000, 12)])
def test_bench_two_ranks_one_gpu(engine, d, r):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", str(d),
           "--latent-rank", str(r), "--timesteps", "400", "--cpu-steps", "150", "--no-extras", "--comm", "gloo", "--one-device", "--engine", engine]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]          # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["config"]["engine"] == engine
    par = out["parity_vs_cpu_oracle"]
    assert par["ranks"] == 2 and max(par["C"], par["V"], par["mu"], par["P"]) < 1e-5, par
