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
    # ... and nothing else reaches stdout (gloo's connection line and RCCL's version banner used to: bench.py sends them to stderr)
    assert [ln for ln in pr.stdout.splitlines() if ln.strip()] == lines, pr.stdout[:2000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["config"]["engine"] == engine
    par = out["parity_vs_cpu_oracle"]
    assert par["ranks"] == 2 and max(par["C"], par["V"], par["mu"], par["P"]) < 1e-5, par
    assert par["replicated_state_bit_identical"] is True
    # the line says what carried the exchange, how many ranks it spanned and which GPU each rank drove
    ex = out["config"]["exchange"]
    assert ex["transport"].startswith("host-mediated") and ex["rccl_ranks"] is None and ex["world_size"] == 2 and ex["ranks"] == [2, 2]
    assert [p["rank"] for p in ex["rank_devices"]] == [0, 1] and ex["distinct_gpus"] == 1          # --one-device: both ranks on GPU 0
    assert all(len(p["pci_bus_id"]) >= 7 for p in ex["rank_devices"])
    if engine == "block":
        assert ex["messages"]["cross_gram_per_block_bytes"] == 128 * 64 * 8
        gaps = out["roofline"]["gap_between_blocks_us_per_rank"]
        assert len(gaps) == 2 and all(g is not None and g >= 0.0 for g in gaps)
    else:
        assert ex["messages"]["per_timestep_bytes"] == (r + 1) * 8


def test_bench_four_ranks_one_gpu_uneven_rows():
    """Four ranks (the most a one-GPU box lets share its card is six processes) with a row count that does not divide: every rank's
    shard, the exchange description with four entries, the N > 1 fields of the line (transport_fallback, scaling_note,
    whole_job_frac) and the sharded parity."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d, r = 10_002, 32
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--rows", str(d),
           "--latent-rank", str(r), "--timesteps", "400", "--cpu-steps", "150", "--no-extras", "--comm", "gloo", "--one-device"]
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=dict(os.environ, OMP_NUM_THREADS="2"), cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["value"] > 0
    ex = out["config"]["exchange"]
    assert ex["world_size"] == 4 and ex["ranks"] == [4, 4, 4, 4] and [p["rank"] for p in ex["rank_devices"]] == [0, 1, 2, 3]
    assert out["transport_fallback"] is True and "flat BY DESIGN" in out["scaling_note"]        # gloo, not RCCL: the line says so at top level
    assert 0.0 < out["roofline"]["whole_job_frac"] < 1.0
    assert len(out["roofline"]["gap_between_blocks_us_per_rank"]) == 4
    par = out["parity_vs_cpu_oracle"]
    assert par["ranks"] == 4 and max(par["C"], par["V"], par["mu"], par["P"]) < 1e-5 and par["replicated_state_bit_identical"] is True


def test_bench_falls_back_to_gloo_when_rccl_does_not_come_up():
    """The driver's N > 1 run asks for RCCL (the default).  If the communicator cannot be built on every rank the bench must still
    deliver its line -- over the host-mediated transport, saying so with RCCL's own message -- instead of losing the measurement.
    Provoked here the one way a one-GPU box can: two ranks on ONE device, which RCCL refuses."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "8000",
           "--latent-rank", "32", "--timesteps", "400", "--cpu-steps", "150", "--no-extras", "--one-device"]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    out = json.loads(lines[0])
    ex = out["config"]["exchange"]
    assert ex["rccl_init_error"] and ex["transport"].startswith("host-mediated") and ex["rccl_ranks"] is None and ex["ranks"] == [2, 2]
    assert out["transport_fallback"] is True
    par = out["parity_vs_cpu_oracle"]
    assert par["ranks"] == 2 and max(par["C"], par["V"], par["mu"], par["P"]) < 1e-5 and par["replicated_state_bit_identical"] is True


def test_bench_two_ranks_parity_against_stored_answers(tmp_path):
    """The in-run parity of a sharded run at BASELINE size cannot re-run the oracle (0.1 s per timestep): bench.py compares with
    STORED oracle answers of the unsharded workload (tests/golden/fullsize_E_*.npz at d = 100 000; format of
    tests/golden/make_golden_fullsize.py) -- replicated state bit-identical across ranks, the sketch S^T C summed over the
    shards.  Rehearsed here with a small fixture made on the spot by the same generator and handed over with --parity-fixture."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden.make_golden_fullsize import generate

    d, r, T = 8000, 32, 400
    fx = generate(False, d=d, r=r, T=T, epochs=1, checkpoints=(100, 300), log=lambda *a, **k: None)
    path = str(tmp_path / "fixture.npz")
    np.savez_compressed(path, **fx)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", str(d),
           "--latent-rank", str(r), "--timesteps", str(T), "--cpu-steps", "150", "--no-extras", "--comm", "gloo", "--one-device",
           "--parity-fixture", path]
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=dict(os.environ, OMP_NUM_THREADS="4"), cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-3000:]
    out = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][0])
    par = out["parity_vs_cpu_oracle"]
    assert par["checkpoints"] == [100, 300] and par["ranks"] == 2 and par["replicated_state_bit_identical"] is True and par["ok"] is True, par
    assert max(par[k] for k in ("V", "P", "mu", "eta", "N", "StC")) < 1e-5, par


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_weak_scaling_two_ranks_one_gpu():
    """`--scaling weak`: every rank holds --rows rows, the filter has N x as many; the line says so."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "4000",
           "--latent-rank", "32", "--timesteps", "400", "--cpu-steps", "150", "--no-extras", "--comm", "gloo", "--one-device", "--scaling", "weak"]
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=dict(os.environ, OMP_NUM_THREADS="4"), cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-3000:]
    out = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and out["n_gpus"] == 2
    assert out["config"]["d"] == 8000 and out["config"]["d_per_gpu"] == 4000 and "weak scaling" in out["config"]["workload"]
    par = out["parity_vs_cpu_oracle"]
    assert par["ranks"] == 2 and max(par["C"], par["V"], par["mu"], par["P"]) < 1e-5, par


def test_forced_collective_single_rank_bench_costs_nothing():
    """PSMF_FORCE_COLLECTIVE=1: bench.py at N = 1 with the RCCL all-reduce kernels of the sharded blocked engine really enqueued
    on the bulk stream (1-rank communicator) -- what each rank of an N > 1 run does besides waiting for its peers.  The exchange
    is one block ahead of its use and off the critical path: throughput within 3 % of the plain run."""
    vals = {}
    for forced in (False, True, False, True):          # interleaved: box drift hits both alike; best of two each
        env = dict(os.environ, OMP_NUM_THREADS="4")
        env.pop("PSMF_FORCE_COLLECTIVE", None)
        if forced:
            env["PSMF_FORCE_COLLECTIVE"] = "1"
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--timesteps", "3200", "--cpu-steps", "0", "--no-extras"]
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
        assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-3000:]
        out = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][0])
        assert (out["config"]["exchange"] is not None) == forced
        if forced:          # the communicator as RCCL itself reports it (ncclCommCount)
            assert out["config"]["exchange"]["rccl_ranks"] == 1 and out["config"]["exchange"]["forced_single_rank"] is True
        assert out["value_config_E_literal"] == out["cold_pass_steps_per_s"]
        vals[forced] = max(vals.get(forced, 0.0), out["value"])
    assert vals[True] > 0.97 * vals[False], vals
    print("timesteps/s plain vs forced collective:", vals)


def test_survivor_exits_when_its_peer_is_killed(tmp_path):
    """Fault injection on the N > 1 path: two ranks (fresh child processes, gloo host communicator, one GPU), rank 1 is killed
    inside the timed region; rank 0 must notice at its next exchange and exit non-zero -- within the 20 s bound that every wait
    of the sharded engines has -- instead of hanging in a collective or a device-side wait."""
    import signal
    import time

    port = _free_port()
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, OMP_NUM_THREADS="2", RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100000", "--warmup", "1", "--rows", "8000",
               "--latent-rank", "32", "--timesteps", "640", "--cpu-steps", "0", "--no-extras", "--comm", "gloo", "--one-device",
               "--pid-dir", str(tmp_path)]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    try:
        t0 = time.time()
        while not all(os.path.exists(tmp_path / f"rank{q}.timed") for q in (0, 1)):
            assert time.time() - t0 < 200, "ranks did not reach the timed region"
            assert all(p.poll() is None for p in procs), [p.stderr.read()[-2000:] for p in procs if p.poll() is not None]
            time.sleep(0.2)
        time.sleep(1.0)                                   # well inside the timed loop (it would run for hours)
        procs[1].send_signal(signal.SIGKILL)              # the exact PID this test started
        t_kill = time.time()
        rc0 = procs[0].wait(timeout=25)
        waited = time.time() - t_kill
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait(timeout=30)
    err0 = procs[0].stderr.read()
    assert rc0 not in (0, None), err0[-1500:]
    assert waited < 20.0, waited
    assert "all-reduce" in err0 or "allreduce" in err0.lower() or "Connection" in err0, err0[-1500:]
