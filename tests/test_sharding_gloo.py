"""The N > 1 logic on CPU, two processes over torch.distributed gloo (world_size 2), against the unsharded CPU oracle:
row sharding + the exchange of each device engine (tests/host_models.py) -- per-step: r + 1 partial sums per timestep;
blocked: one Gram per block; blocked pipelined (what bench.py --gpus N runs): first-block Gram, then one cross-Gram per
block with the next K assembled from it and the tracked Gram.  CPU only."""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, relerr
from oracle import psmf_oracle as O
from host_models import sharded_epoch_host
from rpsmf_amd.sharding import shard_bounds, shard_rows

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from rpsmf_amd.sharding import shard_rows
from host_models import sharded_epoch_host, blocked_epoch_host, blocked_pipelined_epoch_host, gloo_allreduce
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
z = np.load(sys.argv[2])
row0, dl = shard_rows(z["C0"].shape[0], world, rank)
robust = bool(int(sys.argv[4]))
model = sys.argv[5]
d = z["C0"].shape[0]
Cl, Yl = z["C0"][row0:row0 + dl], z["Y"][:, row0:row0 + dl]
if model == "step":
    C, V, P, mu, rho, lam, Yp = sharded_epoch_host(Cl, Yl, z["V0"], z["P0"], z["Q"], z["mu0"], 1.0, d, robust=robust, lambda0=1.8, dist=dist)
elif model == "block":
    C, V, P, mu, rho, lam, Yp = blocked_epoch_host(Cl, Yl, z["V0"], z["P0"], z["Q"], z["mu0"], 1.0, B=7, robust=robust, lambda0=1.8,
                                                   gram_allreduce=gloo_allreduce(dist), d_global=d)
else:
    C, V, P, mu, rho, lam, Yp = blocked_pipelined_epoch_host(Cl, Yl, z["V0"], z["P0"], z["Q"], z["mu0"], 1.0, B=7, robust=robust, lambda0=1.8,
                                                             allreduce=gloo_allreduce(dist), d_global=d)
np.savez(sys.argv[3] + f".{rank}.npz", C=C, V=V, P=P, mu=mu, rho=rho, lam=lam, Yp=Yp, row0=row0)
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_bounds_cover_all_rows():
    for d, w in ((100000, 8), (10, 3), (7, 7), (12500, 1)):
        b = shard_bounds(d, w)
        assert b[0] == 0 and b[-1] == d and np.all(np.diff(b) >= d // w) and np.all(np.diff(b) <= d // w + 1)
        assert [shard_rows(d, w, r) for r in range(w)] == [(int(b[r]), int(b[r + 1] - b[r])) for r in range(w)]
    with pytest.raises(ValueError):
        shard_rows(3, 4, 0)


def test_host_model_single_rank_equals_oracle():
    """The device algorithm's host model (tracked Gram) == the oracle's per-step exact algebra."""
    rng = np.random.default_rng(0)
    d, r, T = 120, 6, 40
    Y = O.synthetic_series(d, r, T, 3, dtype=np.float64)
    C0 = 0.1 * rng.standard_normal((d, r))
    for robust in (False, True):
        st = O.State(C=C0, V=0.1 * np.eye(r), mu=np.zeros(r), P=np.eye(r), Q=0.1 * np.eye(r), rho=1.0, lam=1.8)
        st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=robust), O.RandomWalkDyn())
        C, V, P, mu, rho, lam, Yp2 = sharded_epoch_host(C0, Y, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r),
                                                        1.0, d, robust=robust, lambda0=1.8)
        for a, b in ((C, st.C), (V, st.V), (P, st.P), (mu, st.mu), (Yp2, Yp)):
            assert relerr(a, b) < 1e-10


@pytest.mark.parametrize("model", ["step", "block", "block_pipelined"])
@pytest.mark.parametrize("robust", [0, 1])
def test_two_rank_gloo_equals_unsharded(tmp_path, robust, model):
    rng = np.random.default_rng(1)
    d, r, T = 101, 5, 30          # odd d: unequal shards
    Y = O.synthetic_series(d, r, T, 5, dtype=np.float64)
    C0 = 0.1 * rng.standard_normal((d, r))
    inp = tmp_path / "in.npz"
    np.savez(inp, C0=C0, Y=Y, V0=0.1 * np.eye(r), P0=np.eye(r), Q=0.1 * np.eye(r), mu0=np.zeros(r))
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(inp), str(tmp_path / "out"), str(robust), model],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()[-2000:]
    st = O.State(C=C0, V=0.1 * np.eye(r), mu=np.zeros(r), P=np.eye(r), Q=0.1 * np.eye(r), rho=1.0, lam=1.8)
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=bool(robust)), O.RandomWalkDyn())
    parts = [np.load(str(tmp_path / "out") + f".{rank}.npz") for rank in range(2)]
    C = np.vstack([p["C"] for p in parts])
    Ypr = np.hstack([p["Yp"] for p in parts])
    assert relerr(C, st.C) < 1e-10 and relerr(Ypr, Yp) < 1e-10
    for p in parts:   # replicated state agrees with the oracle and is bit-identical across ranks
        assert relerr(p["V"], st.V) < 1e-10 and relerr(p["P"], st.P) < 1e-10 and relerr(p["mu"], st.mu) < 1e-10
    for k in ("V", "P", "mu", "rho", "lam"):
        assert np.array_equal(parts[0][k], parts[1][k]), k
