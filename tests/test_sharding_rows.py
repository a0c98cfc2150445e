"""Row sharding bookkeeping for 1, 2, 4 and 8 ranks, CPU only: contiguous, near-equal, covering shards -- BASELINE
config E's 100 000 rows over 8 GPUs are 12 500 each -- and the bounded RCCL start of bench.py (a start that hangs must end the
process with a message, never look like a slow run)."""

import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from rpsmf_amd.sharding import shard_bounds, shard_rows


@pytest.mark.parametrize("d", [100_000, 100_003, 10_002, 17])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shards_cover_the_rows(d, world):
    b = shard_bounds(d, world)
    assert b[0] == 0 and b[-1] == d and len(b) == world + 1
    sizes = np.diff(b)
    assert sizes.min() >= d // world and sizes.max() <= -(-d // world)
    rows = [shard_rows(d, world, k) for k in range(world)]
    assert rows[0][0] == 0 and all(rows[k][0] + rows[k][1] == rows[k + 1][0] for k in range(world - 1))
    assert sum(n for _, n in rows) == d


def test_config_E_over_eight_gpus():
    assert [shard_rows(100_000, 8, k) for k in range(8)] == [(12_500 * k, 12_500) for k in range(8)]
    with pytest.raises(ValueError):
        shard_rows(4, 8, 0)
    with pytest.raises(ValueError):
        shard_rows(100, 4, 4)


def test_a_hung_communicator_start_ends_the_process_with_a_message(tmp_path):
    """bench.py runs psmf_comm_init in a helper thread and gives it PSMF_COMM_INIT_TIMEOUT seconds.  The same pattern, with a
    stand-in for the call that never returns: the process must exit with code 3 and say why."""
    src = textwrap.dedent("""
        import os, sys, threading, time
        box = {}
        def _init():
            time.sleep(3600)          # ncclCommInitRank with a peer that never joins
        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(float(os.environ.get("PSMF_COMM_INIT_TIMEOUT", "240")))
        if th.is_alive():
            print("[bench] rank 0: RCCL communicator initialisation did not return", file=sys.stderr, flush=True)
            os._exit(3)
        print("unreachable")
    """)
    # the pattern above is bench.py's, verbatim in structure; check that bench.py still contains it
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert "PSMF_COMM_INIT_TIMEOUT" in text and "os._exit(3)" in text and "th.join(" in text and "comm_abort" in text
    pr = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=60, env=dict(os.environ, PSMF_COMM_INIT_TIMEOUT="0.5"))
    assert pr.returncode == 3 and "did not return" in pr.stderr and "unreachable" not in pr.stdout


@pytest.mark.parametrize("batch,parts", [(50, 8), (50, 1), (100, 8), (3, 8), (8, 8)])
def test_replica_slices_for_config_D(batch, parts):
    """BASELINE config D's 50 seeds over the GPUs of a node ("replicas only")."""
    from rpsmf_amd.impute import replica_slices

    sl = replica_slices(batch, parts)
    assert len(sl) == parts and sl[0][0] == 0 and sl[-1][1] == batch
    assert all(sl[i][1] == sl[i + 1][0] for i in range(parts - 1))
    n = [b - a for a, b in sl]
    assert max(n) - min(n) <= 1 and sum(n) == batch
