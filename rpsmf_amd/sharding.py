"""Row sharding of the d x r dictionary over GPUs (one process per GPU).

The filter shards along the rows of C, y_k and y_hat_k; every r-sized quantity is replicated and every rank repeats the
identical float64 r x r / coefficient-space arithmetic on all-reduced inputs, so the replicated state stays bit-identical
across ranks.  What is exchanged depends on the engine:

* blocked engine (default, r <= 32): ONE sum-all-reduce per block of min(64 - r, 48) timesteps -- the 128 x 64 float64
  cross-Gram [Z | Y_next]^T Y_next (64 KB), on the bulk stream, one block ahead of its use, i.e. off the critical path --
  plus the 64 x 64 Gram of the first block of a run;
* per-step engine (r > 32, host-stepped dynamics, non-uniform R): the r + 1 partial sums (h = C^T e, ee = e^T e) once per
  timestep, between the local reduction kernel and the serial stage.

Transport: RCCL on the library's streams (`DeviceFilter.comm_init`), or any host transport through the host-mediated
communicator (`comm_init_host`: gloo in bench.py's one-GPU rehearsal, MPI, ...).
"""

import numpy as np

__all__ = ["shard_rows", "shard_bounds"]


def shard_bounds(d, world):
    """Contiguous, near-equal row blocks: bounds[i]..bounds[i+1] for rank i."""
    base, extra = divmod(int(d), int(world))
    sizes = [base + (1 if i < extra else 0) for i in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def shard_rows(d, world, rank):
    """(row0, d_local) of `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if world > d:
        raise ValueError("more ranks than rows")
    b = shard_bounds(d, world)
    return int(b[rank]), int(b[rank + 1] - b[rank])
