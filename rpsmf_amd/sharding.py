"""Row sharding of the d x r dictionary over GPUs (one process per GPU).

The filter shards naturally along the rows of C and y_k: each rank sweeps its own rows, the
ranks exchange the r+1 partial sums (h = C^T e, ee = e^T e) once per timestep with one
RCCL all-reduce, and every rank repeats the identical r x r float64 arithmetic, so the
replicated state stays bit-identical across ranks.
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
