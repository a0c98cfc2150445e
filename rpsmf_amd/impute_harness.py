"""Host-side harness pieces of the imputation experiment (numpy only, no device work).

Semantics mirror ExperimentImpute/common.py of the reference so that the stored known
answers (ExperimentImpute/output/*.json: blake2b hashes of Y, C, X and the resulting errors)
can be replayed:

  prepare_missing            common.py:50-76   random block-missing mask; consumes the GLOBAL
                                               numpy RNG in the same order (one randint per row
                                               per sweep), so `np.random.seed(123)` reproduces
                                               the reference's masks bit for bit
  RMSEM                      common.py:79-84
  compute_number_inside_bars common.py:87-94   (vectorised; the reference is a Python double loop)
  matrix_hash                common.py:108-111
"""

import hashlib

import numpy as np

__all__ = ["prepare_missing", "RMSEM", "compute_number_inside_bars", "matrix_hash", "draw_problem"]


def prepare_missing(Ymiss, missRatio, misSeg=20):
    """Punch random segments of length ``misSeg`` into ``Ymiss`` (in place, set to NaN) until at
    least ``missRatio`` of all entries are missing.  Returns (achieved ratio, Mmiss) where
    Mmiss is 1 on the artificially removed entries."""
    d, n = Ymiss.shape
    native = int(np.isnan(Ymiss).sum())
    Mmiss = np.zeros_like(Ymiss)
    total = d * n
    removed = 0                      # == Mmiss.sum(), kept as a running count (the reference re-sums d*n entries per sweep)
    ratio = native / total
    rows = np.repeat(np.arange(d), misSeg)
    offs = np.tile(np.arange(misSeg), d)
    while ratio < missRatio:
        # one segment start per row, drawn in row order from the global RNG exactly as d scalar randint calls would
        starts = np.random.randint(1, n - misSeg, size=d)
        cols = np.repeat(starts, misSeg) + offs
        fresh = ~np.isnan(Ymiss[rows, cols])
        Mmiss[rows[fresh], cols[fresh]] = 1
        Ymiss[rows, cols] = np.nan
        removed += int(fresh.sum())
        ratio = (native + removed) / total
    return ratio, Mmiss


def RMSEM(Y1, Y2, M):
    """Root-mean-square difference of Y1 and Y2 over the entries where M == 1."""
    diff = (Y1 - Y2) * M
    return np.sqrt(np.sum(diff * diff) / np.sum(M))


def compute_number_inside_bars(Mmiss, m, n, Yorg, YrecL, YrecH):
    """Fraction of held-out entries lying strictly inside (YrecL, YrecH)."""
    sel = Mmiss[:m, :n] == 1
    hit = sel & (Yorg[:m, :n] < YrecH[:m, :n]) & (YrecL[:m, :n] < Yorg[:m, :n])
    return hit.sum() / np.sum(Mmiss)


def matrix_hash(A):
    return hashlib.blake2b(A.tobytes(), digest_size=16).hexdigest()


def draw_problem(Yorig, percentage, r):
    """One repeat of the experiment's input draw (PSMF.py:138-158): mask, zero-filled data,
    C ~ U(0,1)^{d x r}, X ~ U(0,1)^{r x n}, all from the global numpy RNG, in that order."""
    d, n = Yorig.shape
    Ymiss = np.copy(Yorig)
    ratio, Mmiss = prepare_missing(Ymiss, percentage / 100)
    M = np.array(~np.isnan(Ymiss), dtype=int)
    Y = np.nan_to_num(Ymiss, nan=0.0)
    C = np.random.rand(d, r)
    X = np.random.rand(r, n)
    return dict(Y=Y, M=M, Mmiss=Mmiss, C=C, X=X, ratio=ratio)
