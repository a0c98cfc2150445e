"""Implicit d x d inverse innovation matrix  S^-1 = D - U K U^T  (diagonal + rank r).

The reference materialises S^-1 as a dense d x d array in
`_compute_inverse_coefficient_innovation` (pypsmf/psmf/psmf.py:140-153) and hands it to the two
coefficient-update hooks, which only ever multiply by it.  This operator supports exactly those
products (`S @ v`, `v.T @ S`, `S.T`) in O(d r) so that experiment subclasses overriding the hooks
keep working unchanged without any d x d object.
"""

import numpy as np


class InverseInnovation:
    __array_priority__ = 1000.0   # make numpy defer `ndarray @ self` to __rmatmul__

    def __init__(self, diag, U=None, K=None):
        self.diag = np.asarray(diag, dtype=float).reshape(-1)
        self.U = U          # (d, r) = diag[:, None] * C
        self.K = K          # (r, r)
        self.shape = (self.diag.size, self.diag.size)

    @property
    def T(self):
        return self  # symmetric

    def __matmul__(self, other):
        other = np.asarray(other)
        vec = other.ndim == 1
        B = other.reshape(self.shape[0], -1)
        out = self.diag[:, None] * B
        if self.U is not None:
            out = out - self.U @ (self.K @ (self.U.T @ B))
        return out.reshape(-1) if vec else out

    def __rmatmul__(self, other):
        other = np.asarray(other)
        vec = other.ndim == 1
        A = other.reshape(-1, self.shape[0])
        out = (self @ A.T).T
        return out.reshape(-1) if vec else out

    def toarray(self):
        return self @ np.eye(self.shape[0])
