"""Error tracking / logging mixin with the call surface of pypsmf/psmf/tracking.py (`TrackingMixin`).

`errors_init / errors_update / log / logs_save` keep the reference's attribute names (`_E_y`, `_E_train`, `_E_pred`,
`_E_theta`, `_logs`) and log-line format, so experiment `run()` methods written against the reference work unchanged.
With backend="hip" the three Frobenius norms of `Y_pred - Y` (tracking.py:63-76: full / train / prediction windows) are
reduced ON THE DEVICE -- `psmf_sq_error` over the resident training series and its y_hat buffer, `psmf_predict_sq_error`
for the roll-out window -- instead of copying the (T + n_pred) x d prediction matrix back (4 GB at d = 10^5, T = 10^4).
Plotting (`figures_*`) is out of scope: the methods exist and do nothing.
"""

import numpy as np

__all__ = ["TrackingMixin"]


class TrackingMixin:
    _tracking_on_device = 0      # number of errors_update calls served by the device reductions

    def log(self, i, n_iter, delta_t, verbose=True, prefix=""):
        c = {
            "iter": "[%s%03i/%i]" % (prefix, i, n_iter),
            "full": "||y - Cx||^2 = %.5f" % self._E_y[i],
            "train": "||y - Cx||^2 (train) = %.5f" % self._E_train[i],
            "pred": "||y - Cx||^2 (pred) = %.5f" % self._E_pred[i],
            "time": "Δt = %.3f" % delta_t,
        }
        parts = ["{full}", "{train}", "{pred}"] + ([] if self._E_theta is None else ["{theta}"]) + ["{time}"]
        if self._E_theta is not None:
            c["theta"] = "||θ* - θ||^2 = %.5f" % self._E_theta[i]
        line = ("{iter} " + ", ".join(parts)).format(**c)
        if not hasattr(self, "_logs"):
            self._logs = []
        self._logs.append(line)
        if verbose:
            print(line, flush=True)

    def logs_save(self, filename):
        if hasattr(self, "_logs"):
            with open(filename, "w") as fp:
                fp.write("\n".join(self._logs))

    @staticmethod
    def _stack(y, k0, k1):
        """y[k0..k1] (dict of (d, 1) arrays or array-like) as a (k1 - k0 + 1, d) array"""
        if k1 < k0:
            return np.zeros((0, 0))
        return np.concatenate([np.asarray(y[k], dtype=float).reshape(1, -1) for k in range(k0, k1 + 1)], axis=0)

    def errors_init(self, y, T, n_iter, n_pred, theta_true=None):
        self._E_y, self._E_train, self._E_pred = {}, {}, {}
        self._E_theta = None if theta_true is None else {}
        Y = self._stack(y, 1, T + n_pred)           # the initial prediction is C0 @ 0 = 0 (tracking.py:52)
        self._E_y[0] = float(np.linalg.norm(Y))
        self._E_train[0] = float(np.linalg.norm(Y[:T]))
        self._E_pred[0] = float(np.linalg.norm(Y[T:]))
        if theta_true is not None:
            self._E_theta[0] = float(np.linalg.norm(self.theta0 - theta_true))

    def _device_norms(self, y, T, n_pred):
        """(train^2, pred^2) from device reductions, or None when the device does not hold what is being compared."""
        dev = getattr(self, "_dev", None)
        if getattr(self, "backend", None) != "hip" or dev is None or not dev.store_y_pred or getattr(self, "_host_stepped", lambda: True)():
            return None
        from .psmf import _YPred, _content_hash

        if not isinstance(self._y_pred, _YPred) or self._y_pred._T != T or self._series_key is None:
            return None
        Yt = np.ascontiguousarray(self._stack(y, 1, T))
        if (T, Yt.dtype.str, _content_hash(Yt)) != self._series_key:      # the norms are against THIS y: it must be the resident series
            return None
        if any(dict.__contains__(self._y_pred, k) is False for k in range(T + 1, T + n_pred + 1)):
            return None                                                     # predict() has not been called for this window
        train2 = dev.sq_error(0, T)
        pred2 = dev.predict_sq_error(T, self._stack(y, T + 1, T + n_pred)) if n_pred else 0.0
        return train2, pred2

    def errors_update(self, i, y, T, n_pred, theta_true=None):
        norms = self._device_norms(y, T, n_pred)
        if norms is not None:
            train2, pred2 = norms
            self._tracking_on_device += 1
        else:
            Yp = np.concatenate([np.asarray(self._y_pred[k], dtype=float).reshape(1, -1) for k in range(1, T + n_pred + 1)], axis=0)
            D = Yp - self._stack(y, 1, T + n_pred)
            train2, pred2 = float(np.sum(D[:T] ** 2)), float(np.sum(D[T:] ** 2))
        self._E_y[i] = float(np.sqrt(train2 + pred2))
        self._E_train[i] = float(np.sqrt(train2))
        self._E_pred[i] = float(np.sqrt(pred2))
        if theta_true is not None:
            self._E_theta[i] = float(np.linalg.norm(self._theta[i] - theta_true))

    # plotting is not part of the hot path (out of scope); kept so that experiment run() methods call through
    def figures_init(self, live_plot=False):
        pass

    def figures_update(self, y_obs, T, n_pred, live_plot=False, x_true=None):
        pass

    def figures_close(self):
        pass
