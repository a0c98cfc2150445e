"""Learning-rate schedules for the theta optimiser (host-side scalars).

Interface of pypsmf/psmf/learning_rate.py: an object with ``get(t) -> float``.
"""

__all__ = ["BaseLearningRate", "ConstantLearningRate", "ExponentialLearningRate"]


class BaseLearningRate:
    def get(self, t):
        raise NotImplementedError


class ConstantLearningRate(BaseLearningRate):
    def __init__(self, lr):
        self.lr = float(lr)

    def get(self, t):
        return self.lr


class ExponentialLearningRate(BaseLearningRate):
    """Geometric interpolation from lr_start (t = 0) to lr_end (t = steps)."""

    def __init__(self, lr_start, lr_end, steps):
        self.lr_start, self.lr_end, self.steps = float(lr_start), float(lr_end), float(steps)

    def get(self, t):
        return self.lr_start * (self.lr_end / self.lr_start) ** (t / self.steps)
