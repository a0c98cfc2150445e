"""rpsmf_amd: MI355X-native PSMF / rPSMF recursive filter behind the pypsmf call surface.

    from rpsmf_amd import PSMFIter, rPSMFIter, PSMFRecursive, rPSMFRecursive
    from rpsmf_amd.impute import ProbabilisticSequentialMatrixFactorizer, robust_PSMF

The device path (backend="hip", default) needs rpsmf_amd/lib/libpsmf_hip.so
(`python -m rpsmf_amd.build`) and a gfx950 GPU; it never falls back to the CPU.
"""

from .learning_rate import BaseLearningRate, ConstantLearningRate, ExponentialLearningRate  # noqa: F401
from .nonlinearities import (BaseNonLinearity, CosPhase, FourierBasis, RandomWalk, ScaledWalk, Sinusoid,  # noqa: F401
                             wrap_nonlinearity)
from .psmf import PSMFIter, PSMFIterMissing, PSMFRecursive  # noqa: F401
from .rpsmf import rPSMFIter, rPSMFIterMissing, rPSMFRecursive  # noqa: F401
from .tracking import TrackingMixin  # noqa: F401

__version__ = "0.1.0"
