"""Import alias: `from psmf import PSMFIter` -- the reference's own import lines (pypsmf/psmf/__init__.py:5-6, as used by
ExperimentSynthetic/synthetic_psmf.py:10-11, synthetic_rpsmf.py, beijing_psmf.py, synthetic_recursive_psmf.py) resolve to the
MI355X implementation in `rpsmf_amd` when this repository is on sys.path instead of pypsmf."""

from rpsmf_amd import __version__  # noqa: F401
from rpsmf_amd.psmf import PSMFIter, PSMFIterMissing, PSMFRecursive  # noqa: F401
from rpsmf_amd.rpsmf import rPSMFIter, rPSMFIterMissing, rPSMFRecursive  # noqa: F401
