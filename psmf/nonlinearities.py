"""Import alias of rpsmf_amd.nonlinearities (pypsmf/psmf/nonlinearities.py's module path)."""

from rpsmf_amd.nonlinearities import *  # noqa: F401,F403
from rpsmf_amd.nonlinearities import __all__  # noqa: F401
