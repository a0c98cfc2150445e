"""Import alias of rpsmf_amd.tracking (pypsmf/psmf/tracking.py's module path)."""

from rpsmf_amd.tracking import *  # noqa: F401,F403
from rpsmf_amd.tracking import __all__  # noqa: F401
