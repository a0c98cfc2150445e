"""Import alias of rpsmf_amd.learning_rate (pypsmf/psmf/learning_rate.py's module path)."""

from rpsmf_amd.learning_rate import *  # noqa: F401,F403
from rpsmf_amd.learning_rate import __all__  # noqa: F401
