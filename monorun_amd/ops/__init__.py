"""Drop-in mirror of the reference's ``monorun.ops`` surface (/root/reference/monorun/ops/__init__.py:1)."""
from .least_squares import *  # noqa: F401,F403
from .least_squares import __all__  # noqa: F401
