"""Mirror of /root/reference/monorun/ops/least_squares/__init__.py:1-5 (same names, same order)."""
from .pnp_uncert_cpu import u2d_pnp_cpu
from .pnp_uncert import PnPUncert, pnp_uncert
from .builder import build_pnp, PNP

__all__ = ['u2d_pnp_cpu', 'build_pnp', 'PnPUncert', 'pnp_uncert', 'PNP']
