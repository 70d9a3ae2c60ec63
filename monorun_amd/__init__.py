"""monorun_amd — MI355X-native (gfx950 HIP) implementation of MonoRUn's uncertainty-aware PnP hot path.

Drop-in surface: ``monorun_amd.ops`` mirrors the reference's ``monorun.ops``
(build_pnp / PnPUncert / pnp_uncert / u2d_pnp_cpu / PNP).  Around it: ``pose_head`` (NOC-head decode, pose head mirror,
fused head -> pose launch, RoIAlign), ``consumers`` (3-D box packing, rotated-BEV NMS), ``evaluation`` (KITTI evaluator and
wire format), ``parallel`` (object sharding, RCCL exchange), ``synthetic`` (seeded workloads).  See DESIGN.md and INTEGRATION.md.
"""
from . import _lib  # noqa: F401
from .ops import build_pnp, PnPUncert, pnp_uncert, u2d_pnp_cpu, PNP  # noqa: F401
from .ops.least_squares.pnp_uncert import PnPLaunch, PnPEpnpLaunch, PnPEpnpGroupLaunch, PnPPipeline, pnp_uncert_device  # noqa: F401

__version__ = '0.1.0'
