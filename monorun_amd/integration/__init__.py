"""Glue a maintainer of the reference adds on the MonoRUn side (no mmdet / mmcv import at module level)."""
from .dump_hook import PoseStageDump  # noqa: F401
