"""Multi-GPU sharding of the PnP hot path: one process per GPU, objects split into contiguous shards,
ONE all-gather of a packed per-object result over RCCL/xGMI (the reference has no
inference-time sharding — test.py:74-75 refuses >1 GPU — so this is new surface, not a port).

Objects are independent (no cross-object term in R1-R13), so there is no data-path collective besides
the final exchange.  The packed row is 85 bytes/object -> rounded to 88:
    [pose 4xf32 | cov 16xf32 | tr f32 | valid u8 + 3 pad]
1024 objects/rank -> 88 KiB per rank: latency-bound on xGMI, hence a single fused all-gather of one
byte buffer instead of one collective per tensor.
"""
import torch
import torch.distributed as dist

ROW_BYTES = 88


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n objects for `rank`; every rank gets ceil(n/world) slots."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


class PackedResults:
    """One flat uint8 buffer holding the per-object outputs of a shard; typed views alias it, so the
    kernel writes straight into the buffer that the collective sends (no packing kernels)."""

    def __init__(self, n, device):
        self.n = n
        self.buf = torch.zeros(max(n, 1) * ROW_BYTES, dtype=torch.uint8, device=device)
        o = 0
        self.pose = self.buf[o:o + n * 16].view(torch.float32).view(n, 4); o += n * 16
        self.cov = self.buf[o:o + n * 64].view(torch.float32).view(n, 4, 4); o += n * 64
        self.tr = self.buf[o:o + n * 4].view(torch.float32); o += n * 4
        self.valid = self.buf[o:o + n]

    @staticmethod
    def unpack(flat, n):
        """flat: (world, per*ROW_BYTES) uint8 -> dict of (world*per, ...) tensors, truncated by the caller."""
        w = flat.shape[0]
        per = flat.shape[1] // ROW_BYTES
        o = 0
        pose = flat[:, o:o + per * 16].contiguous().view(torch.float32).view(w * per, 4); o += per * 16
        cov = flat[:, o:o + per * 64].contiguous().view(torch.float32).view(w * per, 4, 4); o += per * 64
        tr = flat[:, o:o + per * 4].contiguous().view(torch.float32).view(w * per); o += per * 4
        valid = flat[:, o:o + per].contiguous().view(w * per)
        # shards are padded to `per`; the global order is rank-major, object n' = rank*per + i
        keep = torch.arange(w * per, device=flat.device) < n if n < w * per else None
        out = dict(pose=pose, cov=cov, tr=tr, valid=valid.bool())
        if keep is not None:
            out = {k: v[:n] for k, v in out.items()}
        return out


def all_gather_results(packed, per, group=None):
    """All-gather the packed shard buffers.  Returns (world, per*ROW_BYTES) uint8 on every rank."""
    world = dist.get_world_size(group)
    assert packed.buf.numel() == max(per, 1) * ROW_BYTES or packed.n == per
    out = torch.empty(world * packed.buf.numel(), dtype=torch.uint8, device=packed.buf.device)
    dist.all_gather_into_tensor(out, packed.buf, group=group)
    return out.view(world, -1)


def sharded_pnp(solve_shard, n_objects, device, group=None):
    """Run `solve_shard(lo, hi, packed)` on this rank's contiguous shard (it must fill `packed`'s views
    for hi-lo objects) and exchange the results.  Returns the unpacked global results on every rank."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi, per = shard_bounds(n_objects, rank, world)
    packed = PackedResults(per, device)
    if hi > lo:
        solve_shard(lo, hi, packed)
    flat = all_gather_results(packed, per, group)
    return PackedResults.unpack(flat, n_objects)
