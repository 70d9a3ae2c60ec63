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
    kernel writes straight into the buffer that the collective sends (no packing kernels).
    extra_f32: additional float32 columns per object riding in the same row (e.g. the decoded dimensions a consumer on another rank
    needs next to the pose: tools/kitti_val.py) — view ``extra`` (n, extra_f32); row = 88 + 4 extra_f32 bytes."""

    def __init__(self, n, device, buf=None, extra_f32=0):
        """buf: an existing uint8 tensor of max(n,1)*row_bytes bytes to alias (e.g. one slot of a larger buffer that is exchanged
        as a whole: several steps' results in ONE collective), or None to allocate."""
        self.n = n
        self.extra_f32 = int(extra_f32)
        self.row_bytes = ROW_BYTES + 4 * self.extra_f32
        if buf is None:
            buf = torch.zeros(max(n, 1) * self.row_bytes, dtype=torch.uint8, device=device)
        assert buf.dtype == torch.uint8 and buf.is_contiguous() and buf.numel() == max(n, 1) * self.row_bytes and buf.data_ptr() % 4 == 0
        self.buf = buf
        o = 0
        self.pose = self.buf[o:o + n * 16].view(torch.float32).view(n, 4); o += n * 16
        self.cov = self.buf[o:o + n * 64].view(torch.float32).view(n, 4, 4); o += n * 64
        self.tr = self.buf[o:o + n * 4].view(torch.float32); o += n * 4
        self.extra = None
        if self.extra_f32:
            self.extra = self.buf[o:o + n * 4 * self.extra_f32].view(torch.float32).view(n, self.extra_f32); o += n * 4 * self.extra_f32
        self.valid = self.buf[o:o + n]

    @staticmethod
    def unpack(flat, n, extra_f32=0):
        """flat: (world, per*row_bytes) uint8 -> dict of (world*per, ...) tensors, truncated to the n real objects."""
        w = flat.shape[0]
        per = flat.shape[1] // (ROW_BYTES + 4 * int(extra_f32))
        o = 0
        pose = flat[:, o:o + per * 16].contiguous().view(torch.float32).view(w * per, 4); o += per * 16
        cov = flat[:, o:o + per * 64].contiguous().view(torch.float32).view(w * per, 4, 4); o += per * 64
        tr = flat[:, o:o + per * 4].contiguous().view(torch.float32).view(w * per); o += per * 4
        out = dict(pose=pose, cov=cov, tr=tr)
        if extra_f32:
            out['extra'] = flat[:, o:o + per * 4 * extra_f32].contiguous().view(torch.float32).view(w * per, extra_f32); o += per * 4 * extra_f32
        out['valid'] = flat[:, o:o + per].contiguous().view(w * per).bool()
        # shards are padded to `per`; the global order is rank-major, object n' = rank*per + i
        if n < w * per:
            out = {k: v[:n] for k, v in out.items()}
        return out


def all_gather_results(packed, per, group=None):
    """All-gather the packed shard buffers.  Returns (world, per*ROW_BYTES) uint8 on every rank."""
    world = dist.get_world_size(group)
    assert packed.buf.numel() == max(per, 1) * packed.row_bytes or packed.n == per
    out = torch.empty(world * packed.buf.numel(), dtype=torch.uint8, device=packed.buf.device)
    dist.all_gather_into_tensor(out, packed.buf, group=group)
    return out.view(world, -1)


def sharded_pnp(solve_shard, n_objects, device, group=None, extra_f32=0, exchange=None):
    """Run `solve_shard(lo, hi, packed)` on this rank's contiguous shard (it must fill `packed`'s views
    for hi-lo objects) and exchange the results in ONE all-gather.  Returns the unpacked global results on every rank.
    exchange: None = torch.distributed's all_gather_into_tensor (RCCL under the 'nccl' backend, gloo on CPU), or a
    ``RcclAllGather`` (private communicator on a side stream; the current stream then waits for its completion event)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi, per = shard_bounds(n_objects, rank, world)
    packed = PackedResults(per, device, extra_f32=extra_f32)
    if hi > lo:
        solve_shard(lo, hi, packed)
    if exchange is None:
        flat = all_gather_results(packed, per, group)
    else:
        recv = torch.empty(world * packed.buf.numel(), dtype=torch.uint8, device=packed.buf.device)
        torch.cuda.current_stream(packed.buf.device).wait_event(exchange.gather(packed.buf, recv))
        exchange.forget(packed.buf)
        flat = recv.view(world, -1)
    return PackedResults.unpack(flat, n_objects, extra_f32=extra_f32)


# ---------------------------------------------------------------------------------------------------
# Direct RCCL path.  A launch of the PnP kernel lasts ~70 us; a c10d collective costs ~40-50 us of host time per call,
# which makes a per-step all-gather host- or latency-bound.  `RcclAllGather` owns its own RCCL communicator (bootstrapped
# through the existing torch.distributed group) and enqueues ncclAllGather straight onto a side stream (~5 us of host
# time), ordered against the compute stream with events, so the exchange of step i overlaps the kernel of step i+1.
import ctypes
import os


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]               # rccl.h: NCCL_UNIQUE_ID_BYTES


def _rccl():
    # the RCCL build torch itself links (so that one process holds one RCCL), else the system's; MR_RCCL_LIBRARY overrides (tests use it
    # to provoke the set-up failure whose fallback `agreed_rccl_all_gather` handles)
    path = os.environ.get('MR_RCCL_LIBRARY') or os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    lib = ctypes.CDLL(path if (os.environ.get('MR_RCCL_LIBRARY') or os.path.exists(path)) else 'librccl.so')
    lib.ncclGetUniqueId.restype = ctypes.c_int
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_NcclUniqueId)]
    lib.ncclCommInitRank.restype = ctypes.c_int
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
    lib.ncclAllGather.restype = ctypes.c_int
    lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclCommCount.restype = ctypes.c_int
    lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.ncclCommDestroy.restype = ctypes.c_int
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    return lib


def _agree(ok, device, group):
    """MIN over the ranks of a local outcome (1 = fine), through the job's own c10d group: the ONE kind of collective every rank issues in every
    phase below, whatever happened to it locally."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend(group) == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item()) == 1


def agreed_rccl_all_gather(device, group=None):
    """``RcclAllGather(device)`` on EVERY rank or on none: returns (exchange | None, reason | None).

    The private communicator is an optimisation (its collective costs ~5 us of host time against ~45 us through c10d); a rank that
    cannot set it up — library not found, a symbol missing, ncclGetUniqueId or ncclCommInitRank refusing — must not leave the others
    waiting inside a collective it will never join.  The set-up therefore runs in AGREED PHASES, and no rank ever skips a c10d collective
    another rank issues:
      1. every rank loads the library, rank 0 also takes the unique id (both inside try); all_reduce(MIN) of the outcomes;
      2. only if every rank is fine: broadcast of the id (every rank takes part);
      3. every rank calls ncclCommInitRank and checks ncclCommCount against the job; all_reduce(MIN) again.
    If any phase fails anywhere, all ranks drop to ``torch.distributed``'s all-gather together (the caller says so in its output).  What this
    does NOT cover: a bootstrap that hangs instead of failing — e.g. a rank whose ncclCommInitRank returns an error while the others are still
    inside theirs — (then the job's own timeout applies).  Failure modes handled: see DESIGN.md section 8."""
    why, lib, uid = None, None, _NcclUniqueId()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    try:                                                         # phase 1: local set-up, nothing collective inside the try
        lib = _rccl()
        if rank == 0:
            rc = lib.ncclGetUniqueId(ctypes.byref(uid))
            if rc != 0:
                raise RuntimeError(f'ncclGetUniqueId: RCCL error {rc}: {lib.ncclGetErrorString(rc).decode()}')
    except Exception as e:                                       # noqa: BLE001 — any set-up problem
        why = f'{type(e).__name__}: {e}'
    if not _agree(why is None, device, group):
        return None, why or 'another rank could not load RCCL / take the unique id for the private communicator'
    box = [ctypes.string_at(ctypes.byref(uid), 128)]             # phase 2 (every rank is fine, so every rank is here): all 128 raw bytes of the id
    dist.broadcast_object_list(box, src=0, group=group)
    ag = None
    try:                                                         # phase 3: the communicator itself
        ag = RcclAllGather(device, group, _lib=lib, _uid_bytes=box[0])
        if ag.nranks() != world:
            why = f'ncclCommCount reports {ag.nranks()} ranks, the job has {world}'
    except Exception as e:                                       # noqa: BLE001
        why = f'{type(e).__name__}: {e}'
    if _agree(why is None, device, group):
        return ag, None
    if ag is not None:
        try:
            ag.close()
        except Exception:                                        # noqa: BLE001
            pass
    return None, why or 'another rank could not set up its private RCCL communicator'


class RcclAllGather:
    """Byte all-gather over a private RCCL communicator, enqueued on a side stream.

        ag = RcclAllGather(device)                       # inside an initialised torch.distributed job (any backend)
        done = ag.gather(send_u8, recv_u8)               # send must have been produced on the CURRENT stream
        torch.cuda.current_stream().wait_event(done)     # ... when (and where) the gathered bytes are consumed

    `gather` makes the side stream wait for everything enqueued so far on the current stream, enqueues the collective
    there and returns the event that marks its completion (one event per send buffer, re-recorded on every call: keep a ring of
    send buffers and wait on a buffer's event before rewriting it); it never blocks the host."""

    NCCL_UINT8 = 1

    def __init__(self, device, group=None, _lib=None, _uid_bytes=None):
        """_lib / _uid_bytes: the library handle and the unique id ``agreed_rccl_all_gather`` obtained in its own agreed phases (then nothing
        collective happens in here before ncclCommInitRank).  Without them the constructor does both itself — a set-up failure on ONE rank
        then leaves the others inside the broadcast: use ``agreed_rccl_all_gather`` in jobs of more than one rank."""
        self.lib = _lib if _lib is not None else _rccl()
        self.dev = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        uid = _NcclUniqueId()
        if _uid_bytes is None:
            if self.rank == 0:
                self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)))
            box = [ctypes.string_at(ctypes.byref(uid), 128)]        # all 128 raw bytes (a c_char array converts only up to a NUL)
            dist.broadcast_object_list(box, src=0, group=group)
            _uid_bytes = box[0]
        ctypes.memmove(ctypes.byref(uid), _uid_bytes, 128)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.dev):
            self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank))
        # a HIGH-PRIORITY stream: it is served by a hardware queue of its own.  A default-priority torch stream can land on the queue
        # the compute stream uses, and then the collective (and the event wait in front of it) sits IN FRONT of the next kernel
        # launch instead of beside it: measured 77.9 vs 65.1 us/step at world size 1 (tools/rccl_step_cost.py)
        self.stream = torch.cuda.Stream(device=self.dev, priority=-1)
        self._ready = torch.cuda.Event()
        self._done = {}          # data_ptr -> (send tensor, completion event): events are created ONCE per buffer (see gather)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f'RCCL error {rc}: {self.lib.ncclGetErrorString(rc).decode()}')

    def nranks(self):
        """Number of ranks in the communicator as RCCL itself reports it (ncclCommCount)."""
        n = ctypes.c_int(0)
        self._check(self.lib.ncclCommCount(self.comm, ctypes.byref(n)))
        return int(n.value)

    def gather(self, send, recv, after=None):
        """after: the event that marks `send` complete (e.g. PnPPipeline.submit's return value, when the producer ran on another
        stream); None = everything enqueued so far on the CURRENT stream."""
        assert send.dtype == torch.uint8 and recv.dtype == torch.uint8 and recv.numel() == self.world * send.numel()
        if after is None:
            self._ready.record(torch.cuda.current_stream(self.dev))
            after = self._ready
        self.stream.wait_event(after)
        with torch.cuda.device(self.dev):
            self._check(self.lib.ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), self.NCCL_UINT8, self.comm,
                                               self.stream.cuda_stream))
        # one completion event per send buffer, re-recorded at every use: creating (and dropping) an event per call costs the
        # COMPUTE stream ~12 us per step on this stack (tools/rccl_step_cost.py: 77.9 -> 65.1 us/step at world size 1).
        # The table is keyed by the buffer's address AND holds the tensor, which pins the allocation: the address cannot be
        # freed and handed to another tensor (which would then inherit a stale event) while its entry exists; `forget` drops it.
        ent = self._done.get(send.data_ptr())
        if ent is None or ent[0].numel() != send.numel():
            ent = self._done[send.data_ptr()] = (send, torch.cuda.Event())
        ent[1].record(self.stream)
        return ent[1]

    def forget(self, send):
        """Drop the completion event (and the reference) kept for a send buffer that will not be used again."""
        self._done.pop(send.data_ptr(), None)

    def close(self):
        if self.comm:
            self.stream.synchronize()
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()
