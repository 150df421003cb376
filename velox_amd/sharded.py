"""One track (or a corpus of tracks) across the GPUs of a node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Sub-blocks are independent — every predictor / entropy state is reset per sub-block and stream
(VeloxCore.h:94-96, :141-142) — so the reference's thread-pool fan-out over sub-blocks
(VeloxCore.h:255-296) becomes contiguous runs of whole sub-blocks, one per rank.  What
``ProcessBlock`` computes over the whole input is the only thing exchanged:

* ``high_res_mode`` (VeloxCore.h:232-236): a 1-int MAX all-reduce,
* the global offset table (the in-order join of VeloxCore.h:299-303): an all-gather of the
  per-chunk ``uint32`` byte counts — every rank then knows where its bytes sit in the payload,
* optionally the payload itself: each rank sends its bytes straight into their final place in the
  root's buffer (point-to-point, one xGMI link per peer — no ring), and the root ORs the one byte
  two neighbouring shards share.

Nothing here does codec arithmetic; the per-shard work is handed in as callables so that the same
exchange logic runs over the HIP library (``GpuShard``) and, in the CPU tests, over shards cut out
of a reference payload.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._lib import SUB_BLOCK


def shard_ranges(n_samples, world):
    """Contiguous runs of whole sub-blocks: [(first_sample, n_samples)] per rank (VeloxCore.h:255-258)."""
    n_chunks = (int(n_samples) + SUB_BLOCK - 1) // SUB_BLOCK
    base, extra = divmod(n_chunks, world)
    out, c0 = [], 0
    for r in range(world):
        nc = base + (1 if r < extra else 0)
        s0 = c0 * SUB_BLOCK
        s1 = min((c0 + nc) * SUB_BLOCK, int(n_samples))
        out.append((min(s0, int(n_samples)), max(0, s1 - s0)))
        c0 += nc
    return out


def offset_table(chunk_sizes_per_rank):
    """Byte offset of every rank's shard in the whole payload, each shard's byte length, and the
    payload length.  A chunk takes its u32 size prefix plus its bytes (VeloxCore.h:301-302); the
    stream header takes 2 bits, so a shard of B body bytes spans B + 1 bytes and shares its first
    byte with its predecessor's last."""
    body = [int(np.asarray(s, dtype=np.uint64).sum()) + 4 * len(s) for s in chunk_sizes_per_rank]
    off = np.concatenate(([0], np.cumsum(body)[:-1])).astype(np.int64)
    seg_len = np.asarray(body, dtype=np.int64) + 1
    return off, seg_len, int(sum(body)) + 1


class ShardedPayload:
    """What one rank holds after a sharded encode: its own shard bytes plus the global table."""

    def __init__(self, rank, world, seg, sizes_per_rank, high_res):
        self.rank, self.world = rank, world
        self.seg = seg                                  # uint8 tensor, this rank's shard payload
        self.sizes_per_rank = sizes_per_rank            # list of np.uint32 arrays (all ranks)
        self.high_res = int(high_res)
        self.off, self.seg_len, self.total_len = offset_table(sizes_per_rank)

    @property
    def local_sizes(self):
        return self.sizes_per_rank[self.rank]


def _all_gather_sizes(sizes, device, group):
    """All-gather of variable-length uint32 chunk-size lists (the offset-table exchange)."""
    world = dist.get_world_size(group)
    cnt = torch.tensor([len(sizes)], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    width = max(max(counts), 1)
    mine = torch.zeros(width, dtype=torch.int64, device=device)
    if len(sizes):
        mine[: len(sizes)] = torch.from_numpy(np.asarray(sizes, dtype=np.int64)).to(device)
    table = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(table, mine, group=group)
    return [table[r][: counts[r]].cpu().numpy().astype(np.int64) for r in range(world)]


def encode_track_sharded(scan_fn, encode_shard_fn, device, group=None):
    """Sharded ``Encoder::ProcessBlock`` of one integer-PCM track.

    scan_fn()                         -> 0/1, "any |s| > 65536" over this rank's samples
    encode_shard_fn(high_res, first)  -> (uint8 tensor with the shard payload, np.uint32 chunk sizes)
    Returns a ShardedPayload (collective call: every rank of the group must enter)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    flag = torch.tensor([int(scan_fn())], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    high_res = int(flag.item())
    seg, sizes = encode_shard_fn(high_res, rank == 0)
    sizes_per_rank = [a.astype(np.uint32) for a in _all_gather_sizes(sizes, device, group)]
    sp = ShardedPayload(rank, world, seg, sizes_per_rank, high_res)
    assert int(seg.numel()) == int(sp.seg_len[rank]), "shard payload length disagrees with its chunk sizes"
    return sp


def gather_payload(sp, root=0, group=None):
    """Materialise the whole payload on ``root`` (None elsewhere): every rank's bytes go straight
    into their final place, then the root ORs in the shared seam bytes."""
    rank, world = sp.rank, sp.world
    device = sp.seg.device
    seam = sp.seg[:1].clone() if sp.seg.numel() else torch.zeros(1, dtype=torch.uint8, device=device)
    seams = [torch.zeros_like(seam) for _ in range(world)]
    dist.all_gather(seams, seam, group=group)
    if rank == root:
        final = torch.zeros(sp.total_len, dtype=torch.uint8, device=device)
        ops = []
        for r in range(world):
            o, ln = int(sp.off[r]), int(sp.seg_len[r])
            if ln <= 1:
                continue
            if r == rank:
                final[o + 1: o + ln] = sp.seg[1:]
            else:
                ops.append(dist.P2POp(dist.irecv, final[o + 1: o + ln], r, group=group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for r in range(world):                          # each shard's first byte lands on its predecessor's last
            final[int(sp.off[r])] |= seams[r][0]
        return final
    if int(sp.seg_len[rank]) > 1:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, sp.seg[1:].contiguous(), root, group=group)]):
            w.wait()
    return None


def merge_shards_host(segs, sizes_per_rank):
    """The same placement on the host (numpy), for single-process use and tests."""
    off, seg_len, total = offset_table(sizes_per_rank)
    final = np.zeros(total, dtype=np.uint8)
    for r, seg in enumerate(segs):
        seg = np.asarray(seg, dtype=np.uint8)
        assert seg.size == seg_len[r]
        final[off[r] + 1: off[r] + seg.size] = seg[1:]
        final[off[r]] |= seg[0]
    return final


def decode_ranges(chunks, world):
    """SURVEY §8e, decode side: the chunk list of a payload (one host walk of the size prefixes,
    ``codec.walk_chunks``) cut into contiguous runs, one per rank.  Output offsets are static (8192 samples
    per chunk, VeloxCore.h:346-347), so nothing is exchanged.  Returns, per rank,
    ``(c_lo, c_hi, byte_lo, byte_hi, out_lo, out_n)``: its chunks, the payload bytes that hold them (from the
    byte of its first u32 size prefix) and its slice of the track's samples."""
    n = len(chunks)
    base, extra = divmod(n, world)
    out, c0 = [], 0
    for r in range(world):
        c1 = c0 + base + (1 if r < extra else 0)
        if c1 > c0:
            # A chunk whose u32 size prefix was cut by the end of the data sits at bit_off = end of data with
            # n_bits = 0 (vlx_walk_chunks): its prefix does not lie 32 bits in front of it.  Only the LAST chunk
            # of a walk can be such a chunk; starting the range up to 4 bytes early is harmless (shard_chunks
            # re-bases every bit_off onto byte_lo), a negative byte_lo is not.
            byte_lo = max(0, (int(chunks["bit_off"][c0]) - 32) >> 3)
            byte_hi = (int(chunks["bit_off"][c1 - 1]) + int(chunks["n_bits"][c1 - 1]) + 7) >> 3
            out_lo = int(chunks["out_off"][c0])
            out_n = int(chunks["out_off"][c1 - 1]) + int(chunks["n_out"][c1 - 1]) - out_lo
        else:
            byte_lo = byte_hi = out_lo = out_n = 0
        out.append((c0, c1, byte_lo, byte_hi, out_lo, out_n))
        c0 = c1
    return out


def shard_chunks(chunks, rng):
    """A rank's chunks re-based onto its own byte range and its own output slice."""
    c_lo, c_hi, byte_lo, _, out_lo, _ = rng
    ch = np.array(chunks[c_lo:c_hi], copy=True)
    ch["bit_off"] -= np.uint64(8 * byte_lo)
    ch["out_off"] -= np.uint64(out_lo)
    return ch


def shard_as_payload(payload, chunks, rng):
    """The same byte range as a payload of its own: every chunk's size prefix sits at bit 2 of a byte
    (2 header bits + whole bytes, VeloxCore.h:226-236, :301-302), so the range, with the two header bits
    (is_float = 0, high_res) put into the low bits of its first byte, is what an encoder would have produced
    for these sub-blocks alone.  Used by the CPU tests, which decode shards with the oracle."""
    _, _, byte_lo, byte_hi, _, _ = rng
    seg = np.array(np.asarray(payload, dtype=np.uint8)[byte_lo:byte_hi], copy=True)
    if seg.size:
        seg[0] = (seg[0] & 0xFC) | ((int(chunks["high_res"][0]) & 1) << 1)
    return seg


def decode_track_sharded(ctx, payload, chunks, rank, world, out_fmt, device):
    """This rank's share of a track's decode on the HIP library: only its byte range goes up to its GPU.
    Returns (first sample index, uint8 tensor of its samples in `out_fmt`)."""
    from ._lib import FMT_I64
    rng = decode_ranges(chunks, world)[rank]
    _, _, byte_lo, byte_hi, out_lo, out_n = rng
    w = 8 if out_fmt == FMT_I64 else out_fmt // 8
    out = torch.zeros(out_n * w + 16, dtype=torch.uint8, device=device)
    if out_n:
        seg = torch.from_numpy(np.ascontiguousarray(np.asarray(payload, dtype=np.uint8)[byte_lo:byte_hi])).to(device)
        torch.cuda.synchronize(device)                  # the context's stream is not ordered against torch's
        ctx.decode_device(seg.data_ptr(), seg.numel(), shard_chunks(chunks, rng), out.data_ptr(), out_fmt)
    return out_lo, out[: out_n * w]


class NativeNode:
    """The same exchange steps done natively: libvelox_amd.so calls RCCL itself (vlx_node_*, include/velox_amd.h),
    the reduced high_res flag never leaves HBM.  One per rank; needs a GPU per rank.  `uid` = the 128 id bytes rank 0
    made with NativeNode.unique_id() and every rank received (e.g. through torch.distributed.broadcast)."""

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = (C.c_uint8 * 128)()
        rc = _lib.load().vlx_node_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"vlx_node_unique_id -> {_lib.ERR_NAMES.get(rc, rc)} (no RCCL?)")
        return np.frombuffer(bytes(buf), dtype=np.uint8).copy()

    def __init__(self, ctx, uid, rank, world, local_group=None, collectives=None):
        """uid: RCCL id bytes (one process per GPU).  local_group: a LocalGroup instead (ranks = threads of this
        process, vlx_node_init_local).  collectives: a _lib.VlxCollectives instead (the caller's own transport,
        vlx_node_init_custom; keep it alive as long as the node)."""
        import ctypes as C
        self.ctx, self.rank, self.world, self.L = ctx, int(rank), int(world), ctx.L
        h = C.c_void_p()
        if local_group is not None:
            ctx._chk(self.L.vlx_node_init_local(ctx.h, local_group.h, self.rank, C.byref(h)))
        elif collectives is not None:
            self._coll = collectives
            ctx._chk(self.L.vlx_node_init_custom(ctx.h, C.byref(collectives), self.rank, self.world, C.byref(h)))
        else:
            u = np.ascontiguousarray(uid, dtype=np.uint8)
            ctx._chk(self.L.vlx_node_init(ctx.h, u.ctypes.data_as(C.POINTER(C.c_uint8)), self.rank, self.world, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.vlx_node_destroy(self.h)
            self.h = None

    def encode_track(self, pcm_ptr, fmt, n_samples, payload_ptr, payload_cap, max_chunks_per_rank=0):
        """-> (shard_len, shard_off, total_len, high_res); collective."""
        import ctypes as C
        a, b, c_, hr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        self.ctx._chk(self.L.vlx_node_encode_track(self.h, C.c_void_p(int(pcm_ptr)), int(fmt), int(n_samples), int(max_chunks_per_rank),
                                                   C.c_void_p(int(payload_ptr)), int(payload_cap), C.byref(a), C.byref(b), C.byref(c_), C.byref(hr)))
        return int(a.value), int(b.value), int(c_.value), int(hr.value)

    def chunk_sizes(self, rank=None):
        import ctypes as C
        r = self.rank if rank is None else int(rank)
        n = C.c_size_t(0)
        self.ctx._chk(self.L.vlx_node_chunk_sizes(self.h, r, None, 0, C.byref(n)))
        out = np.empty(max(n.value, 1), dtype=np.uint32)
        self.ctx._chk(self.L.vlx_node_chunk_sizes(self.h, r, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size, C.byref(n)))
        return out[: n.value]

    def gather_payload(self, root, shard_ptr, shard_len, full_ptr=0, full_cap=0):
        import ctypes as C
        self.ctx._chk(self.L.vlx_node_gather_payload(self.h, int(root), C.c_void_p(int(shard_ptr)), int(shard_len),
                                                     C.c_void_p(int(full_ptr)) if full_ptr else None, int(full_cap)))

    def all_gather_lengths(self, lengths, max_per_rank):
        import ctypes as C
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        table = np.zeros((self.world, int(max_per_rank)), dtype=np.uint64)
        counts = np.zeros(self.world, dtype=np.uint32)
        self.ctx._chk(self.L.vlx_node_all_gather_lengths(self.h, ln.ctypes.data_as(C.POINTER(C.c_uint64)), ln.size, int(max_per_rank),
                                                         table.ctypes.data_as(C.POINTER(C.c_uint64)), counts.ctypes.data_as(C.POINTER(C.c_uint32))))
        return [table[r, : counts[r]].copy() for r in range(self.world)]


class TorchTransport:
    """The six collectives vlx_node_* needs (include/velox_amd.h: vlx_collectives) carried by torch.distributed — any
    backend, gloo on the CPU included: device buffers are staged through host memory with vlx_node_copy.  Not a fast
    path (RCCL inside the library is): it exists so that the native node layer runs with several ranks wherever
    torch.distributed does — several processes sharing one GPU in tests and rehearsals of `bench.py --gpus N`.
    Hand `.table` to NativeNode(collectives=...) and keep this object alive as long as the node."""

    def __init__(self, group=None):
        import ctypes as C
        from . import _lib
        self.L = _lib.load()
        self.group, self.world = group, dist.get_world_size(group)
        self.pending = []
        self._fns = (_lib.COLL_ALL_REDUCE(self._all_reduce), _lib.COLL_ALL_GATHER(self._all_gather), _lib.COLL_SEND(self._send),
                     _lib.COLL_RECV(self._recv), _lib.COLL_GROUP_START(self._group_start), _lib.COLL_GROUP_END(self._group_end))
        self.table = _lib.VlxCollectives(None, *self._fns)
        self._C = C

    def _d2h(self, ptr, n, st):
        buf = np.empty(max(n, 1), dtype=np.uint8)
        if self.L.vlx_node_copy(buf.ctypes.data, ptr, n, st) != 0:
            raise RuntimeError("vlx_node_copy (device -> host) failed")
        return buf[:n]

    def _h2d(self, ptr, arr, st):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        if self.L.vlx_node_copy(ptr, arr.ctypes.data, arr.size, st) != 0:
            raise RuntimeError("vlx_node_copy (host -> device) failed")

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:  # noqa: BLE001 - a callback must not raise through the C frames
            print(f"[TorchTransport] {type(e).__name__}: {e}", flush=True)
            return 1

    def _all_reduce(self, user, d, n_words, st):
        def go():
            t = torch.from_numpy(self._d2h(d, 4 * n_words, st).view(np.uint32).astype(np.int64))
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            self._h2d(d, t.numpy().astype(np.uint32).view(np.uint8), st)
        return self._guard(go)

    def _all_gather(self, user, d_send, d_recv, nbytes, st):
        def go():
            mine = torch.from_numpy(self._d2h(d_send, nbytes, st).copy())
            parts = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
            self._h2d(d_recv, torch.cat(parts).numpy(), st)
        return self._guard(go)

    def _send(self, user, d, nbytes, peer, st):
        self.pending.append(("s", d, nbytes, peer))
        return 0

    def _recv(self, user, d, nbytes, peer, st):
        self.pending.append(("r", d, nbytes, peer))
        return 0

    def _group_start(self, user):
        self.pending.clear()
        return 0

    def _group_end(self, user, st):
        def go():
            ops, bufs = [], []
            for kind, d, nbytes, peer in self.pending:
                if kind == "s":
                    ops.append(dist.P2POp(dist.isend, torch.from_numpy(self._d2h(d, nbytes, st).copy()), peer, group=self.group))
                else:
                    t = torch.zeros(nbytes, dtype=torch.uint8)
                    ops.append(dist.P2POp(dist.irecv, t, peer, group=self.group))
                    bufs.append((d, t))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            for d, t in bufs:
                self._h2d(d, t.numpy(), st)
            self.pending.clear()
        return self._guard(go)


class LocalGroup:
    """vlx_local_group: the meeting point of `world` ranks that are threads of this process."""

    def __init__(self, world, timeout_s=None):
        import ctypes as C
        from . import _lib
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.vlx_local_group_create(int(world), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"vlx_local_group_create -> {rc}")
        self.h, self.world = h, int(world)
        if timeout_s:
            self.L.vlx_local_group_set_timeout(h, int(timeout_s))

    def close(self):
        if getattr(self, "h", None):
            self.L.vlx_local_group_destroy(self.h)
            self.h = None


def native_decode_range(chunks, rank, world):
    """vlx_node_decode_range: the same cut as decode_ranges()[rank], computed by the library."""
    import ctypes as C
    from . import _lib
    from .codec import CHUNK_DTYPE
    ch = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
    r = _lib.VlxDecodeRange()
    rc = _lib.load().vlx_node_decode_range(ch.ctypes.data_as(C.POINTER(_lib.VlxChunk)), ch.size, int(rank), int(world), C.byref(r))
    if rc != 0:
        raise RuntimeError(f"vlx_node_decode_range -> {rc}")
    return (int(r.chunk_lo), int(r.chunk_hi), int(r.byte_lo), int(r.byte_hi), int(r.out_lo), int(r.out_n))


def native_decode_shard(ctx, payload, chunks, rank, world, out_ptr, out_fmt):
    """vlx_node_decode_shard: this rank's share of a track's decode (only its byte range goes up to its GPU; out_ptr
    receives its own slice).  Returns the range tuple of native_decode_range."""
    import ctypes as C
    from . import _lib
    from .codec import CHUNK_DTYPE
    ch = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
    p = np.ascontiguousarray(payload, dtype=np.uint8)
    r = _lib.VlxDecodeRange()
    ctx._chk(ctx.L.vlx_node_decode_shard(ctx.h, p.ctypes.data_as(C.POINTER(C.c_uint8)), p.size, ch.ctypes.data_as(C.POINTER(_lib.VlxChunk)), ch.size,
                                         int(rank), int(world), C.c_void_p(int(out_ptr)), int(out_fmt), C.byref(r)))
    return (int(r.chunk_lo), int(r.chunk_hi), int(r.byte_lo), int(r.byte_hi), int(r.out_lo), int(r.out_n))


def native_shard_range(total_samples, rank, world):
    """vlx_node_shard_range: the same cut as shard_ranges()[rank], computed by the library."""
    import ctypes as C
    from . import _lib
    a, b = C.c_uint64(0), C.c_uint64(0)
    _lib.load().vlx_node_shard_range(int(total_samples), int(rank), int(world), C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def native_offset_table(chunk_sizes_per_rank):
    """vlx_node_offset_table (host arithmetic only): same answer as offset_table()."""
    import ctypes as C
    from . import _lib
    world = len(chunk_sizes_per_rank)
    stride = max([len(s) for s in chunk_sizes_per_rank] + [1])
    rows = np.zeros((world, stride), dtype=np.uint32)
    counts = np.zeros(world, dtype=np.uint32)
    for r, sz in enumerate(chunk_sizes_per_rank):
        counts[r] = len(sz)
        rows[r, : len(sz)] = np.asarray(sz, dtype=np.uint32)
    off = np.zeros(world, dtype=np.uint64); seg = np.zeros(world, dtype=np.uint64); total = C.c_uint64(0)
    rc = _lib.load().vlx_node_offset_table(counts.ctypes.data_as(C.POINTER(C.c_uint32)), rows.ctypes.data_as(C.POINTER(C.c_uint32)), world, stride,
                                           off.ctypes.data_as(C.POINTER(C.c_uint64)), seg.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(total))
    if rc != 0:
        raise RuntimeError(f"vlx_node_offset_table -> {rc}")
    return off.astype(np.int64), seg.astype(np.int64), int(total.value)


def all_gather_lengths(lengths, device, group=None):
    """Corpus mode: whole tracks per rank, no chunk crosses a rank.  The only exchange is the table of
    payload lengths (an archive's offset table): returns the per-rank np.int64 length lists."""
    return _all_gather_sizes(np.asarray(lengths, dtype=np.int64), device, group)


class GpuShard:
    """The per-rank work of encode_track_sharded on the HIP library: PCM and payload stay in HBM."""

    def __init__(self, ctx, pcm, fmt, n_samples):
        self.ctx, self.pcm, self.fmt, self.n = ctx, pcm, int(fmt), int(n_samples)
        self.cap = ctx.encode_bound([self.n])
        self.payload = torch.empty(self.cap, dtype=torch.uint8, device=pcm.device)

    def scan(self):
        return self.ctx.scan_high_res(self.pcm.data_ptr(), self.fmt, self.n)

    def encode(self, high_res, first):
        ln = self.ctx.encode_shard(self.pcm.data_ptr(), self.fmt, self.n, high_res, first, self.payload.data_ptr(), self.cap)
        return self.payload[:ln], self.ctx.last_chunk_sizes()
