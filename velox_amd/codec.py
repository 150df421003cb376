"""Python host mirror of the reference's two entry points, over the C-ABI.

``Encoder.ProcessBlock`` / ``StreamingDecoder.DecodeNext`` keep the reference's names and argument
meaning (VeloxCore.h:210, :322, :335) so the parity tests read like a caller of the reference.
``Context.encode_device`` / ``decode_device`` are the HBM-resident batch entry points bench.py
measures; they take torch CUDA tensors only as device memory (data_ptr), never as a compute
backend.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FMT_I64, FMT_S16, FMT_S24, FMT_S32, SUB_BLOCK, VlxChunk

_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)

CHUNK_DTYPE = np.dtype([("bit_off", "<u8"), ("n_bits", "<u8"), ("out_off", "<u8"), ("frames", "<u4"),
                        ("n_out", "<u4"), ("high_res", "<u4"), ("reserved", "<u4")])
assert CHUNK_DTYPE.itemsize == C.sizeof(VlxChunk)
TRACK_SIZES_DTYPE = np.dtype([("base_bit_off", "<u8"), ("out_base", "<u8"), ("total_samples", "<u8"), ("n_sizes", "<u4"),
                              ("high_res", "<u4"), ("hdr_bits", "<u4"), ("reserved", "<u4")])
assert TRACK_SIZES_DTYPE.itemsize == C.sizeof(_lib.VlxTrackSizes)
HIGH_RES_FROM_PAYLOAD = 0xFFFFFFFF       # vlx_track_sizes.high_res: let the device read the bit from the payload


class VeloxError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"{_lib.ERR_NAMES.get(code, code)}: {msg}")


class Context:
    """One GPU + one HIP stream + a grow-only device workspace (replaces the reference's
    process-wide ThreadPool, VeloxCore.h:208)."""

    def __init__(self, device=0, stream=None):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.vlx_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise VeloxError(rc, "vlx_ctx_create (no GPU? velox_amd has no CPU fallback)")
        self.h = h
        self.device = int(device)
        if stream is not None:
            self.L.vlx_ctx_set_stream(self.h, C.c_void_p(int(stream)))

    def close(self):
        if getattr(self, "h", None):
            self.L.vlx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, encode_chunks=0, encode_tracks=1, decode_chunks=0, fmt=FMT_S24):
        """Size the workspace up front (vlx_ctx_reserve_for): later device batches of that size and format allocate nothing."""
        self._chk(self.L.vlx_ctx_reserve_for(self.h, int(fmt), int(encode_chunks), int(encode_tracks), int(decode_chunks)))

    def trim(self):
        """Give every workspace buffer back (vlx_ctx_trim)."""
        self._chk(self.L.vlx_ctx_trim(self.h))

    def workspace_bytes(self):
        return int(self.L.vlx_ctx_workspace_bytes(self.h))

    def _chk(self, rc):
        if rc < 0:
            raise VeloxError(rc, (self.L.vlx_last_error(self.h) or b"").decode())
        return rc

    # ---- host-pointer drop-ins --------------------------------------------------------------
    def encode_block(self, samples, is_float=False, exps=None, raw=None):
        s = np.ascontiguousarray(samples, dtype=np.int64).copy()
        e = np.ascontiguousarray(exps, dtype=np.uint8) if exps is not None else None
        r = np.ascontiguousarray(raw, dtype=np.uint8) if raw is not None else None
        out, n = _u8p(), C.c_size_t(0)
        self._chk(self.L.vlx_encode_block(
            self.h, s.ctypes.data_as(_i64p), s.size, int(bool(is_float)),
            e.ctypes.data_as(_u8p) if e is not None else None, 0 if e is None else e.size,
            r.ctypes.data_as(_u8p) if r is not None else None, C.byref(out), C.byref(n)))
        res = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[:n.value].copy()
        self.L.vlx_free(out)
        return res

    def encode_pcm(self, pcm_bytes, bits):
        p = np.ascontiguousarray(pcm_bytes, dtype=np.uint8)
        out, n = _u8p(), C.c_size_t(0)
        self._chk(self.L.vlx_encode_pcm(self.h, p.ctypes.data_as(_u8p), p.size // (bits // 8), bits,
                                        C.byref(out), C.byref(n)))
        res = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[:n.value].copy()
        self.L.vlx_free(out)
        return res

    # ---- device-resident batch API ------------------------------------------------------------
    def decode_pcm(self, payload, total, bits):
        """vlx_decode_pcm: integer-PCM payload (host bytes) -> (packed little-endian PCM bytes of `total`
        samples, number of samples the stream actually reached)."""
        p = np.ascontiguousarray(np.frombuffer(payload, dtype=np.uint8) if isinstance(payload, (bytes, bytearray, memoryview)) else payload, dtype=np.uint8)
        out = np.empty(int(total) * (bits // 8), dtype=np.uint8)
        n = C.c_size_t(0)
        self._chk(self.L.vlx_decode_pcm(self.h, p.ctypes.data_as(_u8p), p.size, int(total), int(bits),
                                        out.ctypes.data_as(_u8p), out.size, C.byref(n)))
        return out, int(n.value)

    def encode_bound(self, track_samples):
        ns = np.ascontiguousarray(track_samples, dtype=np.uint64)
        return int(self.L.vlx_encode_bound(ns.ctypes.data_as(_u64p), ns.size))

    def encode_device(self, pcm_ptr, fmt, track_byte_off, track_samples, payload_ptr, payload_cap):
        off = np.ascontiguousarray(track_byte_off, dtype=np.uint64)
        ns = np.ascontiguousarray(track_samples, dtype=np.uint64)
        poff = np.zeros(ns.size, dtype=np.uint64)
        plen = np.zeros(ns.size, dtype=np.uint64)
        self._chk(self.L.vlx_encode_device(self.h, C.c_void_p(int(pcm_ptr)), int(fmt), off.ctypes.data_as(_u64p),
                                           ns.ctypes.data_as(_u64p), ns.size, C.c_void_p(int(payload_ptr)),
                                           int(payload_cap), poff.ctypes.data_as(_u64p), plen.ctypes.data_as(_u64p)))
        return poff, plen

    # ---- float32 tracks, device-resident ------------------------------------------------------
    def encode_f32_device(self, f32_ptr, n_samples, payload_ptr, payload_cap):
        """vlx_encode_f32_device: float32 samples in HBM -> payload at payload_ptr.  Returns (payload bytes, float_mode)."""
        n = C.c_uint64(0)
        fm = C.c_int(0)
        self._chk(self.L.vlx_encode_f32_device(self.h, C.c_void_p(int(f32_ptr)), int(n_samples), C.c_void_p(int(payload_ptr)),
                                               int(payload_cap), C.byref(n), C.byref(fm)))
        return int(n.value), int(fm.value)

    def decode_f32_device(self, payload_ptr, payload_bytes, total, sizes, out_ptr):
        """vlx_decode_f32_device: a float track's payload in HBM -> float32 samples at out_ptr.
        Returns (float_mode, chunks decoded)."""
        sz = np.ascontiguousarray(sizes, dtype=np.uint32)
        fm = C.c_int(0)
        n = C.c_uint64(0)
        self._chk(self.L.vlx_decode_f32_device(self.h, C.c_void_p(int(payload_ptr)), int(payload_bytes), int(total),
                                               sz.ctypes.data_as(C.POINTER(C.c_uint32)), sz.size, C.c_void_p(int(out_ptr)),
                                               C.byref(fm), C.byref(n)))
        return int(fm.value), int(n.value)

    # ---- multi-GPU building blocks (include/velox_amd.h) ---------------------------------------
    def scan_high_res(self, pcm_ptr, fmt, n_samples):
        flag = C.c_int(0)
        self._chk(self.L.vlx_scan_high_res_device(self.h, C.c_void_p(int(pcm_ptr)), int(fmt), int(n_samples), C.byref(flag)))
        return int(flag.value)

    def encode_shard(self, pcm_ptr, fmt, n_samples, high_res, first_shard, payload_ptr, payload_cap):
        n = C.c_uint64(0)
        self._chk(self.L.vlx_encode_shard_device(self.h, C.c_void_p(int(pcm_ptr)), int(fmt), int(n_samples), int(high_res),
                                                 int(bool(first_shard)), C.c_void_p(int(payload_ptr)), int(payload_cap), C.byref(n)))
        return int(n.value)

    def last_chunk_sizes(self):
        n = C.c_size_t(0)
        self._chk(self.L.vlx_encode_last_chunk_sizes(self.h, None, 0, C.byref(n)))
        out = np.empty(max(n.value, 1), dtype=np.uint32)
        self._chk(self.L.vlx_encode_last_chunk_sizes(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size, C.byref(n)))
        return out[:n.value]

    def decode_device(self, payload_ptr, payload_bytes, chunks, out_ptr, out_fmt):
        ch = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
        self._chk(self.L.vlx_decode_device(self.h, C.c_void_p(int(payload_ptr)), int(payload_bytes),
                                           ch.ctypes.data_as(C.POINTER(VlxChunk)), ch.size,
                                           C.c_void_p(int(out_ptr)), int(out_fmt)))

    def decode_device_noalloc(self, payload_ptr, payload_bytes, chunks, out_ptr, out_fmt):
        """vlx_decode_device_noalloc: the same call over the library's file-scope device scratch (<= 1024 chunks)."""
        ch = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
        self._chk(self.L.vlx_decode_device_noalloc(self.h, C.c_void_p(int(payload_ptr)), int(payload_bytes),
                                                   ch.ctypes.data_as(C.POINTER(VlxChunk)), ch.size,
                                                   C.c_void_p(int(out_ptr)), int(out_fmt)))

    def decode_tracks_device(self, payload_ptr, payload_bytes, tracks, sizes, out_ptr, out_fmt):
        """vlx_decode_tracks_device: `tracks` is a TRACK_SIZES_DTYPE array (one entry per track), `sizes`
        the tracks' chunk byte sizes back to back; the chunk table is built on the device.  Returns the
        number of chunks decoded."""
        tr = np.ascontiguousarray(tracks, dtype=TRACK_SIZES_DTYPE)
        sz = np.ascontiguousarray(sizes, dtype=np.uint32)
        n = C.c_size_t(0)
        self._chk(self.L.vlx_decode_tracks_device(self.h, C.c_void_p(int(payload_ptr)), int(payload_bytes),
                                                  tr.ctypes.data_as(C.POINTER(_lib.VlxTrackSizes)), tr.size,
                                                  sz.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_void_p(int(out_ptr)), int(out_fmt),
                                                  C.byref(n)))
        return int(n.value)

    def set_profiling(self, on=True):
        self.L.vlx_set_profiling(self.h, int(on))

    def last_redo_chunks(self):
        """Diagnostic: chunks of the last decode batch that the careful kernel had to decode again (0 on ordinary material)."""
        f = self.L.vlx_debug_last_redo_chunks
        f.argtypes, f.restype = [C.c_void_p], C.c_long
        return int(f(self.h))

    def last_kernel_times(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = self.L.vlx_last_kernel_times(self.h, names, ms, 32)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]


class Pool:
    """vlx_pool: the host-pointer drop-ins over several GPUs from one caller thread (one worker thread + one context
    per entry of `devices`; an entry may repeat: several shards in flight on one GPU).  Same bytes as Context's calls."""

    def __init__(self, devices):
        self.L = _lib.load()
        dev = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self.L.vlx_pool_create(dev, len(devices), C.byref(h))
        if rc != 0:
            raise VeloxError(rc, "vlx_pool_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.vlx_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self.L.vlx_pool_size(self.h))

    def _chk(self, rc):
        if rc < 0:
            raise VeloxError(rc, (self.L.vlx_pool_last_error(self.h) or b"").decode())
        return rc

    def _take(self, out, n):
        res = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[:n.value].copy()
        self.L.vlx_free(out)
        return res

    def encode_pcm(self, pcm_bytes, bits):
        p = np.ascontiguousarray(pcm_bytes, dtype=np.uint8)
        out, n = _u8p(), C.c_size_t(0)
        self._chk(self.L.vlx_pool_encode_pcm(self.h, p.ctypes.data_as(_u8p), p.size // (bits // 8), bits, C.byref(out), C.byref(n)))
        return self._take(out, n)

    def encode_block(self, samples, is_float=False, exps=None, raw=None):
        s = np.ascontiguousarray(samples, dtype=np.int64).copy()
        e = np.ascontiguousarray(exps, dtype=np.uint8) if exps is not None else None
        r = np.ascontiguousarray(raw, dtype=np.uint8) if raw is not None else None
        out, n = _u8p(), C.c_size_t(0)
        self._chk(self.L.vlx_pool_encode_block(
            self.h, s.ctypes.data_as(_i64p), s.size, int(bool(is_float)),
            e.ctypes.data_as(_u8p) if e is not None else None, 0 if e is None else e.size,
            r.ctypes.data_as(_u8p) if r is not None else None, C.byref(out), C.byref(n)))
        return self._take(out, n)

    def decode_pcm(self, payload, total, bits):
        p = np.ascontiguousarray(payload, dtype=np.uint8)
        out = np.empty(int(total) * (bits // 8), dtype=np.uint8)
        n = C.c_size_t(0)
        self._chk(self.L.vlx_pool_decode_pcm(self.h, p.ctypes.data_as(_u8p), p.size, int(total), int(bits),
                                             out.ctypes.data_as(_u8p), out.size, C.byref(n)))
        return out, int(n.value)


def set_decode_shape(shape):
    """Wave shape of the fast decode kernel, process-wide: "auto" (mix by cost), "stream", "chunk"."""
    rc = _lib.load().vlx_set_decode_shape({"auto": 0, "stream": 1, "chunk": 2, "pipe": 3, "pipe16": 4, "pipe64": 5, "pipe32": 6}[shape])
    if rc != 0:
        raise ValueError(f"vlx_set_decode_shape({shape!r}) -> {rc}")


def set_encode_shape(shape):
    """Shape of the sample-serial encode kernel, process-wide: "auto", "stream" (throughput shape), "pipe", "pipe16", "pipe64"."""
    rc = _lib.load().vlx_set_encode_shape({"auto": 0, "stream": 1, "pipe": 3, "pipe16": 4, "pipe64": 5, "pipe32": 6}[shape])
    if rc != 0:
        raise ValueError(f"vlx_set_encode_shape({shape!r}) -> {rc}")


def walk_chunks(payload, total, base_bit_off=0, out_base=0):
    """Host walk of one integer-PCM payload's size-prefixed chunk list (VeloxCore.h:338-353)."""
    L = _lib.load()
    p = np.ascontiguousarray(payload, dtype=np.uint8)
    cap = int(total) // SUB_BLOCK + 2
    out = np.zeros(cap, dtype=CHUNK_DTYPE)
    n = L.vlx_walk_chunks(p.ctypes.data_as(_u8p), p.size, int(total), int(base_bit_off), int(out_base),
                          out.ctypes.data_as(C.POINTER(VlxChunk)), cap)
    if n < 0:
        raise VeloxError(int(n), "vlx_walk_chunks")
    return out[:n]


def chunks_from_sizes(sizes, total, base_bit_off, out_base, high_res, hdr_bits=2, out=None):
    """Chunk table of one integer-PCM track from its known chunk byte sizes (what
    vlx_encode_last_chunk_sizes returns): vlx_chunks_from_sizes, no walk of the payload needed.
    `out` (a CHUNK_DTYPE array of at least len(sizes) entries) is filled in place when given."""
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    ch = out if out is not None and out.dtype == CHUNK_DTYPE and out.size >= sizes.size else np.zeros(sizes.size, dtype=CHUNK_DTYPE)
    n = _lib.load().vlx_chunks_from_sizes(sizes.ctypes.data_as(C.POINTER(C.c_uint32)), sizes.size, int(total),
                                          int(base_bit_off), int(out_base), int(high_res), int(hdr_bits),
                                          ch.ctypes.data_as(C.POINTER(VlxChunk)), ch.size)
    if n < 0:
        raise RuntimeError(f"vlx_chunks_from_sizes -> {n}")
    return ch[:n]


def refill_many(decoders):
    """One GPU batch for every decoder in the list that has run dry (BASELINE config 5)."""
    if not decoders:
        return
    arr = (C.c_void_p * len(decoders))(*[d.h for d in decoders])
    decoders[0].ctx._chk(decoders[0].L.vlx_decoder_refill_many(arr, len(decoders)))


class Encoder:
    """VeloxCodec::Encoder (VeloxCore.h:206-307)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or Context()

    def ProcessBlock(self, samples, is_float=False, exps=None, raw_bytes=None):
        return self.ctx.encode_block(samples, is_float, exps, raw_bytes)


class StreamingDecoder:
    """VeloxCodec::StreamingDecoder (VeloxCore.h:309-389): borrowed data pointer, pull API."""

    def __init__(self, data, total, ctx=None, lookahead=1, _handle=None, size=None):
        self.ctx = ctx or Context()
        self.L = self.ctx.L
        self._data = np.ascontiguousarray(data, dtype=np.uint8)       # keeps the borrowed bytes alive
        self.total = int(total)
        if _handle is not None:
            self.h = _handle
            return
        h = C.c_void_p()
        self.ctx._chk(self.L.vlx_decoder_open(self.ctx.h, self._data.ctypes.data_as(_u8p),
                                              self._data.size if size is None else int(size), self.total, C.byref(h)))
        self.h = h
        if lookahead != 1:
            self.L.vlx_decoder_set_lookahead(self.h, int(lookahead))

    def extend(self, new_size):
        self.ctx._chk(self.L.vlx_decoder_extend(self.h, int(new_size)))

    def bytes_needed(self, samples=1):
        """Bytes of the buffer that must be valid before the next `samples` samples can be read (0: none)."""
        n = C.c_size_t(0)
        self.ctx._chk(self.L.vlx_decoder_bytes_needed(self.h, int(samples), C.byref(n)))
        return int(n.value)

    def IsFloat(self):
        return bool(self.L.vlx_decoder_is_float(self.h))

    def GetFloatMode(self):
        return int(self.L.vlx_decoder_float_mode(self.h))

    def DecodeNext(self):
        """-> (ok, value, exp)"""
        v, e = C.c_int64(0), C.c_uint8(0)
        rc = self.ctx._chk(self.L.vlx_decoder_next(self.h, C.byref(v), C.byref(e)))
        return bool(rc), int(v.value), int(e.value)

    def read(self, cap):
        vals = np.zeros(max(cap, 1), dtype=np.int64)
        exps = np.zeros(max(cap, 1), dtype=np.uint8)
        n = self.ctx._chk(self.L.vlx_decoder_read(self.h, vals.ctypes.data_as(_i64p), exps.ctypes.data_as(_u8p), int(cap)))
        return vals[:n], exps[:n]

    def seek(self, sample):
        """Next sample handed out = number `sample` (the players' restart-and-fast-forward, without the decoding)."""
        self.ctx._chk(self.L.vlx_decoder_seek(self.h, int(sample)))

    def clone(self):
        h = C.c_void_p()
        self.ctx._chk(self.L.vlx_decoder_clone(self.h, C.byref(h)))
        return StreamingDecoder(self._data, self.total, ctx=self.ctx, _handle=h)

    def close(self):
        if getattr(self, "h", None):
            self.L.vlx_decoder_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
