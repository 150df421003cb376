"""ctypes bindings for the TEST oracles (never imported by the product package).

* ``Oracle``  — oracle/libvelox_oracle.so, the C restatement (built by ``make -C oracle``).
* ``Reference`` — oracle/_ref/libvelox_ref.so, the real reference compiled in the build container
  (present on the GPU box as a prebuilt artefact; tests that need it skip when it is absent).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libvelox_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libvelox_ref.so")

_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "velox_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvelox_oracle.so"],
                              stdout=subprocess.DEVNULL)


class StreamParams(C.Structure):
    _fields_ = [("silent", C.c_int32), ("shift_lsb", C.c_int32), ("lpc_shift", C.c_int32),
                ("coefs", C.c_int32 * 8)]


class Oracle:
    def __init__(self):
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.vlxo_encode_mt.restype = C.c_size_t
        L.vlxo_encode_mt.argtypes = [_i64p, C.c_size_t, C.c_int, _u8p, C.c_size_t, _u8p, C.c_int,
                                     C.POINTER(_u8p)]
        L.vlxo_free.argtypes = [C.c_void_p]
        L.vlxo_decode.restype = C.c_size_t
        L.vlxo_decode.argtypes = [_u8p, C.c_size_t, C.c_size_t, _i64p, _u8p, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]
        L.vlxo_decode_mt.restype = C.c_size_t
        L.vlxo_decode_mt.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_int, _i64p]
        L.vlxo_compute_lpc.argtypes = [_i64p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vlxo_encode_stream.restype = C.c_size_t
        L.vlxo_encode_stream.argtypes = [_i64p, C.c_size_t, C.c_int, _u8p, C.c_size_t,
                                         C.POINTER(StreamParams), C.POINTER(C.c_uint64)]
        L.vlxo_decode_stream.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_int, _i64p]
        L.vlxo_encode_chunk.restype = C.c_size_t
        L.vlxo_encode_chunk.argtypes = [_i64p, C.c_size_t, C.c_int, _u8p, C.c_size_t, C.POINTER(C.c_int)]
        L.vlxo_detect_pseudo_float.restype = C.c_int
        L.vlxo_detect_pseudo_float.argtypes = [_u8p, C.c_size_t]
        L.vlxo_split_float32.argtypes = [_u8p, C.c_size_t, _i64p, _u8p]
        L.vlxo_merge_float32.argtypes = [_i64p, _u8p, C.c_size_t, _u8p]
        L.vlxo_merge_float32.restype = None
        L.vlxo_promote_int_to_float.argtypes = [_i64p, C.c_size_t, C.c_int, _u8p]
        L.vlxo_promote_int_to_float.restype = None
        L.vlxo_bytes_to_samples.argtypes = [_u8p, C.c_size_t, C.c_int, _i64p]
        L.vlxo_samples_to_bytes.argtypes = [_i64p, C.c_size_t, C.c_int, _u8p]
        L.vlxo_verify_payload.argtypes = [_u8p, C.c_int, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_long)]
        L.vlxo_verify_payload.restype = C.c_long
        self.L = L

    def encode(self, samples, is_float=False, exps=None, raw=None, threads=1):
        s = np.ascontiguousarray(samples, dtype=np.int64).copy()
        out = _u8p()
        e = np.ascontiguousarray(exps, dtype=np.uint8) if exps is not None else None
        r = np.ascontiguousarray(raw, dtype=np.uint8) if raw is not None else None
        n = self.L.vlxo_encode_mt(_p(s, _i64p), s.size, int(is_float), _p(e, _u8p),
                                  0 if e is None else e.size, _p(r, _u8p), threads, C.byref(out))
        res = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy()
        self.L.vlxo_free(out)
        return res

    def decode(self, payload, total, want_meta=False):
        p = np.ascontiguousarray(payload, dtype=np.uint8)
        out = np.zeros(max(total, 1), dtype=np.int64)
        exps = np.zeros(max(total, 1), dtype=np.uint8)
        isf, fm = C.c_int(0), C.c_int(0)
        n = self.L.vlxo_decode(_p(p, _u8p), p.size, total, _p(out, _i64p), _p(exps, _u8p),
                               C.byref(isf), C.byref(fm))
        if want_meta:
            return out[:n], exps[:n], isf.value, fm.value
        return out[:n]

    def decode_mt(self, payload, total, threads):
        p = np.ascontiguousarray(payload, dtype=np.uint8)
        out = np.zeros(max(total, 1), dtype=np.int64)
        n = self.L.vlxo_decode_mt(_p(p, _u8p), p.size, total, threads, _p(out, _i64p))
        return out[:n]

    def verify_payload(self, pcm_bytes, bits, total, payload, threads=1):
        """EVERY chunk of a whole integer-PCM payload against encode_chunk of the same sub-block of the packed PCM
        (vlxo_verify_payload, `threads` host threads).  -> (chunks compared, mismatching chunks or -1 when the header /
        chunk list / length is wrong, index of the first bad chunk or -1)."""
        pcm = np.ascontiguousarray(pcm_bytes, dtype=np.uint8)
        p = np.ascontiguousarray(payload, dtype=np.uint8)
        n, first = C.c_size_t(0), C.c_long(-1)
        bad = self.L.vlxo_verify_payload(_p(pcm, _u8p), int(bits), int(total), _p(p, _u8p), p.size, int(threads), C.byref(n), C.byref(first))
        return int(n.value), int(bad), int(first.value)

    def compute_lpc(self, x, order=8):
        x = np.ascontiguousarray(x, dtype=np.int64)
        c = (C.c_int * order)()
        sh = C.c_int(0)
        self.L.vlxo_compute_lpc(_p(x, _i64p), x.size, order, c, C.byref(sh))
        return list(c), sh.value

    def encode_stream(self, x, high_res):
        x = np.ascontiguousarray(x, dtype=np.int64)
        cap = x.size * 14 + 64
        out = np.zeros(cap, dtype=np.uint8)
        P = StreamParams()
        nb = C.c_uint64(0)
        n = self.L.vlxo_encode_stream(_p(x, _i64p), x.size, int(high_res), _p(out, _u8p), cap,
                                      C.byref(P), C.byref(nb))
        return out[:n].copy(), P, nb.value

    def decode_stream(self, data, count, high_res):
        d = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(max(count, 1), dtype=np.int64)
        self.L.vlxo_decode_stream(_p(d, _u8p), d.size, count, int(high_res), _p(out, _i64p))
        return out[:count]

    def encode_chunk(self, inter, high_res):
        x = np.ascontiguousarray(inter, dtype=np.int64)
        cap = x.size * 14 + 64
        out = np.zeros(cap, dtype=np.uint8)
        ms = C.c_int(0)
        n = self.L.vlxo_encode_chunk(_p(x, _i64p), x.size // 2, int(high_res), _p(out, _u8p), cap, C.byref(ms))
        return out[:n].copy(), ms.value

    def detect_pseudo_float(self, raw):
        r = np.ascontiguousarray(raw, dtype=np.uint8)
        return self.L.vlxo_detect_pseudo_float(_p(r, _u8p), r.size // 4)

    def split_float32(self, raw):
        r = np.ascontiguousarray(raw, dtype=np.uint8)
        n = r.size // 4
        m = np.zeros(max(n, 1), dtype=np.int64)
        e = np.zeros(max(n, 1), dtype=np.uint8)
        self.L.vlxo_split_float32(_p(r, _u8p), n, _p(m, _i64p), _p(e, _u8p))
        return m[:n], e[:n]

    def merge_float32(self, mant, exps):
        m = np.ascontiguousarray(mant, dtype=np.int64)
        e = np.ascontiguousarray(exps, dtype=np.uint8)
        o = np.zeros(max(4 * m.size, 1), dtype=np.uint8)
        self.L.vlxo_merge_float32(_p(m, _i64p), _p(e, _u8p), m.size, _p(o, _u8p))
        return o[:4 * m.size]

    def promote_int_to_float(self, vals, bits):
        v = np.ascontiguousarray(vals, dtype=np.int64)
        o = np.zeros(max(4 * v.size, 1), dtype=np.uint8)
        self.L.vlxo_promote_int_to_float(_p(v, _i64p), v.size, int(bits), _p(o, _u8p))
        return o[:4 * v.size]

    def decode_f32(self, payload, total):
        """What main.cpp:253-270 writes for a float track: decode, zero-fill, Merge / Promote."""
        vals, exps, isf, fm = self.decode(payload, total, want_meta=True)
        assert isf or fm in (1, 2)
        v = np.zeros(total, dtype=np.int64); v[:vals.size] = vals
        e = np.zeros(total, dtype=np.uint8); e[:exps.size] = exps
        return (self.merge_float32(v, e) if fm == 0 else self.promote_int_to_float(v, 16 if fm == 1 else 24)), fm

    def bytes_to_samples(self, raw, bits):
        r = np.ascontiguousarray(raw, dtype=np.uint8)
        n = r.size // (bits // 8)
        o = np.zeros(max(n, 1), dtype=np.int64)
        self.L.vlxo_bytes_to_samples(_p(r, _u8p), n, bits, _p(o, _i64p))
        return o[:n]

    def samples_to_bytes(self, s, bits):
        s = np.ascontiguousarray(s, dtype=np.int64)
        o = np.zeros(max(s.size * (bits // 8), 1), dtype=np.uint8)
        self.L.vlxo_samples_to_bytes(_p(s, _i64p), s.size, bits, _p(o, _u8p))
        return o[:s.size * (bits // 8)]


class Reference:
    """The real reference (compiled from /root/reference).  ``Reference.available()`` gates use."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, threads=None):
        L = C.CDLL(REF_SO)
        L.ref_encode.restype = C.c_size_t
        L.ref_encode.argtypes = [_i64p, C.c_size_t, C.c_int, _u8p, C.c_size_t, _u8p, C.POINTER(_u8p)]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_decode.restype = C.c_size_t
        L.ref_decode.argtypes = [_u8p, C.c_size_t, C.c_size_t, _i64p, _u8p, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)]
        L.ref_compute_lpc.argtypes = [_i64p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_try_compress_channel.restype = C.c_size_t
        L.ref_try_compress_channel.argtypes = [_i64p, C.c_size_t, C.c_int, _u8p, C.c_size_t]
        L.ref_decode_channel.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_int, _i64p]
        L.ref_detect_pseudo_float.restype = C.c_int
        L.ref_detect_pseudo_float.argtypes = [_u8p, C.c_size_t]
        L.ref_split_float32.argtypes = [_u8p, C.c_size_t, _i64p, _u8p]
        L.ref_bytes_to_samples.argtypes = [_u8p, C.c_size_t, C.c_int, _i64p]
        L.ref_samples_to_bytes.argtypes = [_i64p, C.c_size_t, C.c_int, _u8p]
        L.ref_limit_threads.restype = C.c_int
        L.ref_pool_threads.restype = C.c_int
        self.L = L
        if threads is not None:
            L.ref_limit_threads(int(threads))

    def pool_threads(self):
        return self.L.ref_pool_threads()

    def encode(self, samples, is_float=False, exps=None, raw=None):
        s = np.ascontiguousarray(samples, dtype=np.int64).copy()
        out = _u8p()
        e = np.ascontiguousarray(exps, dtype=np.uint8) if exps is not None else None
        r = np.ascontiguousarray(raw, dtype=np.uint8) if raw is not None else None
        n = self.L.ref_encode(_p(s, _i64p), s.size, int(is_float), _p(e, _u8p),
                              0 if e is None else e.size, _p(r, _u8p), C.byref(out))
        res = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy()
        self.L.ref_free(out)
        return res

    def decode(self, payload, total, want_meta=False):
        p = np.ascontiguousarray(payload, dtype=np.uint8)
        out = np.zeros(max(total, 1), dtype=np.int64)
        exps = np.zeros(max(total, 1), dtype=np.uint8)
        isf, fm = C.c_int(0), C.c_int(0)
        n = self.L.ref_decode(_p(p, _u8p), p.size, total, _p(out, _i64p), _p(exps, _u8p),
                              C.byref(isf), C.byref(fm))
        if want_meta:
            return out[:n], exps[:n], isf.value, fm.value
        return out[:n]

    def compute_lpc(self, x, order=8):
        x = np.ascontiguousarray(x, dtype=np.int64)
        c = (C.c_int * order)()
        sh = C.c_int(0)
        self.L.ref_compute_lpc(_p(x, _i64p), x.size, order, c, C.byref(sh))
        return list(c), sh.value

    def try_compress_channel(self, x, high_res):
        x = np.ascontiguousarray(x, dtype=np.int64)
        cap = x.size * 14 + 64
        out = np.zeros(cap, dtype=np.uint8)
        n = self.L.ref_try_compress_channel(_p(x, _i64p), x.size, int(high_res), _p(out, _u8p), cap)
        return out[:n].copy()

    def decode_channel(self, data, count, high_res):
        d = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(max(count, 1), dtype=np.int64)
        self.L.ref_decode_channel(_p(d, _u8p), d.size, count, int(high_res), _p(out, _i64p))
        return out[:count]

    def detect_pseudo_float(self, raw):
        r = np.ascontiguousarray(raw, dtype=np.uint8)
        return self.L.ref_detect_pseudo_float(_p(r, _u8p), r.size // 4)
