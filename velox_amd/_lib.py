"""ctypes loader for libvelox_amd.so (the C-ABI declared in include/velox_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C velox_amd/csrc``.  There is
no CPU fallback: if the shared object is missing the import fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VELOX_AMD_LIB: another build of the same library (A/B measurements of kernel variants: tools/); default = the in-tree one
LIB_PATH = os.environ.get("VELOX_AMD_LIB") or os.path.join(HERE, "lib", "libvelox_amd.so")

VLX_OK = 0
ERR_NAMES = {0: "VLX_OK", -1: "VLX_ERR_ARG", -2: "VLX_ERR_NO_DEVICE", -3: "VLX_ERR_HIP",
             -4: "VLX_ERR_NOMEM", -5: "VLX_ERR_UNSUPPORTED", -6: "VLX_ERR_CAPACITY", -7: "VLX_ERR_PEER"}
FMT_S16, FMT_S24, FMT_S32, FMT_I64 = 16, 24, 32, 64
SUB_BLOCK = 8192

# every symbol include/velox_amd.h declares (tests check the library exports all of them)
EXPORTS = [
    "vlx_device_count", "vlx_ctx_create", "vlx_ctx_destroy", "vlx_ctx_set_stream", "vlx_last_error",
    "vlx_strerror", "vlx_free", "vlx_encode_block", "vlx_encode_pcm", "vlx_decoder_open",
    "vlx_decoder_set_lookahead", "vlx_decoder_extend", "vlx_decoder_bytes_needed", "vlx_decoder_header_complete", "vlx_decoder_next", "vlx_decoder_read",
    "vlx_decoder_is_float", "vlx_decoder_float_mode", "vlx_decoder_clone", "vlx_decoder_close",
    "vlx_encode_bound", "vlx_encode_device", "vlx_encode_last_chunk_sizes", "vlx_walk_chunks", "vlx_chunks_from_sizes",
    "vlx_decode_device", "vlx_decode_tracks_device", "vlx_decode_pcm", "vlx_last_kernel_times", "vlx_set_profiling", "vlx_set_decode_shape", "vlx_scan_high_res_device",
    "vlx_encode_shard_device", "vlx_decoder_refill_many", "vlx_decoder_seek", "vlx_decode_device_noalloc",
    "vlx_set_encode_shape", "vlx_ctx_reserve", "vlx_ctx_reserve_for",
    "vlx_node_unique_id", "vlx_node_init", "vlx_node_destroy", "vlx_node_shard_range", "vlx_node_encode_track", "vlx_node_chunk_sizes",
    "vlx_node_gather_payload", "vlx_node_all_gather_lengths", "vlx_node_offset_table", "vlx_ctx_trim", "vlx_ctx_workspace_bytes", "vlx_decoder_set_ctx",
    "vlx_encode_f32_device", "vlx_decode_f32_device",
    "vlx_local_group_create", "vlx_local_group_set_timeout", "vlx_local_group_destroy", "vlx_node_init_local", "vlx_node_init_custom", "vlx_node_copy",
    "vlx_node_decode_range", "vlx_node_decode_shard",
    "vlx_pool_create", "vlx_pool_destroy", "vlx_pool_size", "vlx_pool_last_error", "vlx_pool_encode_block", "vlx_pool_encode_pcm", "vlx_pool_decode_pcm",
]


class VlxTrackSizes(C.Structure):
    _fields_ = [("base_bit_off", C.c_uint64), ("out_base", C.c_uint64), ("total_samples", C.c_uint64),
                ("n_sizes", C.c_uint32), ("high_res", C.c_uint32), ("hdr_bits", C.c_uint32),
                ("reserved", C.c_uint32)]


class VlxChunk(C.Structure):
    _fields_ = [("bit_off", C.c_uint64), ("n_bits", C.c_uint64), ("out_off", C.c_uint64),
                ("frames", C.c_uint32), ("n_out", C.c_uint32), ("high_res", C.c_uint32),
                ("reserved", C.c_uint32)]


class VlxDecodeRange(C.Structure):
    _fields_ = [("chunk_lo", C.c_uint64), ("chunk_hi", C.c_uint64), ("byte_lo", C.c_uint64), ("byte_hi", C.c_uint64),
                ("out_lo", C.c_uint64), ("out_n", C.c_uint64)]


# vlx_collectives: a caller's transport for vlx_node_init_custom (device pointers, the context's stream; 0 = ok)
COLL_ALL_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
COLL_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
COLL_SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
COLL_RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
COLL_GROUP_START = C.CFUNCTYPE(C.c_int, C.c_void_p)
COLL_GROUP_END = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class VlxCollectives(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_reduce_max_u32", COLL_ALL_REDUCE), ("all_gather", COLL_ALL_GATHER),
                ("send", COLL_SEND), ("recv", COLL_RECV), ("group_start", COLL_GROUP_START), ("group_end", COLL_GROUP_END)]


_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C velox_amd/csrc). "
            "velox_amd has no CPU fallback.")
    if not os.environ.get("VELOX_AMD_NO_TORCH"):
        # torch wheels bundle their own HIP runtime (soname libamdhip64.so.7, the one this library
        # links against).  Two HIP runtimes in one process cannot both own the GPU, so when torch
        # is present let it load first: the dynamic loader then binds us to the same runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.vlx_device_count.argtypes = [C.POINTER(C.c_int)]
    L.vlx_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vlx_ctx_destroy.argtypes = [vp]
    L.vlx_ctx_destroy.restype = None
    L.vlx_ctx_set_stream.argtypes = [vp, vp]
    L.vlx_ctx_reserve.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64]
    L.vlx_ctx_reserve_for.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint64]
    L.vlx_ctx_trim.argtypes = [vp]
    L.vlx_ctx_workspace_bytes.argtypes = [vp]
    L.vlx_ctx_workspace_bytes.restype = C.c_size_t
    L.vlx_decoder_set_ctx.argtypes = [vp, vp]
    L.vlx_last_error.argtypes = [vp]
    L.vlx_last_error.restype = C.c_char_p
    L.vlx_strerror.argtypes = [C.c_int]
    L.vlx_strerror.restype = C.c_char_p
    L.vlx_free.argtypes = [vp]
    L.vlx_free.restype = None
    L.vlx_encode_block.argtypes = [vp, _i64p, C.c_size_t, C.c_int, _u8p, C.c_size_t, _u8p,
                                   C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.vlx_encode_pcm.argtypes = [vp, _u8p, C.c_size_t, C.c_int, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.vlx_decoder_open.argtypes = [vp, _u8p, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vlx_decoder_set_lookahead.argtypes = [vp, C.c_uint]
    L.vlx_decoder_extend.argtypes = [vp, C.c_size_t]
    L.vlx_decoder_bytes_needed.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.vlx_decoder_header_complete.argtypes = [vp]
    L.vlx_decoder_next.argtypes = [vp, _i64p, _u8p]
    L.vlx_decoder_read.argtypes = [vp, _i64p, _u8p, C.c_size_t]
    L.vlx_decoder_read.restype = C.c_long
    L.vlx_decoder_is_float.argtypes = [vp]
    L.vlx_decoder_float_mode.argtypes = [vp]
    L.vlx_decoder_clone.argtypes = [vp, C.POINTER(vp)]
    L.vlx_decoder_refill_many.argtypes = [C.POINTER(vp), C.c_size_t]
    L.vlx_decoder_seek.argtypes = [vp, C.c_size_t]
    L.vlx_decoder_close.argtypes = [vp]
    L.vlx_decoder_close.restype = None
    L.vlx_encode_bound.argtypes = [_u64p, C.c_uint32]
    L.vlx_encode_bound.restype = C.c_size_t
    L.vlx_encode_device.argtypes = [vp, vp, C.c_int, _u64p, _u64p, C.c_uint32, vp, C.c_size_t, _u64p, _u64p]
    L.vlx_encode_f32_device.argtypes = [vp, vp, C.c_uint64, vp, C.c_size_t, _u64p, C.POINTER(C.c_int)]
    L.vlx_decode_f32_device.argtypes = [vp, vp, C.c_size_t, C.c_uint64, _u32p, C.c_uint32, vp, C.POINTER(C.c_int), _u64p]
    L.vlx_scan_high_res_device.argtypes = [vp, vp, C.c_int, C.c_uint64, C.POINTER(C.c_int)]
    L.vlx_encode_shard_device.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_int, C.c_int, vp, C.c_size_t, _u64p]
    L.vlx_encode_last_chunk_sizes.argtypes = [vp, _u32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.vlx_walk_chunks.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64,
                                  C.POINTER(VlxChunk), C.c_size_t]
    L.vlx_walk_chunks.restype = C.c_long
    L.vlx_chunks_from_sizes.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64,
                                        C.c_uint32, C.c_uint32, C.POINTER(VlxChunk), C.c_size_t]
    L.vlx_chunks_from_sizes.restype = C.c_long
    L.vlx_decode_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(VlxChunk), C.c_size_t, vp, C.c_int]
    L.vlx_decode_device_noalloc.argtypes = [vp, vp, C.c_size_t, C.POINTER(VlxChunk), C.c_size_t, vp, C.c_int]
    L.vlx_decode_tracks_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(VlxTrackSizes), C.c_uint32, C.POINTER(C.c_uint32),
                                           vp, C.c_int, C.POINTER(C.c_size_t)]
    L.vlx_decode_pcm.argtypes = [vp, _u8p, C.c_size_t, C.c_size_t, C.c_int, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.vlx_last_kernel_times.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    L.vlx_set_profiling.argtypes = [vp, C.c_int]
    L.vlx_set_decode_shape.argtypes = [C.c_int]
    L.vlx_set_encode_shape.argtypes = [C.c_int]
    L.vlx_node_unique_id.argtypes = [_u8p]
    L.vlx_node_init.argtypes = [vp, _u8p, C.c_int, C.c_int, C.POINTER(vp)]
    L.vlx_node_destroy.argtypes = [vp]
    L.vlx_node_destroy.restype = None
    L.vlx_node_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, _u64p, _u64p]
    L.vlx_node_shard_range.restype = None
    L.vlx_node_encode_track.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint32, vp, C.c_size_t, _u64p, _u64p, _u64p, C.POINTER(C.c_int)]
    L.vlx_node_chunk_sizes.argtypes = [vp, C.c_int, _u32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.vlx_node_gather_payload.argtypes = [vp, C.c_int, vp, C.c_uint64, vp, C.c_size_t]
    L.vlx_node_all_gather_lengths.argtypes = [vp, _u64p, C.c_uint32, C.c_uint32, _u64p, _u32p]
    L.vlx_node_offset_table.argtypes = [_u32p, _u32p, C.c_uint32, C.c_uint32, _u64p, _u64p, _u64p]
    L.vlx_local_group_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vlx_local_group_set_timeout.argtypes = [vp, C.c_int]
    L.vlx_local_group_destroy.argtypes = [vp]
    L.vlx_local_group_destroy.restype = None
    L.vlx_node_init_local.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.vlx_node_init_custom.argtypes = [vp, C.POINTER(VlxCollectives), C.c_int, C.c_int, C.POINTER(vp)]
    L.vlx_node_copy.argtypes = [vp, vp, C.c_size_t, vp]
    L.vlx_node_decode_range.argtypes = [C.POINTER(VlxChunk), C.c_size_t, C.c_int, C.c_int, C.POINTER(VlxDecodeRange)]
    L.vlx_node_decode_shard.argtypes = [vp, _u8p, C.c_size_t, C.POINTER(VlxChunk), C.c_size_t, C.c_int, C.c_int, vp, C.c_int,
                                        C.POINTER(VlxDecodeRange)]
    L.vlx_pool_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.vlx_pool_destroy.argtypes = [vp]
    L.vlx_pool_destroy.restype = None
    L.vlx_pool_size.argtypes = [vp]
    L.vlx_pool_last_error.argtypes = [vp]
    L.vlx_pool_last_error.restype = C.c_char_p
    L.vlx_pool_encode_block.argtypes = [vp, _i64p, C.c_size_t, C.c_int, _u8p, C.c_size_t, _u8p, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.vlx_pool_encode_pcm.argtypes = [vp, _u8p, C.c_size_t, C.c_int, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.vlx_pool_decode_pcm.argtypes = [vp, _u8p, C.c_size_t, C.c_size_t, C.c_int, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    _lib = L
    return L
