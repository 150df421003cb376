// Host-side internals shared by vlx_api.cpp (single-GPU entry points) and vlx_node.cpp (the multi-GPU layers).  Not installed.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "vlx_internal.h"

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Grow-only pinned host memory: copies to and from it are truly asynchronous on the stream.
struct HostBuf {
    void *p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8 + 256;
        const hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct KernelTime { const char *name; hipEvent_t a, b; };

struct vlx_ctx {
    int device = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    hipStream_t side = nullptr; hipEvent_t side_go = nullptr;   // the chunk sizes of an encode come down on a stream of their own, beside k_assemble
    std::string err;
    // encode
    DevBuf params, res, low, staging, stream_bits, chunk_use_ms, chunk_info, chunk_off, chunk_track, track_hires, tracks, track_len, hdr_bytes, asm_desc;
    DevBuf seq_buf;                         // sequential-autocorrelation slots of a batch of 4-byte samples (SeqSlot)
    DevBuf in_stage, out_stage;             // host-pointer entry points stage through these
    int fin_fmt_override = 0; const uint8_t *fin_exps = nullptr;   // vlx_decode_f32_device: float32 output format of k_decode_finalize + its exponent plane
    DevBuf f_flags, f_samples, f_exps, f_work, f_rle, f_hdr;   // float front-end: detect flags, int64 samples, exponent plane, RLE scan space, pairs, header
    uint64_t last_n_chunks = 0;
    uint32_t last_redo = 0;                 // chunks of the last decode batch that the careful kernel decoded again
    // decode
    DevBuf dchunks, dtracks, dscan, scratch, hi_pool, wide_count, chunk_state, dec_in, dec_out;
    std::vector<uint8_t> h_in; std::vector<int64_t> h_out;     // host staging of batched decoder refills
    HostBuf h_setup, h_chunk_info, h_lens, h_chunks;                     // pinned: track table + headers up, chunk sizes and lengths down
    // allocation-free decode (vlx_decode_device_noalloc): this device's copy of the file-scope scratch
    // array and a pinned staging area for the chunk table + the wide-chunk counter, both set up at create
    uint8_t *na_scratch = nullptr; uint8_t *na_host = nullptr;
    // profiling
    bool prof = false;
    std::vector<KernelTime> times;
    std::vector<std::pair<const char *, float>> last_times;
};

#define HIPCHK(ctx, call)                                                                        \
    do { hipError_t e__ = (call); if (e__ != hipSuccess) {                                       \
        (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__); return VLX_ERR_HIP; } } while (0)


void vlx_prof_begin(vlx_ctx *c, const char *name);
void vlx_prof_end(vlx_ctx *c);
void vlx_prof_collect(vlx_ctx *c);

// Exchange hooks of a sharded encode (vlx_node_encode_track), both run on the context's stream.
//   flag_exchange  between the high-res scan and the analysis, on the device flag words themselves (n_tracks x
//                  VLX_HIRES_STRIDE u32, one flag per 64-B line; word 1 of a line is free for the caller's use).
//   after_layout   once the chunk sizes are final (k_chunk_layout has run), before k_assemble: `status` is VLX_OK or the
//                  error this rank is about to return (capacity), so that the peers learn of it in the same collective.
struct EncHook {
    int (*flag_exchange)(void *self, uint32_t *d_flags, size_t n_words, hipStream_t st);
    int (*after_layout)(void *self, const ChunkInfo *d_chunk_info, uint64_t n_chunks, int status, hipStream_t st);
    void *self;
};

// force_hires[t]: -1 = scan on the device; -3 = scan on the device, continuation shard (its header bit is not
// emitted); else bit 0 = value, bit 1 = continuation shard.
extern "C" int vlx_encode_device_impl(vlx_ctx *c, const void *d_pcm, int fmt, const uint64_t *off, const uint64_t *ns, uint32_t n_tracks,
                           const std::vector<std::vector<uint8_t>> *hdr_bytes, const uint64_t *hdr_bits, const int *force_hires,
                           uint8_t *d_payload, size_t payload_cap, uint64_t *h_payload_off, uint64_t *h_payload_len,
                           const EncHook *hook = nullptr, const uint8_t *d_hdr = nullptr);
static inline uint64_t vlx_track_chunks(uint64_t n_samples) { return (n_samples + VLX_SUB_BLOCK - 1) / VLX_SUB_BLOCK; }
