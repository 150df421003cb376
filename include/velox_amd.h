/*
 * velox_amd.h — C-ABI of the MI355X-native Velox frame-codec hot path (libvelox_amd.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Each entry point
 * cites the reference interface it replaces (paths relative to the reference tree).  Host code
 * in the reference's own language sits above it (velox_amd/cpp/VeloxCodecAMD.h re-creates
 * VeloxCodec::Encoder / VeloxCodec::StreamingDecoder on top of these calls; INTEGRATION.md shows
 * the three-line change a maintainer makes in main.cpp / the players).
 *
 * All compute runs in hand-written HIP kernels for gfx950.  There is NO CPU fallback: every call
 * returns an error code when no GPU / no kernel image is available.
 *
 * Threading: a vlx_ctx is confined to one host thread at a time (like the reference's Encoder
 * call, VeloxCore.h:210, and a StreamingDecoder instance).  Use one ctx per thread or per GPU.
 */
#ifndef VELOX_AMD_H
#define VELOX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------------------------- */
#define VLX_OK               0
#define VLX_ERR_ARG         -1   /* bad argument (null pointer, bad bits, misaligned offset)      */
#define VLX_ERR_NO_DEVICE   -2   /* no HIP device / device index out of range                    */
#define VLX_ERR_HIP         -3   /* a HIP runtime call or kernel launch failed (see last_error)  */
#define VLX_ERR_NOMEM       -4   /* host or device allocation failed                             */
#define VLX_ERR_UNSUPPORTED -5   /* input shape the GPU path does not take (documented limits)   */
#define VLX_ERR_CAPACITY    -6   /* caller-provided output buffer too small                      */
#define VLX_ERR_PEER        -7   /* collective call: another rank failed or never arrived        */

/* sample formats of the device/batch API */
#define VLX_FMT_S16  16          /* little-endian int16, interleaved                             */
#define VLX_FMT_S24  24          /* little-endian packed 3-byte                                  */
#define VLX_FMT_S32  32          /* little-endian int32                                          */
#define VLX_FMT_I64  64          /* velox_sample_t (VeloxArch.h:10) as the reference boundary passes it */

#define VLX_SUB_BLOCK 8192       /* interleaved samples per chunk (VeloxCore.h:239)              */
#define VLX_ODD_MAX_SAMPLES (1ull << 26)   /* longest track with an ODD sample total (see vlx_encode_device) */

typedef struct vlx_ctx vlx_ctx;
typedef struct vlx_decoder vlx_decoder;

/* ---- context --------------------------------------------------------------------------------
 * Replaces the process-wide ThreadPool behind Encoder::GetPool() (VeloxCore.h:208,
 * VeloxThreads.h:12-62): a context owns one GPU, one HIP stream and a grow-only device workspace. */
int  vlx_device_count(int *count);
int  vlx_ctx_create(int device, vlx_ctx **out);
void vlx_ctx_destroy(vlx_ctx *ctx);
/* The context's own stream is non-blocking: its work is NOT ordered against other streams (the null
 * stream included).  Device inputs must be complete before a call; every call returns with its device
 * outputs complete.  To order against a producer stream instead, run on that stream:
 * a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own. */
int  vlx_ctx_set_stream(vlx_ctx *ctx, void *hip_stream);
/* Workspace policy.  The workspace (device buffers + a few pinned host staging areas) grows on demand inside
 * the batch calls (hipFree + hipMalloc when a batch is larger than any before) and is never shrunk by them:
 * after one configs[1] step about 7 GB stay resident.  vlx_ctx_reserve_for sizes it up front for
 * vlx_encode_device batches of up to `encode_chunks` sub-blocks of `fmt` samples in `encode_tracks` tracks and
 * for vlx_decode_device / vlx_decode_tracks_device batches of `decode_chunks` chunks (of up to 16 tracks), so
 * that no later call of those entry points and that size allocates device or pinned memory — 4-byte and wider
 * formats include the slots of the sequential autocorrelation (33 KB per chunk, at most 65,536 of them).
 * Not covered, because they are sized by what the data turns out to hold or by the caller's sample count: the
 * pool of high words of a decode batch with MORE than 16 chunks whose values leave int32 (only the int64
 * boundary gets there), the float32 entry points, the host-pointer entry points' staging copies.
 * vlx_ctx_reserve is vlx_ctx_reserve_for(VLX_FMT_S24): right for 16- and 24-bit material.
 * vlx_ctx_trim waits for the stream and gives every workspace buffer back (the next call grows it again);
 * vlx_ctx_workspace_bytes reports what is held now, device and pinned.  No reference counterpart (its pool
 * threads hold their 4 MiB writers for the life of the process, VeloxEntropy.h:16). */
int    vlx_ctx_reserve(vlx_ctx *ctx, uint64_t encode_chunks, uint32_t encode_tracks, uint64_t decode_chunks);
int    vlx_ctx_reserve_for(vlx_ctx *ctx, int fmt, uint64_t encode_chunks, uint32_t encode_tracks, uint64_t decode_chunks);
int    vlx_ctx_trim(vlx_ctx *ctx);
size_t vlx_ctx_workspace_bytes(const vlx_ctx *ctx);
const char *vlx_last_error(const vlx_ctx *ctx);
const char *vlx_strerror(int code);
void vlx_free(void *p);          /* frees buffers returned by vlx_encode_block / vlx_encode_pcm   */

/* ---- encode: host-pointer drop-ins -----------------------------------------------------------
 * vlx_encode_block == VeloxCodec::Encoder::ProcessBlock (VeloxCore.h:210-306; sole caller
 * main.cpp:136-137).  `samples` = all interleaved samples of the file as int64; it is rewritten in
 * pseudo-float mode exactly as the reference rewrites its vector (VeloxCore.h:219,222).  `exps`
 * and `raw_bytes` are read only when is_float (raw_bytes = the original little-endian float32
 * PCM).  The payload (malloc'd, free with vlx_free) is byte-identical to the reference's. */
int vlx_encode_block(vlx_ctx *ctx, int64_t *samples, size_t n_samples, int is_float,
                     const uint8_t *exps, size_t n_exps, const uint8_t *raw_bytes,
                     uint8_t **payload, size_t *payload_len);

/* Same result for integer PCM, but takes the container bytes directly (16/24/32-bit LE), i.e.
 * FormatHandler::BytesToSamples (VeloxFormat.h:130-158) is fused into the GPU load instead of
 * widening to 8 B/sample on the host (main.cpp:133). */
int vlx_encode_pcm(vlx_ctx *ctx, const uint8_t *pcm, size_t n_samples, int bits,
                   uint8_t **payload, size_t *payload_len);

/* ---- decode: host-pointer drop-ins -----------------------------------------------------------
 * vlx_decoder_* == VeloxCodec::StreamingDecoder (VeloxCore.h:309-389).  `data` is BORROWED: the
 * caller keeps it alive and unmodified behind the read position (VeloxStreamClient.cpp:379 hands
 * in a buffer that is still being downloaded; pass the bytes available so far as `size` and call
 * vlx_decoder_extend when more have arrived).  Chunks are decoded lazily, `lookahead` chunks per
 * GPU launch (1 = the reference's pacing; larger batches amortise launches for file decode). */
int  vlx_decoder_open(vlx_ctx *ctx, const uint8_t *data, size_t size, size_t total_samples,
                      vlx_decoder **out);                       /* ctor, VeloxCore.h:322-330      */
int  vlx_decoder_set_lookahead(vlx_decoder *d, unsigned chunks);
int  vlx_decoder_extend(vlx_decoder *d, size_t new_size);       /* more bytes of `data` are valid */
/* How many bytes of `data` must be valid before the next `samples` samples can be handed out
 * (vlx_decoder_next x samples, or one vlx_decoder_read of that many) without a refill reading bytes that
 * have not arrived; 0 when they are all buffered already or the stream is over.  When even a chunk's
 * u32 size prefix is missing the answer reaches just past that prefix: ask again after more bytes
 * arrived.  The reference client has no such query and waits on a linear estimate
 * (VeloxStreamClient.cpp:429-437); a decoder that is fed too early reads zero bits, as the
 * reference's does (VeloxEntropy.h:61-62) — and a zero size prefix ends its stream (VeloxCore.h:341). */
int  vlx_decoder_bytes_needed(const vlx_decoder *d, size_t samples, size_t *bytes);
/* 1 when the stream header the constructor reads (VeloxCore.h:324-329: is_float, float_mode, the
 * exponent runs of a true-float track, high_res) lay wholly inside the `size` given to
 * vlx_decoder_open; 0 means the buffer was too short at that moment: close the decoder and open it
 * again when more bytes have arrived. */
int  vlx_decoder_header_complete(const vlx_decoder *d);
/* DecodeNext (VeloxCore.h:335-388): returns 1 and one sample, 0 at end / zero chunk size, <0 on a
 * GPU error. */
int  vlx_decoder_next(vlx_decoder *d, int64_t *val, uint8_t *exp);
/* Pull up to `cap` samples in one call (what every caller's loop does: main.cpp:257-261,
 * VeloxQtPlayerEngine.cpp:664, VeloxStreamClient.cpp:454).  exps may be NULL.  Returns count. */
long vlx_decoder_read(vlx_decoder *d, int64_t *vals, uint8_t *exps, size_t cap);
int  vlx_decoder_is_float(const vlx_decoder *d);                /* IsFloat(),     VeloxCore.h:332 */
int  vlx_decoder_float_mode(const vlx_decoder *d);              /* GetFloatMode(),VeloxCore.h:333 */
/* Copy-assignment of the reference class (VeloxStreamClient.cpp:396, VeloxQtPlayerEngine.cpp:634) */
int  vlx_decoder_clone(const vlx_decoder *d, vlx_decoder **out);
/* A decoder runs its GPU batches on the context it was opened on.  Hand it to another thread together
 * with that thread's context (a vlx_ctx is confined to one thread at a time): every later refill / seek
 * then uses `ctx`.  Both contexts must sit on the same process; the decoder holds no device memory of
 * its own, so nothing moves.  The reference class has no thread affinity at all (its state is plain
 * members), which is what velox_amd/cpp/VeloxCodecAMD.h restores with this call. */
int  vlx_decoder_set_ctx(vlx_decoder *d, vlx_ctx *ctx);
void vlx_decoder_close(vlx_decoder *d);
/* Many concurrent streams (BASELINE config 5; a server or player process driving S decoders,
 * VeloxStreamClient.cpp:449-459 x S): refill, in ONE GPU batch, every listed decoder whose block
 * buffer is exhausted — its next `lookahead` chunks.  Decoders that still hold samples, or are at
 * their end, are skipped; all must belong to one context.  Afterwards vlx_decoder_next/read serve
 * from the buffers without touching the GPU until they run dry again. */
int  vlx_decoder_refill_many(vlx_decoder **decoders, size_t n);
/* Position the decoder so that the next sample handed out is number `sample` of the stream.  The
 * players seek by constructing a fresh decoder and pulling `sample` values through DecodeNext
 * (VeloxQtPlayerEngine.cpp:629-649, VeloxStreamClient.cpp:391-426); this gives the same state by
 * hopping over the size-prefixed chunk list on the host and decoding only the target chunk. */
int  vlx_decoder_seek(vlx_decoder *d, size_t sample);

/* ---- device-resident batch API (the measured hot path) ---------------------------------------
 * Replaces the sub-block loop + ThreadPool fan-out + in-order join of ProcessBlock
 * (VeloxCore.h:255-303) for one or many tracks whose PCM already sits in HBM.
 *
 *   d_pcm            device buffer holding every track back to back (any gaps allowed)
 *   fmt              VLX_FMT_*
 *   h_track_byte_off [n_tracks] byte offset of each track's first sample in d_pcm (multiple of 16)
 *   h_track_samples  [n_tracks] interleaved sample count of each track.  A track with an ODD total is one
 *                    stream over the whole input (VeloxCore.h:242-253) and takes the odd path: one lane runs
 *                    its sample-serial part, about 0.4 us per sample (a 60 s stereo 44.1 kHz track: ~2 s).
 *                    Bounded: beyond VLX_ODD_MAX_SAMPLES (2^26 samples, ~27 s of kernel time) every entry
 *                    point answers VLX_ERR_UNSUPPORTED rather than hold the GPU in one wave for minutes.
 *                    The reference's own decoder cannot read such a payload (its DecodeNext always
 *                    expects two streams), so this is encode parity only.
 *                    Odd tracks are placed behind the batch of even ones and do not appear in
 *                    vlx_encode_last_chunk_sizes.
 *   d_payload        device output buffer of payload_cap bytes, 16-byte aligned (vlx_encode_bound() is
 *                    always enough)
 *   h_payload_off    [n_tracks] out: byte offset of track t's payload in d_payload (256-aligned)
 *   h_payload_len    [n_tracks] out: its length — bytes identical to ProcessBlock's return value
 * The call returns after the stream has drained (the lengths are needed on the host). */
size_t vlx_encode_bound(const uint64_t *h_track_samples, uint32_t n_tracks);
int vlx_encode_device(vlx_ctx *ctx, const void *d_pcm, int fmt,
                      const uint64_t *h_track_byte_off, const uint64_t *h_track_samples,
                      uint32_t n_tracks, uint8_t *d_payload, size_t payload_cap,
                      uint64_t *h_payload_off, uint64_t *h_payload_len);

/* Chunk index of the tracks encoded by the LAST vlx_encode_device call on this ctx (kept on the
 * device): lets a decode follow an encode without walking the size-prefixed chunk list.
 * Copies [n_chunks] entries of chunk byte sizes to the host. */
int vlx_encode_last_chunk_sizes(vlx_ctx *ctx, uint32_t *h_sizes, size_t cap, size_t *n_chunks);

/* ---- float32 tracks, device-resident ----------------------------------------------------------
 * The float prologue of ProcessBlock (VeloxCore.h:214-230) and the callers' float epilogue
 * (main.cpp:262-270) with the samples staying in HBM.
 *
 * vlx_encode_f32_device: d_f32 = n_samples interleaved IEEE float32 samples (16-byte aligned, like
 * every PCM buffer of the device API; an odd count takes the odd path, see vlx_encode_device)
 * exactly as the container holds them.  Runs FormatHandler::DetectPseudoFloat
 * (VeloxFormat.h:14-58) as a reduction, then either DemoteFloatToInt (:60-70) or SplitFloat32
 * (:84-105) + VeloxCodec::EncodeRLE (VeloxCore.h:187-195, a scan) and the integer codec; the payload
 * in d_payload is byte-identical to ProcessBlock(samples, true, exps, raw_bytes).  *float_mode
 * (may be NULL) = 0 true float, 1 / 2 pseudo float of 16 / 24 bits.  payload_cap: the exponent runs of
 * a true-float track can take up to 2 bytes per sample in front of the chunks, so
 * vlx_encode_bound() + 2 * n_samples + 256 is always enough; a smaller buffer is checked before
 * anything is written (VLX_ERR_CAPACITY).  The payload starts at d_payload (16-byte aligned).
 * vlx_encode_last_chunk_sizes works after it as after vlx_encode_device. */
int vlx_encode_f32_device(vlx_ctx *ctx, const void *d_f32, uint64_t n_samples,
                          uint8_t *d_payload, size_t payload_cap, uint64_t *payload_len, int *float_mode);

/* vlx_decode_f32_device: the whole of a float track's payload -> float32 samples in HBM.  Reads the
 * stream header as the StreamingDecoder constructor does (VeloxCore.h:322-330; the exponent runs by
 * VeloxCodec::DecodeRLE, :196-203, as a scan over the fixed-stride pairs), decodes the chunks named by
 * h_sizes (their byte sizes in order, as vlx_encode_last_chunk_sizes or a walk of the size prefixes
 * gives them), then FormatHandler::MergeFloat32 (VeloxFormat.h:107-128) or PromoteIntToFloat (:72-82)
 * into d_f32_out (total_samples x 4 bytes, 4-byte aligned) — what main.cpp:262-270 builds before
 * writing the file.  Samples the stream does not reach are 0.0f.  d_payload: 4-byte aligned with 16
 * readable bytes beyond payload_bytes.  A payload whose is_float bit is clear gives VLX_ERR_UNSUPPORTED. */
int vlx_decode_f32_device(vlx_ctx *ctx, const uint8_t *d_payload, size_t payload_bytes, uint64_t total_samples,
                          const uint32_t *h_sizes, uint32_t n_sizes, void *d_f32_out,
                          int *float_mode, uint64_t *n_chunks_decoded);

/* ---- multi-GPU building blocks ----------------------------------------------------------------
 * Sub-blocks are independent (all predictor / entropy state is reset per sub-block and stream,
 * VeloxCore.h:94-96,141-142), so ONE track can be cut into contiguous runs of whole sub-blocks
 * (multiples of VLX_SUB_BLOCK samples, the last run takes the tail), one run per GPU.  What the
 * reference computes over the whole input, the caller exchanges between ranks:
 *   1. vlx_scan_high_res_device on every shard, MAX-reduced over ranks  (VeloxCore.h:232-236)
 *   2. vlx_encode_shard_device with that file-global flag.  The shard payload is ProcessBlock's
 *      output restricted to the shard's chunks: 2 leading bits (is_float = 0 and high_res on the
 *      first shard, zeros on the others), then the size-prefixed chunks.  Every shard's first chunk
 *      therefore sits at bit 2 of a byte, exactly where it sits in the whole payload.
 *   3. vlx_encode_last_chunk_sizes, all-gathered: rank r's bytes start at
 *      off_r = sum over earlier ranks of sum(chunk size + 4) in the whole payload; the one byte
 *      two neighbouring shards share (last byte of r, first byte of r+1) is the OR of both.
 * velox_amd/sharded.py does exactly this over torch.distributed (RCCL on xGMI).
 * Decode shards without any exchange: vlx_walk_chunks once on the host, contiguous runs of the chunk
 * table per GPU, each GPU gets the bytes of its run (a run starts at bit 2 of a byte, like a payload of
 * its own) and its chunks, re-based, go to vlx_decode_device; output offsets are static. */
int vlx_scan_high_res_device(vlx_ctx *ctx, const void *d_pcm, int fmt, uint64_t n_samples, int *high_res);
int vlx_encode_shard_device(vlx_ctx *ctx, const void *d_pcm, int fmt, uint64_t n_samples, int high_res,
                            int first_shard, uint8_t *d_payload, size_t payload_cap, uint64_t *payload_len);

/* ---- one node, many GPUs: the exchange steps, native (RCCL over xGMI) -------------------------------
 * One process (or thread) per GPU, each with its own vlx_ctx; a vlx_node ties the contexts of the ranks
 * together with an RCCL communicator.  These calls replace, across GPUs, what ProcessBlock does across
 * its pool threads: the whole-input high_res scan (VeloxCore.h:232-236) and the in-order join of the
 * sub-block futures (VeloxCore.h:299-303) — nothing else is exchanged, and nothing at all for decode.
 * librccl.so.1 is bound at run time (the copy already in the process, e.g. torch's, else ROCm's), so the
 * library itself loads without it; without it vlx_node_init returns VLX_ERR_UNSUPPORTED.
 *
 *   vlx_node_unique_id   rank 0 makes the communicator id (ncclGetUniqueId) and hands the 128 bytes to the
 *                        other ranks by any means (a file, MPI, torch.distributed's broadcast).
 *   vlx_node_init        collective: every rank calls it with the same id, its rank and the world size;
 *                        the communicator is created on the context's GPU.
 *   vlx_node_shard_range the standard cut of a track of total_samples into contiguous runs of whole
 *                        sub-blocks, one per rank (first sample, sample count); the last run takes the tail.
 *   vlx_node_encode_track  collective.  This rank's run of ONE integer-PCM track sits in d_pcm (n_samples, a
 *                        multiple of VLX_SUB_BLOCK except on the last rank).  Steps, all on the context's
 *                        stream: high-res scan of the shard -> ncclAllReduce(MAX) of the device flag (the
 *                        flag never visits the host) -> analysis, serial encode, assembly with that flag ->
 *                        one ncclAllGather of [chunk count, chunk byte sizes...] rows.  Out: the shard's
 *                        payload in d_payload (as vlx_encode_shard_device lays it out), its length, its byte
 *                        offset in the whole payload, the whole payload's length and the track's high_res.
 *                        max_chunks_per_rank is the row length of that all-gather and must be the SAME on
 *                        every rank: >= the largest chunk count of any rank (for vlx_node_shard_range cuts:
 *                        ceil(ceil(total / VLX_SUB_BLOCK) / world)).  0 = let the ranks agree on it: the
 *                        largest count rides in the flag all-reduce, at the price of one more stream
 *                        synchronisation per track.  With an explicit bound the call synchronises once.
 *                        Every refusal is collective: a rank whose arguments are bad, whose buffer is too
 *                        small or whose share is odd (an odd total is ONE stream, VeloxCore.h:242-253, and
 *                        cannot be cut; with world == 1 it takes the odd path of vlx_encode_device) still
 *                        makes both exchanges and says so in them, and its peers return VLX_ERR_PEER
 *                        instead of waiting for ever.
 *   vlx_node_chunk_sizes the gathered size table of rank r after vlx_node_encode_track (what a decoder of the
 *                        whole payload needs: vlx_chunks_from_sizes / vlx_decode_tracks_device).
 *   vlx_node_gather_payload  collective, optional (only when one GPU wants the whole file): every rank sends
 *                        its bytes straight into their final place in root's d_full (grouped ncclSend /
 *                        ncclRecv: one xGMI link per peer, no ring), root ORs in the one byte neighbours share.
 *   vlx_node_all_gather_lengths  collective, corpus mode (whole tracks per rank): the table of per-track
 *                        payload lengths, h_table[world][max_per_rank] with h_counts[world] valid entries.
 *   vlx_node_offset_table  the arithmetic alone (no GPU): rank r's bytes start at the sum over earlier ranks
 *                        of sum(size + 4); each shard spans its body + 1 byte; the payload is all bodies + 1. */
typedef struct vlx_node vlx_node;
#define VLX_NODE_ID_BYTES 128
int  vlx_node_unique_id(uint8_t id[VLX_NODE_ID_BYTES]);
int  vlx_node_init(vlx_ctx *ctx, const uint8_t id[VLX_NODE_ID_BYTES], int rank, int world, vlx_node **out);
/* The same node over other transports (the calls above it are the same code):
 *   vlx_node_init_local   the ranks are THREADS of this process, one vlx_ctx each, on any devices (the same GPU
 *                         several times included): a host barrier + device-to-device copies, no RCCL.  Make one
 *                         vlx_local_group for `world` ranks, hand it to every thread, and let each thread make its
 *                         node; a rank that does not reach a collective within the group's time limit (120 s,
 *                         vlx_local_group_set_timeout) breaks the group: every call then fails with VLX_ERR_PEER
 *                         instead of hanging.  The group is freed when it and every node on it are destroyed.
 *   vlx_node_init_custom  bring your own transport (MPI, the pipes of a test harness): every function gets DEVICE
 *                         pointers on the context's GPU and the context's HIP stream, returns 0 on success, and is
 *                         called in the same order on every rank; send / recv are issued between group_start and
 *                         group_end (which may be NULL) and may complete there. */
typedef struct vlx_local_group vlx_local_group;
int  vlx_local_group_create(int world, vlx_local_group **out);
int  vlx_local_group_set_timeout(vlx_local_group *g, int seconds);
void vlx_local_group_destroy(vlx_local_group *g);
int  vlx_node_init_local(vlx_ctx *ctx, vlx_local_group *g, int rank, vlx_node **out);
typedef struct {
    void *user;
    int (*all_reduce_max_u32)(void *user, void *d_words, size_t n_words, void *hip_stream);           /* in place */
    int (*all_gather)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, void *hip_stream);
    int (*send)(void *user, const void *d_buf, size_t bytes, int peer, void *hip_stream);
    int (*recv)(void *user, void *d_buf, size_t bytes, int peer, void *hip_stream);
    int (*group_start)(void *user);
    int (*group_end)(void *user, void *hip_stream);
} vlx_collectives;
int  vlx_node_init_custom(vlx_ctx *ctx, const vlx_collectives *fns, int rank, int world, vlx_node **out);
/* For a vlx_collectives table written where no HIP binding is at hand (velox_amd/sharded.py:TorchTransport carries the
 * six calls over any torch.distributed backend): waits for hip_stream, then copies n bytes between any two host or
 * device addresses (hipMemcpyDefault). */
int  vlx_node_copy(void *dst, const void *src, size_t n, void *hip_stream);
void vlx_node_destroy(vlx_node *node);
void vlx_node_shard_range(uint64_t total_samples, int rank, int world, uint64_t *first_sample, uint64_t *n_samples);
int  vlx_node_encode_track(vlx_node *node, const void *d_pcm, int fmt, uint64_t n_samples, uint32_t max_chunks_per_rank,
                           uint8_t *d_payload, size_t payload_cap, uint64_t *shard_len, uint64_t *shard_off,
                           uint64_t *total_len, int *high_res);
int  vlx_node_chunk_sizes(const vlx_node *node, int rank, uint32_t *h_sizes, size_t cap, size_t *n_chunks);
int  vlx_node_gather_payload(vlx_node *node, int root, const uint8_t *d_shard, uint64_t shard_len, uint8_t *d_full, size_t full_cap);
int  vlx_node_all_gather_lengths(vlx_node *node, const uint64_t *h_lengths, uint32_t n_local, uint32_t max_per_rank,
                                 uint64_t *h_table, uint32_t *h_counts);
int  vlx_node_offset_table(const uint32_t *counts, const uint32_t *size_rows, uint32_t world, uint32_t row_stride,
                           uint64_t *off, uint64_t *seg_len, uint64_t *total_len);

/* ---- one process, many GPUs: the host-pointer drop-ins over a pool of contexts ------------------------------
 * The reference's only caller is ONE thread of ONE process (main.cpp:136-137, :253-261).  A vlx_pool puts every GPU
 * behind that call: one worker thread + one vlx_ctx per entry of `devices` (an entry may repeat: several shards in
 * flight on one GPU), the track cut into contiguous runs of sub-blocks (vlx_node_shard_range), each worker uploading
 * its run, the two exchanges of vlx_node_encode_track over the in-process transport (vlx_node_init_local), and each
 * worker's bytes downloaded straight into their place in the malloc'd payload (the one byte two neighbours share is
 * OR-ed in by the caller's thread).  Decode: one host walk of the size prefixes (VeloxCore.h:338-353), contiguous
 * chunk runs per worker (vlx_node_decode_range), packed PCM downloaded into place.  Same bytes as the one-GPU calls
 * of the same names; what does not shard — odd totals, float tracks (their prologue looks at the whole input),
 * tracks of fewer sub-blocks than workers — runs on the first worker's context.  The pool's calls are made from one
 * thread at a time. */
typedef struct vlx_pool vlx_pool;
int  vlx_pool_create(const int *devices, int n, vlx_pool **out);
void vlx_pool_destroy(vlx_pool *pool);
int  vlx_pool_size(const vlx_pool *pool);
const char *vlx_pool_last_error(const vlx_pool *pool);
int  vlx_pool_encode_block(vlx_pool *pool, int64_t *samples, size_t n_samples, int is_float,
                           const uint8_t *exps, size_t n_exps, const uint8_t *raw_bytes,
                           uint8_t **payload, size_t *payload_len);      /* == vlx_encode_block */
int  vlx_pool_encode_pcm(vlx_pool *pool, const uint8_t *pcm, size_t n_samples, int bits,
                         uint8_t **payload, size_t *payload_len);        /* == vlx_encode_pcm   */
int  vlx_pool_decode_pcm(vlx_pool *pool, const uint8_t *payload, size_t size, size_t total_samples, int bits,
                         uint8_t *pcm, size_t pcm_cap, size_t *n_decoded); /* == vlx_decode_pcm */

/* One chunk of a payload as the decoder's refill step sees it (VeloxCore.h:340-353). */
typedef struct {
    uint64_t bit_off;       /* bit position of the chunk body in the payload buffer (after its u32 size) */
    uint64_t n_bits;        /* 8*chunkSize, clipped to the bits actually present (reads beyond are 0,
                               VeloxEntropy.h:61-62)                                                */
    uint64_t out_off;       /* index of its first output sample in the output buffer             */
    uint32_t frames;        /* min(4096, remaining/2), or `remaining` when that is 0 (<= 4096)   */
    uint32_t n_out;         /* samples it contributes (2*frames, clipped to the track total)     */
    uint32_t high_res;      /* the track's high_res_mode bit (VeloxCore.h:329)                   */
    uint32_t reserved;
} vlx_chunk;

/* Host-side walk of the size-prefixed chunk list of ONE integer-PCM payload (StreamingDecoder's
 * refill logic, VeloxCore.h:338-353, minus the decoding).  base_bit_off / out_base are added to
 * the positions so that many tracks can share one device buffer.  Returns the number of chunks
 * written, or a negative VLX_ERR_* (VLX_ERR_UNSUPPORTED for float payloads — use vlx_decoder_*). */
long vlx_walk_chunks(const uint8_t *payload, size_t size, size_t total_samples,
                     uint64_t base_bit_off, uint64_t out_base, vlx_chunk *chunks, size_t cap);

/* The same chunk table when the chunk byte sizes are already known (vlx_encode_last_chunk_sizes, or
 * the all-gathered size table of a sharded encode): no payload bytes are read.  hdr_bits = bits in
 * front of the first u32 size (2 for integer PCM: is_float, high_res; VeloxCore.h:226-236).  A zero
 * size ends the list as it ends the reference's walk (VeloxCore.h:341).  Returns the chunk count. */
long vlx_chunks_from_sizes(const uint32_t *sizes, size_t n_sizes, size_t total_samples,
                           uint64_t base_bit_off, uint64_t out_base, uint32_t high_res, uint32_t hdr_bits,
                           vlx_chunk *chunks, size_t cap);

/* Decode a batch of chunks that sit in HBM.  out_fmt: VLX_FMT_I64 writes velox_sample_t exactly
 * as DecodeNext hands them out; VLX_FMT_S16/S24/S32 fuses FormatHandler::SamplesToBytes
 * (VeloxFormat.h:160-189).  d_out is indexed in samples by vlx_chunk.out_off.
 * Buffer contract (also for vlx_decode_tracks_device and vlx_decode_device_noalloc): d_payload is 4-byte
 * aligned (checked: VLX_ERR_ARG otherwise) and readable up to the next 16-byte boundary past
 * d_payload + payload_bytes — the kernels fetch whole aligned dwords / 16-byte groups and mask what lies
 * beyond a chunk's end, they never use those bytes.  Any hipMalloc'd buffer satisfies this; a tight
 * sub-allocation that ends within 15 bytes of an unmapped page does not.  n_chunks < 2^31. */
int vlx_decode_device(vlx_ctx *ctx, const uint8_t *d_payload, size_t payload_bytes,
                      const vlx_chunk *h_chunks, size_t n_chunks, void *d_out, int out_fmt);

/* Decode partitioning across ranks (the refill walk of VeloxCore.h:338-353 done once, then cut): rank
 * `rank` of `world` gets the contiguous run [chunk_lo, chunk_hi) of the chunk table of ONE payload, the payload bytes
 * [byte_lo, byte_hi) that hold it (from the byte of its first u32 size prefix: a run starts at bit 2 of a byte, like
 * a payload of its own) and the samples [out_lo, out_lo + out_n) of the track.  Output offsets are static, so ranks
 * exchange nothing.  vlx_node_decode_shard does the rank's share: uploads only those bytes, re-bases its chunks onto
 * them and onto d_out (which receives out_n samples in out_fmt, its own slice only) and runs vlx_decode_device. */
typedef struct {
    uint64_t chunk_lo, chunk_hi;
    uint64_t byte_lo, byte_hi;
    uint64_t out_lo, out_n;
} vlx_decode_range;
int vlx_node_decode_range(const vlx_chunk *chunks, size_t n_chunks, int rank, int world, vlx_decode_range *out);
int vlx_node_decode_shard(vlx_ctx *ctx, const uint8_t *payload, size_t size, const vlx_chunk *chunks, size_t n_chunks,
                          int rank, int world, void *d_out, int out_fmt, vlx_decode_range *range);

/* One track of a batch whose chunk byte sizes are known (vlx_encode_last_chunk_sizes, a container's
 * index, the all-gathered size table of a sharded encode). */
typedef struct {
    uint64_t base_bit_off;  /* bit position of the track's payload in the device buffer            */
    uint64_t out_base;      /* index of its first output sample in the output buffer              */
    uint64_t total_samples; /* samples the track decodes to (the container header's count)        */
    uint32_t n_sizes;       /* entries of the size table that belong to this track                */
    uint32_t high_res;      /* the track's high_res_mode bit (VeloxCore.h:329), or
                               VLX_HIGH_RES_FROM_PAYLOAD: the device reads it, the last header bit */
    uint32_t hdr_bits;      /* bits in front of the first u32 size (2 for integer PCM)            */
    uint32_t reserved;
} vlx_track_sizes;
#define VLX_HIGH_RES_FROM_PAYLOAD 0xFFFFFFFFu

/* vlx_chunks_from_sizes for every track + vlx_decode_device in one call, with the chunk table built
 * on the device: 4 bytes per chunk go up instead of a 40-byte descriptor, and the host does one pass
 * over the sizes.  h_sizes holds the tracks' size tables back to back.  Same results and the same
 * stopping rules as the two-call form (a zero size or total_samples ends a track, VeloxCore.h:336,341);
 * *n_chunks (optional) receives the number of chunks decoded.  No reference counterpart: the
 * reference walks and decodes one chunk at a time (VeloxCore.h:338-353). */
int vlx_decode_tracks_device(vlx_ctx *ctx, const uint8_t *d_payload, size_t payload_bytes,
                             const vlx_track_sizes *h_tracks, uint32_t n_tracks, const uint32_t *h_sizes,
                             void *d_out, int out_fmt, size_t *n_chunks);

/* Whole integer-PCM payload in host memory -> packed little-endian PCM in host memory: the mirror of
 * vlx_encode_pcm.  Replaces the DecodeNext loop of main.cpp:253-261 followed by
 * FormatHandler::SamplesToBytes (VeloxFormat.h:160-189) with one upload, one GPU batch and one download
 * of 2-4 bytes per sample.  pcm receives total_samples * bits/8 bytes; samples the stream does not reach
 * (zero chunk size, truncated payload) are zero, *n_decoded (optional) says how many it did reach.
 * Float payloads: VLX_ERR_UNSUPPORTED (use vlx_decoder_*, which also hands out the exponents). */
int vlx_decode_pcm(vlx_ctx *ctx, const uint8_t *payload, size_t size, size_t total_samples, int bits,
                   uint8_t *pcm, size_t pcm_cap, size_t *n_decoded);

/* ---- allocation-free decode -------------------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate: tight serving loops (BASELINE config
 * 5: a tick decodes the next chunk of every live stream), harnesses that forbid hipMalloc, processes whose
 * device memory is spoken for.  (It is NOT capturable into a HIP graph as it stands: it copies the wide-chunk
 * counter back and synchronises the stream before it returns.)  Same arguments, same results and the same error behaviour as vlx_decode_device, but
 * the whole device workspace (chunk table, per-chunk state, the int32 stream scratch, the hi-word pool)
 * is a file-scope `static __device__` array of the library, VLX_NOALLOC_SCRATCH_BYTES per GPU, resolved
 * once per context by vlx_ctx_create (hipGetSymbolAddress); the chunk table is staged through pinned host
 * memory that vlx_ctx_create set aside.  The call itself runs no hipMalloc / hipFree / hipHostMalloc and
 * creates no events (vlx_set_profiling has no effect on it).
 * Limits: n_chunks <= VLX_NOALLOC_MAX_CHUNKS, else VLX_ERR_CAPACITY and nothing is launched; at most
 * VLX_NOALLOC_HI_SLOTS chunks of one batch may carry values beyond int32 (only the int64 boundary and
 * 33-bit side terms do), else VLX_ERR_CAPACITY after the batch ran (d_out is then incomplete).
 * Every call on a GPU shares that GPU's array, which keeps the last call's contents: calls that overlap on
 * one GPU (two contexts of one device, two threads) race — serialise them. */
#define VLX_NOALLOC_MAX_CHUNKS 1024
#define VLX_NOALLOC_HI_SLOTS   16
#define VLX_NOALLOC_SCRATCH_BYTES (40960u + 32768u + 256u + 16u * 32768u + 1024u * 32768u)
int vlx_decode_device_noalloc(vlx_ctx *ctx, const uint8_t *d_payload, size_t payload_bytes,
                              const vlx_chunk *h_chunks, size_t n_chunks, void *d_out, int out_fmt);

/* ---- introspection used by tests / bench ----------------------------------------------------- */
/* Per-kernel device time (ms) of the last batch call, measured with HIP events on the ctx stream.
 * names/ms hold up to cap entries; returns the number of kernels recorded.  The sample-serial stage is
 * named after the kernel the batch size selected: "k_encode_serial" or "k_encode_pipe", "k_decode_streams"
 * or "k_decode_pipe". */
int vlx_last_kernel_times(vlx_ctx *ctx, const char **names, float *ms, int cap);
/* Enable/disable that event timing (off by default: events serialise nothing but cost launches). */
int vlx_set_profiling(vlx_ctx *ctx, int on);
/* Wave shape of the fast decode kernel, process-wide: VLX_SHAPE_AUTO (default) mixes the two shapes
 * by cost, VLX_SHAPE_STREAM / VLX_SHAPE_CHUNK force one (lane per stream with a pre-parse of stream L
 * / lane per chunk, L then R).  Same results either way; tests run both, tools use it for tuning.
 * No reference counterpart. */
#define VLX_SHAPE_AUTO   0
#define VLX_SHAPE_STREAM 1
#define VLX_SHAPE_CHUNK  2
#define VLX_SHAPE_PIPE   3       /* small-batch shapes: parse / LMS / LPC stages on separate waves (lowest latency per chunk) */
#define VLX_SHAPE_PIPE16 4       /*   ... forced to 16 chunks per workgroup (LMS stage on quads of lanes)                  */
#define VLX_SHAPE_PIPE64 5       /*   ... forced to 64 chunks per workgroup                                                */
#define VLX_SHAPE_PIPE32 6       /*   ... forced to 32 chunks per workgroup (LMS stage on pairs of lanes)                 */
int vlx_set_decode_shape(int shape);
/* The same for the sample-serial encode kernel: VLX_SHAPE_AUTO (default: by batch size), VLX_SHAPE_STREAM
 * (the throughput shape, one lane per stream, LMS and Rice coder in one wave), VLX_SHAPE_PIPE / _PIPE16 /
 * _PIPE64 (the small-batch shape: LMS and Rice coder on separate waves).  Same bytes either way. */
int vlx_set_encode_shape(int shape);

#ifdef __cplusplus
}
#endif
#endif /* VELOX_AMD_H */
