// Internal structures shared by the HIP kernels and the host C-ABI layer.  Not installed.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <hip/hip_runtime.h>
#include "../../include/velox_amd.h"

#define VLX_STREAM_LEN      4096        // samples per stream (half a sub-block, VeloxCore.h:258)
#define VLX_LPC_ORDER       8           // VeloxCore.h:88,136
#define VLX_HIRES_STRIDE    16          // dwords between two tracks' high_res flags
#define VLX_TAPS            12          // VeloxCore.h:17
// Staging slot per stream, in dwords.  A chunk goes raw once its packed size reaches 5 B/sample
// (VeloxCore.h:286) = 40960 bytes for a full chunk, so a stream never needs more than that.
#define VLX_SLOT_DWORDS     (40960 / 4 + 64)
#define VLX_SLOT_LIMIT_BITS (40960u * 8u + 64u)

// stream flag bits
#define SF_SILENT   1u      // IsSilence() fired (VeloxAdvanced.h:14): the stream is the single bit '1'
#define SF_WIDE     2u      // values/residuals do not fit the 32-bit tiers -> generic 64-bit kernel
#define SF_FAST16   4u      // every residual fits int16: the stream has no high plane in `res`

struct __attribute__((aligned(64))) StreamParams {
    int32_t  n;             // samples in this stream
    uint32_t flags;         // SF_*
    int32_t  shift_lsb;     // LSBShifter::Analyze result (VeloxFormat.h:196-212)
    int32_t  lpc_shift;     // 11, or 0 for a zero-energy stream (VeloxCore.h:53,64)
    int32_t  coefs[VLX_LPC_ORDER];
    int32_t  maxres;        // unused (kept for the 64-byte record layout)
    int32_t  pad[3];
};

struct ChunkInfo {          // written by k_chunk_layout
    uint32_t bytes;         // chunk byte size as stored in the u32 prefix
    uint32_t flags;         // bit0 raw, bit1 use_ms
};

struct TrackDesc {          // one per track, uploaded by the host
    uint64_t byte_off;      // first sample in the PCM buffer
    uint64_t n_samples;     // interleaved samples
    uint64_t chunk_base;    // index of its first chunk in the flat chunk list
    uint64_t out_off;       // byte offset of its payload in the output buffer (written on device)
    uint64_t hdr_bits;      // bits before the first chunk INCLUDING the trailing high_res bit
                            // (2 for integer PCM: is_float, high_res; VeloxCore.h:226-236)
    uint64_t hdr_byte_off;  // where those bits (minus high_res, which the device ORs in) sit in hdr_bytes
    uint32_t high_res;      // bit 0x100: forced by the host to (value & 1), else computed on device (track_hires);
                            // bit 0x200: continuation shard of a sharded track (the high_res header bit is not emitted);
                            // bit 0x400: the header dwords up to two in front of the first chunk are copied into the
                            //            payload by the host-issued memcpy (device-built float headers can be large)
    uint32_t n_chunks;
};

// Everything k_assemble needs to know about one chunk, flattened by k_asm_desc after the layout scans: one 64-byte
// record instead of the chain chunk -> track -> offsets -> stream lengths (four dependent memory latencies per workgroup).
struct AsmDesc {
    uint64_t S;             // absolute bit position of the chunk's u32 size prefix in the output buffer
    uint64_t D0;            // first output dword this chunk's workgroup writes
    uint32_t bytes;         // chunk byte size (the prefix value)
    uint32_t next_bytes;    // the next chunk's, 0 for the last chunk of a track
    uint32_t bits[2];       // code bits of streams L / R
    uint32_t n;             // frames per stream
    uint32_t flags;         // ADF_*
    uint32_t track;
    uint32_t pad[5];
};
#define ADF_RAW      1u
#define ADF_MS       2u
#define ADF_HIRES    4u
#define ADF_LOW_L    8u      // stream L is followed by its low bytes (high-res and not silent)
#define ADF_LOW_R    16u
#define ADF_FIRST    32u     // first chunk of its track: the stream header sits in front of it
#define ADF_LAST     64u     // last chunk of its track

// Chunks whose autocorrelation has to be the reference's sequential fp64 sum (energy >= 2^53: loud 32-bit material,
// VeloxCore.h:48-52) in a batch of 4-byte samples: k_analyze parks them in slots, k_seq_chains runs their 18 chains,
// k_analyze_finish takes them up again.  A batch has min(n_chunks, VLX_SEQ_MAX_SLOTS) slots; chunks beyond that run
// their chains inside k_analyze as all other formats do.
#define VLX_SEQ_PAD 8                       // zeros in front of a stream's values, as in k_analyze's LDS copy
#define VLX_SEQ_ROW (VLX_SEQ_PAD + VLX_STREAM_LEN)
#define VLX_SEQ_MAX_SLOTS 65536u            // 2.1 GB of stream values at most
struct SeqSlot {
    uint32_t chunk; int32_t npairs;
    uint32_t flags;                         // 1 / 2 stream L / R needs the chains, 4 / 8 silent, 16 / 32 wide, 64 mid/side
    uint32_t shl0, shl1;                    // LSB shifts of the two streams
    uint32_t pad[3];
};

struct EncWorkspace {
    StreamParams *params;   // [n_streams]
    int32_t      *res;      // [n_streams] LPC residuals (resLPC, VeloxCore.h:105) as two int16 planes of 4096:
                            // low halves, then high halves (present only without SF_FAST16)
    uint8_t      *low;      // [n_streams][4096] low bytes split off in high-res mode (:73-79)
    uint32_t     *staging;  // [n_streams][VLX_SLOT_DWORDS] per-stream packed bits (header + codes)
    uint32_t     *stream_bits; // [n_streams]
    uint32_t     *chunk_use_ms;// [n_chunks]
    ChunkInfo    *chunk_info;  // [n_chunks]
    uint64_t     *chunk_off;   // [n_chunks+1] exclusive scan of (bytes+4), global over the batch
    uint32_t     *track_hires; // [n_tracks * VLX_HIRES_STRIDE]: one flag per 64-byte line
    TrackDesc    *tracks;      // [n_tracks]
    uint32_t     *chunk_track; // [n_chunks] index of the track a chunk belongs to
    uint64_t     *track_len;   // [n_tracks] payload length out, then [n_tracks+1] payload offsets
    uint8_t      *hdr_bytes;   // per-track header bits (is_float, float_mode, RLE exponents), dword padded
    AsmDesc      *asm_desc;    // [n_chunks] written by k_asm_desc (vlx_launch_layout), read by k_assemble
    // sequential-autocorrelation slots (see SeqSlot); seq_cap = 0: none (formats under 4 bytes per sample)
    int32_t      *seq_xs;      // [seq_cap][2][VLX_SEQ_ROW] shifted stream values
    double       *seq_ac;      // [seq_cap][2][9] autocorrelation: exact sums from k_analyze, the chains' results over them
    SeqSlot      *seq_slot;    // [seq_cap]
    uint32_t     *seq_count;   // slots asked for in this batch (may exceed seq_cap)
    uint32_t      seq_cap;
};

// launchers (vlx_encode.hip / vlx_decode.hip)
hipError_t vlx_launch_chunk_track(hipStream_t st, const EncWorkspace &ws, uint32_t n_tracks, uint64_t n_chunks);
// h_stage: pinned host memory [track table, tr_bytes][...][header bytes at hdr_off]; d_hdr may be null (header bits already in HBM)
hipError_t vlx_launch_enc_setup(hipStream_t st, const void *h_stage, void *d_tracks, size_t tr_bytes, void *d_hdr, size_t hdr_off, size_t hdr_bytes,
                                uint32_t *zero_a, size_t n_a, uint32_t *zero_b);
hipError_t vlx_launch_hires_scan(hipStream_t st, const void *pcm, int fmt, const EncWorkspace &ws,
                                 uint32_t n_tracks, uint64_t n_chunks);
hipError_t vlx_launch_analyze(hipStream_t st, const void *pcm, int fmt, const EncWorkspace &ws,
                              uint32_t n_tracks, uint64_t n_chunks);
hipError_t vlx_launch_encode_serial(hipStream_t st, const EncWorkspace &ws, uint64_t n_streams);
const char *vlx_encode_kernel_label(uint64_t n_streams);
const char *vlx_decode_kernel_label(uint64_t n_chunks);
hipError_t vlx_launch_encode_pipe(hipStream_t st, const EncWorkspace &ws, uint64_t n_streams, int group);   // vlx_pipe.hip
void vlx_encode_shape_set(int shape);
hipError_t vlx_launch_or_seams(hipStream_t st, uint8_t *full, const uint64_t *off, const uint32_t *first_bytes, uint32_t world);
hipError_t vlx_launch_encode_wide(hipStream_t st, const void *pcm, int fmt, const EncWorkspace &ws,
                                  uint32_t n_tracks, uint64_t n_chunks);
hipError_t vlx_launch_encode_odd(hipStream_t st, const void *pcm, int fmt, uint64_t n, int force_hires, const uint8_t *hdr, uint64_t HB,
                                 uint32_t *out, uint64_t cap_dwords, uint64_t *result);
hipError_t vlx_launch_layout(hipStream_t st, const EncWorkspace &ws, uint32_t n_tracks, uint64_t n_chunks);
hipError_t vlx_launch_assemble(hipStream_t st, const void *pcm, int fmt, const EncWorkspace &ws,
                               uint32_t n_tracks, uint64_t n_chunks, uint8_t *out);

// float front-end (vlx_float.hip)
hipError_t vlx_launch_f32_detect(hipStream_t st, const void *raw, uint64_t count, uint32_t *d_flags);
hipError_t vlx_launch_f32_demote(hipStream_t st, const void *raw, uint64_t count, int bits, int64_t *out);
hipError_t vlx_launch_f32_exps(hipStream_t st, const void *raw, uint64_t count, uint8_t *exps);
size_t     vlx_rle_work_entries(uint64_t n);
hipError_t vlx_launch_rle_count(hipStream_t st, const uint8_t *exps, uint64_t n, uint64_t *work);     // work[2*nb + nb] = number of pairs
hipError_t vlx_launch_rle_emit(hipStream_t st, const uint8_t *exps, uint64_t n, uint64_t *work, uint8_t *rle);
hipError_t vlx_launch_float_header(hipStream_t st, const uint8_t *rle, uint64_t n_rle, uint32_t float_mode, uint8_t *hdr, uint64_t n_hdr);
hipError_t vlx_launch_rle_runs(hipStream_t st, const uint8_t *payload, uint64_t payload_bytes, uint64_t p0, uint64_t n_pairs, uint64_t *steps);
hipError_t vlx_launch_rle_expand(hipStream_t st, const uint8_t *payload, uint64_t payload_bytes, uint64_t p0, uint64_t n_pairs, const uint64_t *scan,
                                 uint64_t base, uint64_t total, uint8_t *exps, unsigned long long *first_done);
#define VLX_RLE_BLOCK 4096          // samples per workgroup of the RLE kernels (vlx_float.hip: FL_BLOCK)

struct DecWorkspace {
    vlx_chunk *chunks;      // [n_chunks] device copy
    int32_t   *scratch;     // [n_chunks][2][4096] reconstructed stream values before LSB/low-byte restore (stream-major)
    uint32_t  *chunk_state; // [n_chunks][8] per-chunk results of the serial phase (positions, shifts, flags, hi slot)
    // Values beyond int32 are rare (only the int64 boundary and 33-bit side terms reach them): the careful
    // kernel parks their high words in a small pool, one slot of [2][4096] per flagged chunk.
    int32_t   *hi_pool;     // [hi_cap][2][4096]
    uint32_t   hi_cap;      // slots in the pool
    uint32_t  *wide_count;  // device counter of flagged chunks (slots handed out; may exceed hi_cap)
    const uint8_t *exps;    // VLX_FMT_F32_MERGE output only: the exponent byte of every output sample (DecodeRLE's vector)
};
// float32 output formats of k_decode_finalize (internal: reached through vlx_decode_f32_device): the callers' float
// epilogue fused into the store — FormatHandler::MergeFloat32 (VeloxFormat.h:107-128) with the exponent plane, and
// PromoteIntToFloat (VeloxFormat.h:72-82) from 16 / 24 bits
// float32 INPUT formats of the encode kernels (internal: reached through vlx_encode_f32_device): the float
// prologue fused into the sample load — FormatHandler::SplitFloat32's mantissa (VeloxFormat.h:84-105) and
// DemoteFloatToInt to 16 / 24 bits (VeloxFormat.h:60-70); 4 bytes per sample in memory
#define VLX_FMT_F32_SPLIT 0x2F20
#define VLX_FMT_F32_D16   0x2F10
#define VLX_FMT_F32_D24   0x2F18
__host__ __device__ static inline int vlx_fmt_bytes(int fmt)
{
    return fmt == VLX_FMT_I64 ? 8 : ((fmt == VLX_FMT_F32_SPLIT || fmt == VLX_FMT_F32_D16 || fmt == VLX_FMT_F32_D24) ? 4 : fmt / 8);
}
#define VLX_FMT_F32_MERGE 0x1F20
#define VLX_FMT_F32_P16   0x1F10
#define VLX_FMT_F32_P24   0x1F18
struct DecTrack {           // one per track of vlx_decode_tracks_device, uploaded by the host
    uint64_t first_bit;     // bit position of the track's first u32 chunk size in the payload buffer
    uint64_t out_base;      // index of its first output sample
    uint64_t total;         // samples the track decodes to
    uint64_t first_chunk;   // index of its first chunk in the flat chunk list
    uint32_t high_res;      // the track's high_res_mode bit, or VLX_HIGH_RES_FROM_PAYLOAD
    uint32_t pad;
};
// chunk_state: [chunks][CS_WORDS] u32 written by the serial decode kernels for k_decode_finalize:
// word 0 flags, 1 LSB shifts (L | R << 8), 2 / 3 bit position (relative to the chunk) where stream L / R's
// codes end = where its low bytes start, CS_HI_SLOT the chunk's slot in the hi-word pool
#define CS_WORDS 8
#define CSF_MODE     1u
#define CSF_MS       2u
#define CSF_SILENT_A 4u
#define CSF_SILENT_B 8u
#define CSF_WIDE     16u
#define CSF_VALID    32u
#define CSF_DONE     64u                     // a stream wave of k_decode_streams<FUSE> stored the chunk's PCM itself: finalize skips it
#define CS_HI_SLOT   4                       // chunk_state word: slot in the hi-word pool (wide chunks), CS_NO_SLOT = pool was full
#define CS_NO_SLOT   0xFFFFFFFFu
#define DS_TILE 8                       // samples between two refill points of the bit ring
#define DS_RING 32                      // dwords of compressed bits buffered per lane (in LDS)
void vlx_decode_shape_set(int shape);
int  vlx_decode_shape_get();
// Batch sizes up to which the pipelined small-batch shapes are used (vlx_pipe.hip): 16 chunks per workgroup
// (LMS stage on quads of lanes) up to two workgroups per CU on 256 CUs (measured: 0.75 ms at 64 chunks, 0.83 at
// 4096, 1.04 at 8192, against 0.96 / 1.03 / 1.12 with 64 per workgroup and 1.94 / 1.17 at 11,250), 64 per workgroup
// beyond that, the throughput shapes of k_decode_streams above.
#define VLX_PIPE16_MAX_CHUNKS 8192        // a forced G = 16 beyond this only adds waves per SIMD
// launcher policy, measured on one MI355X (tools/bench_decode_shapes.py, bench_encode_shapes.py, 32-bit material):
// decode  16 chunks per workgroup up to 4096 chunks, 32 up to 8192 (256 pipelines = one per CU each), 64 (with two LMS waves for stream R) up to
// 16384;  encode  16 streams per pipeline up to 8192 streams, 32 up to 16384 (every wave alone on its SIMD up to there), 64 with the LMS stage on
// two pair-lane waves up to 32768, on one wave (and four pipelines per workgroup) up to 65536
#define VLX_DPIPE_G16_MAX 4096
#define VLX_DPIPE_G32_MAX 8192
#define VLX_EPIPE_G16_MAX 8192
#define VLX_EPIPE_G32_MAX 16384
#define VLX_EPIPE_MAX_STREAMS 65536       // beyond: k_encode_serial (the pipelines keep one workgroup per CU up to here, see k_encode_pipe)
#define VLX_PIPE_MAX_CHUNKS   16384
hipError_t vlx_launch_decode_pipe(hipStream_t st, const uint8_t *payload, uint64_t payload_bytes, const DecWorkspace &ws, uint64_t n_chunks, int group);
// in-place inclusive scan of data[1..n]; the buffer holds VLX_SCAN_ROOM(n) entries (data + tile totals)
#define VLX_SCAN_ROOM(n) ((size_t)(n) + 1 + ((size_t)(n) + 4095) / 4096 + 1)
hipError_t vlx_launch_scan_u64(hipStream_t st, uint64_t *data, uint64_t n);
// h_stage: pinned host memory [track table, track_bytes (a multiple of 16)][n_chunks sizes], read by the first kernel; d_stage: its device copy
hipError_t vlx_launch_chunks_build(hipStream_t st, const uint8_t *payload, const void *h_stage, void *d_stage, size_t track_bytes, uint64_t *scan,
                                   uint32_t n_tracks, vlx_chunk *chunks, uint64_t n_chunks, uint32_t *zero_me);
// out / fuse_fmt: the packed-PCM output buffer and its format when the batch may hold chunks a stream wave can finish
// itself (VLX_FMT_S16 / S24 and some chunk without high-res low bytes), else nullptr / 0
hipError_t vlx_launch_decode_streams(hipStream_t st, const uint8_t *payload, uint64_t payload_bytes, const DecWorkspace &ws, uint64_t n_chunks,
                                     void *out = nullptr, int fuse_fmt = 0);
hipError_t vlx_launch_decode_serial(hipStream_t st, const uint8_t *payload, const DecWorkspace &ws, uint64_t n_chunks);
hipError_t vlx_launch_decode_finalize(hipStream_t st, const uint8_t *payload, const DecWorkspace &ws, uint64_t n_chunks,
                                      void *out, int out_fmt);
