// Host side of the C-ABI (include/velox_amd.h): context, workspaces, launch sequencing, and the
// reference-shaped host entry points.  No codec arithmetic happens here except the float-path
// prologue, which the reference also runs serially on the calling thread (VeloxCore.h:214-230).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "vlx_host.h"

void vlx_prof_begin(vlx_ctx *c, const char *name)
{
    if (!c->prof) return;
    KernelTime k; k.name = name;
    if (hipEventCreate(&k.a) != hipSuccess || hipEventCreate(&k.b) != hipSuccess) return;
    (void)hipEventRecord(k.a, c->stream);
    c->times.push_back(k);
}
void vlx_prof_end(vlx_ctx *c)
{
    if (!c->prof || c->times.empty()) return;
    (void)hipEventRecord(c->times.back().b, c->stream);
}
void vlx_prof_collect(vlx_ctx *c)
{
    if (!c->prof) return;
    c->last_times.clear();
    for (auto &k : c->times) {
        float ms = 0;
        (void)hipEventSynchronize(k.b);
        (void)hipEventElapsedTime(&ms, k.a, k.b);
        c->last_times.push_back({k.name, ms});
        (void)hipEventDestroy(k.a); (void)hipEventDestroy(k.b);
    }
    c->times.clear();
}

#define prof_begin vlx_prof_begin
#define prof_end vlx_prof_end
#define prof_collect vlx_prof_collect

// ---- workspace of vlx_decode_device_noalloc: a file-scope device array, one copy per GPU -----------
// Layout (every part 256-B aligned): chunk table, per-chunk state, wide-chunk counter, hi-word pool, int32
// stream scratch.  In host code the NAME `scratch` is only a handle for hipGetSymbolAddress — never a
// device address: vlx_ctx_create resolves it with the context's device current and every later use goes
// through the pointer kept in the context.
#define NA_OFF_CHUNKS  0u
#define NA_OFF_STATE   (NA_OFF_CHUNKS + 40960u)                              /* 1024 x sizeof(vlx_chunk)        */
#define NA_OFF_COUNT   (NA_OFF_STATE + 32768u)                               /* 1024 x 8 u32                    */
#define NA_OFF_HI      (NA_OFF_COUNT + 256u)
#define NA_OFF_SCRATCH (NA_OFF_HI + VLX_NOALLOC_HI_SLOTS * 32768u)           /* slots of [2][4096] int32        */
#define NA_TOTAL       (NA_OFF_SCRATCH + VLX_NOALLOC_MAX_CHUNKS * 32768u)    /* [chunks][2][4096] int32         */
static_assert(NA_TOTAL == VLX_NOALLOC_SCRATCH_BYTES, "header and layout disagree");
static_assert(sizeof(vlx_chunk) * VLX_NOALLOC_MAX_CHUNKS <= 40960u && VLX_NOALLOC_MAX_CHUNKS % 64 == 0, "chunk table part too small");
// (No __attribute__((used)) here: hipcc makes a static device variable visible to the runtime only because
// host code names it — HIP_SYMBOL(scratch) in vlx_ctx_create; marked `used` it stays a LOCAL symbol of the
// code object and hipGetSymbolAddress aborts with "Cannot create GlobalVar Obj".)
static __device__ __align__(256) unsigned char scratch[VLX_NOALLOC_SCRATCH_BYTES];
#define NA_HOST_BYTES (sizeof(vlx_chunk) * VLX_NOALLOC_MAX_CHUNKS + 64)

extern "C" {

const char *vlx_strerror(int code)
{
    switch (code) {
    case VLX_OK: return "ok";
    case VLX_ERR_ARG: return "bad argument";
    case VLX_ERR_NO_DEVICE: return "no usable HIP device";
    case VLX_ERR_HIP: return "HIP runtime error";
    case VLX_ERR_NOMEM: return "out of memory";
    case VLX_ERR_UNSUPPORTED: return "unsupported input for the GPU path";
    case VLX_ERR_CAPACITY: return "output buffer too small";
    case VLX_ERR_PEER: return "another rank of the collective call failed or never arrived";
    default: return "unknown error";
    }
}

int vlx_device_count(int *count)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    if (count) *count = n;
    return n > 0 ? VLX_OK : VLX_ERR_NO_DEVICE;
}

int vlx_ctx_create(int device, vlx_ctx **out)
{
    if (!out) return VLX_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return VLX_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return VLX_ERR_NO_DEVICE;
    {   // the code objects in this library are gfx950 only (and the kernels rely on it: vlx_device.h)
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VLX_ERR_NO_DEVICE;
    }
    vlx_ctx *c = new (std::nothrow) vlx_ctx();
    if (!c) return VLX_ERR_NOMEM;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return VLX_ERR_HIP; }
    c->own_stream = true;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->side_go, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (c->side) (void)hipStreamDestroy(c->side);
        (void)hipStreamDestroy(c->stream); delete c; return VLX_ERR_HIP;
    }
    // the allocation-free decode path's set-up: this device's address of the scratch array, pinned staging
    void *sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)) != hipSuccess || !sym ||
        hipHostMalloc((void **)&c->na_host, NA_HOST_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventDestroy(c->side_go); (void)hipStreamDestroy(c->side);
        (void)hipStreamDestroy(c->stream); delete c; return VLX_ERR_HIP;
    }
    c->na_scratch = (uint8_t *)sym;
    *out = c;
    return VLX_OK;
}

#define VLX_ALL_DEVBUFS(c) {&(c)->params, &(c)->res, &(c)->low, &(c)->staging, &(c)->stream_bits, &(c)->chunk_use_ms, &(c)->chunk_info, &(c)->asm_desc, &(c)->seq_buf, \
                     &(c)->chunk_off, &(c)->chunk_track, &(c)->track_hires, &(c)->tracks, &(c)->track_len, &(c)->hdr_bytes, &(c)->in_stage, &(c)->out_stage, \
                     &(c)->dchunks, &(c)->dtracks, &(c)->dscan, &(c)->scratch, &(c)->hi_pool, &(c)->wide_count, &(c)->chunk_state, &(c)->dec_in, &(c)->dec_out, \
                     &(c)->f_flags, &(c)->f_samples, &(c)->f_exps, &(c)->f_work, &(c)->f_rle, &(c)->f_hdr}

void vlx_ctx_destroy(vlx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *all[] = VLX_ALL_DEVBUFS(c);
    for (DevBuf *b : all) b->release();
    c->h_setup.release(); c->h_chunk_info.release(); c->h_lens.release(); c->h_chunks.release();
    if (c->na_host) (void)hipHostFree(c->na_host);
    if (c->side_go) (void)hipEventDestroy(c->side_go);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int vlx_ctx_trim(vlx_ctx *c)
{
    if (!c) return VLX_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    DevBuf *all[] = VLX_ALL_DEVBUFS(c);
    for (DevBuf *b : all) b->release();
    c->h_setup.release(); c->h_chunk_info.release(); c->h_lens.release(); c->h_chunks.release();
    c->last_n_chunks = 0;                                                 // the chunk index of the last encode went with it
    return VLX_OK;
}

size_t vlx_ctx_workspace_bytes(const vlx_ctx *c)
{
    if (!c) return 0;
    const DevBuf *all[] = VLX_ALL_DEVBUFS(c);
    size_t n = 0;
    for (const DevBuf *b : all) n += b->cap;
    return n + c->h_setup.cap + c->h_chunk_info.cap + c->h_lens.cap + c->h_chunks.cap;   // device buffers + the pinned staging areas
}

int vlx_ctx_set_stream(vlx_ctx *c, void *hip_stream)
{
    if (!c) return VLX_ERR_ARG;
    if (c->own_stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = (hipStream_t)hip_stream;
    return VLX_OK;
}

const char *vlx_last_error(const vlx_ctx *c) { return c ? c->err.c_str() : "null context"; }
void vlx_free(void *p) { free(p); }

int vlx_set_profiling(vlx_ctx *c, int on) { if (!c) return VLX_ERR_ARG; c->prof = on != 0; return VLX_OK; }
int vlx_set_decode_shape(int shape)
{
    if (shape < VLX_SHAPE_AUTO || shape > VLX_SHAPE_PIPE32) return VLX_ERR_ARG;
    vlx_decode_shape_set(shape);
    return VLX_OK;
}
int vlx_set_encode_shape(int shape)
{
    if (shape < VLX_SHAPE_AUTO || shape > VLX_SHAPE_PIPE32 || shape == VLX_SHAPE_CHUNK) return VLX_ERR_ARG;
    vlx_encode_shape_set(shape);
    return VLX_OK;
}
// Test hook, not in the public header: at most `n` sequential-autocorrelation slots per encode batch (SeqSlot; -1 = the
// default, 0 = none: every chain runs inside k_analyze), so that a small batch exercises the overflow path too.
static std::atomic<long> g_seq_slot_limit{-1};
extern "C" __attribute__((visibility("default"))) void vlx_debug_set_seq_slots(long n) { g_seq_slot_limit.store(n, std::memory_order_relaxed); }
// bytes of the sequential-autocorrelation slots of a batch of n_chunks sub-blocks of 4-byte (or wider) samples
static size_t seq_buf_bytes(uint64_t n_chunks, long limit, size_t *cap_out)
{
    const size_t cap = (size_t)std::min<uint64_t>(n_chunks, limit >= 0 ? (uint64_t)limit : VLX_SEQ_MAX_SLOTS);
    if (cap_out) *cap_out = cap;
    return sizeof(int32_t) * 2 * VLX_SEQ_ROW * cap + sizeof(double) * 18 * cap + sizeof(SeqSlot) * cap + 64;
}


// Diagnostic, not in the public header (tests/, tools/): how many chunks of the context's last decode batch the fast
// kernel handed to the careful one.  Ordinary material must give 0: a fast path that flags chunks it could have decoded
// still round-trips bit-exactly, only many times slower — which no parity test sees.
extern "C" __attribute__((visibility("default"))) long vlx_debug_last_redo_chunks(vlx_ctx *c) { return c ? (long)c->last_redo : -1; }

int vlx_last_kernel_times(vlx_ctx *c, const char **names, float *ms, int cap)
{
    if (!c) return 0;
    int n = 0;
    for (auto &k : c->last_times) { if (n >= cap) break; if (names) names[n] = k.first; if (ms) ms[n] = k.second; n++; }
    return n;
}

// ------------------------------------------------------------------------------------ encode
#define track_chunks vlx_track_chunks

size_t vlx_encode_bound(const uint64_t *h_track_samples, uint32_t n_tracks)
{
    size_t total = 0;
    for (uint32_t t = 0; t < n_tracks; t++) {
        const uint64_t n = h_track_samples[t];
        // per chunk: u32 prefix + at most the raw block (10 B per pair + 1); +1 header byte; 256-aligned + slack
        size_t len = 1 + (size_t)track_chunks(n) * 5 + (size_t)(n / 2) * 10 + 8;
        total += (len + 4 + 255) & ~(size_t)255;
    }
    return total + 256;
}

static int dec_workspace_ensure(vlx_ctx *c, size_t n_chunks);

// Device workspace of an encode batch of n_chunks sub-blocks in n_tracks tracks.
static int enc_workspace_ensure(vlx_ctx *c, uint64_t n_chunks, uint32_t n_tracks)
{
    const uint64_t n_streams = 2 * n_chunks;
    HIPCHK(c, c->params.ensure(sizeof(StreamParams) * std::max<uint64_t>(n_streams, 1)));
    HIPCHK(c, c->res.ensure(sizeof(int32_t) * VLX_STREAM_LEN * std::max<uint64_t>(n_streams, 1)));
    HIPCHK(c, c->low.ensure((size_t)VLX_STREAM_LEN * std::max<uint64_t>(n_streams, 1) + 64));
    HIPCHK(c, c->staging.ensure(sizeof(uint32_t) * VLX_SLOT_DWORDS * std::max<uint64_t>(n_streams, 1) + 64));
    HIPCHK(c, c->stream_bits.ensure(sizeof(uint32_t) * std::max<uint64_t>(n_streams, 1)));
    HIPCHK(c, c->chunk_use_ms.ensure(sizeof(uint32_t) * std::max<uint64_t>(n_chunks, 1)));
    HIPCHK(c, c->chunk_info.ensure(sizeof(ChunkInfo) * std::max<uint64_t>(n_chunks, 1)));
    HIPCHK(c, c->asm_desc.ensure(sizeof(AsmDesc) * std::max<uint64_t>(n_chunks, 1)));
    HIPCHK(c, c->chunk_off.ensure(sizeof(uint64_t) * VLX_SCAN_ROOM(n_chunks)));
    HIPCHK(c, c->chunk_track.ensure(sizeof(uint32_t) * std::max<uint64_t>(n_chunks, 1)));
    HIPCHK(c, c->track_hires.ensure(sizeof(uint32_t) * VLX_HIRES_STRIDE * std::max<uint32_t>(n_tracks, 1)));
    HIPCHK(c, c->tracks.ensure(sizeof(TrackDesc) * std::max<uint32_t>(n_tracks, 1)));
    HIPCHK(c, c->track_len.ensure(sizeof(uint64_t) * ((size_t)n_tracks + VLX_SCAN_ROOM(n_tracks))));   // lengths, then the scanned offsets
    return VLX_OK;
}

// Shared driver.  hdr_bits/hdr_bytes describe, per track, the stream header WITHOUT the high_res bit
// (NULL = integer PCM: the single is_float=0 bit).  force_hires and the exchange hooks: vlx_host.h.
int vlx_encode_device_impl(vlx_ctx *c, const void *d_pcm, int fmt, const uint64_t *off, const uint64_t *ns, uint32_t n_tracks,
                           const std::vector<std::vector<uint8_t>> *hdr_bytes, const uint64_t *hdr_bits, const int *force_hires,
                           uint8_t *d_payload, size_t payload_cap, uint64_t *h_payload_off, uint64_t *h_payload_len,
                           const EncHook *hook, const uint8_t *d_hdr)
{
    // d_hdr (one track only): the stream header as a bit string ALREADY IN HBM (device-built float header,
    // vlx_float.hip), hdr_bits[0] bits long + 12 readable bytes beyond; its whole dwords are copied into the
    // payload by one device-to-device copy, k_assemble only builds the last ones (TrackDesc.high_res bit 0x400).
    if (d_hdr && n_tracks != 1) return VLX_ERR_ARG;
    if (!c || !ns || !n_tracks || !h_payload_off || !h_payload_len) return VLX_ERR_ARG;
    if (fmt != VLX_FMT_S16 && fmt != VLX_FMT_S24 && fmt != VLX_FMT_S32 && fmt != VLX_FMT_I64 &&
        fmt != VLX_FMT_F32_SPLIT && fmt != VLX_FMT_F32_D16 && fmt != VLX_FMT_F32_D24) return VLX_ERR_ARG;   // (the float32 formats: vlx_encode_f32_device only)
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<TrackDesc> tr(n_tracks);
    std::vector<uint8_t> hb;
    uint64_t n_chunks = 0;
    for (uint32_t t = 0; t < n_tracks; t++) {
        if (ns[t] & 1) { c->err = "odd sample totals go through vlx_encode_block"; return VLX_ERR_UNSUPPORTED; }
        if (off && (off[t] & 15)) { c->err = "track byte offsets must be multiples of 16"; return VLX_ERR_ARG; }
        if (ns[t] && !d_pcm) return VLX_ERR_ARG;
        TrackDesc &T = tr[t];
        T.byte_off = off ? off[t] : 0; T.n_samples = ns[t]; T.chunk_base = n_chunks;
        T.n_chunks = (uint32_t)track_chunks(ns[t]); T.out_off = 0;
        T.hdr_byte_off = hb.size();
        if (d_hdr) { T.hdr_bits = hdr_bits[t] + 1; hb.push_back(0); }
        else if (hdr_bytes) {
            T.hdr_bits = hdr_bits[t] + 1;
            hb.insert(hb.end(), (*hdr_bytes)[t].begin(), (*hdr_bytes)[t].end());
        } else { T.hdr_bits = 2; hb.push_back(0); }
        while (hb.size() % 4) hb.push_back(0);
        for (int k = 0; k < 8; k++) hb.push_back(0);                       // gather reads one dword ahead
        // force: -1 = scan on device; else bit0 = value, bit1 = continuation shard (the bit itself is not emitted)
        T.high_res = (force_hires && force_hires[t] >= 0) ? (0x100u | (uint32_t)(force_hires[t] & 1) | ((force_hires[t] & 2) ? 0x200u : 0u))
                                                          : ((force_hires && force_hires[t] == -3) ? 0x200u : 0u);
        if (d_hdr) T.high_res |= 0x400u;
        n_chunks += T.n_chunks;
    }
    if (n_chunks >= (1ull << 31)) { c->err = "too many chunks in one batch"; return VLX_ERR_UNSUPPORTED; }
    const uint64_t n_streams = 2 * n_chunks;
    size_t bound = vlx_encode_bound(ns, n_tracks);
    if (hdr_bits) for (uint32_t t = 0; t < n_tracks; t++) bound += (size_t)(hdr_bits[t] / 8) + 256;
    if (n_chunks && !d_payload) return VLX_ERR_ARG;
    if ((uintptr_t)d_payload & 15) { c->err = "d_payload must be 16-byte aligned"; return VLX_ERR_ARG; }

    { const int rc = enc_workspace_ensure(c, n_chunks, n_tracks); if (rc != VLX_OK) return rc; }
    HIPCHK(c, c->hdr_bytes.ensure(hb.size()));

    EncWorkspace ws;
    ws.params = (StreamParams *)c->params.p; ws.res = (int32_t *)c->res.p; ws.low = (uint8_t *)c->low.p;
    ws.staging = (uint32_t *)c->staging.p; ws.stream_bits = (uint32_t *)c->stream_bits.p;
    ws.chunk_use_ms = (uint32_t *)c->chunk_use_ms.p; ws.chunk_info = (ChunkInfo *)c->chunk_info.p; ws.asm_desc = (AsmDesc *)c->asm_desc.p;
    ws.chunk_off = (uint64_t *)c->chunk_off.p; ws.chunk_track = (uint32_t *)c->chunk_track.p; ws.track_hires = (uint32_t *)c->track_hires.p;
    ws.tracks = (TrackDesc *)c->tracks.p; ws.track_len = (uint64_t *)c->track_len.p; ws.hdr_bytes = (uint8_t *)c->hdr_bytes.p;
    if (d_hdr) ws.hdr_bytes = const_cast<uint8_t *>(d_hdr);
    ws.seq_xs = nullptr; ws.seq_ac = nullptr; ws.seq_slot = nullptr; ws.seq_count = nullptr; ws.seq_cap = 0;
    if (vlx_fmt_bytes(fmt) >= 4 && n_chunks && g_seq_slot_limit.load(std::memory_order_relaxed) != 0) {   // slots for chunks whose autocorrelation is the sequential fp64 sum
        const long lim = g_seq_slot_limit.load(std::memory_order_relaxed);
        size_t cap = 0;
        HIPCHK(c, c->seq_buf.ensure(seq_buf_bytes(n_chunks, lim, &cap)));
        const size_t xs_bytes = sizeof(int32_t) * 2 * VLX_SEQ_ROW * cap, ac_bytes = sizeof(double) * 18 * cap, sl_bytes = sizeof(SeqSlot) * cap;
        uint8_t *p = (uint8_t *)c->seq_buf.p;
        ws.seq_xs = (int32_t *)p; ws.seq_ac = (double *)(p + xs_bytes); ws.seq_slot = (SeqSlot *)(p + xs_bytes + ac_bytes);
        ws.seq_count = (uint32_t *)(p + xs_bytes + ac_bytes + sl_bytes); ws.seq_cap = (uint32_t)cap;
    }

    hipStream_t st = c->stream;
    // the track table and the header bits go up from pinned memory the context owns: no wait for the copy
    const size_t tr_bytes = sizeof(TrackDesc) * n_tracks, tr_pad = (tr_bytes + 15) & ~(size_t)15;
    HIPCHK(c, c->h_setup.ensure(tr_pad + hb.size()));
    memcpy(c->h_setup.p, tr.data(), tr_bytes);
    memcpy((uint8_t *)c->h_setup.p + tr_pad, hb.data(), hb.size());
    // (one kernel reads the staging area itself and clears the flags and the slot counter; a header of megabytes — the
    // exponent runs of a float stream — still goes up through the copy engine)
    const bool hdr_by_kernel = !d_hdr && hb.size() <= (1u << 20);
    if (!d_hdr && !hdr_by_kernel) HIPCHK(c, hipMemcpyAsync(ws.hdr_bytes, (uint8_t *)c->h_setup.p + tr_pad, hb.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, vlx_launch_enc_setup(st, c->h_setup.p, ws.tracks, tr_bytes, hdr_by_kernel ? ws.hdr_bytes : nullptr, tr_pad, hb.size(),
                                   ws.track_hires, (size_t)VLX_HIRES_STRIDE * n_tracks, ws.seq_count));

    HIPCHK(c, vlx_launch_chunk_track(st, ws, n_tracks, n_chunks));
    prof_begin(c, "k_hires_scan"); HIPCHK(c, vlx_launch_hires_scan(st, d_pcm, fmt, ws, n_tracks, n_chunks)); prof_end(c);
    if (hook && hook->flag_exchange) {                                    // file-global high_res over every rank's samples: stays in HBM
        prof_begin(c, "x_high_res_allreduce");
        const int rc = hook->flag_exchange(hook->self, ws.track_hires, (size_t)VLX_HIRES_STRIDE * n_tracks, st);
        prof_end(c);
        if (rc != VLX_OK) { (void)hipStreamSynchronize(st); prof_collect(c); return rc; }
    }
    prof_begin(c, "k_analyze"); HIPCHK(c, vlx_launch_analyze(st, d_pcm, fmt, ws, n_tracks, n_chunks)); prof_end(c);
    prof_begin(c, vlx_encode_kernel_label(n_streams)); HIPCHK(c, vlx_launch_encode_serial(st, ws, n_streams)); prof_end(c);
    prof_begin(c, "k_encode_wide"); HIPCHK(c, vlx_launch_encode_wide(st, d_pcm, fmt, ws, n_tracks, n_chunks)); prof_end(c);
    prof_begin(c, "k_layout"); HIPCHK(c, vlx_launch_layout(st, ws, n_tracks, n_chunks)); prof_end(c);

    const size_t n_lens = 2 * (size_t)n_tracks + 1;
    HIPCHK(c, c->h_lens.ensure(sizeof(uint64_t) * n_lens));
    HIPCHK(c, c->h_chunk_info.ensure(sizeof(ChunkInfo) * std::max<uint64_t>(n_chunks, 1)));
    uint64_t *lens = (uint64_t *)c->h_lens.p;
    // chunk byte sizes for vlx_encode_last_chunk_sizes come down BESIDE k_assemble, on the side stream (in front of it on
    // the context's stream the copy of cfg2's 675 KB held the kernel back by ~35 us)
    if (n_chunks) {
        HIPCHK(c, hipEventRecord(c->side_go, st));
        HIPCHK(c, hipStreamWaitEvent(c->side, c->side_go, 0));
        HIPCHK(c, hipMemcpyAsync(c->h_chunk_info.p, ws.chunk_info, sizeof(ChunkInfo) * n_chunks, hipMemcpyDeviceToHost, c->side));
    }
    const bool checked_first = payload_cap < bound;
    int status = VLX_OK;
    if (checked_first) {
        HIPCHK(c, hipMemcpyAsync(lens, ws.track_len, sizeof(uint64_t) * n_lens, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        const uint64_t need = lens[n_tracks + n_tracks];                 // offset past the last track
        if (need > payload_cap) { c->err = "payload buffer too small"; status = VLX_ERR_CAPACITY; }
    }
    if (hook && hook->after_layout) {                                     // the offset-table exchange rides in front of k_assemble; a rank
        prof_begin(c, "x_sizes_allgather");                               // that is about to fail still takes part and says so
        const int rc = hook->after_layout(hook->self, ws.chunk_info, n_chunks, status, st);
        prof_end(c);
        if (rc != VLX_OK && status == VLX_OK) status = rc;
    }
    if (status != VLX_OK) { (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(c->side); prof_collect(c); return status; }
    if (d_hdr && n_chunks) {                                              // the single track sits at payload offset 0
        const uint64_t S = tr[0].hdr_bits, nd = (S >> 5) > 2 ? (S >> 5) - 2 : 0;
        if (nd) HIPCHK(c, hipMemcpyAsync(d_payload, d_hdr, 4 * nd, hipMemcpyDeviceToDevice, st));
    }
    prof_begin(c, "k_assemble"); HIPCHK(c, vlx_launch_assemble(st, d_pcm, fmt, ws, n_tracks, n_chunks, d_payload)); prof_end(c);
    if (!checked_first)
        HIPCHK(c, hipMemcpyAsync(lens, ws.track_len, sizeof(uint64_t) * n_lens, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipStreamSynchronize(c->side));
    for (uint32_t t = 0; t < n_tracks; t++) { h_payload_len[t] = lens[t]; h_payload_off[t] = lens[n_tracks + t]; }
    // tracks without any chunk consist of their header bits only (VeloxCore.h:226-236, :305)
    for (uint32_t t = 0; t < n_tracks; t++) {
        if (tr[t].n_chunks) continue;
        std::vector<uint8_t> h((size_t)h_payload_len[t] + 4, 0);
        const uint8_t *src = hb.data() + tr[t].hdr_byte_off;
        const uint64_t nb = tr[t].hdr_bits - 1;
        for (uint64_t b = 0; b < nb; b++) if (src[b >> 3] >> (b & 7) & 1) h[b >> 3] |= (uint8_t)(1u << (b & 7));
        if (d_payload) HIPCHK(c, hipMemcpy(d_payload + h_payload_off[t], h.data(), (size_t)h_payload_len[t], hipMemcpyHostToDevice));
    }
    c->last_n_chunks = n_chunks;
    prof_collect(c);
    return VLX_OK;
}

int vlx_ctx_reserve_for(vlx_ctx *c, int fmt, uint64_t encode_chunks, uint32_t encode_tracks, uint64_t decode_chunks)
{
    if (!c) return VLX_ERR_ARG;
    if (fmt != VLX_FMT_S16 && fmt != VLX_FMT_S24 && fmt != VLX_FMT_S32 && fmt != VLX_FMT_I64) return VLX_ERR_ARG;
    if (encode_chunks >= (1ull << 31) || decode_chunks >= (1ull << 31)) { c->err = "too many chunks in one batch"; return VLX_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));                           // growing a buffer frees the old one
    if (encode_chunks) {
        const uint32_t nt = std::max<uint32_t>(encode_tracks, 1);
        const int rc = enc_workspace_ensure(c, encode_chunks, nt);
        if (rc != VLX_OK) return rc;
        // per track the host stages a TrackDesc and 1 header byte padded to a dword + 8 (vlx_encode_device_impl)
        const size_t tr_pad = (sizeof(TrackDesc) * nt + 15) & ~(size_t)15, hb = 12 * (size_t)nt;
        HIPCHK(c, c->hdr_bytes.ensure(hb));
        HIPCHK(c, c->h_setup.ensure(tr_pad + hb));
        HIPCHK(c, c->h_lens.ensure(sizeof(uint64_t) * (2 * (size_t)nt + 1)));
        HIPCHK(c, c->h_chunk_info.ensure(sizeof(ChunkInfo) * encode_chunks));
        if (vlx_fmt_bytes(fmt) >= 4 && g_seq_slot_limit.load(std::memory_order_relaxed) != 0)      // loud 32-bit material: the slots of k_seq_chains
            HIPCHK(c, c->seq_buf.ensure(seq_buf_bytes(encode_chunks, g_seq_slot_limit.load(std::memory_order_relaxed), nullptr)));
    }
    if (decode_chunks) {
        const int rc = dec_workspace_ensure(c, (size_t)decode_chunks);
        if (rc != VLX_OK) return rc;
        HIPCHK(c, c->dchunks.ensure(sizeof(vlx_chunk) * decode_chunks));
        HIPCHK(c, c->dscan.ensure(sizeof(uint64_t) * VLX_SCAN_ROOM(decode_chunks)));
        HIPCHK(c, c->dtracks.ensure(sizeof(DecTrack) * 16 + sizeof(uint32_t) * decode_chunks + 32));
        HIPCHK(c, c->h_chunks.ensure(std::max(sizeof(vlx_chunk) * (size_t)decode_chunks, sizeof(DecTrack) * 16 + sizeof(uint32_t) * (size_t)decode_chunks + 32)));
    }
    return VLX_OK;
}

int vlx_ctx_reserve(vlx_ctx *c, uint64_t encode_chunks, uint32_t encode_tracks, uint64_t decode_chunks)
{
    return vlx_ctx_reserve_for(c, VLX_FMT_S24, encode_chunks, encode_tracks, decode_chunks);
}

static int encode_odd_stage(vlx_ctx *c, const void *d_samples, int fmt, uint64_t n, const uint8_t *d_hdr, uint64_t hdr_bits, int force_hires, uint64_t *len);
static int upload_hdr(vlx_ctx *c, const std::vector<uint8_t> &hdr);

int vlx_encode_device(vlx_ctx *c, const void *d_pcm, int fmt, const uint64_t *off, const uint64_t *ns, uint32_t n_tracks,
                      uint8_t *d_payload, size_t payload_cap, uint64_t *h_payload_off, uint64_t *h_payload_len)
{
    if (fmt != VLX_FMT_S16 && fmt != VLX_FMT_S24 && fmt != VLX_FMT_S32 && fmt != VLX_FMT_I64) return VLX_ERR_ARG;
    if (!c || !ns || !n_tracks || !h_payload_off || !h_payload_len) return VLX_ERR_ARG;
    bool any_odd = false;
    for (uint32_t t = 0; t < n_tracks; t++) any_odd |= (ns[t] & 1) != 0;
    if (!any_odd)
        return vlx_encode_device_impl(c, d_pcm, fmt, off, ns, n_tracks, nullptr, nullptr, nullptr, d_payload, payload_cap,
                                      h_payload_off, h_payload_len);
    // Tracks with an odd total are one stream each (VeloxCore.h:242-253): the even ones go through the batch kernels as
    // one batch, every odd one through k_encode_odd, placed behind the batch at the next 256-byte boundary.
    if (!d_payload || ((uintptr_t)d_payload & 15)) { c->err = "d_payload must be 16-byte aligned"; return VLX_ERR_ARG; }
    std::vector<uint32_t> ev; std::vector<uint64_t> eoff, ens;
    for (uint32_t t = 0; t < n_tracks; t++) if (!(ns[t] & 1)) { ev.push_back(t); eoff.push_back(off ? off[t] : 0); ens.push_back(ns[t]); }
    uint64_t end = 0;
    c->last_n_chunks = 0;
    if (!ev.empty()) {
        std::vector<uint64_t> po(ev.size()), pl(ev.size());
        const int rc = vlx_encode_device_impl(c, d_pcm, fmt, eoff.data(), ens.data(), (uint32_t)ev.size(), nullptr, nullptr, nullptr, d_payload, payload_cap, po.data(), pl.data());
        if (rc != VLX_OK) return rc;
        for (size_t i = 0; i < ev.size(); i++) { h_payload_off[ev[i]] = po[i]; h_payload_len[ev[i]] = pl[i]; end = std::max(end, po[i] + pl[i]); }
    }
    HIPCHK(c, hipSetDevice(c->device));
    { const std::vector<uint8_t> hdr(1, 0); const int rc = upload_hdr(c, hdr); if (rc != VLX_OK) return rc; }   // is_float = 0
    for (uint32_t t = 0; t < n_tracks; t++) {
        if (!(ns[t] & 1)) continue;
        if (off && (off[t] & 15)) { c->err = "track byte offsets must be multiples of 16"; return VLX_ERR_ARG; }
        if (!d_pcm) return VLX_ERR_ARG;
        uint64_t len = 0;
        const int rc = encode_odd_stage(c, (const uint8_t *)d_pcm + (off ? off[t] : 0), fmt, ns[t], (const uint8_t *)c->hdr_bytes.p, 1, -1, &len);
        if (rc != VLX_OK) return rc;
        const uint64_t pos = (end + 255) & ~255ull;
        if (pos + len > payload_cap) { c->err = "payload buffer too small"; return VLX_ERR_CAPACITY; }
        HIPCHK(c, hipMemcpy(d_payload + pos, c->out_stage.p, len, hipMemcpyDeviceToDevice));
        h_payload_off[t] = pos; h_payload_len[t] = len; end = pos + len;
    }
    return VLX_OK;
}

// ---- multi-GPU building blocks: a track is cut into contiguous runs of sub-blocks, one per GPU
int vlx_scan_high_res_device(vlx_ctx *c, const void *d_pcm, int fmt, uint64_t n_samples, int *high_res)
{
    if (!c || !high_res || (n_samples && !d_pcm)) return VLX_ERR_ARG;
    if (fmt != VLX_FMT_S16 && fmt != VLX_FMT_S24 && fmt != VLX_FMT_S32 && fmt != VLX_FMT_I64) return VLX_ERR_ARG;
    *high_res = 0;
    if (fmt == VLX_FMT_S16 || n_samples < 2) {                            // |int16| never exceeds 65536
        if (fmt != VLX_FMT_S16 && n_samples == 1) return VLX_ERR_UNSUPPORTED;
        return VLX_OK;
    }
    if (n_samples & 1) { c->err = "shards hold an even number of samples"; return VLX_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    TrackDesc T{}; T.n_samples = n_samples; T.n_chunks = (uint32_t)track_chunks(n_samples); T.hdr_bits = 2;
    HIPCHK(c, c->tracks.ensure(sizeof(TrackDesc)));
    HIPCHK(c, c->track_hires.ensure(sizeof(uint32_t) * VLX_HIRES_STRIDE));
    HIPCHK(c, c->chunk_track.ensure(sizeof(uint32_t) * std::max<uint64_t>(T.n_chunks, 1)));
    EncWorkspace ws{}; ws.tracks = (TrackDesc *)c->tracks.p; ws.track_hires = (uint32_t *)c->track_hires.p;
    ws.chunk_track = (uint32_t *)c->chunk_track.p;
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemsetAsync(ws.chunk_track, 0, sizeof(uint32_t) * std::max<uint64_t>(T.n_chunks, 1), st));
    HIPCHK(c, hipMemcpyAsync(ws.tracks, &T, sizeof(T), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(ws.track_hires, 0, sizeof(uint32_t), st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, vlx_launch_hires_scan(st, d_pcm, fmt, ws, 1, T.n_chunks));
    uint32_t flag = 0;
    HIPCHK(c, hipMemcpyAsync(&flag, ws.track_hires, sizeof(flag), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *high_res = flag ? 1 : 0;
    return VLX_OK;
}

int vlx_encode_shard_device(vlx_ctx *c, const void *d_pcm, int fmt, uint64_t n_samples, int high_res, int first_shard,
                            uint8_t *d_payload, size_t payload_cap, uint64_t *payload_len)
{
    if (!c || !payload_len) return VLX_ERR_ARG;
    const uint64_t off = 0; uint64_t poff = 0;
    const int force = (high_res ? 1 : 0) | (first_shard ? 0 : 2);
    const int rc = vlx_encode_device_impl(c, d_pcm, fmt, &off, &n_samples, 1, nullptr, nullptr, &force, d_payload, payload_cap, &poff, payload_len);
    return rc;
}

int vlx_encode_last_chunk_sizes(vlx_ctx *c, uint32_t *h_sizes, size_t cap, size_t *n_chunks)
{
    if (!c || !n_chunks) return VLX_ERR_ARG;
    *n_chunks = (size_t)c->last_n_chunks;
    if (!h_sizes) return VLX_OK;
    if (cap < c->last_n_chunks) return VLX_ERR_CAPACITY;
    if (!c->last_n_chunks) return VLX_OK;
    const ChunkInfo *ci = (const ChunkInfo *)c->h_chunk_info.p;            // downloaded by the encode call itself
    for (size_t i = 0; i < c->last_n_chunks; i++) h_sizes[i] = ci[i].bytes;
    return VLX_OK;
}

// ---- float front-end on the device (vlx_float.hip) -------------------------------------------------
// DetectPseudoFloat over raw float32 samples in HBM: 16, 24 or 0 (VeloxFormat.h:14-58).
static int f32_detect(vlx_ctx *c, const void *d_raw, uint64_t n, int *det)
{
    HIPCHK(c, c->f_flags.ensure(64));
    hipStream_t st = c->stream;
    uint32_t ones[2] = {1u, 1u}, got[2] = {1u, 1u};
    HIPCHK(c, hipMemcpyAsync(c->f_flags.p, ones, sizeof(ones), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    prof_begin(c, "k_f32_detect"); HIPCHK(c, vlx_launch_f32_detect(st, d_raw, n, (uint32_t *)c->f_flags.p)); prof_end(c);
    HIPCHK(c, hipMemcpyAsync(got, c->f_flags.p, sizeof(got), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *det = got[0] ? 16 : (got[1] ? 24 : 0);
    return VLX_OK;
}

// EncodeRLE of an exponent plane in HBM + the three header bits in front: leaves the header bit string in
// c->f_hdr and says how many bits it has (3 + 16 per pair).
static int f32_build_header(vlx_ctx *c, const uint8_t *d_exps, uint64_t n_exps, uint64_t *hdr_bits)
{
    hipStream_t st = c->stream;
    uint64_t pairs = 0;
    if (n_exps) {
        HIPCHK(c, c->f_work.ensure(sizeof(uint64_t) * vlx_rle_work_entries(n_exps)));
        prof_begin(c, "k_rle_count"); HIPCHK(c, vlx_launch_rle_count(st, d_exps, n_exps, (uint64_t *)c->f_work.p)); prof_end(c);
        const uint64_t nb = (n_exps + VLX_RLE_BLOCK - 1) / VLX_RLE_BLOCK;
        HIPCHK(c, hipMemcpyAsync(&pairs, (uint64_t *)c->f_work.p + 2 * nb + nb, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        HIPCHK(c, c->f_rle.ensure(2 * pairs + 16));
        prof_begin(c, "k_rle_emit"); HIPCHK(c, vlx_launch_rle_emit(st, d_exps, n_exps, (uint64_t *)c->f_work.p, (uint8_t *)c->f_rle.p)); prof_end(c);
    } else HIPCHK(c, c->f_rle.ensure(16));
    const uint64_t n_hdr = (2 * pairs + 1 + 16 + 3) & ~3ull;               // header bytes + what k_assemble may read beyond
    HIPCHK(c, c->f_hdr.ensure(n_hdr));
    HIPCHK(c, vlx_launch_float_header(st, (const uint8_t *)c->f_rle.p, 2 * pairs, 0u, (uint8_t *)c->f_hdr.p, n_hdr));
    *hdr_bits = 3 + 16 * pairs;
    return VLX_OK;
}

int vlx_encode_f32_device(vlx_ctx *c, const void *d_f32, uint64_t n, uint8_t *d_payload, size_t payload_cap,
                          uint64_t *payload_len, int *float_mode)
{
    if (!c || !payload_len || (n && !d_f32)) return VLX_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int det = 16;                                                         // what the reference's loop says about an empty input
    if (n) { const int rc = f32_detect(c, d_f32, n, &det); if (rc != VLX_OK) return rc; }
    const int fm = det == 16 ? 1 : (det == 24 ? 2 : 0);
    if (float_mode) *float_mode = fm;
    const uint64_t off = 0; uint64_t poff = 0;
    hipStream_t st = c->stream;
    if ((uintptr_t)d_f32 & 15) { c->err = "float32 samples must be 16-byte aligned"; return VLX_ERR_ARG; }
    // The integers the codec works on are formed inside the encode kernels' sample loads (VLX_FMT_F32_D16 / _D24 /
    // _SPLIT): no int64 copy of the track is ever written.
    if (n & 1) {                                                          // one stream over the whole input (VeloxCore.h:242-253): k_encode_odd
        if (!d_payload) return VLX_ERR_ARG;
        uint64_t hdr_bits = 3, len = 0;
        const uint8_t *d_hdr = nullptr;
        if (fm) {
            const std::vector<uint8_t> hv(1, (uint8_t)(1u | ((unsigned)fm << 1)));
            const int rc = upload_hdr(c, hv); if (rc != VLX_OK) return rc;
            d_hdr = (const uint8_t *)c->hdr_bytes.p;
        } else {
            HIPCHK(c, c->f_exps.ensure(n + 64));
            prof_begin(c, "k_f32_exps"); HIPCHK(c, vlx_launch_f32_exps(st, d_f32, n, (uint8_t *)c->f_exps.p)); prof_end(c);
            const int rc = f32_build_header(c, (const uint8_t *)c->f_exps.p, n, &hdr_bits); if (rc != VLX_OK) return rc;
            d_hdr = (const uint8_t *)c->f_hdr.p;
        }
        const int rc = encode_odd_stage(c, d_f32, fm == 1 ? VLX_FMT_F32_D16 : (fm == 2 ? VLX_FMT_F32_D24 : VLX_FMT_F32_SPLIT), n, d_hdr, hdr_bits, fm ? -1 : 0, &len);
        if (rc != VLX_OK) return rc;
        if (len > payload_cap) { c->err = "payload buffer too small"; return VLX_ERR_CAPACITY; }
        HIPCHK(c, hipMemcpy(d_payload, c->out_stage.p, len, hipMemcpyDeviceToDevice));
        *payload_len = len; c->last_n_chunks = 0;
        return VLX_OK;
    }
    if (fm) {                                                             // VeloxCore.h:217-223: the demoted integers are what gets encoded
        std::vector<std::vector<uint8_t>> hv(1, std::vector<uint8_t>(1, (uint8_t)(1u | ((unsigned)fm << 1))));
        const uint64_t hdr_bits = 3; const int force = -1;                // :233: the high-res scan runs for pseudo float
        return vlx_encode_device_impl(c, d_f32, fm == 1 ? VLX_FMT_F32_D16 : VLX_FMT_F32_D24, &off, &n, 1, &hv, &hdr_bits, &force, d_payload, payload_cap, &poff, payload_len);
    }
    HIPCHK(c, c->f_exps.ensure(std::max<uint64_t>(n, 1) + 64));
    prof_begin(c, "k_f32_exps"); HIPCHK(c, vlx_launch_f32_exps(st, d_f32, n, (uint8_t *)c->f_exps.p)); prof_end(c);
    uint64_t hdr_bits = 3;
    { const int rc = f32_build_header(c, (const uint8_t *)c->f_exps.p, n, &hdr_bits); if (rc != VLX_OK) return rc; }
    const int force = 0;                                                  // :233: no high-res mode for true float
    return vlx_encode_device_impl(c, d_f32, VLX_FMT_F32_SPLIT, &off, &n, 1, nullptr, &hdr_bits, &force, d_payload, payload_cap, &poff, payload_len,
                              nullptr, (const uint8_t *)c->f_hdr.p);
}

int vlx_decode_f32_device(vlx_ctx *c, const uint8_t *d_payload, size_t payload_bytes, uint64_t total, const uint32_t *h_sizes, uint32_t n_sizes,
                          void *d_f32_out, int *float_mode, uint64_t *n_chunks_out)
{
    if (float_mode) *float_mode = 0;
    if (n_chunks_out) *n_chunks_out = 0;
    if (!c || !d_payload || !payload_bytes || (total && !d_f32_out) || (n_sizes && !h_sizes)) return VLX_ERR_ARG;
    if ((uintptr_t)d_payload & 3) { c->err = "d_payload must be 4-byte aligned"; return VLX_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    uint8_t first = 0;
    HIPCHK(c, hipMemcpyAsync(&first, d_payload, 1, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (!(first & 1)) { c->err = "not a float payload"; return VLX_ERR_UNSUPPORTED; }
    const int fm = (first >> 1) & 3;                                      // VeloxCore.h:325-326
    if (float_mode) *float_mode = fm;
    if (fm == 3) { c->err = "float_mode 3: no encoder writes it, the reference's callers treat it as integer PCM"; return VLX_ERR_UNSUPPORTED; }
    if (!total) return VLX_OK;
    if ((uintptr_t)d_f32_out & 3) { c->err = "d_f32_out must be 4-byte aligned"; return VLX_ERR_ARG; }
    HIPCHK(c, hipMemsetAsync(d_f32_out, 0, sizeof(float) * total, st));                // samples the stream does not reach stay 0 (main.cpp:254-261): +0.0f either way
    uint64_t hdr_bits = 3;
    if (fm == 0) {                                                        // DecodeRLE (VeloxCore.h:196-203, :327), in slabs of pairs
        HIPCHK(c, c->f_exps.ensure(total + 64));
        HIPCHK(c, hipMemsetAsync(c->f_exps.p, 0, total, st));
        const uint64_t max_pairs = payload_bytes * 8 > 3 ? (payload_bytes * 8 - 3 + 15) / 16 : 0;
        const uint64_t SLAB = 1ull << 22;
        HIPCHK(c, c->f_work.ensure(sizeof(uint64_t) * (VLX_SCAN_ROOM(SLAB) + 8)));
        HIPCHK(c, c->f_flags.ensure(64));
        uint64_t base = 0, p0 = 0; bool done = false;
        while (!done && p0 < max_pairs) {
            const uint64_t np = std::min<uint64_t>(SLAB, max_pairs - p0);
            unsigned long long sentinel = ~0ull;
            HIPCHK(c, hipMemcpyAsync(c->f_flags.p, &sentinel, sizeof(sentinel), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));
            HIPCHK(c, vlx_launch_rle_runs(st, d_payload, payload_bytes, p0, np, (uint64_t *)c->f_work.p));
            HIPCHK(c, vlx_launch_rle_expand(st, d_payload, payload_bytes, p0, np, (const uint64_t *)c->f_work.p, base, total,
                                            (uint8_t *)c->f_exps.p, (unsigned long long *)c->f_flags.p));
            unsigned long long first_done = 0; uint64_t slab_sum = 0;
            HIPCHK(c, hipMemcpyAsync(&first_done, c->f_flags.p, sizeof(first_done), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(&slab_sum, (uint64_t *)c->f_work.p + np, sizeof(slab_sum), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (first_done != ~0ull) { hdr_bits = 3 + 16 * (p0 + first_done + 1); done = true; }
            else { base += slab_sum; p0 += np; }
        }
        if (!done) hdr_bits = 3 + 16 * max_pairs;                         // truncated: the reference would read zero pairs for ever; nothing is left to decode
    }
    vlx_track_sizes T{};
    T.base_bit_off = 0; T.out_base = 0; T.total_samples = total; T.n_sizes = n_sizes; T.high_res = VLX_HIGH_RES_FROM_PAYLOAD; T.hdr_bits = (uint32_t)(hdr_bits + 1);
    if (hdr_bits + 1 > 0xFFFFFFFFull) { c->err = "float header too long for this entry point"; return VLX_ERR_UNSUPPORTED; }
    size_t n_chunks = 0;
    if (n_sizes && hdr_bits + 1 < 8ull * payload_bytes) {
        // MergeFloat32 / PromoteIntToFloat (main.cpp:262-281) ride in k_decode_finalize's store: no int64 pass in between
        c->fin_fmt_override = fm == 0 ? VLX_FMT_F32_MERGE : (fm == 1 ? VLX_FMT_F32_P16 : VLX_FMT_F32_P24);
        c->fin_exps = fm == 0 ? (const uint8_t *)c->f_exps.p : nullptr;
        const int rc = vlx_decode_tracks_device(c, d_payload, payload_bytes, &T, 1, h_sizes, d_f32_out, VLX_FMT_S32, &n_chunks);
        c->fin_fmt_override = 0; c->fin_exps = nullptr;
        if (rc != VLX_OK) return rc;
    }
    if (n_chunks_out) *n_chunks_out = n_chunks;
    HIPCHK(c, hipStreamSynchronize(st));
    prof_collect(c);
    return VLX_OK;
}

// ---- float-path prologue (host, serial — as in the reference) --------------------------------
// DetectPseudoFloat (VeloxFormat.h:14-58)
static int detect_pseudo_float(const uint8_t *raw, size_t count)
{
    bool fit16 = true, fit24 = true;
    const size_t stride = count > 100000 ? 4 : 1;
    for (size_t i = 0; i < count; i += stride) {
        float f; memcpy(&f, raw + 4 * i, 4);
        if (f == 0.0f || !std::isfinite(f)) continue;
        if (fit16) { float s = f * 32768.0f; int32_t q = (int32_t)std::round(s); if ((float)q * (1.0f / 32768.0f) != f) fit16 = false; }
        if (fit24) { float s = f * 8388608.0f; int32_t q = (int32_t)std::round(s); if ((float)q * (1.0f / 8388608.0f) != f) fit24 = false; }
        if (!fit16 && !fit24) return 0;
    }
    return fit16 ? 16 : (fit24 ? 24 : 0);
}

struct HostBits {
    std::vector<uint8_t> b; uint64_t n = 0;
    void put(uint64_t v, int k) { for (int i = 0; i < k; i++) { if ((n & 7) == 0) b.push_back(0); if ((v >> i) & 1) b.back() |= (uint8_t)(1u << (n & 7)); n++; } }
};

static int encode_host_common(vlx_ctx *c, const void *h_in, size_t n_samples, int fmt, const std::vector<uint8_t> *hdr, uint64_t hdr_bits,
                              int force_hires, uint8_t **payload, size_t *payload_len);

// Odd totals: the whole input is ONE stream and there is no use_MS bit (VeloxCore.h:242-253): k_encode_odd, one
// lane for the sample-serial part (about 0.4 us per sample: the reference's own decoder cannot read these payloads,
// SURVEY §4, so this is encode parity and nothing anybody waits for).  Leaves the payload at the start of c->out_stage.
// d_hdr: the header bits WITHOUT high_res, in device memory.
static int encode_odd_stage(vlx_ctx *c, const void *d_samples, int fmt, uint64_t n, const uint8_t *d_hdr, uint64_t hdr_bits, int force_hires, uint64_t *len)
{
    // One lane at ~0.4 us per sample: VLX_ODD_MAX_SAMPLES keeps the kernel under half a minute, well inside the compute
    // queue's lockup limit (a longer single-wave kernel risks a reset of the GPU for everybody on it).
    if (n > VLX_ODD_MAX_SAMPLES) { c->err = "odd sample total beyond VLX_ODD_MAX_SAMPLES: the one-stream path runs on one lane"; return VLX_ERR_UNSUPPORTED; }
    const uint64_t cap_dwords = (hdr_bits + 34 + 40 * n + 63) / 32 + 2;
    HIPCHK(c, c->out_stage.ensure(cap_dwords * 4));
    HIPCHK(c, c->track_len.ensure(2 * sizeof(uint64_t)));
    hipStream_t st = c->stream;
    prof_begin(c, "k_encode_odd");
    HIPCHK(c, vlx_launch_encode_odd(st, d_samples, fmt, n, force_hires, d_hdr, hdr_bits, (uint32_t *)c->out_stage.p, cap_dwords, (uint64_t *)c->track_len.p));
    prof_end(c);
    uint64_t res[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(res, c->track_len.p, sizeof(res), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    prof_collect(c);
    if (res[0] > cap_dwords * 4) { c->err = "odd-path payload larger than its bound"; return VLX_ERR_CAPACITY; }
    *len = res[0];
    return VLX_OK;
}

static int upload_hdr(vlx_ctx *c, const std::vector<uint8_t> &hdr)
{
    HIPCHK(c, c->hdr_bytes.ensure(hdr.size() + 8));
    if (!hdr.empty()) HIPCHK(c, hipMemcpyAsync(c->hdr_bytes.p, hdr.data(), hdr.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                            // hdr is caller-scoped pageable memory
    return VLX_OK;
}

static int vlx_encode_odd_device(vlx_ctx *c, const void *d_samples, int fmt, uint64_t n, const std::vector<uint8_t> &hdr, uint64_t hdr_bits,
                                 int force_hires, uint8_t **payload, size_t *payload_len)
{
    { const int rc = upload_hdr(c, hdr); if (rc != VLX_OK) return rc; }
    uint64_t len = 0;
    { const int rc = encode_odd_stage(c, d_samples, fmt, n, (const uint8_t *)c->hdr_bytes.p, hdr_bits, force_hires, &len); if (rc != VLX_OK) return rc; }
    uint8_t *out = (uint8_t *)malloc(len ? len : 1);
    if (!out) return VLX_ERR_NOMEM;
    if (hipMemcpy(out, c->out_stage.p, len, hipMemcpyDeviceToHost) != hipSuccess) { free(out); c->err = "download of the odd-path payload failed"; return VLX_ERR_HIP; }
    *payload = out; *payload_len = (size_t)len;
    return VLX_OK;
}

// vlx_encode_block's float case for even totals: DetectPseudoFloat, DemoteFloatToInt and EncodeRLE as device
// kernels over the uploaded buffers.  The caller's `samples` receive the demoted integers, as ProcessBlock
// leaves them (VeloxCore.h:217-223); for true float its mantissas and exponent bytes are what is encoded.
static int encode_float_block_device(vlx_ctx *c, int64_t *samples, size_t n, const uint8_t *exps, size_t n_exps,
                                     const uint8_t *raw_bytes, uint8_t **payload, size_t *payload_len)
{
    *payload = nullptr; *payload_len = 0;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIPCHK(c, c->in_stage.ensure(4 * n + 64));
    HIPCHK(c, c->f_samples.ensure(sizeof(int64_t) * n + 64));
    HIPCHK(c, hipMemcpyAsync(c->in_stage.p, raw_bytes, 4 * n, hipMemcpyHostToDevice, st));
    int det = 0;
    { const int rc = f32_detect(c, c->in_stage.p, n, &det); if (rc != VLX_OK) return rc; }
    uint64_t ns = n, off = 0, poff = 0, plen = 0, hdr_bits = 3;
    int rc;
    if (det) {
        prof_begin(c, "k_f32_demote"); HIPCHK(c, vlx_launch_f32_demote(st, c->in_stage.p, n, det, (int64_t *)c->f_samples.p)); prof_end(c);
        HIPCHK(c, hipMemcpyAsync(samples, c->f_samples.p, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
        HIPCHK(c, c->out_stage.ensure(vlx_encode_bound(&ns, 1) + 256));
        std::vector<std::vector<uint8_t>> hv(1, std::vector<uint8_t>(1, (uint8_t)(1u | ((det == 16 ? 1u : 2u) << 1))));
        const int force = -1;
        rc = vlx_encode_device_impl(c, c->f_samples.p, VLX_FMT_I64, &off, &ns, 1, &hv, &hdr_bits, &force, (uint8_t *)c->out_stage.p, c->out_stage.cap, &poff, &plen);
    } else {
        HIPCHK(c, hipMemcpyAsync(c->f_samples.p, samples, sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
        HIPCHK(c, c->f_exps.ensure(n_exps + 64));
        if (n_exps) HIPCHK(c, hipMemcpyAsync(c->f_exps.p, exps, n_exps, hipMemcpyHostToDevice, st));
        rc = f32_build_header(c, (const uint8_t *)c->f_exps.p, n_exps, &hdr_bits);
        if (rc != VLX_OK) return rc;
        HIPCHK(c, c->out_stage.ensure(vlx_encode_bound(&ns, 1) + (size_t)(hdr_bits / 8) + 256));
        const int force = 0;
        rc = vlx_encode_device_impl(c, c->f_samples.p, VLX_FMT_I64, &off, &ns, 1, nullptr, &hdr_bits, &force, (uint8_t *)c->out_stage.p, c->out_stage.cap, &poff, &plen,
                                nullptr, (const uint8_t *)c->f_hdr.p);
    }
    if (rc != VLX_OK) return rc;
    uint8_t *out = (uint8_t *)malloc(plen ? plen : 1);
    if (!out) return VLX_ERR_NOMEM;
    if (hipMemcpy(out, (uint8_t *)c->out_stage.p + poff, plen, hipMemcpyDeviceToHost) != hipSuccess) { free(out); c->err = "download of the payload failed"; return VLX_ERR_HIP; }
    *payload = out; *payload_len = (size_t)plen;
    return VLX_OK;
}

int vlx_encode_block(vlx_ctx *c, int64_t *samples, size_t n, int is_float, const uint8_t *exps, size_t n_exps,
                     const uint8_t *raw_bytes, uint8_t **payload, size_t *payload_len)
{
    if (!c || !payload || !payload_len || (n && !samples)) return VLX_ERR_ARG;
    HostBits hb; int force = -1;
    if (is_float && n && !(n & 1)) {                                      // the float prologue runs on the device (vlx_float.hip)
        if (!raw_bytes || (n_exps && !exps)) return VLX_ERR_ARG;
        return encode_float_block_device(c, samples, n, exps, n_exps, raw_bytes, payload, payload_len);
    }
    if (is_float) {                                                       // VeloxCore.h:214-230; empty and odd totals only (cold)
        if (!raw_bytes && n) return VLX_ERR_ARG;
        int float_mode = 0;
        const int det = detect_pseudo_float(raw_bytes, n);
        if (det == 16 || det == 24) {                                     // DemoteFloatToInt (VeloxFormat.h:60-70)
            float_mode = det == 16 ? 1 : 2;
            const double scale = det == 16 ? 32768.0 : 8388608.0;
            for (size_t i = 0; i < n; i++) { float f; memcpy(&f, raw_bytes + 4 * i, 4); samples[i] = (int64_t)std::round(f * scale); }
        }
        hb.put(1, 1); hb.put((uint64_t)float_mode, 2);
        if (float_mode == 0) {                                            // EncodeRLE (VeloxCore.h:187-195)
            if (n_exps) {
                uint8_t last = exps[0]; int run = 0;
                for (size_t i = 0; i < n_exps; i++) {
                    if (exps[i] == last && run < 255) run++;
                    else { hb.put((uint64_t)run, 8); hb.put(last, 8); last = exps[i]; run = 1; }
                }
                hb.put((uint64_t)run, 8); hb.put(last, 8);
            }
            force = 0;                                                    // :233 — no high-res scan for true float
        }
    } else hb.put(0, 1);
    return encode_host_common(c, samples, n, VLX_FMT_I64, &hb.b, hb.n, force, payload, payload_len);
}

int vlx_encode_pcm(vlx_ctx *c, const uint8_t *pcm, size_t n, int bits, uint8_t **payload, size_t *payload_len)
{
    if (!c || !payload || !payload_len || (n && !pcm)) return VLX_ERR_ARG;
    if (bits != 16 && bits != 24 && bits != 32) return VLX_ERR_ARG;
    HostBits hb; hb.put(0, 1);
    return encode_host_common(c, pcm, n, bits, &hb.b, hb.n, -1, payload, payload_len);
}

static int encode_host_common(vlx_ctx *c, const void *h_in, size_t n, int fmt, const std::vector<uint8_t> *hdr, uint64_t hdr_bits,
                              int force_hires, uint8_t **payload, size_t *payload_len)
{
    *payload = nullptr; *payload_len = 0;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bps = fmt == VLX_FMT_I64 ? 8 : (size_t)fmt / 8;
    HIPCHK(c, c->in_stage.ensure(n * bps + 64));
    if (n) HIPCHK(c, hipMemcpyAsync(c->in_stage.p, h_in, n * bps, hipMemcpyHostToDevice, c->stream));
    if (n & 1) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return vlx_encode_odd_device(c, c->in_stage.p, fmt, n, *hdr, hdr_bits, force_hires, payload, payload_len);
    }
    uint64_t ns = n, off = 0, poff = 0, plen = 0;
    std::vector<std::vector<uint8_t>> hv(1, *hdr);
    size_t bound = vlx_encode_bound(&ns, 1) + (size_t)(hdr_bits / 8) + 256;
    HIPCHK(c, c->out_stage.ensure(bound));
    int rc = vlx_encode_device_impl(c, c->in_stage.p, fmt, &off, &ns, 1, &hv, &hdr_bits, &force_hires,
                                (uint8_t *)c->out_stage.p, c->out_stage.cap, &poff, &plen);
    if (rc != VLX_OK) return rc;
    uint8_t *out = (uint8_t *)malloc(plen ? plen : 1);
    if (!out) return VLX_ERR_NOMEM;
    if (hipMemcpy(out, (uint8_t *)c->out_stage.p + poff, plen, hipMemcpyDeviceToHost) != hipSuccess) { free(out); c->err = "download of the payload failed"; return VLX_ERR_HIP; }
    *payload = out; *payload_len = (size_t)plen;
    return VLX_OK;
}

// ------------------------------------------------------------------------------------ decode
// LSB-first reader over host bytes (BitStreamReader, VeloxEntropy.h:46-89): reads past the end are 0.
struct HostReader {
    const uint8_t *d; uint64_t size_bits; uint64_t pos;
    uint64_t read(int n)
    {
        uint64_t v = 0;
        for (int i = 0; i < n; i++) { if (pos < size_bits) { v |= (uint64_t)((d[pos >> 3] >> (pos & 7)) & 1) << i; pos++; } }
        return v;
    }
};

// One refill step of StreamingDecoder::DecodeNext (VeloxCore.h:338-353) without the decoding:
// returns 1 and fills *ch, 0 on a zero chunk size.
static int walk_one(HostReader &r, uint64_t total, uint64_t &decoded, uint64_t base_bit, uint64_t out_base, uint32_t hires, vlx_chunk *ch)
{
    const uint64_t before = r.pos;
    uint32_t sz = (uint32_t)r.read(32);
    if (r.pos - before < 32) r.pos = before + 32 > r.size_bits ? r.size_bits : before + 32;
    if (sz == 0) return 0;
    const uint64_t body = before + 32;                                     // where the body would start
    const uint64_t body_bits = 8ull * sz;
    uint64_t avail = body < r.size_bits ? r.size_bits - body : 0;
    const uint64_t remaining = total - decoded;
    uint64_t frames = std::min<uint64_t>(4096, remaining / 2);
    if (frames == 0 && remaining > 0) frames = remaining;
    // a size prefix cut short by the end of the data still names a chunk (the missing bits read as zero,
    // VeloxEntropy.h:61-62): its body holds no bits and is placed at the end of the data, not beyond it
    ch->bit_off = base_bit + std::min(body, r.size_bits);
    ch->n_bits = std::min(body_bits, avail);
    ch->out_off = out_base + decoded;
    ch->frames = (uint32_t)frames;
    ch->n_out = (uint32_t)std::min<uint64_t>(2 * frames, remaining);
    ch->high_res = hires; ch->reserved = 0;
    decoded += ch->n_out;
    r.pos = body + body_bits;                                              // may run past the end; reads there give 0
    return 1;
}

long vlx_walk_chunks(const uint8_t *payload, size_t size, size_t total, uint64_t base_bit_off, uint64_t out_base,
                     vlx_chunk *chunks, size_t cap)
{
    if ((!payload && size) || (!chunks && cap)) return VLX_ERR_ARG;
    HostReader r{payload, 8ull * size, 0};
    if (r.read(1)) return VLX_ERR_UNSUPPORTED;                             // float payloads: vlx_decoder_*
    const uint32_t hires = (uint32_t)r.read(1);
    uint64_t decoded = 0; size_t n = 0;
    while (decoded < total) {
        vlx_chunk ch;
        if (r.pos >= r.size_bits) break;
        if (!walk_one(r, total, decoded, base_bit_off, out_base, hires, &ch)) break;
        if (n >= cap) return VLX_ERR_CAPACITY;
        chunks[n++] = ch;
    }
    return (long)n;
}

long vlx_chunks_from_sizes(const uint32_t *sizes, size_t n_sizes, size_t total, uint64_t base_bit_off, uint64_t out_base,
                           uint32_t high_res, uint32_t hdr_bits, vlx_chunk *chunks, size_t cap)
{
    if ((!sizes && n_sizes) || (!chunks && cap)) return VLX_ERR_ARG;
    uint64_t pos = hdr_bits, decoded = 0; size_t n = 0;                   // the same walk, sizes from the table
    for (; n < n_sizes && decoded < total; n++) {
        if (sizes[n] == 0) break;                                         // VeloxCore.h:341
        if (n >= cap) return VLX_ERR_CAPACITY;
        const uint64_t remaining = total - decoded;
        uint64_t frames = std::min<uint64_t>(4096, remaining / 2);
        if (frames == 0) frames = remaining;
        vlx_chunk &ch = chunks[n];
        ch.bit_off = base_bit_off + pos + 32;
        ch.n_bits = 8ull * sizes[n];
        ch.out_off = out_base + decoded;
        ch.frames = (uint32_t)frames;
        ch.n_out = (uint32_t)std::min<uint64_t>(2 * frames, remaining);
        ch.high_res = high_res; ch.reserved = 0;
        decoded += ch.n_out;
        pos += 32 + ch.n_bits;
    }
    return (long)n;
}

static int dec_workspace_ensure(vlx_ctx *c, size_t n_chunks)
{
    const size_t scratch_bytes = sizeof(int32_t) * ((n_chunks + 63) / 64 * 64) * 2 * VLX_STREAM_LEN;   // whole waves of 64 chunks
    HIPCHK(c, c->scratch.ensure(scratch_bytes));
    HIPCHK(c, c->chunk_state.ensure(sizeof(uint32_t) * 8 * std::max<size_t>(n_chunks, 1)));
    HIPCHK(c, c->wide_count.ensure(sizeof(uint32_t)));
    const size_t slot_bytes = sizeof(int32_t) * 2 * VLX_STREAM_LEN;
    if (!c->hi_pool.p) HIPCHK(c, c->hi_pool.ensure(slot_bytes * 16));     // 16 slots = 512 KB to start with
    return VLX_OK;
}

// Decode the n_chunks descriptors that sit in c->dchunks (uploaded or built there, in stream order).
// may_fuse: some chunk of the batch may be without high-res low bytes, so that a stream wave can finish it by
// itself when the output is packed 16 / 24-bit PCM (k_decode_streams<FUSE>); each chunk is still checked on the device.
// counter_cleared: the caller's last kernel has zeroed the wide-chunk counter already (vlx_launch_chunks_build).
static int decode_chunks_impl(vlx_ctx *c, const uint8_t *d_payload, size_t payload_bytes, size_t n_chunks, void *d_out, int out_fmt, bool may_fuse,
                              bool counter_cleared = false)
{
    const int fuse_fmt = may_fuse && (out_fmt == VLX_FMT_S16 || out_fmt == VLX_FMT_S24) ? out_fmt : 0;
    { const int rc = dec_workspace_ensure(c, n_chunks); if (rc != VLX_OK) return rc; }
    const size_t slot_bytes = sizeof(int32_t) * 2 * VLX_STREAM_LEN;
    hipStream_t st = c->stream;
    for (int attempt = 0; attempt < 2; attempt++) {
        DecWorkspace ws; ws.chunks = (vlx_chunk *)c->dchunks.p; ws.scratch = (int32_t *)c->scratch.p; ws.chunk_state = (uint32_t *)c->chunk_state.p;
        ws.hi_pool = (int32_t *)c->hi_pool.p; ws.hi_cap = (uint32_t)(c->hi_pool.cap / slot_bytes); ws.wide_count = (uint32_t *)c->wide_count.p;
        ws.exps = c->fin_exps;
        if (attempt || !counter_cleared) HIPCHK(c, hipMemsetAsync(ws.wide_count, 0, sizeof(uint32_t), st));
        prof_begin(c, vlx_decode_kernel_label(n_chunks)); HIPCHK(c, vlx_launch_decode_streams(st, d_payload, payload_bytes, ws, n_chunks, d_out, fuse_fmt)); prof_end(c);
        prof_begin(c, "k_decode_serial_wide"); HIPCHK(c, vlx_launch_decode_serial(st, d_payload, ws, n_chunks)); prof_end(c);
        prof_begin(c, "k_decode_finalize"); HIPCHK(c, vlx_launch_decode_finalize(st, d_payload, ws, n_chunks, d_out, c->fin_fmt_override ? c->fin_fmt_override : out_fmt)); prof_end(c);
        uint32_t n_wide = 0;
        HIPCHK(c, hipMemcpyAsync(&n_wide, ws.wide_count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        prof_collect(c);
        c->last_redo = n_wide;
        if (n_wide <= ws.hi_cap) break;
        // more chunks carried values beyond int32 than the pool holds (only the int64 boundary gets here):
        // grow the pool to fit and decode the batch once more
        HIPCHK(c, c->hi_pool.ensure(slot_bytes * (size_t)n_wide));
    }
    return VLX_OK;
}

int vlx_decode_device(vlx_ctx *c, const uint8_t *d_payload, size_t payload_bytes, const vlx_chunk *h_chunks, size_t n_chunks,
                      void *d_out, int out_fmt)
{
    if (!c || (n_chunks && (!d_payload || !h_chunks || !d_out))) return VLX_ERR_ARG;
    if (out_fmt != VLX_FMT_S16 && out_fmt != VLX_FMT_S24 && out_fmt != VLX_FMT_S32 && out_fmt != VLX_FMT_I64) return VLX_ERR_ARG;
    if (!n_chunks) return VLX_OK;
    if (!payload_bytes) { c->err = "empty payload buffer"; return VLX_ERR_ARG; }
    if ((uintptr_t)d_payload & 3) { c->err = "d_payload must be 4-byte aligned"; return VLX_ERR_ARG; }
    if (n_chunks >= (1ull << 31)) { c->err = "too many chunks in one batch"; return VLX_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    // checked while being copied into pinned memory, from where the upload is asynchronous
    HIPCHK(c, c->h_chunks.ensure(sizeof(vlx_chunk) * n_chunks));
    vlx_chunk *staged = (vlx_chunk *)c->h_chunks.p;
    bool any_lowres = false;
    for (size_t i = 0; i < n_chunks; i++) {
        const vlx_chunk ch = h_chunks[i];
        if (ch.frames > VLX_STREAM_LEN || ch.n_out > 2 * ch.frames) { c->err = "bad chunk descriptor"; return VLX_ERR_ARG; }
        if (ch.n_bits > 8ull * payload_bytes || ch.bit_off > 8ull * payload_bytes - ch.n_bits) { c->err = "chunk beyond payload"; return VLX_ERR_ARG; }
        any_lowres |= ch.high_res == 0;
        staged[i] = ch;
    }
    HIPCHK(c, c->dchunks.ensure(sizeof(vlx_chunk) * n_chunks));
    HIPCHK(c, hipMemcpyAsync(c->dchunks.p, staged, sizeof(vlx_chunk) * n_chunks, hipMemcpyHostToDevice, c->stream));
    return decode_chunks_impl(c, d_payload, payload_bytes, n_chunks, d_out, out_fmt, any_lowres);
}

// vlx_decode_device without any allocation: the workspace is this GPU's copy of `scratch`, the chunk table
// goes up from the pinned area vlx_ctx_create set aside.  Same kernels, same order.
int vlx_decode_device_noalloc(vlx_ctx *c, const uint8_t *d_payload, size_t payload_bytes, const vlx_chunk *h_chunks, size_t n_chunks,
                              void *d_out, int out_fmt)
{
    if (!c || (n_chunks && (!d_payload || !h_chunks || !d_out))) return VLX_ERR_ARG;
    if (out_fmt != VLX_FMT_S16 && out_fmt != VLX_FMT_S24 && out_fmt != VLX_FMT_S32 && out_fmt != VLX_FMT_I64) return VLX_ERR_ARG;
    if (!n_chunks) return VLX_OK;
    if (n_chunks > VLX_NOALLOC_MAX_CHUNKS) { c->err = "vlx_decode_device_noalloc takes at most VLX_NOALLOC_MAX_CHUNKS (1024) chunks per call"; return VLX_ERR_CAPACITY; }
    if (!payload_bytes) { c->err = "empty payload buffer"; return VLX_ERR_ARG; }
    if ((uintptr_t)d_payload & 3) { c->err = "d_payload must be 4-byte aligned"; return VLX_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    vlx_chunk *staged = (vlx_chunk *)c->na_host;
    uint32_t *h_wide = (uint32_t *)(c->na_host + sizeof(vlx_chunk) * VLX_NOALLOC_MAX_CHUNKS);
    for (size_t i = 0; i < n_chunks; i++) {
        const vlx_chunk ch = h_chunks[i];
        if (ch.frames > VLX_STREAM_LEN || ch.n_out > 2 * ch.frames) { c->err = "bad chunk descriptor"; return VLX_ERR_ARG; }
        if (ch.n_bits > 8ull * payload_bytes || ch.bit_off > 8ull * payload_bytes - ch.n_bits) { c->err = "chunk beyond payload"; return VLX_ERR_ARG; }
        staged[i] = ch;
    }
    uint8_t *const base = c->na_scratch;                                   // the device address, never the array's name
    DecWorkspace ws;
    ws.chunks = (vlx_chunk *)(base + NA_OFF_CHUNKS); ws.chunk_state = (uint32_t *)(base + NA_OFF_STATE);
    ws.wide_count = (uint32_t *)(base + NA_OFF_COUNT); ws.hi_pool = (int32_t *)(base + NA_OFF_HI); ws.hi_cap = VLX_NOALLOC_HI_SLOTS;
    ws.scratch = (int32_t *)(base + NA_OFF_SCRATCH); ws.exps = nullptr;
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemcpyAsync(ws.chunks, staged, sizeof(vlx_chunk) * n_chunks, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(ws.wide_count, 0, sizeof(uint32_t), st));    // the one word that must start at zero
    HIPCHK(c, vlx_launch_decode_streams(st, d_payload, payload_bytes, ws, n_chunks));
    HIPCHK(c, vlx_launch_decode_serial(st, d_payload, ws, n_chunks));
    HIPCHK(c, vlx_launch_decode_finalize(st, d_payload, ws, n_chunks, d_out, out_fmt));
    HIPCHK(c, hipMemcpyAsync(h_wide, ws.wide_count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (*h_wide > VLX_NOALLOC_HI_SLOTS) { c->err = "more than VLX_NOALLOC_HI_SLOTS chunks with values beyond int32: use vlx_decode_device"; return VLX_ERR_CAPACITY; }
    return VLX_OK;
}

// Chunks of a track that the reference's walk visits (VeloxCore.h:336-353) when every size is non-zero:
// 8192 samples per chunk, an even tail, and one more chunk for the last sample of an odd total.
static inline uint64_t chunks_for_total(uint64_t total)
{
    const uint64_t rem = total % (2 * VLX_STREAM_LEN);
    return total / (2 * VLX_STREAM_LEN) + (rem >= 2 ? 1 : 0) + (rem & 1);
}

int vlx_decode_tracks_device(vlx_ctx *c, const uint8_t *d_payload, size_t payload_bytes, const vlx_track_sizes *h_tracks, uint32_t n_tracks,
                             const uint32_t *h_sizes, void *d_out, int out_fmt, size_t *n_chunks_out)
{
    if (n_chunks_out) *n_chunks_out = 0;
    if (!c || (n_tracks && !h_tracks)) return VLX_ERR_ARG;
    if (out_fmt != VLX_FMT_S16 && out_fmt != VLX_FMT_S24 && out_fmt != VLX_FMT_S32 && out_fmt != VLX_FMT_I64) return VLX_ERR_ARG;
    uint64_t n_in = 0;
    for (uint32_t t = 0; t < n_tracks; t++) n_in += h_tracks[t].n_sizes;
    if (n_in && !h_sizes) return VLX_ERR_ARG;
    if (n_in >= (1ull << 31)) { c->err = "too many chunks in one batch"; return VLX_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    // pinned staging: [n_tracks] DecTrack, then the sizes that are actually walked, back to back
    const size_t trk_bytes = (sizeof(DecTrack) * n_tracks + 15) & ~(size_t)15;
    HIPCHK(c, c->h_chunks.ensure(trk_bytes + sizeof(uint32_t) * n_in + 16));
    DecTrack *trk = (DecTrack *)c->h_chunks.p;
    uint32_t *sz = (uint32_t *)((uint8_t *)c->h_chunks.p + trk_bytes);
    uint64_t n_chunks = 0; const uint32_t *src = h_sizes;
    bool any_lowres = false;                                              // (a 16-bit container never holds a high-res track: |s| <= 32768)
    for (uint32_t t = 0; t < n_tracks; t++) {
        const vlx_track_sizes &T = h_tracks[t];
        any_lowres |= T.high_res == 0 || (T.high_res == VLX_HIGH_RES_FROM_PAYLOAD && out_fmt == VLX_FMT_S16);
        if (T.high_res == VLX_HIGH_RES_FROM_PAYLOAD ? T.hdr_bits == 0 : T.high_res > 1) { c->err = "bad high_res / hdr_bits in the track table"; return VLX_ERR_ARG; }
        const uint64_t m_max = std::min<uint64_t>(T.n_sizes, chunks_for_total(T.total_samples));
        uint64_t m = 0, bits = T.hdr_bits;
        for (; m < m_max && src[m]; m++) { sz[n_chunks + m] = src[m]; bits += 32 + 8ull * src[m]; }   // a zero size ends the walk (:341)
        if (m && (bits > 8ull * payload_bytes || T.base_bit_off > 8ull * payload_bytes - bits)) { c->err = "chunk beyond payload"; return VLX_ERR_ARG; }
        trk[t].first_bit = T.base_bit_off + T.hdr_bits; trk[t].out_base = T.out_base; trk[t].total = T.total_samples;
        trk[t].first_chunk = n_chunks; trk[t].high_res = T.high_res; trk[t].pad = 0;
        n_chunks += m; src += T.n_sizes;
    }
    if (n_chunks_out) *n_chunks_out = (size_t)n_chunks;
    if (!n_chunks) return VLX_OK;
    if (!d_payload || !d_out) return VLX_ERR_ARG;
    if ((uintptr_t)d_payload & 3) { c->err = "d_payload must be 4-byte aligned"; return VLX_ERR_ARG; }
    HIPCHK(c, c->dtracks.ensure(trk_bytes + sizeof(uint32_t) * n_chunks + 16));
    HIPCHK(c, c->dscan.ensure(sizeof(uint64_t) * VLX_SCAN_ROOM(n_chunks)));
    HIPCHK(c, c->dchunks.ensure(sizeof(vlx_chunk) * n_chunks));
    HIPCHK(c, c->wide_count.ensure(sizeof(uint32_t)));
    hipStream_t st = c->stream;
    // no copy-engine transfer: the first kernel reads the pinned staging area itself (and the synchronisation at the end of
    // decode_chunks_impl keeps the host from refilling it before that)
    HIPCHK(c, vlx_launch_chunks_build(st, d_payload, c->h_chunks.p, c->dtracks.p, trk_bytes, (uint64_t *)c->dscan.p,
                                      n_tracks, (vlx_chunk *)c->dchunks.p, n_chunks, (uint32_t *)c->wide_count.p));
    return decode_chunks_impl(c, d_payload, payload_bytes, (size_t)n_chunks, d_out, out_fmt, any_lowres, true);
}

// The decode loop of main.cpp:253-261 + SamplesToBytes (VeloxFormat.h:160-189) for an integer-PCM payload
// in host memory: one upload, one batch, one download of packed PCM (2-4 bytes per sample, not 8).
int vlx_decode_pcm(vlx_ctx *c, const uint8_t *payload, size_t size, size_t total, int bits, uint8_t *pcm, size_t cap, size_t *n_decoded)
{
    if (n_decoded) *n_decoded = 0;
    if (!c || (!payload && size) || (!pcm && total)) return VLX_ERR_ARG;
    if (bits != 16 && bits != 24 && bits != 32) return VLX_ERR_ARG;
    const size_t W = (size_t)bits / 8;
    if (total > cap / W) { c->err = "pcm buffer too small"; return VLX_ERR_CAPACITY; }
    if (!total) return VLX_OK;
    std::vector<vlx_chunk> chs((size_t)chunks_for_total(total) + 1);
    const long n = vlx_walk_chunks(payload, size, total, 0, 0, chs.data(), chs.size());
    if (n < 0) return (int)n;                                             // VLX_ERR_UNSUPPORTED: float payload -> vlx_decoder_*
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->dec_in.ensure(size + 32));
    HIPCHK(c, c->dec_out.ensure(total * W + 32));
    hipStream_t st = c->stream;
    if (size) HIPCHK(c, hipMemcpyAsync(c->dec_in.p, payload, size, hipMemcpyHostToDevice, st));
    // samples the stream does not reach stay zero, as in the caller's zero-initialised vector (main.cpp:254-261)
    HIPCHK(c, hipMemsetAsync(c->dec_out.p, 0, total * W, st));
    size_t done = 0;
    for (long i = 0; i < n; i++) done += chs[(size_t)i].n_out;
    if (n) {
        const int rc = vlx_decode_device(c, (const uint8_t *)c->dec_in.p, size, chs.data(), (size_t)n, c->dec_out.p, bits);
        if (rc != VLX_OK) return rc;
    }
    HIPCHK(c, hipMemcpyAsync(pcm, c->dec_out.p, total * W, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (n_decoded) *n_decoded = done;
    return VLX_OK;
}

} // extern "C"

// ---- StreamingDecoder drop-in ------------------------------------------------------------------
struct vlx_decoder {
    vlx_ctx *ctx;
    const uint8_t *data; size_t size; uint64_t total;
    uint64_t pos_bits;                // next chunk's u32 size prefix
    uint64_t start_bits;              // pos_bits right after the stream header (for seek)
    uint64_t decoded;                 // samples handed to the block buffer so far (walk position)
    uint64_t delivered;               // decoded_count of the reference (samples returned to the caller)
    bool is_float; int float_mode; uint32_t high_res;
    std::vector<uint8_t> exps; size_t exp_idx;
    std::vector<int64_t> block; size_t block_ptr;
    unsigned lookahead;
    bool hit_zero;
    bool header_complete;             // the stream header (incl. exponent runs) lay inside `size` at open
};

extern "C" {

int vlx_decoder_open(vlx_ctx *c, const uint8_t *data, size_t size, size_t total, vlx_decoder **out)
{
    if (!c || !out || (!data && size)) return VLX_ERR_ARG;
    vlx_decoder *d = new (std::nothrow) vlx_decoder();
    if (!d) return VLX_ERR_NOMEM;
    d->ctx = c; d->data = data; d->size = size; d->total = total;
    d->decoded = 0; d->delivered = 0; d->exp_idx = 0; d->block_ptr = 0; d->lookahead = 1; d->hit_zero = false; d->float_mode = 0;
    HostReader r{data, 8ull * size, 0};
    d->is_float = r.read(1) != 0;                                         // VeloxCore.h:324-329
    if (d->is_float) {
        d->float_mode = (int)r.read(2);
        if (d->float_mode == 0) {                                         // DecodeRLE :196-203
            d->exps.reserve(std::min<uint64_t>(total, 128ull * size));      // a (run, value) byte pair yields at most 255 exponents
            while (d->exps.size() < total) {
                if (r.pos >= r.size_bits) break;                          // truncated: the reference would spin; we stop
                int run = (int)r.read(8), val = (int)r.read(8);
                for (int i = 0; i < run; i++) d->exps.push_back((uint8_t)val);
            }
        }
    }
    d->header_complete = r.pos < r.size_bits;                             // a header that ends exactly at `size` may be cut short
    d->high_res = (uint32_t)r.read(1);
    d->pos_bits = r.pos; d->start_bits = r.pos;
    *out = d;
    return VLX_OK;
}

int vlx_decoder_set_lookahead(vlx_decoder *d, unsigned chunks) { if (!d || !chunks) return VLX_ERR_ARG; d->lookahead = chunks; return VLX_OK; }
int vlx_decoder_extend(vlx_decoder *d, size_t new_size) { if (!d || new_size < d->size) return VLX_ERR_ARG; d->size = new_size; return VLX_OK; }
int vlx_decoder_is_float(const vlx_decoder *d) { return d && d->is_float && d->float_mode == 0; }
int vlx_decoder_float_mode(const vlx_decoder *d) { return d ? d->float_mode : 0; }

int vlx_decoder_header_complete(const vlx_decoder *d) { return d && d->header_complete; }

int vlx_decoder_bytes_needed(const vlx_decoder *d, size_t samples, size_t *bytes)
{
    if (!d || !bytes) return VLX_ERR_ARG;
    *bytes = 0;
    const size_t buffered = d->block.size() - d->block_ptr;
    if (samples <= buffered || d->decoded >= d->total) return VLX_OK;     // no refill on the way to `samples`
    uint64_t pos = d->pos_bits, dec = d->decoded, have = buffered;
    while (have < samples && dec < d->total) {                            // one refill = `lookahead` chunks
        for (unsigned i = 0; i < d->lookahead && dec < d->total; i++) {
            const uint64_t prefix_end = (pos + 32 + 7) >> 3;
            *bytes = (size_t)prefix_end;
            if (prefix_end > d->size) return VLX_OK;                      // the size prefix itself has not arrived yet
            HostReader r{d->data, 8ull * d->size, pos};
            const uint32_t sz = (uint32_t)r.read(32);
            if (sz == 0) return VLX_OK;                                   // end marker (VeloxCore.h:341)
            pos += 32 + 8ull * sz;
            *bytes = (size_t)((pos + 7) >> 3);
            const uint64_t remaining = d->total - dec;
            uint64_t frames = std::min<uint64_t>(4096, remaining / 2);
            if (frames == 0) frames = remaining;
            dec += std::min<uint64_t>(2 * frames, remaining);
            have += 2 * frames;
        }
    }
    return VLX_OK;
}

int vlx_decoder_clone(const vlx_decoder *d, vlx_decoder **out)
{
    if (!d || !out) return VLX_ERR_ARG;
    vlx_decoder *n = new (std::nothrow) vlx_decoder(*d);
    if (!n) return VLX_ERR_NOMEM;
    *out = n;
    return VLX_OK;
}
void vlx_decoder_close(vlx_decoder *d) { delete d; }
int vlx_decoder_set_ctx(vlx_decoder *d, vlx_ctx *c) { if (!d || !c) return VLX_ERR_ARG; d->ctx = c; return VLX_OK; }

// One decoder's next refill: up to `lookahead` chunks walked on the host (StreamingDecoder's refill
// step, VeloxCore.h:338-353), not yet decoded.
struct RefillPlan {
    vlx_decoder *d; std::vector<vlx_chunk> chs;
    uint64_t first_byte; size_t nbytes; uint64_t out_total; uint64_t dec, new_pos;
};

// Returns 1 when chunks were planned, 0 when the next chunk size is zero / the data is exhausted.
static int plan_refill(vlx_decoder *d, RefillPlan &p)
{
    p.d = d; p.chs.clear();
    d->block.clear(); d->block_ptr = 0;
    HostReader r{d->data, 8ull * d->size, d->pos_bits};
    uint64_t dec = d->decoded;
    while (p.chs.size() < d->lookahead && dec < d->total) {
        vlx_chunk ch;
        const uint64_t before = r.pos;
        if (!walk_one(r, d->total, dec, 0, 0, d->high_res, &ch)) {
            if (p.chs.empty()) { d->pos_bits = r.pos; return 0; }         // zero chunk size: DecodeNext returns false (:341)
            r.pos = before;                                               // leave it for the next refill to report
            break;
        }
        p.chs.push_back(ch);
    }
    if (p.chs.empty()) return 0;
    p.first_byte = p.chs.front().bit_off >> 3;
    uint64_t last_bit = 0;
    for (auto &ch : p.chs) last_bit = std::max(last_bit, ch.bit_off + ch.n_bits);
    const uint64_t last_byte = std::min<uint64_t>((last_bit + 7) >> 3, d->size);
    p.nbytes = last_byte > p.first_byte ? (size_t)(last_byte - p.first_byte) : 0;
    p.out_total = 0;
    // the reference's block buffer always holds 2*frames samples per chunk (VeloxCore.h:357-377)
    for (auto &ch : p.chs) { ch.bit_off -= 8 * p.first_byte; ch.out_off = p.out_total; ch.n_out = 2 * ch.frames; p.out_total += 2ull * ch.frames; }
    p.dec = dec; p.new_pos = r.pos;
    return 1;
}

// Decode every planned refill in ONE GPU batch and hand the samples to the decoders' block buffers.
static int run_refills(vlx_ctx *c, std::vector<RefillPlan> &plans)
{
    if (plans.empty()) return VLX_OK;
    size_t in_total = 0; uint64_t out_total = 0; size_t n_chunks = 0;
    std::vector<size_t> in_off(plans.size()); std::vector<uint64_t> out_off(plans.size());
    for (size_t i = 0; i < plans.size(); i++) {
        in_off[i] = in_total; out_off[i] = out_total;
        in_total += (plans[i].nbytes + 16 + 15) & ~(size_t)15;
        out_total += plans[i].out_total; n_chunks += plans[i].chs.size();
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->dec_in.ensure(in_total + 16));
    HIPCHK(c, c->dec_out.ensure(sizeof(int64_t) * (out_total + 2)));
    std::vector<vlx_chunk> dev; dev.reserve(n_chunks);
    if (plans.size() == 1) {                                              // straight from the caller's buffer
        const RefillPlan &p = plans[0];
        if (p.nbytes) HIPCHK(c, hipMemcpyAsync(c->dec_in.p, p.d->data + p.first_byte, p.nbytes, hipMemcpyHostToDevice, c->stream));
        dev = p.chs;
    } else {                                                              // one upload for all streams
        c->h_in.assign(in_total, 0);
        for (size_t i = 0; i < plans.size(); i++) {
            const RefillPlan &p = plans[i];
            if (p.nbytes) memcpy(c->h_in.data() + in_off[i], p.d->data + p.first_byte, p.nbytes);
            for (vlx_chunk ch : p.chs) { ch.bit_off += 8ull * in_off[i]; ch.out_off += out_off[i]; dev.push_back(ch); }
        }
        HIPCHK(c, hipMemcpyAsync(c->dec_in.p, c->h_in.data(), in_total, hipMemcpyHostToDevice, c->stream));
    }
    int rc = vlx_decode_device(c, (const uint8_t *)c->dec_in.p, in_total, dev.data(), dev.size(), c->dec_out.p, VLX_FMT_I64);
    if (rc != VLX_OK) return rc;
    if (plans.size() <= 4) {                                              // few streams: straight into their block buffers
        for (size_t i = 0; i < plans.size(); i++) {
            RefillPlan &p = plans[i];
            p.d->block.resize(p.out_total);
            if (p.out_total) HIPCHK(c, hipMemcpy(p.d->block.data(), (const int64_t *)c->dec_out.p + out_off[i], sizeof(int64_t) * p.out_total, hipMemcpyDeviceToHost));
        }
    } else {                                                              // many streams: one download, then split
        c->h_out.resize(out_total);
        if (out_total) HIPCHK(c, hipMemcpy(c->h_out.data(), c->dec_out.p, sizeof(int64_t) * out_total, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < plans.size(); i++)
            plans[i].d->block.assign(c->h_out.begin() + out_off[i], c->h_out.begin() + out_off[i] + plans[i].out_total);
    }
    for (RefillPlan &p : plans) { p.d->decoded = p.dec; p.d->pos_bits = p.new_pos; }
    return VLX_OK;
}

// Refill the block buffer: up to `lookahead` chunks in one GPU batch.  Returns 1 when samples were
// added, 0 when the next chunk size is zero / data exhausted, <0 on error.
static int decoder_refill(vlx_decoder *d)
{
    std::vector<RefillPlan> plans(1);
    if (!plan_refill(d, plans[0])) return 0;
    const int rc = run_refills(d->ctx, plans);
    return rc == VLX_OK ? 1 : rc;
}

int vlx_decoder_refill_many(vlx_decoder **ds, size_t n)
{
    if (!ds && n) return VLX_ERR_ARG;
    std::vector<RefillPlan> plans; plans.reserve(n);
    vlx_ctx *c = nullptr;
    for (size_t i = 0; i < n; i++) {
        vlx_decoder *d = ds[i];
        if (!d) return VLX_ERR_ARG;
        if (!c) c = d->ctx; else if (c != d->ctx) return VLX_ERR_ARG;      // one batch = one GPU
        if (d->block_ptr < d->block.size() || d->delivered >= d->total) continue;   // still has samples / finished
        RefillPlan p;
        if (plan_refill(d, p)) plans.push_back(std::move(p));
    }
    return c ? run_refills(c, plans) : VLX_OK;
}

int vlx_decoder_seek(vlx_decoder *d, size_t sample)
{
    if (!d) return VLX_ERR_ARG;
    if (sample > d->total) sample = d->total;
    d->block.clear(); d->block_ptr = 0;
    d->pos_bits = d->start_bits; d->decoded = 0;
    HostReader r{d->data, 8ull * d->size, d->pos_bits};
    uint64_t dec = 0;
    while (dec < d->total) {                                              // hop over whole chunks: sizes only, nothing is decoded
        const uint64_t before = r.pos, dec_before = dec;
        vlx_chunk ch;
        if (r.pos >= r.size_bits || !walk_one(r, d->total, dec, 0, 0, d->high_res, &ch)) { r.pos = before; break; }
        if (dec > sample) { r.pos = before; dec = dec_before; break; }    // the target lies inside this chunk
    }
    d->pos_bits = r.pos; d->decoded = dec;
    if (dec > sample) return VLX_ERR_ARG;                                 // cannot happen
    if (sample > dec && dec < d->total) {
        const int rc = decoder_refill(d);
        if (rc < 0) return rc;
        if (rc > 0) d->block_ptr = std::min<size_t>(d->block.size(), (size_t)(sample - dec));
        else sample = dec;                                                // stream ends before the target
    } else if (sample > dec) sample = dec;
    d->delivered = sample;
    d->exp_idx = std::min<size_t>(d->exps.size(), sample);
    return VLX_OK;
}

int vlx_decoder_next(vlx_decoder *d, int64_t *val, uint8_t *exp)
{
    if (!d || !val) return VLX_ERR_ARG;
    if (d->delivered >= d->total) return 0;                               // VeloxCore.h:336
    if (d->block_ptr >= d->block.size()) {
        int rc = decoder_refill(d);
        if (rc <= 0) return rc;
    }
    *val = d->block[d->block_ptr++];
    uint8_t e = 0;
    if (d->is_float && d->float_mode == 0 && d->exp_idx < d->exps.size()) e = d->exps[d->exp_idx++];   // :383
    if (exp) *exp = e;
    d->delivered++;
    return 1;
}

long vlx_decoder_read(vlx_decoder *d, int64_t *vals, uint8_t *exps, size_t cap)
{
    if (!d || (!vals && cap)) return VLX_ERR_ARG;
    size_t n = 0;
    while (n < cap && d->delivered < d->total) {                          // the same steps as `cap` DecodeNext calls, in runs
        if (d->block_ptr >= d->block.size()) {
            const int rc = decoder_refill(d);
            if (rc < 0) return rc;
            if (rc == 0) break;
        }
        const size_t k = std::min({cap - n, d->block.size() - d->block_ptr, (size_t)(d->total - d->delivered)});
        memcpy(vals + n, d->block.data() + d->block_ptr, k * sizeof(int64_t));
        if (exps) {
            size_t e = 0;
            if (d->is_float && d->float_mode == 0) {                      // VeloxCore.h:383
                e = std::min(k, d->exps.size() - std::min(d->exps.size(), d->exp_idx));
                memcpy(exps + n, d->exps.data() + d->exp_idx, e);
            }
            memset(exps + n + e, 0, k - e);
        }
        if (d->is_float && d->float_mode == 0) d->exp_idx = std::min(d->exps.size(), d->exp_idx + k);
        d->block_ptr += k; d->delivered += k; n += k;
    }
    return (long)n;
}

} // extern "C"

