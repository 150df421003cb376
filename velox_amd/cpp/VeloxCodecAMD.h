// VeloxCodecAMD.h — the reference's two codec entry points, backed by the MI355X library.
//
// Drop-in for the part of VeloxCore.h that applications use (reference VeloxCore.h:205-389):
//
//     VeloxCodec::Encoder enc;
//     std::vector<uint8_t> payload = enc.ProcessBlock(samples, is_float, exps, raw_bytes);   // main.cpp:136-137
//     VeloxCodec::StreamingDecoder dec(data, size, total);                                    // main.cpp:253
//     while (dec.DecodeNext(val, exp)) ...                                                    // main.cpp:257-261
//
// Same names, same argument meaning, same by-value / borrowed-pointer ownership, same "no
// exceptions, no error codes" surface: a GPU failure makes ProcessBlock return an empty vector and
// DecodeNext return false, and the reason is kept in VeloxCodec::LastError().  The decoder is
// copy- and move-assignable like the reference class (VeloxStreamClient.cpp:396,
// VeloxQtPlayerEngine.cpp:634).  Everything below the calls is libvelox_amd.so (include/velox_amd.h).
#ifndef VELOX_CODEC_AMD_H
#define VELOX_CODEC_AMD_H

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#if defined(__has_include)
#  if __has_include("VeloxArch.h")
#    include "VeloxArch.h"          // velox_sample_t, VeloxHeader: the unchanged container surface
#    define VELOX_AMD_HAVE_ARCH 1
#  endif
#endif
#ifndef VELOX_AMD_HAVE_ARCH
typedef int64_t velox_sample_t;    // VeloxArch.h:10
#endif

#include "velox_amd.h"

class VeloxCodec {
    // One context per host thread: a vlx_ctx is confined to one thread at a time, exactly like a
    // reference StreamingDecoder instance.  Replaces Encoder::GetPool() (VeloxCore.h:208).
    struct Ctx {
        vlx_ctx *h = nullptr; int rc = VLX_OK;
        Ctx() { rc = vlx_ctx_create(Device(), &h); if (rc != VLX_OK) Error() = std::string("vlx_ctx_create: ") + vlx_strerror(rc); }
        ~Ctx() { if (h) vlx_ctx_destroy(h); }
    };
    static vlx_ctx *Context() { static thread_local Ctx c; return c.h; }
    static std::string &Error() { static thread_local std::string e; return e; }
    static int &Device() { static int d = 0; return d; }
    static unsigned &Lookahead() { static unsigned n = 1; return n; }

    // Several GPUs behind the one caller thread the reference has (main.cpp:136-137): a process-wide vlx_pool, made on
    // first use when more than one device is named — by SetDevices, or by VELOX_AMD_DEVICES in the environment
    // ("all", or a list such as "0,1,2,3"; a device may repeat: several shards in flight on one GPU).
    static std::vector<int> DevicesFromEnv()
    {
        std::vector<int> d;
        const char *e = std::getenv("VELOX_AMD_DEVICES");
        if (!e || !*e) return d;
        if (std::string(e) == "all") { int n = 0; vlx_device_count(&n); for (int i = 0; i < n; i++) d.push_back(i); return d; }
        for (const char *p = e; *p;) { char *q = nullptr; const long v = std::strtol(p, &q, 10); if (q == p) break; d.push_back((int)v); p = (*q == ',') ? q + 1 : q; }
        return d;
    }
    struct PoolBox {
        std::mutex m; std::vector<int> devices = DevicesFromEnv(); vlx_pool *h = nullptr; bool tried = false;
        ~PoolBox() { if (h) vlx_pool_destroy(h); }
    };
    static PoolBox &Box() { static PoolBox b; return b; }
    // the pool with Box().m held, or null (one device: the calling thread's own context does the work)
    static vlx_pool *PoolLocked(PoolBox &b)
    {
        if (b.devices.size() < 2) return nullptr;
        if (!b.tried) {
            b.tried = true;
            const int rc = vlx_pool_create(b.devices.data(), (int)b.devices.size(), &b.h);
            if (rc != VLX_OK) { b.h = nullptr; Error() = std::string("vlx_pool_create: ") + vlx_strerror(rc); }
        }
        return b.h;
    }

public:
    // Which GPU new contexts use (call before the first encode/decode on a thread).
    static void SetDevice(int device) { Device() = device; }
    // ProcessBlock / EncodePCM / DecodePCM over several GPUs: the track is cut into contiguous runs of sub-blocks, one
    // per entry (VeloxCore.h:255-303 fanned out over GPUs instead of pool threads), the payload is the same bytes.
    // Call before the first encode; an empty or one-entry list goes back to one GPU (that entry, else SetDevice's).
    static void SetDevices(const std::vector<int> &devices)
    {
        PoolBox &b = Box();
        std::lock_guard<std::mutex> lk(b.m);
        if (b.h) { vlx_pool_destroy(b.h); b.h = nullptr; }
        b.devices = devices; b.tried = false;
        if (devices.size() == 1) Device() = devices[0];
    }
    // Chunks decoded per GPU launch by StreamingDecoder: 1 = the reference's pacing (streaming
    // client), larger = whole-file decode (CLI, Win32 player cache fill).
    static void SetDecodeLookahead(unsigned chunks) { Lookahead() = chunks ? chunks : 1; }
    static const std::string &LastError() { return Error(); }

    class Encoder {
    public:
        // VeloxCore.h:210-306.  `samples` is rewritten in pseudo-float mode like the reference's vector.
        std::vector<uint8_t> ProcessBlock(std::vector<velox_sample_t> &samples, bool is_float,
                                          const std::vector<uint8_t> &exps, const uint8_t *raw_bytes)
        {
            std::vector<uint8_t> out;
            uint8_t *p = nullptr; size_t n = 0;
            static_assert(sizeof(velox_sample_t) == sizeof(int64_t), "velox_sample_t must be int64");
            {
                PoolBox &b = Box();
                std::lock_guard<std::mutex> lk(b.m);
                if (vlx_pool *pool = PoolLocked(b)) {
                    const int rc = vlx_pool_encode_block(pool, reinterpret_cast<int64_t *>(samples.data()), samples.size(), is_float ? 1 : 0,
                                                         exps.data(), exps.size(), raw_bytes, &p, &n);
                    if (rc != VLX_OK) { Error() = std::string("vlx_pool_encode_block: ") + vlx_strerror(rc) + ": " + vlx_pool_last_error(pool); return out; }
                    out.assign(p, p + n);
                    vlx_free(p);
                    return out;
                }
            }
            vlx_ctx *c = Context();
            if (!c) return out;
            const int rc = vlx_encode_block(c, reinterpret_cast<int64_t *>(samples.data()), samples.size(), is_float ? 1 : 0,
                                            exps.data(), exps.size(), raw_bytes, &p, &n);
            if (rc != VLX_OK) { Error() = std::string("vlx_encode_block: ") + vlx_strerror(rc) + ": " + vlx_last_error(c); return out; }
            out.assign(p, p + n);
            vlx_free(p);
            return out;
        }
    };

    // Packed little-endian PCM straight through, for callers that hold file bytes rather than a sample
    // vector (no reference counterpart as functions; they stand for BytesToSamples + ProcessBlock,
    // main.cpp:120-137, and for the DecodeNext loop + SamplesToBytes, main.cpp:253-268, of integer PCM).
    static std::vector<uint8_t> EncodePCM(const uint8_t *pcm, size_t n_samples, int bits)
    {
        std::vector<uint8_t> out;
        uint8_t *p = nullptr; size_t n = 0;
        {
            PoolBox &b = Box();
            std::lock_guard<std::mutex> lk(b.m);
            if (vlx_pool *pool = PoolLocked(b)) {
                const int rc = vlx_pool_encode_pcm(pool, pcm, n_samples, bits, &p, &n);
                if (rc != VLX_OK) { Error() = std::string("vlx_pool_encode_pcm: ") + vlx_strerror(rc) + ": " + vlx_pool_last_error(pool); return out; }
                out.assign(p, p + n);
                vlx_free(p);
                return out;
            }
        }
        vlx_ctx *c = Context();
        if (!c) return out;
        const int rc = vlx_encode_pcm(c, pcm, n_samples, bits, &p, &n);
        if (rc != VLX_OK) { Error() = std::string("vlx_encode_pcm: ") + vlx_strerror(rc) + ": " + vlx_last_error(c); return out; }
        out.assign(p, p + n);
        vlx_free(p);
        return out;
    }
    // false for float payloads (use StreamingDecoder) and on errors; `reached` = samples the stream held
    static bool DecodePCM(const uint8_t *payload, size_t size, size_t total, int bits, std::vector<uint8_t> &pcm, size_t *reached = nullptr)
    {
        if (bits != 16 && bits != 24 && bits != 32) return false;
        pcm.assign(total * (size_t)(bits / 8), 0);
        {
            PoolBox &b = Box();
            std::lock_guard<std::mutex> lk(b.m);
            if (vlx_pool *pool = PoolLocked(b)) {
                const int rc = vlx_pool_decode_pcm(pool, payload, size, total, bits, pcm.data(), pcm.size(), reached);
                if (rc != VLX_OK) { Error() = std::string("vlx_pool_decode_pcm: ") + vlx_strerror(rc) + ": " + vlx_pool_last_error(pool); return false; }
                return true;
            }
        }
        vlx_ctx *c = Context();
        if (!c) return false;
        const int rc = vlx_decode_pcm(c, payload, size, total, bits, pcm.data(), pcm.size(), reached);
        if (rc != VLX_OK) { Error() = std::string("vlx_decode_pcm: ") + vlx_strerror(rc) + ": " + vlx_last_error(c); return false; }
        return true;
    }

    // Threading: like the reference class, a StreamingDecoder may be constructed on one thread and used on
    // another (one thread at a time).  Contexts are per thread here, so every call that can reach the GPU
    // first points the decoder at the CALLING thread's context (vlx_decoder_set_ctx: a pointer store); a
    // decoder therefore never touches the context of a thread that may have exited.
    class StreamingDecoder {
        vlx_decoder *d = nullptr;
        bool bind() const { vlx_ctx *c = Context(); return d && c && vlx_decoder_set_ctx(d, c) == VLX_OK; }
        void open(const uint8_t *data, size_t size, size_t total)
        {
            vlx_ctx *c = Context();
            if (!c) return;
            const int rc = vlx_decoder_open(c, data, size, total, &d);
            if (rc != VLX_OK) { d = nullptr; Error() = std::string("vlx_decoder_open: ") + vlx_strerror(rc); return; }
            vlx_decoder_set_lookahead(d, Lookahead());
        }
    public:
        StreamingDecoder(const uint8_t *data, size_t size, size_t total) { open(data, size, total); }   // VeloxCore.h:322-330
        StreamingDecoder(const StreamingDecoder &o) { if (o.d) vlx_decoder_clone(o.d, &d); }
        StreamingDecoder(StreamingDecoder &&o) noexcept : d(o.d) { o.d = nullptr; }
        StreamingDecoder &operator=(const StreamingDecoder &o)
        {
            if (this != &o) { vlx_decoder *n = nullptr; if (o.d) vlx_decoder_clone(o.d, &n); if (d) vlx_decoder_close(d); d = n; }
            return *this;
        }
        StreamingDecoder &operator=(StreamingDecoder &&o) noexcept { std::swap(d, o.d); return *this; }
        ~StreamingDecoder() { if (d) vlx_decoder_close(d); }

        bool IsFloat() const { return d && vlx_decoder_is_float(d); }          // :332
        int GetFloatMode() const { return d ? vlx_decoder_float_mode(d) : 0; } // :333

        bool DecodeNext(velox_sample_t &out_val, uint8_t &out_exp)             // :335-388
        {
            if (!bind()) return false;
            int64_t v = 0; uint8_t e = 0;
            const int rc = vlx_decoder_next(d, &v, &e);
            if (rc < 0) { Error() = std::string("vlx_decoder_next: ") + vlx_strerror(rc); return false; }
            if (rc == 0) return false;
            out_val = (velox_sample_t)v; out_exp = e;
            return true;
        }
        // Extension: what every caller's loop does, in one call (main.cpp:257-261).
        size_t DecodeMany(velox_sample_t *vals, uint8_t *exps, size_t cap)
        {
            if (!bind()) return 0;
            const long n = vlx_decoder_read(d, reinterpret_cast<int64_t *>(vals), exps, cap);
            if (n < 0) { Error() = std::string("vlx_decoder_read: ") + vlx_strerror((int)n); return 0; }
            return (size_t)n;
        }
        // Extension for VeloxStreamClient.cpp:432: more bytes of the borrowed buffer have arrived.
        void Extend(size_t new_size) { if (d) vlx_decoder_extend(d, new_size); }
        // False when the buffer was too short at construction to hold the whole stream header (a true-float
        // track's exponent runs can be long): construct again once more bytes have arrived.
        bool HeaderComplete() const { return d && vlx_decoder_header_complete(d); }
        // Bytes of the buffer that must be valid before the next `samples` DecodeNext calls (0: none needed).
        size_t BytesNeeded(size_t samples = 1) const { size_t n = 0; if (d) vlx_decoder_bytes_needed(d, samples, &n); return n; }
        // Extension: seek without decoding what lies before (the players restart and fast-forward,
        // VeloxQtPlayerEngine.cpp:629-649); same state afterwards.
        bool Seek(size_t sample) { return bind() && vlx_decoder_seek(d, sample) == VLX_OK; }
        // Extension for processes that drive many streams: one GPU batch refills every decoder that ran dry.
        static bool RefillMany(StreamingDecoder *const *decs, size_t n)
        {
            std::vector<vlx_decoder *> hs;
            for (size_t i = 0; i < n; i++) if (decs[i] && decs[i]->bind()) hs.push_back(decs[i]->d);
            return vlx_decoder_refill_many(hs.data(), hs.size()) == VLX_OK;
        }
    };
};

#endif // VELOX_CODEC_AMD_H
