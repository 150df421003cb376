// TEST INFRASTRUCTURE ONLY — never linked into or called from the product path.
//
// A thin extern "C" harness around the *real* reference codec.  It is compiled by
// oracle/Makefile directly against the headers where they lie under /root/reference
// (-I/root/reference); nothing from the reference is copied into this repository.
// The resulting oracle/_ref/libvelox_ref.so is git-ignored, travels to the GPU box as a
// built artefact, and is used (a) to pin oracle/velox_oracle.c, (b) to generate the
// golden fixtures under tests/golden/, (c) as bench.py's `cpu_baseline` (kind "reference").
//
// Entry points mirrored:
//   VeloxCodec::Encoder::ProcessBlock      VeloxCore.h:210
//   VeloxCodec::StreamingDecoder           VeloxCore.h:309-389
//   VeloxCodec::ComputeLPC (private)       VeloxCore.h:44
//   VeloxCodec::TryCompressChannel (priv.) VeloxCore.h:69
//   VeloxCodec::DecodeChannelWorker (priv.)VeloxCore.h:130
#include <vector>
#include <queue>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <future>
#include <numeric>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdint>
#include <algorithm>
#include <limits>
#include <sched.h>

// private statics of VeloxCodec are reached with -fno-access-control (see Makefile)
#include "VeloxCore.h"

extern "C" {

// Limit the static ThreadPool (VeloxCore.h:208 sizes it from hardware_concurrency(),
// which glibc derives from the affinity mask).  Must run before the first encode.
int ref_limit_threads(int n)
{
    cpu_set_t cur, set;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return -1;
    CPU_ZERO(&set);
    int kept = 0;
    for (int c = 0; c < CPU_SETSIZE && kept < n; ++c)
        if (CPU_ISSET(c, &cur)) { CPU_SET(c, &set); ++kept; }
    if (kept == 0) return -1;
    if (sched_setaffinity(0, sizeof(set), &set) != 0) return -1;
    return kept;
}

int ref_pool_threads(void) { return (int)std::thread::hardware_concurrency(); }

// ProcessBlock.  `samples` may be rewritten (pseudo-float demotion), as in the reference.
// Returns a malloc'd payload in *out (caller frees with ref_free).
size_t ref_encode(int64_t* samples, size_t n, int is_float,
                  const uint8_t* exps, size_t n_exps, const uint8_t* raw_bytes, uint8_t** out)
{
    std::vector<velox_sample_t> s(samples, samples + n);
    std::vector<uint8_t> e;
    if (exps && n_exps) e.assign(exps, exps + n_exps);
    VeloxCodec::Encoder enc;
    std::vector<uint8_t> r = enc.ProcessBlock(s, is_float != 0, e, raw_bytes);
    if (is_float) std::copy(s.begin(), s.end(), samples);
    *out = (uint8_t*)malloc(r.size() ? r.size() : 1);
    memcpy(*out, r.data(), r.size());
    return r.size();
}

void ref_free(void* p) { free(p); }

// StreamingDecoder pull loop (main.cpp:253-261).  Returns samples produced.
size_t ref_decode(const uint8_t* data, size_t size, size_t total,
                  int64_t* out, uint8_t* out_exp, int* is_float, int* float_mode)
{
    VeloxCodec::StreamingDecoder dec(data, size, total);
    size_t i = 0;
    for (; i < total; ++i) {
        velox_sample_t v; uint8_t e;
        if (!dec.DecodeNext(v, e)) break;
        out[i] = v;
        if (out_exp) out_exp[i] = e;
    }
    if (is_float) *is_float = dec.IsFloat() ? 1 : 0;
    if (float_mode) *float_mode = dec.is_float ? dec.GetFloatMode() : 0;
    return i;
}

void ref_compute_lpc(const int64_t* x, size_t n, int order, int* coeffs, int* shift)
{
    std::vector<velox_sample_t> d(x, x + n);
    std::vector<int> c; int sh = 0;
    VeloxCodec::ComputeLPC(d, order, c, sh);
    c.resize(order, 0);
    for (int i = 0; i < order; ++i) coeffs[i] = c[i];
    *shift = sh;
}

// One stream through TryCompressChannel; returns the byte count after Flush().
size_t ref_try_compress_channel(const int64_t* x, size_t n, int high_res, uint8_t* out, size_t cap)
{
    std::vector<velox_sample_t> d(x, x + n);
    BitStreamWriter bs;
    VeloxCodec::TryCompressChannel(d, bs, high_res != 0);
    bs.Flush();
    const std::vector<uint8_t>& b = bs.GetData();
    size_t m = b.size() < cap ? b.size() : cap;
    memcpy(out, b.data(), m);
    return b.size();
}

void ref_decode_channel(const uint8_t* data, size_t size, size_t count, int high_res, int64_t* out)
{
    BitStreamReader bs(data, size);
    std::vector<velox_sample_t> o;
    VeloxCodec::DecodeChannelWorker(bs, count, o, high_res != 0);
    std::copy(o.begin(), o.end(), out);
}

// FormatHandler helpers used by the float path (VeloxFormat.h:14-128).
int ref_detect_pseudo_float(const uint8_t* raw, size_t count)
{ return FormatHandler::DetectPseudoFloat(raw, count); }

void ref_split_float32(const uint8_t* raw, size_t count, int64_t* mant, uint8_t* exps)
{
    std::vector<velox_sample_t> m; std::vector<uint8_t> e;
    FormatHandler::SplitFloat32(raw, count, m, e);
    std::copy(m.begin(), m.end(), mant); std::copy(e.begin(), e.end(), exps);
}

void ref_bytes_to_samples(const uint8_t* bytes, size_t count, int bits, int64_t* out)
{
    std::vector<velox_sample_t> o;
    FormatHandler::BytesToSamples(bytes, count, bits, o);
    std::copy(o.begin(), o.end(), out);
}

void ref_samples_to_bytes(const int64_t* in, size_t count, int bits, uint8_t* out)
{
    std::vector<velox_sample_t> i(in, in + count);
    std::vector<uint8_t> b;
    FormatHandler::SamplesToBytes(i, bits, b);
    memcpy(out, b.data(), b.size());
}

} // extern "C"
