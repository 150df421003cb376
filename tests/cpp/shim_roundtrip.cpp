// C++ caller of the drop-in shim, written the way main.cpp / the players call the reference
// (main.cpp:136-137, :253-261; VeloxStreamClient.cpp:396 copy-assign).  Checks every byte and
// sample against the CPU oracle (test infrastructure, linked only into this test binary).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "VeloxCodecAMD.h"

extern "C" {
size_t vlxo_encode(int64_t *samples, size_t total, int is_float, const uint8_t *exps, size_t n_exps,
                   const uint8_t *raw, uint8_t **out);
size_t vlxo_decode(const uint8_t *data, size_t size, size_t total, int64_t *out, uint8_t *out_exp, int *is_float, int *float_mode);
void vlxo_free(void *p);
}

static int check(const char *name, std::vector<velox_sample_t> s)
{
    std::vector<velox_sample_t> for_ref = s, for_gpu = s;
    uint8_t *rp = nullptr;
    const size_t rn = vlxo_encode(for_ref.data(), for_ref.size(), 0, nullptr, 0, nullptr, &rp);
    VeloxCodec::Encoder enc;
    std::vector<uint8_t> exps;
    std::vector<uint8_t> payload = enc.ProcessBlock(for_gpu, false, exps, nullptr);
    if (payload.size() != rn || memcmp(payload.data(), rp, rn) != 0) {
        fprintf(stderr, "%s: ENCODE MISMATCH (%zu vs %zu bytes) %s\n", name, payload.size(), rn, VeloxCodec::LastError().c_str());
        return 1;
    }
    std::vector<int64_t> want(s.size() + 1); std::vector<uint8_t> wexp(s.size() + 1); int isf = 0, fm = 0;
    const size_t wn = vlxo_decode(rp, rn, s.size(), want.data(), wexp.data(), &isf, &fm);
    vlxo_free(rp);
    VeloxCodec::StreamingDecoder dec(payload.data(), payload.size(), s.size());
    velox_sample_t v; uint8_t e; size_t i = 0;
    const size_t half = s.size() / 2;
    for (; i < half; i++) if (!dec.DecodeNext(v, e) || v != want[i]) { fprintf(stderr, "%s: DECODE MISMATCH at %zu\n", name, i); return 1; }
    VeloxCodec::StreamingDecoder copy = dec;                     // seek-style copy-assignment mid-stream
    dec = VeloxCodec::StreamingDecoder(payload.data(), payload.size(), s.size());
    for (size_t k = 0; k < half; k++) if (!dec.DecodeNext(v, e)) { fprintf(stderr, "%s: fast-forward failed\n", name); return 1; }
    for (; i < wn; i++) {
        velox_sample_t v2; uint8_t e2;
        if (!dec.DecodeNext(v, e) || !copy.DecodeNext(v2, e2) || v != want[i] || v2 != want[i]) { fprintf(stderr, "%s: DECODE MISMATCH at %zu\n", name, i); return 1; }
    }
    if (dec.DecodeNext(v, e) || dec.IsFloat()) { fprintf(stderr, "%s: decoder did not end\n", name); return 1; }
    // packed PCM straight through: same payload from the packed bytes, same bytes back
    bool fits24 = true;
    for (velox_sample_t x : s) if (x < -8388608 || x > 8388607) fits24 = false;
    if (fits24) {
        std::vector<uint8_t> pcm(s.size() * 3);
        for (size_t k = 0; k < s.size(); k++) for (int j = 0; j < 3; j++) pcm[3 * k + j] = (uint8_t)((uint64_t)s[k] >> (8 * j));
        const std::vector<uint8_t> p2 = VeloxCodec::EncodePCM(pcm.data(), s.size(), 24);
        if (p2 != payload) { fprintf(stderr, "%s: EncodePCM differs from ProcessBlock\n", name); return 1; }
        std::vector<uint8_t> back; size_t reached = 0;
        if (!VeloxCodec::DecodePCM(payload.data(), payload.size(), s.size(), 24, back, &reached) || reached != wn) {
            fprintf(stderr, "%s: DecodePCM failed (%s)\n", name, VeloxCodec::LastError().c_str()); return 1;
        }
        for (size_t k = 0; k < wn; k++) for (int j = 0; j < 3; j++)
            if (back[3 * k + j] != (uint8_t)((uint64_t)want[k] >> (8 * j))) { fprintf(stderr, "%s: DecodePCM MISMATCH at %zu\n", name, k); return 1; }
    }
    printf("%s: ok (%zu samples -> %zu bytes)\n", name, s.size(), payload.size());
    return 0;
}

int main()
{
    int bad = 0;
    std::vector<velox_sample_t> a(2 * 30000), b(2 * 9000), z;
    unsigned rng = 12345;
    for (size_t i = 0; i < a.size() / 2; i++) {
        rng = rng * 1664525u + 1013904223u;
        const double t = (double)i / 96000.0;
        a[2 * i] = (velox_sample_t)std::llround(2.0e6 * std::sin(2 * M_PI * 440 * t)) + (int)(rng >> 20) - 2048;
        a[2 * i + 1] = (velox_sample_t)std::llround(1.5e6 * std::sin(2 * M_PI * 554 * t + 0.3)) + (int)((rng >> 8) & 4095) - 2048;
    }
    for (size_t i = 0; i < b.size(); i++) { rng = rng * 1664525u + 1013904223u; b[i] = (int16_t)(rng >> 16); }
    VeloxCodec::SetDecodeLookahead(3);
    bad += check("s24_tone_noise", a);
    bad += check("s16_white_noise", b);
    bad += check("empty", z);
    // the same calls with a pool of contexts behind them (VeloxCodec::SetDevices: ProcessBlock / EncodePCM / DecodePCM cut
    // into runs of sub-blocks, one per entry; three shards on the one GPU of a test box), and back to one GPU
    VeloxCodec::SetDevices({0, 0, 0});
    bad += check("s24_tone_noise, three shards", a);
    bad += check("s16_white_noise, three shards", b);
    bad += check("empty, three shards", z);
    VeloxCodec::SetDevices({0});
    bad += check("s24_tone_noise, one device again", a);
    return bad ? 1 : 0;
}
