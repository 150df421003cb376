// Decode kernels of the Velox frame-codec hot path, hand-written HIP for gfx950 (wave64).
//
//   k_decode_streams   the fast path, one LANE per STREAM, in two wave shapes the launcher mixes by cost.
//                      Stream wave (32 chunks): a half-wave pre-phase parses stream L's entropy codes
//                      to learn where stream R starts (VeloxCore.h:355-356), then both streams run Rice
//                      decode + LMS cascade + quantised-LPC synthesis (:130-162; VeloxEntropy.h:121-138).
//                      Chunk wave (64 chunks): lane = chunk, stream L in full, then stream R.  32-bit
//                      arithmetic, compressed bits staged through a per-lane LDS ring, values to a
//                      stream-major scratch.
//   k_decode_serial    the careful path, one lane per CHUNK, only for chunks the fast path flagged:
//                      int64 arithmetic, every read clamped to the chunk (bits past its end are 0,
//                      VeloxEntropy.h:61-62), streams L then R in sequence.
//   k_decode_finalize  streaming: LSB-shift restore and high-res low bytes (VeloxCore.h:164-171), raw
//                      40-bit blocks (:179-184), inverse M/S (:357-364), int64 or packed 16/24/32-bit
//                      PCM out (FormatHandler::SamplesToBytes, VeloxFormat.h:160-189), 16-B accesses.
#include <atomic>
#include <type_traits>
#include "vlx_device.h"


// Aligned dword of the payload with everything at or beyond `end_bit` (the chunk's end: the
// reference copies the chunk into its own buffer, VeloxCore.h:343-345, so reads past it are 0,
// VeloxEntropy.h:61-62) forced to zero.
__device__ __forceinline__ uint32_t vlx_load_clamped(const uint32_t *base, uint64_t d, uint64_t end_bit)
{
    const uint64_t bp = d << 5;
    if (bp >= end_bit) return 0u;
    uint32_t w = base[d];
    const uint64_t keep = end_bit - bp;
    if (keep < 32) w &= (1u << keep) - 1u;
    return w;
}

// `cnt` (<= 40) bits at absolute bit position p, LSB first.
__device__ __forceinline__ uint64_t vlx_read_abs(const uint32_t *base, uint64_t p, int cnt, uint64_t end_bit)
{
    const uint64_t d = p >> 5; const int sh = (int)(p & 31);
    uint64_t lo = vlx_load_clamped(base, d, end_bit) | ((uint64_t)vlx_load_clamped(base, d + 1, end_bit) << 32);
    uint64_t v = lo >> sh;
    if (sh + cnt > 64) v |= (uint64_t)vlx_load_clamped(base, d + 2, end_bit) << (64 - sh);
    return cnt >= 64 ? v : (v & ((1ull << cnt) - 1ull));
}

// Per-lane LSB-first bit reader (BitStreamReader, VeloxEntropy.h:46-89) over aligned dword loads.
struct LaneReader {
    const uint32_t *base; uint64_t end_bit; uint64_t next_dw; uint64_t buf; int nb;
    __device__ __forceinline__ void init(const uint32_t *b, uint64_t abs_bit, uint64_t end)
    {
        base = b; end_bit = end; next_dw = abs_bit >> 5;
        const int sh = (int)(abs_bit & 31);
        buf = (uint64_t)(vlx_load_clamped(base, next_dw, end_bit) >> sh); nb = 32 - sh; next_dw++;
    }
    __device__ __forceinline__ uint64_t position() const { return (next_dw << 5) - (uint64_t)nb; }
    __device__ __forceinline__ void refill()
    {
        if (nb <= 32) { buf |= (uint64_t)vlx_load_clamped(base, next_dw, end_bit) << nb; nb += 32; next_dw++; }
    }
    __device__ __forceinline__ void consume(int n) { buf = n >= 64 ? 0ull : (buf >> n); nb -= n; }
    __device__ __forceinline__ uint32_t bits(int k)                      // k in [0,32]
    {
        refill();
        uint32_t v = k >= 32 ? (uint32_t)buf : ((uint32_t)buf & ((1u << k) - 1u));
        consume(k);
        return v;
    }
    __device__ __forceinline__ uint64_t bits64(int k)                    // k in [0,63]
    {
        if (k <= 32) return bits(k);
        uint64_t lo = bits(32);
        return lo | ((uint64_t)bits(k - 32) << 32);
    }
    // `while (ReadBit()) q++` (VeloxEntropy.h:123-125)
    __device__ __forceinline__ uint64_t unary()
    {
        uint64_t q = 0;
        for (;;) {
            refill();
            const uint64_t inv = ~buf;
            const int t = inv ? __builtin_ctzll(inv) : 64;
            if (t < nb) { q += (uint64_t)t; consume(t + 1); return q; }
            q += (uint64_t)nb; buf = 0; nb = 0;                           // all buffered bits were ones
            if ((next_dw << 5) >= end_bit) return q;                      // past the end every bit reads 0
        }
    }
};

template <bool WIDE>
__global__ __launch_bounds__(64) void k_decode_serial(const uint32_t *payload, DecWorkspace ws, uint64_t n_chunks)
{
    typedef typename std::conditional<WIDE, int64_t, int32_t>::type X;
    const int lane = threadIdx.x;
    const uint64_t g = blockIdx.x;
    const uint64_t ci = g * 64 + lane;
    const bool valid = ci < n_chunks;
    vlx_chunk ch; ch.bit_off = 0; ch.n_bits = 0; ch.frames = 0; ch.out_off = 0; ch.n_out = 0; ch.high_res = 0; ch.reserved = 0;
    if (valid) ch = ws.chunks[ci];
    uint32_t *cs = ws.chunk_state + ci * CS_WORDS;
    uint32_t prev_flags = 0;
    if (WIDE) {                                                           // second pass: only flagged chunks
        prev_flags = valid ? cs[0] : 0;
        if (!__any(prev_flags & CSF_WIDE)) return;
    }
    const bool mine = valid && (!WIDE || (prev_flags & CSF_WIDE));
    uint32_t hi_slot = CS_NO_SLOT;                                        // where this chunk's high words go
    if (WIDE && mine) {
        hi_slot = atomicAdd(ws.wide_count, 1u);
        if (hi_slot >= ws.hi_cap) hi_slot = CS_NO_SLOT;                   // pool full: the host grows it and runs the batch again
    }
    const uint64_t end_bit = ch.bit_off + ch.n_bits;
    LaneReader rd; rd.init(payload, ch.bit_off, mine ? end_bit : 0);
    uint32_t flags = CSF_VALID;
    const uint32_t mode = rd.bits(1);                                     // VeloxCore.h:347
    const uint32_t use_ms = rd.bits(1);                                   // :353 / :366
    if (mode) flags |= CSF_MODE;
    if (use_ms) flags |= CSF_MS;
    const int frames = mine ? (int)ch.frames : 0;
    const int fmax = vlx_wave_max((mine && mode) ? frames : 0);
    uint32_t shifts = 0, lowpos[2] = {0, 0};
    bool wide = false;

    for (int s = 0; s < 2; s++) {
        bool act = mine && mode;
        uint32_t silent = 0; int lshift = 0; int32_t cf[VLX_LPC_ORDER];
#pragma unroll
        for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = 0;
        if (act) {
            silent = rd.bits(1);                                          // :132
            if (!silent) {
                shifts |= rd.bits(5) << (8 * s);                          // :135
                lshift = (int)rd.bits(5);                                 // :137
#pragma unroll
                for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = (int32_t)(int16_t)rd.bits(16);   // ReadS(16) :139
            } else flags |= (s ? CSF_SILENT_B : CSF_SILENT_A);
        }
        act = act && !silent;
        int32_t h[VLX_TAPS], w[VLX_TAPS]; X hist[VLX_LPC_ORDER];
#pragma unroll
        for (int i = 0; i < VLX_TAPS; i++) { h[i] = 0; w[i] = 0; }
#pragma unroll
        for (int i = 0; i < VLX_LPC_ORDER; i++) hist[i] = 0;
        uint64_t run_avg = 512;
        int32_t *dst = ws.scratch + (ci * 2 + s) * VLX_STREAM_LEN;          // stream-major: [chunk][stream][sample]
        int32_t *dst_hi = (WIDE && hi_slot != CS_NO_SLOT) ? ws.hi_pool + ((size_t)hi_slot * 2 + s) * VLX_STREAM_LEN : nullptr;
        const int smax = vlx_wave_max(act ? frames : 0);
        for (int i = 0; i < smax; i++) {
            if (act && i < frames) {
                const int k = 63 - __builtin_clzll(run_avg);              // :146
                const uint64_t q = rd.unary();
                uint64_t m;
                if (q < 64) m = (q << k) | (k ? rd.bits64(k) : 0ull);     // VeloxEntropy.h:128-132
                else m = rd.bits64(40);                                   // :135
                const int64_t fin = vlx_dezigzag(m);
                int64_t ns = 0;
#pragma unroll
                for (int t = 0; t < VLX_TAPS; t++) ns += (int64_t)h[t] * w[t];
                const int32_t pred = (int32_t)(ns >> 11);
                const int64_t res = (int64_t)((uint64_t)fin + (uint64_t)(int64_t)pred);   // :150
                uint64_t sum = 0;
#pragma unroll
                for (int j = 0; j < VLX_LPC_ORDER; j++) sum += (uint64_t)((int64_t)cf[j] * (int64_t)hist[j]);   // :152-154
                const int64_t x = (int64_t)((uint64_t)res + (uint64_t)((int64_t)sum >> lshift));   // :155
                if (!WIDE && x != (int64_t)(int32_t)x) wide = true;
                dst[i] = (int32_t)x;
                if (WIDE && dst_hi) dst_hi[i] = (int32_t)(x >> 32);
                const int32_t actual = (int32_t)res;
                const int32_t err = (int32_t)((uint32_t)actual - (uint32_t)pred);
                if (err != 0) {
                    const int32_t aerr = err < 0 ? (int32_t)(0u - (uint32_t)err) : err;
                    const int32_t sd = (err > 0 ? 1 : -1) * ((aerr > 1024) ? 16 : 4);
#pragma unroll
                    for (int t = 0; t < VLX_TAPS; t++) { const int hs = (h[t] > 0) - (h[t] < 0); w[t] += hs * sd; }
                    w[0] -= (w[0] > 0) - (w[0] < 0);
                    w[8] -= (w[8] > 0) - (w[8] < 0);
#pragma unroll
                    for (int t = VLX_TAPS - 1; t > 0; t--) h[t] = h[t - 1];
                    h[0] = actual;
                }
#pragma unroll
                for (int j = VLX_LPC_ORDER - 1; j > 0; j--) hist[j] = hist[j - 1];
                hist[0] = (X)x;
                run_avg = run_avg - (run_avg >> 3) + (m >> 3);
                if (run_avg < 1) run_avg = 1;
            }
        }
        if (mine && mode) {
            const uint64_t pos = rd.position();
            lowpos[s] = (uint32_t)(pos - ch.bit_off);
            if (ch.high_res && !silent) rd.init(payload, pos + 8ull * (uint64_t)frames, end_bit);   // skip the low bytes (:166-170)
        }
    }
    (void)fmax;
    if (mine) {
        if (wide) flags |= CSF_WIDE;
        if (WIDE) flags |= CSF_WIDE;
        cs[0] = flags; cs[1] = shifts; cs[2] = lowpos[0]; cs[3] = lowpos[1];
        if (WIDE) cs[CS_HI_SLOT] = hi_slot;
    }
}

// ----------------------------------------------------------------------------------------------
// Fast path: one LANE per STREAM, in two wave shapes that share one body.
//
// "Stream wave" (DS_CPW chunks): phase A, lanes 0..DS_CPW-1 parse stream L of "their" chunk (entropy
// only: k depends on the decoded values alone, never on the predictors) to learn where stream R
// starts (VeloxCore.h:355-356); phase B, lanes 2c / 2c+1 decode streams L / R of chunk c in full.
// Shortest latency per chunk, but the pre-parse is paid on half a wave.
// "Chunk wave" (64 chunks): lane c decodes stream L of chunk c in full, which also tells it where R
// starts, then stream R: no pre-parse at all (28 % fewer wave instructions per chunk) for twice the
// serial length.  The launcher mixes the two shapes so that the busiest SIMD has the least to do.
// Everything is 32-bit and branch-light; whatever leaves that envelope — values beyond
// int32, codes of a malformed stream, reads past the chunk end (which the reference defines as zero
// bits, VeloxEntropy.h:61-62) — flags the chunk CSF_WIDE and k_decode_serial<true> redoes it.
#define DS_CPW 32
#define DS_WAVES 4


// Compressed bits come through a per-lane ring in LDS ([DS_RING][64] dwords per wave: lane l only ever
// touches column l, so every access is conflict-free and needs no barrier), read through PipeWin
// (vlx_device.h): a window of two dwords + a bit offset, one v_alignbit per look, branch-free advance.
// Global loads are issued one tile ahead (16 B per lane and load) at wave-uniform refill points, so
// no wait on HBM latency sits inside the sample loop.  A lane that consumes faster than the ring is
// topped up (sustained > 32 bits per sample) underruns; that is detected and the chunk is redone
// by the careful kernel.  (Round 1 kept a 64-bit accumulator topped up under a divergent branch: 5 more
// VALU instructions per sample and phase, k_decode_streams 2.80 -> 2.67 ms on cfg2 with the window.)
//
// FUSE = 16 / 24 (packed PCM output of that width; 0 = off): a stream wave finishes the chunks that need nothing
// from behind their codes — mode 1, no high-res low bytes, whole tiles — by itself: lanes 2c / 2c+1 swap half a
// tile with a DPP neighbour swap, restore the LSB shift (VeloxFormat.h:220-226), undo M/S (VeloxCore.h:357-364)
// and store packed PCM (FormatHandler::SamplesToBytes, VeloxFormat.h:160-189) straight from the sample loop;
// the chunk is flagged CSF_DONE and k_decode_finalize skips it.  32-bit arithmetic is exact for these widths: only
// the low 16 / 24 bits of the results are kept, and bit b of (v + 1) >> 1 depends on bits <= b + 1 of v alone.
// A chunk that turns out bad is still flagged CSF_WIDE and redone in full (k_decode_serial<true> + finalize
// overwrite whatever was stored).  High-res chunks cannot be finished here: stream R's low bytes start where its
// codes end.
// (3 waves per SIMD: the launcher's mixes need up to three resident; without the bound hipcc takes 184 VGPRs)
template <int FUSE>
__global__ __launch_bounds__(64 * DS_WAVES, 3) void k_decode_streams(const uint8_t *payload, uint64_t payload_bytes,
                                                                  DecWorkspace ws, uint64_t n_chunks, uint32_t n_chunk_waves, void *out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(blockIdx.x * DS_WAVES + (threadIdx.x >> 6));   // wave-uniform
    const bool cw = wv < n_chunk_waves;                                    // chunk wave (64 chunks) or stream wave (DS_CPW)
    const uint64_t c0 = cw ? (uint64_t)wv * 64 : (uint64_t)n_chunk_waves * 64 + (uint64_t)(wv - n_chunk_waves) * DS_CPW;
    if (c0 >= n_chunks) return;                                           // whole wave idle
    __shared__ uint32_t rings[DS_WAVES][DS_RING * 64];
    uint32_t *const myring = &rings[threadIdx.x >> 6][lane];
    const uint64_t addr = (uint64_t)payload;
    const uint4 *abase = (const uint4 *)(addr & ~15ull);
    const uint64_t dbits = (addr & 15ull) * 8ull;
    const uint32_t last_g = (uint32_t)((payload_bytes + (addr & 15ull) - 1ull) >> 4);

    uint32_t rstart = 0, hdr = 0; bool bad = false;
    if (!cw) {
        // ---- phase A (stream waves): where does stream R start?
        uint32_t a_rstart = 0, a_hdr = 0; bool a_bad = false;
        // lanes DS_CPW..63 shadow lanes 0..DS_CPW-1 (same chunk, same instructions, results unused), so that
        // the whole wave stays on the unchecked copy of the loop below for as long as lanes 0..DS_CPW-1 do
        const uint64_t ci = c0 + (lane & (DS_CPW - 1));
        const bool mine = ci < n_chunks;
        vlx_chunk ch; ch.bit_off = 0; ch.n_bits = 0; ch.frames = 0; ch.high_res = 0;
        if (mine) ch = ws.chunks[ci];
        PipeWin rd; rd.init(myring, abase, last_g, ch.bit_off + dbits);
        const uint32_t mode = rd.take(1), use_ms = rd.take(1);            // VeloxCore.h:347,353
        a_hdr = mode | (use_ms << 1);
        bool act = mine && mode;
        const uint32_t silent = act ? rd.take(1) : 1u;                    // :132
        act = act && !silent;
        if (act) { rd.take(10); for (int j = 0; j < 4; j++) rd.take(32); }   // shift_lsb, lpc_shift, 8 x 16-bit coefs
        int frames = act ? (int)ch.frames : 0;
        const int fmax = vlx_wave_max(frames);
        uint32_t run_avg = 512;
        // While every lane of the wave still has samples left the per-sample activity test is dropped
        // (a lane that turned bad decodes garbage that nobody reads: its chunk is redone); only the
        // ragged end of a batch runs the checked copy of the body.
        auto tile_a = [&](auto checked, int t0) {
            rd.refill_point();
#pragma unroll
            for (int j = 0; j < DS_TILE; j++) {
                if (!decltype(checked)::value || t0 + j < frames) {
                    const uint32_t k = 31u - (uint32_t)__builtin_clz(run_avg);
                    bool b = false;
                    const uint32_t m = dp_sample(rd, k, b);
                    if (b) { a_bad = true; frames = 0; }
                    run_avg = run_avg - (run_avg >> 3) + (m >> 3);      // (the floor of 1, :161, never fires: x - (x >> 3) >= 1 for x >= 1, and it starts at 512)
                }
            }
        };
        const int fmin = vlx_wave_min(frames) & ~(DS_TILE - 1);
        int t0 = 0;
        for (; t0 < fmin; t0 += DS_TILE) tile_a(std::false_type{}, t0);
        for (; t0 < fmax; t0 += DS_TILE) tile_a(std::true_type{}, t0);
        rd.refill_point();
        if (rd.under) a_bad = true;
        a_rstart = rd.used() + ((act && ch.high_res) ? 8u * ch.frames : 0u);   // low bytes sit between L and R (:166-170)
        rstart = __shfl(a_rstart, lane >> 1);
        hdr = __shfl(a_hdr, lane >> 1);
        bad = __shfl((int)a_bad, lane >> 1) != 0;
    }

    // ---- full decode: stream waves, one pass with lane -> (chunk lane/2, stream lane&1);
    //      chunk waves, lane -> chunk, pass 0 = stream L, pass 1 = stream R
    const uint64_t ci = c0 + (cw ? lane : (lane >> 1));
    const bool mine = ci < n_chunks;
    vlx_chunk ch; ch.bit_off = 0; ch.n_bits = 0; ch.frames = 0; ch.high_res = 0;
    if (mine) ch = ws.chunks[ci];
    char *const sbase = (char *)(ws.scratch + c0 * 2 * VLX_STREAM_LEN);   // stream-major [chunk][stream][sample]
    uint32_t l_silent = 0, l_shift = 0, l_end = 0;                        // stream L's results (chunk waves)
    uint32_t silent = 0, shift_lsb = 0, endpos = 0;
    bool fuse = false;                                                    // this lane pair finishes its chunk itself
    const int n_pass = cw ? 2 : 1;
#pragma unroll 1
    for (int pass = 0; pass < n_pass; pass++) {
        const int s = cw ? pass : (lane & 1);
        const uint32_t init_rel = cw ? (pass ? rstart : 0u) : (s ? rstart : 2u);
        PipeWin rd; rd.init(myring, abase, last_g, ch.bit_off + dbits + init_rel);
        if (cw && pass == 0) hdr = rd.take(2);                            // mode, use_MS (VeloxCore.h:347,353)
        const uint32_t mode = hdr & 1u;
        bool act = mine && mode && !bad;
        uint32_t lshift = 0; int32_t cf[VLX_LPC_ORDER];
        silent = 0; shift_lsb = 0;
#pragma unroll
        for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = 0;
        if (act) {
            silent = rd.take(1);
            if (!silent) {
                shift_lsb = rd.take(5); lshift = rd.take(5);
#pragma unroll
                for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = (int32_t)(int16_t)rd.take(16);   // ReadS(16), :139
            }
        }
        uint32_t sh_oth = 0; char *obase = nullptr;
        if (FUSE && !cw) {
            constexpr int W = FUSE / 8;
            const uint32_t mine_hdr = (act ? 1u : 0u) | (silent << 1) | (shift_lsb << 2);   // act: mode 1 and not bad, silent or not
            const uint32_t oth_hdr = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine_hdr, 0xB1, 0xF, 0xF, true);
            sh_oth = oth_hdr >> 2;
            const uint64_t A0 = (uint64_t)out + ch.out_off * (uint64_t)W;
            fuse = mine && (mine_hdr & 1u) && (oth_hdr & 1u) && !(silent && ((oth_hdr >> 1) & 1u)) && !ch.high_res &&
                   ch.frames && (ch.frames & (2 * DS_TILE - 1)) == 0 && ch.n_out == 2 * ch.frames && (A0 & 3ull) == 0;
            obase = (char *)A0 + (uint64_t)((lane & 1) * DS_TILE * 2) * W;                  // even lane: frames 0-15 of a tile, odd lane: 16-31
        }
        act = act && !silent;
        int frames = act ? (int)ch.frames : 0;
        const int fmax = vlx_wave_max(frames);
        VlxLms lms; lms.reset();
        int32_t hist[VLX_LPC_ORDER];
#pragma unroll
        for (int j = 0; j < VLX_LPC_ORDER; j++) hist[j] = 0;
        uint32_t run_avg = 512;
        // every lane appends a whole 64-B line of its stream per 16 samples
        uint32_t soff = (cw ? (uint32_t)(2 * lane + pass) : (uint32_t)lane) * (VLX_STREAM_LEN * 4u);
        // (unchecked while every lane has samples left, checked for the ragged end: see phase A)
        auto tile_b = [&](auto checked, int t0) {
            int32_t xs[2 * DS_TILE];
            static_assert((2 * DS_TILE) % 16 == 0, "whole turns of the predictor's ring per tile");
            vlx_static_for<2 * DS_TILE>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if (j % DS_TILE == 0) rd.refill_point();
                int32_t x = 0;
                if (!decltype(checked)::value || t0 + j < frames) {
                    const uint32_t k = 31u - (uint32_t)__builtin_clz(run_avg);                   // :146
                    bool b = false;
                    const uint32_t m = dp_sample(rd, k, b);
                    const int32_t fin = (int32_t)((m >> 1) ^ (0u - (m & 1u)));                   // DeZigZag, VeloxEntropy.h:96
                    const int32_t pred = lms.template predict2<j & 15>();                         // (the tile is whole turns of the predictor's ring)
                    const int32_t res = (int32_t)((uint32_t)fin + (uint32_t)pred);               // (int32)resLPC, :150 / :27
                    int64_t sum = 0, sum1 = 0;
#pragma unroll
                    for (int jj = 0; jj < VLX_LPC_ORDER; jj += 2) { sum += (int64_t)cf[jj] * hist[jj]; sum1 += (int64_t)cf[jj + 1] * hist[jj + 1]; }   // :152-154
                    sum += sum1;
                    const int64_t x64 = (int64_t)fin + (int64_t)pred + (sum >> lshift);          // :155
                    x = (int32_t)x64;
                    if (b || x64 != (int64_t)x) { bad = true; frames = 0; }
                    lms.template update<j & 15>(fin, res);
                    run_avg = run_avg - (run_avg >> 3) + (m >> 3);                               // :159-161 (the floor never fires, as above)
                }
                xs[j] = x;
                // a lane past its own frame count shifts zeros into a history nobody reads again
#pragma unroll
                for (int jj = VLX_LPC_ORDER - 1; jj > 0; jj--) hist[jj] = hist[jj - 1];
                hist[0] = x;
            });
            if (FUSE && fuse) {
                if (t0 + 2 * DS_TILE <= (int)ch.frames) {                                           // same answer on both lanes of the pair
                    constexpr int W = FUSE / 8;
                    const bool odd = lane & 1;
                    const bool ms = (hdr & 2u) != 0;
                    uint32_t o0[DS_TILE], o1[DS_TILE];
#pragma unroll
                    for (int j = 0; j < DS_TILE; j++) {
                        const int32_t keep = odd ? xs[DS_TILE + j] : xs[j];
                        const int32_t give = odd ? xs[j] : xs[DS_TILE + j];
                        const int32_t got = __builtin_amdgcn_mov_dpp(give, 0xB1, 0xF, 0xF, true);       // the neighbour's half
                        const uint32_t vk = (uint32_t)keep << shift_lsb, vg = (uint32_t)got << sh_oth;   // Restore
                        const uint32_t vL = odd ? vg : vk, vR = odd ? vk : vg;
                        o0[j] = ms ? vL + (uint32_t)((int32_t)(vR + 1u) >> 1) : vL;                      // VeloxCore.h:357-364
                        o1[j] = ms ? vL - (uint32_t)((int32_t)vR >> 1) : vR;
                    }
                    char *dst = obase + (uint64_t)t0 * 2 * W;
                    if (W == 2) {
#pragma unroll
                        for (int q = 0; q < DS_TILE / 4; q++) {
                            uint32_t d[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) d[e] = (o0[4 * q + e] & 0xFFFFu) | (o1[4 * q + e] << 16);
                            vlx_gstore16_a4(dst + 16 * q, make_uint4(d[0], d[1], d[2], d[3]));
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < DS_TILE / 8; q++) {                                         // 8 frames = 16 samples = 12 dwords
                            uint32_t t[16], d[12];
#pragma unroll
                            for (int e = 0; e < 8; e++) { t[2 * e] = o0[8 * q + e] & 0xFFFFFFu; t[2 * e + 1] = o1[8 * q + e] & 0xFFFFFFu; }
#pragma unroll
                            for (int g = 0; g < 4; g++) {                                               // 4 samples -> 3 dwords
                                d[3 * g] = t[4 * g] | (t[4 * g + 1] << 24);
                                d[3 * g + 1] = (t[4 * g + 1] >> 8) | (t[4 * g + 2] << 16);
                                d[3 * g + 2] = (t[4 * g + 2] >> 16) | (t[4 * g + 3] << 8);
                            }
#pragma unroll
                            for (int g = 0; g < 3; g++) vlx_gstore16_a4(dst + 48 * q + 16 * g, make_uint4(d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]));
                        }
                    }
                }
            } else if (mine) {
                // samples past a lane's own frame count are never read
#pragma unroll
                for (int q = 0; q < 2 * DS_TILE / 4; q++) vlx_gstore16(sbase + soff + 16 * q, make_uint4((uint32_t)xs[4 * q], (uint32_t)xs[4 * q + 1], (uint32_t)xs[4 * q + 2], (uint32_t)xs[4 * q + 3]));
            }
            soff += 8u * DS_TILE;
        };
        const int fmin = vlx_wave_min(frames) & ~(2 * DS_TILE - 1);
        int t0 = 0;
        for (; t0 < fmin; t0 += 2 * DS_TILE) tile_b(std::false_type{}, t0);
        for (; t0 < fmax; t0 += 2 * DS_TILE) tile_b(std::true_type{}, t0);
        rd.refill_point();
        if (rd.under) bad = true;
        endpos = init_rel + rd.used();                                      // where this stream's low bytes start
        if (act && (uint64_t)endpos + (ch.high_res ? 8ull * ch.frames : 0ull) > ch.n_bits) bad = true;   // ran past the chunk end
        if (mine && mode && silent && (uint64_t)endpos > ch.n_bits) bad = true;
        if (mine && ch.n_bits < 2) bad = true;                            // even the mode bits lie past the end
        if (cw && pass == 0) {
            l_silent = silent; l_shift = shift_lsb; l_end = endpos;
            rstart = endpos + ((act && ch.high_res) ? 8u * ch.frames : 0u);   // low bytes sit between L and R (:166-170)
        }
    }
    // ---- per-chunk state for k_decode_finalize (and the redo flag for k_decode_serial<true>)
    const uint32_t mode = hdr & 1u;
    if (cw) {
        if (mine) {
            uint32_t *cs = ws.chunk_state + ci * CS_WORDS;
            uint32_t flags = CSF_VALID | (mode ? CSF_MODE : 0u) | ((hdr & 2u) ? CSF_MS : 0u);
            if (mode && l_silent) flags |= CSF_SILENT_A;
            if (mode && silent) flags |= CSF_SILENT_B;
            if (bad) flags |= CSF_WIDE;
            cs[0] = flags; cs[1] = l_shift | (shift_lsb << 8); cs[2] = l_end; cs[3] = endpos;
        }
    } else {
        const int s = lane & 1;
        const uint32_t o_silent = (uint32_t)__shfl_xor((int)silent, 1);
        const uint32_t o_shift = (uint32_t)__shfl_xor((int)shift_lsb, 1);
        const bool o_bad = __shfl_xor((int)bad, 1) != 0;
        if (mine) {
            uint32_t *cs = ws.chunk_state + ci * CS_WORDS;
            if (s == 0) {
                uint32_t flags = CSF_VALID | (mode ? CSF_MODE : 0u) | ((hdr & 2u) ? CSF_MS : 0u);
                if (mode && silent) flags |= CSF_SILENT_A;
                if (mode && o_silent) flags |= CSF_SILENT_B;
                if (bad || o_bad) flags |= CSF_WIDE;
                else if (FUSE && fuse) flags |= CSF_DONE;
                cs[0] = flags; cs[1] = shift_lsb | (o_shift << 8); cs[2] = endpos;
            } else cs[3] = endpos;
        }
    }
}

template <int OUTFMT> __device__ __forceinline__ void vlx_store_sample(void *out, uint64_t idx, int64_t v)
{
    if (OUTFMT == VLX_FMT_I64) ((int64_t *)out)[idx] = v;
    else if (OUTFMT == VLX_FMT_S32) ((int32_t *)out)[idx] = (int32_t)v;
    else if (OUTFMT == VLX_FMT_S16) ((int16_t *)out)[idx] = (int16_t)v;
    else { uint8_t *p = (uint8_t *)out + idx * 3; int32_t x = (int32_t)v; p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); p[2] = (uint8_t)(x >> 16); }
}

// vector stores whose address is only known to be dword aligned

// One workgroup per chunk, fully streaming: thread t takes frames {1024 r + 4 t .. +3}, r = 0..3, of
// both streams — 16-byte scratch reads, one unaligned 32-bit window of low bytes per stream, and
// 4 frames x 2 channels of packed PCM out (24 contiguous bytes per lane for 24-bit).  Restores the
// LSB shift and the high-res low bytes (VeloxCore.h:164-171), reads raw blocks (:179-184), undoes
// M/S (:357-364) and packs (FormatHandler::SamplesToBytes, VeloxFormat.h:160-189).
template <int OUTFMT>
__global__ __launch_bounds__(256) void k_decode_finalize(const uint32_t *payload, DecWorkspace ws, uint64_t n_chunks, void *out)
{
    constexpr bool F32 = OUTFMT == VLX_FMT_F32_MERGE || OUTFMT == VLX_FMT_F32_P16 || OUTFMT == VLX_FMT_F32_P24;
    constexpr int STOREFMT = F32 ? VLX_FMT_S32 : OUTFMT;                  // float32 goes out as its bit pattern
    constexpr int W = OUTFMT == VLX_FMT_I64 ? 8 : (F32 ? 4 : OUTFMT / 8);
    const uint64_t ci = blockIdx.x;
    const vlx_chunk ch = ws.chunks[ci];
    const uint32_t *cs = ws.chunk_state + ci * CS_WORDS;
    const uint32_t flags = cs[0], shifts = cs[1];
    if (flags & CSF_DONE) return;                                         // a stream wave stored this chunk's PCM itself
    const uint64_t end_bit = ch.bit_off + ch.n_bits;
    const int frames = (int)ch.frames;
    const bool lows = ch.high_res != 0;
    const bool use_hi = (flags & CSF_WIDE) && cs[CS_HI_SLOT] != CS_NO_SLOT;
    const int32_t *const hi_base = use_hi ? ws.hi_pool + (size_t)cs[CS_HI_SLOT] * 2 * VLX_STREAM_LEN : nullptr;
    const uint64_t A0 = (uint64_t)out + ch.out_off * (uint64_t)W;
    const bool aligned = (A0 & (W == 8 ? 7ull : 3ull)) == 0;
    for (int r = 0; r < 4; r++) {
        const int f0 = r * 1024 + 4 * (int)threadIdx.x;
        if (r * 1024 >= frames) break;
        if (f0 >= frames) continue;
        int64_t v[2][4];
        if (flags & CSF_MODE) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (flags & (s ? CSF_SILENT_B : CSF_SILENT_A)) {                             // :133
#pragma unroll
                    for (int e = 0; e < 4; e++) v[s][e] = 0;
                    continue;
                }
                const size_t o = (ci * 2 + s) * VLX_STREAM_LEN + (size_t)f0;
                const vlx_v4u xt = __builtin_nontemporal_load((const vlx_g_v4u *)(ws.scratch + o));   // read once: non-temporal, like the stores below (1.085 -> 1.062 ms on cfg2, same box)
                const int4 xl = make_int4((int)xt.x, (int)xt.y, (int)xt.z, (int)xt.w);
                int64_t x[4] = {xl.x, xl.y, xl.z, xl.w};
                if (use_hi) {
                    const int4 xh = *(const int4 *)(hi_base + (size_t)s * VLX_STREAM_LEN + (size_t)f0);
                    x[0] = (int64_t)(((uint64_t)(uint32_t)xh.x << 32) | (uint32_t)xl.x);
                    x[1] = (int64_t)(((uint64_t)(uint32_t)xh.y << 32) | (uint32_t)xl.y);
                    x[2] = (int64_t)(((uint64_t)(uint32_t)xh.z << 32) | (uint32_t)xl.z);
                    x[3] = (int64_t)(((uint64_t)(uint32_t)xh.w << 32) | (uint32_t)xl.w);
                }
                const int sh = (int)((shifts >> (8 * s)) & 0xFF);
                uint32_t low4 = 0;
                if (lows) {                                                                   // 4 consecutive low bytes
                    const uint64_t P = ch.bit_off + cs[2 + s] + 8ull * (uint64_t)f0;
                    const uint64_t w2 = ((uint64_t)vlx_load_clamped(payload, (P >> 5) + 1, end_bit) << 32) |
                                        vlx_load_clamped(payload, P >> 5, end_bit);
                    low4 = (uint32_t)(w2 >> (P & 31ull));
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    uint64_t u = (uint64_t)x[e] << sh;                                        // Restore (VeloxFormat.h:220-226)
                    if (lows) u = (u << 8) | ((low4 >> (8 * e)) & 0xFFu);                     // :166-170
                    v[s][e] = (int64_t)u;
                }
            }
        } else {                                                                              // ReadRawBlock :179-184
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint64_t jj = (uint64_t)(f0 + e < frames ? f0 + e : f0);
                v[0][e] = vlx_dezigzag(vlx_read_abs(payload, ch.bit_off + 2 + 40ull * jj, 40, end_bit));
                v[1][e] = vlx_dezigzag(vlx_read_abs(payload, ch.bit_off + 2 + 40ull * ((uint64_t)frames + jj), 40, end_bit));
            }
        }
        int64_t o[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            o[2 * e] = v[0][e]; o[2 * e + 1] = v[1][e];
            if (flags & CSF_MS) { o[2 * e] = v[0][e] + ((v[1][e] + 1) >> 1); o[2 * e + 1] = v[0][e] - (v[1][e] >> 1); }   // :357-364
        }
        const uint32_t k0 = 2u * (uint32_t)f0;                                                // first output sample of this lane
        if (F32) {                                                                            // main.cpp:262-281, fused into the store
#pragma unroll
            for (int q = 0; q < 8; q++) {
                uint32_t u;
                if (OUTFMT == VLX_FMT_F32_MERGE) {                                            // MergeFloat32, VeloxFormat.h:119-125
                    int64_t m = o[q]; uint32_t sign = 0;
                    if (m < 0) { sign = 1; m = (int64_t)(0ull - (uint64_t)m); }
                    const uint32_t e = (f0 + (q >> 1) < frames && k0 + (uint32_t)q < ch.n_out) ? ws.exps[ch.out_off + k0 + q] : 0u;
                    u = (sign << 31) | (e << 23) | ((uint32_t)m & 0x7FFFFFu);
                } else u = __float_as_uint((float)o[q] * (OUTFMT == VLX_FMT_F32_P16 ? (1.0f / 32768.0f) : (1.0f / 8388608.0f)));   // PromoteIntToFloat, :80
                o[q] = (int64_t)u;
            }
        }
        if (aligned && k0 + 8u <= ch.n_out && f0 + 4 <= frames) {
            uint8_t *dst = (uint8_t *)A0 + (uint64_t)k0 * W;
            // (global stores by address space: `dst` comes out of integer arithmetic, and through a generic pointer hipcc
            // emitted flat_store for the one store that carries all of this kernel's output)
            if (W == 8) {
#pragma unroll
                for (int q = 0; q < 4; q++)                                                       // int64 output is 8-B aligned
                    vlx_gstore16_a4(dst + 16 * q, make_uint4((uint32_t)o[2 * q], (uint32_t)((uint64_t)o[2 * q] >> 32), (uint32_t)o[2 * q + 1], (uint32_t)((uint64_t)o[2 * q + 1] >> 32)));
            } else if (W == 4) {
                vlx_gstore16_a4(dst, make_uint4((uint32_t)o[0], (uint32_t)o[1], (uint32_t)o[2], (uint32_t)o[3]));
                vlx_gstore16_a4(dst + 16, make_uint4((uint32_t)o[4], (uint32_t)o[5], (uint32_t)o[6], (uint32_t)o[7]));
            } else if (W == 2) {
                uint32_t d[4];
#pragma unroll
                for (int q = 0; q < 4; q++) d[q] = ((uint32_t)o[2 * q] & 0xFFFFu) | ((uint32_t)o[2 * q + 1] << 16);
                vlx_gstore16_a4(dst, make_uint4(d[0], d[1], d[2], d[3]));
            } else {                                                                          // 8 x 24 bit = 6 dwords
                uint32_t t[8];
#pragma unroll
                for (int q = 0; q < 8; q++) t[q] = (uint32_t)o[q] & 0xFFFFFFu;
                uint32_t d[6];
                d[0] = t[0] | (t[1] << 24); d[1] = (t[1] >> 8) | (t[2] << 16); d[2] = (t[2] >> 16) | (t[3] << 8);
                d[3] = t[4] | (t[5] << 24); d[4] = (t[5] >> 8) | (t[6] << 16); d[5] = (t[6] >> 16) | (t[7] << 8);
                { vlx_v4u t; t.x = d[0]; t.y = d[1]; t.z = d[2]; t.w = d[3]; __builtin_nontemporal_store(t, (vlx_g_v4u_a4 *)dst);
                  vlx_v2u u; u.x = d[4]; u.y = d[5]; __builtin_nontemporal_store(u, (vlx_g_v2u_a4 *)(dst + 16)); }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (f0 + (q >> 1) < frames && k0 + (uint32_t)q < ch.n_out) vlx_store_sample<STOREFMT>(out, ch.out_off + k0 + q, o[q]);
        }
    }
}

// The careful pass: only chunks the fast kernel flagged CSF_WIDE do any work here.
hipError_t vlx_launch_decode_serial(hipStream_t st, const uint8_t *payload, const DecWorkspace &ws, uint64_t n_chunks)
{
    if (!n_chunks) return hipSuccess;
    const unsigned groups = (unsigned)((n_chunks + 63) / 64);
    k_decode_serial<true><<<dim3(groups), dim3(64), 0, st>>>((const uint32_t *)payload, ws, n_chunks);
    return hipGetLastError();
}

// Wave-shape mix.  Measured on MI355X (tools/bench_decode_shapes.py, 24-bit material): a SIMD that
// issues for several waves spends about 1.12 ms per stream wave (DS_CPW chunks) and 1.78 ms per chunk
// wave (64 chunks); a wave that has the SIMD to itself is latency-bound instead and takes 1.7 / 2.4 ms.
// With `a` rounds of chunk waves and `b` rounds of stream waves per SIMD the busiest SIMD therefore
// needs max(a * CW + b * SW, latency of the longest wave): take the cheapest mix that holds every
// chunk.  Small batches come out as stream waves only (lowest latency), mid-size ones as one chunk
// wave plus one stream wave per SIMD, large ones as chunk waves plus a stream-wave tail.
// Units: 10 us.
#define DS_COST_CW 172
#define DS_COST_SW 108
#define DS_LAT_CW  222
#define DS_LAT_SW  163
static std::atomic<int> g_decode_shape{0};                             // 0 = by cost, 1 = stream waves only, 2 = chunk waves only, 3 = pipe
// (batches up to VLX_PIPE_MAX_CHUNKS go to the pipelined small-batch shapes, vlx_pipe.hip)
void vlx_decode_shape_set(int shape) { g_decode_shape.store(shape, std::memory_order_relaxed); }
int  vlx_decode_shape_get() { return g_decode_shape.load(std::memory_order_relaxed); }

static uint32_t vlx_chunk_wave_count(uint64_t n_chunks)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const uint64_t simds = 4ull * (uint64_t)cus;
    const int shape = g_decode_shape.load(std::memory_order_relaxed);
    const uint64_t full = n_chunks / 64;
    if (shape == 1) return 0;
    if (shape == 2) return (uint32_t)((n_chunks + 63) / 64);
    const uint64_t a_max = (full + simds - 1) / simds;
    uint64_t best_cost = ~0ull, best_w2 = 0;
    for (uint64_t a = 0; a <= a_max; a++) {
        const uint64_t w2 = a * simds < full ? a * simds : full;
        const uint64_t w1 = (n_chunks - 64 * w2 + DS_CPW - 1) / DS_CPW;
        const uint64_t rounds2 = (w2 + simds - 1) / simds, rounds1 = (w1 + simds - 1) / simds;
        const uint64_t issue = rounds2 * DS_COST_CW + rounds1 * DS_COST_SW;
        const uint64_t lat = rounds2 ? DS_LAT_CW : DS_LAT_SW;
        const uint64_t cost = issue > lat ? issue : lat;
        if (cost < best_cost) { best_cost = cost; best_w2 = w2; }
    }
    return (uint32_t)best_w2;
}

// Name of the kernel vlx_launch_decode_streams launches for a batch of this size (the label of its profiling slot).
const char *vlx_decode_kernel_label(uint64_t n_chunks)
{
    const int shape = g_decode_shape.load(std::memory_order_relaxed);
    const bool pipe = shape == VLX_SHAPE_PIPE16 || shape == VLX_SHAPE_PIPE32 || shape == VLX_SHAPE_PIPE64 || shape == VLX_SHAPE_PIPE ||
                      (shape == VLX_SHAPE_AUTO && n_chunks <= VLX_PIPE_MAX_CHUNKS);
    return pipe ? "k_decode_pipe" : "k_decode_streams";
}

hipError_t vlx_launch_decode_streams(hipStream_t st, const uint8_t *payload, uint64_t payload_bytes, const DecWorkspace &ws, uint64_t n_chunks,
                                     void *out, int fuse_fmt)
{
    if (!n_chunks) return hipSuccess;
    const int shape = g_decode_shape.load(std::memory_order_relaxed);
    if (shape == VLX_SHAPE_PIPE16 || shape == VLX_SHAPE_PIPE32 || shape == VLX_SHAPE_PIPE64)
        return vlx_launch_decode_pipe(st, payload, payload_bytes, ws, n_chunks, shape == VLX_SHAPE_PIPE16 ? 16 : (shape == VLX_SHAPE_PIPE32 ? 32 : 64));
    if (shape == VLX_SHAPE_PIPE || (shape == VLX_SHAPE_AUTO && n_chunks <= VLX_PIPE_MAX_CHUNKS))
        return vlx_launch_decode_pipe(st, payload, payload_bytes, ws, n_chunks, 0);
    const uint32_t w2 = vlx_chunk_wave_count(n_chunks);
    const uint64_t rest = n_chunks > 64ull * w2 ? n_chunks - 64ull * w2 : 0;
    const uint64_t waves = w2 + (rest + DS_CPW - 1) / DS_CPW;
    const dim3 grid((unsigned)((waves + DS_WAVES - 1) / DS_WAVES)), block(64 * DS_WAVES);
    if (out && fuse_fmt == VLX_FMT_S16) k_decode_streams<16><<<grid, block, 0, st>>>(payload, payload_bytes, ws, n_chunks, w2, out);
    else if (out && fuse_fmt == VLX_FMT_S24) k_decode_streams<24><<<grid, block, 0, st>>>(payload, payload_bytes, ws, n_chunks, w2, out);
    else k_decode_streams<0><<<grid, block, 0, st>>>(payload, payload_bytes, ws, n_chunks, w2, nullptr);
    return hipGetLastError();
}

hipError_t vlx_launch_decode_finalize(hipStream_t st, const uint8_t *payload, const DecWorkspace &ws, uint64_t n_chunks,
                                      void *out, int out_fmt)
{
    if (!n_chunks) return hipSuccess;
    const dim3 grid((unsigned)n_chunks), block(256);
    const uint32_t *p = (const uint32_t *)payload;
    switch (out_fmt) {
    case VLX_FMT_I64: k_decode_finalize<VLX_FMT_I64><<<grid, block, 0, st>>>(p, ws, n_chunks, out); break;
    case VLX_FMT_S32: k_decode_finalize<VLX_FMT_S32><<<grid, block, 0, st>>>(p, ws, n_chunks, out); break;
    case VLX_FMT_S24: k_decode_finalize<VLX_FMT_S24><<<grid, block, 0, st>>>(p, ws, n_chunks, out); break;
    case VLX_FMT_S16: k_decode_finalize<VLX_FMT_S16><<<grid, block, 0, st>>>(p, ws, n_chunks, out); break;
    case VLX_FMT_F32_MERGE: k_decode_finalize<VLX_FMT_F32_MERGE><<<grid, block, 0, st>>>(p, ws, n_chunks, out); break;
    case VLX_FMT_F32_P16: k_decode_finalize<VLX_FMT_F32_P16><<<grid, block, 0, st>>>(p, ws, n_chunks, out); break;
    case VLX_FMT_F32_P24: k_decode_finalize<VLX_FMT_F32_P24><<<grid, block, 0, st>>>(p, ws, n_chunks, out); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// Chunk table built in HBM from the chunk byte sizes (vlx_decode_tracks_device): what
// vlx_chunks_from_sizes computes on the host, i.e. the positions the reference's refill walk
// (VeloxCore.h:338-353) would visit.  Only 4 bytes per chunk cross PCIe instead of a 40-byte descriptor.
// `stage` is the caller's PINNED HOST staging area ([track table, stage_words - n_chunks dwords][sizes]): this kernel reads it
// over PCIe and leaves the device copy the later kernels use — a copy-engine transfer in front of a 1 ms batch cost its
// own duration plus a ~25 us hand-over between the engines.  It also clears the batch's wide-chunk counter (one
// stream operation less in front of the decode kernel).
__global__ __launch_bounds__(256) void k_sizes_to_steps(const uint32_t *stage, uint32_t *dev_copy, uint64_t stage_words, uint64_t *scan,
                                                        uint64_t n_chunks, uint32_t *zero_me)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t head = stage_words - n_chunks;                         // dwords of the track table
    if (c == 0) { scan[0] = 0; if (zero_me) *zero_me = 0; }
    if (c < n_chunks) {
        const uint32_t sz = stage[head + c];
        dev_copy[head + c] = sz;
        scan[c + 1] = 32ull + 8ull * sz;                                  // u32 size prefix + body (:340-344)
    }
    for (uint64_t i = c; i < head; i += (uint64_t)gridDim.x * blockDim.x) dev_copy[i] = stage[i];
}

__global__ __launch_bounds__(256) void k_chunks_build(const uint8_t *payload, const uint32_t *sizes, const uint64_t *scan, const DecTrack *tracks,
                                                      uint32_t n_tracks, vlx_chunk *chunks, uint64_t n_chunks)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    uint32_t lo = 0, hi = n_tracks;                                       // last track whose first chunk is <= c
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tracks[mid].first_chunk <= c) lo = mid; else hi = mid; }
    const DecTrack T = tracks[lo];
    const uint64_t idx = c - T.first_chunk;
    // every chunk before the track's tail hands out 8192 samples; an odd tail is followed by a one-sample chunk
    const uint64_t full = T.total / (2 * VLX_STREAM_LEN);
    const uint64_t decoded = idx <= full ? idx * (2 * VLX_STREAM_LEN) : T.total - 1;
    const uint64_t remaining = T.total - decoded;
    uint64_t frames = remaining / 2 < VLX_STREAM_LEN ? remaining / 2 : VLX_STREAM_LEN;   // :346-347
    if (frames == 0) frames = remaining;
    vlx_chunk ch;
    ch.bit_off = T.first_bit + (scan[c] - scan[T.first_chunk]) + 32;
    ch.n_bits = 8ull * sizes[c];
    ch.out_off = T.out_base + decoded;
    ch.frames = (uint32_t)frames;
    ch.n_out = (uint32_t)(2 * frames < remaining ? 2 * frames : remaining);
    // VLX_HIGH_RES_FROM_PAYLOAD: the bit right in front of the first chunk size (VeloxCore.h:329)
    ch.high_res = T.high_res == VLX_HIGH_RES_FROM_PAYLOAD ? (uint32_t)(payload[(T.first_bit - 1) >> 3] >> ((T.first_bit - 1) & 7)) & 1u : T.high_res;
    ch.reserved = 0;
    chunks[c] = ch;
}

hipError_t vlx_launch_chunks_build(hipStream_t st, const uint8_t *payload, const void *h_stage, void *d_stage, size_t track_bytes, uint64_t *scan,
                                   uint32_t n_tracks, vlx_chunk *chunks, uint64_t n_chunks, uint32_t *zero_me)
{
    if (!n_chunks) return hipSuccess;
    const unsigned groups = (unsigned)((n_chunks + 255) / 256);
    const DecTrack *tracks = (const DecTrack *)d_stage;
    const uint32_t *sizes = (const uint32_t *)((const uint8_t *)d_stage + track_bytes);
    k_sizes_to_steps<<<dim3(groups), dim3(256), 0, st>>>((const uint32_t *)h_stage, (uint32_t *)d_stage, track_bytes / 4 + n_chunks, scan, n_chunks, zero_me);
    hipError_t e = vlx_launch_scan_u64(st, scan, n_chunks);
    if (e != hipSuccess) return e;
    k_chunks_build<<<dim3(groups), dim3(256), 0, st>>>(payload, sizes, scan, tracks, n_tracks, chunks, n_chunks);
    return hipGetLastError();
}
