// Small-batch ("pipe") shapes of the two sample-serial kernels, hand-written HIP for gfx950 (wave64).
//
// A wave that has its SIMD to itself issues one instruction every ~4 cycles whatever the instruction
// is, so the latency of one chunk is (instructions per sample) x 4 cycles x 4096 samples: for a batch
// that cannot fill the chip (BASELINE configs 4 and 5: a few hundred to a few thousand chunks on 1,024
// SIMDs) the only way to get a chunk done sooner is to put FEWER instructions per sample on each wave.
// The recurrence cannot be cut along the sample axis, but it can be cut into stages that only feed
// forward:
//
//   decode   bits --[entropy parse: k, Rice code, run_avg]--> m --[LMS cascade]--> resLPC --[LPC synthesis]--> x
//            (VeloxEntropy.h:121-138, VeloxCore.h:146-161)     (VeloxCore.h:16-41, :150)   (:152-155)
//   encode   resLPC --[LMS cascade]--> final --[k, Rice code, bitpack, run_avg]--> bits
//
// Each stage is one WAVE of a workgroup (lane = chunk resp. stream, as in the throughput shapes), the
// stages run concurrently on the CU's four SIMDs and hand tiles of 16 samples to one another in place
// through the stream scratch in HBM/L2; progress counters live in LDS.  A producer never waits for a
// consumer (no back-pressure, hence no cycle to deadlock on); a consumer polls with s_sleep.
//
// Decode, one workgroup of five waves per 64 chunks:
//     wave P   parses stream L's codes, then — knowing now where R starts (VeloxCore.h:355-356) — stream R's
//     wave ML / MR   LMS cascade of stream L / R          wave XL / XR   LPC synthesis of stream L / R
// The chunk's latency is  parse(L) + LMS(R)  instead of  parse(L) + parse(R) + LMS + LPC (stream waves) —
// about 26 + 60 wave instructions per sample step on the critical path instead of 167.
// Everything that leaves the 32-bit envelope flags the chunk CSF_WIDE exactly as k_decode_streams does
// and k_decode_serial<true> redoes it; k_decode_finalize is unchanged.
#include <type_traits>
#include "vlx_device.h"

// Diagnostic stamps (tools/pipe_stamps.py): when the hook below has set a device buffer, lane 0 of every
// wave of workgroup 0 leaves s_memtime readings there: [role][0] start, [1] end, [2] the parser's end of
// stream L.  The buffer is read by no kernel; without it (always, outside that tool) nothing is stamped.
static uint64_t *g_dp_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) void vlx_debug_set_stamp_buffer(void *d_buf) { g_dp_stamps = (uint64_t *)d_buf; }

#define DP_TILE    16                   // samples per hand-off tile = one 64-byte line per lane
#define DP_RING    64                   // dwords of compressed bits buffered per lane of the parser wave: with requests two refill intervals
                                        // deep, 32 leave 13 readable at a refill point, and a few codes of 32+ bits in one interval ran it dry
                                        // (the chunk is then redone by the careful kernel: correct, and 7 ms for cfg4)
#define DPX_SLOTS  4                    // tiles in the LDS ring between an LMS wave and its LPC wave
#define DPX_ROW    20                   // dwords per stream and tile there: 16 samples + 4 of padding = conflict-free 16-byte accesses
#define DP_WAVES   5
#define DP_ROLE_XL 0                    // (the two light LPC waves are the first and the last of the workgroup: with
#define DP_ROLE_P  1                    //  five waves on four SIMDs, waves 0 and 4 are the ones likely to share one)
#define DP_ROLE_ML 2
#define DP_ROLE_MR 3
#define DP_ROLE_XR 4
#define DP_ROLE_MR1 5                   // MRS = 2: the second LMS wave of stream R (chunks 32-63 of the group)

// DP_PROFILE (a measurement build, tools/pipe_stamps.py --waits): cycles every wave of workgroup 0 spends waiting
// for its producer (dp_wait), draining its memory operations before a publish (dp_publish) and waiting for the
// next tile's loads; never defined in the shipped library.
#ifdef DP_PROFILE
#define DP_PROF_DECL uint64_t pf_wait = 0, pf_pub = 0, pf_load = 0
#define DP_PROF_T0 const uint64_t pf_t0 = __builtin_amdgcn_s_memtime()
#define DP_PROF_ADD(acc) acc += __builtin_amdgcn_s_memtime() - pf_t0
#else
#define DP_PROF_DECL
#define DP_PROF_T0
#define DP_PROF_ADD(acc)
#endif

// ---- progress counters in LDS ------------------------------------------------------------------
// Producer side: a tile is announced half a tile AFTER its stores were issued (in the middle of the next
// tile, or after the loop): by then they have long completed, so the s_waitcnt that makes sure of it
// (every vector-memory operation of the wave done = the tile is in this CU's L1/L2) does not stall.
// Consumer side: poll, then plain loads — all waves of a workgroup share the CU's vector L1, which is
// write-through and updated by stores.
__device__ __forceinline__ void dp_publish(uint32_t *ctr, uint32_t value, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(ctr, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The parser's publish inside a tile: exactly two vector-memory instructions (the loads of the refill point at the
// head of the tile, PipeWinT<true>) were issued after the stores it announces; the counter retires in order, so
// vmcnt(2) says the stores and everything older are done and leaves those two loads in flight.
__device__ __forceinline__ void dp_publish_behind2(uint32_t *ctr, uint32_t value, int lane)
{
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (lane == 0) __hip_atomic_store(ctr, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void dp_wait(uint32_t *ctr, uint32_t need)
{
    for (;;) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (v >= need) break;
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                // keeps the loads that follow behind the poll
}

// Counters for hand-overs through LDS (the LMS -> LPC ring here, the whole encode pipeline below): the poster's LDS
// accesses of the tile are done before the counter moves.
__device__ __forceinline__ void ep_wait(uint32_t *ctr, uint32_t need)
{
    for (;;) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int32_t)(v - need) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void ep_post(uint32_t *ctr, uint32_t value, int lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // this wave's LDS accesses of the tile are done
    if (lane == 0) __hip_atomic_store(ctr, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// NSUB = pipelines per workgroup: 1, 2 (waves A0 B0 A1 B1) or 4 (A0 B0 A1 B1 B2 A2 B3 A3).  A workgroup's wave w
// goes to SIMD w % 4, so two 2-wave workgroups on one CU put both LMS waves on SIMD 0 and both packers on SIMD 1
// and leave SIMDs 2 and 3 idle (measured: 22,500 streams as 352 workgroups of G = 64, 1.09 ms; as 176 workgroups
// of two pipelines, 0.95 ms = what 256 single pipelines take).  The launcher keeps the batch at one workgroup per
// CU as long as it can; with four pipelines every SIMD carries one LMS wave and one packer.
// (the bit window PipeWin and its sample decoder dp_sample live in vlx_device.h: k_decode_streams uses them too)

// (the predictor of the LMS waves — one, two or four lanes per stream — is VlxLmsT in vlx_device.h)
// How an LMS wave of these pipelines treats the streams whose error is zero (VlxLmsT's REPAIR), by lanes per stream.
// Measured on one box, all three policies for each (profiles/round3_pipe_repair_ab.txt): one and two lanes per stream, the
// out-of-line branch (encode 32-bit, 11,250 chunks: 0.607 ms against 0.647 in line and 0.708 with selects; decode 0.870
// against 0.917 and 0.895); four lanes per stream in the 16-stream pipelines, selects (decode, 64 chunks: 0.715 ms against
// 0.748 out of line and 0.793 in line).
#ifndef VLX_PIPE_REPAIR_1
#define VLX_PIPE_REPAIR_1 2
#endif
#ifndef VLX_PIPE_REPAIR_2
#define VLX_PIPE_REPAIR_2 2
#endif
#ifndef VLX_PIPE_REPAIR_4
#define VLX_PIPE_REPAIR_4 0
#endif
#define VLX_PIPE_REPAIR(LPS) ((LPS) == 1 ? VLX_PIPE_REPAIR_1 : (LPS) == 2 ? VLX_PIPE_REPAIR_2 : VLX_PIPE_REPAIR_4)

__device__ __forceinline__ void dp_store_tile(int32_t *dst, const uint32_t (&v)[DP_TILE])
{
#pragma unroll
    for (int q = 0; q < DP_TILE / 4; q++) vlx_gstore16((uint4 *)dst + q, make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]));
}
__device__ __forceinline__ void dp_load_tile(const int32_t *src, uint32_t (&v)[DP_TILE])
{
#pragma unroll
    for (int q = 0; q < DP_TILE / 4; q++) { const uint4 t = vlx_gload16((const uint4 *)src + q); v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}

// G = chunks per workgroup: 64 (one lane per stream in every stage) or 16 (the LMS stage spreads a stream over
// a quad of lanes, VlxLmsT<4>; the other stages leave 48 lanes idle — lanes are free when the batch is small).
// NSUB = pipelines per workgroup, 1 or 2.  A workgroup's wave w goes to SIMD w % 4: two 5-wave workgroups on one CU
// would put both parsers on one SIMD and both critical LMS waves of stream R on another.  With two pipelines (a, b)
// in one workgroup the ten waves are ordered  P_a P_b MR_a MR_b | ML_b ML_a XL_b XL_a | XR_b XR_a : each MR shares
// its SIMD only with the other pipeline's light XL, each parser (which has priority) with an ML and an XR.
// MRS = LMS waves of stream R, 1 or 2 (G = 64, NSUB = 1 only).  The chunk's latency is parse(L) + LMS(R): with two
// waves for LMS(R), 32 chunks each on pairs of lanes (six taps per lane, as G = 32 has them everywhere), the critical
// stage is a third shorter while the workgroup grows by one wave only — XL P ML MR0 | MR1 XR: MR0 has SIMD 3 to
// itself, MR1 shares SIMD 0 with the light XL.
template <int G, int NSUB, int MRS = 1>
__global__ __launch_bounds__(64 * (DP_WAVES + MRS - 1) * NSUB) void k_decode_pipe(const uint8_t *payload, uint64_t payload_bytes, DecWorkspace ws, uint64_t n_chunks, uint64_t *stamps)
{
    __shared__ uint32_t ring_s[NSUB][DP_RING * 64];
    __shared__ int32_t  s_cf_s[NSUB][2][VLX_LPC_ORDER][64];     // quantised LPC coefficients of stream s, lane-contiguous
    __shared__ uint32_t s_meta_s[NSUB][2][64];                  // frames | lpc_shift << 16, 0 = nothing to decode (silent, raw, bad, no chunk)
    static_assert(MRS == 1 || (MRS == 2 && G == 64 && NSUB == 1), "two LMS waves for stream R: G = 64, one pipeline per workgroup");
    __shared__ uint32_t s_prog_s[NSUB][5];                      // [s] 1 + tiles of stream s parsed (0 = header not parsed yet); [2 + s] likewise, LMS done ([4]: MR1's)
    __shared__ uint32_t s_bad_s[NSUB][64];
    __shared__ __attribute__((aligned(16))) int32_t s_xq_s[NSUB][2][DPX_SLOTS][64 * DPX_ROW];   // LMS -> LPC hand-over of stream s: [slot][stream of the group][sample]
    __shared__ uint32_t s_xdone_s[NSUB][2];                        // tiles of stream s the LPC wave has taken out of s_xq
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // wave -> (role, pipeline): one pipeline XL P ML MR XR; two, see above (4 bits per wave: role | sub << 3)
    constexpr uint64_t WMAP2 = (uint64_t)(DP_ROLE_P) | (uint64_t)(DP_ROLE_P | 8) << 4 | (uint64_t)(DP_ROLE_MR) << 8 | (uint64_t)(DP_ROLE_MR | 8) << 12 |
                               (uint64_t)(DP_ROLE_ML | 8) << 16 | (uint64_t)(DP_ROLE_ML) << 20 | (uint64_t)(DP_ROLE_XL | 8) << 24 | (uint64_t)(DP_ROLE_XL) << 28 |
                               (uint64_t)(DP_ROLE_XR | 8) << 32 | (uint64_t)(DP_ROLE_XR) << 36;
    const int wcode = MRS == 2 ? (wv == 4 ? DP_ROLE_MR1 : (wv == 5 ? DP_ROLE_XR : wv)) : (NSUB == 1 ? wv : (int)((WMAP2 >> (4 * wv)) & 15u));
    const int role = wcode & 7, sub = wcode >> 3;
    uint32_t *const ring = ring_s[sub];
    int32_t (*const s_cf)[VLX_LPC_ORDER][64] = s_cf_s[sub];
    uint32_t (*const s_meta)[64] = s_meta_s[sub];
    uint32_t *const s_prog = s_prog_s[sub];
    uint32_t *const s_bad = s_bad_s[sub];
    int32_t (*const s_xq)[DPX_SLOTS][64 * DPX_ROW] = s_xq_s[sub];
    uint32_t *const s_xdone = s_xdone_s[sub];
    const uint64_t c0 = ((uint64_t)blockIdx.x * NSUB + sub) * G;
    const uint64_t ci = c0 + lane;
    const bool mine = lane < G && ci < n_chunks;
    if (role == DP_ROLE_P) { if (lane < 5) s_prog[lane] = 0; if (lane < 2) s_xdone[lane] = 0; s_bad[lane] = 0; }
    __syncthreads();
    const bool stamp = stamps && blockIdx.x == 0 && sub == 0 && lane == 0;
    if (stamp) stamps[role * 4 + 0] = __builtin_amdgcn_s_memtime();
    // stream-major scratch [chunk][stream][sample]; the rows of a workgroup's G lanes exist even where the batch
    // ends (the scratch holds whole groups of 64 chunks), so stores are only guarded by lane < G
    int32_t *const row0 = ws.scratch + (c0 + (lane < G ? lane : 0)) * 2 * VLX_STREAM_LEN;
    uint32_t cs_flags = 0, cs_shifts = 0, cs_end[2] = {0, 0};
    DP_PROF_DECL;

    if (role == DP_ROLE_P) {
        __builtin_amdgcn_s_setprio(3);                                    // the critical wave: wins issue arbitration on a shared SIMD
        const uint64_t addr = (uint64_t)payload;
        const uint4 *abase = (const uint4 *)(addr & ~15ull);
        const uint64_t dbits = (addr & 15ull) * 8ull;
        const uint32_t last_g = (uint32_t)((payload_bytes + (addr & 15ull) - 1ull) >> 4);
        vlx_chunk ch; ch.bit_off = 0; ch.n_bits = 0; ch.frames = 0; ch.high_res = 0;
        if (mine) ch = ws.chunks[ci];
        PipeWinT<true, DP_RING> w; w.init(ring + lane, abase, last_g, ch.bit_off + dbits);
        const uint32_t mode = w.take(1), use_ms = w.take(1);              // VeloxCore.h:347,353
        bool bad = mine && ch.n_bits < 2;                                 // even the mode bits lie past the end
        uint32_t pos = 0;                                                 // bit position inside the chunk at which the window was initialised
        cs_flags = CSF_VALID | (mode ? CSF_MODE : 0u) | (use_ms ? CSF_MS : 0u);
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            bool act = mine && mode && !bad;
            uint32_t silent = 0, shift_lsb = 0, lshift = 0; int32_t cf[VLX_LPC_ORDER];
#pragma unroll
            for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = 0;
            if (act) {
                silent = w.take(1);                                       // :132
                if (!silent) {
                    shift_lsb = w.take(5); lshift = w.take(5);            // :135,137
#pragma unroll
                    for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = (int32_t)(int16_t)w.take(16);   // ReadS(16), :139
                }
            }
            if (mine && mode && silent) cs_flags |= (s ? CSF_SILENT_B : CSF_SILENT_A);
            cs_shifts |= shift_lsb << (8 * s);
            act = act && !silent;
            const int frames = act ? (int)ch.frames : 0;
            const uint32_t used_hdr = w.used();                           // a lane without codes to parse stops here, whatever it runs along on
#pragma unroll
            for (int j = 0; j < VLX_LPC_ORDER; j++) s_cf[s][j][lane] = cf[j];
            s_meta[s][lane] = (uint32_t)frames | (lshift << 16);
            const int fmax = vlx_wave_max(frames);
            const int n_tiles = (fmax + DP_TILE - 1) / DP_TILE;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the header words are in LDS before the counter says so
            if (lane == 0) __hip_atomic_store(&s_prog[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // lanes with nothing to decode run along on whatever bits they see (all reads are clamped to the
            // payload, their scratch rows exist, nobody reads what they store): the wave then stays on the
            // unchecked copy of the loop unless a real chunk is shorter than the others (a track's tail)
            const int flane = act ? frames : fmax;
            uint32_t run_avg = 512;                                       // VeloxCore.h:141
            int32_t *dst = row0 + s * VLX_STREAM_LEN;
            // One half tile = the DS_TILE samples between two refill points.  A scalar branch on a vector
            // compare costs a lone wave about 30 cycles, a quarter of a sample: so the samples are parsed
            // WITHOUT the test for codes the fast form cannot take (32 bits and more), only their lengths
            // are remembered; one test per half tile then sends the lanes that met such a code back to the
            // half tile's start (window and run_avg are six registers; the ring does not change between
            // refill points) to parse it again through the side path.
            auto half = [&](auto checked, auto phase, int t0, uint32_t (&mv)[DP_TILE]) {
                constexpr int PH = decltype(phase)::value, j0 = PH * DS_TILE;
                w.template refill_point<PH>();
                const uint32_t s_w0 = w.w0, s_w1 = w.w1, s_qn = w.qn, s_bo = w.bo, s_d = w.d, s_ra = run_avg;
                uint32_t worst = 0;
#pragma unroll
                for (int j = 0; j < DS_TILE; j++) {
                    uint32_t m = 0;
                    if (!decltype(checked)::value || t0 + j0 + j < flane) {
                        const uint32_t k = 31u - (uint32_t)__builtin_clz(run_avg);               // :146
                        const uint32_t lo = w.next32();
                        const uint32_t q = (uint32_t)__builtin_ctz(~lo | 0x80000000u);           // ones before the first 0
                        const uint32_t q1 = q + 1u, len = q1 + k;
                        m = (q << k) | __builtin_amdgcn_ubfe(lo, q1, k);
                        worst = worst > len ? worst : len;
                        w.advance(len);                                                          // (nonsense beyond 31 bits, but harmless: redone below)
                        run_avg = run_avg - (run_avg >> 3) + (m >> 3);                             // :159-161 (never reaches 0 from >= 1)
                    }
                    mv[j0 + j] = m;
                }
                if (__builtin_expect(__any(worst > 31u), 0)) {
                    if (worst > 31u) {
                        w.w0 = s_w0; w.w1 = s_w1; w.qn = s_qn; w.bo = s_bo; w.d = s_d; run_avg = s_ra;
#pragma unroll 1
                        for (int j = 0; j < DS_TILE; j++) {
                            uint32_t m = 0;
                            if (t0 + j0 + j < flane) {
                                const uint32_t k = 31u - (uint32_t)__builtin_clz(run_avg);
                                bool b = false;
                                m = dp_sample(w, k, b);
                                if (b) bad = bad || act;
                                run_avg = run_avg - (run_avg >> 3) + (m >> 3);
                            }
                            mv[j0 + j] = m;
                        }
                    }
                }
            };
            auto tile = [&](auto checked, int t0) {
                uint32_t mv[DP_TILE];
                half(checked, std::integral_constant<int, 0>{}, t0, mv);
                // (announce BEFORE the next refill point issues new loads: the wait inside then covers only
                // operations that are at least half a tile old)
                { DP_PROF_T0; dp_publish_behind2(&s_prog[s], 1u + (uint32_t)(t0 / DP_TILE), lane); DP_PROF_ADD(pf_pub); }   // the tiles BEFORE this one
                half(checked, std::integral_constant<int, 1>{}, t0, mv);
                if (G == 64 || lane < G) dp_store_tile(dst + t0, mv);
            };
            const int fmin = vlx_wave_min(flane) & ~(DP_TILE - 1);
            int t0 = 0;
            for (; t0 < fmin; t0 += DP_TILE) tile(std::false_type{}, t0);
            for (; t0 < fmax; t0 += DP_TILE) tile(std::true_type{}, t0);
            dp_publish(&s_prog[s], 1u + (uint32_t)n_tiles, lane);
            w.refill_point();
            if (w.under && act) bad = true;
            const uint32_t endpos = pos + (act ? w.used() : used_hdr);                           // where this stream's low bytes start
            cs_end[s] = endpos;
            if (act && (uint64_t)endpos + (ch.high_res ? 8ull * ch.frames : 0ull) > ch.n_bits) bad = true;   // ran past the chunk end
            if (mine && mode && silent && (uint64_t)endpos > ch.n_bits) bad = true;
            if (s == 0) {
                if (stamp) stamps[role * 4 + 2] = __builtin_amdgcn_s_memtime();
                pos = endpos + ((act && ch.high_res) ? 8u * ch.frames : 0u);                     // low bytes sit between L and R (:166-170)
                w.init(ring + lane, abase, last_g, ch.bit_off + dbits + pos);
            }
        }
        if (bad) atomicOr(&s_bad[lane], 1u);
    } else if (role == DP_ROLE_ML || role == DP_ROLE_MR || role == DP_ROLE_MR1) {
      // ---- LMS cascade of one stream: m -> (int32)resLPC (VeloxCore.h:147-150, :16-41).  lps_c = lanes per stream: 1, 2
      //      (six taps each) or 4 (three taps each); a wave carries 64 / LPS streams, from `sbase` on
      auto m_wave = [&](auto lps_c, const int s, const int sbase, uint32_t *const my_prog) {
        constexpr int LPS = decltype(lps_c)::value;
        const int sl = sbase + lane / LPS;                                // the stream (= chunk of this workgroup) this lane works on
        dp_wait(&s_prog[s], 1u);
        const int frames = (int)(s_meta[s][sl] & 0xFFFFu);
        // (every LMS wave of a stream runs as many tiles as the longest chunk of the whole group has: the LPC wave waits for all of them)
        const int fmax = vlx_wave_max(64 / LPS < G ? max(frames, (int)(s_meta[s][(sl + 64 / LPS) & 63] & 0xFFFFu)) : frames);
        const int n_tiles = (fmax + DP_TILE - 1) / DP_TILE;
        const bool act = frames > 0;
        const int flane = act ? frames : fmax;
        int32_t ovf = 0;
        VlxLmsT<LPS, VLX_PIPE_REPAIR(LPS)> lms; lms.reset(lane % LPS);
        const bool storer = lane % LPS == 0;                              // every lane of a group holds the same resLPC
        int32_t *row = ws.scratch + ((c0 + sl) * 2 + s) * VLX_STREAM_LEN;
        uint32_t cur[DP_TILE], nxt[DP_TILE];
        // (the first tile is waited for before the loop: while its registers count as "maybe still in flight" at the
        // loop head, hipcc puts an s_waitcnt in front of every tile's first sample that in steady state waits for
        // the stores and the first load issued a moment before — a memory round trip per tile)
        if (n_tiles) { dp_wait(&s_prog[s], 2u); dp_load_tile(row, nxt); }
#pragma unroll
        for (int j = 0; j < DP_TILE; j++) { cur[j] = nxt[j]; asm volatile("" : "+v"(cur[j])); }   // a use hipcc sees: it waits HERE, once
        uint32_t x_known = 0;                                             // tiles the LPC wave is known to have taken
        auto tile = [&](auto checked, int t0) {
            uint32_t out[DP_TILE];
            static_assert(DP_TILE % 16 == 0, "whole turns of the predictor's ring per tile");
            vlx_static_for<DP_TILE>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if (j == DP_TILE / 2) ep_post(my_prog, 1u + (uint32_t)(t0 / DP_TILE), lane);   // the tiles BEFORE this one: their LDS writes are long done
                uint32_t r = cur[j];
                if (!decltype(checked)::value || t0 + j < flane) {
                    const uint32_t m = cur[j];
                    const int32_t fin = (int32_t)((m >> 1) ^ (0u - (m & 1u)));                   // DeZigZag, VeloxEntropy.h:96
                    int32_t pred;
                    if constexpr (LPS == 1) pred = lms.template predict2<j>(); else pred = lms.template predict<j>();
                    const int32_t res = (int32_t)((uint32_t)fin + (uint32_t)pred);               // (int32)resLPC, :150 / :27
                    ovf |= ~(fin ^ pred) & (fin ^ res);                                          // sign bit: fin + pred left int32 (the careful kernel's business)
                    lms.template update<j>(fin, res);
                    r = (uint32_t)res;
                }
                out[j] = r;
            });
            // hand the tile to the LPC wave through LDS; it has to have taken tile i - DPX_SLOTS out before tile i
            // goes in (it is the lighter stage and follows within a tile: this rarely waits, and rarely even polls)
            const uint32_t ti = (uint32_t)(t0 / DP_TILE);
            if (x_known + DPX_SLOTS <= ti) {                              // (x_known: what the counter said last time: no poll while that still covers the slot)
                DP_PROF_T0;
                for (;;) {
                    x_known = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&s_xdone[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (x_known + DPX_SLOTS > ti) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                DP_PROF_ADD(pf_pub);
            }
            if (storer) {
                uint4 *dst = (uint4 *)&s_xq[s][ti % DPX_SLOTS][sl * DPX_ROW];
#pragma unroll
                for (int q = 0; q < DP_TILE / 4; q++) dst[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
            }
        };
        const int fmin = vlx_wave_min(flane) & ~(DP_TILE - 1);
        for (int i = 0; i < n_tiles; i++) {
            const int t0 = i * DP_TILE;
            // the next tile's loads are issued every time (after the last tile: this tile's again, unused), so that
            // nxt is always "loaded a tile ago" when it is copied and nothing of cur is in flight at the loop head
            const bool more = i + 1 < n_tiles;
            if (more) { DP_PROF_T0; dp_wait(&s_prog[s], 2u + (uint32_t)(i + 1)); DP_PROF_ADD(pf_wait); }
            dp_load_tile(row + (more ? t0 + DP_TILE : t0), nxt);
            if (t0 < fmin) tile(std::false_type{}, t0); else tile(std::true_type{}, t0);
#pragma unroll
            for (int j = 0; j < DP_TILE; j++) cur[j] = nxt[j];
        }
        ep_post(my_prog, 1u + (uint32_t)n_tiles, lane);
        if (ovf < 0 && act) atomicOr(&s_bad[sl], 1u);
      };
      if (MRS == 2 && role != DP_ROLE_ML) m_wave(std::integral_constant<int, MRS == 2 ? 2 : 1>{}, 1, role == DP_ROLE_MR1 ? 32 : 0, &s_prog[role == DP_ROLE_MR1 ? 4 : 3]);
      else m_wave(std::integral_constant<int, 64 / G>{}, role == DP_ROLE_ML ? 0 : 1, 0, &s_prog[role == DP_ROLE_ML ? 2 : 3]);
    } else {
        // ---- quantised-LPC synthesis of one stream: resLPC -> x, in place (VeloxCore.h:152-157)
        const int s = role == DP_ROLE_XR ? 1 : 0;
        dp_wait(&s_prog[s], 1u);
        const uint32_t meta = lane < G ? s_meta[s][lane] : 0u;
        const int frames = (int)(meta & 0xFFFFu);
        const uint32_t lshift = meta >> 16;
        int32_t cf[VLX_LPC_ORDER];
#pragma unroll
        for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = s_cf[s][j][lane];
        const int fmax = vlx_wave_max(frames);
        const int n_tiles = (fmax + DP_TILE - 1) / DP_TILE;
        const bool act = frames > 0;
        const int flane = act ? frames : fmax;
        bool bad = false;
        int32_t hist[VLX_LPC_ORDER];
#pragma unroll
        for (int j = 0; j < VLX_LPC_ORDER; j++) hist[j] = 0;
        int32_t *row = row0 + s * VLX_STREAM_LEN;
        auto tile = [&](auto checked, int t0) {
            const uint32_t ti = (uint32_t)(t0 / DP_TILE);
            uint32_t cur[DP_TILE], out[DP_TILE];
            { DP_PROF_T0; ep_wait(&s_prog[2 + s], 2u + ti); if (MRS == 2 && s == 1) ep_wait(&s_prog[4], 2u + ti); DP_PROF_ADD(pf_wait); }
            {
                const uint4 *src = (const uint4 *)&s_xq[s][ti % DPX_SLOTS][lane * DPX_ROW];
#pragma unroll
                for (int q = 0; q < DP_TILE / 4; q++) { const uint4 v = src[q]; cur[4 * q] = v.x; cur[4 * q + 1] = v.y; cur[4 * q + 2] = v.z; cur[4 * q + 3] = v.w; }
            }
            ep_post(&s_xdone[s], ti + 1u, lane);                          // (waits for the reads: the slot is free again)
#pragma unroll
            for (int j = 0; j < DP_TILE; j++) {
                int32_t x = 0;
                if (!decltype(checked)::value || t0 + j < flane) {
                    const int32_t res = (int32_t)cur[j];
                    int64_t sum = 0, sum1 = 0;
#pragma unroll
                    for (int jj = 0; jj < VLX_LPC_ORDER; jj += 2) { sum += (int64_t)cf[jj] * hist[jj]; sum1 += (int64_t)cf[jj + 1] * hist[jj + 1]; }   // :152-154
                    sum += sum1;
                    const int64_t x64 = (int64_t)res + (sum >> lshift);                          // :155
                    x = (int32_t)x64;
                    if (x64 != (int64_t)x) bad = true;
                }
                out[j] = (uint32_t)x;
#pragma unroll
                for (int jj = VLX_LPC_ORDER - 1; jj > 0; jj--) hist[jj] = hist[jj - 1];
                hist[0] = x;
            }
            if (G == 64 || lane < G) dp_store_tile(row + t0, out);
        };
        const int fmin = vlx_wave_min(flane) & ~(DP_TILE - 1);
        for (int i = 0; i < n_tiles; i++) {
            const int t0 = i * DP_TILE;
            if (t0 < fmin) tile(std::false_type{}, t0); else tile(std::true_type{}, t0);
        }
        if (bad && act) atomicOr(&s_bad[lane], 1u);
    }
    if (stamp) stamps[role * 4 + 1] = __builtin_amdgcn_s_memtime();
#ifdef DP_PROFILE
    if (stamp) { stamps[role * 4 + 3] = pf_wait; stamps[20 + role] = pf_pub; stamps[25 + role] = pf_load; }
#endif
    __syncthreads();
    // ---- per-chunk state for k_decode_finalize (and the redo flag for k_decode_serial<true>)
    if (role == DP_ROLE_P && mine) {
        uint32_t *cs = ws.chunk_state + ci * CS_WORDS;
        if (s_bad[lane]) cs_flags |= CSF_WIDE;
        cs[0] = cs_flags; cs[1] = cs_shifts; cs[2] = cs_end[0]; cs[3] = cs_end[1];
    }
}

// group = 16, 32 or 64 chunks per workgroup (0: by batch size, see VLX_DPIPE_G16_MAX / _G32_MAX)
hipError_t vlx_launch_decode_pipe(hipStream_t st, const uint8_t *payload, uint64_t payload_bytes, const DecWorkspace &ws, uint64_t n_chunks, int group)
{
    if (!n_chunks) return hipSuccess;
    if (group == 0) group = n_chunks <= VLX_DPIPE_G16_MAX ? 16 : (n_chunks <= VLX_DPIPE_G32_MAX ? 32 : 64);
    const uint64_t groups = (n_chunks + (unsigned)group - 1) / (unsigned)group;
    const unsigned nsub = groups <= 256 ? 1u : 2u;                         // more pipelines than CUs: two per workgroup
    const dim3 grid((unsigned)((groups + nsub - 1) / nsub)), block(64 * DP_WAVES * nsub);
#define VLX_DP_LAUNCH(G_) do { if (nsub == 1) k_decode_pipe<G_, 1><<<grid, block, 0, st>>>(payload, payload_bytes, ws, n_chunks, g_dp_stamps); \
        else k_decode_pipe<G_, 2><<<grid, block, 0, st>>>(payload, payload_bytes, ws, n_chunks, g_dp_stamps); } while (0)
    if (group == 16) VLX_DP_LAUNCH(16); else if (group == 32) VLX_DP_LAUNCH(32);
    else if (nsub == 1) k_decode_pipe<64, 1, 2><<<grid, dim3(64 * (DP_WAVES + 1)), 0, st>>>(payload, payload_bytes, ws, n_chunks, g_dp_stamps);
    else VLX_DP_LAUNCH(64);
#undef VLX_DP_LAUNCH
    return hipGetLastError();
}

// ================================================================================================
// Encode, small batches: one workgroup of TWO waves per G streams.
//     wave A   LMS cascade: resLPC -> final = resLPC - predNeural (VeloxCore.h:106-107, :16-41)
//     wave B   k from run_avg, Rice code, bit packing into the stream's staging slot, run_avg
//              (VeloxCore.h:109-121; VeloxEntropy.h:98-119, :9-43)
// B is the lighter stage, so it follows A at a distance of one tile; the tiles go through a small ring
// in LDS ([slot][sample][lane]: conflict-free dword accesses), with a counter in each direction (A must
// not overwrite a slot B has not read yet).  Same results as k_encode_serial: streams whose 64-bit
// difference leaves int32 are flagged SF_WIDE for k_encode_wide.
#define EP_SLOTS 4
#define EP_ROW   20                     // dwords per stream and tile in the ring (as DPX_ROW)
#define ES_TILE 32
#define ES_STRIDE 36                    // as in k_encode_serial: 16-B aligned rows, conflict-free ds_read_b128

// AS = LMS waves per pipeline, 1 or 4 (G = 64, NSUB <= 2): the packer is the lighter stage and serves 64 streams with one
// lane each, so the LMS stage of those 64 streams can spread over four waves of 16 streams on quads of lanes (three
// taps per lane) — the LMS of G = 16 behind a full packer wave.  Waves: A0..A3 of pipeline a, (A0..A3 of b,) B_a (, B_b):
// every SIMD carries one LMS wave of each pipeline.
template <int G, int NSUB, int AS = 1>
__global__ __launch_bounds__(64 * (AS + 1) * NSUB) void k_encode_pipe(EncWorkspace ws, uint64_t n_streams)
{
    static_assert(AS == 1 || ((AS == 2 || AS == 4) && G == 64 && NSUB <= 2), "several LMS waves per pipeline: G = 64, at most two pipelines per workgroup");
    __shared__ __attribute__((aligned(16))) int32_t tile_s[NSUB][64 * ES_STRIDE];
    __shared__ __attribute__((aligned(16))) int32_t fifo_s[NSUB][EP_SLOTS][64 * EP_ROW];   // [slot][stream][sample]: rows of 16 + 4 dwords, conflict-free 16-byte accesses
    __shared__ uint32_t s_prog_s[NSUB][AS + 1];          // [k] tiles LMS wave k has written, [AS] tiles B has read
    __shared__ uint32_t s_redo_s[NSUB][64];
    constexpr int GA = G / AS;                            // streams per LMS wave
    constexpr int LPS = 64 / GA;                          // lanes per stream in an LMS wave
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int role = AS == 1 ? (wv < 4 ? (wv & 1) : 1 - (wv & 1)) : (wv < AS * NSUB ? 0 : 1);
    const int sub = AS == 1 ? (wv >> 1) : (wv < AS * NSUB ? wv / AS : wv - AS * NSUB);
    const int ak = AS == 1 ? 0 : wv % AS;                 // which LMS wave of its pipeline (role 0)
    const int a_base = ak * GA;                           // its first stream / tile row / fifo column
    int32_t *const tile = tile_s[sub];
    int32_t (*const fifo)[64 * EP_ROW] = fifo_s[sub];
    uint32_t *const s_prog = s_prog_s[sub];
    uint32_t *const s_redo = s_redo_s[sub];
    const uint64_t s0 = ((uint64_t)blockIdx.x * NSUB + sub) * G;
    if (lane <= AS && role == 1) s_prog[lane] = 0;
    if (role == 1) s_redo[lane] = 0;
    __syncthreads();
    // what both waves need to agree on: the number of tiles = from the longest stream of the group
    const int pl = lane < G ? lane : 0;
    const uint64_t ps = s0 + pl;
    int pn = 0; uint32_t pflags = SF_SILENT;
    if (lane < G && ps < n_streams) { pn = ws.params[ps].n; pflags = ws.params[ps].flags; }
    const bool pactive = !(pflags & (SF_SILENT | SF_WIDE));
    const int nmax = vlx_wave_max(pactive ? pn : 0);
    const int n_tiles = (nmax + DP_TILE - 1) / DP_TILE;
    // wave B's bit writer (BitStreamWriter, VeloxEntropy.h:9-43, as in k_encode_serial): declared here because its
    // last word is written after the barrier at the end, once wave A's verdict on the stream is known
    char *const wbase = (char *)(ws.staging + s0 * VLX_SLOT_DWORDS);
    const uint32_t bstart = (uint32_t)pl * (VLX_SLOT_DWORDS * 4u);
    const uint32_t bend = bstart + (VLX_SLOT_DWORDS - 2) * 4u;
    uint32_t boff = bstart, lo = 0, nacc = 0;
    const bool valid = lane < G && ps < n_streams;

    if (role == 0) {
        // ---------------------------------------------------------------- wave A: LMS
        const int sl = lane / LPS;                                        // stream of this wave; a_base + sl in the pipeline
        const uint64_t st = s0 + a_base + sl;
        int n = 0; uint32_t flags = SF_SILENT;
        if (sl < GA && st < n_streams) { n = ws.params[st].n; flags = ws.params[st].flags; }
        const bool active = !(flags & (SF_SILENT | SF_WIDE));
        const int n_eff = active ? n : 0;
        int32_t ovf = 0;
        VlxLmsT<LPS, (AS == 4 ? 2 : VLX_PIPE_REPAIR(LPS))> lms; lms.reset(lane % LPS);   // (four quad-lane waves of one pipeline share SIMDs with its packer: the branch, not the selects)
        // residual tiles: two int16 planes per stream (k_analyze), low halves always, high halves only for streams
        // without SF_FAST16.  Loader: lane -> (row, 8 samples), 16 B per plane and load, as in k_encode_serial.
        const uint64_t hi_mask = __ballot(active && !(flags & SF_FAST16) && lane % LPS == 0);   // per loader ROW below
        constexpr int ROWS = GA, RLOADS = ROWS / 16;
        const int lrow = lane >> 2, lcol = (lane & 3) * 8;
        uint4 nlo[RLOADS], nhi[RLOADS];
        // (ballot bit LPS*row belongs to row `row`)
        auto row_has_hi = [&](int row) { return ((hi_mask >> (LPS * row)) & 1ull) != 0; };
        auto issue = [&](int t0) {
#pragma unroll
            for (int r = 0; r < RLOADS; r++) {
                const int row = lrow + 16 * r;
                const uint64_t rs = s0 + a_base + row;
                const bool ok = rs < n_streams && t0 < nmax;
                const int16_t *src = (const int16_t *)(ws.res + rs * VLX_STREAM_LEN) + t0 + lcol;
                nlo[r] = ok ? vlx_gload16(src) : make_uint4(0, 0, 0, 0);
                nhi[r] = (hi_mask && ok && row_has_hi(row)) ? vlx_gload16(src + VLX_STREAM_LEN) : make_uint4(0, 0, 0, 0);
            }
        };
        auto sext2 = [](uint32_t d, int32_t &a, int32_t &b) { a = (int32_t)(d << 16) >> 16; b = (int32_t)d >> 16; };
        auto join2 = [](uint32_t lo2, uint32_t hi2, int32_t &a, int32_t &b) {
            a = (int32_t)__builtin_amdgcn_perm(hi2, lo2, 0x05040100u); b = (int32_t)__builtin_amdgcn_perm(hi2, lo2, 0x07060302u);
        };
        issue(0);
        uint32_t tile_no = 0;
        for (int t0 = 0; t0 < nmax; t0 += ES_TILE) {
            vlx_wave_lds_sync();
#pragma unroll
            for (int r = 0; r < RLOADS; r++) {
                const int row = lrow + 16 * r;
                int4 u, v;
                if (hi_mask && row_has_hi(row)) {
                    join2(nlo[r].x, nhi[r].x, u.x, u.y); join2(nlo[r].y, nhi[r].y, u.z, u.w);
                    join2(nlo[r].z, nhi[r].z, v.x, v.y); join2(nlo[r].w, nhi[r].w, v.z, v.w);
                } else {
                    sext2(nlo[r].x, u.x, u.y); sext2(nlo[r].y, u.z, u.w); sext2(nlo[r].z, v.x, v.y); sext2(nlo[r].w, v.z, v.w);
                }
                *(int4 *)&tile[(a_base + row) * ES_STRIDE + lcol] = u;
                *(int4 *)&tile[(a_base + row) * ES_STRIDE + lcol + 4] = v;
            }
            vlx_wave_lds_sync();
            issue(t0 + ES_TILE);
#pragma unroll 1
            for (int h0 = 0; h0 < ES_TILE && t0 + h0 < nmax; h0 += DP_TILE, tile_no++) {
                if (tile_no >= EP_SLOTS) ep_wait(&s_prog[AS], tile_no - EP_SLOTS + 1);         // the slot has been read
                int4 *const srow = (int4 *)&fifo[tile_no % EP_SLOTS][(a_base + sl) * EP_ROW];
                // (a copy of the tile without the per-sample length test for waves whose streams all still have the whole
                // tile to go: a compare and a branch cost a lone wave ~30 cycles each, a tenth of a sample step)
                static_assert(DP_TILE % 16 == 0, "whole turns of the predictor's ring per tile");
                auto run = [&](auto checked) {
                    vlx_static_for<DP_TILE / 4>([&](auto ic) __attribute__((always_inline)) {
                        constexpr int i = 4 * decltype(ic)::value;
                        const int4 rv = *(const int4 *)&tile[(a_base + sl) * ES_STRIDE + h0 + i];
                        const int32_t rr[4] = {rv.x, rv.y, rv.z, rv.w};
                        int32_t fo[4];
                        vlx_static_for<4>([&](auto ec) __attribute__((always_inline)) {
                            constexpr int e = decltype(ec)::value;
                            int32_t fin = 0;
                            if (!decltype(checked)::value || t0 + h0 + i + e < n_eff) {
                                const int32_t r = rr[e];
                                int32_t pred;
                                if constexpr (LPS == 1) pred = lms.template predict2<i + e>(); else pred = lms.template predict<i + e>();
                                fin = (int32_t)((uint32_t)r - (uint32_t)pred);                  // :107 when it fits int32
                                ovf |= (r ^ pred) & (r ^ fin);                                   // sign bit: the 64-bit difference left int32
                                lms.template update<i + e>(fin, r);                              // err = (int32)resLPC - pred = fin
                            }
                            fo[e] = fin;
                        });
                        if (lane % LPS == 0) srow[i / 4] = make_int4(fo[0], fo[1], fo[2], fo[3]);
                    });
                };
                if (__all(t0 + h0 + DP_TILE <= n_eff)) run(std::false_type{}); else run(std::true_type{});
                ep_post(&s_prog[ak], tile_no + 1, lane);
            }
        }
        if (ovf < 0 && active && sl < GA) atomicOr(&s_redo[a_base + sl], 1u);
    } else {
        // ---------------------------------------------------------------- wave B: Rice code + bit packing
        const uint64_t st = ps;
        const int n = pn; const uint32_t flags = pflags;
        // (k_encode_serial's store path — finished dwords through a per-lane LDS ring, 16 aligned bytes out at a time — was
        // tried here too: 0.68 -> 0.83 ms with 64 streams on the wave, 0.48 -> 0.65 with 16: a lone wave pays for every
        // extra instruction, and its scattered 4-byte stores wait for nobody)
        auto put = [&](uint32_t v, uint32_t nb) {                         // nb in [0,32], v < 2^nb, nacc < 32 (32-bit throughout, as k_encode_serial's)
            const uint32_t room = 31u - nacc;
            lo |= v << nacc; nacc += nb;
            if (nacc >= 32u) {
                *(uint32_t *)(wbase + (boff < bend ? boff : bend)) = lo;  // past the slot's end everything lands on its spare dword
                boff += 4u; lo = (v >> 1) >> room; nacc -= 32u;           // v >> (32 - nacc_before), also right for nacc_before = 0
            }
        };
        if (valid) {
            const StreamParams *P = &ws.params[st];
            if (flags & SF_SILENT) put(1u, 1);                            // VeloxCore.h:81
            else if (!(flags & SF_WIDE)) {
                put(0u, 1); put((uint32_t)P->shift_lsb & 31u, 5); put((uint32_t)P->lpc_shift & 31u, 5);     // :82-91
#pragma unroll
                for (int j = 0; j < VLX_LPC_ORDER; j++) put((uint32_t)P->coefs[j] & 0xFFFFu, 16);           // :92
            }
        }
        const bool active = valid && pactive;
        const int n_eff = active ? n : 0;
        uint32_t run_avg = 512;                                           // :96
        for (int i = 0; i < n_tiles; i++) {
#pragma unroll
            for (int k = 0; k < AS; k++) ep_wait(&s_prog[k], (uint32_t)i + 1);
            int32_t fv[DP_TILE];
            {
                const int4 *srow = (const int4 *)&fifo[i % EP_SLOTS][pl * EP_ROW];
#pragma unroll
                for (int q = 0; q < DP_TILE / 4; q++) { const int4 v = srow[q]; fv[4 * q] = v.x; fv[4 * q + 1] = v.y; fv[4 * q + 2] = v.z; fv[4 * q + 3] = v.w; }
            }
            ep_post(&s_prog[AS], (uint32_t)i + 1, lane);
#pragma unroll
            for (int j = 0; j < DP_TILE; j++) {
                if (i * DP_TILE + j < n_eff) {
                    const int32_t fin = fv[j];
                    const uint32_t m = ((uint32_t)fin << 1) ^ (uint32_t)(fin >> 31);                // ZigZag, VeloxEntropy.h:95
                    const uint32_t k = 31u - (uint32_t)__builtin_clz(run_avg);                      // :109-114
                    const uint32_t q = m >> k;
                    uint32_t len = q + k + 1u;
                    const uint32_t rem2 = (m & ((1u << k) - 1u)) << 1;                              // remainder above the 0 terminator
                    uint32_t code = ((1u << (q & 31u)) - 1u) | (rem2 << (q & 31u));                 // q ones, 0, k bits (VeloxEntropy.h:104-113)
                    // (codes of 33-64 bits as one more put of a 64-bit code instead of this piecewise path: measured, 5 % slower —
                    // the 64-bit shifts cost every sample more than the rare long code costs the wave)
                    const bool slow = len > 32u;
                    if (__builtin_expect(__any(slow), 0)) {
                        if (slow) {
                            if (q < 64u) {
                                uint32_t qq = q;
                                if (qq >= 32u) { put(0xFFFFFFFFu, 32); qq -= 32u; }
                                put((1u << qq) - 1u, qq + 1u);
                                put(m & ((1u << k) - 1u), k);
                            } else {                                                                // escape: 64 ones, 0, 40 bits (:114-118)
                                put(0xFFFFFFFFu, 32); put(0xFFFFFFFFu, 32); put(0u, 1); put(m, 32); put(0u, 8);
                            }
                            code = 0; len = 0;
                        }
                    }
                    put(code, len);
                    run_avg = run_avg - (run_avg >> 3) + (m >> 3);                                  // :120-121 (never reaches 0 from >= 1)
                }
            }
        }
    }
    __syncthreads();
    if (role == 1 && valid) {
        const bool redo = s_redo[pl] != 0;
        uint32_t bits = 0;
        if (!(pflags & SF_WIDE) && !redo) {
            if (boff < bend) { *(uint32_t *)(wbase + boff) = lo; *(uint32_t *)(wbase + boff + 4) = 0u; }
            bits = (boff - bstart) * 8u + nacc;
        }
        ws.stream_bits[ps] = bits;
        if (redo) ws.params[ps].flags = pflags | SF_WIDE;                 // k_encode_wide re-encodes it from the PCM
    }
}

hipError_t vlx_launch_encode_pipe(hipStream_t st, const EncWorkspace &ws, uint64_t n_streams, int group)
{
    if (!n_streams) return hipSuccess;
    if (group == 0) group = n_streams <= VLX_EPIPE_G16_MAX ? 16 : (n_streams <= VLX_EPIPE_G32_MAX ? 32 : 64);
    const uint64_t groups = (n_streams + (unsigned)group - 1) / (unsigned)group;
    const unsigned nsub = groups <= 256 ? 1u : (groups <= 512 ? 2u : 4u);  // more pipelines than CUs: several per workgroup
    const dim3 grid((unsigned)((groups + nsub - 1) / nsub)), block(128 * nsub);
#define VLX_EP_LAUNCH(G_) do { if (nsub == 1) k_encode_pipe<G_, 1><<<grid, block, 0, st>>>(ws, n_streams); \
        else if (nsub == 2) k_encode_pipe<G_, 2><<<grid, block, 0, st>>>(ws, n_streams); \
        else k_encode_pipe<G_, 4><<<grid, block, 0, st>>>(ws, n_streams); } while (0)
    if (group == 16) VLX_EP_LAUNCH(16); else if (group == 32) VLX_EP_LAUNCH(32);
    else if (groups <= 256) k_encode_pipe<64, 1, 4><<<dim3((unsigned)groups), dim3(64 * 5), 0, st>>>(ws, n_streams);
    else if (groups <= 512) k_encode_pipe<64, 2, 2><<<dim3((unsigned)((groups + 1) / 2)), dim3(64 * 6), 0, st>>>(ws, n_streams);
    else VLX_EP_LAUNCH(64);
#undef VLX_EP_LAUNCH
    return hipGetLastError();
}
