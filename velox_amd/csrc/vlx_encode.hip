// Encode kernels of the Velox frame-codec hot path, hand-written HIP for gfx950 (wave64).
//
//   k_hires_scan     whole-track "any |s| > 65536" reduction            (VeloxCore.h:232-236)
//   k_analyze        one workgroup per 8192-sample sub-block: de-interleave, SAD L/R vs M/S,
//                    M/S transform, high-res byte split, silence, LSB shift, autocorrelation,
//                    Levinson-Durbin, quantised-LPC residual            (VeloxCore.h:255-277, :69-105, :44-66)
//   k_encode_serial  one LANE per stream: sign-sign LMS cascade + adaptive Rice code, bit-packed
//                    into the stream's staging slot                      (VeloxCore.h:16-41, :106-122; VeloxEntropy.h:98-119)
//   k_chunk_layout / k_scan_u64 / k_track_layout
//                    chunk byte sizes, raw-fallback decision, offset table (VeloxCore.h:284-293, :299-302)
//   k_assemble       gathers every output dword of the final payload exactly once: u32 size prefixes,
//                    flag bits, code bits, low bytes, raw 40-bit blocks, all at the payload's
//                    bit offset                                          (VeloxCore.h:175-177, :299-305)
//
// No MFMA: nothing here is a dense contraction.  All integer work is exact; the only floating
// point is the reference's own double-precision LPC analysis, compiled without FMA contraction.
#include <atomic>
#include "vlx_device.h"

#pragma clang fp contract(off)

// ----------------------------------------------------------------------------------------------
// Vector loads of 16 consecutive (L,R) pairs per thread.
template <int FMT> struct Fmt { typedef int32_t S; static constexpr int BYTES = FMT / 8; };
template <> struct Fmt<VLX_FMT_I64> { typedef int64_t S; static constexpr int BYTES = 8; };
template <> struct Fmt<VLX_FMT_F32_SPLIT> { typedef int32_t S; static constexpr int BYTES = 4; };   // |mantissa| < 2^24
template <> struct Fmt<VLX_FMT_F32_D16> { typedef int64_t S; static constexpr int BYTES = 4; };     // what does not fit demotes to anything, INT64_MIN included
template <> struct Fmt<VLX_FMT_F32_D24> { typedef int64_t S; static constexpr int BYTES = 4; };

template <int FMT> __device__ __forceinline__ typename Fmt<FMT>::S vlx_extract(const uint32_t *raw, int k)
{
    if (FMT == VLX_FMT_S16) { uint32_t w = raw[k >> 1]; return (typename Fmt<FMT>::S)(int16_t)((k & 1) ? (w >> 16) : w); }
    if (FMT == VLX_FMT_S24) {
        int b = 3 * k, d = b >> 2, sh = (b & 3) * 8;
        uint32_t u = raw[d] >> sh;
        if (sh > 8) u |= raw[d + 1] << (32 - sh);
        return (typename Fmt<FMT>::S)((int32_t)(u << 8) >> 8);
    }
    if (FMT == VLX_FMT_S32) return (typename Fmt<FMT>::S)(int32_t)raw[k];
    if (FMT == VLX_FMT_F32_SPLIT || FMT == VLX_FMT_F32_D16 || FMT == VLX_FMT_F32_D24) return (typename Fmt<FMT>::S)vlx_from_f32<FMT>(raw[k]);
    return (typename Fmt<FMT>::S)(((uint64_t)raw[2 * k + 1] << 32) | raw[2 * k]);
}

template <int FMT>
__device__ __forceinline__ void vlx_load16pairs(const uint8_t *p, int valid,
                                                typename Fmt<FMT>::S (&L)[16], typename Fmt<FMT>::S (&R)[16])
{
    typedef typename Fmt<FMT>::S S;
    constexpr int NB = 32 * Fmt<FMT>::BYTES;
    if (valid >= 16) {
        uint32_t raw[NB / 4];
        const uint4 *v = (const uint4 *)p;
#pragma unroll
        for (int q = 0; q < NB / 16; q++) { uint4 t = v[q]; raw[4 * q] = t.x; raw[4 * q + 1] = t.y; raw[4 * q + 2] = t.z; raw[4 * q + 3] = t.w; }
#pragma unroll
        for (int i = 0; i < 16; i++) { L[i] = vlx_extract<FMT>(raw, 2 * i); R[i] = vlx_extract<FMT>(raw, 2 * i + 1); }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i < valid) { L[i] = (S)vlx_load_sample<FMT>(p, 2 * i); R[i] = (S)vlx_load_sample<FMT>(p, 2 * i + 1); }
            else { L[i] = 0; R[i] = 0; }
        }
    }
}

// Whole-wave variant for a wave whose 64 x 16 pairs are all valid: the wave's bytes are contiguous,
// so fetch them as fully coalesced 1-KiB rows, park them in the wave's LDS area and let every lane
// read its own run back.  (Lane-strided 16-B global loads of 96-B runs touch ~48 cache lines per
// instruction and thrash the 32-KiB L1: the high-res scan ran at 0.7 TB/s that way.)
template <int FMT>
__device__ __forceinline__ void vlx_load16pairs_wave(const uint8_t *wave_base, uint4 *lds, int lane,
                                                     typename Fmt<FMT>::S (&L)[16], typename Fmt<FMT>::S (&R)[16])
{
    constexpr int NB = 32 * Fmt<FMT>::BYTES, NQ = NB / 16;
    uint4 t[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) t[q] = ((const uint4 *)wave_base)[q * 64 + lane];
#pragma unroll
    for (int q = 0; q < NQ; q++) lds[q * 64 + lane] = t[q];
    vlx_wave_lds_sync();
    uint32_t raw[NB / 4];
#pragma unroll
    for (int q = 0; q < NQ; q++) { const uint4 u = lds[lane * NQ + q]; raw[4 * q] = u.x; raw[4 * q + 1] = u.y; raw[4 * q + 2] = u.z; raw[4 * q + 3] = u.w; }
#pragma unroll
    for (int i = 0; i < 16; i++) { L[i] = vlx_extract<FMT>(raw, 2 * i); R[i] = vlx_extract<FMT>(raw, 2 * i + 1); }
}

// ----------------------------------------------------------------------------------------------
// chunk -> track table, once per batch: a 12-deep dependent search per WORKGROUP (scalar-cache misses
// on a 4096-track table) in every later kernel costs more than the whole high-res scan.
__global__ __launch_bounds__(256) void k_chunk_track(EncWorkspace ws, uint32_t n_tracks, uint64_t n_chunks)
{
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < n_chunks) ws.chunk_track[c] = vlx_find_track(ws.tracks, n_tracks, c);
}

// What an encode batch needs in HBM before its first kernel, in ONE launch instead of two copy-engine transfers and two
// memsets (each a stream operation of its own, ~10 us of hand-over in front of batches that take 1-2 ms): the track
// table and the header bits, read from the context's PINNED HOST staging area over PCIe, and the cleared high-res
// flags and slot counter.
__global__ __launch_bounds__(256) void k_enc_setup(const uint8_t *h_stage, uint8_t *d_tracks, size_t tr_bytes, uint8_t *d_hdr, size_t hdr_off, size_t hdr_bytes,
                                                   uint32_t *zero_a, size_t n_a, uint32_t *zero_b)
{
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < tr_bytes / 4; i += step) ((uint32_t *)d_tracks)[i] = ((const uint32_t *)h_stage)[i];
    if (d_hdr) for (size_t i = i0; i < hdr_bytes; i += step) d_hdr[i] = h_stage[hdr_off + i];
    for (size_t i = i0; i < n_a; i += step) zero_a[i] = 0;
    if (zero_b && i0 == 0) *zero_b = 0;
}
hipError_t vlx_launch_enc_setup(hipStream_t st, const void *h_stage, void *d_tracks, size_t tr_bytes, void *d_hdr, size_t hdr_off, size_t hdr_bytes,
                                uint32_t *zero_a, size_t n_a, uint32_t *zero_b)
{
    const size_t work = std::max(std::max(tr_bytes / 4, d_hdr ? hdr_bytes : 0), n_a);
    const unsigned grid = (unsigned)std::min<size_t>((work + 255) / 256 + 1, 1024);
    k_enc_setup<<<dim3(grid), dim3(256), 0, st>>>((const uint8_t *)h_stage, (uint8_t *)d_tracks, tr_bytes, (uint8_t *)d_hdr, hdr_off, hdr_bytes, zero_a, n_a, zero_b);
    return hipGetLastError();
}

hipError_t vlx_launch_chunk_track(hipStream_t st, const EncWorkspace &ws, uint32_t n_tracks, uint64_t n_chunks)
{
    if (!n_chunks) return hipSuccess;
    k_chunk_track<<<dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st>>>(ws, n_tracks, n_chunks);
    return hipGetLastError();
}

// Corpora of short tracks (round 3): one workgroup per TRACK walks its chunks in order and leaves at the first loud one —
// the early return of the reference's loop (VeloxCore.h:232-236).  Nobody else looks at the track, so there is no flag
// to poll (the contention that made the per-chunk polling form cost 16 ms on cfg3, see below); a quiet track is read to
// its end by its one workgroup, as many tracks in flight as the CUs hold workgroups.  Real 24-bit material is loud
// within its first sub-blocks: the 512-track share of cfg3 0.28 -> 0.01 ms, all 4096 tracks 1.9 -> 0.03 ms.
template <int FMT>
__global__ __launch_bounds__(256) void k_hires_scan_tracks(const uint8_t *pcm, EncWorkspace ws, uint32_t n_tracks)
{
    constexpr int NB = 32 * Fmt<FMT>::BYTES;
    __shared__ uint4 stg[NB <= 128 ? 4 * 64 * NB / 16 : 1];
    for (uint32_t t = blockIdx.x; t < n_tracks; t += gridDim.x) {
        const uint64_t t_samples = ws.tracks[t].n_samples, t_off = ws.tracks[t].byte_off;
        const uint32_t nc = ws.tracks[t].n_chunks;
        if (ws.tracks[t].high_res & 0x100u) continue;                     // forced by the host
        for (uint32_t k = 0; k < nc; k++) {
            const uint64_t start = (uint64_t)k * VLX_SUB_BLOCK;
            const uint64_t end = start + VLX_SUB_BLOCK < t_samples ? start + VLX_SUB_BLOCK : t_samples;
            const int npairs = (int)((end - start) / 2);
            const int p0 = threadIdx.x * 16;
            int valid = npairs - p0; valid = valid < 0 ? 0 : (valid > 16 ? 16 : valid);
            typename Fmt<FMT>::S L[16], R[16];
            const uint8_t *cbase = pcm + t_off + start * Fmt<FMT>::BYTES;
            const int wave = threadIdx.x >> 6;
            if (NB <= 128 && __all(valid >= 16)) vlx_load16pairs_wave<FMT>(cbase + (size_t)wave * 64 * NB, stg + wave * 64 * (NB / 16), threadIdx.x & 63, L, R);
            else vlx_load16pairs<FMT>(cbase + 2 * (uint64_t)p0 * Fmt<FMT>::BYTES, valid, L, R);
            bool loud = false;
#pragma unroll
            for (int i = 0; i < 16; i++) loud |= (int64_t)vlx_uabs((int64_t)L[i]) > 65536ll || (int64_t)vlx_uabs((int64_t)R[i]) > 65536ll;   // (signed compare: see k_hires_scan)
            if (__syncthreads_or(loud)) {                                 // (the barrier also keeps the next chunk's staging behind this one's reads)
                if (threadIdx.x == 0) __hip_atomic_store(&ws.track_hires[t * VLX_HIRES_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void k_hires_scan(const uint8_t *pcm, EncWorkspace ws, uint32_t n_tracks, uint64_t n_chunks)
{
    constexpr int NB = 32 * Fmt<FMT>::BYTES;
    __shared__ uint4 stg[NB <= 128 ? 4 * 64 * NB / 16 : 1];
    // a workgroup walks chunks blockIdx.x, + gridDim.x, ...: a long track whose answer is found in its first chunks
    // costs each workgroup one poll instead of costing the dispatcher one workgroup per chunk (cfg2: 84,375).
    // One barrier per chunk: the "already found" poll (long tracks only: polling one hot word from every workgroup of
    // a corpus of short tracks queues up on a single L2 channel, 16 ms for 4096 tracks x 118 chunks) is issued in front
    // of the chunk's loads and travels through LDS behind the barrier that agrees on the chunk's own verdict, so the
    // whole workgroup takes the same decision; the track's record stays in registers while the walk is inside it.
    __shared__ uint32_t found_s[2];                                        // by iteration parity: a wave is never two barriers behind
    uint64_t t_base = 1, t_end = 0, t_samples = 0, t_off = 0;
    uint32_t t = 0, t_hr = 0;
    int it = 0;
    for (uint64_t c = blockIdx.x; c < n_chunks; c += gridDim.x, it++) {
    if (c < t_base || c >= t_end) {
        t = ws.chunk_track[c];
        t_base = ws.tracks[t].chunk_base; t_end = t_base + ws.tracks[t].n_chunks;
        t_samples = ws.tracks[t].n_samples; t_off = ws.tracks[t].byte_off; t_hr = ws.tracks[t].high_res;
    }
    const bool long_track = t_end - t_base > 512;
    // nothing (more) to learn about this track: on to the next one's chunks — the loop's own step makes it the first
    // c' >= t_end in this residue class
    auto next_track = [&]() { if (t_end > c + gridDim.x) c += (t_end - c - 1) / gridDim.x * gridDim.x; };
    if (t_hr & 0x100u) { next_track(); continue; }                       // forced by the host
    uint32_t found = 0;
    if (long_track && threadIdx.x == 0) found = __hip_atomic_load(&ws.track_hires[t * VLX_HIRES_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t start = (c - t_base) * VLX_SUB_BLOCK;
    const uint64_t end = start + VLX_SUB_BLOCK < t_samples ? start + VLX_SUB_BLOCK : t_samples;
    const int npairs = (int)((end - start) / 2);
    const int p0 = threadIdx.x * 16;
    int valid = npairs - p0; valid = valid < 0 ? 0 : (valid > 16 ? 16 : valid);
    typename Fmt<FMT>::S L[16], R[16];
    const uint8_t *cbase = pcm + t_off + start * Fmt<FMT>::BYTES;
    const int wave = threadIdx.x >> 6;
    if (NB <= 128 && __all(valid >= 16)) vlx_load16pairs_wave<FMT>(cbase + (size_t)wave * 64 * NB, stg + wave * 64 * (NB / 16), threadIdx.x & 63, L, R);
    else vlx_load16pairs<FMT>(cbase + 2 * (uint64_t)p0 * Fmt<FMT>::BYTES, valid, L, R);
    bool loud = false;
#pragma unroll
    // std::abs(INT64_MIN) stays INT64_MIN in the reference's build (negq): such a sample is NOT loud, hence the signed compare
    for (int i = 0; i < 16; i++) loud |= (int64_t)vlx_uabs((int64_t)L[i]) > 65536ll || (int64_t)vlx_uabs((int64_t)R[i]) > 65536ll;
    if (threadIdx.x == 0) found_s[it & 1] = found;
    const bool any = __syncthreads_or(loud);
    const bool known = found_s[it & 1] != 0;
    // One plain flag store per workgroup: every writer stores the same 1, so no read-modify-write is
    // needed (same-line atomics from the tracks in flight serialise at ~90 per microsecond).
    if (any && !known && threadIdx.x == 0) __hip_atomic_store(&ws.track_hires[t * VLX_HIRES_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (long_track && (any || known)) next_track();
    }
}

// ----------------------------------------------------------------------------------------------
// Levinson-Durbin, order 8, reflection clamp +-0.999, quantise to 2^11 (VeloxCore.h:53-65).
// Strict IEEE double, no contraction — the reference binary has no FMA.
__device__ void vlx_levinson(const double *ac, int32_t *coefs, int32_t *shift)
{
    if (fabs(ac[0]) < 1e-9) { *shift = 0; for (int i = 0; i < VLX_LPC_ORDER; i++) coefs[i] = 0; return; }
    double a[VLX_LPC_ORDER + 1][VLX_LPC_ORDER + 1];
    double e[VLX_LPC_ORDER + 1];
#pragma unroll
    for (int i = 0; i <= VLX_LPC_ORDER; i++)
#pragma unroll
        for (int j = 0; j <= VLX_LPC_ORDER; j++) a[i][j] = 0.0;
    e[0] = ac[0];
#pragma unroll
    for (int i = 1; i <= VLX_LPC_ORDER; i++) {
        double k = ac[i];
#pragma unroll
        for (int j = 1; j < i; j++) k -= a[j][i - 1] * ac[i - j];
        k /= e[i - 1];
        if (k > 0.999) k = 0.999;
        if (k < -0.999) k = -0.999;
        a[i][i] = k;
#pragma unroll
        for (int j = 1; j < i; j++) a[j][i] = a[j][i - 1] - k * a[i - j][i - 1];
        e[i] = e[i - 1] * (1 - k * k);
    }
    *shift = 11;
#pragma unroll
    for (int i = 1; i <= VLX_LPC_ORDER; i++) coefs[i - 1] = (int32_t)floor(a[i][VLX_LPC_ORDER] * 2048.0 + 0.5);
}

#define AN_THREADS 256
#define XS_PAD 8                        // leading zeros so x[i-1-j] with i-1-j < 0 reads 0 (VeloxCore.h:102 "if(i > j)")
#define RED_ROW 264                     // 8 segments of 33: conflict-free 8-byte reads for 4 rows x 8 segments per half wave

// Arithmetic type of the pre-pass.  16- and 24-bit containers: L+R, L-R and every |x| fit int32,
// so the whole pre-pass runs in 32-bit registers; 32-bit containers and the int64 boundary need 64.
template <int FMT> struct Arith { typedef int32_t T; typedef uint32_t U; };
template <> struct Arith<VLX_FMT_S32> { typedef int64_t T; typedef uint64_t U; };
template <> struct Arith<VLX_FMT_I64> { typedef int64_t T; typedef uint64_t U; };
template <> struct Arith<VLX_FMT_F32_D16> { typedef int64_t T; typedef uint64_t U; };
template <> struct Arith<VLX_FMT_F32_D24> { typedef int64_t T; typedef uint64_t U; };
__device__ __forceinline__ uint32_t vlx_uabs(int32_t v) { return v < 0 ? (0u - (uint32_t)v) : (uint32_t)v; }

struct OpSum { __device__ __forceinline__ uint64_t operator()(uint64_t a, uint64_t b) const { return a + b; }
               __device__ __forceinline__ void lds(unsigned long long *p, uint64_t v) const { atomicAdd(p, (unsigned long long)v); } };
struct OpOr  { __device__ __forceinline__ uint64_t operator()(uint64_t a, uint64_t b) const { return a | b; }
               __device__ __forceinline__ void lds(unsigned long long *p, uint64_t v) const { atomicOr(p, (unsigned long long)v); } };
struct OpMax { __device__ __forceinline__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; }
               __device__ __forceinline__ void lds(unsigned long long *p, uint64_t v) const { atomicMax(p, (unsigned long long)v); } };

// Workgroup-wide reduction of N values per thread (N <= AN_RED_MAX) through LDS atomics: every thread
// folds its value into red[k][lane] (the four waves hit the same word in different instructions, so
// there is no intra-instruction conflict), then 8 threads per value fold 8 words each, then one thread
// the last 8.  All three operators have identity 0.  Results land in red3[0..N) for every thread to read.
// Replaces 12 ds_bpermute + 12 VALU per 64-bit value.
// `red` must be all zero on entry (the kernel clears it once); every call leaves it zero again for the
// next one, in the shadow of its last fold, so a reduction costs three barriers.
#define AN_RED_MAX 9
template <int N, typename Op>
__device__ __forceinline__ void vlx_block_reduce(const uint64_t (&v)[N], unsigned long long *red, uint64_t *red2, uint64_t *red3, int tid, Op op)
{
#pragma unroll
    for (int k = 0; k < N; k++) op.lds(&red[k * 64 + (tid & 63)], v[k]);
    __syncthreads();
    if (tid < N * 8) {
        const unsigned long long *p = red + (tid >> 3) * 64 + (tid & 7) * 8;
        uint64_t acc = p[0];
#pragma unroll
        for (int i = 1; i < 8; i++) acc = op(acc, p[i]);
        red2[tid] = acc;
    }
    __syncthreads();
    for (int i = tid; i < N * 64; i += AN_THREADS) red[i] = 0;
    if (tid < N) {
        uint64_t acc = red2[tid * 8];
#pragma unroll
        for (int i = 1; i < 8; i++) acc = op(acc, red2[tid * 8 + i]);
        red3[tid] = acc;
    }
    __syncthreads();
}

// The second half of the analysis of one chunk, from the autocorrelation on: Levinson-Durbin on one lane per stream, the
// quantised-LPC residual, its int16 planes, the stream parameters.  Called by all 256 threads of the workgroup, by
// k_analyze itself or — for a chunk whose autocorrelation went through k_seq_chains — by k_analyze_finish.  `red` is
// all zero on entry (vlx_block_reduce); w0 / w1 hold the thread's 16 stream values and the 8 before them.
template <int FMT>
__device__ __forceinline__ void vlx_analyze_tail(const EncWorkspace &ws, uint64_t c, int tid, int npairs, const bool (&silent)[2], const bool (&wide)[2],
                                                 const int (&shl)[2], bool use_ms, const int32_t (&w0)[24], const int32_t (&w1)[24],
                                                 double (*ac_s)[VLX_LPC_ORDER + 1], int32_t (*coef_s)[VLX_LPC_ORDER + 1],
                                                 unsigned long long *red, uint64_t *red2, uint64_t *red3)
{
    const int lane = tid & 63, wave = tid >> 6, p0 = tid * 16;
    if (wave < 2 && lane == 0) {
        int32_t cf[VLX_LPC_ORDER]; int32_t sh = 0;
        if (silent[wave] || wide[wave]) { for (int i = 0; i < VLX_LPC_ORDER; i++) cf[i] = 0; }
        else {
            double ac[VLX_LPC_ORDER + 1];
#pragma unroll
            for (int i = 0; i <= VLX_LPC_ORDER; i++) ac[i] = ac_s[wave][i];
            vlx_levinson(ac, cf, &sh);
        }
#pragma unroll
        for (int i = 0; i < VLX_LPC_ORDER; i++) coef_s[wave][i] = cf[i];
        coef_s[wave][VLX_LPC_ORDER] = sh;
    }
    __syncthreads();

    // ---- quantised-LPC residual resLPC = x - (int32)(sum >> shift) (VeloxCore.h:100-105)
    uint64_t mo[4] = {0, 0, 0, 0};                         // "some residual outside int16" and "left int32" per stream
    uint32_t hi_pk[2][8];                                  // high halves of this thread's residuals, two per dword
#pragma unroll
    for (int s = 0; s < 2; s++) {
        int32_t cf[VLX_LPC_ORDER];
#pragma unroll
        for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = coef_s[s][j];
        const int sh = coef_s[s][VLX_LPC_ORDER];
        const int32_t *w = s ? w1 : w0;
        int32_t res[16]; uint32_t big = 0, over = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            int64_t sum = 0;
#pragma unroll
            for (int j = 0; j < VLX_LPC_ORDER; j++) sum = vlx_mad64(cf[j], w[8 + i - 1 - j], sum);
            const int32_t pred = (int32_t)(sum >> sh), x = w[8 + i];
            res[i] = (int32_t)((uint32_t)x - (uint32_t)pred);                  // int64 in the reference (:105): flag when it leaves int32
            over |= (uint32_t)((x ^ pred) & (x ^ res[i]));                     // sign bit: the subtraction overflowed
            big |= (uint32_t)res[i] + 32768u;                                  // bits 16.. set: outside int16
        }
        mo[s] = big >> 16; mo[2 + s] = over >> 31;                         // non-zero = some residual outside int16 / int32
        // Two int16 planes per stream (low halves, high halves): the low plane goes out now; the high
        // plane is only written — and only read by k_encode_serial — when some residual of the stream
        // lies outside int16, which the reduction below decides.  16-bit and high-res 24-bit material never
        // needs it: 1.38 GB less written here and read there (cfg2).
        uint32_t lo_pk[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            lo_pk[j] = __builtin_amdgcn_perm((uint32_t)res[2 * j + 1], (uint32_t)res[2 * j], 0x05040100u);
            hi_pk[s][j] = __builtin_amdgcn_perm((uint32_t)res[2 * j + 1], (uint32_t)res[2 * j], 0x07060302u);
        }
        if (!silent[s] && !wide[s]) {
            uint4 *dst = (uint4 *)((int16_t *)(ws.res + (2 * c + s) * VLX_STREAM_LEN) + p0);
            dst[0] = make_uint4(lo_pk[0], lo_pk[1], lo_pk[2], lo_pk[3]);
            dst[1] = make_uint4(lo_pk[4], lo_pk[5], lo_pk[6], lo_pk[7]);
        }
    }
    vlx_block_reduce<4>(mo, red, red2, red3, tid, OpMax());
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (!silent[s] && !wide[s] && !red3[2 + s] && red3[s]) {           // uniform over the workgroup
            uint4 *dst = (uint4 *)((int16_t *)(ws.res + (2 * c + s) * VLX_STREAM_LEN) + VLX_STREAM_LEN + p0);
            dst[0] = make_uint4(hi_pk[s][0], hi_pk[s][1], hi_pk[s][2], hi_pk[s][3]);
            dst[1] = make_uint4(hi_pk[s][4], hi_pk[s][5], hi_pk[s][6], hi_pk[s][7]);
        }
    }
    if (tid < 2) {
        const int s = tid;
        const uint32_t big = (uint32_t)red3[s], ov = (uint32_t)red3[2 + s];
        StreamParams P;
        P.n = npairs;
        P.flags = (silent[s] ? SF_SILENT : 0u);
        if (!silent[s] && (wide[s] || ov)) P.flags |= SF_WIDE;
        if (!silent[s] && !(P.flags & SF_WIDE) && !big) P.flags |= SF_FAST16;
        P.shift_lsb = shl[s];
        P.lpc_shift = coef_s[s][VLX_LPC_ORDER];
        for (int j = 0; j < VLX_LPC_ORDER; j++) P.coefs[j] = coef_s[s][j];
        P.maxres = 0;
        P.pad[0] = P.pad[1] = P.pad[2] = 0;
        ws.params[2 * c + s] = P;
        if (s == 0) ws.chunk_use_ms[c] = use_ms ? 1u : 0u;
    }
}

template <int FMT>
__global__ __launch_bounds__(AN_THREADS) void k_analyze(const uint8_t *pcm, EncWorkspace ws, uint32_t n_tracks)
{
    typedef typename Fmt<FMT>::S S;
    typedef typename Arith<FMT>::T AT;
    typedef typename Arith<FMT>::U UT;
    __shared__ __attribute__((aligned(16))) int32_t xs[2][XS_PAD + VLX_STREAM_LEN];
    __shared__ unsigned long long red[AN_RED_MAX * 64];
    __shared__ uint64_t red2[AN_RED_MAX * 8];
    __shared__ uint64_t red3[AN_RED_MAX];
    __shared__ double   ac_s[2][VLX_LPC_ORDER + 1];
    __shared__ int32_t  coef_s[2][VLX_LPC_ORDER + 1];      // [8] = shift
    __shared__ int32_t  seq_s[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < AN_RED_MAX * 64; i += AN_THREADS) red[i] = 0;   // vlx_block_reduce expects and leaves it zero
    const uint64_t c = blockIdx.x;
    const uint32_t t = ws.chunk_track[c];
    const uint64_t start = (c - ws.tracks[t].chunk_base) * VLX_SUB_BLOCK;
    const uint64_t total = ws.tracks[t].n_samples;
    const uint64_t end = start + VLX_SUB_BLOCK < total ? start + VLX_SUB_BLOCK : total;
    const int npairs = (int)((end - start) / 2);
    const bool hires = (ws.tracks[t].high_res & 0x100u) ? (ws.tracks[t].high_res & 1u) : (ws.track_hires[t * VLX_HIRES_STRIDE] != 0);
    const int p0 = tid * 16;
    int valid = npairs - p0; valid = valid < 0 ? 0 : (valid > 16 ? 16 : valid);

    S L[16], R[16];
    {
        constexpr int NB = 32 * Fmt<FMT>::BYTES;
        const uint8_t *cbase = pcm + ws.tracks[t].byte_off + start * Fmt<FMT>::BYTES;
        // staging area = xs, which nothing else touches before the two workgroup reductions below
        if (NB <= 128 && __all(valid >= 16)) vlx_load16pairs_wave<FMT>(cbase + (size_t)wave * 64 * NB, (uint4 *)&xs[0][0] + wave * 64 * (NB / 16), lane, L, R);
        else vlx_load16pairs<FMT>(cbase + 2 * (uint64_t)p0 * Fmt<FMT>::BYTES, valid, L, R);
    }

    // ---- SAD of L/R vs mid/side (VeloxCore.h:262-271); padding pairs are zero and add nothing.
    // Per-thread partial sums of 16 pairs fit UT (2^25 * 32 < 2^32 for the 32-bit arithmetic).
    AT a[16], b[16];                                       // becomes the stream values; starts as mid / side
    {
        UT sl = 0, sm = 0;
        const uint32_t bias = 0x80000000u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const AT l = (AT)L[i], r = (AT)R[i];
            a[i] = (l + r) >> 1; b[i] = l - r;
            if (sizeof(AT) == 4) {
                // |x| + acc as ONE v_sad_u32 on offset-binary operands (x ^ 2^31 orders like x): |l - r| comes straight
                // from l and r, the other three against the offset itself — 7 instructions per pair instead of 12
                const uint32_t lb = (uint32_t)l ^ 0x80000000u, rb = (uint32_t)r ^ 0x80000000u, ab = (uint32_t)a[i] ^ 0x80000000u;
                sl = (UT)vlx_sad(lb, bias, vlx_sad(rb, bias, (uint32_t)sl));
                sm = (UT)vlx_sad(ab, bias, vlx_sad(lb, rb, (uint32_t)sm));
            } else {
                sl += (UT)vlx_uabs(l) + (UT)vlx_uabs(r);
                sm += (UT)vlx_uabs(a[i]) + (UT)vlx_uabs(b[i]);
            }
        }
        const uint64_t v[2] = {(uint64_t)sl, (uint64_t)sm};
        __syncthreads();                                   // red is cleared (top of the kernel), the staged load is done with xs
        vlx_block_reduce<2>(v, red, red2, red3, tid, OpSum());
    }
    const bool use_ms = red3[1] < red3[0];

    // ---- stream values, high-res byte split (VeloxCore.h:272-277, :73-79)
    if (!use_ms) {
#pragma unroll
        for (int i = 0; i < 16; i++) { a[i] = (AT)L[i]; b[i] = (AT)R[i]; }
    }
    if (hires) {
        uint32_t la[4], lb[4];
        // the low bytes of four values as one dword: three v_perm_b32 instead of four and + shift-or pairs
        auto low4 = [](uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
            return __builtin_amdgcn_perm(__builtin_amdgcn_perm(x3, x2, 0x0c0c0400u), __builtin_amdgcn_perm(x1, x0, 0x0c0c0400u), 0x05040100u);
        };
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            la[i >> 2] = low4((uint32_t)a[i], (uint32_t)a[i + 1], (uint32_t)a[i + 2], (uint32_t)a[i + 3]);
            lb[i >> 2] = low4((uint32_t)b[i], (uint32_t)b[i + 1], (uint32_t)b[i + 2], (uint32_t)b[i + 3]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) { a[i] >>= 8; b[i] >>= 8; }
        *(uint4 *)(ws.low + (2 * c) * VLX_STREAM_LEN + p0) = make_uint4(la[0], la[1], la[2], la[3]);
        *(uint4 *)(ws.low + (2 * c + 1) * VLX_STREAM_LEN + p0) = make_uint4(lb[0], lb[1], lb[2], lb[3]);
    }

    // ---- silence + LSB shift: OR of |x| (VeloxAdvanced.h:14-22, VeloxFormat.h:196-212)
    {
        // 16 / 24-bit containers: what is asked of the OR of |x| below — zero or not, its trailing zeros — the OR of the
        // values themselves answers too (negation keeps a value's trailing zeros), and the magnitude tests cannot fire
        // (|x| < 2^25 whatever the M/S choice): one instruction per value instead of three
        constexpr bool SMALL = Fmt<FMT>::BYTES <= 3 && sizeof(AT) == 4;
        UT m0 = 0, m1 = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (SMALL) { m0 |= (UT)a[i]; m1 |= (UT)b[i]; }
            else { m0 |= (UT)vlx_uabs(a[i]); m1 |= (UT)vlx_uabs(b[i]); }
        }
        const uint64_t v[2] = {(uint64_t)m0, (uint64_t)m1};
        vlx_block_reduce<2>(v, red, red2, red3, tid, OpOr());
    }
    int shl[2]; bool silent[2], wide[2], intok[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const uint64_t mask = red3[s];
        silent[s] = mask == 0;
        int tz = silent[s] ? 0 : __builtin_ctzll(mask);
        shl[s] = tz > 32 ? 32 : tz;
        uint64_t top = mask >> shl[s];
        constexpr bool SMALL = Fmt<FMT>::BYTES <= 3 && sizeof(AT) == 4;   // (mask is then the OR of the raw values: see above)
        wide[s] = !SMALL && top >= (1ull << 31);           // shifted values may not fit int32
        intok[s] = SMALL || top < (1ull << 25);            // R[0] <= 4096 * 2^50 fits int64
    }
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        *(int4 *)&xs[0][XS_PAD + p0 + i] = make_int4((int32_t)(a[i] >> shl[0]), (int32_t)(a[i + 1] >> shl[0]), (int32_t)(a[i + 2] >> shl[0]), (int32_t)(a[i + 3] >> shl[0]));
        *(int4 *)&xs[1][XS_PAD + p0 + i] = make_int4((int32_t)(b[i] >> shl[1]), (int32_t)(b[i + 1] >> shl[1]), (int32_t)(b[i + 2] >> shl[1]), (int32_t)(b[i + 3] >> shl[1]));
    }
    if (tid < XS_PAD) { xs[0][tid] = 0; xs[1][tid] = 0; }
    __syncthreads();

    // ---- autocorrelation lags 0..8.  Exact integer sums; they equal the reference's sequential
    // double accumulation whenever R[0] < 2^53 (every partial sum is then an exact integer <= R[0]).
    int32_t w0[24], w1[24];
#pragma unroll
    for (int i = 0; i < 24; i += 4) {
        const int4 u = *(const int4 *)&xs[0][p0 + i], v = *(const int4 *)&xs[1][p0 + i];
        w0[i] = u.x; w0[i + 1] = u.y; w0[i + 2] = u.z; w0[i + 3] = u.w;
        w1[i] = v.x; w1[i + 1] = v.y; w1[i + 2] = v.z; w1[i + 3] = v.w;
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int32_t *w = s ? w1 : w0;
        uint64_t acc[9];
#pragma unroll
        for (int lag = 0; lag <= VLX_LPC_ORDER; lag++) {
            int64_t sm = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) sm = vlx_mad64(w[8 + i], w[8 + i - lag], sm);
            acc[lag] = (uint64_t)sm;
        }
        vlx_block_reduce<9>(acc, red, red2, red3, tid, OpSum());
        if (tid < 9) {
            const int64_t r = (int64_t)red3[tid];
            ac_s[s][tid] = (double)r;
            if (tid == 0) seq_s[s] = (!intok[s] || r >= (1ll << 53)) ? 1 : 0;
        }
    }
    __syncthreads();
    // ---- energy >= 2^53 (loud 32-bit material): replay the reference's sequential double accumulation
    // (VeloxCore.h:48-52).  The additions round, so their order is the reference's: 18 chains (2 streams
    // x 9 lags), one lane each, 4096 dependent fp64 adds.  The products do not depend on one another:
    // all 256 threads form them, RP_T samples of every chain at a time, into LDS (sample-major, so the
    // chain lanes read neighbouring doubles); a chain lane then only loads and adds.  Products with
    // j < lag or j >= npairs are (+-)0 from the zero padding and leave the sum as it is.  Only 32-bit
    // and int64 input can get here unforced (16/24-bit material stays below 2^46); 14 KB of tiles keep
    // those instantiations at three workgroups per CU, which matters more than fewer barriers: the
    // chains of the co-resident workgroups overlap (cfg4: 1.0 ms with 256-sample tiles at two per CU, 0.82 ms
    // with 64-sample tiles at three).
    // The chain is what bounds this phase (4096 dependent additions at ~19 cycles each), so one wave does nothing
    // else: the other three form the products two tiles ahead into a ring of three tiles, one barrier per tile.
    // In iteration k the producers write tile k+2; tiles k and k+1 are complete, so the chain lanes add tile k
    // while their loads of the rest of it and of the head of tile k+1 are in flight: after a barrier the
    // first RP_H additions never wait for LDS.
    constexpr bool RP_OWN = Fmt<FMT>::BYTES >= 4;
    constexpr int RP_T = RP_OWN ? 32 : 8;                                // samples per tile
    constexpr int RP_H = 8;                                              // head of a tile loaded one iteration early
    constexpr int RP_ROW = RP_OWN ? 19 : 18;                             // doubles per sample row (19: conflict-free writes)
    constexpr int RP_NP = (RP_T * 18 + 191) / 192;                       // products per producer thread and tile
    __shared__ double prod_s[RP_OWN ? 3 * RP_T * RP_ROW : 1];
    const bool need0 = seq_s[0] && !silent[0] && !wide[0], need1 = seq_s[1] && !silent[1] && !wide[1];
    if (need0 || need1) {                                                 // uniform over the workgroup
        // 4-byte formats, while the batch's slots last: the chains do not run here (18 lanes of a workgroup that holds
        // 53 KB of LDS: three chains of 19-cycle additions per CU) but in k_seq_chains, 252 chains to a workgroup and five
        // workgroups to a CU; this workgroup leaves its stream values, what it knows about them and the exact sums of
        // the stream that needs no replay in a slot and ends; k_analyze_finish takes the chunk up again from there.
        if (RP_OWN && ws.seq_cap) {
            __shared__ uint32_t slot_s;
            if (tid == 0) slot_s = atomicAdd(ws.seq_count, 1u);
            __syncthreads();
            const uint32_t slot = slot_s;
            if (slot < ws.seq_cap) {
                int4 *dst = (int4 *)(ws.seq_xs + (size_t)slot * 2 * VLX_SEQ_ROW);
                const int4 *src = (const int4 *)&xs[0][0];
                for (int i = tid; i < 2 * VLX_SEQ_ROW / 4; i += AN_THREADS) dst[i] = src[i];
                if (tid < 18) ws.seq_ac[(size_t)slot * 18 + tid] = ac_s[tid / 9][tid % 9];
                if (tid == 0) {
                    SeqSlot q;
                    q.chunk = (uint32_t)c; q.npairs = npairs;
                    q.flags = (need0 ? 1u : 0u) | (need1 ? 2u : 0u) | (silent[0] ? 4u : 0u) | (silent[1] ? 8u : 0u) | (wide[0] ? 16u : 0u) | (wide[1] ? 32u : 0u) |
                              (use_ms ? 64u : 0u);
                    q.shl0 = (uint32_t)shl[0]; q.shl1 = (uint32_t)shl[1]; q.pad[0] = q.pad[1] = q.pad[2] = 0;
                    ws.seq_slot[slot] = q;
                }
                return;
            }
        }
        double *prod = RP_OWN ? prod_s : (double *)red;                   // [3][RP_T samples][RP_ROW >= 18 chains]; 3*8*18*8 B fit in red
        const int ntiles = (npairs + RP_T - 1) / RP_T;
        // a producer's products: e = chain * RP_T + sample (a wave reads consecutive samples), the same every tile
        // (Which wave carries the chains makes no difference: rotating it with blockIdx so that co-resident
        // workgroups use different SIMDs measured the same, the chain is bound by the ~19-cycle latency of a
        // dependent v_add_f64, not by issue slots.)
        constexpr int cw = 0;
        const int pt = wave == cw ? -1 : ((wave < cw ? wave : wave - 1) * 64 + lane);      // producer index 0..191
        int px[RP_NP], plag[RP_NP], pdst[RP_NP];
#pragma unroll
        for (int j = 0; j < RP_NP; j++) {
            const int e = pt < 0 ? -1 : pt + 192 * j;
            const int chn = e / RP_T, i = e - chn * RP_T;
            const int s = chn >= 9 ? 1 : 0;
            plag[j] = chn - 9 * s;
            px[j] = s * (XS_PAD + VLX_STREAM_LEN) + XS_PAD + i;
            pdst[j] = (e >= 0 && e < RP_T * 18) ? i * RP_ROW + chn : -1;
        }
        auto produce = [&](int k, int slot) {
            if (k >= ntiles) return;
            double *dst = prod + slot * (RP_T * RP_ROW);
            const int32_t *xb = &xs[0][0] + RP_T * k;
#pragma unroll
            for (int j = 0; j < RP_NP; j++)
                if (pdst[j] >= 0) dst[pdst[j]] = (double)xb[px[j]] * (double)xb[px[j] - plag[j]];
        };
        if (wave != cw) { produce(0, 0); produce(1, 1); }
        __syncthreads();
        double sum = 0, head[RP_H];
        const bool chain = wave == cw && lane < 18;                       // chain `lane`
        if (chain) {
#pragma unroll
            for (int u = 0; u < RP_H; u++) head[u] = prod[u * RP_ROW + lane];
        }
        int slot = 0;                                                     // of tile k
        for (int k = 0; k < ntiles; k++) {
            const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
            if (wave == cw) {
                if (chain) {
                    const double *cur = prod + slot * (RP_T * RP_ROW) + lane, *nxt = prod + slot1 * (RP_T * RP_ROW) + lane;
                    double rest[RP_T > RP_H ? RP_T - RP_H : 1], nhead[RP_H];
#pragma unroll
                    for (int u = RP_H; u < RP_T; u++) rest[u - RP_H] = cur[u * RP_ROW];
#pragma unroll
                    for (int u = 0; u < RP_H; u++) nhead[u] = nxt[u * RP_ROW];    // beyond the last tile: stale, never added
#pragma unroll
                    for (int u = 0; u < RP_H; u++) sum += head[u];
#pragma unroll
                    for (int u = RP_H; u < RP_T; u++) sum += rest[u - RP_H];
#pragma unroll
                    for (int u = 0; u < RP_H; u++) head[u] = nhead[u];
                }
            } else produce(k + 2, slot2);
            __syncthreads();
            slot = slot1;
        }
        if (chain) {
            const int s = lane >= 9 ? 1 : 0;
            if (s ? need1 : need0) ac_s[s][lane - 9 * s] = sum;
        }
        if (!RP_OWN) for (int i = tid; i < 3 * RP_T * RP_ROW; i += AN_THREADS) red[i] = 0;   // borrowed from the reductions: hand it back cleared
        __syncthreads();                                                  // (xs is read again below: nothing here wrote it)
#pragma unroll
        for (int i = 0; i < 24; i += 4) {                                 // w0 / w1 again, so that they need not live through the chains
            const int4 u = *(const int4 *)&xs[0][p0 + i], v = *(const int4 *)&xs[1][p0 + i];
            w0[i] = u.x; w0[i + 1] = u.y; w0[i + 2] = u.z; w0[i + 3] = u.w;
            w1[i] = v.x; w1[i + 1] = v.y; w1[i + 2] = v.z; w1[i + 3] = v.w;
        }
    }
    __syncthreads();
    vlx_analyze_tail<FMT>(ws, c, tid, npairs, silent, wide, shl, use_ms, w0, w1, ac_s, coef_s, red, red2, red3);
}

// The chains of the parked chunks.  A chain is 4096 dependent fp64 additions in the reference's order (VeloxCore.h:48-52:
// sum += (double)x[i] * (double)x[i-lag]; products with i < lag are the zero padding's +-0 and leave the sum as it is).
// A workgroup takes 28 rows (slot, stream) = 252 chains, lane = (row, lag), and walks them in tiles of SC_T samples staged
// through LDS: a tile of a row (with the 8 samples in front of it, for the lags) is 544 contiguous bytes, fetched by the
// whole workgroup one tile ahead of the chains; the nine lanes of a row read the same words (broadcast).  Five
// workgroups per CU put five waves on every SIMD: the 19 cycles of a dependent addition are hidden, the fp64 rate and
// the LDS reads are what is left.  cfg4: 202,500 chains in 0.10 ms.  Measured before this: one lane per chain streaming
// its row from global memory 16 bytes at a time (two quads ahead, pinned with an empty asm so that hipcc did not turn
// "load the next quad, work on this one" into "load, wait, convert at once"), 0.22 ms — bound by 22,500 rows fetched from
// HBM 128 bytes at a time; three chains per lane (a third of the loads, a third of the waves) 0.21 ms; the int -> double
// conversions as integer tricks 0.23 ms.  And after it: the tile kept as doubles (converted once per value instead of
// twice per product): 0.115 ms — twice the LDS bytes cost more than the conversions saved.
static_assert(VLX_SEQ_PAD == XS_PAD, "the slots hold k_analyze's LDS rows as they are");
#define SC_ROWS 28
#define SC_T 128
#define SC_W (VLX_SEQ_PAD + SC_T)               // words of a row's tile in LDS
__global__ __launch_bounds__(256) void k_seq_chains(EncWorkspace ws)
{
    __shared__ __attribute__((aligned(16))) int32_t buf[2][SC_ROWS * SC_W];
    const int tid = threadIdx.x;
    const uint32_t n = min(*ws.seq_count, ws.seq_cap);
    const uint32_t row0 = blockIdx.x * SC_ROWS;                           // rows are (slot * 2 + stream)
    if (row0 >= 2u * n) return;
    const int rr = tid / 9, lag = tid - rr * 9;                           // chain lanes: tid < 252
    const uint32_t my_row = row0 + (uint32_t)rr;
    const bool chain = tid < SC_ROWS * 9 && my_row < 2u * n && ((ws.seq_slot[my_row >> 1].flags >> (my_row & 1u)) & 1u);
    // loader: the tile of all rows is SC_ROWS * (SC_W / 4) 16-byte pieces; piece p = (row p / QPR, quad p % QPR)
    constexpr int QPR = SC_W / 4, PIECES = SC_ROWS * QPR, LPT = (PIECES + 255) / 256;
    int4 st[LPT];
    auto fetch = [&](int k) {
#pragma unroll
        for (int j = 0; j < LPT; j++) {
            const int p = tid + 256 * j;
            const int r = p / QPR, q = p - r * QPR;
            const uint32_t grow = row0 + (uint32_t)r;
            st[j] = (p < PIECES && grow < 2u * n) ? *(const int4 *)(ws.seq_xs + (size_t)grow * VLX_SEQ_ROW + k * SC_T + 4 * q) : make_int4(0, 0, 0, 0);
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int j = 0; j < LPT; j++) { const int p = tid + 256 * j; if (p < PIECES) *(int4 *)&buf[b][4 * p] = st[j]; }
    };
    fetch(0); stash(0);
    __syncthreads();
    double sum = 0;
    constexpr int NT = VLX_STREAM_LEN / SC_T;
    for (int k = 0; k < NT; k++) {
        if (k + 1 < NT) fetch(k + 1);
        if (chain) {
            const int32_t *xr = &buf[k & 1][rr * SC_W + VLX_SEQ_PAD];
#pragma unroll 8
            for (int i = 0; i < SC_T; i += 4) {
                const int4 a = *(const int4 *)(xr + i);
                const int32_t b0 = xr[i - lag], b1 = xr[i + 1 - lag], b2 = xr[i + 2 - lag], b3 = xr[i + 3 - lag];
                sum += (double)a.x * (double)b0;
                sum += (double)a.y * (double)b1;
                sum += (double)a.z * (double)b2;
                sum += (double)a.w * (double)b3;
            }
        }
        if (k + 1 < NT) stash((k + 1) & 1);
        __syncthreads();
    }
    if (chain) ws.seq_ac[(size_t)(my_row >> 1) * 18 + (my_row & 1u) * 9u + (uint32_t)lag] = sum;
}

// A parked chunk taken up again: its stream values come from the slot instead of the PCM, its autocorrelation from
// seq_ac, the rest is k_analyze's second half.
template <int FMT>
__global__ __launch_bounds__(AN_THREADS) void k_analyze_finish(EncWorkspace ws)
{
    __shared__ unsigned long long red[AN_RED_MAX * 64];
    __shared__ uint64_t red2[AN_RED_MAX * 8];
    __shared__ uint64_t red3[AN_RED_MAX];
    __shared__ double   ac_s[2][VLX_LPC_ORDER + 1];
    __shared__ int32_t  coef_s[2][VLX_LPC_ORDER + 1];
    const uint32_t slot = blockIdx.x;
    if (slot >= min(*ws.seq_count, ws.seq_cap)) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < AN_RED_MAX * 64; i += AN_THREADS) red[i] = 0;
    const SeqSlot q = ws.seq_slot[slot];
    if (tid < 18) ac_s[tid / 9][tid % 9] = ws.seq_ac[(size_t)slot * 18 + tid];
    const bool silent[2] = {(q.flags & 4u) != 0, (q.flags & 8u) != 0}, wide[2] = {(q.flags & 16u) != 0, (q.flags & 32u) != 0};
    const int shl[2] = {(int)q.shl0, (int)q.shl1};
    int32_t w0[24], w1[24];
    const int32_t *r0 = ws.seq_xs + (size_t)slot * 2 * VLX_SEQ_ROW + tid * 16, *r1 = r0 + VLX_SEQ_ROW;
#pragma unroll
    for (int i = 0; i < 24; i += 4) {
        const int4 u = *(const int4 *)(r0 + i), v = *(const int4 *)(r1 + i);
        w0[i] = u.x; w0[i + 1] = u.y; w0[i + 2] = u.z; w0[i + 3] = u.w;
        w1[i] = v.x; w1[i + 1] = v.y; w1[i + 2] = v.z; w1[i + 3] = v.w;
    }
    __syncthreads();
    vlx_analyze_tail<FMT>(ws, q.chunk, tid, q.npairs, silent, wide, shl, (q.flags & 64u) != 0, w0, w1, ac_s, coef_s, red, red2, red3);
}

// ----------------------------------------------------------------------------------------------
// Per-lane LSB-first bit packer writing dwords into the stream's staging slot
// (BitStreamWriter, VeloxEntropy.h:9-43, as a 64-bit accumulator).
struct LaneBits {
    uint32_t *slot; uint64_t acc; int nacc; uint32_t pos;
    __device__ __forceinline__ void init(uint32_t *s) { slot = s; acc = 0; nacc = 0; pos = 0; }
    __device__ __forceinline__ void put(uint32_t v, int nb)           // nb in [0,32], v < 2^nb
    {
        acc |= (uint64_t)v << nacc; nacc += nb;
        if (nacc >= 32) {
            if (pos < VLX_SLOT_DWORDS - 2) slot[pos] = (uint32_t)acc;
            pos++; acc >>= 32; nacc -= 32;
        }
    }
    __device__ __forceinline__ uint32_t finish()
    {
        if (pos < VLX_SLOT_DWORDS - 2) { slot[pos] = (uint32_t)acc; slot[pos + 1] = 0; }
        return pos * 32u + (uint32_t)nacc;
    }
};

// Adaptive Rice code of one residual (VeloxEntropy.h:98-119).
__device__ __forceinline__ void vlx_rice_put(LaneBits &bw, uint64_t m, int k)
{
    uint64_t q = m >> k;
    if (q < 64) {
        uint32_t qq = (uint32_t)q;
        if (qq + 1 + (uint32_t)k <= 31) {
            uint32_t rem = (uint32_t)m & ((1u << k) - 1u);                 // k <= 30, qq <= 30 here
            bw.put(((1u << qq) - 1u) | (rem << (qq + 1)), (int)(qq + 1 + k));
        } else {
            if (qq >= 32) { bw.put(0xFFFFFFFFu, 32); qq -= 32; }
            bw.put((1u << qq) - 1u, (int)qq + 1);                          // qq <= 31: ones then the 0
            uint64_t rem = (k >= 64) ? m : (m & ((1ull << k) - 1ull));
            if (k > 32) { bw.put((uint32_t)rem, 32); bw.put((uint32_t)(rem >> 32), k - 32); }
            else if (k == 32) bw.put((uint32_t)rem, 32);
            else if (k > 0) bw.put((uint32_t)rem, k);
        }
    } else {                                                               // escape: 64 ones, 0, 40 bits of m
        bw.put(0xFFFFFFFFu, 32); bw.put(0xFFFFFFFFu, 32); bw.put(0u, 1);
        bw.put((uint32_t)m, 32); bw.put((uint32_t)(m >> 32) & 0xFFu, 8);
    }
}

#define ES_TILE 32
#define ES_STRIDE 36                    // dwords per tile row: 16-B aligned rows, conflict-free ds_read_b128 (9 is odd)

// One LANE per stream, 64 streams per wave.  Per sample the common case is branch-free 32-bit
// arithmetic: zig-zag value m < 2^32 and a code of <= 32 bits.  Longer codes (escapes, the first
// samples of a stream before run_avg has adapted) take a wave-uniform side path; a residual whose
// 64-bit difference (VeloxCore.h:107) leaves int32 hands the stream to k_encode_wide.
#define ES_WAVES 4                      // independent waves per workgroup: one per SIMD of the CU

//
// Store path, template parameter G.  G = 0: every finished dword goes straight to the lane's staging slot (a
// 4-byte store per lane, 64 different cache lines per wave instruction).  G = 4 / 8: finished dwords go into a
// per-lane ring of ES_RING dwords in LDS (laid out [slot][lane]: one bank per lane, never a conflict), and at
// wave-uniform points every lane that holds G of them writes them out as 16 / 32 aligned bytes — the writer-side
// mirror of the decoder's bit ring (PipeWin).  Between two flush points a lane adds at most 4 dwords (one per sample of
// the unrolled group) and a long code at most 4 more before its own flush, so never more than G + 6 <= ES_RING wait.
#define ES_RING 16

template <int G>
__global__ __launch_bounds__(64 * ES_WAVES) void k_encode_serial(EncWorkspace ws, uint64_t n_streams)
{
    __shared__ __attribute__((aligned(16))) int32_t tiles[ES_WAVES][64 * ES_STRIDE];
    __shared__ uint32_t rings[G ? ES_WAVES : 1][G ? ES_RING * 64 : 1];
    const int lane = threadIdx.x & 63;
    int32_t *const tile = tiles[threadIdx.x >> 6];
    uint32_t *const ring = rings[G ? (threadIdx.x >> 6) : 0] + (G ? lane : 0);
    const uint64_t s0 = (uint64_t)__builtin_amdgcn_readfirstlane(blockIdx.x * ES_WAVES + (threadIdx.x >> 6)) * 64;   // wave-uniform
    const uint64_t s = s0 + lane;
    const bool valid = s < n_streams;
    int n = 0; uint32_t flags = SF_SILENT;

    // BitStreamWriter (VeloxEntropy.h:9-43) as a 32-bit accumulator + byte offset into the wave's slots
    char *const wbase = (char *)(ws.staging + s0 * VLX_SLOT_DWORDS);
    const uint32_t bstart = (uint32_t)lane * (VLX_SLOT_DWORDS * 4u);
    const uint32_t bend = bstart + (VLX_SLOT_DWORDS - 2) * 4u;
    uint32_t boff = bstart, lo = 0, nacc = 0;                             // boff: where the next finished dword belongs
    uint32_t foff = bstart;                                               // G > 0: everything below it has left the ring
    auto put = [&](uint32_t v, uint32_t nb) {                             // nb in [0,32], v < 2^nb, nacc < 32
        // (32-bit throughout: one v_lshl_or_b32 on the common path; the bits that did not fit are fetched again only when a
        // dword fills — the 64-bit shift this replaces cost every sample two VOP3 issues)
        const uint32_t room = 31u - nacc;                                 // 32 - nacc - 1, in [0, 31]
        lo |= v << nacc; nacc += nb;
        if (nacc >= 32u) {
            if (G) ring[(((boff - bstart) >> 2) & (ES_RING - 1)) * 64] = lo;
            else *(uint32_t *)(wbase + (boff < bend ? boff : bend)) = lo;     // past the slot's end everything lands on its spare dword
            boff += 4u; lo = (v >> 1) >> room; nacc -= 32u;               // v >> (32 - nacc_before), also right for nacc_before = 0
        }
    };
    auto flush = [&]() {                                                  // the lanes that are active at the call site empty their rings
        if (!G) return;
        while (__any(boff - foff >= 4u * G)) {
            if (boff - foff >= 4u * G) {
                uint32_t d[G ? G : 1];
                const uint32_t r0 = ((foff - bstart) >> 2) & (ES_RING - 1);          // slots start 128-byte aligned and ES_RING % G == 0: no wrap inside a group
#pragma unroll
                for (int j = 0; j < G; j++) d[j] = ring[(r0 + j) * 64];
                if (foff + 4u * G <= bend) {
#pragma unroll
                    for (int j = 0; j < G; j += 4) vlx_gstore16(wbase + foff + 4 * j, make_uint4(d[j], d[j + 1], d[j + 2], d[j + 3]));
                }
                foff += 4u * G;
            }
        }
    };

    if (valid) {
        const StreamParams *P = &ws.params[s];
        n = P->n; flags = P->flags;
        if (flags & SF_SILENT) put(1u, 1);                                // VeloxCore.h:81
        else if (!(flags & SF_WIDE)) {
            put(0u, 1); put((uint32_t)P->shift_lsb & 31u, 5); put((uint32_t)P->lpc_shift & 31u, 5);     // :82-91
#pragma unroll
            for (int j = 0; j < VLX_LPC_ORDER; j++) put((uint32_t)P->coefs[j] & 0xFFFFu, 16);           // :92
        }
    }
    const bool active = valid && !(flags & (SF_SILENT | SF_WIDE));
    int n_eff = active ? n : 0;                                           // samples this lane still encodes
    const int nmax = vlx_wave_max(n_eff);
    bool redo = false;

    VlxLms lms; lms.reset();
    uint32_t run_avg = 512;                                               // VeloxCore.h:96 (stays < 2^32 while m < 2^32)
    int32_t ovf_acc = 0;

    auto step = [&](const int32_t r, const int idx, auto phase) {          // phase: the predictor's ring position, idx % 16
        constexpr int P = decltype(phase)::value;
        if (idx < n_eff) {
            const int32_t pred = lms.template predict<P>();
            const int32_t fin = (int32_t)((uint32_t)r - (uint32_t)pred);                    // :107 when it fits int32
            ovf_acc |= (r ^ pred) & (r ^ fin);                                               // sign bit: the 64-bit difference left int32 (looked at once per four samples, below)
            const uint32_t m = ((uint32_t)fin << 1) ^ (uint32_t)(fin >> 31);                // ZigZag, VeloxEntropy.h:95
            const uint32_t k = 31u - (uint32_t)__builtin_clz(run_avg);                      // :109-114
            const uint32_t q = m >> k;
            uint32_t len = q + k + 1u;
            const uint32_t rem2 = (m & ((1u << k) - 1u)) << 1;                              // remainder above the 0 terminator
            uint32_t code = ((1u << (q & 31u)) - 1u) | (rem2 << (q & 31u));                 // q ones, 0, k bits (VeloxEntropy.h:104-113)
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
                flush();                                                                    // a long code is up to 4 dwords
            }
            put(code, len);
            lms.template update<P>(fin, r);                                                 // err = (int32)resLPC - pred = fin
            run_avg = run_avg - (run_avg >> 3) + (m >> 3);                                  // :120-121 (the floor of 1 never fires: x - (x >> 3) >= 1 for x >= 1, and it starts at 512)
        }
    };
    // A lane whose difference left int32 hands its stream to k_encode_wide.  It is taken out at the next of these checks
    // (every four samples), not at the sample: what it encodes in between goes into a slot that is rewritten, with bits = 0.
    auto check_ovf = [&]() {
        if (__builtin_expect(__any(ovf_acc < 0), 0)) { if (ovf_acc < 0) { redo = true; n_eff = 0; ovf_acc = 0; } }
    };

    // Residuals come as two int16 planes per stream (k_analyze): low halves always, high halves only for
    // streams without SF_FAST16.  Tile loader: lane -> (row = lane/4 + 16r, 8 samples at (lane%4)*8), 16 B
    // per plane and load; the halves are joined into int32 on the way into the LDS tile.
    const uint64_t hi_mask = __ballot(active && !(flags & SF_FAST16));    // wave-uniform: rows that have a high plane
    const int lrow = lane >> 2, lcol = (lane & 3) * 8;
    uint4 nlo[4], nhi[4];
    auto issue = [&](int t0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = lrow + 16 * r;
            const uint64_t st = s0 + row;
            const bool ok = st < n_streams && t0 < nmax;
            const int16_t *src = (const int16_t *)(ws.res + st * VLX_STREAM_LEN) + t0 + lcol;
            nlo[r] = ok ? *(const uint4 *)src : make_uint4(0, 0, 0, 0);
            if (hi_mask) nhi[r] = (ok && ((hi_mask >> row) & 1ull)) ? *(const uint4 *)(src + VLX_STREAM_LEN) : make_uint4(0, 0, 0, 0);
        }
    };
    auto sext2 = [](uint32_t d, int32_t &a, int32_t &b) { a = (int32_t)(d << 16) >> 16; b = (int32_t)d >> 16; };
    auto join2 = [](uint32_t lo2, uint32_t hi2, int32_t &a, int32_t &b) {
        a = (int32_t)__builtin_amdgcn_perm(hi2, lo2, 0x05040100u); b = (int32_t)__builtin_amdgcn_perm(hi2, lo2, 0x07060302u);
    };
    issue(0);
    flush();                                                              // the stream header is 139 bits
    for (int t0 = 0; t0 < nmax; t0 += ES_TILE) {
        vlx_wave_lds_sync();
        if (!hi_mask) {                                                   // the usual wave: every stream fits int16
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int4 u, v;
                sext2(nlo[r].x, u.x, u.y); sext2(nlo[r].y, u.z, u.w); sext2(nlo[r].z, v.x, v.y); sext2(nlo[r].w, v.z, v.w);
                *(int4 *)&tile[(lrow + 16 * r) * ES_STRIDE + lcol] = u;
                *(int4 *)&tile[(lrow + 16 * r) * ES_STRIDE + lcol + 4] = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = lrow + 16 * r;
                int4 u, v;
                if ((hi_mask >> row) & 1ull) {
                    join2(nlo[r].x, nhi[r].x, u.x, u.y); join2(nlo[r].y, nhi[r].y, u.z, u.w);
                    join2(nlo[r].z, nhi[r].z, v.x, v.y); join2(nlo[r].w, nhi[r].w, v.z, v.w);
                } else {
                    sext2(nlo[r].x, u.x, u.y); sext2(nlo[r].y, u.z, u.w); sext2(nlo[r].z, v.x, v.y); sext2(nlo[r].w, v.z, v.w);
                }
                *(int4 *)&tile[row * ES_STRIDE + lcol] = u;
                *(int4 *)&tile[row * ES_STRIDE + lcol + 4] = v;
            }
        }
        vlx_wave_lds_sync();
        issue(t0 + ES_TILE);
        // 16 samples per trip: one turn of the predictor's ring (VlxLms), so that every phase is a constant
        auto quad = [&](const int i, auto q) {
            constexpr int Q = decltype(q)::value;
            const int4 rv = *(const int4 *)&tile[lane * ES_STRIDE + i + Q];
            step(rv.x, t0 + i + Q, std::integral_constant<int, Q>{}); step(rv.y, t0 + i + Q + 1, std::integral_constant<int, Q + 1>{});
            step(rv.z, t0 + i + Q + 2, std::integral_constant<int, Q + 2>{}); step(rv.w, t0 + i + Q + 3, std::integral_constant<int, Q + 3>{});
            check_ovf();
            flush();
        };
        static_assert(ES_TILE % 16 == 0, "whole turns of the ring per tile");
#pragma unroll 1
        for (int i = 0; i < ES_TILE; i += 16) {
            quad(i, std::integral_constant<int, 0>{}); quad(i, std::integral_constant<int, 4>{});
            quad(i, std::integral_constant<int, 8>{}); quad(i, std::integral_constant<int, 12>{});
        }
    }
    if (valid) {
        uint32_t bits = 0;
        if (!(flags & SF_WIDE) && !redo) {
            if (G) for (; foff < boff; foff += 4u)                         // what is left in the ring: fewer than G dwords
                if (foff < bend) *(uint32_t *)(wbase + foff) = ring[(((foff - bstart) >> 2) & (ES_RING - 1)) * 64];
            if (boff < bend) { *(uint32_t *)(wbase + boff) = lo; *(uint32_t *)(wbase + boff + 4) = 0u; }
            bits = (boff - bstart) * 8u + nacc;
        }
        ws.stream_bits[s] = bits;
        if (redo) ws.params[s].flags = flags | SF_WIDE;                   // k_encode_wide re-encodes it from the PCM
    }
}

// ----------------------------------------------------------------------------------------------
// Generic 64-bit path for streams flagged SF_WIDE (values beyond the 32-bit tiers; only reachable
// through the int64 boundary).  One lane per stream, straight from the input samples.
template <int FMT>
__global__ __launch_bounds__(256) void k_encode_wide(const uint8_t *pcm, EncWorkspace ws, uint32_t n_tracks, uint64_t n_streams)
{
    // (256 lanes per workgroup, one wave per SIMD: nearly every batch has nothing for this kernel to do, and what it then
    // costs is the dispatch of its workgroups, ~6 ns each — cfg2: 2,637 one-wave workgroups 0.017 ms)
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_streams) return;
    if (!(ws.params[s].flags & SF_WIDE)) return;     // one word per stream: the whole record only for the streams that stay
    StreamParams P = ws.params[s];
    const uint64_t c = s >> 1; const int which = (int)(s & 1);
    const uint32_t t = ws.chunk_track[c];
    const uint64_t start = (c - ws.tracks[t].chunk_base) * VLX_SUB_BLOCK;
    const uint8_t *base = pcm + ws.tracks[t].byte_off + start * Fmt<FMT>::BYTES;
    const bool hires = (ws.tracks[t].high_res & 0x100u) ? (ws.tracks[t].high_res & 1u) : (ws.track_hires[t * VLX_HIRES_STRIDE] != 0);
    const bool use_ms = ws.chunk_use_ms[c] != 0;
    const int n = P.n; const int shl = P.shift_lsb;
    auto X = [&](int j) -> int64_t {                 // stream value after M/S, >>8, >>lsb
        int64_t l = vlx_load_sample<FMT>(base, 2 * (uint64_t)j), r = vlx_load_sample<FMT>(base, 2 * (uint64_t)j + 1);
        int64_t v = which ? (use_ms ? (l - r) : r) : (use_ms ? ((l + r) >> 1) : l);
        if (hires) v >>= 8;
        return v >> shl;
    };
    // LPC analysis exactly as written (VeloxCore.h:44-66), sequential double sums
    double ac[VLX_LPC_ORDER + 1];
    for (int lag = 0; lag <= VLX_LPC_ORDER; lag++) {
        double sum = 0;
        for (int j = lag; j < n; j++) sum += (double)X(j) * (double)X(j - lag);
        ac[lag] = sum;
    }
    int32_t cf[VLX_LPC_ORDER]; int32_t sh = 0;
    vlx_levinson(ac, cf, &sh);
    LaneBits bw; bw.init(ws.staging + s * VLX_SLOT_DWORDS);
    bw.put(0u, 1); bw.put((uint32_t)shl & 31u, 5); bw.put((uint32_t)sh & 31u, 5);
    for (int j = 0; j < VLX_LPC_ORDER; j++) bw.put((uint32_t)cf[j] & 0xFFFFu, 16);
    int32_t h[VLX_TAPS], w[VLX_TAPS];
    for (int i = 0; i < VLX_TAPS; i++) { h[i] = 0; w[i] = 0; }
    int64_t hist[VLX_LPC_ORDER];
    for (int i = 0; i < VLX_LPC_ORDER; i++) hist[i] = 0;
    uint64_t run_avg = 512;
    for (int i = 0; i < n; i++) {
        const int64_t x = X(i);
        uint64_t sum = 0;                               // wraps like the reference's int64 (x86, -fwrapv behaviour)
        for (int j = 0; j < VLX_LPC_ORDER; j++) sum += (uint64_t)(int64_t)cf[j] * (uint64_t)hist[j];
        const int32_t pred_lpc = (int32_t)((int64_t)sum >> sh);
        const int64_t res = (int64_t)((uint64_t)x - (uint64_t)(int64_t)pred_lpc);
        int64_t ns = 0;
        for (int tt = 0; tt < VLX_TAPS; tt++) ns += (int64_t)h[tt] * w[tt];
        const int32_t pred = (int32_t)(ns >> 11);
        const int64_t fin = (int64_t)((uint64_t)res - (uint64_t)(int64_t)pred);
        const uint64_t m = vlx_zigzag(fin);
        const int k = 63 - __builtin_clzll(run_avg);
        vlx_rice_put(bw, m, k);
        const int32_t actual = (int32_t)res;
        const int32_t err = (int32_t)((uint32_t)actual - (uint32_t)pred);
        if (err != 0) {
            const int32_t aerr = err < 0 ? (int32_t)(0u - (uint32_t)err) : err;
            const int32_t sd = (err > 0 ? 1 : -1) * ((aerr > 1024) ? 16 : 4);
            for (int tt = 0; tt < VLX_TAPS; tt++) { const int hs = (h[tt] > 0) - (h[tt] < 0); w[tt] += hs * sd; }
            w[0] -= (w[0] > 0) - (w[0] < 0);
            w[8] -= (w[8] > 0) - (w[8] < 0);
            for (int tt = VLX_TAPS - 1; tt > 0; tt--) h[tt] = h[tt - 1];
            h[0] = actual;
        }
        for (int j = VLX_LPC_ORDER - 1; j > 0; j--) hist[j] = hist[j - 1];
        hist[0] = x;
        run_avg = run_avg - (run_avg >> 3) + (m >> 3);
        if (run_avg < 1) run_avg = 1;
    }
    ws.stream_bits[s] = bw.finish();
    P.lpc_shift = sh;
    for (int j = 0; j < VLX_LPC_ORDER; j++) P.coefs[j] = cf[j];
    ws.params[s] = P;
}

// ----------------------------------------------------------------------------------------------
#define VLX_DISPATCH_FMT(fmt, CALL)                                  \
    switch (fmt) {                                                   \
    case VLX_FMT_S16: { constexpr int F = VLX_FMT_S16; CALL; } break; \
    case VLX_FMT_S24: { constexpr int F = VLX_FMT_S24; CALL; } break; \
    case VLX_FMT_S32: { constexpr int F = VLX_FMT_S32; CALL; } break; \
    case VLX_FMT_I64: { constexpr int F = VLX_FMT_I64; CALL; } break; \
    default: return hipErrorInvalidValue;                            \
    }
// the batch kernels and the odd path also take the float32 input formats (vlx_encode_f32_device)
#define VLX_DISPATCH_FMT_F(fmt, CALL)                                \
    switch (fmt) {                                                   \
    case VLX_FMT_S16: { constexpr int F = VLX_FMT_S16; CALL; } break; \
    case VLX_FMT_S24: { constexpr int F = VLX_FMT_S24; CALL; } break; \
    case VLX_FMT_S32: { constexpr int F = VLX_FMT_S32; CALL; } break; \
    case VLX_FMT_I64: { constexpr int F = VLX_FMT_I64; CALL; } break; \
    case VLX_FMT_F32_SPLIT: { constexpr int F = VLX_FMT_F32_SPLIT; CALL; } break; \
    case VLX_FMT_F32_D16: { constexpr int F = VLX_FMT_F32_D16; CALL; } break; \
    case VLX_FMT_F32_D24: { constexpr int F = VLX_FMT_F32_D24; CALL; } break; \
    default: return hipErrorInvalidValue;                            \
    }

// Odd sample totals (VeloxCore.h:242-253): the WHOLE input is one stream, there is no use_MS bit,
// the LPC autocorrelation takes every 4th product once the stream is longer than 4096 (:47), and the
// raw fallback triggers at "> 5 bytes per sample".  The reference's own decoder cannot read these
// payloads, so this is encode parity only: one wave, lane 0 runs the sample-serial
// part, the other lanes stage tiles through LDS and help with the reductions.  The payload is
// written in place from bit 0: header bits, high_res, u32 size, mode bit, stream.
#define OD_TILE 2048

struct OddBits {                                                          // BitStreamWriter for one lane, dwords straight to the output
    uint32_t *dst; uint64_t cap, pos; uint64_t acc; int nacc;
    __device__ __forceinline__ void init(uint32_t *d, uint64_t c) { dst = d; cap = c; pos = 0; acc = 0; nacc = 0; }
    __device__ __forceinline__ void put(uint32_t v, int nb)               // nb in [0,32], v < 2^nb
    {
        acc |= (uint64_t)v << nacc; nacc += nb;
        if (nacc >= 32) { if (pos < cap) dst[pos] = (uint32_t)acc; pos++; acc >>= 32; nacc -= 32; }
    }
    __device__ __forceinline__ void put64(uint64_t v, int nb)             // nb in [0,64]
    {
        if (nb > 32) { put((uint32_t)v, 32); put((uint32_t)(v >> 32) & (nb >= 64 ? 0xFFFFFFFFu : ((1u << (nb - 32)) - 1u)), nb - 32); }
        else put(nb == 32 ? (uint32_t)v : ((uint32_t)v & ((1u << nb) - 1u)), nb);
    }
    __device__ __forceinline__ uint64_t bits() const { return pos * 32ull + (uint64_t)nacc; }
    __device__ __forceinline__ void flush() { if (nacc > 0 && pos < cap) dst[pos] = (uint32_t)acc; }
};

template <int FMT>
__global__ __launch_bounds__(64) void k_encode_odd(const uint8_t *pcm, uint64_t n, int force_hires, const uint8_t *hdr, uint64_t HB,
                                                   uint32_t *out, uint64_t cap_dwords, uint64_t *result)
{
    __shared__ int64_t tile[8 + OD_TILE];
    __shared__ double ac_s[VLX_LPC_ORDER + 1];
    __shared__ int32_t cf_s[VLX_LPC_ORDER + 1];
    const int lane = threadIdx.x;
    // ---- high_res over the whole input (:232-236), silence and LSB shift (:73-86)
    bool hires = force_hires > 0;
    if (force_hires < 0) {
        bool loud = false;
        for (uint64_t i = lane; i < n; i += 64) loud |= (int64_t)vlx_uabs(vlx_load_sample<FMT>(pcm, i)) > 65536ll;
        hires = __any(loud) != 0;
    }
    uint64_t mask = 0;
    for (uint64_t i = lane; i < n; i += 64) { int64_t v = vlx_load_sample<FMT>(pcm, i); if (hires) v >>= 8; mask |= vlx_uabs(v); }
    mask = vlx_wave_or(mask);
    const bool silent = mask == 0;
    int shl = 0;
    if (!silent) { const int tz = __builtin_ctzll(mask); shl = tz > 32 ? 32 : tz; }
    auto X = [&](uint64_t j) -> int64_t { int64_t v = vlx_load_sample<FMT>(pcm, j); if (hires) v >>= 8; return v >> shl; };
    // tile[8 + k] = X(t0 + k), tile[0..7] = the 8 values before t0 (zeros before the start)
    auto stage = [&](uint64_t t0) {
        __syncthreads();
        for (int k = lane; k < 8 + OD_TILE; k += 64) {
            const int64_t j = (int64_t)t0 + k - 8;
            tile[k] = (j >= 0 && (uint64_t)j < n) ? X((uint64_t)j) : 0;
        }
        __syncthreads();
    };
    // ---- ComputeLPC (:44-66): sequential double sums, one lane per lag
    const uint64_t stride = n > 4096 ? 4 : 1;
    double sum = 0;
    if (!silent) {
        for (uint64_t t0 = 0; t0 < n; t0 += OD_TILE) {
            stage(t0);
            if (lane <= VLX_LPC_ORDER) {
                const uint64_t lag = (uint64_t)lane;
                uint64_t j = lag + ((t0 > lag ? t0 - lag + stride - 1 : 0) / stride) * stride;   // first j >= t0 on this lane's grid
                const uint64_t end = t0 + OD_TILE < n ? t0 + OD_TILE : n;
                for (; j < end; j += stride) sum += (double)tile[8 + (j - t0)] * (double)tile[8 + (j - t0) - lag];
            }
        }
        if (lane <= VLX_LPC_ORDER) ac_s[lane] = sum;
        __syncthreads();
        if (lane == 0) {
            double ac[VLX_LPC_ORDER + 1]; int32_t cf[VLX_LPC_ORDER]; int32_t sh = 0;
            for (int i = 0; i <= VLX_LPC_ORDER; i++) ac[i] = ac_s[i];
            vlx_levinson(ac, cf, &sh);
            for (int i = 0; i < VLX_LPC_ORDER; i++) cf_s[i] = cf[i];
            cf_s[VLX_LPC_ORDER] = sh;
        }
        __syncthreads();
    }
    // ---- the payload, from bit 0
    OddBits bw; bw.init(out, cap_dwords);
    if (lane == 0) {
        for (uint64_t b = 0; b < HB; b += 8) { const int nb = HB - b < 8 ? (int)(HB - b) : 8; bw.put((uint32_t)hdr[b >> 3] & ((1u << nb) - 1u), nb); }
        bw.put(hires ? 1u : 0u, 1);                                       // :236
        bw.put(0u, 32);                                                   // u32 size, patched at the end (:301)
        bw.put(1u, 1);                                                    // mode bit, no use_MS (:244)
        if (silent) bw.put(1u, 1);                                        // :81
        else {
            bw.put(0u, 1); bw.put((uint32_t)shl & 31u, 5); bw.put((uint32_t)cf_s[VLX_LPC_ORDER] & 31u, 5);
            for (int j = 0; j < VLX_LPC_ORDER; j++) bw.put((uint32_t)cf_s[j] & 0xFFFFu, 16);
        }
    }
    const uint64_t body0 = HB + 1 + 32;                                   // first bit of the chunk body
    if (!silent) {
        int32_t cf[VLX_LPC_ORDER]; const int sh = cf_s[VLX_LPC_ORDER];
        for (int j = 0; j < VLX_LPC_ORDER; j++) cf[j] = cf_s[j];
        int32_t h[VLX_TAPS], w[VLX_TAPS];
        for (int i = 0; i < VLX_TAPS; i++) { h[i] = 0; w[i] = 0; }
        uint64_t run_avg = 512;
        for (uint64_t t0 = 0; t0 < n; t0 += OD_TILE) {
            stage(t0);
            if (lane == 0) {
                const int cnt = (int)(n - t0 < OD_TILE ? n - t0 : OD_TILE);
                for (int i = 0; i < cnt; i++) {
                    const int64_t x = tile[8 + i];
                    uint64_t s2 = 0;                                      // wraps like the reference's int64 on x86
                    for (int j = 0; j < VLX_LPC_ORDER; j++) s2 += (uint64_t)(int64_t)cf[j] * (uint64_t)tile[8 + i - 1 - j];
                    const int32_t pred_lpc = (int32_t)((int64_t)s2 >> sh);
                    const int64_t res = (int64_t)((uint64_t)x - (uint64_t)(int64_t)pred_lpc);
                    int64_t ns = 0;
                    for (int tt = 0; tt < VLX_TAPS; tt++) ns += (int64_t)h[tt] * w[tt];
                    const int32_t pred = (int32_t)(ns >> 11);
                    const int64_t fin = (int64_t)((uint64_t)res - (uint64_t)(int64_t)pred);
                    const uint64_t m = vlx_zigzag(fin);
                    const int k = 63 - __builtin_clzll(run_avg);
                    const uint64_t q = m >> k;
                    if (q < 64) {                                         // VeloxEntropy.h:104-111
                        uint32_t qq = (uint32_t)q;
                        if (qq >= 32) { bw.put(0xFFFFFFFFu, 32); qq -= 32; }
                        bw.put((1u << qq) - 1u, (int)qq + 1);
                        if (k > 0) bw.put64(k >= 64 ? m : (m & ((1ull << k) - 1ull)), k);
                    } else { bw.put(0xFFFFFFFFu, 32); bw.put(0xFFFFFFFFu, 32); bw.put(0u, 1); bw.put64(m, 40); }
                    const int32_t actual = (int32_t)res;
                    const int32_t err = (int32_t)((uint32_t)actual - (uint32_t)pred);
                    if (err != 0) {
                        const int32_t aerr = err < 0 ? (int32_t)(0u - (uint32_t)err) : err;
                        const int32_t sd = (err > 0 ? 1 : -1) * ((aerr > 1024) ? 16 : 4);
                        for (int tt = 0; tt < VLX_TAPS; tt++) { const int hs = (h[tt] > 0) - (h[tt] < 0); w[tt] = (int32_t)((uint32_t)w[tt] + (uint32_t)(hs * sd)); }
                        w[0] -= (w[0] > 0) - (w[0] < 0);
                        w[8] -= (w[8] > 0) - (w[8] < 0);
                        for (int tt = VLX_TAPS - 1; tt > 0; tt--) h[tt] = h[tt - 1];
                        h[0] = actual;
                    }
                    run_avg = run_avg - (run_avg >> 3) + (m >> 3);
                    if (run_avg < 1) run_avg = 1;
                }
            }
        }
        if (hires) {                                                      // low bytes after the codes (:124-126)
            for (uint64_t t0 = 0; t0 < n; t0 += OD_TILE) {
                __syncthreads();
                for (int k = lane; k < OD_TILE; k += 64) tile[8 + k] = (t0 + k < n) ? (vlx_load_sample<FMT>(pcm, t0 + k) & 0xFF) : 0;
                __syncthreads();
                if (lane == 0) {
                    const int cnt = (int)(n - t0 < OD_TILE ? n - t0 : OD_TILE);
                    for (int i = 0; i < cnt; i++) bw.put((uint32_t)tile[8 + i], 8);
                }
            }
        }
    }
    // ---- size, raw fallback (:246-251), patch the u32
    uint64_t total_bits = __shfl((long long)bw.bits(), 0);
    uint64_t bytes = (total_bits - body0 + 7) >> 3;
    const bool raw = bytes > 5 * n;                                       // uniform
    if (raw) {
        if (lane == 0) {
            bw.init(out, cap_dwords);
            for (uint64_t b = 0; b < HB; b += 8) { const int nb = HB - b < 8 ? (int)(HB - b) : 8; bw.put((uint32_t)hdr[b >> 3] & ((1u << nb) - 1u), nb); }
            bw.put(hires ? 1u : 0u, 1); bw.put(0u, 32); bw.put(0u, 1);    // mode 0 (:248)
        }
        for (uint64_t t0 = 0; t0 < n; t0 += OD_TILE) {
            __syncthreads();
            for (int k = lane; k < OD_TILE; k += 64) tile[8 + k] = (t0 + k < n) ? vlx_load_sample<FMT>(pcm, t0 + k) : 0;
            __syncthreads();
            if (lane == 0) {
                const int cnt = (int)(n - t0 < OD_TILE ? n - t0 : OD_TILE);
                for (int i = 0; i < cnt; i++) bw.put64(vlx_zigzag(tile[8 + i]) & ((1ull << 40) - 1ull), 40);   // WriteRawBlock :175-177
            }
        }
        total_bits = __shfl((long long)bw.bits(), 0);
        bytes = (total_bits - body0 + 7) >> 3;
    }
    if (lane == 0) {
        bw.put(0u, (int)(body0 + 8 * bytes - total_bits));                // the chunk's Flush padding (< 8 bits) ...
        bw.put(0u, 8);                                                    // ... and the payload's own (:305)
        bw.flush();
        __threadfence();
        const uint64_t sp = HB + 1;                                       // bit position of the u32 size
        const uint64_t d = sp >> 5; const int sh2 = (int)(sp & 31);
        const uint32_t sz = (uint32_t)bytes;
        if (d < cap_dwords) out[d] |= sz << sh2;
        if (sh2 && d + 1 < cap_dwords) out[d + 1] |= sz >> (32 - sh2);
        result[0] = (body0 + 8 * bytes + 7) >> 3;                         // payload length after the final Flush (:305)
        result[1] = (uint64_t)(raw ? 1 : 0) | ((uint64_t)(hires ? 1 : 0) << 1);
    }
}

hipError_t vlx_launch_encode_odd(hipStream_t st, const void *pcm, int fmt, uint64_t n, int force_hires, const uint8_t *hdr, uint64_t HB,
                                 uint32_t *out, uint64_t cap_dwords, uint64_t *result)
{
    VLX_DISPATCH_FMT_F(fmt, (k_encode_odd<F><<<dim3(1), dim3(64), 0, st>>>((const uint8_t *)pcm, n, force_hires, hdr, HB, out, cap_dwords, result)));
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// Size of chunk c on the wire (body + u32 prefix); leaves its ChunkInfo.
__device__ __forceinline__ uint64_t vlx_chunk_wire_bytes(const EncWorkspace &ws, uint64_t c)
{
    const uint32_t t = ws.chunk_track[c];
    const bool hires = (ws.tracks[t].high_res & 0x100u) ? (ws.tracks[t].high_res & 1u) : (ws.track_hires[t * VLX_HIRES_STRIDE] != 0);
    const StreamParams *PA = &ws.params[2 * c], *PB = &ws.params[2 * c + 1];
    const uint64_t n = (uint64_t)PA->n;
    uint64_t bits = 2;                                                     // mode, use_MS (VeloxCore.h:281)
    bits += ws.stream_bits[2 * c] + ((hires && !(PA->flags & SF_SILENT)) ? 8 * n : 0);
    bits += ws.stream_bits[2 * c + 1] + ((hires && !(PB->flags & SF_SILENT)) ? 8 * n : 0);
    uint64_t bytes = (bits + 7) >> 3;
    uint32_t raw = 0;
    if (bytes >= 10 * n) { raw = 1; bytes = (2 + 80 * n + 7) >> 3; }       // :286-292
    ChunkInfo ci; ci.bytes = (uint32_t)bytes; ci.flags = raw | (ws.chunk_use_ms[c] << 1);
    ws.chunk_info[c] = ci;
    return bytes + 4;                                                      // u32 size prefix (:301)
}

__global__ void k_chunk_layout(EncWorkspace ws, uint32_t n_tracks, uint64_t n_chunks)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) ws.chunk_off[0] = 0;
    if (c >= n_chunks) return;
    ws.chunk_off[c + 1] = vlx_chunk_wire_bytes(ws, c);
}

// In-place inclusive scan of data[1..n] by ONE workgroup of 1024 threads (data[0] stays): tiles of
// 4096 elements, 4 consecutive per thread (coalesced 32-B accesses), wave scan + LDS for the tile,
// running carry across tiles, the loads of the next tile in flight while this one is scanned.
// 84 k chunks = 21 tiles.
__global__ __launch_bounds__(1024) void k_scan_u64(uint64_t *data, uint64_t n)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t carry = data[0];
    uint64_t nv[4];                                                       // the next tile's elements: loaded one tile ahead
#pragma unroll
    for (int k = 0; k < 4; k++) nv[k] = (1 + 4ull * tid + k <= n) ? data[1 + 4ull * tid + k] : 0;
    for (uint64_t base = 1; base <= n; base += 4096) {
        const uint64_t i0 = base + 4ull * tid;
        uint64_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = nv[k];
#pragma unroll
        for (int k = 0; k < 4; k++) nv[k] = (i0 + 4096 + k <= n) ? data[i0 + 4096 + k] : 0;   // a tile this workgroup has not written yet
        v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
        uint64_t incl = v[3];                                             // inclusive scan of the per-thread totals within the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint64_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const uint64_t off = carry + woff + (incl - v[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) if (i0 + k <= n) data[i0 + k] = v[k] + off;
        if (tid == 1023) carry_s = off + v[3];
        __syncthreads();
        carry = carry_s;
    }
}

// The same scan for more than one tile, in two launches: every workgroup scans its own tile and leaves
// the tile total in totals[tile]; then every workgroup adds data[0] and the totals of the tiles in front
// of it.  84 k chunks: 2 x 3 us instead of 21 tiles one after the other (48 us).
__global__ __launch_bounds__(1024) void k_scan_tiles(uint64_t *data, uint64_t n, uint64_t *totals)
{
    __shared__ uint64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t i0 = 1 + 4096ull * blockIdx.x + 4ull * tid;
    uint64_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (i0 + k <= n) ? data[i0 + k] : 0;
    v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
    uint64_t incl = v[3];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    const uint64_t off = woff + (incl - v[3]);
#pragma unroll
    for (int k = 0; k < 4; k++) if (i0 + k <= n) data[i0 + k] = v[k] + off;
    if (tid == 1023) totals[blockIdx.x] = off + v[3];
}

__global__ __launch_bounds__(1024) void k_scan_apply(uint64_t *data, uint64_t n, const uint64_t *totals)
{
    __shared__ uint64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t part = 0;
    for (uint32_t j = tid; j < blockIdx.x; j += 1024) part += totals[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    uint64_t off = data[0];
#pragma unroll
    for (int w = 0; w < 16; w++) off += wsum[w];
    const uint64_t i0 = 1 + 4096ull * blockIdx.x + 4ull * tid;
#pragma unroll
    for (int k = 0; k < 4; k++) if (i0 + k <= n) data[i0 + k] += off;
}

__global__ void k_track_layout(EncWorkspace ws, uint64_t *track_out /*[n_tracks+1]*/, uint32_t n_tracks)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) track_out[0] = 0;
    if (t >= n_tracks) return;
    const TrackDesc *T = &ws.tracks[t];
    const uint64_t S = ws.chunk_off[T->chunk_base + T->n_chunks] - ws.chunk_off[T->chunk_base];
    const uint64_t len = ((uint64_t)T->hdr_bits + 8 * S + 7) >> 3;          // final Flush (:305)
    ws.track_len[t] = len;
    track_out[t + 1] = (len + 4 + 255) & ~255ull;                          // next track starts 256-aligned
}

// One record per chunk for k_assemble (AsmDesc, vlx_internal.h): runs after both scans, thread per chunk.
// SINGLE: a batch of one track (its payload starts at 0): thread 0 also does k_track_layout's work and the one-element
// scan behind it, two launches less.
template <bool SINGLE>
__global__ void k_asm_desc(EncWorkspace ws, uint64_t *track_out, uint64_t n_chunks)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t t = ws.chunk_track[c];
    const TrackDesc T = ws.tracks[t];
    const uint64_t lc = c - T.chunk_base;
    const bool hires = (T.high_res & 0x100u) ? (T.high_res & 1u) : (ws.track_hires[t * VLX_HIRES_STRIDE] != 0);
    if (SINGLE && c == 0) {
        const uint64_t S = ws.chunk_off[T.chunk_base + T.n_chunks] - ws.chunk_off[T.chunk_base];
        const uint64_t len = ((uint64_t)T.hdr_bits + 8 * S + 7) >> 3;      // final Flush (:305)
        ws.track_len[0] = len;
        track_out[0] = 0; track_out[1] = (len + 4 + 255) & ~255ull;
    }
    const uint64_t base_bits = SINGLE ? 0ull : 8 * track_out[t];
    const ChunkInfo ci = ws.chunk_info[c];
    const bool last = lc + 1 >= T.n_chunks;
    AsmDesc A;
    A.S = base_bits + T.hdr_bits + 8 * (ws.chunk_off[c] - ws.chunk_off[T.chunk_base]);
    A.D0 = (lc == 0) ? (base_bits >> 5) : ((A.S + 31) >> 5);
    A.bytes = ci.bytes;
    A.next_bytes = last ? 0u : ws.chunk_info[c + 1].bytes;
    A.bits[0] = ws.stream_bits[2 * c]; A.bits[1] = ws.stream_bits[2 * c + 1];
    A.n = (uint32_t)ws.params[2 * c].n;
    A.flags = (ci.flags & 3u) | (hires ? ADF_HIRES : 0u) | (lc == 0 ? ADF_FIRST : 0u) | (last ? ADF_LAST : 0u);
    if (hires && !(ws.params[2 * c].flags & SF_SILENT)) A.flags |= ADF_LOW_L;
    if (hires && !(ws.params[2 * c + 1].flags & SF_SILENT)) A.flags |= ADF_LOW_R;
    A.track = t;
#pragma unroll
    for (int i = 0; i < 5; i++) A.pad[i] = 0;
    ws.asm_desc[c] = A;
}

// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t vlx_bits_from_mem(const uint32_t *src, uint64_t o, int cnt)
{
    const uint64_t w = ((uint64_t)src[(o >> 5) + 1] << 32) | src[o >> 5];
    uint32_t v = (uint32_t)(w >> (o & 31));
    return cnt >= 32 ? v : (v & ((1u << cnt) - 1u));
}
__device__ __forceinline__ int64_t vlx_load_sample_rt(int fmt, const uint8_t *base, uint64_t idx)
{
    switch (fmt) {
    case VLX_FMT_S16: return vlx_load_sample<VLX_FMT_S16>(base, idx);
    case VLX_FMT_S24: return vlx_load_sample<VLX_FMT_S24>(base, idx);
    case VLX_FMT_S32: return vlx_load_sample<VLX_FMT_S32>(base, idx);
    case VLX_FMT_F32_SPLIT: return vlx_load_sample<VLX_FMT_F32_SPLIT>(base, idx);
    case VLX_FMT_F32_D16: return vlx_load_sample<VLX_FMT_F32_D16>(base, idx);
    case VLX_FMT_F32_D24: return vlx_load_sample<VLX_FMT_F32_D24>(base, idx);
    default: return vlx_load_sample<VLX_FMT_I64>(base, idx);
    }
}

// Bits of chunk c (which starts, u32 size prefix first, at absolute output bit S) that fall into
// the dword window [lo, lo+32), already positioned relative to lo.
__device__ uint32_t vlx_gather_chunk(const EncWorkspace &ws, const uint8_t *pcm_chunk, int fmt,
                                     uint64_t c, uint64_t S, bool hires, uint64_t lo)
{
    const uint64_t hi = lo + 32;
    const ChunkInfo ci = ws.chunk_info[c];
    uint32_t val = 0;
    auto imm = [&](uint64_t P, int len, uint32_t v) {
        uint64_t a = lo > P ? lo : P, b = hi < P + len ? hi : P + len;
        if (a < b) { uint32_t x = v >> (a - P); int cnt = (int)(b - a); if (cnt < 32) x &= (1u << cnt) - 1u; val |= x << (a - lo); }
    };
    imm(S, 32, ci.bytes);                                                  // Write(size, 32)  VeloxCore.h:301
    const bool raw = ci.flags & 1u; const uint32_t use_ms = (ci.flags >> 1) & 1u;
    imm(S + 32, 2, (raw ? 0u : 1u) | (use_ms << 1));                       // mode, use_MS      :281 / :289
    uint64_t P = S + 34;
    const uint64_t n = (uint64_t)ws.params[2 * c].n;
    if (!raw) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const uint64_t st = 2 * c + s;
            const uint64_t nb = ws.stream_bits[st];
            {
                uint64_t a = lo > P ? lo : P, b = hi < P + nb ? hi : P + nb;
                if (a < b) val |= vlx_bits_from_mem(ws.staging + st * VLX_SLOT_DWORDS, a - P, (int)(b - a)) << (a - lo);
            }
            P += nb;
            if (hires && !(ws.params[st].flags & SF_SILENT)) {                 // low bytes after the codes (:124-126)
                const uint64_t lb = 8 * n;
                uint64_t a = lo > P ? lo : P, b = hi < P + lb ? hi : P + lb;
                if (a < b) val |= vlx_bits_from_mem((const uint32_t *)(ws.low + st * VLX_STREAM_LEN), a - P, (int)(b - a)) << (a - lo);
                P += lb;
            }
        }
    } else {                                                               // WriteRawBlock: 40-bit zig-zag fields (:175-177)
        const uint64_t end = P + 80 * n;
        uint64_t a0 = lo > P ? lo : P, b0 = hi < end ? hi : end;
        if (a0 < b0) {
            uint64_t f0 = (a0 - P) / 40, f1 = (b0 - 1 - P) / 40;
            for (uint64_t f = f0; f <= f1; f++) {
                const uint64_t j = f < n ? f : f - n;
                int64_t l = vlx_load_sample_rt(fmt, pcm_chunk, 2 * j), r = vlx_load_sample_rt(fmt, pcm_chunk, 2 * j + 1);
                int64_t v = (f < n) ? (use_ms ? ((l + r) >> 1) : l) : (use_ms ? (l - r) : r);
                const uint64_t zz = vlx_zigzag(v) & ((1ull << 40) - 1ull);
                const uint64_t FP = P + 40 * f;
                uint64_t a = lo > FP ? lo : FP, b = hi < FP + 40 ? hi : FP + 40;
                if (a < b) { uint32_t x = (uint32_t)(zz >> (a - FP)); int cnt = (int)(b - a); if (cnt < 32) x &= (1u << cnt) - 1u; val |= x << (a - lo); }
            }
        }
    }
    return val;
}

struct __attribute__((packed, aligned(4))) VlxU4 { uint32_t x, y, z, w; };   // 16-B access at a dword-aligned address

// One workgroup per chunk.  The four long segments of a compressed chunk (L codes, L low bytes,
// R codes, R low bytes) are bit-shifted copies of dword arrays: their interior output dwords are a
// funnel shift of two neighbouring source dwords, moved 16 B per lane.  Only the handful of output
// dwords that straddle a segment boundary (plus the track header, the u32 size / flag bits, and raw
// chunks) go through the generic bit gather.  Chunk c owns every output dword whose first bit lies
// in [S, E), i.e. the dword it shares with chunk c+1 is written here, with the neighbour's bits OR-ed in.
__global__ __launch_bounds__(256) void k_assemble(const uint8_t *pcm, int fmt, EncWorkspace ws, const uint64_t *track_out,
                                                  uint32_t n_tracks, uint8_t *out)
{
    const uint64_t c = blockIdx.x;
    const AsmDesc A = ws.asm_desc[c];                                     // the one descriptor load of the common case
    const bool hires = (A.flags & ADF_HIRES) != 0, first_chunk = (A.flags & ADF_FIRST) != 0, last_chunk = (A.flags & ADF_LAST) != 0;
    const uint64_t S = A.S, D0 = A.D0;
    const uint64_t E = S + 32 + 8 * (uint64_t)A.bytes;
    uint32_t *out32 = (uint32_t *)out;
    // any owned dword, from first principles: only raw chunks and the header dwords of a track's first chunk come
    // here, and only they pay for the track descriptor
    auto generic = [&](uint64_t D) {
        const uint32_t t = A.track;
        const TrackDesc T = ws.tracks[t];
        const uint64_t lc = c - T.chunk_base;
        const uint64_t base_bits = 8 * track_out[t];
        const int bps = vlx_fmt_bytes(fmt);
        const uint8_t *pc = pcm + T.byte_off + lc * VLX_SUB_BLOCK * bps;
        const uint64_t lo = D << 5;
        uint32_t val = vlx_gather_chunk(ws, pc, fmt, c, S, hires, lo);
        if (lo + 32 > E && !last_chunk) val |= vlx_gather_chunk(ws, pc + VLX_SUB_BLOCK * bps, fmt, c + 1, E, hires, lo);
        if (lc == 0 && lo < S) {                                          // is_float [float_mode, RLE] high_res (:226-236)
            const uint64_t hb = T.hdr_bits - 1;                           // bits held in hdr_bytes
            const uint64_t a = lo, b = (lo + 32 < base_bits + hb) ? lo + 32 : base_bits + hb;
            if (a < b) val |= vlx_bits_from_mem((const uint32_t *)(ws.hdr_bytes + T.hdr_byte_off), a - base_bits, (int)(b - a));
            const uint64_t hp = base_bits + hb;                           // the high_res bit
            if (hp >= lo && hp < lo + 32 && hires && !(T.high_res & 0x200u)) val |= 1u << (hp - lo);   // 0x200: continuation shard
        }
        out32[D] = val;
    };
    if (A.flags & ADF_RAW) {                                              // raw chunk: rare, all generic
        for (uint64_t D = D0 + threadIdx.x; (D << 5) < E; D += blockDim.x) generic(D);
        return;
    }
    // ---- segment table (uniform across the workgroup)
    uint64_t sb[5]; const uint32_t *src[4];
    sb[0] = S + 34;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const uint64_t st = 2 * c + s;
        sb[2 * s + 1] = sb[2 * s] + A.bits[s];
        sb[2 * s + 2] = sb[2 * s + 1] + ((A.flags & (s ? ADF_LOW_R : ADF_LOW_L)) ? 8ull * A.n : 0ull);
        src[2 * s] = ws.staging + st * VLX_SLOT_DWORDS;
        src[2 * s + 1] = (const uint32_t *)(ws.low + st * VLX_STREAM_LEN);
    }
    // ---- dwords that hold a region boundary (and the header dwords of the track's first chunk).
    // Everything such a dword can contain is already in registers — the size prefix, the two mode bits,
    // the segment table, and of the next chunk at most 31 bits of ITS size prefix — so apart from the
    // track's very first dwords (stream header bits) no descriptor is loaded again.
    const uint32_t next_bytes = A.next_bytes;
    const uint32_t mode_bits = 1u | (((A.flags >> 1) & 1u) << 1);
    auto boundary = [&](uint64_t D) {
        const uint64_t lo = D << 5, hi = lo + 32;
        if (first_chunk && lo < S) { generic(D); return; }
        uint32_t val = 0;
        auto imm = [&](uint64_t P, int len, uint32_t v) {
            const uint64_t a = lo > P ? lo : P, b = hi < P + len ? hi : P + len;
            if (a < b) { uint32_t x = v >> (a - P); const int cnt = (int)(b - a); if (cnt < 32) x &= (1u << cnt) - 1u; val |= x << (a - lo); }
        };
        imm(S, 32, A.bytes);                                              // Write(size, 32)  VeloxCore.h:301
        imm(S + 32, 2, mode_bits);                                        // mode, use_MS      :281
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t a = lo > sb[k] ? lo : sb[k], b = hi < sb[k + 1] ? hi : sb[k + 1];
            if (a < b) val |= vlx_bits_from_mem(src[k], a - sb[k], (int)(b - a)) << (a - lo);
        }
        if (hi > E && !last_chunk) imm(E, 32, next_bytes);                // the neighbour's share of the seam dword
        out32[D] = val;
    };
    if (first_chunk) {
        uint64_t Dh = D0;                                                 // 0x400: the bulk of a long header is already in place
        if ((ws.tracks[A.track].high_res & 0x400u) && (S >> 5) > D0 + 2) Dh = (S >> 5) - 2;
        for (uint64_t D = Dh + threadIdx.x; D < (S >> 5); D += blockDim.x) generic(D);
    }
    if (threadIdx.x < 8) {
        const uint64_t bnd = threadIdx.x == 0 ? S : threadIdx.x == 1 ? S + 32 : threadIdx.x == 7 ? E : sb[threadIdx.x > 6 ? 4 : threadIdx.x - 2];
        const uint64_t D = bnd >> 5;
        if (D >= D0 && (D << 5) < E) boundary(D);
    }
    // ---- interiors: out[first + j] = funnel(src[j], src[j + 1]) >> sh
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t first = (sb[k] + 31) >> 5; if (first < D0) first = D0;
        const uint64_t last = sb[k + 1] >> 5;                             // exclusive
        if (last <= first) continue;
        const uint32_t count = (uint32_t)(last - first);
        const uint32_t o = (uint32_t)(32 * first - sb[k]);               // bit offset of out[first] inside the source
        const uint32_t *sp = src[k] + (o >> 5); const uint32_t sh = o & 31u;
        const uint32_t head = (uint32_t)((4 - (first & 3)) & 3);          // dwords before the first 16-B aligned one
        const uint32_t h = head < count ? head : count;
        const uint32_t body = (count - h) & ~3u;
        const uint32_t tail = count - h - body;
        uint32_t *dp = out32 + first;
        if (threadIdx.x < h + tail) {                                     // ragged ends, one dword per lane
            const uint32_t j = threadIdx.x < h ? threadIdx.x : body + threadIdx.x;
            dp[j] = (uint32_t)((((uint64_t)sp[j + 1] << 32) | sp[j]) >> sh);
        }
        for (uint32_t j = h + 4 * threadIdx.x; j < h + body; j += 4 * blockDim.x) {
            const VlxU4 a = *(const VlxU4 *)(sp + j); const uint32_t e = sp[j + 4];
            uint4 r;
            r.x = (uint32_t)((((uint64_t)a.y << 32) | a.x) >> sh); r.y = (uint32_t)((((uint64_t)a.z << 32) | a.y) >> sh);
            r.z = (uint32_t)((((uint64_t)a.w << 32) | a.z) >> sh); r.w = (uint32_t)((((uint64_t)e << 32) | a.w) >> sh);
            *(uint4 *)(dp + j) = r;
        }
    }
}

// Multi-GPU gather (vlx_node_gather_payload): the one byte two neighbouring shards share is the OR of the
// predecessor's last byte (already received into place) and the shard's own first byte.
__global__ void k_or_seams(uint8_t *full, const uint64_t *off, const uint32_t *first_bytes, uint32_t world)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < world) full[off[r]] |= (uint8_t)first_bytes[r];
}
hipError_t vlx_launch_or_seams(hipStream_t st, uint8_t *full, const uint64_t *off, const uint32_t *first_bytes, uint32_t world)
{
    k_or_seams<<<dim3((world + 63) / 64), dim3(64), 0, st>>>(full, off, first_bytes, world);
    return hipGetLastError();
}

// Multi-GPU offset table (vlx_node_encode_track): this rank's row of the all-gather, [head, high_res, sizes..., 0...],
// straight from k_chunk_layout's records and the (all-reduced) flag word: the sizes never visit the host on the way.
__global__ void k_node_row(const ChunkInfo *ci, uint32_t count, uint32_t head, const uint32_t *flag, uint32_t *row, uint32_t bound)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { row[0] = head; row[1] = flag ? *flag : 0u; }
    if (i < bound) row[2 + i] = i < count ? ci[i].bytes : 0u;
}
hipError_t vlx_launch_node_row(hipStream_t st, const ChunkInfo *ci, uint32_t count, uint32_t head, const uint32_t *d_flag, uint32_t *row, uint32_t bound)
{
    k_node_row<<<dim3((bound + 255) / 256 + 1), dim3(256), 0, st>>>(ci, count, head, d_flag, row, bound);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
hipError_t vlx_launch_hires_scan(hipStream_t st, const void *pcm, int fmt, const EncWorkspace &ws,
                                 uint32_t n_tracks, uint64_t n_chunks)
{
    if (!n_chunks || fmt == VLX_FMT_S16) return hipSuccess;     // |int16| can never exceed 65536
    // long tracks (the early exit applies): few workgroups that walk and skip to the next track once the answer is
    // known; corpora of short tracks scan everything: one workgroup per chunk (cfg3 2.05 ms, HBM-bound).  Two workgroups
    // per CU, measured with one barrier per chunk in the kernel (grid cap 8192 / 2048 / 1024 / 512 / 256): cfg2, found in
    // the first chunks, 0.098 / 0.060 / 0.041 / 0.026 / 0.016 ms (what is left is every workgroup's first chunk); an hour
    // of 24-bit material that never gets loud, all 2 GB scanned, 0.376 / 0.401 / 0.378 / 0.388 / 0.566 ms.
    const uint64_t per_track = n_chunks / (n_tracks ? n_tracks : 1);
    if (per_track <= 512 && n_tracks >= 256) {                            // a corpus: a workgroup per track, gone at the track's first loud chunk
        VLX_DISPATCH_FMT_F(fmt, (k_hires_scan_tracks<F><<<dim3(std::min<uint32_t>(n_tracks, 16384u)), dim3(256), 0, st>>>((const uint8_t *)pcm, ws, n_tracks)));
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)(per_track > 512 ? std::min<uint64_t>(n_chunks, 512) : n_chunks);
    VLX_DISPATCH_FMT_F(fmt, (k_hires_scan<F><<<dim3(grid), dim3(256), 0, st>>>((const uint8_t *)pcm, ws, n_tracks, n_chunks)));
    return hipGetLastError();
}

hipError_t vlx_launch_analyze(hipStream_t st, const void *pcm, int fmt, const EncWorkspace &ws,
                              uint32_t n_tracks, uint64_t n_chunks)
{
    if (!n_chunks) return hipSuccess;
    const bool seq = ws.seq_cap != 0 && vlx_fmt_bytes(fmt) >= 4;      // (the slot counter was cleared by k_enc_setup)
    VLX_DISPATCH_FMT_F(fmt, (k_analyze<F><<<dim3((unsigned)n_chunks), dim3(AN_THREADS), 0, st>>>((const uint8_t *)pcm, ws, n_tracks)));
    if (seq) {                                                            // (both find nothing to do on quiet material: the slot count stays 0)
        k_seq_chains<<<dim3((2u * ws.seq_cap + SC_ROWS - 1) / SC_ROWS), dim3(256), 0, st>>>(ws);
        VLX_DISPATCH_FMT_F(fmt, (k_analyze_finish<F><<<dim3(ws.seq_cap), dim3(AN_THREADS), 0, st>>>(ws)));
    }
    return hipGetLastError();
}

// Shape of the sample-serial encode kernel, process-wide (vlx_set_encode_shape): by batch size, or forced.
static std::atomic<int> g_encode_shape{0};
static std::atomic<int> g_encode_store{4};                                 // k_encode_serial<G>: 0 direct stores, 4 / 8 LDS ring with 16 / 32-byte flushes
extern "C" void vlx_debug_set_encode_store(int g) { g_encode_store.store(g == 4 || g == 8 ? g : 0, std::memory_order_relaxed); }   // measurement hook (tools/), not in the public header
void vlx_encode_shape_set(int shape) { g_encode_shape.store(shape, std::memory_order_relaxed); }

// Name of the kernel vlx_launch_encode_serial launches for a batch of this size (the label of its profiling slot).
const char *vlx_encode_kernel_label(uint64_t n_streams)
{
    const int shape = g_encode_shape.load(std::memory_order_relaxed);
    const bool pipe = shape == VLX_SHAPE_PIPE16 || shape == VLX_SHAPE_PIPE32 || shape == VLX_SHAPE_PIPE64 || shape == VLX_SHAPE_PIPE ||
                      (shape == VLX_SHAPE_AUTO && n_streams <= VLX_EPIPE_MAX_STREAMS);
    return pipe ? "k_encode_pipe" : "k_encode_serial";
}

hipError_t vlx_launch_encode_serial(hipStream_t st, const EncWorkspace &ws, uint64_t n_streams)
{
    if (!n_streams) return hipSuccess;
    const int shape = g_encode_shape.load(std::memory_order_relaxed);
    if (shape == VLX_SHAPE_PIPE16 || shape == VLX_SHAPE_PIPE32 || shape == VLX_SHAPE_PIPE64)
        return vlx_launch_encode_pipe(st, ws, n_streams, shape == VLX_SHAPE_PIPE16 ? 16 : (shape == VLX_SHAPE_PIPE32 ? 32 : 64));
    if (shape == VLX_SHAPE_PIPE || (shape == VLX_SHAPE_AUTO && n_streams <= VLX_EPIPE_MAX_STREAMS)) return vlx_launch_encode_pipe(st, ws, n_streams, 0);
    const dim3 grid((unsigned)((n_streams + 64 * ES_WAVES - 1) / (64 * ES_WAVES))), block(64 * ES_WAVES);
    switch (g_encode_store.load(std::memory_order_relaxed)) {
    case 4: k_encode_serial<4><<<grid, block, 0, st>>>(ws, n_streams); break;
    case 8: k_encode_serial<8><<<grid, block, 0, st>>>(ws, n_streams); break;
    default: k_encode_serial<0><<<grid, block, 0, st>>>(ws, n_streams); break;
    }
    return hipGetLastError();
}

hipError_t vlx_launch_encode_wide(hipStream_t st, const void *pcm, int fmt, const EncWorkspace &ws,
                                  uint32_t n_tracks, uint64_t n_chunks)
{
    if (!n_chunks) return hipSuccess;
    const uint64_t n_streams = 2 * n_chunks;
    VLX_DISPATCH_FMT_F(fmt, (k_encode_wide<F><<<dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, st>>>((const uint8_t *)pcm, ws, n_tracks, n_streams)));
    return hipGetLastError();
}

// In-place inclusive scan of data[1..n] (data[0] is the seed).  The buffer must have room for
// VLX_SCAN_ROOM(n) entries: the tile totals of the two-launch form live behind the data.
hipError_t vlx_launch_scan_u64(hipStream_t st, uint64_t *data, uint64_t n)
{
    if (!n) return hipSuccess;
    if (n <= 4096) k_scan_u64<<<dim3(1), dim3(1024), 0, st>>>(data, n);
    else {
        const unsigned tiles = (unsigned)((n + 4095) / 4096);
        uint64_t *totals = data + n + 1;
        k_scan_tiles<<<dim3(tiles), dim3(1024), 0, st>>>(data, n, totals);
        k_scan_apply<<<dim3(tiles), dim3(1024), 0, st>>>(data, n, totals);
    }
    return hipGetLastError();
}

// track_out lives right behind track_len in the workspace (see vlx_api.cpp): [n_tracks+1] u64
hipError_t vlx_launch_layout(hipStream_t st, const EncWorkspace &ws, uint32_t n_tracks, uint64_t n_chunks)
{
    uint64_t *track_out = ws.track_len + n_tracks;
    // (forming the sizes inside the first scan launch instead, one launch less: 0.023 -> 0.030 ms on cfg4 — three
    // workgroups then do all of k_chunk_layout's record loads)
    k_chunk_layout<<<dim3((unsigned)((n_chunks + 255) / 256 + 1)), dim3(256), 0, st>>>(ws, n_tracks, n_chunks);
    hipError_t e = vlx_launch_scan_u64(st, ws.chunk_off, n_chunks);
    if (e != hipSuccess) return e;
    if (n_tracks == 1 && n_chunks) {
        k_asm_desc<true><<<dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st>>>(ws, track_out, n_chunks);
        return hipGetLastError();
    }
    k_track_layout<<<dim3((n_tracks + 255) / 256), dim3(256), 0, st>>>(ws, track_out, n_tracks);
    e = vlx_launch_scan_u64(st, track_out, n_tracks);
    if (e != hipSuccess) return e;
    if (n_chunks) k_asm_desc<false><<<dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st>>>(ws, track_out, n_chunks);
    return hipGetLastError();
}

hipError_t vlx_launch_assemble(hipStream_t st, const void *pcm, int fmt, const EncWorkspace &ws,
                               uint32_t n_tracks, uint64_t n_chunks, uint8_t *out)
{
    if (!n_chunks) return hipSuccess;
    const uint64_t *track_out = ws.track_len + n_tracks;
    k_assemble<<<dim3((unsigned)n_chunks), dim3(256), 0, st>>>((const uint8_t *)pcm, fmt, ws, track_out, n_tracks, out);
    return hipGetLastError();
}
