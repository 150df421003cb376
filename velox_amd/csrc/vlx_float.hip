// Float front-end of the Velox frame codec on the device, hand-written HIP for gfx950 (wave64): what
// the reference runs serially on the calling thread before and after the integer codec when the
// container holds IEEE float32 samples.
//
//   k_f32_detect    FormatHandler::DetectPseudoFloat (VeloxFormat.h:14-58): "every sampled value is k / 2^15
//                   (or k / 2^23)" as an AND-reduction over every sample (every 4th beyond 100,000)
//   k_f32_demote    FormatHandler::DemoteFloatToInt (VeloxFormat.h:60-70): round(f * 2^15 | 2^23) -> int64
//   k_f32_exps      FormatHandler::SplitFloat32's exponent plane (VeloxFormat.h:84-105); the mantissas — and, for
//                   the device-resident entry point, the demoted integers — are formed inside the encode kernels'
//                   sample loads (VLX_FMT_F32_SPLIT / _D16 / _D24, vlx_device.h: vlx_from_f32)
//   k_rle_*         VeloxCodec::EncodeRLE (VeloxCore.h:187-195): (run <= 255, value) byte pairs of the
//                   exponent plane, as a two-level scan; k_float_header shifts them behind the three
//                   stream-header bits (is_float, float_mode; VeloxCore.h:226-229)
//   k_rle_expand    VeloxCodec::DecodeRLE (VeloxCore.h:196-203)
//   (FormatHandler::MergeFloat32 / PromoteIntToFloat, VeloxFormat.h:72-82, :107-128, are output formats of
//    k_decode_finalize in vlx_decode.hip: VLX_FMT_F32_MERGE / _P16 / _P24)
//
// Bit-exactness notes.  The reference is x86-64 code: a float -> int conversion that does not fit gives
// INT_MIN (cvttss2si / cvttsd2si), where the GPU's conversion saturates, so the conversions are written out.
// Comparisons of floats are done on their bit patterns (both sides are never zero together), which keeps
// them independent of the wave's denormal mode.  Everything else is exact scaling by powers of two.
#include <algorithm>
#include "vlx_device.h"

#pragma clang fp contract(off)

#define FL_THREADS 256
#define FL_PER     16
#define FL_BLOCK   (FL_THREADS * FL_PER)      // samples per workgroup in the RLE kernels

__device__ __forceinline__ int32_t vlx_x86_cvt_i32(float r)               // (int32_t)r as cvttss2si does it
{
    if (!(r < 2147483648.0f) || r < -2147483648.0f) return (int32_t)0x80000000u;    // out of range or NaN
    return (int32_t)r;
}
// flags[0] = every tested value fits 16 bits, flags[1] = ... 24 bits; both start at 1, a thread that finds a
// counter-example stores 0 (the sequential loop's early return changes nothing about the answer).
__global__ __launch_bounds__(256) void k_f32_detect(const uint32_t *raw, uint64_t count, uint64_t stride, uint32_t *flags)
{
    bool fit16 = true, fit24 = true;
    const uint64_t n_tested = (count + stride - 1) / stride;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_tested; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = raw[j * stride];
        if ((u & 0x7FFFFFFFu) == 0 || ((u >> 23) & 0xFFu) == 0xFFu) continue;          // f == 0.0f || !isfinite(f), VeloxFormat.h:27
        const float f = __uint_as_float(u);
        const int32_t i16 = vlx_x86_cvt_i32(roundf(f * 32768.0f));                     // :32-33
        if (__float_as_uint((float)i16 * (1.0f / 32768.0f)) != u) fit16 = false;       // :34-36
        const int32_t i24 = vlx_x86_cvt_i32(roundf(f * 8388608.0f));                   // :42-43
        if (__float_as_uint((float)i24 * (1.0f / 8388608.0f)) != u) fit24 = false;     // :44-46
    }
    if (!__all(fit16) && (threadIdx.x & 63) == 0) __hip_atomic_store(&flags[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!__all(fit24) && (threadIdx.x & 63) == 0) __hip_atomic_store(&flags[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_f32_demote(const uint32_t *raw, uint64_t count, double scale, int64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = vlx_x86_cvt_i64(round((double)__uint_as_float(raw[i]) * scale));      // VeloxFormat.h:68
}

// the exponent plane alone (the mantissas are formed inside the encode kernels' loads, VLX_FMT_F32_SPLIT)
__global__ __launch_bounds__(256) void k_f32_exps(const uint32_t *raw, uint64_t count, uint8_t *exps)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= count) {
        const uint4 u = *(const uint4 *)(raw + i);
        *(uint32_t *)(exps + i) = ((u.x >> 23) & 0xFFu) | (((u.y >> 23) & 0xFFu) << 8) | (((u.z >> 23) & 0xFFu) << 16) | (((u.w >> 23) & 0xFFu) << 24);
    } else for (uint64_t j = i; j < count; j++) exps[j] = (uint8_t)((raw[j] >> 23) & 0xFFu);
}

// ---- EncodeRLE ------------------------------------------------------------------------------------
// Sample i ENDS a (run, value) pair when the next exponent differs, when the pair has reached 255, or at the
// end of the data; its run = ((i - rs) mod 255) + 1 with rs = the position where i's run of equal values
// started.  rs needs a backward look only: an in-block max-scan of run-start positions plus, per block, the
// last run start in front of it (k_rle_last_start -> k_rle_carry).
__global__ __launch_bounds__(FL_THREADS) void k_rle_last_start(const uint8_t *e, uint64_t n, uint64_t *last_start /*[blocks]: position + 1, 0 = none*/)
{
    __shared__ uint64_t red[FL_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * FL_BLOCK;
    uint64_t best = 0;
    for (int k = 0; k < FL_PER; k++) {
        const uint64_t i = b0 + (uint64_t)k * FL_THREADS + threadIdx.x;
        if (i < n && (i == 0 || e[i] != e[i - 1])) best = i + 1 > best ? i + 1 : best;
    }
    best = vlx_wave_max(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t m = red[0]; for (int w = 1; w < FL_THREADS / 64; w++) m = red[w] > m ? red[w] : m; last_start[blockIdx.x] = m; }
}

// carry[b] = position + 1 of the last run start in blocks 0..b-1 (exclusive prefix max), one workgroup.
__global__ __launch_bounds__(1024) void k_rle_carry(const uint64_t *last_start, uint64_t n_blocks, uint64_t *carry)
{
    __shared__ uint64_t wmax[16];
    __shared__ uint64_t run_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t run = 0;
    for (uint64_t base = 0; base < n_blocks; base += 1024) {
        const uint64_t b = base + tid;
        const uint64_t v = b < n_blocks ? last_start[b] : 0;
        uint64_t inc = v;                                                 // inclusive max-scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(inc, o); if (lane >= o && u > inc) inc = u; }
        if (lane == 63) wmax[wave] = inc;
        __syncthreads();
        uint64_t pre = run;
        for (int w = 0; w < wave; w++) pre = wmax[w] > pre ? wmax[w] : pre;
        uint64_t excl = __shfl_up(inc, 1); if (lane == 0) excl = 0;
        excl = excl > pre ? excl : pre;
        if (b < n_blocks) carry[b] = excl;
        if (tid == 1023) run_s = inc > pre ? inc : pre;
        __syncthreads();
        run = run_s;
    }
}

// EMIT = false: counts[1 + block] = pair ends in the block (counts[0] = 0 is written by block 0);
// EMIT = true:  pairs are written at offs[block] + rank inside the block.
template <bool EMIT>
__global__ __launch_bounds__(FL_THREADS) void k_rle_pairs(const uint8_t *e, uint64_t n, const uint64_t *carry, uint64_t *counts_or_offs, uint8_t *rle)
{
    __shared__ uint64_t wsum[FL_THREADS / 64], wmaxs[FL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * FL_BLOCK + (uint64_t)tid * FL_PER;      // FL_PER consecutive samples per thread
    uint8_t v[FL_PER + 2];                                                             // e[i0 - 1 .. i0 + FL_PER]
#pragma unroll
    for (int k = 0; k < FL_PER + 2; k++) { const uint64_t i = i0 + k - 1; v[k] = (i0 + k >= 1 && i < n) ? e[i] : 0; }
    // last run start (position + 1) at or before each of this thread's samples: thread-local, then across the block
    uint64_t loc = 0;
#pragma unroll
    for (int k = 0; k < FL_PER; k++) { const uint64_t i = i0 + k; if (i < n && (i == 0 || v[k + 1] != v[k])) loc = i + 1; }
    uint64_t inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(inc, o); if (lane >= o && u > inc) inc = u; }
    if (lane == 63) wmaxs[wave] = inc;
    __syncthreads();
    uint64_t before = carry[blockIdx.x];
    for (int w = 0; w < wave; w++) before = wmaxs[w] > before ? wmaxs[w] : before;
    { uint64_t ex = __shfl_up(inc, 1); if (lane == 0) ex = 0; before = ex > before ? ex : before; }
    // pair ends
    uint32_t ends = 0; uint8_t runs[FL_PER];
    uint64_t rs1 = before;                                                             // run start + 1
#pragma unroll
    for (int k = 0; k < FL_PER; k++) {
        const uint64_t i = i0 + k;
        if (i < n) {
            if (i == 0 || v[k + 1] != v[k]) rs1 = i + 1;
            const uint64_t pos = i - (rs1 - 1);                                        // position inside the run of equal values
            const bool end = i + 1 == n || v[k + 2] != v[k + 1] || (pos + 1) % 255 == 0;   // VeloxCore.h:191-194
            runs[k] = (uint8_t)(pos % 255 + 1);
            if (end) ends |= 1u << k;
        } else runs[k] = 0;
    }
    const uint64_t cnt = (uint64_t)__builtin_popcount(ends);
    uint64_t sc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(sc, o); if (lane >= o) sc += u; }
    if (lane == 63) wsum[wave] = sc;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    if (!EMIT) {
        if (tid == FL_THREADS - 1) counts_or_offs[1 + blockIdx.x] = woff + sc;
        if (blockIdx.x == 0 && tid == 0) counts_or_offs[0] = 0;
    } else {
        uint64_t p = counts_or_offs[blockIdx.x] + woff + (sc - cnt);
#pragma unroll
        for (int k = 0; k < FL_PER; k++)
            if (ends & (1u << k)) { rle[2 * p] = runs[k]; rle[2 * p + 1] = v[k + 1]; p++; }
    }
}

// The stream header in front of the chunks: bits 0..2 = is_float (1), float_mode (2 bits), then the RLE bytes
// (VeloxCore.h:226-229), i.e. every RLE byte shifted up by three bits; padded with zero bytes.
__global__ __launch_bounds__(256) void k_float_header(const uint8_t *rle, uint64_t n_rle, uint32_t float_mode, uint8_t *hdr, uint64_t n_hdr)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_hdr) return;
    uint32_t b = 0;
    if (j == 0) b = 1u | (float_mode << 1);
    if (j >= 1 && j - 1 < n_rle) b |= (uint32_t)rle[j - 1] >> 5;
    if (j < n_rle) b |= ((uint32_t)rle[j] << 3) & 0xFFu;
    hdr[j] = (uint8_t)b;
}

// ---- DecodeRLE: pairs sit at a fixed stride of 16 bits from bit 3 of the payload -------------------------
// steps[1 + p] = run of pair p (steps[0] = 0); scanned in place by vlx_launch_scan_u64
__global__ __launch_bounds__(256) void k_rle_runs(const uint8_t *payload, uint64_t payload_bytes, uint64_t p0, uint64_t n_pairs, uint64_t *steps)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) steps[0] = 0;
    if (k >= n_pairs) return;
    const uint64_t bit = 3 + 16 * (p0 + k), by = bit >> 3;
    const uint32_t w = (by < payload_bytes ? payload[by] : 0u) | ((by + 1 < payload_bytes ? payload[by + 1] : 0u) << 8);   // reads past the end are 0 bits
    steps[1 + k] = (w >> 3) & 0xFFu;
}
// pair p covers samples [base + scan[p], base + scan[p + 1]); writes the exponents of the pairs that start
// below `total`; *first_done = the first pair (of this slab) at which the running count reaches `total`
__global__ __launch_bounds__(256) void k_rle_expand(const uint8_t *payload, uint64_t payload_bytes, uint64_t p0, uint64_t n_pairs, const uint64_t *scan,
                                                   uint64_t base, uint64_t total, uint8_t *exps, unsigned long long *first_done)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pairs) return;
    const uint64_t a = base + scan[k], b = base + scan[k + 1];
    if (a < total && b >= total) atomicMin(first_done, (unsigned long long)k);
    if (a >= total) return;
    const uint64_t bit = 3 + 16 * (p0 + k) + 8, by = bit >> 3;
    const uint32_t w = (by < payload_bytes ? payload[by] : 0u) | ((by + 1 < payload_bytes ? payload[by + 1] : 0u) << 8);
    const uint8_t val = (uint8_t)(w >> 3);
    for (uint64_t i = a; i < b && i < total; i++) exps[i] = val;                       // DecodeRLE pushes beyond `count`; only the first count are used
}

// ---- launchers ------------------------------------------------------------------------------------
hipError_t vlx_launch_f32_detect(hipStream_t st, const void *raw, uint64_t count, uint32_t *d_flags)
{
    const uint64_t stride = count > 100000 ? 4 : 1;                       // VeloxFormat.h:20-22
    const uint64_t tested = (count + stride - 1) / stride;
    const unsigned blocks = (unsigned)std::min<uint64_t>((tested + 255) / 256, 4096);
    if (!blocks) return hipSuccess;
    k_f32_detect<<<dim3(blocks), dim3(256), 0, st>>>((const uint32_t *)raw, count, stride, d_flags);
    return hipGetLastError();
}
hipError_t vlx_launch_f32_demote(hipStream_t st, const void *raw, uint64_t count, int bits, int64_t *out)
{
    if (!count) return hipSuccess;
    k_f32_demote<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st>>>((const uint32_t *)raw, count, bits == 16 ? 32768.0 : 8388608.0, out);
    return hipGetLastError();
}
// work: [blocks] last_start, [blocks] carry, then VLX_SCAN_ROOM(blocks) counts/offsets (u64 each)
size_t vlx_rle_work_entries(uint64_t n) { const uint64_t nb = (n + FL_BLOCK - 1) / FL_BLOCK; return (size_t)(2 * nb + VLX_SCAN_ROOM(nb) + 8); }
hipError_t vlx_launch_rle_count(hipStream_t st, const uint8_t *exps, uint64_t n, uint64_t *work)
{
    if (!n) return hipSuccess;
    const uint64_t nb = (n + FL_BLOCK - 1) / FL_BLOCK;
    uint64_t *last = work, *carry = work + nb, *counts = work + 2 * nb;
    k_rle_last_start<<<dim3((unsigned)nb), dim3(FL_THREADS), 0, st>>>(exps, n, last);
    k_rle_carry<<<dim3(1), dim3(1024), 0, st>>>(last, nb, carry);
    k_rle_pairs<false><<<dim3((unsigned)nb), dim3(FL_THREADS), 0, st>>>(exps, n, carry, counts, nullptr);
    hipError_t e = vlx_launch_scan_u64(st, counts, nb);                   // counts[b] = pairs in front of block b, counts[nb] = all pairs
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
hipError_t vlx_launch_rle_emit(hipStream_t st, const uint8_t *exps, uint64_t n, uint64_t *work, uint8_t *rle)
{
    if (!n) return hipSuccess;
    const uint64_t nb = (n + FL_BLOCK - 1) / FL_BLOCK;
    k_rle_pairs<true><<<dim3((unsigned)nb), dim3(FL_THREADS), 0, st>>>(exps, n, work + nb, work + 2 * nb, rle);
    return hipGetLastError();
}
hipError_t vlx_launch_float_header(hipStream_t st, const uint8_t *rle, uint64_t n_rle, uint32_t float_mode, uint8_t *hdr, uint64_t n_hdr)
{
    k_float_header<<<dim3((unsigned)((n_hdr + 255) / 256)), dim3(256), 0, st>>>(rle, n_rle, float_mode, hdr, n_hdr);
    return hipGetLastError();
}
hipError_t vlx_launch_rle_runs(hipStream_t st, const uint8_t *payload, uint64_t payload_bytes, uint64_t p0, uint64_t n_pairs, uint64_t *steps)
{
    if (!n_pairs) return hipSuccess;
    k_rle_runs<<<dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st>>>(payload, payload_bytes, p0, n_pairs, steps);
    return vlx_launch_scan_u64(st, steps, n_pairs);
}
hipError_t vlx_launch_rle_expand(hipStream_t st, const uint8_t *payload, uint64_t payload_bytes, uint64_t p0, uint64_t n_pairs, const uint64_t *scan,
                                 uint64_t base, uint64_t total, uint8_t *exps, unsigned long long *first_done)
{
    if (!n_pairs) return hipSuccess;
    k_rle_expand<<<dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st>>>(payload, payload_bytes, p0, n_pairs, scan, base, total, exps, first_done);
    return hipGetLastError();
}
hipError_t vlx_launch_f32_exps(hipStream_t st, const void *raw, uint64_t count, uint8_t *exps)
{
    if (!count) return hipSuccess;
    k_f32_exps<<<dim3((unsigned)((count + 1023) / 1024)), dim3(256), 0, st>>>((const uint32_t *)raw, count, exps);
    return hipGetLastError();
}
