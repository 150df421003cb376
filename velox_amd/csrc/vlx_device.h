// Device-side helpers shared by the encode and decode kernels (gfx950 only).
#pragma once
#include "vlx_internal.h"

#define VLX_WAVE 64

__device__ __forceinline__ uint64_t vlx_zigzag(int64_t n) { return ((uint64_t)n << 1) ^ (uint64_t)(n >> 63); }   // VeloxEntropy.h:95
__device__ __forceinline__ int64_t  vlx_dezigzag(uint64_t n) { return (int64_t)((n >> 1) ^ (0ull - (n & 1ull))); } // VeloxEntropy.h:96
__device__ __forceinline__ uint64_t vlx_uabs(int64_t v) { return v < 0 ? (0ull - (uint64_t)v) : (uint64_t)v; }

// Largest t with tracks[t].chunk_base <= c (tracks with zero chunks share a base with their successor).
__device__ __forceinline__ uint32_t vlx_find_track(const TrackDesc *tr, uint32_t n_tracks, uint64_t c)
{
    uint32_t lo = 0, hi = n_tracks;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (tr[mid].chunk_base <= c) lo = mid; else hi = mid;
    }
    return lo;
}

template <typename T> __device__ __forceinline__ T vlx_wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <typename T> __device__ __forceinline__ T vlx_wave_or(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}
template <typename T> __device__ __forceinline__ T vlx_wave_max(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o); v = u > v ? u : v; }
    return v;
}

template <typename T> __device__ __forceinline__ T vlx_wave_min(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o); v = u < v ? u : v; }
    return v;
}

// One interleaved sample of any supported container, widened to int64 (FormatHandler::BytesToSamples,
// VeloxFormat.h:130-158, fused into the load).  Scalar form for tails and rare paths.
template <int FMT> __device__ __forceinline__ int64_t vlx_load_sample(const uint8_t *base, uint64_t idx)
{
    if (FMT == VLX_FMT_S16) return (int64_t) * (const int16_t *)(base + idx * 2);
    if (FMT == VLX_FMT_S24) {
        const uint8_t *p = base + idx * 3;
        uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
        return (int64_t)((int32_t)(u << 8) >> 8);
    }
    if (FMT == VLX_FMT_S32) return (int64_t) * (const int32_t *)(base + idx * 4);
    return *(const int64_t *)(base + idx * 8);
}

// Loads and stores through pointers that hipcc can no longer prove to be global (struct members, char
// arithmetic) come out as flat_* instructions, and those count on lgkmcnt as well as vmcnt: every later
// wait for an LDS read then also waits for the HBM access.  In the serial kernels, where a per-lane LDS ring
// is read every sample, that put a full memory latency behind each refill.  These helpers pin the address space.
typedef uint32_t vlx_v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) vlx_v4u vlx_g_v4u;
__device__ __forceinline__ uint4 vlx_gload16(const void *p)
{
    const vlx_v4u t = *(const vlx_g_v4u *)p;
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void vlx_gstore16(void *p, const uint4 v)
{
    vlx_v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(vlx_g_v4u *)p = t;
}

// the same store for an address that is only known to be 4-byte aligned (global memory takes it)
typedef vlx_v4u vlx_v4u_a4 __attribute__((aligned(4)));
typedef __attribute__((address_space(1))) vlx_v4u_a4 vlx_g_v4u_a4;
__device__ __forceinline__ void vlx_gstore16_a4(void *p, const uint4 v)
{
    vlx_v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(vlx_g_v4u_a4 *)p = t;
}

__device__ __forceinline__ void vlx_wave_lds_sync()
{   // LDS traffic of ONE wave is processed in issue order; this only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// c + (int64)a * b in ONE v_mad_i64_i32.  Left to itself hipcc sometimes sign-extends a reused int32
// operand once and then multiplies 64 x 32 (v_mad_u64_u32 + 2 v_mul_lo_u32 + v_add3 per product); the
// empty asm statements make every use an opaque 32-bit value, so the extension cannot be shared.
// (Writing the instruction itself as inline asm works too, but hipcc pads every such asm with an s_nop.)
__device__ __forceinline__ int64_t vlx_mad64(int32_t a, int32_t b, int64_t c)
{
    asm("" : "+v"(a));
    asm("" : "+v"(b));
    return c + (int64_t)a * (int64_t)b;
}

__device__ __forceinline__ int32_t vlx_sign(int32_t v)                  // -1 / 0 / +1 in ONE instruction (hipcc emits 4 for the C form)
{
    int32_t d;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(d) : "v"(v));
    return d;
}

// NeuralPredictor (VeloxCore.h:16-41): 12-tap sign-sign LMS, all state in registers.  The signs of
// the history ride along in their own registers so that Update is one v_mad_i32_i24 per tap.
struct VlxLms {
    // (h, sign(h)) as one 64-bit pair per tap with v_mov_b64 shifts was measured: no faster, a
    // 64-bit move issues at half rate.
    int32_t h[VLX_TAPS], sg[VLX_TAPS], w[VLX_TAPS];
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int i = 0; i < VLX_TAPS; i++) { h[i] = 0; sg[i] = 0; w[i] = 0; }
    }
    __device__ __forceinline__ int32_t predict() const                   // Predict(), :22-26
    {
        int64_t sum = 0;
#pragma unroll
        for (int t = 0; t < VLX_TAPS; t++) sum += (int64_t)h[t] * w[t];
        return (int32_t)(sum >> 11);
    }
    // Same sum over two independent accumulator chains: half the dependent-MAC latency for a wave
    // that has its SIMD to itself (the decode kernel's chunk waves), one more 64-bit add.
    __device__ __forceinline__ int32_t predict2() const
    {
        int64_t s0 = 0, s1 = 0;
#pragma unroll
        for (int t = 0; t < VLX_TAPS; t += 2) { s0 += (int64_t)h[t] * w[t]; s1 += (int64_t)h[t + 1] * w[t + 1]; }
        return (int32_t)((s0 + s1) >> 11);
    }
    // Update(actual, pred), :27-40, with err = actual - pred already formed (int32, wrapping).
    // Call only when err != 0 (the reference returns before touching anything otherwise, :30).
    __device__ __forceinline__ void update(int32_t err, int32_t actual)
    {
        const int32_t ne = (int32_t)(0u - (uint32_t)err);
        const int32_t aerr = err > ne ? err : ne;                         // std::abs as x86 computes it (INT_MIN stays)
        const int32_t mag = aerr > 1024 ? 16 : 4;                         // :31
        const int32_t sgn = err >> 31;
        const int32_t sd = (mag ^ sgn) - sgn;
#pragma unroll
        for (int t = 0; t < VLX_TAPS; t++) w[t] += __mul24(sg[t], sd);    // :32-35 (|w| <= 17 * 4096)
        w[0] -= vlx_sign(w[0]);                                           // :36 leak on taps 0 and 8
        w[8] -= vlx_sign(w[8]);
#pragma unroll
        for (int t = VLX_TAPS - 1; t > 0; t--) { h[t] = h[t - 1]; sg[t] = sg[t - 1]; }   // :38-39
        h[0] = actual; sg[0] = vlx_sign(actual);
    }
};
