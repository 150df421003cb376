// Device-side helpers shared by the encode and decode kernels (gfx950 only).
#pragma once
#include <type_traits>
#include <utility>
#include "vlx_internal.h"

#define VLX_WAVE 64

// gfx950 only, and said so where it matters: the wave64 lane tricks (DPP quad_perm, v_med3 signs), the hand-written
// s_waitcnt vmcnt(n) publishes of vlx_pipe.hip (vector-memory operations of a wave retire in order on the gfx9 family)
// and the workgroup-on-one-CU assumption behind them (no threadgroup-split mode: see csrc/Makefile) are not portable.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libvelox_amd.so is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__AMDGCN_CUMODE__) && __AMDGCN_CUMODE__ == 1)
#error "the intra-workgroup hand-overs of vlx_pipe.hip need all waves of a workgroup on one CU (CU mode, -mno-tgsplit)"
#endif

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

// The reference is x86-64 code: a double -> int64 conversion that does not fit gives INT64_MIN (cvttsd2si), where
// the GPU's conversion saturates.
__device__ __forceinline__ int64_t vlx_x86_cvt_i64(double r)
{
    if (!(r < 9223372036854775808.0) || r < -9223372036854775808.0) return (int64_t)0x8000000000000000ull;
    return (int64_t)r;
}
// float32 bit pattern -> the integer the codec works on (the float prologue of ProcessBlock, fused into the load)
template <int FMT> __device__ __forceinline__ int64_t vlx_from_f32(uint32_t u)
{
    if (FMT == VLX_FMT_F32_SPLIT) {                                       // SplitFloat32's mantissa, VeloxFormat.h:94-103
        uint32_t m = u & 0x7FFFFFu;
        if ((u >> 23) & 0xFFu) m |= 0x800000u;
        return (u >> 31) ? -(int64_t)m : (int64_t)m;
    }
    return vlx_x86_cvt_i64(round((double)__uint_as_float(u) * (FMT == VLX_FMT_F32_D16 ? 32768.0 : 8388608.0)));   // DemoteFloatToInt, :68
}

// One interleaved sample of any supported container, widened to int64 (FormatHandler::BytesToSamples,
// VeloxFormat.h:130-158, fused into the load).  Scalar form for tails and rare paths.
template <int FMT> __device__ __forceinline__ int64_t vlx_load_sample(const uint8_t *base, uint64_t idx)
{
    if (FMT == VLX_FMT_F32_SPLIT || FMT == VLX_FMT_F32_D16 || FMT == VLX_FMT_F32_D24) return vlx_from_f32<FMT>(*(const uint32_t *)(base + idx * 4));
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

// 8 bytes to a 4-byte aligned global address
typedef uint32_t vlx_v2u __attribute__((ext_vector_type(2)));
typedef vlx_v2u vlx_v2u_a4 __attribute__((aligned(4)));
typedef __attribute__((address_space(1))) vlx_v2u_a4 vlx_g_v2u_a4;
__device__ __forceinline__ void vlx_gstore8_a4(void *p, uint32_t a, uint32_t b)
{
    vlx_v2u t; t.x = a; t.y = b;
    *(vlx_g_v2u_a4 *)p = t;
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

// |a - b| + c on unsigned operands in ONE v_sad_u32.  (hipcc matches the C form — max minus min plus c — only in
// isolation; inside k_analyze's unrolled loop it came out as v_max_u32 + v_min_u32 + v_sub + v_add.)
__device__ __forceinline__ uint32_t vlx_sad(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t vlx_opaque_s(uint32_t v) { asm("" : "+s"(v)); return v; }

__device__ __forceinline__ int32_t vlx_sign(int32_t v)                  // -1 / 0 / +1 in ONE instruction (hipcc emits 4 for the C form)
{
    int32_t d;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(d) : "v"(v));
    return d;
}

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant
// expression inside the body (template arguments, register-array subscripts that must stay static)
template <class F, int... I> __device__ __forceinline__ void vlx_static_for_impl(F &f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void vlx_static_for(F &&f) { vlx_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// r[(P + 1 - t) & (R - 1)] = r[(P - t) & (R - 1)] for t = 0..T-1, in place (the twelve, six or three history values of a
// lane one ring register forward).  ONE asm statement with read-write operands: written as assignments, hipcc's copies
// for the values that meet again behind the branch around it landed on the common path (19 moves per sample for every
// lane, measured in the ISA) instead of in the rare block.
template <int P, int T, int R> __device__ __forceinline__ void vlx_ring_forward(int32_t (&r)[R])
{
    static_assert((R & (R - 1)) == 0 && R > T && (T == 3 || T == 6 || T == 12), "ring of a power of two registers, longer than the taps");
#define VLX_RR(k) r[(P + 1 - (k)) & (R - 1)]
    if constexpr (T == 3)
        asm("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3" : "+v"(VLX_RR(0)), "+v"(VLX_RR(1)), "+v"(VLX_RR(2)) : "v"(VLX_RR(3)));
    else if constexpr (T == 6)
        asm("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\tv_mov_b32 %4, %5\n\tv_mov_b32 %5, %6"
            : "+v"(VLX_RR(0)), "+v"(VLX_RR(1)), "+v"(VLX_RR(2)), "+v"(VLX_RR(3)), "+v"(VLX_RR(4)), "+v"(VLX_RR(5)) : "v"(VLX_RR(6)));
    else
        asm("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\tv_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\t"
            "v_mov_b32 %6, %7\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %8, %9\n\tv_mov_b32 %9, %10\n\tv_mov_b32 %10, %11\n\tv_mov_b32 %11, %12"
            : "+v"(VLX_RR(0)), "+v"(VLX_RR(1)), "+v"(VLX_RR(2)), "+v"(VLX_RR(3)), "+v"(VLX_RR(4)), "+v"(VLX_RR(5)),
              "+v"(VLX_RR(6)), "+v"(VLX_RR(7)), "+v"(VLX_RR(8)), "+v"(VLX_RR(9)), "+v"(VLX_RR(10)), "+v"(VLX_RR(11))
            : "v"(VLX_RR(12)));
#undef VLX_RR
}

#define DPP_QUAD_XOR1 0xB1              // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E              // quad_perm:[2,3,0,1]
#define DPP_QUAD_PREV 0x93              // quad_perm:[3,0,1,2]: lane q reads lane q-1
#define DPP_PAIR_PREV 0xA0              // quad_perm:[0,0,2,2]: the odd lane of a pair reads the even one

// NeuralPredictor (VeloxCore.h:16-41): 12-tap sign-sign LMS, all state in registers, and a history that does not
// move.  LPS = lanes per stream: 1 (twelve taps in one lane), 2 (six per lane) or 4 (three per lane, lane q of the
// group holds taps T*q .. T*q + T - 1; the lone waves of the pipelined shapes, where lanes are free and every
// instruction counts).  A lane's T history values sit in a ring of R = 16 / 8 / 4 registers: at phase P (a
// compile-time constant of the call site — the callers unroll their sample loops in multiples of 16) its tap j is ring
// register (P - j) & (R - 1), so the shift of the history (:38-39) is a renaming: update<P> writes the value that comes
// in (the new sample, or the neighbour lane's oldest) into register (P + 1) & (R - 1) and the next sample runs at
// phase P + 1.  The signs of the history ride along in a ring of their own so that the weight step is one
// v_mad_i32_i24 per tap.
// A stream whose error is zero must NOT shift (:30 returns before the history is touched) while its neighbours in the
// wave do: its lanes move their values (and signs) one register forward instead, so that phase P + 1 finds them where
// it looks — 2 T moves under a branch that is only taken when some stream of the wave has a zero error at this
// sample, where shifting from register to register cost every lane 2 T moves (or selects) at EVERY sample.  The weight
// step of such a stream is += 0 (sign(err) = 0), its leak is masked.
// With LPS > 1 the 64-bit sum is exchanged over DPP as two 32-bit words that add without carries between them:
//   S = sum_q P_q,  P_q = hi_q * 2^11 + lo_q  (0 <= lo_q < 2^11)   =>   S >> 11 = sum hi_q + ((sum lo_q) >> 11)
// and only the low 32 bits of S >> 11 are needed (Predict returns int, VeloxCore.h:26).
// REPAIR = how the streams that stay are handled: 1 = the branch above, laid out in line (the throughput kernels: several
// waves per SIMD hide it); 2 = the same branch out of line behind __any (a lone wave pays ~30 cycles for a taken branch
// pair and nothing for one that falls through); 0 = no ring and no branch at all: every history word is a select between
// "shifted" and "kept" at every sample (2 T v_bfi), which is what a lone QUAD-lane wave does best — with three taps per
// lane the six selects cost less than any branch.
template <int LPS, int REPAIR = 1>
struct VlxLmsT {
    static constexpr int T = VLX_TAPS / LPS;                              // taps per lane
    static constexpr int R = 16 / LPS;                                    // ring registers per lane
    int32_t h[R], sg[R], w[T];
    int32_t first, leak0, leak8;                                          // LPS > 1, all-ones masks: lane 0 of the group / holder of tap 0 / of tap 8
    template <int CTRL> static __device__ __forceinline__ int32_t dpp(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
    static __device__ __forceinline__ int32_t sel(int32_t m, int32_t a, int32_t b) { return (a & m) | (b & ~m); }   // one v_bfi_b32
    __device__ __forceinline__ void reset(int q = 0)                      // q = this lane's place in its group
    {
#pragma unroll
        for (int i = 0; i < R; i++) { h[i] = 0; sg[i] = 0; }
#pragma unroll
        for (int i = 0; i < T; i++) w[i] = 0;
        first = q == 0 ? -1 : 0; leak0 = first; leak8 = q == 8 / T ? -1 : 0;   // tap 0 = lane 0 local 0, tap 8 = lane 8/T local 8%T
    }
    template <int P> __device__ __forceinline__ int32_t predict() const  // Predict(), :22-26; with LPS > 1 the whole group gets the whole sum
    {
        constexpr int Q = REPAIR ? P : 0;                                 // (no ring: the taps stay where phase 0 has them)
        int64_t p = 0;
#pragma unroll
        for (int t = 0; t < T; t++) p += (int64_t)h[(Q - t) & (R - 1)] * w[t];
        if constexpr (LPS == 1) return (int32_t)(p >> 11);
        int32_t hi = (int32_t)(p >> 11), lo = (int32_t)((uint32_t)p & 0x7FFu);
        hi += dpp<DPP_QUAD_XOR1>(hi); lo += dpp<DPP_QUAD_XOR1>(lo);
        if (LPS == 4) { hi += dpp<DPP_QUAD_XOR2>(hi); lo += dpp<DPP_QUAD_XOR2>(lo); }
        return (int32_t)((uint32_t)hi + ((uint32_t)lo >> 11));
    }
    // LPS = 1: the same sum written as two accumulator chains.  hipcc reassociates the integer additions and emits ONE
    // chain of twelve dependent v_mad_i64_i32 for it all the same (seen in the ISA, round 3); with the two sums made
    // opaque before they meet, so that the chains stay apart, k_decode_streams was SLOWER on the same box (cfg2 2.55 ->
    // 2.60 ms, the 60,416-chunk corpus share 2.44 -> 2.47): the extra 64-bit add and register pair cost more than the
    // shorter chain gives.  Kept in this form because it is what the kernels were measured with.
    template <int P> __device__ __forceinline__ int32_t predict2() const
    {
        static_assert(LPS == 1, "the two-chain sum is for one lane per stream");
        constexpr int Q = REPAIR ? P : 0;
        int64_t s0 = 0, s1 = 0;
#pragma unroll
        for (int t = 0; t < T; t += 2) { s0 += (int64_t)h[(Q - t) & (R - 1)] * w[t]; s1 += (int64_t)h[(Q - t - 1) & (R - 1)] * w[t + 1]; }
        return (int32_t)((s0 + s1) >> 11);
    }
    // Update(actual, pred), :27-40, with err = actual - pred already formed (int32, wrapping).  Called for EVERY sample,
    // a zero error included; the state is at phase P before the call and at phase P + 1 after it.
    template <int PH> __device__ __forceinline__ void update(int32_t err, int32_t actual)
    {
        constexpr int P = REPAIR ? PH : 0;
        const int32_t s = vlx_sign(err);                                  // :29
        const int32_t ne = (int32_t)(0u - (uint32_t)err);
        const int32_t aerr = err > ne ? err : ne;                         // std::abs as x86 computes it (INT_MIN stays)
        const int32_t mag = aerr > 1024 ? 16 : 4;                         // :31
        const int32_t sd = __mul24(s, mag);                               // 0 for a zero error
#pragma unroll
        for (int t = 0; t < T; t++) w[t] += __mul24(sg[(P - t) & (R - 1)], sd);     // :32-35 (|w| <= 17 * 4096)
        int32_t lk = __builtin_amdgcn_sbfe(s, 0, 1);                      // -1 when err != 0, else 0
        asm("" : "+v"(lk));                                               // (opaque: hipcc otherwise folds the extraction into four shifts and masks)
        w[0] += __mul24(vlx_sign(w[0]), LPS == 1 ? lk : (lk & leak0));    // :36 leak on taps 0 and 8
        w[8 % T] += __mul24(vlx_sign(w[8 % T]), LPS == 1 ? lk : (lk & leak8));
        int32_t in_h = actual, in_s = vlx_sign(actual);                   // :38-39 for the streams that shift
        if constexpr (LPS > 1) {                                          // lane 0 takes the new value, the others their neighbour's oldest
            constexpr int O = (P - (T - 1)) & (R - 1);
            in_h = sel(first, in_h, LPS == 4 ? dpp<DPP_QUAD_PREV>(h[O]) : dpp<DPP_PAIR_PREV>(h[O]));
            in_s = sel(first, in_s, LPS == 4 ? dpp<DPP_QUAD_PREV>(sg[O]) : dpp<DPP_PAIR_PREV>(sg[O]));
        }
        if constexpr (REPAIR == 0) {                                      // every word chooses: lk is the "shift" mask
#pragma unroll
            for (int t = T - 1; t > 0; t--) { h[(0 - t) & (R - 1)] = sel(lk, h[(1 - t) & (R - 1)], h[(0 - t) & (R - 1)]); sg[(0 - t) & (R - 1)] = sel(lk, sg[(1 - t) & (R - 1)], sg[(0 - t) & (R - 1)]); }
            h[0] = sel(lk, in_h, h[0]); sg[0] = sel(lk, in_s, sg[0]);
            return;
        }
        // every weight is final here: without this hipcc sinks the weight steps behind the branch below, and the old signs
        // they read must then outlive the rotation — in copies, made on the common path
        if constexpr (T == 12) asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]));
        else if constexpr (T == 6) asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]));
        else asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]));
        h[(P + 1) & (R - 1)] = in_h; sg[(P + 1) & (R - 1)] = in_s;
        // :30 — keep tap j where phase P + 1 reads it
        if constexpr (REPAIR == 2) {
            if (__builtin_expect(__any(err == 0), 0)) {
                if (err == 0) { vlx_ring_forward<P & (R - 1), T, R>(h); vlx_ring_forward<P & (R - 1), T, R>(sg); }
            }
        } else {
            // (a plain divergent branch, laid out in line and skipped when no lane stays: out of line, material that stays at
            // nearly every sample — a noise floor of a few LSB — paid two taken branches per sample, 4-5 % of the decode kernel)
            if (err == 0) { vlx_ring_forward<P & (R - 1), T, R>(h); vlx_ring_forward<P & (R - 1), T, R>(sg); }
        }
    }
};
typedef VlxLmsT<1> VlxLms;

// ---- the parser's bit window ---------------------------------------------------------------------
// LSB-first reader (BitStreamReader, VeloxEntropy.h:46-89) over a per-lane ring in LDS ([DS_RING][64] dwords per
// wave), used by k_decode_streams and by the parser wave of k_decode_pipe: the window is kept as two dwords + a
// bit offset instead of a 64-bit accumulator: one v_alignbit gives the next 32 bits, and advancing is branch-free (two v_cndmask, and the
// look-ahead dword is re-read from LDS every sample instead of under a divergent branch).
// DEEP (the parser wave of k_decode_pipe, alone on its SIMD): requests stay in flight for TWO refill intervals
// instead of one (two register sets, used alternately: refill_point<0>, <1>, <0>, ...), and a refill point always
// issues two load instructions, so that the parser can publish its stores with a counted s_waitcnt that leaves
// exactly those loads in flight.
// Both loads of a DEEP refill point are issued whatever the lanes' fill levels (a lane without room drops the result
// and asks again later): no exec-mask branches, and hipcc can count the loads instead of waiting for vmcnt(0) (loads
// under per-lane conditions were measured too: 8 % slower on an idle chip, 2-5 % faster at one workgroup per CU).
template <bool DEEP, int RING = DS_RING>
struct PipeWinT {
    uint32_t *ring;                            // &wave_ring[lane]
    const uint4 *base; uint32_t last_g;        // 16-B aligned view of the payload; last readable group
    uint32_t w0, w1, qn;                       // ring dwords d, d+1, d+2
    uint32_t bo;                               // bit offset of the read position inside w0
    uint32_t d, d0;                            // dword counter of w0; its value at init()
    uint32_t wr;                               // dwords written into the ring so far
    uint32_t gnext; uint4 pa, pb; uint32_t pend;
    uint4 pc, pd; uint32_t pend2;              // DEEP: the second set
    uint32_t bo0;
    bool under;
    __device__ __forceinline__ uint4 load(uint32_t g) const { return vlx_gload16(base + (g < last_g ? g : last_g)); }
    __device__ __forceinline__ uint32_t peek(uint32_t idx) const { return ring[(idx & (RING - 1)) * 64]; }
    __device__ __forceinline__ void ring_put(const uint4 v)               // wr is a multiple of 4: a group never wraps inside
    {
        uint32_t *p = ring + (wr & (RING - 4)) * 64;
        p[0] = v.x; p[64] = v.y; p[128] = v.z; p[192] = v.w;
        wr += 4;
    }
    __device__ __forceinline__ void init(uint32_t *r, const uint4 *b, uint32_t lg, uint64_t bit)
    {
        ring = r; base = b; last_g = lg; under = false; wr = 0; pend = 0; pend2 = 0;
        // what the ring starts with has to last until the first requests arrive: one refill interval, or two when DEEP
        // — and a stream's first samples are its longest codes (nothing adapted yet: an escape code of 105 bits, then
        // a few of 30 and more), behind a header of 141 bits.  Six groups ran dry on 8-channel 32-bit material with
        // two intervals to bridge (every such chunk went to the careful kernel): ten there.
        constexpr int NG = (DEEP && RING >= 64) ? 10 : 6;
        const uint32_t g = (uint32_t)(bit >> 7);
        uint4 t[NG];
#pragma unroll
        for (int j = 0; j < NG; j++) t[j] = load(g + j);
#pragma unroll
        for (int j = 0; j < NG; j++) ring_put(t[j]);
        gnext = g + NG;
        d = (uint32_t)(bit >> 5) & 3u; d0 = d;
        bo = (uint32_t)bit & 31u; bo0 = bo;
        w0 = peek(d); w1 = peek(d + 1); qn = peek(d + 2);
    }
    __device__ __forceinline__ uint32_t used() const { return 32u * (d - d0) + bo - bo0; }   // bits consumed since init()
    // Wave-uniform refill point, once per DS_TILE samples: what was requested a tile ago goes into the
    // ring, the next 16-B groups are requested.  Dwords d.. are live (d + 2 is re-read every sample).
    template <int PH = 0>
    __device__ __forceinline__ void refill_point()
    {
        if (d + 2 > wr) under = true;                                     // w1 held a dword that was never written
        if constexpr (!DEEP) {
            if (pend >= 1) ring_put(pa);
            if (pend >= 2) ring_put(pb);
            const uint32_t fill = wr - d;                                 // (garbage once under)
            pend = fill <= RING - 8 ? 2u : (fill <= RING - 4 ? 1u : 0u);
            if (under) pend = 0;
            if (pend >= 1) pa = load(gnext);
            if (pend >= 2) pb = load(gnext + 1);
            gnext += pend;
        } else {
            uint4 &qa = PH ? pc : pa, &qb = PH ? pd : pb;                 // requested two refill points ago
            uint32_t &mine = PH ? pend2 : pend;
            const uint32_t others = PH ? pend : pend2;                    // groups requested one refill point ago, still on their way
            if (mine >= 1) ring_put(qa);
            if (mine >= 2) ring_put(qb);
            const uint32_t fill = wr + 4u * others - d;
            mine = fill <= RING - 8 ? 2u : (fill <= RING - 4 ? 1u : 0u);
            if (under) mine = 0;
            qa = load(gnext); qb = load(gnext + 1);
            gnext += mine;
        }
        qn = peek(d + 2);
    }
    __device__ __forceinline__ uint32_t next32() const { return __builtin_amdgcn_alignbit(w1, w0, bo); }
    __device__ __forceinline__ void advance(uint32_t n)                   // n in [0,32]
    {
        const uint32_t b2 = bo + n;
        const bool adv = b2 > 31u;
        w0 = adv ? w1 : w0; w1 = adv ? qn : w1; d += adv ? 1u : 0u;
        bo = b2 & 31u;
        qn = peek(d + 2);
    }
    __device__ __forceinline__ uint32_t take(uint32_t n)                  // n in [0,32]
    {
        const uint32_t lo = next32();
        const uint32_t v = n >= 32u ? lo : (lo & ((1u << n) - 1u));
        advance(n);
        return v;
    }
};
typedef PipeWinT<false> PipeWin;

// DecodeSample (VeloxEntropy.h:121-138) for values that fit 32 bits, any code length; sets bad when the
// code is outside the fast envelope (m >= 2^32, more than 64 ones).  The per-lane form: the parser's hot
// loop inlines the short-code case and comes here only to redo a half tile (see there).
template <typename W>
__device__ __forceinline__ uint32_t dp_sample(W &w, uint32_t k, bool &bad)
{
    const uint32_t lo = w.next32();
    const uint32_t q = (uint32_t)__builtin_ctz(~lo | 0x80000000u);        // ones before the first 0
    const uint32_t q1 = q + 1u;
    uint32_t len = q1 + k;
    uint32_t m = (q << k) | __builtin_amdgcn_ubfe(lo, q1, k);             // q1 == 32 only with len >= 32: side path
    const bool slow = len > 31u;
    {
        if (slow) {
            len = 0;                                                      // this lane consumes its code piecewise
            uint32_t qq = 0; bool done = false;
            for (int it = 0; it < 3 && !done; it++) {                     // unary part, 32 bits at a time
                const uint32_t x = w.next32();
                if (x == 0xFFFFFFFFu) { qq += 32u; w.advance(32); }
                else { const uint32_t t = (uint32_t)__builtin_ctz(~x); qq += t; w.advance(t + 1u); done = true; }
            }
            if (!done || qq > 64u) { bad = true; m = 0; }
            else if (qq < 64u) {
                const uint32_t r = w.take(k);
                const uint64_t mm = ((uint64_t)qq << k) | r;
                if (mm >> 32) bad = true;
                m = (uint32_t)mm;
            } else {                                                      // escape: 40 raw bits of m
                m = w.take(32);
                if (w.take(8)) bad = true;
            }
        }
    }
    w.advance(len);
    return m;
}

