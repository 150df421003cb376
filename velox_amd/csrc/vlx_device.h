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
