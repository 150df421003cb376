// What ONE wave that has its SIMD to itself pays per instruction on gfx950: a chain of DEPENDENT
// instructions against four independent chains, for the instruction forms of the pipelined small-batch
// kernels (vlx_pipe.hip), an LDS read whose address depends on the previous read, and a v_cmp + scalar
// branch pair.  One workgroup of 64 threads; cycles from s_memtime.
// Build: make -C tools ubench_lone      Run: tools/ubench_lone
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 2048

#define DEP(NAME, STR, ...)                                                                  \
    __global__ __launch_bounds__(64) void NAME##_dep(unsigned long long *out, unsigned seed)   \
    {                                                                                        \
        int a0 = threadIdx.x + seed, b = seed | 1, c = (seed + 9) & 31;                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                          \
        _Pragma("unroll 16") for (int i = 0; i < N; i++) { asm volatile(STR : "+v"(a0) : __VA_ARGS__); } \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                          \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                              \
        if (a0 == 0x12345) out[1] = a0;                                                      \
    }                                                                                        \
    __global__ __launch_bounds__(64) void NAME##_ind(unsigned long long *out, unsigned seed)   \
    {                                                                                        \
        int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed | 1, c = (seed + 9) & 31; \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                          \
        _Pragma("unroll 4") for (int i = 0; i < N / 4; i++) {                                \
            asm volatile(STR : "+v"(a0) : __VA_ARGS__); asm volatile(STR : "+v"(a1) : __VA_ARGS__); \
            asm volatile(STR : "+v"(a2) : __VA_ARGS__); asm volatile(STR : "+v"(a3) : __VA_ARGS__); } \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                          \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                              \
        if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345) out[1] = a0;                                     \
    }

DEP(add_u32, "v_add_u32 %0, %0, %1", "v"(b))
DEP(alignbit, "v_alignbit_b32 %0, %0, %1, %2", "v"(b), "v"(c))
DEP(ffbl, "v_ffbl_b32 %0, %0", "v"(b))
DEP(bfe_u32, "v_bfe_u32 %0, %0, %1, %2", "v"(c), "v"(c))
DEP(lshl_or, "v_lshl_or_b32 %0, %0, %1, %2", "v"(c), "v"(b))
DEP(bfi, "v_bfi_b32 %0, %1, %0, %2", "v"(b), "v"(c))
DEP(cndmask, "v_cndmask_b32 %0, %0, %1, vcc", "v"(b))
DEP(cmp_cndmask, "v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc", "v"(c), "v"(b) : "vcc")
DEP(mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2", "v"(b), "v"(c))
DEP(mul_lo, "v_mul_lo_u32 %0, %0, %1", "v"(b))
DEP(cmp_branch, "v_cmp_eq_u32 vcc, 0x7fffffff, %0\n\ts_cbranch_vccnz 1f\n\tv_add_u32 %0, %0, %1\n1:", "v"(b) : "vcc")
DEP(add_nop, "v_add_u32 %0, %0, %1\n\ts_nop 0", "v"(b))

__global__ __launch_bounds__(64) void mad64_dep(unsigned long long *out, unsigned seed)
{
    long long a0 = threadIdx.x + seed; int b = seed | 1, c = seed + 9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; i++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c) : "vcc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a0 == 0x12345) out[1] = a0;
}
__global__ __launch_bounds__(64) void mad64_ind(unsigned long long *out, unsigned seed)
{
    long long a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7; int b = seed | 1, c = seed + 9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int i = 0; i < N / 4; i++) {
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c) : "vcc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345) out[1] = a0;
}
// LDS read whose address is the previous read's result (the table holds a permutation of its own offsets)
__global__ __launch_bounds__(64) void lds_dep(unsigned long long *out, unsigned seed)
{
    __shared__ unsigned tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) tab[i] = ((i + 64) & 2047) * 4;
    __syncthreads();
    unsigned a = threadIdx.x * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; i++) a = *(volatile unsigned *)((char *)tab + a);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a == 0x12345) out[1] = a;
}
// an LDS read issued every 8 independent VALU instructions, consumed 8 instructions later
__global__ __launch_bounds__(64) void lds_beside(unsigned long long *out, unsigned seed)
{
    __shared__ unsigned tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) tab[i] = i;
    __syncthreads();
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed | 1;
    unsigned q = 0, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int i = 0; i < N / 8; i++) {
        acc += q;
        q = *(volatile unsigned *)&tab[(threadIdx.x + i) & 2047];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ acc) == 0x12345) out[1] = a0;
}

typedef void (*kfn)(unsigned long long *, unsigned);
struct Entry { const char *name; kfn fn; int per; };
#define E2(n) { #n " dep", n##_dep, 1 }, { #n " ind", n##_ind, 1 }

int main()
{
    Entry es[] = { E2(add_u32), E2(alignbit), E2(ffbl), E2(bfe_u32), E2(lshl_or), E2(bfi), E2(cndmask), E2(cmp_cndmask), E2(mad_i32_i24), E2(mul_lo),
                   E2(cmp_branch), E2(add_nop), { "mad_i64_i32 dep", mad64_dep, 1 }, { "mad_i64_i32 ind", mad64_ind, 1 }, { "lds_read dep", lds_dep, 1 },
                   { "8 v_add + 1 ds_read (per group of 9)", lds_beside, 8 } };
    unsigned long long *out; hipMalloc(&out, 64);
    for (auto &e : es) {
        unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 3; rep++) { e.fn<<<1, 64>>>(out, 1); hipDeviceSynchronize(); }
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("%-40s %8.2f cycles per %s\n", e.name, (double)h[0] / (N / e.per), e.per == 1 ? "instruction (pair for cmp_*)" : "group");
    }
    return 0;
}
