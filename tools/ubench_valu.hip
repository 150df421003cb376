// VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction per SIMD at full
// occupancy (8 waves/SIMD), for the integer / fp64 forms the serial codec kernels choose between.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define ITERS 4096
#define REP 32

#define BENCH_KERNEL(NAME, DECL, ...)                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed)       \
    {                                                                               \
        DECL;                                                                       \
        for (int it = 0; it < ITERS; it++) {                                        \
            _Pragma("unroll") for (int r = 0; r < REP / 4; r++) { __VA_ARGS__; }           \
        }                                                                           \
        unsigned acc = (unsigned)a0 ^ (unsigned)a1 ^ (unsigned)a2 ^ (unsigned)a3;   \
        if (acc == 0x12345u) out[threadIdx.x] = acc;                                \
    }

#define I32 int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed | 1, c = seed + 9
#define I64 long long a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7; int b = seed | 1, c = seed + 9
#define F64 double a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = 1.0000001, c = 0.5

#define ASM4(STR, ...) asm volatile(STR : "+v"(a0) : __VA_ARGS__); asm volatile(STR : "+v"(a1) : __VA_ARGS__); asm volatile(STR : "+v"(a2) : __VA_ARGS__); asm volatile(STR : "+v"(a3) : __VA_ARGS__)

BENCH_KERNEL(k_mad_i64_i32, I64, ASM4("v_mad_i64_i32 %0, vcc, %1, %2, %0", "v"(b), "v"(c) : "vcc"))
BENCH_KERNEL(k_mad_u64_u32, I64, ASM4("v_mad_u64_u32 %0, vcc, %1, %2, %0", "v"(b), "v"(c) : "vcc"))
BENCH_KERNEL(k_mad_i32_i24, I32, ASM4("v_mad_i32_i24 %0, %1, %2, %0", "v"(b), "v"(c)))
BENCH_KERNEL(k_mul_lo_u32, I32, ASM4("v_mul_lo_u32 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_mul_hi_i32, I32, ASM4("v_mul_hi_i32 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_mul_hi_i32_i24, I32, ASM4("v_mul_hi_i32_i24 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_add_u32, I32, ASM4("v_add_u32 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_add3_u32, I32, ASM4("v_add3_u32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_med3_i32, I32, ASM4("v_med3_i32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_mov_b32, I32, ASM4("v_mov_b32 %0, %1", "v"(b)))
BENCH_KERNEL(k_mov_b64, I64, ASM4("v_mov_b64 %0, %1", "v"(a0)))
BENCH_KERNEL(k_pk_mov_b32, I64, ASM4("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]", "v"(a0)))
BENCH_KERNEL(k_fma_f64, F64, ASM4("v_fma_f64 %0, %1, %2, %0", "v"(b), "v"(c)))
BENCH_KERNEL(k_add_f64, F64, ASM4("v_add_f64 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_lshlrev_b64, I64, ASM4("v_lshlrev_b64 %0, %1, %0", "v"(b)))
BENCH_KERNEL(k_ashrrev_i64, I64, ASM4("v_ashrrev_i64 %0, %1, %0", "v"(b)))
BENCH_KERNEL(k_dot2_i32_i16, I32, ASM4("v_dot2_i32_i16 %0, %1, %2, %0", "v"(b), "v"(c)))
BENCH_KERNEL(k_dot4_i32_i8, I32, ASM4("v_dot4_i32_i8 %0, %1, %2, %0", "v"(b), "v"(c)))
BENCH_KERNEL(k_pk_mad_i16, I32, ASM4("v_pk_mad_i16 %0, %1, %2, %0", "v"(b), "v"(c)))
BENCH_KERNEL(k_pk_add_i16, I32, ASM4("v_pk_add_i16 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_perm_b32, I32, ASM4("v_perm_b32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_alignbit_b32, I32, ASM4("v_alignbit_b32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_bfe_i32, I32, ASM4("v_bfe_i32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_ffbh_u32, I32, ASM4("v_ffbh_u32 %0, %0", "v"(b)))
BENCH_KERNEL(k_ffbl_b32, I32, ASM4("v_ffbl_b32 %0, %0", "v"(b)))
BENCH_KERNEL(k_cndmask, I32, ASM4("v_cndmask_b32 %0, %0, %1, vcc", "v"(b)))
BENCH_KERNEL(k_cvt_f64_i32, F64, ASM4("v_cvt_f64_i32 %0, %1", "v"((int)seed)))
BENCH_KERNEL(k_cvt_i32_f64, I32, ASM4("v_cvt_i32_f64 %0, %1", "v"((double)seed)))
BENCH_KERNEL(k_cvt_f32_i32, I32, ASM4("v_cvt_f32_i32 %0, %0", "v"(b)))
BENCH_KERNEL(k_fma_f32, I32, ASM4("v_fma_f32 %0, %1, %2, %0", "v"(b), "v"(c)))
BENCH_KERNEL(k_pk_fma_f32, I64, ASM4("v_pk_fma_f32 %0, %1, %1, %0", "v"(a0)))
BENCH_KERNEL(k_lshl_add_u32, I32, ASM4("v_lshl_add_u32 %0, %0, 3, %1", "v"(b)))
BENCH_KERNEL(k_sub_co, I32, ASM4("v_sub_co_u32 %0, vcc, %0, %1", "v"(b) : "vcc"))
BENCH_KERNEL(k_max3_i32, I32, ASM4("v_max3_i32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_bfi_b32, I32, ASM4("v_bfi_b32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_xad_u32, I32, ASM4("v_xad_u32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_mbcnt, I32, ASM4("v_mbcnt_lo_u32_b32 %0, %1, %0", "v"(b)))

// forms with fewer VGPR sources (round 3: three-VGPR-source VOP3 forms issue slower at low occupancy — is it the sources?)
BENCH_KERNEL(k_med3_consts, I32, ASM4("v_med3_i32 %0, %0, -1, 1", "v"(b)))
BENCH_KERNEL(k_bfe_consts, I32, ASM4("v_bfe_i32 %0, %0, 0, 17", "v"(b)))
BENCH_KERNEL(k_alignbit_const, I32, ASM4("v_alignbit_b32 %0, %0, %1, 11", "v"(b)))
BENCH_KERNEL(k_mad24_sgpr, I32, ASM4("v_mad_i32_i24 %0, %1, %2, %0", "v"(b), "s"(c)))
BENCH_KERNEL(k_mad24_const, I32, ASM4("v_mad_i32_i24 %0, %1, 4, %0", "v"(b)))
BENCH_KERNEL(k_mul_i32_i24, I32, ASM4("v_mul_i32_i24 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_lshl_or_b32, I32, ASM4("v_lshl_or_b32 %0, %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_lshlrev_b32, I32, ASM4("v_lshlrev_b32 %0, %1, %0", "v"(b)))
BENCH_KERNEL(k_xor_b32, I32, ASM4("v_xor_b32 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_cndmask_sgpr, I32, ASM4("v_cndmask_b32_e64 %0, %0, %1, %2", "v"(b), "s"(0x5555555555555555ull)))
BENCH_KERNEL(k_cmp_cndmask, I32, ASM4("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc", "v"(b), "v"(c) : "vcc"))
BENCH_KERNEL(k_mad64_sgpr, I64, ASM4("v_mad_i64_i32 %0, vcc, %1, %2, %0", "v"(b), "s"(c) : "vcc"))
BENCH_KERNEL(k_bitop3, I32, ASM4("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x48", "v"(b), "v"(c)))
BENCH_KERNEL(k_sub_u32, I32, ASM4("v_sub_u32 %0, %0, %1", "v"(b)))
BENCH_KERNEL(k_max_i32, I32, ASM4("v_max_i32 %0, %0, %1", "v"(b)))

// accumulate-in-place VOP2 forms (two encoded sources, the destination is the third)
BENCH_KERNEL(k_dot2c_i32_i16, I32, ASM4("v_dot2c_i32_i16 %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_dot4c_i32_i8, I32, ASM4("v_dot4c_i32_i8 %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_fmac_f32, I32, ASM4("v_fmac_f32 %0, %1, %2", "v"(b), "v"(c)))
BENCH_KERNEL(k_fmac_f64, F64, ASM4("v_fmac_f64 %0, %1, %2", "v"(b), "v"(c)))

typedef void (*kfn)(unsigned *, unsigned);
struct Entry { const char *name; kfn fn; };
#define E(n) { #n, n }

int main(int argc, char **argv)
{
    Entry es[] = { E(k_mad_i64_i32), E(k_mad_u64_u32), E(k_mad_i32_i24), E(k_mul_lo_u32), E(k_mul_hi_i32), E(k_mul_hi_i32_i24),
                   E(k_add_u32), E(k_add3_u32), E(k_med3_i32), E(k_mov_b32), E(k_mov_b64), E(k_pk_mov_b32), E(k_fma_f64), E(k_add_f64),
                   E(k_lshlrev_b64), E(k_ashrrev_i64), E(k_dot2_i32_i16), E(k_dot4_i32_i8), E(k_pk_mad_i16), E(k_pk_add_i16),
                   E(k_perm_b32), E(k_alignbit_b32), E(k_bfe_i32), E(k_ffbh_u32), E(k_ffbl_b32), E(k_cndmask), E(k_cvt_f64_i32),
                   E(k_cvt_i32_f64), E(k_cvt_f32_i32), E(k_fma_f32), E(k_pk_fma_f32), E(k_lshl_add_u32), E(k_sub_co), E(k_max3_i32),
                   E(k_bfi_b32), E(k_xad_u32), E(k_mbcnt), E(k_med3_consts), E(k_bfe_consts), E(k_alignbit_const), E(k_mad24_sgpr),
                   E(k_mad24_const), E(k_mul_i32_i24), E(k_lshl_or_b32), E(k_lshlrev_b32), E(k_xor_b32), E(k_cndmask_sgpr), E(k_cmp_cndmask),
                   E(k_mad64_sgpr), E(k_bitop3), E(k_sub_u32), E(k_max_i32), E(k_dot2c_i32_i16), E(k_dot4c_i32_i8), E(k_fmac_f32), E(k_fmac_f64) };
    unsigned *out; hipMalloc(&out, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wps = 8; wps >= 1; wps = (wps == 8 ? 3 : wps - 1)) {
        const int blocks = 256 * wps;   // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        printf("---- %d wave(s) per SIMD\n", wps);
        for (auto &e : es) {
            e.fn<<<blocks, 256>>>(out, 1);
            hipDeviceSynchronize();
            hipEventRecord(a);
            e.fn<<<blocks, 256>>>(out, 1);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double inst_per_simd = (double)ITERS * REP * wps;
            printf("%-20s %8.3f ms  %6.2f cyc/inst/SIMD @2.4GHz\n", e.name, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
        }
    }
    return 0;
}
