// Micro-benchmark (diagnostic, not part of the product): issue cost of single instructions on gfx950 at the step kernel's
// occupancy.  Each kernel runs UNROLL independent copies of one instruction per loop iteration (8 independent register chains,
// so dependencies do not limit issue) on a grid of 1 or 2 waves per SIMD; reported: shader cycles per instruction per SIMD
// (s_memtime of wave 0 / instructions issued by the waves of its SIMD).
//   hipcc -O3 --offload-arch=gfx950 inst_rate.hip -o inst_rate && ./inst_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t r[8];
    uint64_t q[8];
    double d[8];
    __shared__ uint32_t lds[64 * 16];
    for (int i = 0; i < 8; ++i) { r[i] = seed * (i + 3) + threadIdx.x; q[i] = ((uint64_t)r[i] << 20) | i; d[i] = 1.0 + r[i] * 1e-9; }
    lds[threadIdx.x] = seed;
    const uint32_t s1 = seed | 1u;
    const double dk = 1.0000001, dm = 0.9999999;
    if (MODE == 13) asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define V32(i, OP) asm volatile(OP " %0, %0, %1" : "+v"(r[i]) : "v"(s1));
#define X0(i) V32(i, "v_add_u32")
#define X1(i) V32(i, "v_mul_lo_u32")
#define X2(i) V32(i, "v_mul_hi_u32")
#define X3(i) V32(i, "v_mul_u32_u24")
#define X4(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(r[i]), "v"(s1) : "vcc");
#define X5(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
#define X6(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dk), "v"(dm));
#define X7(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
#define X8(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dk));
#define X9(i) asm volatile("v_cmp_ge_f64 vcc, %0, %1" :: "v"(d[i]), "v"(dm) : "vcc");
#define X10(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(r[i]));
#define X11(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s1), "v"(seed));
#define X12(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s1));
#define X13(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(s1));
#define X14(i) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(s1));
#define X15(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s1));
#define X16(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r[i]) : "v"(r[(i + 1) & 7]));
#define X17(i) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r[i]) : "v"(r[(i + 1) & 7]));
#define X18(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s1), "v"(seed));
#define X19(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r[i]) : "v"(s1));
#define X20(i) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s1), "v"(seed));
#define X21(i) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(r[i]));
#define X22(i) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
#define X23(i) asm volatile("ds_read_b32 %0, %1" : "=v"(r[i]) : "v"((uint32_t)(threadIdx.x * 4)));
#define X24(i) asm volatile("ds_write_b32 %0, %1" :: "v"((uint32_t)(threadIdx.x * 4)), "v"(r[i]));
#define X25(i) asm volatile("ds_add_u32 %0, %1" :: "v"((uint32_t)(threadIdx.x * 4)), "v"(r[i]));
#define X26(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
#define X27(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s1));
#define X28(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
#define X29(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s1), "v"(seed));
            if (MODE == 0) { REP8(X0) } else if (MODE == 1) { REP8(X1) } else if (MODE == 2) { REP8(X2) } else if (MODE == 3) { REP8(X3) }
            else if (MODE == 4) { REP8(X4) } else if (MODE == 5) { REP8(X5) } else if (MODE == 6) { REP8(X6) } else if (MODE == 7) { REP8(X7) }
            else if (MODE == 8) { REP8(X8) } else if (MODE == 9) { REP8(X9) } else if (MODE == 10) { REP8(X10) } else if (MODE == 11) { REP8(X11) }
            else if (MODE == 12) { REP8(X12) } else if (MODE == 13) { REP8(X13) } else if (MODE == 14) { REP8(X14) } else if (MODE == 15) { REP8(X15) }
            else if (MODE == 16) { REP8(X16) } else if (MODE == 17) { REP8(X17) } else if (MODE == 18) { REP8(X18) } else if (MODE == 19) { REP8(X19) }
            else if (MODE == 20) { REP8(X20) } else if (MODE == 21) { REP8(X21) } else if (MODE == 22) { REP8(X22) } else if (MODE == 23) { REP8(X23) }
            else if (MODE == 24) { REP8(X24) } else if (MODE == 25) { REP8(X25) } else if (MODE == 26) { REP8(X26) } else if (MODE == 27) { REP8(X27) }
            else if (MODE == 28) { REP8(X28) } else { REP8(X29) }
        }
        if (MODE == 23) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = clock64();
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= r[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc + lds[(threadIdx.x * 7) & 63];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}

typedef void (*kern_t)(uint32_t*, int, uint32_t);
template <int M> struct Tab { static void fill(kern_t* t) { t[M] = k<M>; Tab<M - 1>::fill(t); } };
template <> struct Tab<-1> { static void fill(kern_t*) {} };

int main() {
    uint32_t* dmem; hipMalloc(&dmem, ((1 << 20) + 16) * 4);
    const char* names[30] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u64_u32", "v_lshlrev_b64", "v_fma_f64", "v_add_f64", "v_mul_f64",
                             "v_cmp_ge_f64", "v_cvt_f64_u32", "v_perm_b32", "v_bcnt_u32_b32", "v_cndmask_b32", "v_lshrrev_b64 (var)", "v_xor_b32", "v_mov_b32 dpp quad_perm",
                             "v_cvt_f32_i32 sdwa", "v_or3_b32", "v_lshl_add_u32", "v_alignbyte_b32", "v_bfe_u32", "s_add_u32", "ds_read_b32 (+wait per 32)", "ds_write_b32",
                             "ds_add_u32", "v_max_f64", "v_min_u32", "v_lshl_add_u64", "v_mad_u32_u24"};
    kern_t tab[30];
    Tab<29>::fill(tab);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int grid = 1024 * wps;             // 1024 SIMDs
        printf("---- %d wave(s) per SIMD (%d workgroups of 64)\n", wps, grid);
        for (int m = 0; m < 30; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(tab[m], dim3(grid), dim3(64), 0, 0, dmem, iters, 12345u);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            uint32_t cyc; hipMemcpy(&cyc, dmem + (1 << 20), 4, hipMemcpyDeviceToHost);
            const double n = (double)iters * 32.0;
            printf("%-28s wave 0: %6.2f cycles per own instruction = %6.2f cycles per instruction issued on its SIMD   (%.3f ms = %.2f cycles of 2.4 GHz per instruction per SIMD)\n", names[m], cyc / n, cyc / n / wps, ms, ms * 1e-3 * 2.4e9 / (n * wps));
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    return 0;
}
