// Micro-benchmark (diagnostic, not part of the product): does a wave64 VALU / LDS instruction get cheaper when only the first L
// lanes of the wavefront are active?  (The step kernel's second combat round has 18 of 64 items on average: if the hardware skipped
// the inactive 16-lane passes, that round would already be cheap.)  Two waves per SIMD, 8 independent chains per lane, whole-kernel time.
//   hipcc -O3 --offload-arch=gfx950 exec_mask_rate.hip -o exec_mask_rate && ./exec_mask_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, uint32_t seed, int active) {
    uint32_t r[8]; uint64_t q[8]; double d[8];
    for (int i = 0; i < 8; ++i) { r[i] = seed * (i + 3) + threadIdx.x; q[i] = ((uint64_t)r[i] << 20) | i; d[i] = 1.0 + r[i] * 1e-9; }
    const uint32_t s1 = seed | 1u; const double dk = 1.0000001, dm = 0.9999999;
    if ((int)threadIdx.x < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#define X0(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s1));
#define X1(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(r[i]), "v"(s1) : "vcc");
#define X2(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dk), "v"(dm));
#define X3(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s1), "v"(seed));
#define X4(i) asm volatile("ds_add_u32 %0, %1" :: "v"((uint32_t)(threadIdx.x * 4)), "v"(r[i]));
                if (MODE == 0) { REP8(X0) } else if (MODE == 1) { REP8(X1) } else if (MODE == 2) { REP8(X2) } else if (MODE == 3) { REP8(X3) } else { REP8(X4) }
            }
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= r[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
typedef void (*kern_t)(uint32_t*, int, uint32_t, int);
int main() {
    uint32_t* dmem; hipMalloc(&dmem, (1 << 20) * 4);
    const char* names[5] = {"v_add_u32", "v_mad_u64_u32", "v_fma_f64", "v_perm_b32", "ds_add_u32"};
    kern_t tab[5] = {k<0>, k<1>, k<2>, k<3>, k<4>};
    const int iters = 20000, grid = 2048;            // 2 waves per SIMD
    for (int m = 0; m < 5; ++m)
        for (int active = 64; active >= 8; active /= 2) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(tab[m], dim3(grid), dim3(64), 0, 0, dmem, iters, 12345u, active);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-16s lanes active %2d: %.3f ms = %.2f cycles of 2.4 GHz per instruction per SIMD\n", names[m], active, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 32.0 * 2));
        }
    return 0;
}
