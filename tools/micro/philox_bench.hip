// Micro-benchmark (diagnostic, not part of the product): cost of one Philox4x32-10 block per wavefront on gfx950 at the
// step kernel's occupancy (2 waves/SIMD), and of its building blocks.   hipcc -O3 --offload-arch=gfx950 -I../../everglades-ai-wargame_amd/csrc philox_bench.hip -o philox_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "evg_rng.h"
using namespace evg;

template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t acc = threadIdx.x + blockIdx.x * 64, a = seed ^ acc, b = acc * 3u + 1u;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {                       // Philox block, result feeds the next counter (as independent as the kernel's use)
            const uint4 x = rng_block(seed, 7u, acc, 3u, 0, (uint32_t)i, 80, 5, 1, 7);
            acc ^= x.x ^ x.y ^ x.z ^ x.w;
        } else if (MODE == 1) {                // 20 x v_mad_u64_u32, dependent pairs
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                const uint64_t p0 = (uint64_t)0xD2511F53u * a, p1 = (uint64_t)0xCD9E8D57u * b;
                a = (uint32_t)(p1 >> 32) ^ (uint32_t)p0; b = (uint32_t)(p0 >> 32) ^ (uint32_t)p1;
            }
            acc ^= a ^ b;
        } else {                               // 60 plain 32-bit VALU ops
#pragma unroll
            for (int r = 0; r < 30; ++r) { a = (a ^ b) + 0x9E3779B9u; b = (b << 3) ^ a; }
            acc ^= a ^ b;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[1 << 20] = (uint32_t)(t1 - t0); }
}

int main() {
    uint32_t* d; hipMalloc(&d, ((1 << 20) + 16) * 4);
    const int iters = 2000, grid = 2048;      // 2048 waves = every wave slot at 2 waves/SIMD
    const char* names[3] = {"philox4x32-10 block", "20 x mad_u64_u32 (+20 xor)", "60 x 32-bit VALU"};
    for (int m = 0; m < 3; ++m) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, d, iters, 12345u);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, d, iters, 12345u);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, d, iters, 12345u);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint32_t cyc; hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
        printf("%-28s %8.3f ms  -> %.1f ns per iteration per wave pair-slot; wave 0: %.1f memtime ticks / iteration\n", names[m], ms, ms * 1e6 / iters, (double)cyc / iters);
    }
    return 0;
}
