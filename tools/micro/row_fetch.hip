// row_fetch.hip -- what does a 64-byte row read cost the fabric / HBM on gfx950, and how does FETCH_SIZE tally it?
// The step kernel's float64 health rows are aligned 64-byte segments read with 16 bytes per lane (4 lanes per row); the PMC summaries double FETCH_SIZE
// (the guide's correction, calibrated on a wide streaming copy: a 128-byte request is tallied as 64).  If a 64-byte row read is ONE 64-byte request, that
// correction over-counts the rows by 2x; if the L2 fills whole 128-byte lines, the correction is right and half of every fetched line is waste.
// Four kernels over a 4 GiB buffer (16x the Infinity Cache), each lane reading 16 bytes per iteration:
//   full      streams the whole buffer                                  -> bytes requested = B
//   half      4 lanes per 64-byte row, rows = the FIRST half of every 128-byte line  -> requested B / 2, lines touched B / 128
//   halfbuf   streams the first half of the buffer                      -> requested B / 2, lines touched B / 256
//   rows1600  64-byte rows 1 600 bytes apart (the env-major health layout: one row per env) -> requested B / 25
// Times (HIP events, best of 3) say what the HBM moved: half ~ full means whole lines are fetched, half ~ halfbuf means 64-byte requests.  Run it under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- ./row_fetch
// to see the tally of each kernel (tools/profile_rowfetch.sh).
// build: hipcc -O3 --offload-arch=gfx950 -o row_fetch row_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// mode 0 full, 1 half (first 64 B of every 128-B line), 2 halfbuf, 3 rows at stride 1600 B
template <int MODE>
__global__ void __launch_bounds__(256) read_kernel(const uint4* __restrict__ buf, size_t bytes, uint32_t* __restrict__ sink) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    if (MODE == 0 || MODE == 2) {
        const size_t n = (MODE == 0 ? bytes : bytes / 2) / 16;
        for (size_t i = tid; i < n; i += nthreads) { const uint4 v = buf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    } else {
        const size_t stride = MODE == 1 ? 128 : 1600;          // bytes between rows
        const size_t rows = bytes / stride;
        for (size_t i = tid; i < rows * 4; i += nthreads) {     // 4 lanes per 64-byte row
            const size_t r = i >> 2, q = i & 3;
            const uint4 v = buf[(r * stride) / 16 + q];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[tid & 1023] = acc;            // never true for the zero-filled buffer's pattern; keeps the loads alive
}

template <int MODE>
static float run(const uint4* buf, size_t bytes, uint32_t* sink, hipEvent_t a, hipEvent_t b) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, nullptr);
        hipLaunchKernelGGL(read_kernel<MODE>, dim3(256 * 8), dim3(256), 0, nullptr, buf, bytes, sink);
        hipEventRecord(b, nullptr);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    uint4* buf = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const float t_full = run<0>(buf, bytes, sink, a, b), t_half = run<1>(buf, bytes, sink, a, b), t_halfbuf = run<2>(buf, bytes, sink, a, b), t_rows = run<3>(buf, bytes, sink, a, b);
    const double gb = bytes / 1e9;
    printf("buffer %.2f GB (16 x the 256 MiB Infinity Cache); 16 bytes per lane per load\n", gb);
    printf("full      (stream all)                         %8.3f ms  requested %6.2f GB  %6.2f TB/s of requested bytes\n", t_full, gb, gb / t_full);
    printf("half      (first 64 B of every 128-B line)     %8.3f ms  requested %6.2f GB  %6.2f TB/s of requested bytes, %6.2f TB/s if whole lines move\n", t_half, gb / 2, gb / 2 / t_half, gb / t_half);
    printf("halfbuf   (stream the first half)              %8.3f ms  requested %6.2f GB  %6.2f TB/s of requested bytes\n", t_halfbuf, gb / 2, gb / 2 / t_halfbuf);
    printf("rows1600  (64-B rows 1 600 B apart)            %8.3f ms  requested %6.2f GB  %6.2f TB/s of requested bytes, %6.2f TB/s if whole lines move\n", t_rows, gb / 25, gb / 25 / t_rows, gb / 25 * 2 / t_rows);
    printf("half / full = %.2f, half / halfbuf = %.2f  (1.0 / 2.0: the fabric moves whole 128-byte lines; 0.5 / 1.0: 64-byte requests)\n", t_half / t_full, t_half / t_halfbuf);
    return 0;
}
