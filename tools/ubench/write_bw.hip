// HBM write bandwidth on MI355X: what a write-heavy streaming kernel (a GEMM with Cout = 2 Cin writes 2 bytes per byte read) can hope for.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/write_bw tools/ubench/write_bw.hip && /tmp/write_bw
// Variants: pure write (dwordx4, full 128-byte lines per 8 lanes), pure read, copy 1:1, read:write 1:2; grid = 256 / 1024 / 4096 workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int RD, int WR>      // per iteration: RD float4 loads and WR float4 stores per thread
__global__ __launch_bounds__(256) void stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    float4 acc = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        if (RD) {
#pragma unroll
            for (int r = 0; r < RD; ++r) { const float4 v = src[(i + (size_t)r * n4) % (n4 * (RD ? RD : 1))]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
#pragma unroll
        for (int w = 0; w < WR; ++w) dst[i + (size_t)w * n4] = acc;
    }
    if (!WR && acc.x == 12345.f) dst[0] = acc;
}
template <int RD, int WR>
static void run(const char* name, const float4* src, float4* dst, size_t n4, int grid) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream<RD, WR>), dim3(grid), dim3(256), 0, 0, src, dst, n4);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double bytes = (double)n4 * 16 * (RD + WR);
    printf("%-22s grid %5d: %7.1f us  %6.2f TB/s (read %6.2f, write %6.2f)\n", name, grid, best * 1e3, bytes / best * 1e-9,
           (double)n4 * 16 * RD / best * 1e-9, (double)n4 * 16 * WR / best * 1e-9);
}
int main() {
    const size_t n4 = (size_t)6 << 20;           // 96 MB per stream unit
    float4 *src, *dst;
    CHK(hipMalloc(&src, n4 * 16 * 2)); CHK(hipMalloc(&dst, n4 * 16 * 2));
    CHK(hipMemset(src, 0, n4 * 16 * 2)); CHK(hipMemset(dst, 0, n4 * 16 * 2));
    for (int grid : {256, 512, 1024, 4096, 16384}) {
        run<0, 1>("write", src, dst, n4, grid);
        run<0, 2>("write x2", src, dst, n4, grid);
        run<1, 0>("read", src, dst, n4, grid);
        run<1, 1>("copy 1:1", src, dst, n4, grid);
        run<1, 2>("read 1 : write 2", src, dst, n4, grid);
        run<2, 1>("read 2 : write 1", src, dst, n4, grid);
    }
    return 0;
}
