// What does the access pattern of the PointConv contraction kernels cost?  Persistent waves read a [rows][C] fp32 table in
// units of 64 rows; per step a wave loads either a 32-channel piece of its 64 rows (8 x 16-byte lanes per row: 128 B out of
// every C*4-byte row) or whole rows.  Pure streaming: the loaded values are summed and written once per wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rowpattern tools/ubench/rowpattern.hip && /tmp/rowpattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int MODE, int DEPTH>   // MODE 0: 32-channel chunks (strided 128 B pieces), 1: whole rows contiguous; DEPTH: chunks in flight
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, int rows, int C, float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = gridDim.x * 4, units = rows / 64, nch = C / 32;
    float4 acc = make_float4(0, 0, 0, 0);
    const int total = units * nch;
    // step t of this wave: linear index i = w + t * W over (unit, chunk) pairs (MODE 0) or contiguous 8 KB pieces (MODE 1)
    int w = blockIdx.x * 4 + wave;
    float4 buf[DEPTH][8];
    auto issue = [&](int slot, int i) {
        if (MODE == 0) {
            const int u = i / nch, c0 = (i % nch) * 32;
#pragma unroll
            for (int p = 0; p < 8; ++p) buf[slot][p] = *reinterpret_cast<const float4*>(x + ((size_t)u * 64 + p * 8 + (lane >> 3)) * C + c0 + (lane & 7) * 4);
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) buf[slot][p] = *reinterpret_cast<const float4*>(x + (size_t)i * 2048 + p * 256 + lane * 4);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (w + d * W < total) issue(d, w + d * W);
    for (int i = w; i < total; i += W * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (i + d * W >= total) break;
#pragma unroll
            for (int p = 0; p < 8; ++p) { acc.x += buf[d][p].x; acc.y += buf[d][p].y; acc.z += buf[d][p].z; acc.w += buf[d][p].w; }
            if (i + (d + DEPTH) * W < total) issue(d, i + (d + DEPTH) * W);
        }
    }
    out[(size_t)(blockIdx.x * 256 + threadIdx.x)] = acc.x + acc.y + acc.z + acc.w;
}
static bool COLD = false;
template <int MODE, int DEPTH>
void run(const float* x, int rows, int C, float* out, int grid, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // COLD: every launch reads another 134 MB window of a 1.07 GB buffer (4x the 256 MB Infinity Cache), else the same window
    const size_t win = (size_t)rows * C;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(grid), dim3(256), 0, 0, x + (COLD ? (i % 8) * win : 0), rows, C, out);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(grid), dim3(256), 0, 0, x + (COLD ? ((i + 3) % 8) * win : 0), rows, C, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)rows * C * 4;
    printf("  %-40s C=%3d grid=%4d: %7.1f us  %6.2f TB/s\n", name, C, grid, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
}
int main() {
    const int rows = 524288;
    float *x, *out;
    hipMalloc(&x, (size_t)rows * 64 * 4 * 8); hipMalloc(&out, 4096 * 256 * 4);
    hipMemset(x, 0, (size_t)rows * 64 * 4 * 8);
    for (int cold = 0; cold < 2; ++cold)
    for (int C : {64, 128}) {
        COLD = cold != 0;
        printf("--- %s\n", COLD ? "cold: a new 134 MB window of 1.07 GB per launch" : "warm: the same 134 MB every launch");
        const int r = C == 64 ? rows : rows / 2;      // 134 MB either way
        for (int grid : {512, 1024, 2048}) {
            run<0, 1>(x, r, C, out, grid, "32-channel chunks, 1 in flight");
            run<0, 2>(x, r, C, out, grid, "32-channel chunks, 2 in flight");
            run<1, 1>(x, r, C, out, grid, "contiguous 8 KB pieces, 1 in flight");
            run<1, 2>(x, r, C, out, grid, "contiguous 8 KB pieces, 2 in flight");
        }
    }
    return 0;
}
