// What does it cost a streaming kernel to END with fp64 atomics instead of a partial row per workgroup, and what does a dependent
// tiny reduce kernel cost next to a consumer that reduces a few accumulator rows in its own prologue?
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics -o /tmp/atomic_f64 tools/ubench/atomic_f64.hip && /tmp/atomic_f64
// Chain per iteration, the BatchNorm hand-over of two consecutive layers in miniature:
//   A  producer (G workgroups x 512 threads: stream 64 KB each, then a partial row of 2C doubles) -> reduce kernel (rows -> 2C
//      doubles -> C floats) -> consumer (every workgroup reads the C floats, streams 64 KB)
//   B  producer ends with 2C fp64 atomics into one of R accumulator rows (row = blockIdx % R) -> consumer reduces the R rows in
//      its prologue (every workgroup) and streams; the accumulators are cleared by the consumer's workgroup 0 for the next round
//      of ANOTHER buffer (double-buffered), so no memset launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void producer(const float4* __restrict__ src, size_t n4, int C, double* __restrict__ rows, double* __restrict__ acc, int R, float* sink) {
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 512) { float4 v = src[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    const double a = (double)s.x + s.y, b = (double)s.z + s.w;
    if (s.x == 123.456f) sink[0] = s.y;
    for (int c = threadIdx.x; c < 2 * C; c += 512) {
        const double v = (c & 1) ? a : b;
        if (acc) unsafeAtomicAdd(&acc[(size_t)(blockIdx.x % R) * 2 * C + c], v);
        else rows[(size_t)blockIdx.x * 2 * C + c] = v;
    }
}
__global__ __launch_bounds__(256) void reduce(const double* __restrict__ rows, int nrows, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x & 3), ry = threadIdx.x >> 2;
    double s = 0, q = 0;
    for (int r = ry; r < nrows; r += 64) { s += rows[(size_t)r * 2 * C + c]; q += rows[(size_t)r * 2 * C + C + c]; }
    for (int off = 4; off < 64; off <<= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    __shared__ double red[32];
    if ((threadIdx.x & 63) < 4) { red[(threadIdx.x >> 6) * 4 + (threadIdx.x & 3)] = s; red[16 + (threadIdx.x >> 6) * 4 + (threadIdx.x & 3)] = q; }
    __syncthreads();
    if (threadIdx.x < 4) out[c] = (float)((red[threadIdx.x] + red[4 + threadIdx.x] + red[8 + threadIdx.x] + red[12 + threadIdx.x]) + 1e-3 * red[16 + threadIdx.x]);
}
__global__ __launch_bounds__(512) void consumer(const float4* __restrict__ src, size_t n4, int C, const float* __restrict__ cst, double* __restrict__ acc, int R,
                                                double* __restrict__ acc_clear, float* __restrict__ out) {
    __shared__ float sc[1024];
    if (acc) {
        for (int c = threadIdx.x; c < C; c += 512) {
            double s = 0, q = 0;
            for (int r = 0; r < R; ++r) { s += acc[(size_t)r * 2 * C + c]; q += acc[(size_t)r * 2 * C + C + c]; }
            sc[c] = (float)(s + 1e-3 * q);
        }
        if (blockIdx.x == 0) for (int c = threadIdx.x; c < 2 * C * R; c += 512) acc_clear[c] = 0.0;
    } else for (int c = threadIdx.x; c < C; c += 512) sc[c] = cst[c];
    __syncthreads();
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 512) { float4 v = src[i]; const float k = sc[i % C]; s.x += v.x * k; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x == 123.456f) out[0] = s.y + s.z + s.w;
}
int main() {
    const int G = 256, C = 128, ITER = 200;
    const size_t n4 = (size_t)G * 512 * 8;          // 16 MB streamed per kernel: ~5 us of work
    float4* src; double *rows, *acc0, *acc1; float *cst, *out;
    CHK(hipMalloc(&src, n4 * 16)); CHK(hipMemsetD32((hipDeviceptr_t)src, 0x3f800000, n4 * 4));
    CHK(hipMalloc(&rows, (size_t)1024 * 2 * 1024 * 8)); CHK(hipMalloc(&acc0, 64 * 2 * 1024 * 8)); CHK(hipMalloc(&acc1, 64 * 2 * 1024 * 8));
    CHK(hipMemset(acc0, 0, 64 * 2 * 1024 * 8)); CHK(hipMemset(acc1, 0, 64 * 2 * 1024 * 8));
    CHK(hipMalloc(&cst, 4096)); CHK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int g : {256, 1024}) for (int c : {128, 512}) {
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0));
            for (int i = 0; i < ITER; ++i) {
                hipLaunchKernelGGL(producer, dim3(g), dim3(512), 0, 0, src, n4, c, rows, (double*)nullptr, 1, out);
                hipLaunchKernelGGL(reduce, dim3(c / 4), dim3(256), 0, 0, rows, g, c, cst);
                hipLaunchKernelGGL(consumer, dim3(g), dim3(512), 0, 0, src, n4, c, cst, (double*)nullptr, 1, (double*)nullptr, out);
            }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("G=%4d C=%4d  A rows + reduce launch: %7.2f us per (producer, reduce, consumer)\n", g, c, ms * 1e3 / ITER);
        for (int R : {1, 4, 8, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipEventRecord(e0));
                for (int i = 0; i < ITER; ++i) {
                    double* a = (i & 1) ? acc1 : acc0; double* b = (i & 1) ? acc0 : acc1;
                    hipLaunchKernelGGL(producer, dim3(g), dim3(512), 0, 0, src, n4, c, rows, a, R, out);
                    hipLaunchKernelGGL(consumer, dim3(g), dim3(512), 0, 0, src, n4, c, cst, a, R, b, out);
                }
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("G=%4d C=%4d  B atomics into %2d rows, consumer-side reduce: %7.2f us per (producer, consumer)\n", g, c, R, ms * 1e3 / ITER);
        }
        // reference: producer + consumer alone (no hand-over at all)
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0));
            for (int i = 0; i < ITER; ++i) {
                hipLaunchKernelGGL(producer, dim3(g), dim3(512), 0, 0, src, n4, 0, rows, (double*)nullptr, 1, out);
                hipLaunchKernelGGL(consumer, dim3(g), dim3(512), 0, 0, src, n4, c, cst, (double*)nullptr, 1, (double*)nullptr, out);
            }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("G=%4d C=%4d  (no hand-over)                                   : %7.2f us\n", g, c, ms * 1e3 / ITER);
    }
    // correctness of the atomic sums
    CHK(hipMemset(acc0, 0, 64 * 2 * 1024 * 8));
    hipLaunchKernelGGL(producer, dim3(1024), dim3(512), 0, 0, src, n4, 128, rows, acc0, 8, out);
    double h[8 * 256]; CHK(hipMemcpy(h, acc0, sizeof h, hipMemcpyDeviceToHost));
    double t = 0; for (double v : h) t += v;
    printf("sum check: %.1f (expected %.1f)\n", t, 1024.0 * 256 * 16);
    return 0;
}
