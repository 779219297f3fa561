// What does an in-kernel grid barrier cost on MI355X (8 XCDs, one L2 each)?  G resident workgroups, agent-scope release / acquire on one
// counter, each workgroup writes a line before and reads another workgroup's line after every barrier (checked).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier tools/ubench/grid_barrier.hip && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k(unsigned* ctr, float* buf, int nbar, int* bad, long long* cyc) {
    const int G = gridDim.x, b = blockIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < nbar; ++i) {
        buf[(size_t)b * 256 + threadIdx.x] = (float)(i * 1000 + b);
        grid_barrier(ctr, (unsigned)(i + 1) * G);
        const int o = (b + 37) % G;
        const float v = buf[(size_t)o * 256 + threadIdx.x];
        if (v != (float)(i * 1000 + o)) atomicAdd(bad, 1);
        grid_barrier(ctr + 64, (unsigned)(i + 1) * G);      // second barrier: nobody overwrites before everyone has read
    }
    if (b == 0 && threadIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;
}
int main() {
    unsigned* ctr; float* buf; int* bad; long long* cyc;
    CHK(hipMalloc(&ctr, 1024)); CHK(hipMalloc(&buf, 1024 * 256 * 4)); CHK(hipMalloc(&bad, 4)); CHK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int G : {64, 128, 256, 512}) for (int nbar : {0, 8, 64}) {
        float best = 1e9f; int hb = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipMemset(ctr, 0, 1024)); CHK(hipMemset(bad, 0, 4));
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, ctr, buf, nbar, bad, cyc);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        }
        printf("G=%3d barriers=%3d (x2): %8.1f us total  -> %6.2f us per barrier pair   stale reads %d\n", G, nbar, best * 1e3, nbar ? best * 1e3 / nbar : 0.0, hb);
    }
    return 0;
}
