// Micro-benchmark (round 5): which clock do s_memtime, s_memrealtime and GRBM_GUI_ACTIVE count, and what does the shader clock do
// under matrix load?   build: hipcc --offload-arch=gfx950 -O3 -o clock clock.hip ; run: ./clock
//
// A wave issues a chain of N DEPENDENT v_mfma_f32_32x32x2_f32 on one accumulator: 64 shader cycles each (MI355X_MICROARCH.md, "64 cyc
// dependent-accumulator latency"), whatever else the chip does.  Around the chain it reads s_memtime and s_memrealtime (constant
// 100 MHz).  If s_memtime is the shader clock, (s_memtime delta) / N = 64 at every load level, and (s_memtime delta) / (s_memrealtime
// delta x 10 ns) is the TRUE shader clock of that CU during the chain.  Load levels: one workgroup on an idle chip; every CU with one
// chain wave per SIMD; every CU with one chain wave + one full-rate wave (two independent accumulators) per SIMD = the matrix pipe of
// every SIMD saturated, the regime of the fused GEMM kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Rec { unsigned long long rt0, rt1; long long cy0, cy1; };

// waves 0-3 of a workgroup (one per SIMD): the dependent chain + the stamps; waves 4-7 (when launched): two independent accumulators,
// i.e. the pipe's full issue rate, for as long as the chain wave runs (iteration count matched by the host)
__global__ __launch_bounds__(512) void chain_kernel(Rec* out, int n, int n_load, float seed, float* sink) {
    const int wave = threadIdx.x >> 6;
    f32x16 a0 = {0}, a1 = {0};
    float x = seed + threadIdx.x * 1e-7f, y = seed + 1.f;
    if (wave < 4) {
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
        const long long cy0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; i += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        }
        // (the counter read must follow the last MFMA's result: make it depend on the accumulator)
        asm volatile("s_nop 0" :: "v"(a0[0]));
        const long long cy1 = __builtin_readcyclecounter();
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + wave] = Rec{rt0, rt1, cy0, cy1};
    } else {
        for (int i = 0; i < n_load; i += 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            }
        }
    }
    if (a0[0] + a1[3] == 12345.678f) sink[threadIdx.x] = a0[1];
}

__global__ void empty_kernel(float* sink) { if (sink && threadIdx.x == 9999) sink[0] = 1.f; }

static void run(const char* label, int blocks, int threads, int n, int n_load, Rec* d_out, float* d_sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Rec> h(blocks * 4);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(threads), 0, 0, d_out, n, n_load, 1.0f, d_sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d_out, sizeof(Rec) * blocks * 4, hipMemcpyDeviceToHost));
        std::vector<double> per, ghz;
        for (auto& r : h) {
            per.push_back(double(r.cy1 - r.cy0) / n);
            ghz.push_back(double(r.cy1 - r.cy0) / (double(r.rt1 - r.rt0) * 10.0));
        }
        std::sort(per.begin(), per.end()); std::sort(ghz.begin(), ghz.end());
        if (rep == 2)
            printf("%-58s s_memtime ticks per dependent MFMA: median %.2f (min %.2f max %.2f)   s_memtime / s_memrealtime: median %.3f GHz (min %.3f "
                   "max %.3f)   host events %.1f us\n", label, per[per.size() / 2], per.front(), per.back(), ghz[ghz.size() / 2], ghz.front(), ghz.back(), ms * 1e3);
    }
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    Rec* d_out; float* d_sink;
    CK(hipMalloc(&d_out, sizeof(Rec) * 4 * 1024)); CK(hipMalloc(&d_sink, 4096));
    const int n = 40000;            // 40 000 x 64 cycles = 2.56 M cycles ~ 1.1 ms: long enough for the clock to settle under load
    printf("CUs: %d; chain of %d dependent v_mfma_f32_32x32x2_f32 (64 shader cycles each by the ISA's pass count)\n", cus, n);
    run("1 workgroup, 1 chain wave per SIMD, idle chip", 1, 256, n, 0, d_out, d_sink);
    run("every CU, 1 chain wave per SIMD (pipe half busy)", cus, 256, n, 0, d_out, d_sink);
    // the load wave issues 2 MFMAs per 64-cycle slot pair => pipe saturated; it shares the pipe with the chain wave, which then takes
    // longer than 64 cycles per link in WALL terms -- the tick count per link says by how much
    run("every CU, chain wave + full-rate wave per SIMD (pipe saturated)", cus, 512, n, 2 * n, d_out, d_sink);
    run("1 workgroup again (clock recovery)", 1, 256, n, 0, d_out, d_sink);
    // launch overhead seen by events, for the GRBM_GUI_ACTIVE reading of an empty kernel under rocprofv3 --pmc GRBM_GUI_ACTIVE
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, d_sink);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, d_sink);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel, 256 workgroups, back to back: %.2f us per launch (events)\n", ms * 10.f);
    return 0;
}
