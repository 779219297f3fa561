// Micro-benchmark: can a VALU-only wave issue beside an MFMA-only wave of the SAME SIMD on gfx950?
// One workgroup of 8 waves per CU (2 per SIMD): waves 0-3 run `role_a`, waves 4-7 run `role_b`; roles: 0 idle, 1 fp32
// MFMA 32x32x2 chain (2 accumulators), 2 v_fma_f32 chain (8 independent), 3 LDS write+read, 4 MFMA 16x16x4 chain.
// Prints the wall time of {A alone, B alone, A+B}.   build: hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROLE>
__device__ __forceinline__ float work(int iters, float seed, float* lds) {
    if constexpr (ROLE == 1) {
        f32x16 a0 = {0}, a1 = {0};
        float x = seed, y = seed + 1.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            }
        }
        return a0[0] + a1[3];
    } else if constexpr (ROLE == 4) {
        f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = seed, y = seed + 1.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
            }
        }
        return a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (ROLE == 2) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = seed + u;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = fmaf(v[u], 1.0001f, 0.5f);
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
        return s;
    } else if constexpr (ROLE == 3) {
        float s = seed;
        float4* p = reinterpret_cast<float4*>(lds) + threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                *p = make_float4(s, s + 1.f, s + 2.f, s + 3.f);
                s += p[0].y;
            }
        }
        return s;
    } else {
        return seed;
    }
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(int ia, int ib, float* out) {
    __shared__ float lds[512 * 4];
    float r;
    if (threadIdx.x < 256) r = work<RA>(ia, (float)threadIdx.x, lds);
    else r = work<RB>(ib, (float)threadIdx.x, lds);
    if (r == 12345.678f) out[0] = r;
}

template <int RA, int RB>
static float run(int ia, int ib, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, ia, ib, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, ia, ib, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1000.f;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const int IM = 4000;      // 16 MFMA 32x32x2 per iteration = 1024 cycles
    // VALU iterations per iteration: 128 v_fma
    printf("mfma32x32x2 alone         : %8.1f us (%.1f TF)\n", run<1, 0>(IM, 0, d), 2.0 * 32 * 32 * 2 * 16 * IM * 4 * 256 / run<1, 0>(IM, 0, d) * 1e-6);
    printf("mfma16x16x4 alone         : %8.1f us (%.1f TF)\n", run<4, 0>(IM, 0, d), 2.0 * 16 * 16 * 4 * 32 * IM * 4 * 256 / run<4, 0>(IM, 0, d) * 1e-6);
    for (int iv : {1000, 2000, 4000, 8000}) {
        const float tv = run<0, 2>(0, iv, d), tb = run<1, 2>(IM, iv, d);
        printf("valu x%5d alone %8.1f us | mfma32 + valu %8.1f us | sum %8.1f max %8.1f\n", iv, tv, tb, tv + run<1, 0>(IM, 0, d), fmaxf(tv, run<1, 0>(IM, 0, d)));
    }
    for (int iv : {2000, 4000, 8000}) {
        const float tv = run<0, 2>(0, iv, d), tb = run<4, 2>(IM, iv, d);
        printf("valu x%5d alone %8.1f us | mfma16 + valu %8.1f us | mfma16 alone %8.1f\n", iv, tv, tb, run<4, 0>(IM, 0, d));
    }
    for (int il : {2000, 8000}) {
        const float tl = run<0, 3>(0, il, d), tb = run<1, 3>(IM, il, d);
        printf("lds  x%5d alone %8.1f us | mfma32 + lds  %8.1f us\n", il, tl, tb);
    }
    // both roles MFMA: two MFMA waves per SIMD share the pipe
    printf("mfma32 + mfma32 (2 waves/SIMD): %8.1f us\n", run<1, 1>(IM, IM, d));
    return 0;
}
