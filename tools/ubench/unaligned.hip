// Micro-benchmark (round 5): does raw_buffer_load_b128 accept a 4-byte-aligned (not 16-byte-aligned) address on gfx950 / ROCm 7.2, and what
// does it cost?  Rows of 259 floats (the GroupAll level's 3 + 256 columns): lane l reads the 16 bytes at row (l & 31), float (l >> 5) * 4 + 8 s.
//   build: hipcc --offload-arch=gfx950 -O3 -o unaligned unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k(const float* w, int ld, int steps, size_t bytes, float* out, int mode) {
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, (unsigned)bytes, 0x00020000);
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const unsigned row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + lr;
    const unsigned vo = row * ld * 4u + lh * 16u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int s = 0; s < steps; ++s) {
        if (mode == 0) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, s * 32, 0);
            acc.x += __uint_as_float(v.x); acc.y += __uint_as_float(v.y); acc.z += __uint_as_float(v.z); acc.w += __uint_as_float(v.w);
        } else {
            acc.x += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, s * 32, 0));
            acc.y += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo + 4, s * 32, 0));
            acc.z += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo + 8, s * 32, 0));
            acc.w += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo + 12, s * 32, 0));
        }
    }
    out[(size_t)row * 8 + lh * 4 + 0] = acc.x; out[(size_t)row * 8 + lh * 4 + 1] = acc.y;
    out[(size_t)row * 8 + lh * 4 + 2] = acc.z; out[(size_t)row * 8 + lh * 4 + 3] = acc.w;
}

int main() {
    const int rows = 256 * 4 * 32;
    for (int ld : {260, 259, 257}) {
        const int steps = 256 / 8;
        std::vector<float> h((size_t)rows * ld + 64);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f;
        float *d, *o;
        CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, (size_t)rows * 8 * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, ld, steps, h.size() * 4, o, mode);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<float> ho((size_t)rows * 8);
            CK(hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost));
            double err = 0;
            for (int r = 0; r < rows; ++r) for (int q = 0; q < 8; ++q) {
                double ref = 0; for (int s = 0; s < steps; ++s) ref += h[(size_t)r * ld + (q / 4) * 4 + 8 * s + q % 4];
                err = fmax(err, fabs(ref - ho[(size_t)r * 8 + q]));
            }
            printf("row stride %d floats, %s: %.1f us, max |err| %.2e %s\n", ld, mode == 0 ? "buffer_load_dwordx4" : "4 x buffer_load_dword", ms * 1e3, err, err < 1e-2 ? "OK" : "WRONG");
        }
        CK(hipFree(d)); CK(hipFree(o));
    }
    return 0;
}
