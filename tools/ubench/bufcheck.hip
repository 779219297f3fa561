// Does the buffer range check of gfx950 include the scalar offset?  (raw buffer, stride 0)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/bufcheck tools/ubench/bufcheck.hip && /tmp/bufcheck
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* data, float* out) {
    // descriptor covers 64 bytes = 16 floats
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(data, 0, 64, 0x00020000);
    const unsigned lane = threadIdx.x;
    // case 0: voffset in range, soffset pushes the address past the range
    out[lane] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, 128, 0));
    // case 1: voffset out of range, soffset 0
    out[64 + lane] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 128 + lane * 4, 0, 0));
    // case 2: voffset in range only for lanes < 8, soffset 32 (address of lane 8.. is past 64 bytes only with soffset)
    out[128 + lane] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, 32, 0));
    // case 3: b128 load straddling the end: voffset 56 (+16 > 64)
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, 56, 0, 0);
    if (lane < 4) out[192 + lane] = __uint_as_float(v[lane]);
}
int main() {
    float *d, *o, h[256], src[256];
    for (int i = 0; i < 256; ++i) src[i] = 1000 + i;
    hipMalloc(&d, sizeof(src)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, src, sizeof(src), hipMemcpyHostToDevice);
    hipMemset(o, 0, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("case0 (voff<64, soff=128): lane0=%g lane15=%g lane16=%g  -> %s\n", h[0], h[15], h[16], h[0] != 0 ? "soffset NOT checked" : "soffset checked");
    printf("case1 (voff>=128):        lane0=%g\n", h[64]);
    printf("case2 (soff=32): lane7=%g lane8=%g lane15=%g lane16=%g\n", h[128 + 7], h[128 + 8], h[128 + 15], h[128 + 16]);
    printf("case3 (b128 at 56 of 64): %g %g %g %g\n", h[192], h[193], h[194], h[195]);
    return 0;
}
