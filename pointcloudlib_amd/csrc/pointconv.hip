// pointconv.hip -- PointConv's density-weighted per-point contraction for gfx950.
//
// Semantics (/root/reference/misc/pointconv_utils.py:393-394 and :319-320):
//     new_points = new_points * grouped_density                       [B,S,ns,C] * [B,S,ns,1]
//     out        = matmul(new_points^T [C x ns], weights [ns x 16])   -> [B,S,C*16]
// i.e. out[g,c,m] = sum_s feat[g,s,c] * dens[g,s] * w[g,s,m] for every group g = (b, point).
// The reference runs an elementwise multiply, a transpose copy and a batched GEMM of tiny (C x ns)(ns x 16) problems:
// three round trips of the [G,ns,C] tensor.  This is an HBM-bound streaming op (2*C*ns*16 flops per group against
// 4*ns*C + 64*C bytes), so it stays on the vector ALU: one workgroup per group, the density-scaled weights of the group
// in LDS (broadcast reads), one lane per channel streaming the group's rows coalesced.
//   backward:  d_feat[g,s,c] = dens[g,s] * sum_m dout[g,c,m] * w[g,s,m]
//              t[g,s,m]      = sum_c feat[g,s,c] * dout[g,c,m]
//              d_w[g,s,m]    = dens[g,s] * t[g,s,m],   d_dens[g,s] = sum_m w[g,s,m] * t[g,s,m]
#include "common.h"

namespace pcl {

constexpr int PC_M = 16;         // WeightNet's output width (pointconv_utils.py:236: WeightNet(3, 16))
constexpr int PC_SCH = 64;       // rows of a group staged per pass

// out[g,c,:] ; grid = G, block = 64..256 lanes over channels
__global__ __launch_bounds__(256) void pointconv_contract_kernel(const float* __restrict__ feat, const float* __restrict__ dens,
                                                                 const float* __restrict__ w, int ns, int C,
                                                                 float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float swd[PC_SCH * PC_M];
    const int g = blockIdx.x, tid = threadIdx.x;
    const float* F = feat + (size_t)g * ns * C;
    for (int c0 = 0; c0 < C; c0 += blockDim.x) {
        const int c = c0 + tid;
        float acc[PC_M];
#pragma unroll
        for (int m = 0; m < PC_M; ++m) acc[m] = 0.f;
        for (int s0 = 0; s0 < ns; s0 += PC_SCH) {
            const int len = min(PC_SCH, ns - s0);
            __syncthreads();
            for (int e = tid; e < len * PC_M; e += blockDim.x)
                swd[e] = w[((size_t)g * ns + s0) * PC_M + e] * dens[(size_t)g * ns + s0 + e / PC_M];
            __syncthreads();
            if (c < C) {
                for (int s = 0; s < len; ++s) {
                    const float f = F[(size_t)(s0 + s) * C + c];
                    const float4* q = reinterpret_cast<const float4*>(&swd[s * PC_M]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 v = q[j];
                        acc[4 * j] = fmaf(f, v.x, acc[4 * j]); acc[4 * j + 1] = fmaf(f, v.y, acc[4 * j + 1]);
                        acc[4 * j + 2] = fmaf(f, v.z, acc[4 * j + 2]); acc[4 * j + 3] = fmaf(f, v.w, acc[4 * j + 3]);
                    }
                }
            }
        }
        if (c < C) {
            float4* o = reinterpret_cast<float4*>(out + ((size_t)g * C + c) * PC_M);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        }
    }
}

// d_feat[g,s,c] = dens[g,s] * sum_m dout[g,c,m] * w[g,s,m]
__global__ __launch_bounds__(256) void pointconv_contract_bwd_feat_kernel(const float* __restrict__ dout, const float* __restrict__ dens,
                                                                          const float* __restrict__ w, int ns, int C,
                                                                          float* __restrict__ dfeat) {
    __shared__ __attribute__((aligned(16))) float swd[PC_SCH * PC_M];
    const int g = blockIdx.x, tid = threadIdx.x;
    float* DF = dfeat + (size_t)g * ns * C;
    for (int c0 = 0; c0 < C; c0 += blockDim.x) {
        const int c = c0 + tid;
        float d[PC_M];
        if (c < C) {
            const float4* q = reinterpret_cast<const float4*>(dout + ((size_t)g * C + c) * PC_M);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float4 v = q[j]; d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w; }
        }
        for (int s0 = 0; s0 < ns; s0 += PC_SCH) {
            const int len = min(PC_SCH, ns - s0);
            __syncthreads();
            for (int e = tid; e < len * PC_M; e += blockDim.x)
                swd[e] = w[((size_t)g * ns + s0) * PC_M + e] * dens[(size_t)g * ns + s0 + e / PC_M];
            __syncthreads();
            if (c < C) {
                for (int s = 0; s < len; ++s) {
                    float a = 0.f;
#pragma unroll
                    for (int m = 0; m < PC_M; ++m) a = fmaf(d[m], swd[s * PC_M + m], a);
                    DF[(size_t)(s0 + s) * C + c] = a;
                }
            }
        }
    }
}

// t[s,m] = sum_c feat[g,s,c] * dout[g,c,m];  d_w = dens * t;  d_dens = sum_m w * t.
// 256 threads = 64 rows x 4 quarter-rows of m; channels staged through LDS 64 at a time.
__global__ __launch_bounds__(256) void pointconv_contract_bwd_w_kernel(const float* __restrict__ feat, const float* __restrict__ dout,
                                                                       const float* __restrict__ dens, const float* __restrict__ w,
                                                                       int ns, int C, float* __restrict__ dw,
                                                                       float* __restrict__ ddens) {
    constexpr int CCH = 64;
    __shared__ float sf[PC_SCH][CCH + 1];
    __shared__ __attribute__((aligned(16))) float sd[CCH * PC_M];
    const int g = blockIdx.x, tid = threadIdx.x, sl = tid >> 2, mq = tid & 3;
    for (int s0 = 0; s0 < ns; s0 += PC_SCH) {
        const int len = min(PC_SCH, ns - s0);
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < C; c0 += CCH) {
            const int cl = min(CCH, C - c0);
            __syncthreads();
            for (int e = tid; e < len * CCH; e += 256) {
                const int s = e / CCH, c = e - s * CCH;
                sf[s][c] = c < cl ? feat[((size_t)g * ns + s0 + s) * C + c0 + c] : 0.f;
            }
            for (int e = tid; e < CCH * PC_M; e += 256) {
                const int c = e / PC_M;
                sd[e] = c < cl ? dout[((size_t)g * C + c0) * PC_M + e] : 0.f;
            }
            __syncthreads();
            if (sl < len) {
#pragma unroll 8
                for (int c = 0; c < CCH; ++c) {
                    const float f = sf[sl][c];
                    const float4 v = *reinterpret_cast<const float4*>(&sd[c * PC_M + 4 * mq]);
                    t[0] = fmaf(f, v.x, t[0]); t[1] = fmaf(f, v.y, t[1]); t[2] = fmaf(f, v.z, t[2]); t[3] = fmaf(f, v.w, t[3]);
                }
            }
        }
        float dd = 0.f;
        if (sl < len) {
            const size_t row = (size_t)g * ns + s0 + sl;
            const float4 wv = *reinterpret_cast<const float4*>(w + row * PC_M + 4 * mq);
            const float de = dens[row];
            *reinterpret_cast<float4*>(dw + row * PC_M + 4 * mq) = make_float4(de * t[0], de * t[1], de * t[2], de * t[3]);
            dd = wv.x * t[0] + wv.y * t[1] + wv.z * t[2] + wv.w * t[3];
        }
        dd += __shfl_xor(dd, 1); dd += __shfl_xor(dd, 2);            // the four quarter-rows of a row are adjacent lanes
        if (sl < len && mq == 0) ddens[(size_t)g * ns + s0 + sl] = dd;
    }
}

}  // namespace pcl
using namespace pcl;

static int pc_block(int C) { return C >= 256 ? 256 : (C + 63) / 64 * 64; }

extern "C" int pcl_pointconv_contract_f32(const float* feat, const float* density, const float* weights, int G, int ns, int C,
                                          int M, float* out, void* stream) {
    PCL_REQUIRE(feat && density && weights && out, "pcl_pointconv_contract_f32: null pointer");
    PCL_REQUIRE(G >= 1 && ns >= 1 && C >= 1 && M == PC_M, "pcl_pointconv_contract_f32: bad sizes G=%d ns=%d C=%d M=%d (M must be 16)", G, ns, C, M);
    hipLaunchKernelGGL(pointconv_contract_kernel, dim3(G), dim3(pc_block(C)), 0, as_stream(stream), feat, density, weights, ns, C, out);
    return check_launch("pcl_pointconv_contract_f32");
}

extern "C" int pcl_pointconv_contract_bwd_f32(const float* dout, const float* feat, const float* density, const float* weights,
                                              int G, int ns, int C, int M, float* dfeat, float* dweights, float* ddensity,
                                              void* stream) {
    PCL_REQUIRE(dout && feat && density && weights && dfeat && dweights && ddensity, "pcl_pointconv_contract_bwd_f32: null pointer");
    PCL_REQUIRE(G >= 1 && ns >= 1 && C >= 1 && M == PC_M, "pcl_pointconv_contract_bwd_f32: bad sizes G=%d ns=%d C=%d M=%d", G, ns, C, M);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(pointconv_contract_bwd_feat_kernel, dim3(G), dim3(pc_block(C)), 0, st, dout, density, weights, ns, C, dfeat);
    int rc = check_launch("pcl_pointconv_contract_bwd_f32(feat)");
    if (rc) return rc;
    hipLaunchKernelGGL(pointconv_contract_bwd_w_kernel, dim3(G), dim3(256), 0, st, feat, dout, density, weights, ns, C, dweights, ddensity);
    return check_launch("pcl_pointconv_contract_bwd_f32(w)");
}
