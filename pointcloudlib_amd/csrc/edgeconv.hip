// edgeconv.hip -- DGCNN's EdgeConv without the edge tensor, gfx950.
//
// Reference (/root/reference/networks/cls/dgcnn.py:29-50, :72-83, :100-111): gather the k nearest neighbours of every
// point, build e[i,j] = [x_nbr - x_i, x_i] (a [B,N,k,2C] tensor, 671 MB at the last stage), Conv2d 1x1 (no bias) +
// BatchNorm + LeakyReLU(0.2), max over the k neighbours.
// The 1x1 conv is linear, so with W = [Wa | Wb]
//     y[i,j] = Wa (x_nbr - x_i) + Wb x_i = U[nbr(i,j)] + V[i],   U = x Wa^T,  V = x (Wb - Wa)^T,
// one GEMM over the N points instead of the N*k edges (k = 20 times fewer flops, no edge tensor at all).  What is left
// is HBM/L2-bound streaming, done here:
//   * edgeconv_gather_kernel: y = U[nbr] + V on the fly; per channel the BatchNorm batch sums (fp64 partial rows, same
//     workspace layout as the GEMM epilogues in mlp.hip) and, per point, max/min of y over the neighbours with their
//     positions (the sign of the BatchNorm scale is not known yet; pcl_group_minmax_finalize_f32 picks afterwards);
//   * edgeconv_scatter_kernel: BatchNorm + max backward for every edge, dy = [j == arg] a gz - k1 - k2 (y - mean),
//     accumulated into dU[nbr] (atomics) and dV[i].
// Summation order differs from the reference's conv (y is formed from two fp32 products instead of one), within 1e-6.
#include "common.h"

namespace pcl {

constexpr int EC_PB = 32;        // points per workgroup == BatchNorm partial rows per 32 points

// UV [B*N, 2C] (U | V), idx [B*N, k] (neighbour index within the cloud)
__global__ __launch_bounds__(256) void edgeconv_gather_kernel(const float* __restrict__ UV, const int32_t* __restrict__ idx,
                                                              int N, int k, int C, size_t P /* B*N */, float* __restrict__ ymax,
                                                              float* __restrict__ ymin, int32_t* __restrict__ jmax,
                                                              int32_t* __restrict__ jmin, double* __restrict__ stats) {
    extern __shared__ double ssum[];                 // [2][C]
    const int tid = threadIdx.x;
    for (int c = tid; c < 2 * C; c += 256) ssum[c] = 0.0;
    __syncthreads();
    const size_t p0 = (size_t)blockIdx.x * EC_PB;
    const int items = EC_PB * C;
    for (int e = tid; e < items; e += 256) {
        const int pi = e / C, c = e - pi * C;
        const size_t p = p0 + pi;
        if (p >= P) break;
        const size_t base = (p / N) * N;             // first point of this cloud
        const float v = UV[p * 2 * C + C + c];
        const int32_t* I = idx + p * k;
        float vmax = -INFINITY, vmin = INFINITY;
        int imax = 0, imin = 0;
        double s = 0.0, q = 0.0;
        int j = 0;
        for (; j + 4 <= k; j += 4) {                 // four gathers in flight
            float u[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) u[t] = UV[(base + I[j + t]) * 2 * C + c];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float y = u[t] + v;
                s += (double)y; q += (double)y * (double)y;
                if (y > vmax) { vmax = y; imax = j + t; }
                if (y < vmin) { vmin = y; imin = j + t; }
            }
        }
        for (; j < k; ++j) {
            const float y = UV[(base + I[j]) * 2 * C + c] + v;
            s += (double)y; q += (double)y * (double)y;
            if (y > vmax) { vmax = y; imax = j; }
            if (y < vmin) { vmin = y; imin = j; }
        }
        const size_t o = p * C + c;
        ymax[o] = vmax; ymin[o] = vmin; jmax[o] = imax; jmin[o] = imin;
        atomicAdd(&ssum[c], s); atomicAdd(&ssum[C + c], q);
    }
    __syncthreads();
    double* dst = stats + (size_t)blockIdx.x * 2 * C;
    for (int c = tid; c < 2 * C; c += 256) dst[c] = ssum[c];
}

// dUV [B*N, 2C]: the U half must be zero on entry (atomics), the V half is written.
__global__ __launch_bounds__(256) void edgeconv_scatter_kernel(const float* __restrict__ UV, const int32_t* __restrict__ idx,
                                                               const float* __restrict__ gz, const int32_t* __restrict__ arg,
                                                               const float* __restrict__ a_, const float* __restrict__ k1_,
                                                               const float* __restrict__ k2_, const float* __restrict__ mu_,
                                                               int N, int k, int C, size_t P, float* __restrict__ dUV) {
    const int tid = threadIdx.x;
    const size_t p0 = (size_t)blockIdx.x * EC_PB;
    const int items = EC_PB * C;
    for (int e = tid; e < items; e += 256) {
        const int pi = e / C, c = e - pi * C;
        const size_t p = p0 + pi;
        if (p >= P) break;
        const size_t base = (p / N) * N;
        const float v = UV[p * 2 * C + C + c];
        const float a = a_[c], k1 = k1_[c], k2 = k2_[c], mu = mu_[c];
        const float g = a * gz[p * C + c];
        const int ja = arg[p * C + c];
        const int32_t* I = idx + p * k;
        float dv = 0.f;
        int j = 0;
        for (; j + 4 <= k; j += 4) {
            int n[4];
            float u[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { n[t] = I[j + t]; u[t] = UV[(base + n[t]) * 2 * C + c]; }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float y = u[t] + v;
                const float d = (j + t == ja ? g : 0.f) - fmaf(k2, y - mu, k1);
                dv += d;
                unsafeAtomicAdd(&dUV[(base + n[t]) * 2 * C + c], d);
            }
        }
        for (; j < k; ++j) {
            const int n = I[j];
            const float y = UV[(base + n) * 2 * C + c] + v;
            const float d = (j == ja ? g : 0.f) - fmaf(k2, y - mu, k1);
            dv += d;
            unsafeAtomicAdd(&dUV[(base + n) * 2 * C + c], d);
        }
        dUV[p * 2 * C + C + c] = dv;
    }
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_edgeconv_stat_rows(int B, int N) {
    if (B < 1 || N < 1) return 0;
    return (int)(((size_t)B * N + EC_PB - 1) / EC_PB);
}

extern "C" int pcl_edgeconv_gather_f32(const float* UV, const int32_t* idx, int B, int N, int k, int C, float* ymax, float* ymin,
                                       int32_t* jmax, int32_t* jmin, double* stats_ws, void* stream) {
    PCL_REQUIRE(UV && idx && ymax && ymin && jmax && jmin && stats_ws, "pcl_edgeconv_gather_f32: null pointer");
    PCL_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= N && C >= 1 && C <= 2048, "pcl_edgeconv_gather_f32: bad sizes B=%d N=%d k=%d C=%d", B, N, k, C);
    const size_t P = (size_t)B * N;
    const int blocks = pcl_edgeconv_stat_rows(B, N);
    hipLaunchKernelGGL(edgeconv_gather_kernel, dim3(blocks), dim3(256), sizeof(double) * 2 * C, as_stream(stream), UV, idx, N, k, C, P,
                       ymax, ymin, jmax, jmin, stats_ws);
    return check_launch("pcl_edgeconv_gather_f32");
}

extern "C" int pcl_edgeconv_scatter_f32(const float* UV, const int32_t* idx, const float* gz, const int32_t* arg, const float* a,
                                        const float* k1, const float* k2, const float* mu, int B, int N, int k, int C, float* dUV,
                                        void* stream) {
    PCL_REQUIRE(UV && idx && gz && arg && a && k1 && k2 && mu && dUV, "pcl_edgeconv_scatter_f32: null pointer");
    PCL_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= N && C >= 1, "pcl_edgeconv_scatter_f32: bad sizes B=%d N=%d k=%d C=%d", B, N, k, C);
    const size_t P = (size_t)B * N;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(dUV, 0, sizeof(float) * P * 2 * C, st);
    if (e != hipSuccess) return fail(PCL_EHIP, "pcl_edgeconv_scatter_f32: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(edgeconv_scatter_kernel, dim3((int)((P + EC_PB - 1) / EC_PB)), dim3(256), 0, st, UV, idx, gz, arg, a, k1, k2, mu, N, k,
                       C, P, dUV);
    return check_launch("pcl_edgeconv_scatter_f32");
}
