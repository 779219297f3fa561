// three_nn.hip -- 3-nearest-neighbour inverse-distance interpolation (feature propagation) for gfx950.
//
// Semantics: PointNetFeaturePropagation.execute, /root/reference/misc/ops.py:83-93.  The reference
// builds a dense [B,N,S] matrix in matmul form and full-argsorts it; here eight lanes per target point
// scan the S source points (staged in LDS as SoA) keeping running top-3 lists, merged at the end, by
// (direct-form d2, index) -- the definition pinned by oracle/pcl_oracle.c:pclo_three_nn_f32.
#include "common.h"

namespace pcl {

constexpr int TN_T = 256;
constexpr int TN_S = 8;                        // lanes per target point: each scans every TN_S-th source, the eight top-3 lists are merged

// insert (d, s) into the ascending triple (d0,i0) <= (d1,i1) <= (d2,i2), ordered by (distance, index)
__device__ __forceinline__ void tn_insert(float d, int s, float& d0, int& i0, float& d1, int& i1, float& d2, int& i2) {
    const bool b0 = d < d0 || (d == d0 && s < i0), b1 = d < d1 || (d == d1 && s < i1), b2 = d < d2 || (d == d2 && s < i2);
    d2 = b1 ? d1 : (b2 ? d : d2); i2 = b1 ? i1 : (b2 ? s : i2);
    d1 = b0 ? d0 : (b1 ? d : d1); i1 = b0 ? i0 : (b1 ? s : i1);
    d0 = b0 ? d : d0;             i0 = b0 ? s : i0;
}

// One lane per target scanned S sources in a chain of data-dependent branches: 2 048 targets x 512 sources on 128 workgroups took 70 us of the
// part-seg step (16.8 M distances).  Eight lanes per target each scan every eighth source in ascending order (strict <: among equal distances
// the lower index stays ahead, as in the single scan), branch-free inserts, then three xor-shuffle rounds merge the eight triples by
// (distance, index) -- the same top-3 in the same order as the sequential scan, whatever the split.
__global__ __launch_bounds__(TN_T) void three_nn_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                        int N, int S, int chunk, int32_t* __restrict__ idx3,
                                                        float* __restrict__ w3) {
    extern __shared__ __attribute__((aligned(16))) float s_src[];   // x[chunk] y[chunk] z[chunk]
    const int b = blockIdx.y, tid = threadIdx.x, sub = tid % TN_S;
    const int n = blockIdx.x * (TN_T / TN_S) + tid / TN_S;
    const float* P2 = xyz2 + (size_t)b * S * 3;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (n < N) { const float* p = xyz1 + ((size_t)b * N + n) * 3; px = p[0]; py = p[1]; pz = p[2]; }
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff;          // (empty slots lose every (distance, index) comparison)
    for (int s0 = 0; s0 < S; s0 += chunk) {
        const int len = min(chunk, S - s0);
        __syncthreads();
        for (int i = tid; i < 3 * len; i += TN_T) {
            const int k = i / 3, c = i - 3 * k;
            s_src[c * chunk + k] = P2[(size_t)s0 * 3 + i];
        }
        __syncthreads();
#pragma unroll 4
        for (int k = sub; k < len; k += TN_S) {
            const float d = sq_dist3(px, py, pz, s_src[k], s_src[chunk + k], s_src[2 * chunk + k]);
            const int s = s0 + k;
            const bool b0 = d < d0, b1 = d < d1, b2 = d < d2;      // ascending s within a lane: strict < keeps the lower index ahead
            d2 = b1 ? d1 : (b2 ? d : d2); i2 = b1 ? i1 : (b2 ? s : i2);
            d1 = b0 ? d0 : (b1 ? d : d1); i1 = b0 ? i0 : (b1 ? s : i1);
            d0 = b0 ? d : d0;             i0 = b0 ? s : i0;
        }
    }
#pragma unroll
    for (int off = 1; off < TN_S; off <<= 1) {
        const float e0 = __shfl_xor(d0, off), e1 = __shfl_xor(d1, off), e2 = __shfl_xor(d2, off);
        const int j0 = __shfl_xor(i0, off), j1 = __shfl_xor(i1, off), j2 = __shfl_xor(i2, off);
        tn_insert(e0, j0, d0, i0, d1, i1, d2, i2);
        tn_insert(e1, j1, d0, i0, d1, i1, d2, i2);
        tn_insert(e2, j2, d0, i0, d1, i1, d2, i2);
    }
    if (n >= N || sub != 0) return;
    int32_t* oi = idx3 + ((size_t)b * N + n) * 3;
    float* ow = w3 + ((size_t)b * N + n) * 3;
    if (S == 1) { oi[0] = oi[1] = oi[2] = 0; ow[0] = 1.f; ow[1] = 0.f; ow[2] = 0.f; return; }
    if (S == 2) { d2 = d1; i2 = i1; }
    const float r0 = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f));
    const float r1 = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f));
    const float r2 = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
    const float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
    oi[0] = i0; oi[1] = i1; oi[2] = i2;
    ow[0] = __fdiv_rn(r0, norm); ow[1] = __fdiv_rn(r1, norm); ow[2] = __fdiv_rn(r2, norm);
}

__global__ __launch_bounds__(256) void three_interp_kernel(const float* __restrict__ p2, const int32_t* __restrict__ idx3,
                                                           const float* __restrict__ w3, int N, int S, int D,
                                                           size_t total, float* __restrict__ out) {
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t row = g / D;                 // b*N + n
        const int c = (int)(g - row * D);
        const size_t b = row / N;
        const int32_t* ii = idx3 + row * 3;
        const float* ww = w3 + row * 3;
        const float* base = p2 + b * S * D + c;
        float acc = __fmul_rn(base[(size_t)ii[0] * D], ww[0]);
        acc = __fadd_rn(acc, __fmul_rn(base[(size_t)ii[1] * D], ww[1]));
        acc = __fadd_rn(acc, __fmul_rn(base[(size_t)ii[2] * D], ww[2]));
        out[g] = acc;
    }
}

__global__ __launch_bounds__(256) void three_interp_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ idx3,
                                                               const float* __restrict__ w3, int N, int S, int D,
                                                               size_t total, float* __restrict__ gp2) {
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t row = g / D;
        const int c = (int)(g - row * D);
        const size_t b = row / N;
        const int32_t* ii = idx3 + row * 3;
        const float* ww = w3 + row * 3;
        const float go = gout[g];
        float* base = gp2 + b * S * D + c;
#pragma unroll
        for (int j = 0; j < 3; ++j) unsafeAtomicAdd(&base[(size_t)ii[j] * D], go * ww[j]);
    }
}

static inline int grid_for(size_t total) {
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    return (int)(blocks ? blocks : 1);
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_three_nn_f32(const float* xyz1, const float* xyz2, int B, int N, int S, int32_t* idx3, float* w3,
                                void* stream) {
    PCL_REQUIRE(xyz1 && xyz2 && idx3 && w3, "pcl_three_nn_f32: null pointer");
    PCL_REQUIRE(B >= 0 && N >= 0 && S >= 1 && B <= 65535, "pcl_three_nn_f32: bad sizes B=%d N=%d S=%d", B, N, S);
    if (B == 0 || N == 0) return PCL_OK;
    const int chunk = S < 4096 ? S : 4096;
    hipLaunchKernelGGL(three_nn_kernel, dim3((N + TN_T / TN_S - 1) / (TN_T / TN_S), B), dim3(TN_T), sizeof(float) * 3 * chunk,
                       as_stream(stream), xyz1, xyz2, N, S, chunk, idx3, w3);
    return check_launch("pcl_three_nn_f32");
}

extern "C" int pcl_three_interp_f32(const float* points2, const int32_t* idx3, const float* w3, int B, int N, int S,
                                    int D, float* out, void* stream) {
    PCL_REQUIRE(points2 && idx3 && w3 && out && B >= 0 && N >= 0 && S >= 1 && D >= 1, "pcl_three_interp_f32: bad arguments");
    const size_t total = (size_t)B * N * D;
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(three_interp_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), points2, idx3, w3, N, S, D, total, out);
    return check_launch("pcl_three_interp_f32");
}

extern "C" int pcl_three_interp_bwd_f32(const float* gout, const int32_t* idx3, const float* w3, int B, int N, int S,
                                        int D, float* gpoints2, void* stream) {
    PCL_REQUIRE(gout && idx3 && w3 && gpoints2 && B >= 0 && N >= 0 && S >= 1 && D >= 1, "pcl_three_interp_bwd_f32: bad arguments");
    hipStream_t st = as_stream(stream);
    if ((size_t)B * S * D) {
        hipError_t e = hipMemsetAsync(gpoints2, 0, sizeof(float) * (size_t)B * S * D, st);
        if (e != hipSuccess) return fail(PCL_EHIP, "pcl_three_interp_bwd_f32: memset: %s", hipGetErrorString(e));
    }
    const size_t total = (size_t)B * N * D;
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(three_interp_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, gout, idx3, w3, N, S, D, total, gpoints2);
    return check_launch("pcl_three_interp_bwd_f32");
}
