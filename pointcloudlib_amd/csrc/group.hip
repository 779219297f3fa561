// group.hip -- grouped gathers and their scatter-add gradients for gfx950.
//
// Semantics: BallQueryGrouper.execute, /root/reference/misc/ops.py:383-407 (three Var.reindex gathers,
// `new_pointset - new_xyz.unsqueeze(2)`, concat [local_xyz, feature]); GroupAll.execute :415-419;
// index_points :12-27.  HBM-bound copies: one thread per output dword, lanes run along the channel
// axis so both the table reads (rows of C floats) and the output writes coalesce; the feature table
// ([B,N,C] <= a few MB) stays L2-resident across the ns-fold re-reads.
#include "common.h"

namespace pcl {

// caller-supplied point indices are clamped into [0, N) before they address memory (the reference's CUDA would read or
// atomically add out of bounds); valid lists are unaffected
__device__ __forceinline__ int in_cloud(int k, int N) { return min(max(k, 0), N - 1); }

constexpr int GT = 256;

__global__ __launch_bounds__(GT) void group_fwd_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                       const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                       int N, int m, int ns, int C, int use_xyz, size_t total,
                                                       float* __restrict__ out) {
    const int D = (use_xyz ? 3 : 0) + C;
    const int off = use_xyz ? 3 : 0;
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t row = g / D;               // (b*m + j)*ns + s
        const int c = (int)(g - row * D);
        const size_t bj = row / ns;             // b*m + j
        const size_t b = bj / m;
        const int k = in_cloud(idx[row], N);
        float v;
        if (c < off) v = __fsub_rn(xyz[(b * N + k) * 3 + c], new_xyz[bj * 3 + c]);
        else v = feat[(b * N + k) * C + (c - off)];
        out[g] = v;
    }
}

__global__ __launch_bounds__(GT) void group_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ idx,
                                                       int N, int m, int ns, int C, int use_xyz, size_t total,
                                                       float* __restrict__ gfeat) {
    const int D = (use_xyz ? 3 : 0) + C;
    const int off = use_xyz ? 3 : 0;
    // total = rows * C
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t row = g / C;
        const int c = (int)(g - row * C);
        const size_t b = row / ((size_t)m * ns);
        const int k = in_cloud(idx[row], N);
        unsafeAtomicAdd(&gfeat[(b * N + k) * C + c], gout[row * D + off + c]);
    }
}

__global__ __launch_bounds__(GT) void group_all_kernel(const float* __restrict__ xyz, const float* __restrict__ feat,
                                                       int C, int use_xyz, size_t total, float* __restrict__ out) {
    const int D = (use_xyz ? 3 : 0) + C;
    const int off = use_xyz ? 3 : 0;
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t p = g / D;
        const int c = (int)(g - p * D);
        out[g] = c < off ? xyz[p * 3 + c] : feat[p * C + (c - off)];
    }
}

__global__ __launch_bounds__(GT) void group_all_bwd_kernel(const float* __restrict__ gout, int C, int use_xyz,
                                                           size_t total, float* __restrict__ gfeat) {
    const int D = (use_xyz ? 3 : 0) + C;
    const int off = use_xyz ? 3 : 0;
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t p = g / C;
        const int c = (int)(g - p * C);
        gfeat[g] = gout[p * D + off + c];
    }
}

__global__ __launch_bounds__(GT) void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                         int N, int M, int C, size_t total, float* __restrict__ out) {
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t row = g / C;               // b*M + i
        const int c = (int)(g - row * C);
        const size_t b = row / M;
        out[g] = src[(b * N + in_cloud(idx[row], N)) * C + c];
    }
}

__global__ __launch_bounds__(GT) void gather_rows_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ idx,
                                                             int N, int M, int C, size_t total, float* __restrict__ gsrc) {
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t row = g / C;
        const int c = (int)(g - row * C);
        const size_t b = row / M;
        unsafeAtomicAdd(&gsrc[(b * N + in_cloud(idx[row], N)) * C + c], gout[g]);
    }
}


// DGCNN edge features, get_graph_feature (/root/reference/networks/cls/dgcnn.py:29-50) without the k-fold
// `repeat` of the centres: out[b,n,j,:] = concat(x[b,idx[b,n,j],:] - x[b,n,:], x[b,n,:])  -> [B,N,k,2C].
__global__ __launch_bounds__(GT) void edge_feature_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                          int N, int k, int C, size_t total, float* __restrict__ out) {
    const int D = 2 * C;
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t row = g / D;                // (b*N + n)*k + j
        const int c = (int)(g - row * D);
        const size_t bn = row / k;               // b*N + n
        const size_t b = bn / N;
        if (c < C) out[g] = __fsub_rn(x[(b * N + in_cloud(idx[row], N)) * C + c], x[bn * C + c]);
        else out[g] = x[bn * C + (c - C)];
    }
}

// gradient, centre part: gx[b,n,c] = sum_j (g[b,n,j,C+c] - g[b,n,j,c])   (plain store: initialises gx)
__global__ __launch_bounds__(GT) void edge_feature_bwd_center_kernel(const float* __restrict__ gout, int k, int C,
                                                                     size_t total, float* __restrict__ gx) {
    const int D = 2 * C;
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t bn = g / C;
        const int c = (int)(g - bn * C);
        const float* base = gout + bn * k * D;
        float s = 0.f;
        for (int j = 0; j < k; ++j) s += base[(size_t)j * D + C + c] - base[(size_t)j * D + c];
        gx[g] = s;
    }
}

// gradient, neighbour part: gx[b,idx[b,n,j],c] += g[b,n,j,c]
__global__ __launch_bounds__(GT) void edge_feature_bwd_nbr_kernel(const float* __restrict__ gout, const int32_t* __restrict__ idx,
                                                                  int N, int k, int C, size_t total, float* __restrict__ gx) {
    const int D = 2 * C;
    for (size_t g = (size_t)blockIdx.x * GT + threadIdx.x; g < total; g += (size_t)gridDim.x * GT) {
        const size_t row = g / C;
        const int c = (int)(g - row * C);
        const size_t b = row / ((size_t)N * k);
        unsafeAtomicAdd(&gx[(b * N + in_cloud(idx[row], N)) * C + c], gout[row * D + c]);
    }
}

static inline int grid_for(size_t total) {
    size_t blocks = (total + GT - 1) / GT;
    if (blocks > 256 * 16) blocks = 256 * 16;   // 16 resident blocks per CU, grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_group_f32(const float* xyz, const float* new_xyz, const float* feat, const int32_t* idx, int B,
                             int N, int m, int ns, int C, int use_xyz, float* out, void* stream) {
    PCL_REQUIRE(idx && out, "pcl_group_f32: null pointer");
    PCL_REQUIRE(!use_xyz || (xyz && new_xyz), "pcl_group_f32: use_xyz needs xyz and new_xyz");
    PCL_REQUIRE(C == 0 || feat, "pcl_group_f32: C=%d needs feat", C);
    PCL_REQUIRE(B >= 0 && N >= 1 && m >= 0 && ns >= 1 && C >= 0 && (use_xyz || C > 0), "pcl_group_f32: bad sizes");
    const size_t total = (size_t)B * m * ns * ((use_xyz ? 3 : 0) + C);
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(group_fwd_kernel, dim3(grid_for(total)), dim3(GT), 0, as_stream(stream), xyz, new_xyz, feat, idx,
                       N, m, ns, C, use_xyz, total, out);
    return check_launch("pcl_group_f32");
}

extern "C" int pcl_group_bwd_f32(const float* gout, const int32_t* idx, int B, int N, int m, int ns, int C,
                                 int use_xyz, float* gfeat, void* stream) {
    PCL_REQUIRE(gout && idx && gfeat, "pcl_group_bwd_f32: null pointer");
    PCL_REQUIRE(B >= 0 && N >= 1 && m >= 0 && ns >= 1 && C >= 1, "pcl_group_bwd_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    if ((size_t)B * N * C) {
        hipError_t e = hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * N * C, st);
        if (e != hipSuccess) return fail(PCL_EHIP, "pcl_group_bwd_f32: memset: %s", hipGetErrorString(e));
    }
    const size_t total = (size_t)B * m * ns * C;
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(group_bwd_kernel, dim3(grid_for(total)), dim3(GT), 0, st, gout, idx, N, m, ns, C, use_xyz, total, gfeat);
    return check_launch("pcl_group_bwd_f32");
}

extern "C" int pcl_group_all_f32(const float* xyz, const float* feat, int B, int N, int C, int use_xyz, float* out,
                                 void* stream) {
    PCL_REQUIRE(out && (!use_xyz || xyz) && (C == 0 || feat), "pcl_group_all_f32: null pointer");
    PCL_REQUIRE(B >= 0 && N >= 0 && C >= 0 && (use_xyz || C > 0), "pcl_group_all_f32: bad sizes");
    const size_t total = (size_t)B * N * ((use_xyz ? 3 : 0) + C);
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(group_all_kernel, dim3(grid_for(total)), dim3(GT), 0, as_stream(stream), xyz, feat, C, use_xyz, total, out);
    return check_launch("pcl_group_all_f32");
}

extern "C" int pcl_group_all_bwd_f32(const float* gout, int B, int N, int C, int use_xyz, float* gfeat, void* stream) {
    PCL_REQUIRE(gout && gfeat && C >= 1 && B >= 0 && N >= 0, "pcl_group_all_bwd_f32: bad arguments");
    const size_t total = (size_t)B * N * C;
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(group_all_bwd_kernel, dim3(grid_for(total)), dim3(GT), 0, as_stream(stream), gout, C, use_xyz, total, gfeat);
    return check_launch("pcl_group_all_bwd_f32");
}

extern "C" int pcl_gather_rows_f32(const float* src, const int32_t* idx, int B, int N, int M, int C, float* out,
                                   void* stream) {
    PCL_REQUIRE(src && idx && out && B >= 0 && N >= 1 && M >= 0 && C >= 1, "pcl_gather_rows_f32: bad arguments");
    const size_t total = (size_t)B * M * C;
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(GT), 0, as_stream(stream), src, idx, N, M, C, total, out);
    return check_launch("pcl_gather_rows_f32");
}

extern "C" int pcl_gather_rows_bwd_f32(const float* gout, const int32_t* idx, int B, int N, int M, int C, float* gsrc,
                                       void* stream) {
    PCL_REQUIRE(gout && idx && gsrc && B >= 0 && N >= 1 && M >= 0 && C >= 1, "pcl_gather_rows_bwd_f32: bad arguments");
    hipStream_t st = as_stream(stream);
    if ((size_t)B * N * C) {
        hipError_t e = hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * N * C, st);
        if (e != hipSuccess) return fail(PCL_EHIP, "pcl_gather_rows_bwd_f32: memset: %s", hipGetErrorString(e));
    }
    const size_t total = (size_t)B * M * C;
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(grid_for(total)), dim3(GT), 0, st, gout, idx, N, M, C, total, gsrc);
    return check_launch("pcl_gather_rows_bwd_f32");
}

extern "C" int pcl_edge_feature_f32(const float* x, const int32_t* idx, int B, int N, int k, int C, float* out, void* stream) {
    PCL_REQUIRE(x && idx && out && B >= 0 && N >= 1 && k >= 1 && C >= 1, "pcl_edge_feature_f32: bad arguments");
    const size_t total = (size_t)B * N * k * 2 * C;
    if (!total) return PCL_OK;
    hipLaunchKernelGGL(edge_feature_kernel, dim3(grid_for(total)), dim3(GT), 0, as_stream(stream), x, idx, N, k, C, total, out);
    return check_launch("pcl_edge_feature_f32");
}

extern "C" int pcl_edge_feature_bwd_f32(const float* gout, const int32_t* idx, int B, int N, int k, int C, float* gx,
                                        void* stream) {
    PCL_REQUIRE(gout && idx && gx && B >= 0 && N >= 1 && k >= 1 && C >= 1, "pcl_edge_feature_bwd_f32: bad arguments");
    hipStream_t st = as_stream(stream);
    const size_t tc = (size_t)B * N * C;
    if (!tc) return PCL_OK;
    hipLaunchKernelGGL(edge_feature_bwd_center_kernel, dim3(grid_for(tc)), dim3(GT), 0, st, gout, k, C, tc, gx);
    int rc = check_launch("pcl_edge_feature_bwd_f32(center)");
    if (rc) return rc;
    const size_t tn = tc * k;
    hipLaunchKernelGGL(edge_feature_bwd_nbr_kernel, dim3(grid_for(tn)), dim3(GT), 0, st, gout, idx, N, k, C, tn, gx);
    return check_launch("pcl_edge_feature_bwd_f32(nbr)");
}
