// ball_query.hip -- radius neighbour search for gfx950.
//
// Semantics: query_ball_point_kernel, /root/reference/misc/ops.py:291-330: for each query, the
// first `nsample` point indices in ascending order whose squared distance is < fl(radius*radius),
// padded with the first hit.
// Design: one wave per query, 64 points per test; __ballot gives the 64-bit hit mask, mbcnt the
// ordered slot of each hit (order-preserving compaction with coalesced idx writes), early exit as
// soon as nsample hits are found.  The cloud is staged once per workgroup into LDS as SoA
// (conflict-free lane-consecutive reads) and shared by the QPB queries of the block.
#include "common.h"

namespace pcl {

constexpr int BQ_THREADS = 256;
constexpr int BQ_WAVES = BQ_THREADS / 64;

template <bool USE_LDS>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(const float* __restrict__ new_xyz,
                                                                const float* __restrict__ xyz, int m, int N,
                                                                float radius2, int nsample, int qpb,
                                                                int32_t* __restrict__ idx_out,
                                                                int32_t* __restrict__ cnt_out) {
    extern __shared__ __attribute__((aligned(16))) float s_pts[];   // x[N] y[N] z[N]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* P = xyz + (size_t)b * N * 3;
    float* sx = s_pts; float* sy = s_pts + N; float* sz = s_pts + 2 * N;
    if (USE_LDS) {
        for (int i = tid; i < 3 * N; i += BQ_THREADS) {
            const int k = i / 3, c = i - 3 * k;
            s_pts[c * N + k] = P[i];
        }
        __syncthreads();
    }
    const int q0 = blockIdx.x * qpb;
    const int q1 = min(q0 + qpb, m);
    for (int q = q0 + wid; q < q1; q += BQ_WAVES) {
        const float* Q = new_xyz + ((size_t)b * m + q) * 3;
        const float cx = Q[0], cy = Q[1], cz = Q[2];
        int32_t* row = idx_out + ((size_t)b * m + q) * nsample;
        int cnt = 0, first = 0;
        for (int base = 0; base < N && cnt < nsample; base += 64) {
            const int k = base + lane;
            bool hit = false;
            if (k < N) {
                float x, y, z;
                if (USE_LDS) { x = sx[k]; y = sy[k]; z = sz[k]; }
                else { x = P[3 * k]; y = P[3 * k + 1]; z = P[3 * k + 2]; }
                // (new_x - x)*(new_x - x) + (new_y - y)*(new_y - y) + (new_z - z)*(new_z - z)
                hit = sq_dist3(cx, cy, cz, x, y, z) < radius2;
            }
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                if (cnt == 0) first = base + __ffsll((long long)mask) - 1;
                const int slot = cnt + mbcnt(mask);
                if (hit && slot < nsample) row[slot] = k;
                cnt += __popcll(mask);
            }
        }
        cnt = min(cnt, nsample);
        for (int s = cnt + lane; s < nsample; s += 64) row[s] = first;   // first == 0 when no hit
        if (lane == 0 && cnt_out) cnt_out[(size_t)b * m + q] = cnt;
    }
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_ball_query_f32(const float* new_xyz, const float* xyz, int B, int m, int N, float radius,
                                  int nsample, int32_t* idx_out, int32_t* cnt_out, void* stream) {
    PCL_REQUIRE(new_xyz && xyz && idx_out, "pcl_ball_query_f32: null pointer");
    PCL_REQUIRE(B >= 0 && m >= 0 && N >= 1 && nsample >= 1, "pcl_ball_query_f32: bad sizes B=%d m=%d N=%d ns=%d", B, m, N, nsample);
    PCL_REQUIRE(B <= 65535, "pcl_ball_query_f32: B=%d exceeds grid.y limit", B);
    if (B == 0 || m == 0) return PCL_OK;
    const float radius2 = radius * radius;    // fp32 product, misc/ops.py:306
    const int qpb = 32;
    dim3 grid((m + qpb - 1) / qpb, B);
    const size_t lds = sizeof(float) * 3 * (size_t)N;
    hipStream_t st = as_stream(stream);
    if (lds <= 150 * 1024) {
        auto kern = ball_query_kernel<true>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(PCL_EHIP, "ball_query: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, grid, dim3(BQ_THREADS), lds, st, new_xyz, xyz, m, N, radius2, nsample, qpb, idx_out, cnt_out);
    } else {
        hipLaunchKernelGGL(ball_query_kernel<false>, grid, dim3(BQ_THREADS), 0, st, new_xyz, xyz, m, N, radius2, nsample, qpb, idx_out, cnt_out);
    }
    return check_launch("pcl_ball_query_f32");
}
