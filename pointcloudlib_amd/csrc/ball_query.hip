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

// Several radii around the same centres (multi-scale grouping: PointnetModuleMSG runs one BallQueryGrouper per scale on the same new_xyz,
// networks/seg/pointnet2_partseg.py:93-103): one scan of the cloud per query, the distance of a point formed once and tested against every
// radius; each radius keeps its own count / first hit / output row and stops taking hits at its nsample, the scan ends when all have.
// Per radius the same comparisons in the same order as ball_query_kernel: identical lists.
constexpr int BQ_MAXR = 4;
struct BqMulti { float r2[BQ_MAXR]; int ns[BQ_MAXR]; int32_t* idx[BQ_MAXR]; int32_t* cnt[BQ_MAXR]; };
template <bool USE_LDS, int NR>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_multi_kernel(const float* __restrict__ new_xyz, const float* __restrict__ xyz, int m,
                                                                      int N, int qpb, const BqMulti a) {
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
        const size_t gq = (size_t)b * m + q;
        const float* Q = new_xyz + gq * 3;
        const float cx = Q[0], cy = Q[1], cz = Q[2];
        int cnt[NR], first[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) { cnt[r] = 0; first[r] = 0; }
        for (int base = 0; base < N; base += 64) {
            bool live = false;
#pragma unroll
            for (int r = 0; r < NR; ++r) live |= cnt[r] < a.ns[r];
            if (!live) break;
            const int k = base + lane;
            float d = INFINITY;
            if (k < N) {
                float x, y, z;
                if (USE_LDS) { x = sx[k]; y = sy[k]; z = sz[k]; }
                else { x = P[3 * k]; y = P[3 * k + 1]; z = P[3 * k + 2]; }
                d = sq_dist3(cx, cy, cz, x, y, z);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool hit = d < a.r2[r];                      // (lanes past the cloud carry +inf: never a hit)
                const unsigned long long mask = __ballot(hit);
                if (mask && cnt[r] < a.ns[r]) {
                    if (cnt[r] == 0) first[r] = base + __ffsll((long long)mask) - 1;
                    const int slot = cnt[r] + mbcnt(mask);
                    if (hit && slot < a.ns[r]) a.idx[r][gq * a.ns[r] + slot] = k;
                    cnt[r] += __popcll(mask);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int c = min(cnt[r], a.ns[r]);
            for (int s = c + lane; s < a.ns[r]; s += 64) a.idx[r][gq * a.ns[r] + s] = first[r];   // first == 0 when no hit
            if (lane == 0 && a.cnt[r]) a.cnt[r][gq] = c;
        }
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

template <int NR>
static int launch_bq_multi(const float* new_xyz, const float* xyz, int B, int m, int N, const BqMulti& a, hipStream_t st) {
    const int qpb = 32;
    dim3 grid((m + qpb - 1) / qpb, B);
    const size_t lds = sizeof(float) * 3 * (size_t)N;
    if (lds <= 150 * 1024) {
        auto kern = ball_query_multi_kernel<true, NR>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(PCL_EHIP, "ball_query_multi: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, grid, dim3(BQ_THREADS), lds, st, new_xyz, xyz, m, N, qpb, a);
    } else {
        hipLaunchKernelGGL((ball_query_multi_kernel<false, NR>), grid, dim3(BQ_THREADS), 0, st, new_xyz, xyz, m, N, qpb, a);
    }
    return check_launch("pcl_ball_query_multi_f32");
}

extern "C" int pcl_ball_query_multi_f32(const float* new_xyz, const float* xyz, int B, int m, int N, int n_radii, const float* radii,
                                        const int32_t* nsamples, int32_t* const* idx_out, int32_t* const* cnt_out, void* stream) {
    PCL_REQUIRE(new_xyz && xyz && radii && nsamples && idx_out, "pcl_ball_query_multi_f32: null pointer");
    PCL_REQUIRE(n_radii >= 1 && n_radii <= BQ_MAXR, "pcl_ball_query_multi_f32: n_radii=%d (1..%d)", n_radii, BQ_MAXR);
    PCL_REQUIRE(B >= 0 && m >= 0 && N >= 1 && B <= 65535, "pcl_ball_query_multi_f32: bad sizes B=%d m=%d N=%d", B, m, N);
    BqMulti a = {};
    for (int r = 0; r < n_radii; ++r) {
        PCL_REQUIRE(nsamples[r] >= 1 && idx_out[r], "pcl_ball_query_multi_f32: radius %d: nsample=%d / null idx_out", r, nsamples[r]);
        a.r2[r] = radii[r] * radii[r];        // fp32 product, misc/ops.py:306
        a.ns[r] = nsamples[r]; a.idx[r] = idx_out[r]; a.cnt[r] = cnt_out ? cnt_out[r] : nullptr;
    }
    if (B == 0 || m == 0) return PCL_OK;
    hipStream_t st = as_stream(stream);
    switch (n_radii) {
        case 1: return launch_bq_multi<1>(new_xyz, xyz, B, m, N, a, st);
        case 2: return launch_bq_multi<2>(new_xyz, xyz, B, m, N, a, st);
        case 3: return launch_bq_multi<3>(new_xyz, xyz, B, m, N, a, st);
        default: return launch_bq_multi<4>(new_xyz, xyz, B, m, N, a, st);
    }
}
