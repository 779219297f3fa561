// compact.hip -- duplicate-compacted ("ragged") grouping for ball-query groups, gfx950.
//
// query_ball_point pads every group to nsample slots with copies of its FIRST hit
// (/root/reference/misc/ops.py:321-324): at BASELINE config 2 the SA1 groups hold on average 30 distinct points
// in 64 slots.  Identical rows stay identical through conv/BN/ReLU, so the per-group MLP
// only needs the DISTINCT rows plus each row's multiplicity w (first hit: nsample - cnt + 1, others 1):
//   BatchNorm batch sums = sum_rows w*y, w*y^2 (identical to summing the padded rows), the max over the group is
//   unchanged, and in backward the dense BatchNorm term of a row counts w times.
// This file builds the compacted rows [P_eff, D] in (group, slot) order with their metadata, the ragged max-pool
// and the scatter-add of the input gradient.  P_eff stays on the device (no host sync); kernels take a capacity.
#include "common.h"

namespace pcl {

// goff[g] = sum_{h<g} max(cnt[h],1); goff[G] = P_eff.  One workgroup, 1024 lanes, sequential chunks + LDS scan.
__global__ __launch_bounds__(1024) void group_offsets_kernel(const int32_t* __restrict__ cnt, int G, int32_t* __restrict__ goff) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (G + 1023) / 1024;
    const int lo = min(t * per, G), hi = min(lo + per, G);
    int s = 0;
    for (int g = lo; g < hi; ++g) s += max(cnt[g], 1);
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;                       // exclusive prefix of this thread's chunk
    for (int g = lo; g < hi; ++g) { goff[g] = run; run += max(cnt[g], 1); }
    if (t == 1023) goff[G] = part[1023];
}

// one wave per group: rows off..off+c-1 = concat(xyz[idx]-new_xyz, feat[idx]) for the c distinct slots
__global__ __launch_bounds__(256) void group_compact_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                            const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ cnt, const int32_t* __restrict__ goff,
                                                            int G, int N, int m, int ns, int C, int use_xyz,
                                                            float* __restrict__ rows, int2* __restrict__ rmeta,
                                                            int32_t* __restrict__ rsrc) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    const int D = (use_xyz ? 3 : 0) + C, off3 = use_xyz ? 3 : 0;
    const int c = max(cnt[g], 1), base = goff[g];
    const int b = g / m;
    const int32_t* I = idx + (size_t)g * ns;
    // lanes run over the flattened (slot, channel) pairs so narrow rows (D = 6 at SA1) still fill the wave
    float* o = rows + (size_t)base * D;
    for (int e = lane; e < c * D; e += 64) {
        const int s = e / D, d = e - s * D;
        const int k = I[s];
        float v;
        if (d < off3) v = __fsub_rn(xyz[((size_t)b * N + k) * 3 + d], new_xyz[(size_t)g * 3 + d]);
        else v = feat[((size_t)b * N + k) * C + (d - off3)];
        o[e] = v;
    }
    for (int s = lane; s < c; s += 64) {
        const int mult = s == 0 ? ns - c + 1 : 1;
        rmeta[base + s] = make_int2(g, s | (mult << 16));
        rsrc[base + s] = b * N + I[s];
    }
}

// out[g,c] = max over the group's rows of lrelu(scale*y+shift); arg = compact row-in-group; ymax = y there
__global__ __launch_bounds__(256) void bn_act_max_rows_kernel(const float* __restrict__ Y, const int32_t* __restrict__ goff,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float slope, int G, int C, float* __restrict__ out,
                                                              int32_t* __restrict__ arg, float* __restrict__ ymax) {
    const size_t total = (size_t)G * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t g = e / C;
        const int c = (int)(e - g * C);
        const float a = scale[c], bsh = shift[c];
        const int r0 = goff[g], r1 = goff[g + 1];
        float best = -INFINITY, by = 0.f;
        int bi = 0;
        for (int r = r0; r < r1; ++r) {
            const float yy = Y[(size_t)r * C + c];
            const float u = fmaf(a, yy, bsh);
            const float z = u > 0.f ? u : u * slope;
            if (z > best) { best = z; bi = r - r0; by = yy; }
        }
        out[e] = best; arg[e] = bi; ymax[e] = by;
    }
}

// gfeat[rsrc[r], c] += grows[r, off+c] for r < *n_rows
__global__ __launch_bounds__(256) void scatter_rows_add_kernel(const float* __restrict__ grows, const int32_t* __restrict__ rsrc,
                                                               const int32_t* __restrict__ n_rows, int D, int off, int C,
                                                               float* __restrict__ gfeat) {
    const size_t total = (size_t)(*n_rows) * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t r = e / C;
        const int c = (int)(e - r * C);
        unsafeAtomicAdd(&gfeat[(size_t)rsrc[r] * C + c], grows[r * D + off + c]);
    }
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_group_compact_f32(const float* xyz, const float* new_xyz, const float* feat, const int32_t* idx,
                                     const int32_t* cnt, int B, int N, int m, int ns, int C, int use_xyz, float* rows,
                                     int32_t* row_meta, int32_t* row_src, int32_t* group_off, void* stream) {
    PCL_REQUIRE(idx && cnt && rows && row_meta && row_src && group_off, "pcl_group_compact_f32: null pointer");
    PCL_REQUIRE(!use_xyz || (xyz && new_xyz), "pcl_group_compact_f32: use_xyz needs xyz and new_xyz");
    PCL_REQUIRE(C == 0 || feat, "pcl_group_compact_f32: C=%d needs feat", C);
    PCL_REQUIRE(B >= 1 && N >= 1 && m >= 1 && ns >= 1 && ns < 32768 && C >= 0 && (use_xyz || C > 0), "pcl_group_compact_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    const int G = B * m;
    hipLaunchKernelGGL(group_offsets_kernel, dim3(1), dim3(1024), 0, st, cnt, G, group_off);
    int rc = check_launch("pcl_group_compact_f32(offsets)");
    if (rc) return rc;
    hipLaunchKernelGGL(group_compact_kernel, dim3((G + 3) / 4), dim3(256), 0, st, xyz, new_xyz, feat, idx, cnt, group_off, G, N, m, ns,
                       C, use_xyz, rows, reinterpret_cast<int2*>(row_meta), row_src);
    return check_launch("pcl_group_compact_f32");
}

extern "C" int pcl_bn_act_max_rows_f32(const float* Y, const int32_t* group_off, const float* scale, const float* shift,
                                       float slope, int G, int C, float* out, int32_t* arg, float* ymax, void* stream) {
    PCL_REQUIRE(Y && group_off && scale && shift && out && arg && ymax && G >= 1 && C >= 1, "pcl_bn_act_max_rows_f32: bad arguments");
    const size_t total = (size_t)G * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_act_max_rows_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), Y, group_off, scale, shift, slope, G, C,
                       out, arg, ymax);
    return check_launch("pcl_bn_act_max_rows_f32");
}

extern "C" int pcl_scatter_rows_add_f32(const float* grows, const int32_t* row_src, const int32_t* n_rows_dev, int rows_cap,
                                        int D, int off, int C, int n_dst_rows, float* gfeat, void* stream) {
    PCL_REQUIRE(grows && row_src && n_rows_dev && gfeat && rows_cap >= 1 && D >= 1 && off >= 0 && C >= 1 && off + C <= D && n_dst_rows >= 1,
                "pcl_scatter_rows_add_f32: bad arguments");
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)n_dst_rows * C, st);
    if (e != hipSuccess) return fail(PCL_EHIP, "pcl_scatter_rows_add_f32: memset: %s", hipGetErrorString(e));
    size_t blocks = ((size_t)rows_cap * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scatter_rows_add_kernel, dim3((int)blocks), dim3(256), 0, st, grows, row_src, n_rows_dev, D, off, C, gfeat);
    return check_launch("pcl_scatter_rows_add_f32");
}
