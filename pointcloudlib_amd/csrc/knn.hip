// knn.hip -- brute-force k nearest neighbours in feature space for gfx950 (DGCNN / PointCNN).
//
// Semantics: knn_cuda_global, /root/reference/misc/ops.py:562-638 =
//   compute_distances (:429-502): dist[b,r,q] = sum_c (ref[b,c,r]-qry[b,c,q])^2, `ssd += tmp*tmp` in
//   ascending c with separately rounded multiply and add, then
//   modified_insertion_sort (:504-552): per query the k smallest, ascending by (distance, r).
// Index exactness needs exactly that arithmetic, so the distance pass runs on the VALU with explicit
// single-rounded ops (the a^2+b^2-2ab MFMA form rounds differently and would reorder near-ties).
//
// Design: (1) 64x64 (query x ref) register-tiled distance kernel, channel chunks staged through LDS,
// written query-major so that (2) one wave per query loads its whole distance row into VGPRs
// (Nr/64 per lane) and extracts the k winners by k rounds of {lane-local min, DPP wave min, ballot}
// -- no per-thread serial insertion sort over global memory as in the reference.
#include "common.h"

namespace pcl {

constexpr int KD_T = 256, KD_TILE = 64, KD_CK = 32;

__global__ __launch_bounds__(KD_T) void knn_dist_kernel(const float* __restrict__ ref, const float* __restrict__ qry,
                                                        int C, int Nr, int Nq, float* __restrict__ dist) {
    __shared__ float sR[KD_CK][KD_TILE];
    __shared__ float sQ[KD_CK][KD_TILE];
    const int b = blockIdx.z, r0 = blockIdx.x * KD_TILE, q0 = blockIdx.y * KD_TILE;
    const int tid = threadIdx.x, tr = tid & 15, tq = tid >> 4;
    const float* R = ref + (size_t)b * C * Nr;
    const float* Q = qry + (size_t)b * C * Nq;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int c0 = 0; c0 < C; c0 += KD_CK) {
        for (int e = tid; e < KD_CK * KD_TILE; e += KD_T) {
            const int cc = e >> 6, x = e & 63;
            const int c = c0 + cc;
            sR[cc][x] = (c < C && r0 + x < Nr) ? R[(size_t)c * Nr + r0 + x] : 0.f;
            sQ[cc][x] = (c < C && q0 + x < Nq) ? Q[(size_t)c * Nq + q0 + x] : 0.f;
        }
        __syncthreads();
        const int cend = min(KD_CK, C - c0);
        for (int cc = 0; cc < cend; ++cc) {
            float a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sQ[cc][tq + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = sR[cc][tr + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = __fsub_rn(bb[j], a[i]);               // ref - query, :489
                    acc[i][j] = __fadd_rn(acc[i][j], __fmul_rn(t, t));    // ssd += tmp*tmp, :490
                }
        }
        __syncthreads();
    }
    float* D = dist + (size_t)b * Nq * Nr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + tq + 16 * i;
        if (q >= Nq) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + tr + 16 * j;
            if (r < Nr) D[(size_t)q * Nr + r] = acc[i][j];
        }
    }
}

template <int PPT>
__global__ __launch_bounds__(256) void knn_select_kernel(const float* __restrict__ dist, int Nr, int Nq, int k,
                                                         int32_t* __restrict__ idx_out) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Nq) return;
    const float* row = dist + ((size_t)b * Nq + q) * Nr;
    unsigned key[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int r = j * 64 + lane;
        key[j] = r < Nr ? __float_as_uint(row[r]) : 0xFFFFFFFFu;   // sums of squares: >= +0, bit-monotone
    }
    int32_t* out = idx_out + (size_t)b * k * Nq + q;
    for (int t = 0; t < k; ++t) {
        unsigned bk = 0xFFFFFFFFu, br = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const bool take = key[j] < bk;                          // strict: lowest r kept on ties
            bk = take ? key[j] : bk;
            br = take ? (unsigned)(j * 64 + lane) : br;
        }
        const unsigned wmin = wave_min_u32(bk);
        const unsigned long long tied = __ballot(bk == wmin);
        unsigned r;
        if (__popcll(tied) == 1) {
            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)tied) - 1);
            r = __builtin_amdgcn_readlane(br, l);
        } else {
            r = wave_min_u32(bk == wmin ? br : 0xFFFFFFFFu);
        }
        if (lane == 0) out[(size_t)t * Nq] = (int32_t)r;
        const unsigned rj = r >> 6, rl = r & 63;
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if ((unsigned)j == rj && (unsigned)lane == rl) key[j] = 0xFFFFFFFFu;
    }
}

// Any-Nr fallback: the row stays in the (caller-owned, scratch) workspace; winners are overwritten
// with +inf bits.  O(k*Nr) L2 reads per query.
__global__ __launch_bounds__(256) void knn_select_generic_kernel(float* __restrict__ dist, int Nr, int Nq, int k,
                                                                 int32_t* __restrict__ idx_out) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Nq) return;
    unsigned* row = reinterpret_cast<unsigned*>(dist + ((size_t)b * Nq + q) * Nr);
    int32_t* out = idx_out + (size_t)b * k * Nq + q;
    for (int t = 0; t < k; ++t) {
        unsigned bk = 0xFFFFFFFFu, br = 0xFFFFFFFFu;
        for (int r = lane; r < Nr; r += 64) {
            // agent-scope relaxed load: served by L2, so lane 0's store of the previous round is seen
            const unsigned kk = __hip_atomic_load(&row[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool take = kk < bk;
            bk = take ? kk : bk;
            br = take ? (unsigned)r : br;
        }
        const unsigned wmin = wave_min_u32(bk);
        const unsigned r = wave_min_u32(bk == wmin ? br : 0xFFFFFFFFu);
        if (lane == 0) {
            out[(size_t)t * Nq] = (int32_t)r;
            __hip_atomic_store(&row[r], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

}  // namespace pcl
using namespace pcl;

extern "C" size_t pcl_knn_workspace_bytes(int B, int C, int Nr, int Nq, int k) {
    (void)C; (void)k;
    if (B <= 0 || Nr <= 0 || Nq <= 0) return 0;
    return sizeof(float) * (size_t)B * Nr * Nq;
}

extern "C" int pcl_knn_f32(const float* ref, const float* qry, int B, int C, int Nr, int Nq, int k,
                           int32_t* idx_out, void* workspace, size_t workspace_bytes, void* stream) {
    PCL_REQUIRE(ref && qry && idx_out, "pcl_knn_f32: null pointer");
    PCL_REQUIRE(B >= 0 && C >= 1 && Nr >= 1 && Nq >= 1, "pcl_knn_f32: bad sizes B=%d C=%d Nr=%d Nq=%d", B, C, Nr, Nq);
    PCL_REQUIRE(k >= 1 && k <= Nr, "pcl_knn_f32: need 1 <= k <= Nr (k=%d Nr=%d)", k, Nr);
    PCL_REQUIRE(B <= 65535, "pcl_knn_f32: B=%d exceeds grid limit", B);
    if (B == 0) return PCL_OK;
    const size_t need = pcl_knn_workspace_bytes(B, C, Nr, Nq, k);
    if (!workspace || workspace_bytes < need)
        return fail(PCL_EWS, "pcl_knn_f32: workspace %zu bytes < required %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    float* dist = static_cast<float*>(workspace);
    dim3 g0((Nr + KD_TILE - 1) / KD_TILE, (Nq + KD_TILE - 1) / KD_TILE, B);
    hipLaunchKernelGGL(knn_dist_kernel, g0, dim3(KD_T), 0, st, ref, qry, C, Nr, Nq, dist);
    int rc = check_launch("pcl_knn_f32(dist)");
    if (rc) return rc;
    dim3 g1((Nq + 3) / 4, B);
    const int ppt = (Nr + 63) / 64;
#define PCL_KSEL(P) if (ppt <= P) { hipLaunchKernelGGL(knn_select_kernel<P>, g1, dim3(256), 0, st, dist, Nr, Nq, k, idx_out); return check_launch("pcl_knn_f32(select)"); }
    PCL_KSEL(1) PCL_KSEL(2) PCL_KSEL(4) PCL_KSEL(8) PCL_KSEL(16) PCL_KSEL(32) PCL_KSEL(64)
#undef PCL_KSEL
    hipLaunchKernelGGL(knn_select_generic_kernel, g1, dim3(256), 0, st, dist, Nr, Nq, k, idx_out);
    return check_launch("pcl_knn_f32(select-generic)");
}
