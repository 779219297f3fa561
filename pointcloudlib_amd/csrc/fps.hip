// fps.hip -- farthest point sampling for gfx950.
//
// Semantics: FurthestPointSampler, /root/reference/misc/ops.py:124-234 (see include/pcl_hip.h).
// Design (MI355X-first, not the reference's launch shape):
//   * one workgroup per cloud; the cloud is read from HBM exactly once, coalesced, and staged into
//     LDS as float4 (x,y,z,_) so the "last winner" lookup is one broadcast ds_read_b128;
//   * every lane keeps PPT points (xyz + running min-distance) in VGPRs for the whole
//     m-step chain -- no global or LDS traffic for the O(m*N) part;
//   * per step: lane-local arg-max, a DPP row reduction + 4 readlanes for the wave maximum, a
//     ballot to find the winning lane (the common no-tie case costs one readlane; exact ties fall
//     back to a wave min over the tie rank), one LDS slot per wave and ONE barrier per step
//     (double-buffered slots) to combine waves.
//   The op is a chain of m-1 dependent arg-max steps with only B independent problems: it is
//   latency-bound (reported as cycles/step in DESIGN.md), not HBM-bound.
//
// Tie rule (exact ties only): the reference's S-thread strided scan + tree reduction picks, among
// equal distances, the smallest (bitreverse_{log2 S}(k mod S), k).  We carry that as a 32-bit rank
// rank(k) = bitrev(k mod S) * ceil(N/S) + k div S, smaller wins.
#include "common.h"

namespace pcl {

struct FpsSlot { unsigned key, rank, k, pad; };

template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int N, int m, int log2S,
                                                double skip_thr, const int32_t* __restrict__ start_idx,
                                                int32_t* __restrict__ idx_out, float* __restrict__ new_xyz_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = T / 64;
    FpsSlot* slots = reinterpret_cast<FpsSlot*>(smem);                       // [2][NW]
    float4* s_xyz = reinterpret_cast<float4*>(smem + sizeof(FpsSlot) * 2 * (NW > 1 ? NW : 1));

    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    int32_t* out = idx_out + (size_t)b * m;
    float* oxyz = new_xyz_out ? new_xyz_out + (size_t)b * m * 3 : nullptr;

    // stage the cloud: coalesced dword reads of the AoS xyz, scattered into float4 slots
    float* s_flat = reinterpret_cast<float*>(s_xyz);
    for (int i = tid; i < 3 * N; i += T) {
        const int k = i / 3, c = i - 3 * k;
        s_flat[4 * k + c] = p[i];
    }
    __syncthreads();

    const unsigned S = 1u << log2S;
    const unsigned cnt = ((unsigned)N + S - 1) >> log2S;
    auto rank_of = [&](unsigned k) -> unsigned {
        const unsigned br = log2S ? (__brev(k & (S - 1)) >> (32 - log2S)) : 0u;
        return br * cnt + (k >> log2S);
    };
    float px[PPT], py[PPT], pz[PPT], md[PPT];     // md: running min distance, -1 = never a candidate
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = tid + j * T;
        if (k < N) {
            const float4 v = s_xyz[k];
            px[j] = v.x; py[j] = v.y; pz[j] = v.z;
            const float mag = __fadd_rn(__fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y)), __fmul_rn(v.z, v.z));
            md[j] = ((double)mag <= skip_thr) ? -1.0f : 1e10f;    // skip_thr < 0 disables the rule
        } else {
            px[j] = py[j] = pz[j] = 0.f; md[j] = -1.0f;
        }
    }

    int old = start_idx ? start_idx[b] : 0;
    if (tid == 0) out[0] = old;
    const int lane = tid & 63, wid = tid >> 6;

    for (int step = 1; step < m; ++step) {
        const float4 c = s_xyz[old];
        if (tid == 0 && oxyz) { oxyz[(step - 1) * 3 + 0] = c.x; oxyz[(step - 1) * 3 + 1] = c.y; oxyz[(step - 1) * 3 + 2] = c.z; }
        // lane-local arg-max of the updated min-distances (first maximum; exact ties are fixed up below)
        float best = -1.0f;
        unsigned bestk = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float d = sq_dist3(px[j], py[j], pz[j], c.x, c.y, c.z);
            const float d2 = fminf(d, md[j]);                     // dead points stay at -1 forever
            md[j] = d2;
            const bool gt = d2 > best;
            bestk = gt ? (unsigned)(tid + j * T) : bestk;
            best = gt ? d2 : best;
        }
        // wave arg-max: key is monotone in best for best >= 0; 0 means "no live point"
        const unsigned key = best >= 0.0f ? __float_as_uint(best) + 1u : 0u;
        const unsigned wmax = wave_max_u32(key);
        unsigned wr = 0xFFFFFFFFu, wk = 0;
        if (wmax) {
            const float wbest = __uint_as_float(wmax - 1u);
            int ntied = 0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) ntied += __popcll(__ballot(md[j] == wbest));
            if (ntied == 1) {                                      // the common case: one point attains the max
                const unsigned long long w = __ballot(key == wmax);
                const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)w) - 1);
                wk = __builtin_amdgcn_readlane(bestk, l);
                wr = rank_of(wk);
            } else {                                               // exact tie: smallest rank wins
                unsigned r = 0xFFFFFFFFu, rkk = 0;
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    const unsigned kj = (unsigned)(tid + j * T);
                    const unsigned rj = rank_of(kj);
                    const bool take = (md[j] == wbest) & (rj < r);
                    r = take ? rj : r; rkk = take ? kj : rkk;
                }
                wr = wave_min_u32(r);
                const unsigned long long w = __ballot(r == wr);
                const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)w) - 1);
                wk = __builtin_amdgcn_readlane(rkk, l);
            }
        }
        if (NW == 1) {
            old = wmax ? (int)wk : 0;
        } else {
            FpsSlot* sl = slots + (step & 1) * NW;
            if (lane == 0) { FpsSlot s; s.key = wmax; s.rank = wr; s.k = wk; s.pad = 0; sl[wid] = s; }
            __syncthreads();
            unsigned bk = 0, br = 0xFFFFFFFFu, bi = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const FpsSlot s = sl[w];
                const bool take = (s.key > bk) | ((s.key == bk) & (s.rank < br));
                bk = take ? s.key : bk; br = take ? s.rank : br; bi = take ? s.k : bi;
            }
            old = bk ? (int)bi : 0;
        }
        if (tid == 0) out[step] = old;
    }
    if (tid == 0 && oxyz) {
        const float4 c = s_xyz[old];
        oxyz[(m - 1) * 3 + 0] = c.x; oxyz[(m - 1) * 3 + 1] = c.y; oxyz[(m - 1) * 3 + 2] = c.z;
    }
}

// Large-cloud variant (N*16 B does not fit LDS next to the slots): only the running min-distance
// (4 B/point, -1 = never a candidate) lives in LDS, coordinates are re-read from L2 every step.
// Same arithmetic and tie rule as above.  Clouds beyond ~40k points are refused (PCL_ENOSUP).
template <int T>
__global__ __launch_bounds__(T) void fps_kernel_lds(const float* __restrict__ xyz, int N, int m, int log2S,
                                                    double skip_thr, const int32_t* __restrict__ start_idx,
                                                    int32_t* __restrict__ idx_out, float* __restrict__ new_xyz_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = T / 64;
    FpsSlot* slots = reinterpret_cast<FpsSlot*>(smem);
    float* s_md = reinterpret_cast<float*>(smem + sizeof(FpsSlot) * 2 * NW);  // [N] running min (-1 = dead)
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    int32_t* out = idx_out + (size_t)b * m;
    float* oxyz = new_xyz_out ? new_xyz_out + (size_t)b * m * 3 : nullptr;
    const unsigned S = 1u << log2S;
    const unsigned cnt = ((unsigned)N + S - 1) >> log2S;
    for (int k = tid; k < N; k += T) {
        const float x = p[3 * k], y = p[3 * k + 1], z = p[3 * k + 2];
        const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
        s_md[k] = ((double)mag <= skip_thr) ? -1.0f : 1e10f;
    }
    __syncthreads();
    int old = start_idx ? start_idx[b] : 0;
    if (tid == 0) out[0] = old;
    const int lane = tid & 63, wid = tid >> 6;
    for (int step = 1; step < m; ++step) {
        const float cx = p[3 * old], cy = p[3 * old + 1], cz = p[3 * old + 2];
        if (tid == 0 && oxyz) { oxyz[(step - 1) * 3 + 0] = cx; oxyz[(step - 1) * 3 + 1] = cy; oxyz[(step - 1) * 3 + 2] = cz; }
        float best = -1.0f;
        unsigned bestr = 0xFFFFFFFFu, bestk = 0;
        for (int k = tid; k < N; k += T) {
            const float mdk = s_md[k];
            const float d = sq_dist3(p[3 * k], p[3 * k + 1], p[3 * k + 2], cx, cy, cz);
            const float d2 = fminf(d, mdk);
            s_md[k] = d2;
            const unsigned lo = (unsigned)k & (S - 1);
            const unsigned brv = log2S ? (__brev(lo) >> (32 - log2S)) : 0u;
            const unsigned r = mdk < 0.f ? 0xFFFFFFFFu : brv * cnt + ((unsigned)k >> log2S);
            const bool take = (d2 > best) | ((d2 == best) & (r < bestr));
            best = take ? d2 : best; bestr = take ? r : bestr; bestk = take ? (unsigned)k : bestk;
        }
        const unsigned key = best >= 0.0f ? __float_as_uint(best) + 1u : 0u;
        const unsigned wmax = wave_max_u32(key);
        const unsigned r = key == wmax ? bestr : 0xFFFFFFFFu;
        const unsigned wr = wave_min_u32(r);
        const unsigned long long t2 = __ballot(key == wmax && r == wr);
        const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)t2) - 1);
        const unsigned wk = __builtin_amdgcn_readlane(bestk, l);
        FpsSlot* sl = slots + (step & 1) * NW;
        if (lane == 0) { FpsSlot s; s.key = wmax; s.rank = wr; s.k = wk; s.pad = 0; sl[wid] = s; }
        __syncthreads();
        unsigned bk = 0, br = 0xFFFFFFFFu, bi = 0;
        for (int w = 0; w < NW; ++w) {
            const FpsSlot s = sl[w];
            const bool take = (s.key > bk) | ((s.key == bk) & (s.rank < br));
            bk = take ? s.key : bk; br = take ? s.rank : br; bi = take ? s.k : bi;
        }
        old = bk ? (int)bi : 0;
        if (tid == 0) out[step] = old;
    }
    if (tid == 0 && oxyz) {
        oxyz[(m - 1) * 3 + 0] = p[3 * old]; oxyz[(m - 1) * 3 + 1] = p[3 * old + 1]; oxyz[(m - 1) * 3 + 2] = p[3 * old + 2];
    }
}

template <int T, int PPT>
static int launch_fps(const float* xyz, int B, int N, int m, int log2S, double thr, const int32_t* start,
                      int32_t* idx, float* nx, hipStream_t st) {
    constexpr int NW = T / 64;
    const size_t lds = sizeof(FpsSlot) * 2 * (NW > 1 ? NW : 1) + sizeof(float4) * (size_t)N;
    auto kern = fps_kernel<T, PPT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(PCL_EHIP, "fps: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(T), lds, st, xyz, N, m, log2S, thr, start, idx, nx);
    return check_launch("pcl_fps_f32");
}

}  // namespace pcl

using namespace pcl;

extern "C" int pcl_fps_f32(const float* xyz, int B, int N, int m, int tie_stride, double skip_sqnorm_le,
                           const int32_t* start_idx, int32_t* idx_out, float* new_xyz_out, void* stream) {
    PCL_REQUIRE(xyz && idx_out, "pcl_fps_f32: null pointer");
    PCL_REQUIRE(B >= 0 && N >= 1 && m >= 1 && m <= N, "pcl_fps_f32: need 1 <= m <= N (B=%d N=%d m=%d)", B, N, m);
    PCL_REQUIRE(tie_stride >= 1 && tie_stride <= 512 && (tie_stride & (tie_stride - 1)) == 0,
                "pcl_fps_f32: tie_stride must be a power of two in [1,512], got %d", tie_stride);
    if (B == 0) return PCL_OK;
    int log2S = 0;
    while ((1 << log2S) < tie_stride) ++log2S;
    hipStream_t st = as_stream(stream);
    static int env_t = -1;
    if (env_t < 0) { const char* e = getenv("PCL_FPS_THREADS"); env_t = e ? atoi(e) : 0; }
    // threads per cloud: ~4-8 points per lane keeps the VALU part and the cross-wave part balanced
    int T = env_t > 0 ? env_t : (N <= 256 ? 64 : N <= 1024 ? 256 : N <= 4096 ? 512 : 1024);
    const int ppt = (N + T - 1) / T;
#define PCL_FPS_CASE(TT, PP) \
    if (T == TT && ppt <= PP) return launch_fps<TT, PP>(xyz, B, N, m, log2S, skip_sqnorm_le, start_idx, idx_out, new_xyz_out, st);
    if ((size_t)N * 16 <= 150 * 1024) {
        PCL_FPS_CASE(64, 1) PCL_FPS_CASE(64, 2) PCL_FPS_CASE(64, 4) PCL_FPS_CASE(64, 8) PCL_FPS_CASE(64, 16) PCL_FPS_CASE(64, 32)
        PCL_FPS_CASE(128, 1) PCL_FPS_CASE(128, 2) PCL_FPS_CASE(128, 4) PCL_FPS_CASE(128, 8) PCL_FPS_CASE(128, 16)
        PCL_FPS_CASE(256, 1) PCL_FPS_CASE(256, 2) PCL_FPS_CASE(256, 4) PCL_FPS_CASE(256, 8) PCL_FPS_CASE(256, 16)
        PCL_FPS_CASE(512, 1) PCL_FPS_CASE(512, 2) PCL_FPS_CASE(512, 4) PCL_FPS_CASE(512, 8) PCL_FPS_CASE(512, 16)
        PCL_FPS_CASE(1024, 1) PCL_FPS_CASE(1024, 2) PCL_FPS_CASE(1024, 4) PCL_FPS_CASE(1024, 8) PCL_FPS_CASE(1024, 16)
    }
#undef PCL_FPS_CASE
    // large clouds: min-distance array in LDS (4 B/point), coordinates re-read from L2 each step
    const size_t lds = sizeof(FpsSlot) * 2 * 16 + sizeof(float) * (size_t)N;
    if (lds > 160 * 1024) return fail(PCL_ENOSUP, "pcl_fps_f32: N=%d exceeds the LDS-resident limit (%d points)", N, (160 * 1024 - 512) / 4);
    auto kern = fps_kernel_lds<1024>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(PCL_EHIP, "fps: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(1024), lds, st, xyz, N, m, log2S, skip_sqnorm_le, start_idx, idx_out, new_xyz_out);
    return check_launch("pcl_fps_f32(lds)");
}
