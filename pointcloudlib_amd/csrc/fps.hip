// fps.hip -- farthest point sampling for gfx950.
//
// Semantics: FurthestPointSampler, /root/reference/misc/ops.py:124-234 (see include/pcl_hip.h).
// Design (MI355X-first, not the reference's launch shape):
//   * one workgroup per cloud; the cloud is read from HBM exactly once, coalesced, and staged into
//     LDS as float4 (x,y,z,_) so the "last winner" lookup is one broadcast ds_read_b128;
//   * every lane keeps PPT points (xyz + running min-distance) in VGPRs for the whole
//     m-step chain -- no global or LDS traffic for the O(m*N) part;
//   * per step: lane-local max of a 64-bit key (distance bits | inverted tie rank) with one v_max_f64 per
//     point, a 6-step DPP reduction for the wave maximum, one 8-byte LDS slot per wave and ONE barrier per
//     step (double-buffered slots) to combine waves; the winner's index is recovered from the key by shifts;
//   * nothing is written to global memory inside the chain (indices are buffered in LDS).
//   The op is a chain of m-1 dependent arg-max steps with only B independent problems: it is
//   latency-bound (~940 shader cycles per step at N=1024, of which ~45 % are LDS/barrier waits; PMC in
//   profiles/), not HBM-bound.
//
// Tie rule (exact ties only): the reference's S-thread strided scan + tree reduction picks, among
// equal distances, the smallest (bitreverse_{log2 S}(k mod S), k).  We carry that as a 32-bit rank
// rank'(k) = (bitrev(k mod S) << sh) | (k div S), smaller wins (see the key layout below).
#include "common.h"

namespace pcl {

struct FpsSlot { unsigned key, rank, k, pad; };

// 64-bit arg-max key: high word = bits of the (non-negative) running min-distance, low word = ~rank'(k), so
// one unsigned 64-bit max picks the largest distance and, among exactly equal distances, the smallest rank.
//   rank'(k) = (bitrev_{log2 S}(k mod S) << sh) | (k div S),  sh = bits of ((N-1) div S)
// is order-isomorphic to the reference's (bitrev(k mod S), k) priority and invertible with shifts only.
// Points that may never be sampled carry key 0 (distance 0, low word 0) and lose to every live point;
// a cloud without live points reduces to key 0 -> index 0 (misc/ops.py:152-153).
// The keys are compared as IEEE doubles: for bit patterns with a clear sign bit and a non-NaN exponent (high
// word = bits of a finite non-negative float, so the f64 exponent field is < 0x7F8) numeric order == unsigned
// integer order, and v_max_f64 does in ONE instruction what a 64-bit integer max needs five for.  The step
// chain is latency-bound, so instruction count on the dependent path is what matters.
__device__ __forceinline__ double key_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)u, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(u >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// the same for the stages in which EVERY lane receives a valid value (full row mask, a permutation inside the row): no `old` operand, so
// hipcc does not copy the value first (two v_mov_b32 + an s_nop per stage on the dependent chain; round 6)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_perm(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// wave-wide max of a key: 4 DPP butterfly steps inside each row of 16 lanes, row_bcast:15 / row_bcast:31 fold
// the four rows, the total is read from lane 63 (wave-uniform).  Measured on MI355X: 382 ns/step at N=1024
// with 4 waves; a two-pass 32-bit variant with the DPP modifier fused into v_max_u32 (fewer instructions, two
// VALU->SGPR hops) measured 401 ns/step, the original compare/select/ballot form 582 ns/step.
__device__ __forceinline__ double wave_key_max_lane63(double v) {            // the wave maximum, valid in lane 63 (only)
    v = key_max(v, dpp_f64_perm<0xB1>(v));      // quad_perm [1,0,3,2]
    v = key_max(v, dpp_f64_perm<0x4E>(v));      // quad_perm [2,3,0,1]
    v = key_max(v, dpp_f64_perm<0x141>(v));     // row_half_mirror
    v = key_max(v, dpp_f64_perm<0x140>(v));     // row_mirror  -> every lane holds its row's max
    // the two row folds with FULL row masks: rows that receive nothing read 0 (bound_ctrl) -- the identity of a max over keys >= +0.0 -- and
    // the rows that receive more than they need only see more of the same maximum; row 3 ends with all four rows either way
    v = key_max(v, dpp_f64_perm<0x142>(v));     // row_bcast:15: lane 15 of a row into the next row
    v = key_max(v, dpp_f64_perm<0x143>(v));     // row_bcast:31: lane 31 into rows 2,3 -> lane 63 has the wave max
    return v;
}
__device__ __forceinline__ unsigned long long wave_key_max(double v) {
    v = wave_key_max_lane63(v);
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

#ifndef PCL_EXP
#define PCL_EXP 0
#endif
#if PCL_EXP == 6
// lab build (`make EXP=6`, tools/fps_budget.py): shader cycles (s_memtime) per phase of the dependent step, summed over the chain by
// thread 0 of cloud 0 -- [0] centre broadcast (LDS), [1] distances + lane-local key max, [2] wave reduction (DPP + readlane),
// [3] cross-wave exchange (LDS slot, barrier, slot reads, maxima), [4] rank decode, [5] steps, [6] whole chain
__device__ long long g_fps_ph[8];
#define FPS_STAMP(i, dep) { asm volatile("" : "+v"(dep)); __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_readcyclecounter(); \
                            __builtin_amdgcn_sched_barrier(0); ph[i] += t_ - tl; tl = t_; }
#else
#define FPS_STAMP(i, dep)
#endif

// Round 6 -- the step chain, shortened where the cycle budget (tools/fps_budget.py, `make EXP=6`) showed instructions that carry no
// information: (1) the cloud sits in LDS indexed by the tie RANK rank'(k), not by k, so the winner's key IS the address of the next
// centre -- `~low word` clamped -- and the ten-instruction rank -> index decode left the chain (the m indices are decoded together,
// in parallel, after it); (2) the next centre's LDS read is issued before the step's index store; (3) the four in-row DPP stages move
// the key without first copying it (dpp_f64_perm); (4) with several waves, lane 63 stores the wave's maximum itself instead of going
// through two v_readlane + a move back into a vector register.  Same arithmetic, same keys, same winners.
template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int N, int m, int log2S,
                                                double skip_thr, const int32_t* __restrict__ start_idx,
                                                int32_t* __restrict__ idx_out, float* __restrict__ new_xyz_out, int prio) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = T / 64;
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem);      // [2][NW] (NW > 1)
    // [RT + 1] entries indexed by rank'(k); entry RT = point 0.  16 bytes per entry (one ds_read_b96 for the next centre) up to 1 024 points;
    // beyond, 12 bytes: at N = 4 096 the table is 48 KB instead of 64 and the chain still fits beside the 105 KB workgroups of the first
    // level's forward GEMMs when it runs a batch ahead on the side stream (with 70 KB it did not: those kernels waited for the CUs it held)
    constexpr int ES = T * PPT > 1024 ? 3 : 4;
    float* s_flat = reinterpret_cast<float*>(smem + 256);

    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    int32_t* out = idx_out + (size_t)b * m;
    float* oxyz = new_xyz_out ? new_xyz_out + (size_t)b * m * 3 : nullptr;
    // a serial chain: `prio` (pcl_set_fps_tuning) decides who wins issue arbitration when it is co-resident with throughput kernels.
    // Inline in a forward pass the chain IS the critical path (3); one batch ahead on a side stream it has a whole backward pass to
    // finish in and every slot it takes is taken from a kernel the step waits for (0).
    if (prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);

    const unsigned S = 1u << log2S;
    int sh = 0;
    while ((((unsigned)N - 1) >> log2S) >> sh) ++sh;                    // bits of (N-1) div S
    const unsigned RT = S << sh;                                         // ranks are < RT (a power of two, N <= RT < 2 N)
    unsigned* s_out = reinterpret_cast<unsigned*>(smem + 256 + (((size_t)ES * 4 * ((size_t)RT + 1) + 15) & ~(size_t)15));  // [m] sampled RANKS
    auto entry = [&](unsigned rk) -> float4 {
        if constexpr (ES == 4) return *reinterpret_cast<const float4*>(s_flat + 4 * rk);
        else { const float* q = s_flat + 3 * rk; return make_float4(q[0], q[1], q[2], 0.f); }
    };
    auto rank_of = [&](unsigned k) -> unsigned {
        const unsigned br = log2S ? (__brev(k & (S - 1)) >> (32 - log2S)) : 0u;
        return (br << sh) | (k >> log2S);
    };
    auto index_of = [&](unsigned r) -> unsigned {                       // inverse of rank_of; the clamp value RT (no live point) -> 0
        if (r >= RT) return 0u;
        const unsigned kdiv = r & ((1u << sh) - 1u), br = r >> sh;
        return (kdiv << log2S) | (log2S ? (__brev(br) >> (32 - log2S)) : 0u);
    };

    // stage the cloud: coalesced dword reads of the AoS xyz, scattered into the float4 slot of the point's rank
    for (int i = tid; i < 3 * N; i += T) {
        const int k = i / 3, c = i - 3 * k;
        s_flat[ES * rank_of((unsigned)k) + c] = p[i];
    }
    if (tid < 3) s_flat[ES * RT + tid] = p[tid];                          // entry RT: point 0 (misc/ops.py:152-153: a cloud without live points samples index 0)
    __syncthreads();

    float px[PPT], py[PPT], pz[PPT], md[PPT];
    unsigned lo[PPT];                                                    // ~rank'(k), 0 = never a candidate
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const unsigned k = (unsigned)(tid + j * T);
        px[j] = py[j] = pz[j] = 0.f; md[j] = 0.f; lo[j] = 0u;
        if (k < (unsigned)N) {
            const unsigned rk = rank_of(k);
            const float4 v = entry(rk);
            px[j] = v.x; py[j] = v.y; pz[j] = v.z;
            const float mag = __fadd_rn(__fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y)), __fmul_rn(v.z, v.z));
            if (!((double)mag <= skip_thr)) {                            // skip_thr < 0 disables the rule
                lo[j] = ~rk;
                md[j] = 1e10f;
            }
        }
    }

    // NOTE: nothing is stored to global memory inside the chain: a global store per step would be drained
    // (vmcnt(0)) by every __syncthreads(), adding an HBM round trip to each of the m-1 dependent steps.
    const int old0 = start_idx ? min(max(start_idx[b], 0), N - 1) : 0;   // a caller's start index is clamped into the cloud, never trusted
    unsigned r = rank_of((unsigned)old0);
    if (tid == 0) s_out[0] = r;
    const int lane = tid & 63, wid = tid >> 6;
    float4 c = entry(r);

#if PCL_EXP == 6
    long long ph[5] = {0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
    const long long t_chain = tl;
#endif
    for (int step = 1; step < m; ++step) {
        FPS_STAMP(0, c.x)
        double kq[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float d = sq_dist3(px[j], py[j], pz[j], c.x, c.y, c.z);
            const float d2 = fminf(d, md[j]);                     // never-candidates stay at 0 forever
            md[j] = d2;
            kq[j] = __longlong_as_double(((unsigned long long)__float_as_uint(d2) << 32) | lo[j]);
        }
        // the lane's maximum as a TREE over its PPT keys (depth log2 PPT; a running maximum is a dependent chain of PPT v_max_f64)
#pragma unroll
        for (int w = 1; w < PPT; w <<= 1)
#pragma unroll
            for (int j = 0; j + w < PPT; j += 2 * w) kq[j] = key_max(kq[j], kq[j + w]);
        double best = kq[0];
        FPS_STAMP(1, best)
        double v = wave_key_max_lane63(best);                     // lane 63: the wave's maximum
        unsigned wlo;
        if (NW > 1) {
            FPS_STAMP(2, v)
            unsigned long long* sl = slots + (step & 1) * NW;
            if (lane == 63) sl[wid] = __double_as_longlong(v);
            __syncthreads();
            double wd = __longlong_as_double(sl[0]);
#pragma unroll
            for (int i = 1; i < NW; ++i) wd = key_max(wd, __longlong_as_double(sl[i]));
            wlo = (unsigned)__double_as_longlong(wd);
        } else {
            wlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)__double_as_longlong(v), 63);
            FPS_STAMP(2, wlo)
        }
        FPS_STAMP(3, wlo)
        // the winner's rank IS the next centre's slot: low word 0 can only win when no point is live -> ~0 clamps to RT = point 0
        r = min(~wlo, RT);
        c = entry(r);
        if (tid == 0) s_out[step] = r;
        FPS_STAMP(4, r)
    }
#if PCL_EXP == 6
    if (b == 0 && tid == 0) {
        for (int i = 0; i < 5; ++i) g_fps_ph[i] = ph[i];
        g_fps_ph[5] = m - 1; g_fps_ph[6] = __builtin_readcyclecounter() - t_chain; g_fps_ph[7] = T * 1000 + PPT;
    }
#endif
    __syncthreads();
    for (int j = tid; j < m; j += T) out[j] = (int32_t)index_of(s_out[j]);
    if (oxyz)
        for (int i = tid; i < 3 * m; i += T) {
            const int j = i / 3, cc = i - 3 * j;
            oxyz[i] = s_flat[ES * s_out[j] + cc];
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
    int old = start_idx ? min(max(start_idx[b], 0), N - 1) : 0;      // a caller's start index is clamped into the cloud, never trusted
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

static int g_fps_threads = 0, g_fps_prio = 3;       // pcl_set_fps_tuning (process-wide, set between calls)

// LDS of the register kernel: 256 B of slots + (RT + 1) entries of 16 | 12 bytes (RT = the power of two the tie ranks live under, N <= RT < 2 N) + m ranks
static size_t fps_lds_bytes(int N, int m, int log2S, int entry_floats) {
    int sh = 0;
    while ((((unsigned)N - 1) >> log2S) >> sh) ++sh;
    const size_t RT = (size_t)1 << (log2S + sh);
    return 256 + (((size_t)entry_floats * 4 * (RT + 1) + 15) & ~(size_t)15) + sizeof(int) * (size_t)m;
}

template <int T, int PPT>
static int launch_fps(const float* xyz, int B, int N, int m, int log2S, double thr, const int32_t* start,
                      int32_t* idx, float* nx, hipStream_t st) {
    const size_t lds = fps_lds_bytes(N, m, log2S, T * PPT > 1024 ? 3 : 4);           // slots + rank-indexed xyz + sampled ranks
    auto kern = fps_kernel<T, PPT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(PCL_EHIP, "fps: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(T), lds, st, xyz, N, m, log2S, thr, start, idx, nx, g_fps_prio);
    return check_launch("pcl_fps_f32");
}

}  // namespace pcl

using namespace pcl;

#if PCL_EXP == 6
// lab build only (not declared in include/pcl_hip.h): the phase cycles of the last launch, after a device synchronisation
extern "C" int pcl_lab_fps_read(long long* dst) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fps_ph), sizeof(g_fps_ph), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

extern "C" void pcl_set_fps_tuning(int threads_per_cloud, int issue_priority) {
    g_fps_threads = threads_per_cloud > 0 ? threads_per_cloud : 0;
    g_fps_prio = issue_priority < 0 ? 3 : issue_priority > 3 ? 3 : issue_priority;
}

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
    // threads per cloud: ~4-8 points per lane keeps the VALU part and the cross-wave part balanced
    // (raced on MI355X, B = 32 / 16, round 4 -- ns per dependent step at T = 64 / 128 / 256 / 512 / 1024:
    //  N = 512: 336 / 355 / 374 / - / -;  N = 1024: 479 / 410 / 381 / 416 / 736;  N = 2048: - / 550 / 450 / 475 / 766;  N = 4096: - / - / 589 / 611 / 869.
    //  One wave per cloud -- no barrier, no LDS exchange, DPP only -- wins up to 512 points; beyond, four waves: the cross-wave
    //  exchange costs less than the eight more distance evaluations per lane)
    int T = g_fps_threads > 0 ? g_fps_threads : (N <= 512 ? 64 : N <= 4096 ? 256 : N <= 8192 ? 512 : 1024);
    const int ppt = (N + T - 1) / T;
#define PCL_FPS_CASE(TT, PP) \
    if (T == TT && ppt <= PP) return launch_fps<TT, PP>(xyz, B, N, m, log2S, skip_sqnorm_le, start_idx, idx_out, new_xyz_out, st);
    if (fps_lds_bytes(N, m, log2S, N > 1024 ? 3 : 4) <= 158 * 1024) {
        PCL_FPS_CASE(64, 1) PCL_FPS_CASE(64, 2) PCL_FPS_CASE(64, 4) PCL_FPS_CASE(64, 8) PCL_FPS_CASE(64, 16) PCL_FPS_CASE(64, 32)
        PCL_FPS_CASE(128, 1) PCL_FPS_CASE(128, 2) PCL_FPS_CASE(128, 4) PCL_FPS_CASE(128, 8) PCL_FPS_CASE(128, 16)
        PCL_FPS_CASE(256, 1) PCL_FPS_CASE(256, 2) PCL_FPS_CASE(256, 4) PCL_FPS_CASE(256, 8) PCL_FPS_CASE(256, 16)
        PCL_FPS_CASE(512, 1) PCL_FPS_CASE(512, 2) PCL_FPS_CASE(512, 4) PCL_FPS_CASE(512, 8) PCL_FPS_CASE(512, 16)
        PCL_FPS_CASE(1024, 1) PCL_FPS_CASE(1024, 2) PCL_FPS_CASE(1024, 4) PCL_FPS_CASE(1024, 8)
        // (1024 threads x 16 points per lane does not fit the 128 registers of a 1024-thread workgroup -- it spilled 16 of them into the step
        //  loop; clouds of 8 193 .. 10 100 points take the LDS-resident kernel below like the larger ones)
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
