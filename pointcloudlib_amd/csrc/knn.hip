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
#pragma clang fp contract(off)      // index-producing arithmetic: every operation rounds on its own

namespace pcl {

constexpr int KD_T = 256, KD_TILE = 64, KD_CK = 32;

// FMA = the library's NAMED SECOND DEFINITION of the distance (pcl_knn_fma_f32): `ssd = fma(tmp, tmp, ssd)`, what nvcc's default
// -fmad=true makes of `ssd += tmp*tmp` (misc/ops.py:490) -- 2 VALU operations per (query, reference, channel) instead of 3.
// The default (FMA = false) rounds the product and the sum separately, as the source text reads.
template <bool FMA>
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
                    if constexpr (FMA) acc[i][j] = __fmaf_rn(t, t, acc[i][j]);
                    else acc[i][j] = __fadd_rn(acc[i][j], __fmul_rn(t, t));    // ssd += tmp*tmp, :490
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

// ---- fused distance + select: no [Nq x Nr] matrix ---------------------------------------------------------------------
// One wave owns QW queries of a cloud and ALL Nr reference points: lane l holds, per query, the PPT = Nr/64 running sums of
// the references r = 256*J + 4*l + e (e < 4: one ds_read_b128 of the staged channel row per J), i.e. QW x PPT accumulators
// in VGPRs (128 at every supported size).  The workgroup (4 waves, 4*QW queries) streams the cloud's reference matrix
// [C][Nr] once through a double-buffered LDS stage in chunks of CK channels, queries ride along as a [CK][4*QW] tile read
// back as broadcasts.  Arithmetic per (query, ref, channel) is the reference's: t = ref - qry; ssd = ssd + t*t, each
// operation rounded separately, channels ascending (misc/ops.py:488-491) -- VALU, not MFMA, because a^2+b^2-2ab rounds
// differently and reorders near-ties.  Then each query's row is in registers already and the k winners come out by k rounds
// of {lane-local min, DPP wave min, ballot}, ascending by (distance, r) (misc/ops.py:528-550).
// Traffic: the reference matrix is read Nq/(4*QW) times per cloud from L2 (C*Nr*4 bytes each), nothing else but the
// queries and the k*Nq indices: no O(Nq*Nr) workspace.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4k __attribute__((ext_vector_type(4)));

// Range-checked loads through a buffer descriptor over one cloud's [C][N] matrix: a channel past C is past the end and
// reads 0 with no branch (a load under a divergent branch makes hipcc drain the whole memory queue at the join).  Points
// past N within a row read the next row's values -- finite, and every consumer masks those columns by index.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t knn_rsrc(const float* base, size_t bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned n = bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
__device__ __forceinline__ float4 knn_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4k v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float knn_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// Workgroup -> (cloud, query block), XCD-aware: hardware hands consecutive workgroup ids to the 8 XCDs round-robin, each with
// its own 4 MB L2.  With the plain (x = query block, y = cloud) grid the 32 workgroups of a cloud land on all 8 XCDs and every
// L2 sees every cloud's reference matrix (16 MB at C = 128: misses to the Infinity Cache on each of the Nq/32 passes).  Here
// XCD j takes the clouds j, j+8, ...: one matrix (<= 2 MB) is streamed by workgroups that share an L2 and run back to back.
__device__ __forceinline__ void knn_block(int gx, int B, int& b, int& qb) {
    const int lin = blockIdx.y * gx + blockIdx.x;
    if ((B & 7) == 0) { const int slot = lin >> 3; b = (lin & 7) + 8 * (slot / gx); qb = slot % gx; }
    else { b = blockIdx.y; qb = blockIdx.x; }
}

template <int PPT>
struct KnnCfg {
    static constexpr int QW = 128 / PPT > 8 ? 8 : 128 / PPT;     // queries per wave
    static constexpr int NRP = PPT * 64;                         // padded reference count
    static constexpr int CK = 8192 / NRP > 32 ? 32 : 8192 / NRP; // channels per LDS stage (<= 32 KB of references)
};

template <int PPT, bool FMA = false>
__global__ __launch_bounds__(256) void knn_fused_kernel(const float* __restrict__ ref, const float* __restrict__ qry, int C, int Nr,
                                                        int Nq, int k, int32_t* __restrict__ idx_out, int out_nk) {
    int b, qblk;
    knn_block(gridDim.x, gridDim.y, b, qblk);
    using Cfg = KnnCfg<PPT>;
    constexpr int QW = Cfg::QW, NRP = Cfg::NRP, CK = Cfg::CK, QB = 4 * QW, NJ = PPT / 4;
    constexpr int RV = CK * NRP / 4 / 256;                       // 16-byte reference pieces per thread and stage
    constexpr int QV = (CK * QB + 255) / 256;                   // query values per thread and stage
    static_assert(RV >= 1, "staging maps");
    __shared__ __attribute__((aligned(16))) float sR[2][CK][NRP];
    __shared__ __attribute__((aligned(16))) float sQ[2][CK][QB];
    __shared__ uint2 sCand[4][64];                               // per wave: the candidates of the query being selected
    const int q0 = qblk * QB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* R = ref + (size_t)b * C * Nr;
    const float* Q = qry + (size_t)b * C * Nq;

    // two references per register pair: v_pk_add_f32 / v_pk_mul_f32 are separately rounded like their scalar forms and run at
    // twice the rate (contraction into an FMA is switched off for this file, whatever the command line says)
    f32x2 acc[QW][PPT / 2];
#pragma unroll
    for (int i = 0; i < QW; ++i)
#pragma unroll
        for (int j = 0; j < PPT / 2; ++j) acc[i][j] = f32x2{0.f, 0.f};

    // stage registers: piece e = tid + 256*v -> channel e / (NRP/4), references 4*(e % (NRP/4)) ..+3; one query value
    float4 pr[RV];
    float pq[QV];
    const __amdgpu_buffer_rsrc_t rsR = knn_rsrc(R, (size_t)C * Nr * 4), rsQ = knn_rsrc(Q, (size_t)C * Nq * 4);
    auto fetch = [&](int c0) {
#pragma unroll
        for (int v = 0; v < RV; ++v) {
            const int e = tid + 256 * v, cc = e / (NRP / 4), r = 4 * (e % (NRP / 4));
            pr[v] = knn_ld4(rsR, (unsigned)(((size_t)(c0 + cc) * Nr + r) * 4));
        }
#pragma unroll
        for (int v = 0; v < QV; ++v) {
            const int e = tid + 256 * v, cc = e / QB, qi = e % QB;
            pq[v] = knn_ld1(rsQ, (unsigned)(((size_t)(c0 + cc) * Nq + q0 + qi) * 4));
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int v = 0; v < RV; ++v) {
            const int e = tid + 256 * v;
            *reinterpret_cast<float4*>(&sR[buf][e / (NRP / 4)][4 * (e % (NRP / 4))]) = pr[v];
        }
#pragma unroll
        for (int v = 0; v < QV; ++v) {
            const int e = tid + 256 * v;
            if (e < CK * QB) sQ[buf][e / QB][e % QB] = pq[v];
        }
    };

    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < C; c0 += CK, buf ^= 1) {
        const bool more = c0 + CK < C;
        if (more) fetch(c0 + CK);
        // channels in pairs with two operand sets: the LDS reads of channel cc+1 are issued before the arithmetic of channel cc
        // (2 waves per SIMD do not hide an LDS round trip per 4 reference registers).  A channel past C was staged as zeros on
        // both sides: (0 - 0)^2 added to a sum >= +0 leaves it bit-identical, so an odd tail just runs one idle channel.
        const int cend = ((C - c0 < CK ? C - c0 : CK) + 1) & ~1;
        struct Ops { float qv[QW]; float4 rv[NJ]; };
        auto ld = [&](int cc) -> Ops {
            Ops o;
#pragma unroll
            for (int i = 0; i < QW; i += 4) {
                const float4 t = *reinterpret_cast<const float4*>(&sQ[buf][cc][wave * QW + i]);     // broadcast read
                o.qv[i] = t.x;
                if (i + 1 < QW) o.qv[i + 1] = t.y;
                if (i + 2 < QW) o.qv[i + 2] = t.z;
                if (i + 3 < QW) o.qv[i + 3] = t.w;
            }
#pragma unroll
            for (int J = 0; J < NJ; ++J) o.rv[J] = *reinterpret_cast<const float4*>(&sR[buf][cc][256 * J + 4 * lane]);
            return o;
        };
        auto compute = [&](const Ops& o) {
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
                const f32x2 r01 = {o.rv[J].x, o.rv[J].y}, r23 = {o.rv[J].z, o.rv[J].w};
                // (three passes over the 2*QW pairs, not three dependent instructions per pair: back-to-back dependent packed
                // ops cost wait states -- hipcc filled the fused form with 45 s_nop per channel)
                f32x2 t[QW][2];
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    const f32x2 qq = {o.qv[i], o.qv[i]};
                    t[i][0] = r01 - qq; t[i][1] = r23 - qq;                                       // ref - query, :489
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (FMA) {                                                              // ssd = fma(tmp, tmp, ssd): v_pk_fma_f32
#pragma unroll
                    for (int i = 0; i < QW; ++i) {
                        acc[i][2 * J] = __builtin_elementwise_fma(t[i][0], t[i][0], acc[i][2 * J]);
                        acc[i][2 * J + 1] = __builtin_elementwise_fma(t[i][1], t[i][1], acc[i][2 * J + 1]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < QW; ++i) { t[i][0] = t[i][0] * t[i][0]; t[i][1] = t[i][1] * t[i][1]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < QW; ++i) {                                                // ssd += tmp*tmp, :490
                        acc[i][2 * J] = acc[i][2 * J] + t[i][0];
                        acc[i][2 * J + 1] = acc[i][2 * J + 1] + t[i][1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        Ops A = ld(0);
        for (int cc = 0; cc < cend; cc += 2) {
            const Ops B = ld(cc + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(A);
            A = ld(cc + 2 < CK ? cc + 2 : 0);                   // (the wrap-around read of the last pair is unused)
            __builtin_amdgcn_sched_barrier(0);
            compute(B);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }

    // ---- select: per query the k smallest (distance bits, r), ascending; sums of squares are >= +0 -> bit-monotone
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        const int q = q0 + wave * QW + i;
        if (q >= Nq) break;                                        // (wave-uniform)
        unsigned key[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int r = 256 * (j >> 2) + 4 * lane + (j & 3);
            key[j] = r < Nr ? __float_as_uint(acc[i][j >> 1][j & 1]) : 0xFFFFFFFFu;
        }
        // out_nk: the lists as [B, Nq, k] rows (what the EdgeConv gathers read) instead of the reference's [B, k, Nq] (round 6: the permute
        // copy after every search is gone)
        int32_t* out = out_nk ? idx_out + ((size_t)b * Nq + q) * k : idx_out + (size_t)b * k * Nq + q;
        const size_t ostr = out_nk ? 1 : (size_t)Nq;
        // Fast path (k <= 64): T = the k-th smallest of the 64 lane-local minima is an upper bound of the k-th smallest key
        // (those are 64 distinct elements), found by a bitwise search with one ballot per bit; the keys <= T (>= k of
        // them, typically ~1.5 k) are compacted into LDS and ranked by (key, r) -- ~400 issue slots instead of ~100 per
        // extraction round.  More than 64 candidates (mass ties) or k > 64: the rounds below, same result.
        bool done = false;
        if (k <= 64) {
            unsigned m = key[0];
#pragma unroll
            for (int j = 1; j < PPT; ++j) m = min(m, key[j]);
            unsigned T = 0;
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned trial = T | (1u << bit);
                if (__popcll(__ballot(m < trial)) < k) T = trial;            // (uniform)
            }
            int n = 0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) n += __popcll(__ballot(key[j] <= T));
            if (n <= 64) {
                uint2* cand = sCand[wave];
                int base = 0;
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    const bool pred = key[j] <= T;
                    const unsigned long long mask = __ballot(pred);
                    if (mask) {                                              // (uniform)
                        if (pred) cand[base + mbcnt(mask)] = make_uint2(key[j], 256u * (j >> 2) + 4u * lane + (j & 3u));
                        base += __popcll(mask);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const uint2 mine = cand[lane < n ? lane : 0];
                int rank = 0;
                for (int c0 = 0; c0 < n; c0 += 8) {                          // eight broadcast reads in flight: one LDS latency per
                    uint2 o[8];                                              // eight candidates, not per candidate
#pragma unroll
                    for (int u = 0; u < 8; ++u) o[u] = cand[(c0 + u) & 63];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        rank += (c0 + u < n && (o[u].x < mine.x || (o[u].x == mine.x && o[u].y < mine.y))) ? 1 : 0;
                }
                if (lane < n && rank < k) out[(size_t)rank * ostr] = (int32_t)mine.y;
                __builtin_amdgcn_wave_barrier();                             // (the next query reuses the list)
                done = true;
            }
        }
        for (int t = 0; t < k && !done; ++t) {
            unsigned bk = 0xFFFFFFFFu, bj = 0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {                        // increasing j = increasing r within the lane
                const bool take = key[j] < bk;                     // strict: lowest r kept on ties
                bk = take ? key[j] : bk;
                bj = take ? (unsigned)j : bj;
            }
            const unsigned br = 256u * (bj >> 2) + 4u * lane + (bj & 3u);
            const unsigned wmin = wave_min_u32(bk);
            const unsigned long long tied = __ballot(bk == wmin);
            unsigned r;
            if (__popcll(tied) == 1) {
                const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)tied) - 1);
                r = __builtin_amdgcn_readlane(br, l);
            } else {
                r = wave_min_u32(bk == wmin ? br : 0xFFFFFFFFu);
            }
            if (lane == 0) out[(size_t)t * ostr] = (int32_t)r;
            const unsigned rj = 4u * (r >> 8) + (r & 3u), rl = (r & 255u) >> 2;
            const bool mine = (unsigned)lane == rl;
#pragma unroll
            for (int j = 0; j < PPT; ++j)
                if ((unsigned)j == rj && mine) key[j] = 0xFFFFFFFFu;
        }
    }
}


// ---- PointConv's knn_point in the reference's OWN arithmetic (named second definition, round 4) --------------------------------
// /root/reference/misc/pointconv_utils.py:34-53 + :120-131: dist = -2 matmul(src, dst^T) + sum(src^2) + sum(dst^2), the first k of a
// full ascending argsort.  The library's definition of these groups is the direct form (pcl_knn_f32 on the coordinates); this
// kernel evaluates the matmul form exactly as oracle/pcl_oracle.c::pclo_knn_point_matmul_f32 defines it -- dot over c = 0, 1, 2
// ascending (fma chain, or every product and sum rounded), squares summed ascending, the three terms added in source order,
// stable order (distance, index) -- so the two agree bit for bit and a network can be run on the reference's own groups.
// One wave per query row at a time (QW queries per wave), a lane owns PPT references (their coordinates and |p|^2 stay in
// registers for all of the wave's queries); distances can be negative, so the select runs on order-preserving unsigned keys.
template <int PPT>
__device__ __forceinline__ void knn_select_keys(unsigned (&key)[PPT], int k, int32_t* out, uint2* cand, int lane) {
    bool done = false;
    if (k <= 64) {          // threshold T = the k-th smallest of the 64 lane minima, candidates <= T ranked by (key, r): see knn_fused_kernel
        unsigned m = key[0];
#pragma unroll
        for (int j = 1; j < PPT; ++j) m = min(m, key[j]);
        unsigned T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned trial = T | (1u << bit);
            if (__popcll(__ballot(m < trial)) < k) T = trial;
        }
        int n = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) n += __popcll(__ballot(key[j] <= T));
        if (n <= 64) {
            int base = 0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const bool pred = key[j] <= T;
                const unsigned long long mask = __ballot(pred);
                if (mask) {
                    if (pred) cand[base + mbcnt(mask)] = make_uint2(key[j], 256u * (j >> 2) + 4u * lane + (j & 3u));
                    base += __popcll(mask);
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const uint2 mine = cand[lane < n ? lane : 0];
            int rank = 0;
            for (int c = 0; c < n; ++c) {
                const uint2 o = cand[c];
                rank += (o.x < mine.x || (o.x == mine.x && o.y < mine.y)) ? 1 : 0;
            }
            if (lane < n && rank < k) out[rank] = (int32_t)mine.y;
            __builtin_amdgcn_wave_barrier();
            done = true;
        }
    }
    for (int t = 0; t < k && !done; ++t) {
        unsigned bk = 0xFFFFFFFFu, bj = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const bool take = key[j] < bk;
            bk = take ? key[j] : bk;
            bj = take ? (unsigned)j : bj;
        }
        const unsigned br = 256u * (bj >> 2) + 4u * lane + (bj & 3u);
        const unsigned wmin = wave_min_u32(bk);
        const unsigned r = wave_min_u32(bk == wmin ? br : 0xFFFFFFFFu);
        if (lane == 0) out[t] = (int32_t)r;
        const unsigned rj = 4u * (r >> 8) + (r & 3u), rl = (r & 255u) >> 2;
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if ((unsigned)j == rj && (unsigned)lane == rl) key[j] = 0xFFFFFFFFu;
    }
}

template <int PPT>
__global__ __launch_bounds__(256) void knn_point_matmul_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S,
                                                               int k, int fma_dot, int32_t* __restrict__ idx_out) {
    constexpr int QW = 8, NJ = PPT / 4;
    __shared__ uint2 sCand[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* P = xyz + (size_t)b * N * 3;
    const float* Q = new_xyz + (size_t)b * S * 3;
    const __amdgpu_buffer_rsrc_t rsP = knn_rsrc(P, (size_t)N * 12);
    float px[PPT], py[PPT], pz[PPT], dn[PPT];
#pragma unroll
    for (int J = 0; J < NJ; ++J) {                   // references 256 J + 4 lane .. + 3: twelve consecutive floats
        const unsigned off = (unsigned)(256 * J + 4 * lane) * 12u;
        const float4 a = knn_ld4(rsP, off), c = knn_ld4(rsP, off + 16u), e = knn_ld4(rsP, off + 32u);
        px[4 * J] = a.x; py[4 * J] = a.y; pz[4 * J] = a.z;
        px[4 * J + 1] = a.w; py[4 * J + 1] = c.x; pz[4 * J + 1] = c.y;
        px[4 * J + 2] = c.z; py[4 * J + 2] = c.w; pz[4 * J + 2] = e.x;
        px[4 * J + 3] = e.y; py[4 * J + 3] = e.z; pz[4 * J + 3] = e.w;
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j)                    // sum(dst**2): squares summed ascending, each operation rounded (:52)
        dn[j] = __fadd_rn(__fadd_rn(__fmul_rn(px[j], px[j]), __fmul_rn(py[j], py[j])), __fmul_rn(pz[j], pz[j]));
    for (int i = 0; i < QW; ++i) {
        const int q = (blockIdx.x * 4 + wave) * QW + i;
        if (q >= S) break;                           // (wave-uniform)
        const float qx = Q[3 * q], qy = Q[3 * q + 1], qz = Q[3 * q + 2];
        const float sn = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));          // :51
        unsigned key[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int r = 256 * (j >> 2) + 4 * lane + (j & 3);
            float dot;
            if (fma_dot) dot = __fmaf_rn(qz, pz[j], __fmaf_rn(qy, py[j], __fmul_rn(qx, px[j])));
            else dot = __fadd_rn(__fadd_rn(__fmul_rn(qx, px[j]), __fmul_rn(qy, py[j])), __fmul_rn(qz, pz[j]));
            float d = __fmul_rn(-2.0f, dot);                                                                        // :50
            d = __fadd_rn(d, sn);
            d = __fadd_rn(d, dn[j]);
            d = __fadd_rn(d, 0.0f);                  // -0 -> +0: the comparison sort treats them as equal (ties go by index)
            unsigned u = __float_as_uint(d);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            key[j] = r < N ? u : 0xFFFFFFFFu;
        }
        knn_select_keys<PPT>(key, k, idx_out + ((size_t)b * S + q) * k, sCand[wave], lane);
    }
}

}  // namespace pcl
using namespace pcl;

// up to 4096 references the fused kernel keeps a query's whole distance row in registers (no workspace); beyond: two passes
constexpr int KNN_FUSED_MAX_NR = 4096;
// (An MFMA-filter variant -- approximate distances on the matrix cores choosing <= 64 candidates per query for the exact
// arithmetic to rank -- was built in round 2, exact but not faster (C = 64: 220 against 200 us), and removed in round 3;
// DESIGN.md section 3.4 keeps the measurements, the code is in the history.)
extern "C" size_t pcl_knn_workspace_bytes(int B, int C, int Nr, int Nq, int k) {
    if (B <= 0 || Nr <= 0 || Nq <= 0) return 0;
    if (Nr <= KNN_FUSED_MAX_NR) return 0;
    return sizeof(float) * (size_t)B * Nr * Nq;
}

template <bool FMA>
static int knn_impl(const float* ref, const float* qry, int B, int C, int Nr, int Nq, int k,
                    int32_t* idx_out, void* workspace, size_t workspace_bytes, void* stream, int out_nk = 0) {
    PCL_REQUIRE(ref && qry && idx_out, "pcl_knn_f32: null pointer");
    PCL_REQUIRE(B >= 0 && C >= 1 && Nr >= 1 && Nq >= 1, "pcl_knn_f32: bad sizes B=%d C=%d Nr=%d Nq=%d", B, C, Nr, Nq);
    PCL_REQUIRE(k >= 1 && k <= Nr, "pcl_knn_f32: need 1 <= k <= Nr (k=%d Nr=%d)", k, Nr);
    PCL_REQUIRE(B <= 65535, "pcl_knn_f32: B=%d exceeds grid limit", B);
    if (B == 0) return PCL_OK;
    if (Nr <= KNN_FUSED_MAX_NR) {
        hipStream_t st = as_stream(stream);
        const int ppt4 = (Nr + 255) / 256 * 4;                    // multiples of 4 registers per lane
#define PCL_KF(P) if (ppt4 <= P) { PCL_LAUNCH_TIMED((knn_fused_kernel<P, FMA>), dim3((Nq + 4 * KnnCfg<P>::QW - 1) / (4 * KnnCfg<P>::QW), B), dim3(256), st, ref, qry, C, Nr, Nq, k, idx_out, out_nk); return check_launch("pcl_knn_f32(fused)"); }
        PCL_KF(4) PCL_KF(8) PCL_KF(16) PCL_KF(32) PCL_KF(64)
#undef PCL_KF
    }
    PCL_REQUIRE(!out_nk, "pcl_knn_nk_f32: the [B, Nq, k] layout comes from the fused kernel only (Nr <= %d)", KNN_FUSED_MAX_NR);
    const size_t need = Nr <= KNN_FUSED_MAX_NR ? 0 : sizeof(float) * (size_t)B * Nr * Nq;
    if (need && (!workspace || workspace_bytes < need))
        return fail(PCL_EWS, "pcl_knn_f32: workspace %zu bytes < required %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    float* dist = static_cast<float*>(workspace);
    dim3 g0((Nr + KD_TILE - 1) / KD_TILE, (Nq + KD_TILE - 1) / KD_TILE, B);
    hipLaunchKernelGGL(knn_dist_kernel<FMA>, g0, dim3(KD_T), 0, st, ref, qry, C, Nr, Nq, dist);
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

extern "C" int pcl_knn_nk_supported(int Nr) { return Nr >= 1 && Nr <= KNN_FUSED_MAX_NR; }
extern "C" int pcl_knn_nk_f32(const float* ref, const float* qry, int B, int C, int Nr, int Nq, int k, int32_t* idx_out, void* stream) {
    return knn_impl<false>(ref, qry, B, C, Nr, Nq, k, idx_out, nullptr, 0, stream, 1);
}
extern "C" int pcl_knn_f32(const float* ref, const float* qry, int B, int C, int Nr, int Nq, int k,
                           int32_t* idx_out, void* workspace, size_t workspace_bytes, void* stream) {
    return knn_impl<false>(ref, qry, B, C, Nr, Nq, k, idx_out, workspace, workspace_bytes, stream);
}

extern "C" int pcl_knn_fma_f32(const float* ref, const float* qry, int B, int C, int Nr, int Nq, int k,
                               int32_t* idx_out, void* workspace, size_t workspace_bytes, void* stream) {
    return knn_impl<true>(ref, qry, B, C, Nr, Nq, k, idx_out, workspace, workspace_bytes, stream);
}

// reference: misc/pointconv_utils.py:120-131 (knn_point) in the arithmetic of :34-53 (square_distance); see knn_point_matmul_kernel
extern "C" int pcl_knn_point_matmul_f32(const float* xyz, const float* new_xyz, int B, int N, int S, int k, int fma_dot, int32_t* idx_out,
                                        void* stream) {
    PCL_REQUIRE(xyz && new_xyz && idx_out, "pcl_knn_point_matmul_f32: null pointer");
    PCL_REQUIRE(B >= 0 && N >= 1 && S >= 1 && k >= 1 && k <= N, "pcl_knn_point_matmul_f32: bad sizes B=%d N=%d S=%d k=%d", B, N, S, k);
    PCL_REQUIRE(B <= 65535 && N <= KNN_FUSED_MAX_NR, "pcl_knn_point_matmul_f32: B=%d (<= 65535), N=%d (<= %d: the references of a cloud live in one wave's registers)", B, N, KNN_FUSED_MAX_NR);
    PCL_REQUIRE((reinterpret_cast<uintptr_t>(xyz) & 15) == 0 && ((size_t)N * 12) % 16 == 0, "pcl_knn_point_matmul_f32: xyz must be 16-byte aligned and N a multiple of 4");
    if (B == 0) return PCL_OK;
    hipStream_t st = as_stream(stream);
    const int ppt4 = (N + 255) / 256 * 4;
    const dim3 grid((S + 31) / 32, B), blk(256);
#define PCL_KPM(P) if (ppt4 <= P) { hipLaunchKernelGGL(knn_point_matmul_kernel<P>, grid, blk, 0, st, xyz, new_xyz, N, S, k, fma_dot, idx_out); return check_launch("pcl_knn_point_matmul_f32"); }
    PCL_KPM(4) PCL_KPM(8) PCL_KPM(16) PCL_KPM(32) PCL_KPM(64)
#undef PCL_KPM
    return fail(PCL_ENOSUP, "pcl_knn_point_matmul_f32: N=%d", N);
}
