// frag.hip -- GEMMs whose MFMA fragments come straight from global memory / L2, for layers with FEW ROWS (round 5).
//
// Where: the GroupAll level of PointNet++ (networks/cls/pointnet2.py:131-136: [259 -> 256 -> 512 -> 1024] on B * 128 = 4 096 rows), the
// feature-propagation stacks of the part-seg decoder (networks/seg/pointnet2_partseg.py:146-148: 1280 / 1664 -> 256 -> 256 on B * 128 rows,
// 384 / 576 -> 256 -> 128 on B * 512), the per-point products of the folded first layers and DGCNN's U | V product (networks/cls/dgcnn.py:72-83).
// The staged kernels of mlp.hip are built for 10^5 .. 10^6 rows: persistent 128-row tiles, two block barriers per 32-wide k step, split-K
// partial tiles by the megabyte.  With 2 048 .. 8 192 rows there are fewer output tiles than SIMDs and one workgroup per CU, so every
// barrier and every global -> LDS -> register hop is exposed (r4: fwd259x256 36 TF, dx256x259 22 TF, dw256x259 24 TF of 157).
//
// Here a WAVE owns a 32 x (32 TN) output tile and reads its operands as MFMA fragments directly: lane (lr = l & 31, lh = l >> 5) of
// v_mfma_f32_32x32x2_f32 holds A[row lr][k = lh] and B[col lr][k = lh]; over a step of 8 k the lane loads the 16 bytes A[lr][8s + 4lh ..+3]
// (one buffer_load_dwordx4) and uses element j in MFMA j -- the contraction index is permuted the same way on both operands.  No LDS
// staging, no barrier in the k loop, 4-16 waves per CU hide each other's L2 latency; the operands of these layers (<= 16 MB) live in
// L2 / Infinity Cache.  A workgroup is 2 x 2 such waves (a 64 x 64 TN block: the row / column fragments are shared through L1) times KSW
// wave groups that split K and are reduced through LDS in fixed order at the end (deterministic), so that even a 4 096 x 256 output
// (1 024 wave tiles) puts two waves on every SIMD.
//
// Accuracy option (FLUSH = 8 or 32): every FLUSH contraction steps the fp32 accumulators are added into fp64 ones and cleared, so an
// fp32 fma chain never runs longer than FLUSH terms from zero: the error of a K-term dot product drops by ~sqrt(FLUSH / K) (K = 1664,
// FLUSH = 32: 7 x) for 16 (cvt + add) per 16 MFMAs of a tile.  Used where the part-seg decoder's and DGCNN's distance from the fp64
// evaluation was accumulation error (tools/dbg/partseg_local_err.py).
//
// The same kernel is the dX GEMM (B consumed K-major: W[k][n] as stored; epilogue = ReLU mask of the layer below + its BatchNorm-backward
// sums) on a dy tensor that frag_dy_kernel forms ONCE per layer (dense, from dU | (arg, gz) and Y) instead of once per reading wave, and
// frag_dw_kernel is the weight gradient dW = dy^T z with the rows split over the waves of a workgroup and over KSG workgroups whose
// partial tiles the LAST ARRIVER of a tile sums in fixed order (agent-scope release / acquire around one counter per tile).
#include "common.h"
#include <type_traits>

namespace pcl {
namespace fg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned SOFF_DEAD = 0x80000000u;        // a scalar offset beyond every descriptor here (operands < 2 GiB): reads 0

__device__ __forceinline__ rsrc_t rsrc(const void* base, size_t first_byte, size_t total_bytes) {
    const size_t left = total_bytes > first_byte ? total_bytes - first_byte : 0;
    const unsigned n = left > 0x7fffffffull ? 0x7fffffffu : (unsigned)left;
    const uintptr_t a = reinterpret_cast<uintptr_t>(base) + first_byte;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
__device__ __forceinline__ float4 ld4(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float ld1(rsrc_t r, unsigned voff, unsigned soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ void st1(rsrc_t r, unsigned voff, unsigned soff, float x) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0); }
__device__ __forceinline__ float lrelu_max(float t, float slope) { return fmaxf(t, t * slope); }      // slope in [0, 1]

// XCD-aware block -> (row tile, column tile): ids 8 apart (same XCD, dispatched back to back) share the row tile, so the A rows a
// column sweep re-reads are hits of that XCD's L2 (mlp.hip: tile_of_block, measured there)
__device__ __forceinline__ void tile_of(int id, int nrt, int nct, int& rt, int& ct) {
    const int full = nrt & ~7;
    if (id < full * nct) {
        const int j = id >> 3;
        ct = j % nct;
        rt = (j / nct) * 8 + (id & 7);
    } else {
        const int r = id - full * nct, rem = nrt - full;
        rt = full + r % rem;
        ct = r / rem;
    }
}

enum { EP_STORE = 0, EP_STATS = 1, EP_MASK_STATS = 2 };

struct GemmArgs {
    const float* A; int lda;                  // [M][lda >= K]
    const float* B; int ldb;                  // BKN = false: W[n][ldb] (k contiguous); BKN = true: W[k][ldb] (n contiguous)
    const float* bias;                        // [N] or null
    const float* asc; const float* ash; float aslope;      // ACT: A element k becomes lrelu(asc[k] * x + ash[k])
    float* C; int ldc;                        // [M][ldc], columns n_begin .. N written
    float* Clo;                               // FLUSH only, optional: the fp64 sum's residual, C + Clo = the sum to ~2^-48 (same layout as C)
    double* stats;                            // EP_STATS / EP_MASK_STATS: [nrt][2][N] fp64 (sum c, sum c^2 | sum c * yprev), or null
    const float* Yp; int ldyp;                // EP_MASK_STATS: pre-BatchNorm output of the layer below [M][ldyp]
    const float* esc; const float* esh; float eslope;      //                 and its folded BatchNorm
    int M, N, K, n_begin;
    int nrt, nct;
};

constexpr int FG_D = 3;                       // k steps in flight per wave

// C[m][n] = sum_k act(A[m][k]) * B(n, k)  (+ bias[n]); one workgroup = 2 x 2 waves x KSW k-groups, wave tile 32 x (32 TN)
template <int TN, int FLUSH, bool ACT, bool BKN, int EP>
__global__ __launch_bounds__(1024) void frag_gemm_kernel(const GemmArgs p) {
    constexpr bool F64 = FLUSH > 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int KSW = blockDim.x >> 8;
    const int ks = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    int rt, ct;
    tile_of(blockIdx.x, p.nrt, p.nct, rt, ct);
    const int m0 = rt * 64, n0 = p.n_begin + ct * 64 * TN;
    // k steps of 8; the last one may be partial (K % 8 != 0: K = 259 at the GroupAll level, 132 / 324 / 644 on the grouped levels): it is
    // requested into its own registers with the elements past K forced to zero on the A side, so neither operand needs padding -- rows
    // may have any stride and any 4-byte alignment (buffer_load_dwordx4 takes it at full rate: tools/ubench/unaligned.hip)
    const int SF = p.K >> 3, rem = p.K - 8 * SF;                 // full steps, elements of the partial step
    const bool has_tail = rem > 0;
    const int S = SF + (has_tail ? 1 : 0);
    // this wave group's full steps [s0, s1); the partial step, if any, belongs to the last group
    const int per = (SF + KSW - 1) / KSW;
    const int s0 = min(SF, ks * per), s1 = min(SF, s0 + per);
    // LDS: [ACT: sc[KP], sh[KP]] [reduction buffer: 4 waves x TN x 16 x 64 values]
    const int KP = (S + FG_D + 1) * 8;
    float* const sSc = reinterpret_cast<float*>(lds_raw);
    float* const sSh = sSc + (ACT ? KP : 0);
    unsigned char* const sRed = reinterpret_cast<unsigned char*>(sSh + (ACT ? KP : 0));
    if constexpr (ACT) {
        for (int k = tid; k < KP; k += blockDim.x) { const bool in = k < p.K; sSc[k] = in ? p.asc[k] : 0.f; sSh[k] = in ? p.ash[k] : 0.f; }
        __syncthreads();
    }
    // descriptors: A rows of this row tile (rows past M read 0), B rows / columns of this column tile
    const unsigned lda_b = (unsigned)p.lda * 4, ldb_b = (unsigned)p.ldb * 4;
    const rsrc_t rA = rsrc(p.A, (size_t)m0 * lda_b, (size_t)p.M * lda_b);
    const rsrc_t rB = BKN ? rsrc(p.B, 0, (size_t)p.K * ldb_b) : rsrc(p.B, (size_t)n0 * ldb_b, (size_t)p.N * ldb_b);
    const unsigned vA = (unsigned)(wr * 32 + lr) * lda_b + (unsigned)lh * 16u;
    unsigned vB[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int cl = wc * 32 * TN + t * 32 + lr;               // column within the column tile
        if constexpr (BKN) {
            // W[k][n]: k = 8 s + 4 lh + j -> per-lane (4 lh) rows + column; j and s go into the scalar / immediate offsets.
            // a column past N must read 0: its per-lane offset is pushed out of range
            vB[t] = n0 + cl < p.N ? (unsigned)(4 * lh) * ldb_b + (unsigned)(n0 + cl) * 4u : 0xfffffff0u;
        } else vB[t] = (unsigned)cl * ldb_b + (unsigned)lh * 16u;
    }
    f32x16 acc[TN];
    double accd[F64 ? TN : 1][16];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; if constexpr (F64) accd[t][r] = 0.0; }

    float4 qa[FG_D], qb[FG_D][TN];
    auto issue = [&](int slot, int s) {                          // slot: compile-time after unrolling
        const bool live = s < s1;
        const unsigned so = live ? (unsigned)s * 32u : SOFF_DEAD;
        qa[slot] = ld4(rA, vA, so);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            if constexpr (BKN) {
                const unsigned sb = live ? (unsigned)s * 8u * ldb_b : SOFF_DEAD;
                qb[slot][t].x = ld1(rB, vB[t], sb); qb[slot][t].y = ld1(rB, vB[t], sb + ldb_b);
                qb[slot][t].z = ld1(rB, vB[t], sb + 2 * ldb_b); qb[slot][t].w = ld1(rB, vB[t], sb + 3 * ldb_b);
            } else qb[slot][t] = ld4(rB, vB[t], so);
        }
    };
    auto consume = [&](float4 a, const float4 (&b)[TN], int s) {
        if constexpr (ACT) {
            const float4 c = *reinterpret_cast<const float4*>(sSc + s * 8 + lh * 4), h = *reinterpret_cast<const float4*>(sSh + s * 8 + lh * 4);
            a.x = lrelu_max(fmaf(c.x, a.x, h.x), p.aslope); a.y = lrelu_max(fmaf(c.y, a.y, h.y), p.aslope);
            a.z = lrelu_max(fmaf(c.z, a.z, h.z), p.aslope); a.w = lrelu_max(fmaf(c.w, a.w, h.w), p.aslope);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
        }
    };
    auto flush = [&]() {
        if constexpr (F64) {
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { accd[t][r] += (double)acc[t][r]; acc[t][r] = 0.f; }
        }
    };
    // the partial step of the last k group: requested first (its own registers), consumed last
    float4 ta = make_float4(0, 0, 0, 0), tb[TN];
    const bool own_tail = has_tail && ks == KSW - 1;
#pragma unroll
    for (int t = 0; t < TN; ++t) tb[t] = make_float4(0, 0, 0, 0);
    if (own_tail) {
        // element j of this lane's quad is k = 8 SF + 4 lh + j: valid while j < nv.  Dword loads with the invalid elements' offsets pushed
        // out of range -- a 16-byte load that straddles the end of the last row would depend on how the range check treats a partly
        // covered access
        const int nv = rem - 4 * lh;
        const unsigned d0 = nv > 0 ? 0u : 0xfffffff0u, d1 = nv > 1 ? 0u : 0xfffffff0u, d2 = nv > 2 ? 0u : 0xfffffff0u, d3 = nv > 3 ? 0u : 0xfffffff0u;
        const unsigned so = (unsigned)SF * 32u;
        ta.x = ld1(rA, vA | d0, so); ta.y = ld1(rA, (vA + 4u) | d1, so); ta.z = ld1(rA, (vA + 8u) | d2, so); ta.w = ld1(rA, (vA + 12u) | d3, so);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            if constexpr (BKN) {
                const unsigned sb = (unsigned)SF * 8u * ldb_b;   // (rows k >= K of W[k][n] are past the descriptor anyway)
                tb[t].x = ld1(rB, vB[t] | d0, sb); tb[t].y = ld1(rB, vB[t] | d1, sb + ldb_b);
                tb[t].z = ld1(rB, vB[t] | d2, sb + 2 * ldb_b); tb[t].w = ld1(rB, vB[t] | d3, sb + 3 * ldb_b);
            } else {
                tb[t].x = ld1(rB, vB[t] | d0, so); tb[t].y = ld1(rB, (vB[t] + 4u) | d1, so);
                tb[t].z = ld1(rB, (vB[t] + 8u) | d2, so); tb[t].w = ld1(rB, (vB[t] + 12u) | d3, so);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < FG_D; ++d) issue(d, s0 + d);
    constexpr int FSTEPS = F64 ? FLUSH / 8 : 1;                  // k steps between two flushes
    int since = 0;
    for (int s = s0; s < s1; s += FG_D) {
#pragma unroll
        for (int d = 0; d < FG_D; ++d) {
            const float4 a = qa[d];
            float4 b[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = qb[d][t];
            issue(d, s + d + FG_D);
            consume(a, b, min(s + d, S + FG_D));                 // (dead steps: A = B = 0; the clamp keeps the constants' index inside the padded arrays)
            if constexpr (F64) { if (++since == FSTEPS) { flush(); since = 0; } }
        }
    }
    if (own_tail) { consume(ta, tb, SF); }
    if constexpr (F64) flush();

    // ---- reduce the KSW k-groups into group 0, in fixed order (round r: group r writes, group 0 adds)
    using red_t = typename std::conditional<F64, double, float>::type;
    red_t* const red = reinterpret_cast<red_t*>(sRed) + (size_t)(wave & 3) * TN * 16 * 64;
    for (int r = 1; r < KSW; ++r) {
        __syncthreads();
        if (ks == r) {
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(t * 16 + i) * 64 + lane] = F64 ? (red_t)accd[F64 ? t : 0][i] : (red_t)acc[t][i];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (F64) accd[t][i] += (double)red[(t * 16 + i) * 64 + lane];
                    else acc[t][i] += (float)red[(t * 16 + i) * 64 + lane];
                }
        }
    }

    // ---- epilogue (k group 0): C/D layout col = lr, row = (i & 3) + 8 (i >> 2) + 4 lh
    double st_s[TN], st_q[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { st_s[t] = 0.0; st_q[t] = 0.0; }
    if (ks == 0) {
        const unsigned ldc_b = (unsigned)p.ldc * 4;
        const rsrc_t rC = rsrc(p.C, (size_t)(m0 + wr * 32) * ldc_b, (size_t)p.M * ldc_b);
        const rsrc_t rLo = rsrc(F64 && p.Clo ? p.Clo : p.C, (size_t)(m0 + wr * 32) * ldc_b, F64 && p.Clo ? (size_t)p.M * ldc_b : 0);
        const bool full_rows = m0 + wr * 32 + 32 <= p.M;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int col = n0 + wc * 32 * TN + t * 32 + lr;
            const bool cin = col < p.N;
            const float bias = (p.bias && cin) ? p.bias[col] : 0.f;
            const unsigned vC = cin ? (unsigned)(4 * lh) * ldc_b + (unsigned)col * 4u : 0xfffffff0u;      // columns past N: stores dropped
            float yv[EP == EP_MASK_STATS ? 16 : 1];
            float esc = 0.f, esh = 0.f;
            if constexpr (EP == EP_MASK_STATS) {
                const unsigned ldy_b = (unsigned)p.ldyp * 4;
                const rsrc_t rY = rsrc(p.Yp, (size_t)(m0 + wr * 32) * ldy_b, (size_t)p.M * ldy_b);
                const unsigned vY = cin ? (unsigned)(4 * lh) * ldy_b + (unsigned)col * 4u : 0xfffffff0u;
#pragma unroll
                for (int i = 0; i < 16; ++i) yv[i] = ld1(rY, vY, (unsigned)((i & 3) + 8 * (i >> 2)) * ldy_b);
                esc = cin ? p.esc[col] : 0.f; esh = cin ? p.esh[col] : 0.f;
            }
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = (i & 3) + 8 * (i >> 2);
                float c;
                if constexpr (F64) {
                    const double cd = accd[t][i] + (double)bias;
                    c = (float)cd;
                    if (p.Clo) st1(rLo, vC, (unsigned)rl * ldc_b, (float)(cd - (double)c));
                } else c = acc[t][i] + bias;
                if constexpr (EP == EP_MASK_STATS) c = fmaf(esc, yv[i], esh) > 0.f ? c : c * p.eslope;
                st1(rC, vC, (unsigned)rl * ldc_b, c);
                if constexpr (EP != EP_STORE) {
                    // rows past M (last row tile only): A read as zeros, the value is act(shift) . W or the bias -- not part of the sums
                    const bool rin = full_rows || (m0 + wr * 32 + rl + 4 * lh < p.M);
                    const double cd = rin ? (double)c : 0.0;
                    s += cd;
                    if constexpr (EP == EP_MASK_STATS) q = fma(cd, (double)yv[i], q);
                    else q = fma(cd, cd, q);
                }
            }
            st_s[t] = s; st_q[t] = q;
        }
    }
    if constexpr (EP != EP_STORE) {
        if (p.stats) {
            // lanes l and l ^ 32 hold the same column; then the two row waves (wr = 0, 1) of k group 0 through LDS
            __syncthreads();
            double* const rs = reinterpret_cast<double*>(sRed);                   // [2 wr][2 wc][TN][32][2]
            if (ks == 0) {
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    double s = st_s[t], q = st_q[t];
                    s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
                    if (lh == 0) { double* d = rs + ((((wr * 2 + wc) * TN + t) * 32 + lr) * 2); d[0] = s; d[1] = q; }
                }
            }
            __syncthreads();
            if (tid < 64 * TN) {
                const int wcc = tid / (32 * TN), t = (tid >> 5) % TN, l = tid & 31;
                const int col = n0 + wcc * 32 * TN + t * 32 + l;
                if (col < p.N) {
                    const double* d0 = rs + ((((0 * 2 + wcc) * TN + t) * 32 + l) * 2);
                    const double* d1 = rs + ((((1 * 2 + wcc) * TN + t) * 32 + l) * 2);
                    double* dst = p.stats + (size_t)rt * 2 * p.N;
                    dst[col] = d0[0] + d1[0]; dst[p.N + col] = d0[1] + d1[1];
                }
            }
        }
    }
}

// ---- dy of one layer, formed once: dy[m][c] = a[c] du[m][c] - (k1[c] + k2[c] (y[m][c] - mean[c])), du dense or the sparse max-pool gradient
struct DyArgs {
    const float* dU; const float* Y; const float* a; const float* k1; const float* k2; const float* mu;
    const int32_t* arg; const float* gz; int ns;
    float* dy; int M, C;
};
__global__ __launch_bounds__(256) void frag_dy_kernel(const DyArgs p) {
    const int c4 = p.C >> 2;
    const size_t n4 = (size_t)p.M * c4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        const int m = (int)(e / c4), c = (int)(e % c4) * 4;
        const float4 y = *reinterpret_cast<const float4*>(p.Y + (size_t)m * p.C + c);
        const float4 a = *reinterpret_cast<const float4*>(p.a + c), k1 = *reinterpret_cast<const float4*>(p.k1 + c);
        const float4 k2 = *reinterpret_cast<const float4*>(p.k2 + c), mu = *reinterpret_cast<const float4*>(p.mu + c);
        float4 du;
        if (p.dU) du = *reinterpret_cast<const float4*>(p.dU + (size_t)m * p.C + c);
        else {
            const int g = m / p.ns, srow = m - g * p.ns;
            const int4 ar = *reinterpret_cast<const int4*>(p.arg + (size_t)g * p.C + c);
            const float4 gz = *reinterpret_cast<const float4*>(p.gz + (size_t)g * p.C + c);
            du.x = ar.x == srow ? gz.x : 0.f; du.y = ar.y == srow ? gz.y : 0.f; du.z = ar.z == srow ? gz.z : 0.f; du.w = ar.w == srow ? gz.w : 0.f;
        }
        float4 d;        // (the arithmetic of mlp.hip's dy-forming loaders, so that the two paths agree bit for bit on dy)
        d.x = fmaf(a.x, du.x, -fmaf(k2.x, y.x - mu.x, k1.x)); d.y = fmaf(a.y, du.y, -fmaf(k2.y, y.y - mu.y, k1.y));
        d.z = fmaf(a.z, du.z, -fmaf(k2.z, y.z - mu.z, k1.z)); d.w = fmaf(a.w, du.w, -fmaf(k2.w, y.w - mu.w, k1.w));
        *reinterpret_cast<float4*>(p.dy + (size_t)m * p.C + c) = d;
    }
}

// ---- weight gradient dW[i][j] = sum_m dy[m][i] z[m][j], z = lrelu(bsc[j] x[m][j] + bsh[j]) or x ---------------------------------------
// A workgroup owns a (32 TM) x (32 TN) x 4-wave ... see the launcher: 2 x 2 waves of 32 TM x 32 TN tiles times KSW row groups; KSG
// workgroups share an output tile (rows split again); the last of them to arrive sums the KSG partial tiles in fixed order.
struct DwArgs2 {
    const float* dy; int ldd;                 // [M][ldd]
    const float* X; int ldx;                  // [M][ldx]
    const float* bsc; const float* bsh; float bslope;      // null: plain x
    float* dW; int ldo;                       // [I][ldo], columns < J written
    float* part;                              // [KSG][nti * ntj][64 TM x 64 TN] partial tiles (KSG > 1)
    unsigned* counters;                       // [nti * ntj], zero before the launch; the last arriver re-zeroes its counter
    int M, I, J;
    int nti, ntj, ksg;
};

template <int TM, int TN>
__global__ __launch_bounds__(1024) void frag_dw_kernel(const DwArgs2 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int KSW = blockDim.x >> 8;
    const int ks = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    const int ntile = p.nti * p.ntj;
    const int tile = blockIdx.x % ntile, kg = blockIdx.x / ntile;          // (consecutive ids = different tiles of one row slab: X / dy rows shared in L2)
    const int i0 = (tile % p.nti) * 64 * TM, j0 = (tile / p.nti) * 64 * TN;
    // rows of this workgroup's k group kg, then of this wave group: steps of 8 rows
    const int S = (p.M + 7) >> 3;
    const int per_g = (S + p.ksg - 1) / p.ksg;
    const int g0 = kg * per_g, g1 = min(S, g0 + per_g);
    const int per_w = (max(g1 - g0, 0) + KSW - 1) / KSW;
    const int s0 = g0 + ks * per_w, s1 = min(g1, s0 + per_w);
    const unsigned ldd_b = (unsigned)p.ldd * 4, ldx_b = (unsigned)p.ldx * 4;
    const rsrc_t rD = rsrc(p.dy, 0, (size_t)p.M * ldd_b), rX = rsrc(p.X, 0, (size_t)p.M * ldx_b);
    // lane (lr, lh), element j of step s: row 8 s + 4 lh + j, column (tile column) + lr: coalesced dword loads; columns past I / J read 0
    unsigned vA[TM], vB[TN];
    float zsc[TN], zsh[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) { const int c = i0 + (wr * TM + t) * 32 + lr; vA[t] = c < p.I ? (unsigned)(4 * lh) * ldd_b + (unsigned)c * 4u : 0xfffffff0u; }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int c = j0 + (wc * TN + t) * 32 + lr;
        vB[t] = c < p.J ? (unsigned)(4 * lh) * ldx_b + (unsigned)c * 4u : 0xfffffff0u;
        zsc[t] = (p.bsc && c < p.J) ? p.bsc[c] : 1.f; zsh[t] = (p.bsc && c < p.J) ? p.bsh[c] : 0.f;
    }
    const bool act = p.bsc != nullptr;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int D = 2;
    float4 qa[D][TM], qb[D][TN];
    auto issue = [&](int slot, int s) {
        const bool live = s < s1;
        const unsigned sa = live ? (unsigned)s * 8u * ldd_b : SOFF_DEAD, sb = live ? (unsigned)s * 8u * ldx_b : SOFF_DEAD;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            qa[slot][t].x = ld1(rD, vA[t], sa); qa[slot][t].y = ld1(rD, vA[t], sa + ldd_b);
            qa[slot][t].z = ld1(rD, vA[t], sa + 2 * ldd_b); qa[slot][t].w = ld1(rD, vA[t], sa + 3 * ldd_b);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            qb[slot][t].x = ld1(rX, vB[t], sb); qb[slot][t].y = ld1(rX, vB[t], sb + ldx_b);
            qb[slot][t].z = ld1(rX, vB[t], sb + 2 * ldx_b); qb[slot][t].w = ld1(rX, vB[t], sb + 3 * ldx_b);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, s0 + d);
    for (int s = s0; s < s1; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = qa[d][t];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = qb[d][t];
            issue(d, s + d + D);
            // rows past M (the last step) and dead steps read 0 on the dy side: whatever z is there multiplies 0
            if (act) {
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    b[t].x = lrelu_max(fmaf(zsc[t], b[t].x, zsh[t]), p.bslope); b[t].y = lrelu_max(fmaf(zsc[t], b[t].y, zsh[t]), p.bslope);
                    b[t].z = lrelu_max(fmaf(zsc[t], b[t].z, zsh[t]), p.bslope); b[t].w = lrelu_max(fmaf(zsc[t], b[t].w, zsh[t]), p.bslope);
                }
            }
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb) {
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta].x, b[tb].x, acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta].y, b[tb].y, acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta].z, b[tb].z, acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta].w, b[tb].w, acc[ta][tb], 0, 0, 0);
                }
        }
    }
    // ---- the KSW row groups of the workgroup -> group 0, fixed order
    float* const red = reinterpret_cast<float*>(lds_raw) + (size_t)(wave & 3) * TM * TN * 16 * 64;
    for (int r = 1; r < KSW; ++r) {
        __syncthreads();
        if (ks == r) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) red[((a * TN + b) * 16 + i) * 64 + lane] = acc[a][b][i];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] += red[((a * TN + b) * 16 + i) * 64 + lane];
        }
    }
    // ---- across the KSG workgroups of the tile: partial tiles in the accumulator layout ([a][b][i][wave 0..3][lane]: coalesced), one
    // counter per tile; the last arriver adds the others' partials to its own in the order kg = 0, 1, ... (its own at its place)
    constexpr int TILE_F = TM * TN * 16 * 256;               // floats per partial tile
    bool writer = true;
    if (p.ksg > 1) {
        float* mine = p.part + ((size_t)kg * ntile + tile) * TILE_F;
        if (ks == 0) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) mine[((a * TN + b) * 16 + i) * 256 + (wave & 3) * 64 + lane] = acc[a][b][i];
        }
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_ticket = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        writer = s_ticket == (unsigned)p.ksg - 1;
        if (writer) {
            if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); p.counters[tile] = 0; }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float sum = 0.f;
                            for (int g = 0; g < p.ksg; ++g) {
                                const float* src = p.part + ((size_t)g * ntile + tile) * TILE_F;
                                // (sc1: served by L2 / memory, never by this CU's L1 -- the other workgroups' tiles were written elsewhere)
                                sum += __hip_atomic_load(src + ((a * TN + b) * 16 + i) * 256 + (wave & 3) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            acc[a][b][i] = sum;
                        }
            }
        }
    }
    if (writer && ks == 0) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = j0 + (wc * TN + b) * 32 + lr;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = i0 + (wr * TM + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                    if (row < p.I && col < p.J) p.dW[(size_t)row * p.ldo + col] = acc[a][b][i];
                }
            }
    }
}

static int cu_count() {
    static const int n = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        (void)hipGetLastError();
        return cu;
    }();
    return n;
}

// k groups per workgroup so that the launch puts >= 2 waves on every SIMD when the output alone does not (<= 4: 1024 threads), while a
// wave keeps >= 8 k steps
static int pick_ksw(int tiles, int steps) {
    const int want = 2 * 4 * cu_count();                     // waves
    int ksw = 1;
    while (ksw < 4 && tiles * 4 * ksw < want && steps / (2 * ksw) >= 8) ksw *= 2;
    return ksw;
}

#define FG_LAUNCH(kernel, grid, blk, lds, st, ...)                                                                      \
    do {                                                                                                                \
        ::pcl::TimeHook& h_ = ::pcl::time_hook();                                                                       \
        h_.last_kernel = #kernel;                                                                                       \
        if (h_.start && ::pcl::time_hook_matches(h_)) {                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, blk, lds, st, h_.start, h_.stop, 0, __VA_ARGS__);                       \
            h_.start = h_.stop = nullptr;                                                                               \
        } else hipLaunchKernelGGL(kernel, grid, blk, lds, st, __VA_ARGS__);                                             \
    } while (0)

template <int TN, int FLUSH, bool ACT, bool BKN, int EP>
static void launch_gemm_t(const GemmArgs& g, int ksw, hipStream_t st) {
    const int S = g.K / 8 + ((g.K & 7) ? 1 : 0);
    const size_t cst = ACT ? (size_t)2 * (S + FG_D + 1) * 8 * 4 : 0;
    size_t red = (size_t)4 * TN * 16 * 64 * (FLUSH > 0 ? 8 : 4);                       // k-group reduction buffer (also the statistics exchange)
    const size_t lds = cst + red;
    if (lds > 64 * 1024) {
        // the folded input BatchNorm's constants (2 x 32 bytes per 8 k) beside the reduction buffer pass the 64 KB default from Cin ~ 4 000 on
        // (the entry point admits Cin <= 8 192: 66 KB + 32 KB): raise the kernel's limit, once per instantiation
        static size_t raised = 0;
        if (lds > raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(frag_gemm_kernel<TN, FLUSH, ACT, BKN, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            raised = lds;
        }
    }
    FG_LAUNCH((frag_gemm_kernel<TN, FLUSH, ACT, BKN, EP>), dim3(g.nrt * g.nct), dim3(256 * ksw), lds, st, g);
}

template <int FLUSH, bool ACT, bool BKN, int EP>
static void launch_gemm(GemmArgs g, hipStream_t st, int force_tn, int force_ksw) {
    const int width = g.N - g.n_begin;
    g.nrt = (g.M + 63) / 64;
    // 128-column wave pairs (two B fragments per A fragment) when the output still fills the chip twice over with them
    int tn = (width >= 128 && g.nrt * ((width + 127) / 128) * 4 >= 8 * cu_count()) ? 2 : 1;
    // ... or one workgroup per CU with them on a long contraction (4 096 x 1024 -> 512: 48.1 -> 41.2 us with two k groups; 512 -> 1024 forward:
    // 55.9 -> 51.4; tools/dbg/frag_time.py)
    if (width >= 256 && g.K >= 512 && g.nrt * ((width + 127) / 128) >= cu_count()) tn = 2;
    if (force_tn) tn = force_tn;
    if (FLUSH > 0) tn = 1;              // (fp64 accumulators: 32 more registers per tile -- two tiles do not fit 128 registers at 4 waves per SIMD)
    g.nct = (width + 64 * tn - 1) / (64 * tn);
    const int ksw = force_ksw ? force_ksw : pick_ksw(g.nrt * g.nct, g.K / 8);
    if constexpr (FLUSH == 0) { if (tn == 2) { launch_gemm_t<2, FLUSH, ACT, BKN, EP>(g, ksw, st); return; } }
    launch_gemm_t<1, FLUSH, ACT, BKN, EP>(g, ksw, st);
}

static int g_force_tn = 0, g_force_ksw = 0, g_force_dw[4] = {0, 0, 0, 0};     // lab / test knobs (pcl_frag_set_tuning)

// dW launch shape: tile (64 TM x 64 TN), KSW row groups per workgroup, KSG workgroups per tile
struct DwShape { int tm, tn, ksw, ksg, nti, ntj; };
static DwShape dw_shape(int M, int I, int J) {
    DwShape s;
    s.tm = 1; s.tn = 1;
    if (g_force_dw[0]) { s.tm = g_force_dw[0]; s.tn = g_force_dw[1]; }
    s.nti = (I + 64 * s.tm - 1) / (64 * s.tm); s.ntj = (J + 64 * s.tn - 1) / (64 * s.tn);
    const int tiles = s.nti * s.ntj, steps = (M + 7) / 8;
    s.ksw = 4;
    while (s.ksw > 1 && steps / s.ksw < 8) s.ksw >>= 1;
    int ksg = (cu_count() + tiles - 1) / tiles;                                  // one workgroup per CU
    const int cap = steps / (s.ksw * 8) > 0 ? steps / (s.ksw * 8) : 1;           // >= 8 steps per wave
    if (ksg > cap) ksg = cap;
    if (ksg > 16) ksg = 16;
    if (ksg < 1) ksg = 1;
    s.ksg = ksg;
    if (g_force_dw[2]) s.ksw = g_force_dw[2];
    if (g_force_dw[3]) s.ksg = g_force_dw[3];
    return s;
}
static size_t dw_tile_floats(const DwShape& s) { return (size_t)s.tm * s.tn * 16 * 256; }

}  // namespace fg

// What the staged kernels' callers (stack.hip) ask: may a plain stack of P rows take the fragment path, and how many statistics rows
// does a forward / dX GEMM write then (one per 64-row tile)?
static int g_frag_max_rows = 8192;
bool frag_rows_eligible(int P) { return P >= 1 && P <= g_frag_max_rows; }
int frag_stat_rows(int P) { return (P + 63) / 64; }

}  // namespace pcl
using namespace pcl;
using namespace pcl::fg;

extern "C" void pcl_frag_set_tuning(int max_rows, int force_tn, int force_ksw, int dw_tm, int dw_tn, int dw_ksw, int dw_ksg) {
    if (max_rows >= 0) g_frag_max_rows = max_rows;
    g_force_tn = force_tn; g_force_ksw = force_ksw;
    g_force_dw[0] = dw_tm; g_force_dw[1] = dw_tn; g_force_dw[2] = dw_ksw; g_force_dw[3] = dw_ksg;
}
extern "C" int pcl_frag_max_rows(void) { return g_frag_max_rows; }
extern "C" int pcl_frag_stat_rows(int P) { return P < 1 ? 1 : frag_stat_rows(P); }

static bool small_enough(const void* base, size_t bytes) { (void)base; return bytes < 0x7fffffffull; }

extern "C" int pcl_frag_linear_fwd_f32(const float* X, int ldx, const float* W, int ldw, const float* bias, const float* in_scale,
                                       const float* in_shift, float in_slope, int P, int Cin, int Cout, float* Y, int ldy, float* Y_lo, double* stats_ws,
                                       int flush_k, void* stream) {
    PCL_REQUIRE(X && W && Y, "pcl_frag_linear_fwd_f32: null pointer");
    PCL_REQUIRE(P >= 1 && Cin >= 1 && Cout >= 1 && ldx >= Cin && ldw >= Cin && ldy >= Cout, "pcl_frag_linear_fwd_f32: bad sizes P=%d Cin=%d Cout=%d ldx=%d ldw=%d ldy=%d", P, Cin,
                Cout, ldx, ldw, ldy);
    PCL_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "pcl_frag_linear_fwd_f32: in_scale / in_shift come together");
    PCL_REQUIRE(flush_k == 0 || flush_k == 8 || flush_k == 32, "pcl_frag_linear_fwd_f32: flush_k = %d (0: fp32 accumulation; 8 or 32: fp64 every that many terms)", flush_k);
    PCL_REQUIRE(in_slope >= 0.f && in_slope <= 1.f, "pcl_frag_linear_fwd_f32: slope %f outside [0,1]", in_slope);
    PCL_REQUIRE(small_enough(X, (size_t)P * ldx * 4) && small_enough(W, (size_t)Cout * ldw * 4) && small_enough(Y, (size_t)P * ldy * 4),
                "pcl_frag_linear_fwd_f32: operands beyond 2 GiB take the staged kernels");
    PCL_REQUIRE(Cin <= 8192, "pcl_frag_linear_fwd_f32: Cin = %d (the folded BatchNorm of the input lives in LDS: <= 8192)", Cin);
    PCL_REQUIRE(!Y_lo || flush_k != 0, "pcl_frag_linear_fwd_f32: Y_lo (the residual of the fp64 sum) needs flush_k = 8 or 32");
    hipStream_t st = as_stream(stream);
    GemmArgs g = {};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.bias = bias; g.asc = in_scale; g.ash = in_shift; g.aslope = in_slope;
    g.C = Y; g.ldc = ldy; g.Clo = Y_lo; g.stats = stats_ws; g.M = P; g.N = Cout; g.K = Cin; g.n_begin = 0;
    const bool act = in_scale != nullptr;
#define FWD(FL) do { if (stats_ws) { if (act) launch_gemm<FL, true, false, EP_STATS>(g, st, g_force_tn, g_force_ksw); else launch_gemm<FL, false, false, EP_STATS>(g, st, g_force_tn, g_force_ksw); } \
                     else { if (act) launch_gemm<FL, true, false, EP_STORE>(g, st, g_force_tn, g_force_ksw); else launch_gemm<FL, false, false, EP_STORE>(g, st, g_force_tn, g_force_ksw); } } while (0)
    if (flush_k == 0) FWD(0); else if (flush_k == 8) FWD(8); else FWD(32);
#undef FWD
    return check_launch("pcl_frag_linear_fwd_f32");
}

extern "C" int pcl_frag_dy_f32(const float* dU, const float* Y, const float* a, const float* k1, const float* k2, const float* mu,
                               const int32_t* arg, const float* gz, int ns, int P, int C, float* dy, uint32_t* zero_words, int n_zero, void* stream) {
    PCL_REQUIRE(Y && a && k1 && k2 && mu && dy && P >= 1 && C >= 4 && C % 4 == 0, "pcl_frag_dy_f32: bad arguments (C must be a multiple of 4)");
    PCL_REQUIRE((dU != nullptr) != (arg != nullptr && gz != nullptr), "pcl_frag_dy_f32: pass dU or (arg, gz)");
    PCL_REQUIRE(dU || ns >= 1, "pcl_frag_dy_f32: ns = %d", ns);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    PCL_REQUIRE(al16(Y) && al16(a) && al16(k1) && al16(k2) && al16(mu) && al16(dy) && (!dU || al16(dU)) && (!gz || (al16(gz) && al16(arg))),
                "pcl_frag_dy_f32: operands must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (zero_words && n_zero > 0) {
        if (hipMemsetAsync(zero_words, 0, (size_t)n_zero * 4, st) != hipSuccess) return fail(PCL_EHIP, "pcl_frag_dy_f32: memset failed");
    }
    DyArgs d = {dU, Y, a, k1, k2, mu, arg, gz, ns, dy, P, C};
    const size_t n4 = (size_t)P * (C / 4);
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(frag_dy_kernel, dim3(blocks), dim3(256), 0, st, d);
    return check_launch("pcl_frag_dy_f32");
}

extern "C" int pcl_frag_linear_bwd_dx_f32(const float* dy, const float* W, int ldw, int P, int Cout, int Cin, const float* Yprev, int ldyp,
                                          const float* prev_scale, const float* prev_shift, float prev_slope, float* dUprev, int ldu,
                                          double* stats_ws, int first_col, void* stream) {
    PCL_REQUIRE(dy && W && dUprev && P >= 1 && Cout >= 1 && Cin >= 1 && ldw >= Cin && ldu >= Cin, "pcl_frag_linear_bwd_dx_f32: bad arguments");
    PCL_REQUIRE(first_col >= 0 && first_col < Cin, "pcl_frag_linear_bwd_dx_f32: first_col = %d", first_col);
    PCL_REQUIRE((Yprev == nullptr) == (prev_scale == nullptr) && (Yprev == nullptr) == (prev_shift == nullptr), "pcl_frag_linear_bwd_dx_f32: Yprev / prev_scale / prev_shift come together");
    PCL_REQUIRE(!Yprev || (stats_ws && ldyp >= Cin), "pcl_frag_linear_bwd_dx_f32: a masked epilogue needs stats_ws and ldyp >= Cin");
    PCL_REQUIRE(prev_slope >= 0.f && prev_slope <= 1.f, "pcl_frag_linear_bwd_dx_f32: slope %f outside [0,1]", prev_slope);
    PCL_REQUIRE(small_enough(dy, (size_t)P * Cout * 4) && small_enough(dUprev, (size_t)P * ldu * 4), "pcl_frag_linear_bwd_dx_f32: operands beyond 2 GiB take the staged kernels");
    hipStream_t st = as_stream(stream);
    GemmArgs g = {};
    g.A = dy; g.lda = Cout; g.B = W; g.ldb = ldw; g.C = dUprev; g.ldc = ldu; g.stats = stats_ws; g.M = P; g.N = Cin; g.K = Cout; g.n_begin = first_col;
    g.Yp = Yprev; g.ldyp = ldyp; g.esc = prev_scale; g.esh = prev_shift; g.eslope = prev_slope;
    if (Yprev) launch_gemm<0, false, true, EP_MASK_STATS>(g, st, g_force_tn, g_force_ksw);
    else launch_gemm<0, false, true, EP_STORE>(g, st, g_force_tn, g_force_ksw);
    return check_launch("pcl_frag_linear_bwd_dx_f32");
}

extern "C" size_t pcl_frag_dw_workspace_bytes(int P, int Cout, int Cin) {
    if (P < 1 || Cout < 1 || Cin < 1) return 0;
    const DwShape s = dw_shape(P, Cout, Cin);
    const size_t tiles = (size_t)s.nti * s.ntj;
    return 256 + tiles * 4 + (s.ksg > 1 ? (size_t)s.ksg * tiles * dw_tile_floats(s) * 4 : 0) + 256;
}
/* the words pcl_frag_dy_f32 (or the caller) must have cleared before the dW launch: the head of the workspace */
extern "C" int pcl_frag_dw_counter_words(int P, int Cout, int Cin) {
    if (P < 1 || Cout < 1 || Cin < 1) return 0;
    const DwShape s = dw_shape(P, Cout, Cin);
    return s.nti * s.ntj;
}

extern "C" int pcl_frag_linear_bwd_dw_f32(const float* dy, const float* X, int ldx, const float* prev_scale, const float* prev_shift, float prev_slope,
                                          int P, int Cout, int Cin, float* dW, int ldo, void* workspace, size_t workspace_bytes, int counters_cleared,
                                          void* stream) {
    PCL_REQUIRE(dy && X && dW && P >= 1 && Cout >= 1 && Cin >= 1 && ldx >= Cin && ldo >= Cin, "pcl_frag_linear_bwd_dw_f32: bad arguments");
    PCL_REQUIRE((prev_scale == nullptr) == (prev_shift == nullptr), "pcl_frag_linear_bwd_dw_f32: prev_scale / prev_shift come together");
    PCL_REQUIRE(prev_slope >= 0.f && prev_slope <= 1.f, "pcl_frag_linear_bwd_dw_f32: slope %f outside [0,1]", prev_slope);
    PCL_REQUIRE(small_enough(dy, (size_t)P * Cout * 4) && small_enough(X, (size_t)P * ldx * 4), "pcl_frag_linear_bwd_dw_f32: operands beyond 2 GiB take the staged kernels");
    const size_t need = pcl_frag_dw_workspace_bytes(P, Cout, Cin);
    if (!workspace || workspace_bytes < need) return fail(PCL_EWS, "pcl_frag_linear_bwd_dw_f32: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    const DwShape s = dw_shape(P, Cout, Cin);
    const size_t tiles = (size_t)s.nti * s.ntj;
    char* base = static_cast<char*>(workspace);
    base += (256 - (reinterpret_cast<uintptr_t>(base) & 255)) & 255;
    unsigned* counters = reinterpret_cast<unsigned*>(base);
    float* part = reinterpret_cast<float*>(base + ((tiles * 4 + 255) & ~(size_t)255));
    if (!counters_cleared && s.ksg > 1) {
        if (hipMemsetAsync(counters, 0, tiles * 4, st) != hipSuccess) return fail(PCL_EHIP, "pcl_frag_linear_bwd_dw_f32: memset failed");
    }
    DwArgs2 d = {};
    d.dy = dy; d.ldd = Cout; d.X = X; d.ldx = ldx; d.bsc = prev_scale; d.bsh = prev_shift; d.bslope = prev_slope; d.dW = dW; d.ldo = ldo;
    d.part = part; d.counters = counters; d.M = P; d.I = Cout; d.J = Cin; d.nti = s.nti; d.ntj = s.ntj; d.ksg = s.ksg;
    const dim3 grid((unsigned)(tiles * s.ksg)), blk(256 * s.ksw);
    const size_t lds = (size_t)4 * s.tm * s.tn * 16 * 64 * 4;
    if (s.tm == 1 && s.tn == 1) FG_LAUNCH((frag_dw_kernel<1, 1>), grid, blk, lds, st, d);
    else if (s.tm == 1 && s.tn == 2) FG_LAUNCH((frag_dw_kernel<1, 2>), grid, blk, lds, st, d);
    else if (s.tm == 2 && s.tn == 1) FG_LAUNCH((frag_dw_kernel<2, 1>), grid, blk, lds, st, d);
    else FG_LAUNCH((frag_dw_kernel<2, 2>), grid, blk, lds, st, d);
    return check_launch("pcl_frag_linear_bwd_dw_f32");
}
