// mlp.hip -- per-group pointwise MLP (1x1 conv + BatchNorm(train) + (Leaky)ReLU [+ max over the group])
// for gfx950, forward and backward, on fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak).
//
// Reference arithmetic: build_mlps / PointNetModuleBase.execute, /root/reference/networks/cls/pointnet2.py:18-62
// (nn.Conv(k=1,bias=not bn) + nn.BatchNorm + nn.ReLU, then argmax(dim=2)[1] = max VALUE over the group),
// DGCNN conv1-4 networks/cls/dgcnn.py:72-83, FP stacks misc/ops.py:54-64.  A 1x1 conv over [B,C,m,ns] is a
// row-wise linear map over P = B*m*ns channel-last rows, so everything here is a thin-K GEMM on [P,C] rows.
//
// What is fused (the activations are 0.25-0.5 GB each; HBM traffic, not FLOPs, bounds the big layers):
//   * the previous layer's folded BatchNorm + activation is applied while STAGING the A operand
//     (never materialised);
//   * BatchNorm batch statistics (sum, sum of squares per channel) are produced by the GEMM epilogue as
//     per-workgroup fp64 partials (no atomics; summed by the tiny finalize kernel);
//   * backward: dy = a*du - k1 - k2*(y - mean) (BatchNorm backward, affine per channel once the two channel sums
//     are known) is formed while staging, for the dense case and for the sparse max-pool gradient; the ReLU mask
//     of the layer below and ITS two channel sums are produced by the dX GEMM's epilogue; the dX GEMM reads the
//     weight matrix as stored (K-major B tile, transposed on its way into LDS);
//   * statistics are summed about a per-lane pivot in fp32 and shifted back in fp64 (|mean| >> std loses nothing).
//
// Tiling (wave64): workgroup = 4 waves, block tile 128x128 (forward, wide N), 128x64 (N <= 64 and every backward
// GEMM) or 64x64 (small M), K step 32; each wave owns a (32*TM)x(32*TN) tile of MFMA 32x32 accumulators.  LDS rows are padded to 36 dwords so the per-lane ds_read_b128
// fragment reads (4 consecutive k per lane; lanes 0-31 take k..k+3, lanes 32-63 take k+4..k+7 of each
// 8-wide k block, identically for A and B) are bank-conflict-free.  Grids are persistent over row tiles so
// statistics are reduced in registers across tiles.
#include <cstdlib>
#include "common.h"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mlp.hip is written for gfx950 (MI355X) only: its buffer addressing relies on the range check covering the scalar offset (tools/ubench/bufcheck.hip)"
#endif

// PCL_EXP: timing experiments only (csrc/Makefile EXP=n builds a separate library; the product build has PCL_EXP == 0)
#ifndef PCL_EXP
#define PCL_EXP 0
#endif

namespace pcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, LDS_LD = BK + 4;
constexpr int MLP_T = 256;
constexpr int STAT_ROWS = 1024;  // max workgroups along the row axis == max rows of a stats workspace

enum AMode { A_PLAIN = 0, A_BNACT = 1, A_DY = 2, A_DY_SPARSE = 3 };
enum EMode { E_STORE_STATS = 0, E_MASK_STORE_STATS = 1, E_STORE = 2 };

struct LinArgs {
    const float* A;      // [M,K] primary rows (X / Y_prev / dU)
    const float* A2;     // [M,K] pre-BN output y of THIS layer (A_DY, A_DY_SPARSE)
    int ldb;             // BT only: row stride of B
    const float* B;      // forward: W[N=Cout][K=Cin]; dX (BT): the same W[K=Cout][ldb >= N=Cin], consumed K-major
    const float* bias;   // [N] or null
    const float* sc;     // [K] A_BNACT: scale; A_DY*: a = gamma*invstd
    const float* sh;     // [K] A_BNACT: shift; A_DY*: k1
    const float* k2;     // [K] A_DY*
    const float* mu;     // [K] A_DY*: batch mean of Y (the k2 term is applied to y - mu)
    const int32_t* arg;  // [M/ns, K] A_DY_SPARSE: row-in-group of the max
    const float* gz;     // [M/ns, K] A_DY_SPARSE: gradient at the max (already masked by the activation)
    float* C;            // [M,N], row stride ldc (>= N)
    int ldc;
    double* stats;       // [STAT_ROWS][2][N] partials (E_STORE_STATS, E_MASK_STORE_STATS)
    const float* Yprev;  // [M,N] E_MASK_STORE_STATS: pre-BN output of the layer below
    const float* esc;    // [N]   its folded BN scale
    const float* esh;    // [N]   its folded BN shift
    float* gmax;         // [M/ns, N] E_STORE_STATS + GM: per-group max / min of the stored rows and the row-in-group
    float* gmin;         //           attaining them (first occurrence), for the fused max-pool
    int32_t* gamax;
    int32_t* gamin;
    const int* m_dev;    // optional device-resident row count (ragged / duplicate-compacted rows): M = *m_dev
    const int2* rmeta;   // optional per-row {group id, pos-in-group | multiplicity << 16} of compacted rows
    float slope, eslope;
    int M, N, K, ns;
    int n_begin;         // first output column computed (columns below it are never written): input-gradient
                         // GEMMs skip the xyz columns of a grouped tensor, which have no consumer
    int a_mode, e_mode;
    int gx, nt;          // logical grid: gx persistent row-workgroups x nt column tiles (see tile_of_block)
};

// XCD-aware block -> (row-workgroup, column tile).  The hardware hands consecutive workgroup ids to the 8 XCDs round
// robin, each XCD has its own L2.  With a 2-D grid the nt column tiles of a row tile are gx ids apart: they land on the
// same XCD only by accident and run a whole grid pass apart, so the shared A rows come from HBM / Infinity Cache nt
// times (PMC: 571 MB for 361 MB algorithmic on dx 256->128).  Here ids 8 apart share the row-workgroup index: same XCD,
// dispatched back to back, resident together -- the second tile's A rows are L2 hits.
__device__ __forceinline__ void tile_of_block(int id, int gx, int nt, int& bx, int& by) {
    const int full = gx & ~7;                       // row-workgroups in complete groups of 8
    if (id < full * nt) {
        const int j = id >> 3;
        by = j % nt;
        bx = (j / nt) * 8 + (id & 7);
    } else {
        const int r = id - full * nt, rem = gx - full;
        bx = full + r % rem;
        by = r / rem;
    }
}

// Duplicate-compacted ("ragged") rows: ball query pads a group with copies of its first hit (misc/ops.py:321-324);
// identical input rows give identical activations, so the stack runs once per DISTINCT row and carries the
// multiplicity w: BatchNorm sums use w*y and w*y^2, and in backward the dense BatchNorm term of a row counts w
// times (the sparse max-pool gradient goes to the first occurrence only).  Results equal the padded computation
// up to fp32 summation order.
__device__ __forceinline__ void row_meta(const int2 m, int& g, int& srow, float& w) {
    g = m.x; srow = m.y & 0xffff; w = (float)(m.y >> 16);
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

// ---- staging: global -> registers (raw) ... MFMAs of the current step ... -> transform -> LDS ------------
// The raw operands of step i+1 are requested before the MFMAs of step i and only touched (fused transform +
// LDS store) after them, so HBM/L2 latency hides under the matrix pipe.
// Vector path (K % 4 == 0): thread t owns k4 = (t&7)*4 of rows (t>>3) + 32*i.
// Scalar path: thread t owns k = t&31 of rows (t>>5) + 8*i.
template <bool VEC, int ROWS>
struct Stage {
    static constexpr int NI = VEC ? ROWS / 32 : ROWS / 8;
    float4 v[VEC ? NI : 1], v2[VEC ? NI : 1], vg[VEC ? NI : 1];
    int4 vi[VEC ? NI : 1];
    float s[VEC ? 1 : NI], s2[VEC ? 1 : NI], sg[VEC ? 1 : NI];
    int si[VEC ? 1 : NI];
    float4 c_sc, c_sh, c_k2, c_mu; // per-k constants of this step (vector path)
    float f_sc, f_sh, f_k2, f_mu;  // (scalar path)
    int m0;                        // first row of the staged tile (for the sparse row-in-group test)
};

// Per-row metadata of the tile being staged (group id, row-in-group, multiplicity).  It depends on the ROW only, so it
// is fetched once per tile -- one tile ahead, as raw int2 records -- instead of once per K step: with compacted rows the
// sparse loader needs the group id before it can address (arg, gz), and a metadata load inside every K step put a full
// memory round trip between the barrier and the MFMAs of that step (the dominant dX launch ran at 0.40 of the MFMA roof).
template <bool VEC, int ROWS>
struct RowInfo {
    static constexpr int NI = VEC ? ROWS / 32 : ROWS / 8;
    int g[NI];                     // group id (sparse max gradient)
    int rs[NI];                    // row-in-group
    float rw[NI];                  // row multiplicity (ragged rows), 1 otherwise
    int2 raw[NI];                  // prefetched records of the tile after the staged one
};

template <int AM, bool VEC, int ROWS, bool RAG>
__device__ __forceinline__ void rowinfo_fetch(const LinArgs& p, int m0, int tid, RowInfo<VEC, ROWS>& ri) {
    if constexpr (AM >= A_DY && RAG) {
#pragma unroll
        for (int i = 0; i < RowInfo<VEC, ROWS>::NI; ++i) {
            const int r = min(m0 + (VEC ? (tid >> 3) + 32 * i : (tid >> 5) + 8 * i), p.M - 1);
            ri.raw[i] = p.rmeta[r];
        }
    }
}

template <int AM, bool VEC, int ROWS, bool RAG>
__device__ __forceinline__ void rowinfo_adopt(const LinArgs& p, int m0, int tid, RowInfo<VEC, ROWS>& ri) {
    if constexpr (AM >= A_DY) {
#pragma unroll
        for (int i = 0; i < RowInfo<VEC, ROWS>::NI; ++i) {
            int g = 0, srow = 0;
            float w = 1.f;
            if constexpr (RAG) row_meta(ri.raw[i], g, srow, w);
            else if constexpr (AM == A_DY_SPARSE) {
                const int r = min(m0 + (VEC ? (tid >> 3) + 32 * i : (tid >> 5) + 8 * i), p.M - 1);
                g = r / p.ns; srow = r - g * p.ns;
            }
            ri.g[i] = g; ri.rs[i] = srow; ri.rw[i] = w;
        }
    }
}

// ---- vector path: buffer (descriptor) addressing -------------------------------------------------------------------------
// fp32-input MFMA executes on the vector ALU datapath of gfx950: an MFMA-only wave and a VALU-only wave on one SIMD take
// the SUM of their times, not the max (tools/ubench/coissue.hip: 1726 us + 1357 us -> 3177 us), so every VALU instruction
// of the staging and the epilogue is paid in matrix throughput, one for one.  The vector path therefore spends none on
// addresses or masks: operands are read through buffer descriptors rebased per tile (32-bit per-lane offsets computed once
// per kernel, the k block in the scalar offset), and the hardware range check returns zeros for rows past M, rows of W
// past N and k past K.  What is left unmasked is harmless: a zero-filled A row becomes act(shift) or a nonzero dy, but a
// C row depends on its own A row only, is not stored (partial tiles take the checked epilogue) and is not summed; an
// A value at k >= K multiplies a zero of B.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned BUF_OOB = 0xfffffff0u;           // a per-lane offset no descriptor here covers: reads 0

// descriptor over [base + first_byte, base + total_bytes); both wave-uniform (read-first-laned so that hipcc can prove it)
__device__ __forceinline__ rsrc_t buf_rsrc(const void* base, size_t first_byte, size_t total_bytes) {
    const size_t left = total_bytes > first_byte ? total_bytes - first_byte : 0;
    const unsigned n = left > 0xffffffffull ? 0xffffffffu : (unsigned)left;
    const uintptr_t a = reinterpret_cast<uintptr_t>(base) + first_byte;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ int4 buf_ld4i(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
}
__device__ __forceinline__ float buf_ld1(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st1(rsrc_t r, unsigned voff, unsigned soff, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st4(rsrc_t r, unsigned voff, unsigned soff, float4 x) {
    const u32x4 v = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}
__device__ __forceinline__ int2 buf_ld2i(rsrc_t r, unsigned voff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
    return make_int2((int)v.x, (int)v.y);
}

// ---- fp32 operands as three bf16 planes (round 4; the method is described at linear_fwd_split_kernel) --------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pack(float x, float y) {          // v_cvt_pk_bf16_f32 (round to nearest even)
    bf16x2 v = {(__bf16)x, (__bf16)y};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// (x, y) -> three packed words, x in the low half: x = lo(p0) + lo(p1) + lo(p2) exactly
__device__ __forceinline__ void split3(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = bf16_pack(x, y);
    const float rx = x - bf16_lo(p0), ry = y - bf16_hi(p0);
    p1 = bf16_pack(rx, ry);
    p2 = bf16_pack(rx - bf16_lo(p1), ry - bf16_hi(p1));
}
struct Split8 { bf16x8 pl[3]; };
__device__ __forceinline__ Split8 split8(float4 u, float4 v) {
    uint4 w0, w1, w2;
    split3(u.x, u.y, w0.x, w1.x, w2.x); split3(u.z, u.w, w0.y, w1.y, w2.y);
    split3(v.x, v.y, w0.z, w1.z, w2.z); split3(v.z, v.w, w0.w, w1.w, w2.w);
    Split8 s;
    s.pl[0] = __builtin_bit_cast(bf16x8, w0); s.pl[1] = __builtin_bit_cast(bf16x8, w1); s.pl[2] = __builtin_bit_cast(bf16x8, w2);
    return s;
}
// acc += A . B over 16 k, A and B as three planes each: the nine partial products, smallest first
__device__ __forceinline__ f32x16 mfma_split9(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

constexpr int PLD = BK * 2 + 16;           // bytes per row of a staged bf16 plane (32 k + 16 bytes: conflict-free 16-byte fragment reads)

// per-thread offsets (bytes) of the vector path, computed once per kernel: A rows (tid>>3)+32i at k quad tid&7;
// B forward: rows n = (tid>>3)+32i of W[N][K]; B backward (BT): row k = tid&31, four n at (tid>>5)*4+32i of W[K][ldb]
template <int TBM, int TBN, bool BT>
struct VecOff {
    unsigned a[TBM / 32], b[TBN / 32], kq;
    __device__ __forceinline__ void init(const LinArgs& p, int tid) {
        const unsigned rowb = (unsigned)p.K * 4;
        kq = (tid & 7) * 16;
#pragma unroll
        for (int i = 0; i < TBM / 32; ++i) a[i] = ((tid >> 3) + 32 * i) * rowb + kq;
#pragma unroll
        for (int i = 0; i < TBN / 32; ++i)
            b[i] = BT ? ((tid & 31) * (unsigned)p.ldb + (tid >> 5) * 4 + 32 * i) * 4u : ((tid >> 3) + 32 * i) * rowb + kq;
    }
};

// descriptors of the staged tile (A side: rebuilt when the tile changes) and of the whole-kernel operands
struct VecRsrc { rsrc_t A, A2, R, B, arg, gz, sc, sh, k2, mu; };

template <int AM, bool BT, bool RAG>
__device__ __forceinline__ void vec_rsrc_kernel(const LinArgs& p, int n0, VecRsrc& r) {
    const size_t rowb = (size_t)p.K * 4;
    r.B = BT ? buf_rsrc(p.B, (size_t)n0 * 4, (size_t)p.K * p.ldb * 4) : buf_rsrc(p.B, (size_t)n0 * rowb, (size_t)p.N * rowb);
    r.sc = buf_rsrc(p.sc, 0, AM != A_PLAIN ? rowb : 0); r.sh = buf_rsrc(p.sh, 0, AM != A_PLAIN ? rowb : 0);
    r.k2 = buf_rsrc(p.k2, 0, AM >= A_DY ? rowb : 0); r.mu = buf_rsrc(p.mu, 0, AM >= A_DY ? rowb : 0);
    // [groups][K] tables of the sparse max gradient: group ids come from the row records (or r / ns): always valid
    r.arg = buf_rsrc(p.arg, 0, AM == A_DY_SPARSE ? 0xffffffffull : 0); r.gz = buf_rsrc(p.gz, 0, AM == A_DY_SPARSE ? 0xffffffffull : 0);
}
template <int AM, bool RAG>
__device__ __forceinline__ void vec_rsrc_tile(const LinArgs& p, int m0, VecRsrc& r) {
    const size_t rowb = (size_t)p.K * 4;
    r.A = buf_rsrc(p.A, (size_t)m0 * rowb, AM != A_DY_SPARSE ? (size_t)p.M * rowb : 0);
    r.A2 = buf_rsrc(p.A2, (size_t)m0 * rowb, AM >= A_DY ? (size_t)p.M * rowb : 0);
}

template <int AM, int ROWS, bool RAG>
__device__ __forceinline__ void vload_a(const LinArgs& p, const VecRsrc& r, const unsigned (&voff)[ROWS / 32], unsigned kq, int m0, int k0,
                                        Stage<true, ROWS>& st, const RowInfo<true, ROWS>& ri) {
    st.m0 = m0;
    const unsigned kb = (unsigned)k0 * 4;
    if constexpr (AM != A_PLAIN) {
        st.c_sc = buf_ld4(r.sc, kb + kq, 0); st.c_sh = buf_ld4(r.sh, kb + kq, 0);
        if constexpr (AM >= A_DY) { st.c_k2 = buf_ld4(r.k2, kb + kq, 0); st.c_mu = buf_ld4(r.mu, kb + kq, 0); }
    }
    // k tail (K not a multiple of 32): a lane whose k quad lies past K must read zeros, not the next row
    const bool kin = kb + kq < (unsigned)p.K * 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const unsigned vo = kin ? voff[i] : BUF_OOB;
        if constexpr (AM != A_DY_SPARSE) st.v[i] = buf_ld4(r.A, vo, kb);
        if constexpr (AM >= A_DY) st.v2[i] = buf_ld4(r.A2, vo, kb);
        if constexpr (AM == A_DY_SPARSE) {
            const unsigned go = (unsigned)ri.g[i] * ((unsigned)p.K * 4) + kq;
            st.vi[i] = buf_ld4i(r.arg, go, kb);
            st.vg[i] = buf_ld4(r.gz, go, kb);
        }
    }
}

template <int ROWS, bool BT>
__device__ __forceinline__ void vload_b(const LinArgs& p, const VecRsrc& r, const unsigned (&voff)[ROWS / 32], unsigned kq, int k0,
                                        Stage<true, ROWS>& st) {
    if constexpr (BT) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) st.v[i] = buf_ld4(r.B, voff[i], (unsigned)k0 * (unsigned)p.ldb * 4u);     // k >= K: past the end -> 0
    } else {
        const unsigned kb = (unsigned)k0 * 4;
        const bool kin = kb + kq < (unsigned)p.K * 4;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) st.v[i] = buf_ld4(r.B, kin ? voff[i] : BUF_OOB, kb);
    }
}

// fused transform + LDS store, vector path: no masks (see the section header).  lrelu(t) = max(t, slope*t), 0 <= slope <= 1.
template <int AM, int ROWS, bool SP = false>
__device__ __forceinline__ void vstore_a(const LinArgs& p, float* sX, int tid, const Stage<true, ROWS>& st, const RowInfo<true, ROWS>& ri) {
    const float4 sc = st.c_sc, sh = st.c_sh, k2 = st.c_k2, mu = st.c_mu;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        float4 a;
        if constexpr (AM == A_PLAIN) {
            a = st.v[i];
        } else if constexpr (AM == A_BNACT) {
            const float4 x = st.v[i];
            const float tx = fmaf(sc.x, x.x, sh.x), ty = fmaf(sc.y, x.y, sh.y), tz = fmaf(sc.z, x.z, sh.z), tw = fmaf(sc.w, x.w, sh.w);
            a.x = fmaxf(tx, tx * p.slope); a.y = fmaxf(ty, ty * p.slope); a.z = fmaxf(tz, tz * p.slope); a.w = fmaxf(tw, tw * p.slope);
        } else {
            float4 du;
            if constexpr (AM == A_DY) du = st.v[i];
            else {
                const int srow = ri.rs[i];
                const int4 ar = st.vi[i];
                const float4 gz = st.vg[i];
                du.x = ar.x == srow ? gz.x : 0.f; du.y = ar.y == srow ? gz.y : 0.f;
                du.z = ar.z == srow ? gz.z : 0.f; du.w = ar.w == srow ? gz.w : 0.f;
            }
            const float4 y = st.v2[i];
            const float w = ri.rw[i];            // dense BatchNorm term counts once per duplicate
            a.x = fmaf(sc.x, du.x, -w * fmaf(k2.x, y.x - mu.x, sh.x)); a.y = fmaf(sc.y, du.y, -w * fmaf(k2.y, y.y - mu.y, sh.y));
            a.z = fmaf(sc.z, du.z, -w * fmaf(k2.z, y.z - mu.z, sh.z)); a.w = fmaf(sc.w, du.w, -w * fmaf(k2.w, y.w - mu.w, sh.w));
        }
        if constexpr (SP) {                    // three bf16 planes [plane][row][32 k]
            uint2 w0, w1, w2;
            split3(a.x, a.y, w0.x, w1.x, w2.x); split3(a.z, a.w, w0.y, w1.y, w2.y);
            unsigned char* q = reinterpret_cast<unsigned char*>(sX) + ((tid >> 3) + 32 * i) * PLD + (tid & 7) * 8;
            *reinterpret_cast<uint2*>(q) = w0; *reinterpret_cast<uint2*>(q + ROWS * PLD) = w1; *reinterpret_cast<uint2*>(q + 2 * ROWS * PLD) = w2;
        } else {
            *reinterpret_cast<float4*>(&sX[((tid >> 3) + 32 * i) * LDS_LD + (tid & 7) * 4]) = a;
        }
    }
}

template <int ROWS, bool BT, bool SP = false>
__device__ __forceinline__ void vstore_b(float* sX, int tid, const Stage<true, ROWS>& st) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const float4 v = st.v[i];
        if constexpr (SP) {
            unsigned char* base = reinterpret_cast<unsigned char*>(sX);
            if constexpr (BT) {          // W[k][4 n] -> planes [n][k]: 2-byte writes, lanes run over k
                uint2 w0, w1, w2;
                split3(v.x, v.y, w0.x, w1.x, w2.x); split3(v.z, v.w, w0.y, w1.y, w2.y);
                unsigned char* q = base + ((tid >> 5) * 4 + 32 * i) * PLD + (tid & 31) * 2;
                const uint2 ww[3] = {w0, w1, w2};
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    unsigned char* r = q + pl * ROWS * PLD;
                    *reinterpret_cast<unsigned short*>(r) = (unsigned short)(ww[pl].x & 0xffffu);
                    *reinterpret_cast<unsigned short*>(r + PLD) = (unsigned short)(ww[pl].x >> 16);
                    *reinterpret_cast<unsigned short*>(r + 2 * PLD) = (unsigned short)(ww[pl].y & 0xffffu);
                    *reinterpret_cast<unsigned short*>(r + 3 * PLD) = (unsigned short)(ww[pl].y >> 16);
                }
            } else {
                uint2 w0, w1, w2;
                split3(v.x, v.y, w0.x, w1.x, w2.x); split3(v.z, v.w, w0.y, w1.y, w2.y);
                unsigned char* q = base + ((tid >> 3) + 32 * i) * PLD + (tid & 7) * 8;
                *reinterpret_cast<uint2*>(q) = w0; *reinterpret_cast<uint2*>(q + ROWS * PLD) = w1; *reinterpret_cast<uint2*>(q + 2 * ROWS * PLD) = w2;
            }
            continue;
        }
        if constexpr (BT) {              // W[k][4 n] -> [n][k]: lanes run over k, the four dword writes are conflict-free
            const int rl = (tid >> 5) * 4 + 32 * i;
            sX[(rl + 0) * LDS_LD + (tid & 31)] = v.x; sX[(rl + 1) * LDS_LD + (tid & 31)] = v.y;
            sX[(rl + 2) * LDS_LD + (tid & 31)] = v.z; sX[(rl + 3) * LDS_LD + (tid & 31)] = v.w;
        } else {
            *reinterpret_cast<float4*>(&sX[((tid >> 3) + 32 * i) * LDS_LD + (tid & 7) * 4]) = v;
        }
    }
}

// All staging loads are UNCONDITIONAL with clamped (always valid) addresses: a load under a divergent branch
// makes hipcc drain vmcnt(0) at the branch join, which would serialise the prefetch against the MFMAs.
// Out-of-range elements are zeroed later, in the transform/store stage.
template <int AM, bool VEC, int ROWS, bool RAG>
__device__ __forceinline__ void load_a(const LinArgs& p, int m0, int k0, int tid, Stage<VEC, ROWS>& st, const RowInfo<VEC, ROWS>& ri) {
    st.m0 = m0;
    if constexpr (VEC) {
        const int k = min(k0 + (tid & 7) * 4, p.K - 4);
        if constexpr (AM != A_PLAIN) {
            st.c_sc = *reinterpret_cast<const float4*>(p.sc + k);
            st.c_sh = *reinterpret_cast<const float4*>(p.sh + k);
            if constexpr (AM >= A_DY) { st.c_k2 = *reinterpret_cast<const float4*>(p.k2 + k); st.c_mu = *reinterpret_cast<const float4*>(p.mu + k); }
        }
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int r = min(m0 + (tid >> 3) + 32 * i, p.M - 1);
            const size_t off = (size_t)r * p.K + k;
            const int g = ri.g[i];
            if constexpr (AM == A_PLAIN || AM == A_BNACT) {
                st.v[i] = *reinterpret_cast<const float4*>(p.A + off);
            } else if constexpr (AM == A_DY) {
                st.v[i] = *reinterpret_cast<const float4*>(p.A + off);
                st.v2[i] = *reinterpret_cast<const float4*>(p.A2 + off);
            } else {
                const size_t go = (size_t)g * p.K + k;
                st.vi[i] = *reinterpret_cast<const int4*>(p.arg + go);
                st.vg[i] = *reinterpret_cast<const float4*>(p.gz + go);
                st.v2[i] = *reinterpret_cast<const float4*>(p.A2 + off);
            }
        }
    } else {
        const int k = min(k0 + (tid & 31), p.K - 1);
        if constexpr (AM != A_PLAIN) {
            st.f_sc = p.sc[k]; st.f_sh = p.sh[k];
            if constexpr (AM >= A_DY) { st.f_k2 = p.k2[k]; st.f_mu = p.mu[k]; }
        }
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int r = min(m0 + (tid >> 5) + 8 * i, p.M - 1);
            const size_t off = (size_t)r * p.K + k;
            const int g = ri.g[i];
            if constexpr (AM == A_PLAIN || AM == A_BNACT) {
                st.s[i] = p.A[off];
            } else if constexpr (AM == A_DY) {
                st.s[i] = p.A[off];
                st.s2[i] = p.A2[off];
            } else {
                const size_t go = (size_t)g * p.K + k;
                st.si[i] = p.arg[go];
                st.sg[i] = p.gz[go];
                st.s2[i] = p.A2[off];
            }
        }
    }
}

// fused transform + LDS store of a staged A tile.  Rows/columns outside the matrix were loaded as zeros and
// must stay exactly zero (they feed MFMAs whose results are discarded or masked, but k-padding feeds real
// outputs), so the affine modes are applied only to in-range elements.
template <int AM, bool VEC, int ROWS>
__device__ __forceinline__ void store_a(const LinArgs& p, float* sX, int k0, int tid, const Stage<VEC, ROWS>& st, const RowInfo<VEC, ROWS>& ri) {
    if constexpr (VEC) {
        const bool kin = k0 + (tid & 7) * 4 < p.K;
        const float4 sc = st.c_sc, sh = st.c_sh, k2 = st.c_k2, mu = st.c_mu;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int rl = (tid >> 3) + 32 * i;
            const int r = st.m0 + rl;
            float4 a = make_float4(0, 0, 0, 0);
            if (kin && r < p.M) {
                if constexpr (AM == A_PLAIN) {
                    a = st.v[i];
                } else if constexpr (AM == A_BNACT) {
                    const float4 x = st.v[i];
                    a.x = lrelu(fmaf(sc.x, x.x, sh.x), p.slope); a.y = lrelu(fmaf(sc.y, x.y, sh.y), p.slope);
                    a.z = lrelu(fmaf(sc.z, x.z, sh.z), p.slope); a.w = lrelu(fmaf(sc.w, x.w, sh.w), p.slope);
                } else {
                    float4 du;
                    if constexpr (AM == A_DY) du = st.v[i];
                    else {
                        const int srow = ri.rs[i];
                        const int4 ar = st.vi[i];
                        const float4 gz = st.vg[i];
                        du.x = ar.x == srow ? gz.x : 0.f; du.y = ar.y == srow ? gz.y : 0.f;
                        du.z = ar.z == srow ? gz.z : 0.f; du.w = ar.w == srow ? gz.w : 0.f;
                    }
                    const float4 y = st.v2[i];
                    const float w = ri.rw[i];            // dense BatchNorm term counts once per duplicate
                    a.x = fmaf(sc.x, du.x, -w * fmaf(k2.x, y.x - mu.x, sh.x)); a.y = fmaf(sc.y, du.y, -w * fmaf(k2.y, y.y - mu.y, sh.y));
                    a.z = fmaf(sc.z, du.z, -w * fmaf(k2.z, y.z - mu.z, sh.z)); a.w = fmaf(sc.w, du.w, -w * fmaf(k2.w, y.w - mu.w, sh.w));
                }
            }
            *reinterpret_cast<float4*>(&sX[rl * LDS_LD + (tid & 7) * 4]) = a;
        }
    } else {
        const bool kin = k0 + (tid & 31) < p.K;
        const float sc = st.f_sc, sh = st.f_sh, k2 = st.f_k2, mu = st.f_mu;
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int rl = (tid >> 5) + 8 * i;
            const int r = st.m0 + rl;
            float a = 0.f;
            if (kin && r < p.M) {
                if constexpr (AM == A_PLAIN) a = st.s[i];
                else if constexpr (AM == A_BNACT) a = lrelu(fmaf(sc, st.s[i], sh), p.slope);
                else {
                    float du;
                    if constexpr (AM == A_DY) du = st.s[i];
                    else du = st.si[i] == ri.rs[i] ? st.sg[i] : 0.f;
                    a = fmaf(sc, du, -ri.rw[i] * fmaf(k2, st.s2[i] - mu, sh));
                }
            }
            sX[rl * LDS_LD + (tid & 31)] = a;
        }
    }
}

// BT: the B operand is given K-major, B[k][n] with row stride ldb (the dX GEMMs consume the layer's weight W[Cout][Cin]
// as it is stored instead of a transposed copy); a lane then reads 4 consecutive n for one k and the tile is transposed
// on its way into LDS (lanes run over k, so the four scalar LDS writes are conflict-free).
template <bool VEC, int ROWS, bool BT>
__device__ __forceinline__ void load_b(const LinArgs& p, int n0, int k0, int tid, Stage<VEC, ROWS>& st) {
    if constexpr (BT && VEC) {
        const int k = min(k0 + (tid & 31), p.K - 1);
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int n = min(n0 + (tid >> 5) * 4 + 32 * i, p.ldb - 4);
            st.v[i] = *reinterpret_cast<const float4*>(p.B + (size_t)k * p.ldb + n);
        }
    } else if constexpr (BT) {
        const int k = min(k0 + (tid & 31), p.K - 1);
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int n = min(n0 + (tid >> 5) + 8 * i, p.N - 1);
            st.s[i] = p.B[(size_t)k * p.ldb + n];
        }
    } else if constexpr (VEC) {
        const int k = min(k0 + (tid & 7) * 4, p.K - 4);
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int n = min(n0 + (tid >> 3) + 32 * i, p.N - 1);
            st.v[i] = *reinterpret_cast<const float4*>(p.B + (size_t)n * p.K + k);
        }
    } else {
        const int k = min(k0 + (tid & 31), p.K - 1);
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int n = min(n0 + (tid >> 5) + 8 * i, p.N - 1);
            st.s[i] = p.B[(size_t)n * p.K + k];
        }
    }
}

template <bool VEC, int ROWS, bool BT>
__device__ __forceinline__ void store_b(const LinArgs& p, float* sX, int n0, int k0, int tid, const Stage<VEC, ROWS>& st) {
    if constexpr (BT && VEC) {
        const bool kin = k0 + (tid & 31) < p.K;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int rl = (tid >> 5) * 4 + 32 * i;
            const float4 v = st.v[i];
            sX[(rl + 0) * LDS_LD + (tid & 31)] = (kin && n0 + rl + 0 < p.N) ? v.x : 0.f;
            sX[(rl + 1) * LDS_LD + (tid & 31)] = (kin && n0 + rl + 1 < p.N) ? v.y : 0.f;
            sX[(rl + 2) * LDS_LD + (tid & 31)] = (kin && n0 + rl + 2 < p.N) ? v.z : 0.f;
            sX[(rl + 3) * LDS_LD + (tid & 31)] = (kin && n0 + rl + 3 < p.N) ? v.w : 0.f;
        }
    } else if constexpr (VEC) {
        const bool kin = k0 + (tid & 7) * 4 < p.K;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int rl = (tid >> 3) + 32 * i;
            const bool in = kin && n0 + rl < p.N;
            *reinterpret_cast<float4*>(&sX[rl * LDS_LD + (tid & 7) * 4]) = in ? st.v[i] : make_float4(0, 0, 0, 0);
        }
    } else {
        const bool kin = k0 + (tid & 31) < p.K;
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int rl = (tid >> 5) + 8 * i;
            sX[rl * LDS_LD + (tid & 31)] = (kin && n0 + rl < p.N) ? st.s[i] : 0.f;
        }
    }
}

// C[M,N] = A'[M,K] * B[N,K]^T with fused A transform (AM) and epilogue (EM).
// 2 x 2 waves, each owning a 64 x (32*TN) tile: block tile 128 x (64*TN); TN = 1 for N <= 64.
// GM (E_STORE_STATS only): 0 = off; 32 / 64 = also emit per-group (ns = GM rows) max/min/argmax/argmin of the
// raw outputs -- BatchNorm's scale is not known yet, so both extremes are kept and the tiny finalize kernel
// picks max for scale >= 0 and min for scale < 0 (the activation is monotone).

// In-place 4x4 transpose across the four lanes of a quad: afterwards register j of lane i holds what register i of
// lane j held.  The MFMA C/D layout gives a lane ONE column and four consecutive rows per register group; transposed,
// a lane holds four consecutive columns of ONE row, i.e. a 16-byte piece of a C row -> global_store_dwordx4 on whole
// 128-byte lines instead of four dword stores (the epilogue is store-issue bound otherwise).
__device__ __forceinline__ float quad_xchg1(float x) { return __uint_as_float(dpp_u32<0xB1>(__float_as_uint(x), __float_as_uint(x))); }
__device__ __forceinline__ float quad_xchg2(float x) { return __uint_as_float(dpp_u32<0x4E>(__float_as_uint(x), __float_as_uint(x))); }
__device__ __forceinline__ void quad_transpose4(float& v0, float& v1, float& v2, float& v3, bool b0, bool b1) {
    float y;
    y = quad_xchg1(b0 ? v0 : v1); v0 = b0 ? y : v0; v1 = b0 ? v1 : y;
    y = quad_xchg1(b0 ? v2 : v3); v2 = b0 ? y : v2; v3 = b0 ? v3 : y;
    y = quad_xchg2(b1 ? v0 : v2); v0 = b1 ? y : v0; v2 = b1 ? v2 : y;
    y = quad_xchg2(b1 ? v1 : v3); v1 = b1 ? y : v1; v3 = b1 ? v3 : y;
}

// (`block`: the workgroup's id in the kernel's logical grid -- blockIdx.x for the plain launch; linear_bwd_pair_kernel hands the ids
//  behind its dW workgroups to this body)
template <int AM, int EM, bool VEC, int TN, int GM, bool RAG, int TM = 2, bool SP = false>
__device__ __forceinline__ void linear_nt_body(const LinArgs& p_in, const int block) {
    static_assert(GM == 0 || TM == 2, "the fused group max works on 64-row wave slabs");
    static_assert(!SP || (VEC && GM == 0), "bf16-plane operands: vector path only");
    LinArgs p = p_in;
    if (p.m_dev) p.M = *p.m_dev;                       // compacted rows: the row count lives on the device
    constexpr int TBM = 64 * TM, TBN = 64 * TN;        // TM = 1: 64-row tiles for small M (twice the workgroups)
    constexpr bool BT = AM >= A_DY;              // backward: B is the weight matrix as stored, [K][N]
    __shared__ __attribute__((aligned(16))) float sA[SP ? 3 * TBM * PLD / 4 : TBM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float sB[SP ? 3 * TBN * PLD / 4 : TBN * LDS_LD];
    // multiplicities of the tile's rows for the BatchNorm sums of compacted rows: prefetched with the operands and kept
    // in LDS (a global load in the epilogue would sit behind the tile's own stores in the in-order vmcnt queue)
    constexpr bool NEEDW = RAG && EM == E_STORE_STATS;
    __shared__ __attribute__((aligned(16))) float sW[NEEDW ? TBM : 4];
    int wmeta = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int bx, by;
    tile_of_block(block, p.gx, p.nt, bx, by);
    const int n0 = p.n_begin + by * TBN;
    const int m_tiles = (p.M + TBM - 1) / TBM;
    const int lr = lane & 31, lh = lane >> 5;
    // rows of C (and Yprev) can be moved as 16-byte pieces
    // (not for the masked epilogue: moving Yprev the same way costs two transposes per register group and measured slower)
    const bool vec_c = EM != E_MASK_STORE_STATS && (p.ldc % 4 == 0) && (p.n_begin % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);

    double st_s[TN], st_q[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { st_s[t] = 0.0; st_q[t] = 0.0; }
    float ep_bias[TN], ep_sc[TN], ep_sh[TN];        // per-column epilogue constants (this lane's columns)
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = min(n0 + wc * 32 * TN + t * 32 + lr, p.N - 1);
        ep_bias[t] = p.bias ? p.bias[col] : 0.f;
        ep_sc[t] = EM == E_MASK_STORE_STATS ? p.esc[col] : 0.f;
        ep_sh[t] = EM == E_MASK_STORE_STATS ? p.esh[col] : 0.f;
    }

    Stage<VEC, TBM> ra;
    Stage<VEC, TBN> rb;
    RowInfo<VEC, TBM> ri;
    VecOff<TBM, TBN, BT> vo;
    VecRsrc vr;
    if constexpr (VEC) { vo.init(p, tid); vec_rsrc_kernel<AM, BT, RAG>(p, n0, vr); }
    // row records of a tile through its own descriptor (vector path): rows past M read 0 = {group 0, row 0, multiplicity 0}
    auto vec_fetch_rows = [&](int m0) {
        if constexpr (VEC && RAG && (AM >= A_DY || NEEDW)) {
            vr.R = buf_rsrc(p.rmeta, (size_t)m0 * 8, (size_t)p.M * 8);
            if constexpr (AM >= A_DY) {
#pragma unroll
                for (int i = 0; i < TBM / 32; ++i) ri.raw[i] = buf_ld2i(vr.R, ((tid >> 3) + 32 * i) * 8);
            }
        }
    };
    // request the operands of step (tile mt_, k block k0_); `newtile`: the staged tile changes with this step
    auto request = [&](int mt_, int k0_, bool newtile, bool first) {
        const int m0_ = mt_ * TBM;
        if constexpr (VEC) {
            if (newtile) {
                if (first) vec_fetch_rows(m0_);
                rowinfo_adopt<AM, VEC, TBM, RAG>(p, m0_, tid, ri);        // the records fetched one tile ahead
                if constexpr (NEEDW) wmeta = buf_ld2i(vr.R, (tid & (TBM - 1)) * 8).y;   // (vr.R still describes this tile)
                vec_rsrc_tile<AM, RAG>(p, m0_, vr);
            }
            vload_a<AM, TBM, RAG>(p, vr, vo.a, vo.kq, m0_, k0_, ra, ri);
            vload_b<TBN, BT>(p, vr, vo.b, vo.kq, k0_, rb);
            if (newtile) vec_fetch_rows((mt_ + p.gx) * TBM);              // after this step's loads: nothing waits for them early
        } else {
            if (newtile) {
                if (first) rowinfo_fetch<AM, VEC, TBM, RAG>(p, m0_, tid, ri);
                rowinfo_adopt<AM, VEC, TBM, RAG>(p, m0_, tid, ri);
                rowinfo_fetch<AM, VEC, TBM, RAG>(p, (mt_ + p.gx) * TBM, tid, ri);      // clamped addresses: always valid
                if constexpr (NEEDW) wmeta = p.rmeta[min(m0_ + (tid & (TBM - 1)), p.M - 1)].y;
            }
            load_a<AM, VEC, TBM, RAG>(p, m0_, k0_, tid, ra, ri);
            load_b<VEC, TBN, BT>(p, n0, k0_, tid, rb);
        }
    };
    int mt = bx;
    int staged_k0 = 0;
    if (mt < m_tiles) request(mt, 0, true, true);
    for (; mt < m_tiles; mt += p.gx) {
        const int m0 = mt * TBM;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int k0 = 0; k0 < p.K; k0 += BK) {
            __syncthreads();                       // previous step's fragment reads are done
            if constexpr (VEC) {
                vstore_a<AM, TBM, SP>(p, sA, tid, ra, ri);
                vstore_b<TBN, BT, SP>(sB, tid, rb);
            } else {
                store_a<AM, VEC, TBM>(p, sA, staged_k0, tid, ra, ri);
                store_b<VEC, TBN, BT>(p, sB, n0, staged_k0, tid, rb);
            }
            if constexpr (NEEDW) { if (tid < TBM) sW[tid] = (float)(wmeta >> 16); }
            __syncthreads();
            // request the next step's operands (next k block, or the first k block of this workgroup's next tile)
            {
                int nk = k0 + BK, nmt = mt;
                if (nk >= p.K) { nk = 0; nmt = mt + p.gx; }
                staged_k0 = nk;
                if (nmt < m_tiles) request(nmt, nk, nk == 0, false);     // a new tile adopts its prefetched row records
            }
            const int kc = min(BK, p.K - k0);
            if constexpr (SP) {                // nine bf16 MFMAs per 16 k and 32x32 tile (k past K was staged as zeros)
                const unsigned char* pa = reinterpret_cast<const unsigned char*>(sA) + (wr * (32 * TM) + lr) * PLD + lh * 16;
                const unsigned char* pb = reinterpret_cast<const unsigned char*>(sB) + (wc * 32 * TN + lr) * PLD + lh * 16;
                const int nks = (kc + 15) >> 4;
                for (int ks = 0; ks < nks; ++ks) {
                    bf16x8 af[TM][3], bq[TN][3];
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) af[t][pl] = *reinterpret_cast<const bf16x8*>(pa + pl * TBM * PLD + t * 32 * PLD + ks * 32);
#pragma unroll
                    for (int t = 0; t < TN; ++t)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) bq[t][pl] = *reinterpret_cast<const bf16x8*>(pb + pl * TBN * PLD + t * 32 * PLD + ks * 32);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_split9(af[tm], bq[tn], acc[tm][tn]);
                }
            } else {
            const int nkk = (kc + 7) >> 3;
            for (int kk = 0; kk < nkk; ++kk) {
                float4 a4[TM], b4[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    a4[t] = *reinterpret_cast<const float4*>(&sA[(wr * (32 * TM) + t * 32 + lr) * LDS_LD + kk * 8 + lh * 4]);
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    b4[t] = *reinterpret_cast<const float4*>(&sB[(wc * 32 * TN + t * 32 + lr) * LDS_LD + kk * 8 + lh * 4]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        const float av = s == 0 ? a4[tm].x : s == 1 ? a4[tm].y : s == 2 ? a4[tm].z : a4[tm].w;
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) {
                            const float bv = s == 0 ? b4[tn].x : s == 1 ? b4[tn].y : s == 2 ? b4[tn].z : b4[tn].w;
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tm][tn], 0, 0, 0);
                        }
                    }
                }
            }
            }
        }

        // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        // Full tiles take a branch-free path (a store or load under a per-element branch makes hipcc wait
        // vmcnt(0) at every join, serialising the 64 stores of a lane).
        // (decided per 32-column block of a wave, not per tile: N = 96 -- the MSG stacks' 64->96->128 layers -- leaves the last
        //  block of the last column tile empty, and taking the whole tile down the per-element path cost 0.77 ms per launch)
        const bool full_rows = m0 + TBM <= p.M;
        float tw_all = 0.f;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int cb0 = n0 + wc * 32 * TN + tn * 32;        // this wave's 32-column block (wave-uniform)
            if (cb0 >= p.N) continue;                           // entirely past N: nothing to store, sums stay 0
            const bool full = full_rows && (cb0 + 32 <= p.N);
            const int col = cb0 + lr;
            const bool cin = col < p.N;
            const float bias = ep_bias[tn], esc = ep_sc[tn], esh = ep_sh[tn];
            float ts = 0.f, tq = 0.f, tw = 0.f, piv = 0.f;
            if (full && VEC) {
                // Vector path: the tile leaves (and Yprev arrives) in the MFMA C/D layout as dword buffer accesses -- a lane's 16
                // values of a 32x32 tile sit in one column, a wave instruction covers two full 128-byte row segments -- with the
                // row of register r in the SCALAR offset: no address arithmetic and no quad transposes on the vector ALU,
                // which is the matrix pipe here (section header of the vector path).  16 dword stores per 32x32 tile instead
                // of 4 x 16-byte stores + 64 VALU: the stores queue beside the other wave's MFMAs, the VALU work would not.
                const unsigned ldcb = (unsigned)p.ldc * 4;
                const rsrc_t rC = buf_rsrc(p.C, (size_t)m0 * ldcb, (size_t)p.M * ldcb);
                const unsigned vC = (unsigned)(wr * (32 * TM) + 4 * lh) * ldcb + (unsigned)col * 4;
                float yv[EM == E_MASK_STORE_STATS ? TM : 1][16];
                if constexpr (EM == E_MASK_STORE_STATS) {          // (Yprev has the row stride of C there)
                    const rsrc_t rY = buf_rsrc(p.Yprev, (size_t)m0 * ldcb, (size_t)p.M * ldcb);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) yv[tm][r] = buf_ld1(rY, vC, (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * ldcb);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    if (tm == 0) piv = EM == E_MASK_STORE_STATS ? yv[0][0] : acc[0][tn][0] + bias;
                    float wv[16];
                    if constexpr (NEEDW) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 t = *reinterpret_cast<const float4*>(&sW[wr * (32 * TM) + tm * 32 + 8 * q + 4 * lh]);
                            wv[4 * q] = t.x; wv[4 * q + 1] = t.y; wv[4 * q + 2] = t.z; wv[4 * q + 3] = t.w;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float c = acc[tm][tn][r] + bias;
                        if constexpr (EM == E_MASK_STORE_STATS) {
                            const float y = yv[tm][r];
                            c = fmaf(esc, y, esh) > 0.f ? c : c * p.eslope;
                            ts += c; tq = fmaf(c, y - piv, tq);
                        } else if constexpr (EM == E_STORE_STATS) {
                            const float d = c - piv;
                            if constexpr (RAG) { const float w = wv[r]; if (tn == 0) tw += w; ts = fmaf(w, d, ts); tq = fmaf(w * d, d, tq); }
                            else { ts += d; tq = fmaf(d, d, tq); }
                        }
                        buf_st1(rC, vC, (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * ldcb, c);
                    }
                }
                if constexpr (RAG && EM == E_STORE_STATS) { if (tn == 0) tw_all = tw; else tw = tw_all; }     // sum of multiplicities: same for every column
                if constexpr (EM == E_MASK_STORE_STATS) {
                    st_s[tn] += (double)ts; st_q[tn] += (double)tq + (double)piv * (double)ts;
                } else if constexpr (EM == E_STORE_STATS) {
                    const double n = RAG ? (double)tw : 16.0 * TM, pv = piv;
                    st_s[tn] += (double)ts + n * pv; st_q[tn] += (double)tq + 2.0 * pv * (double)ts + n * pv * pv;
                }
            } else if (full) {
                const bool b0 = lane & 1, b1 = lane & 2;
                const int qcol = n0 + wc * 32 * TN + tn * 32 + (lr & ~3);      // first of this lane's 4 columns after the transpose
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int rbase = m0 + wr * (32 * TM) + tm * 32 + 4 * lh + (lane & 3);   // + 8*(r>>2): this lane's row after the transpose
                    float yv[16];                          // one 32-row MFMA tile at a time: 16 VGPRs
                    if constexpr (EM == E_MASK_STORE_STATS) {
                        if (vec_c) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 t = *reinterpret_cast<const float4*>(p.Yprev + (size_t)(rbase + 8 * q) * p.N + qcol);
                                yv[4 * q] = t.x; yv[4 * q + 1] = t.y; yv[4 * q + 2] = t.z; yv[4 * q + 3] = t.w;
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) quad_transpose4(yv[4 * q], yv[4 * q + 1], yv[4 * q + 2], yv[4 * q + 3], b0, b1);
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = m0 + wr * (32 * TM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                yv[r] = p.Yprev[(size_t)row * p.N + col];
                            }
                        }
                    }
                    if (tm == 0) piv = EM == E_MASK_STORE_STATS ? yv[0] : acc[0][tn][0] + bias;
                    float cv[16];
                    float wv[16];
                    if constexpr (NEEDW) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 t = *reinterpret_cast<const float4*>(&sW[wr * (32 * TM) + tm * 32 + 8 * q + 4 * lh]);
                            wv[4 * q] = t.x; wv[4 * q + 1] = t.y; wv[4 * q + 2] = t.z; wv[4 * q + 3] = t.w;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float c = acc[tm][tn][r] + bias;
                        if constexpr (EM == E_MASK_STORE_STATS) {
                            const float y = yv[r];
                            c = fmaf(esc, y, esh) > 0.f ? c : c * p.eslope;
                            ts += c; tq = fmaf(c, y - piv, tq);
                        } else if constexpr (EM == E_STORE_STATS) {
                            const float d = c - piv;
                            if constexpr (RAG) { const float w = wv[r]; tw += w; ts = fmaf(w, d, ts); tq = fmaf(w * d, d, tq); }
                            else { ts += d; tq = fmaf(d, d, tq); }
                        }
                        cv[r] = c;
                    }
                    if (vec_c) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            quad_transpose4(cv[4 * q], cv[4 * q + 1], cv[4 * q + 2], cv[4 * q + 3], b0, b1);
                            *reinterpret_cast<float4*>(p.C + (size_t)(rbase + 8 * q) * p.ldc + qcol) =
                                make_float4(cv[4 * q], cv[4 * q + 1], cv[4 * q + 2], cv[4 * q + 3]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = m0 + wr * (32 * TM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            p.C[(size_t)row * p.ldc + col] = cv[r];
                        }
                    }
                }
                // Sums were taken about a pivot (this lane's first value of the column) so that fp32 accumulation
                // loses nothing when |mean| >> std; undo the shift in fp64.
                if constexpr (EM == E_MASK_STORE_STATS) {
                    st_s[tn] += (double)ts; st_q[tn] += (double)tq + (double)piv * (double)ts;
                } else if constexpr (EM == E_STORE_STATS) {
                    const double n = RAG ? (double)tw : 16.0 * TM, pv = piv;
                    st_s[tn] += (double)ts + n * pv; st_q[tn] += (double)tq + 2.0 * pv * (double)ts + n * pv * pv;
                }
            } else {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wr * (32 * TM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (cin && row < p.M) {
                            const size_t off = (size_t)row * p.N + col;
                            float c = acc[tm][tn][r] + bias;
                            if constexpr (EM == E_MASK_STORE_STATS) {
                                const float y = p.Yprev[off];
                                c = fmaf(esc, y, esh) > 0.f ? c : c * p.eslope;
                                st_s[tn] += (double)c; st_q[tn] += (double)c * (double)y;
                            } else if constexpr (EM == E_STORE_STATS) {
                                const double w = RAG ? (double)(p.rmeta[row].y >> 16) : 1.0;
                                st_s[tn] += w * (double)c; st_q[tn] += w * (double)c * (double)c;
                            }
                            p.C[(size_t)row * p.ldc + col] = c;
                        }
                    }
            }
            if constexpr (GM != 0) {
                // rows of this lane, ascending: s = tm*32 + (r&3) + 8*(r>>2) + 4*lh; a group is GM consecutive rows
                constexpr int NG = 64 / GM;                      // groups per 64-row wave slab (1 or 2)
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    float vmax = -INFINITY, vmin = INFINITY;
                    int imax = 0, imin = 0;
#pragma unroll
                    for (int tm = gi * (2 / NG); tm < (gi + 1) * (2 / NG); ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int sl = (tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) - gi * GM;   // row in group
                            const float c = acc[tm][tn][r] + bias;
                            if (c > vmax) { vmax = c; imax = sl; }
                            if (c < vmin) { vmin = c; imin = sl; }
                        }
                    const float omax = __shfl_xor(vmax, 32), omin = __shfl_xor(vmin, 32);
                    const int oimax = __shfl_xor(imax, 32), oimin = __shfl_xor(imin, 32);
                    if (omax > vmax || (omax == vmax && oimax < imax)) { vmax = omax; imax = oimax; }
                    if (omin < vmin || (omin == vmin && oimin < imin)) { vmin = omin; imin = oimin; }
                    const int row0 = m0 + wr * (32 * TM) + gi * GM;
                    if (lh == 0 && cin && row0 < p.M) {
                        const size_t o = (size_t)(row0 / GM) * p.N + col;
                        p.gmax[o] = vmax; p.gmin[o] = vmin; p.gamax[o] = imax; p.gamin[o] = imin;
                    }
                }
            }
        }
    }

    if constexpr (EM != E_STORE) {
        // lanes l and l^32 hold the same column: fold, then combine the two waves (wr = 0,1) sharing the columns
        __syncthreads();
        double* red = reinterpret_cast<double*>(sA);           // [4 waves][TN][32][2]
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            double s = st_s[tn], q = st_q[tn];
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (lh == 0) { red[((wave * TN + tn) * 32 + lr) * 2 + 0] = s; red[((wave * TN + tn) * 32 + lr) * 2 + 1] = q; }
        }
        __syncthreads();
        if (tid < TBN) {
            const int c = tid, wcc = c / (32 * TN), tn = (c >> 5) % TN, l = c & 31;
            const int col = n0 + c;
            if (col < p.N) {
                double s = 0.0, q = 0.0;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int wv = w * 2 + wcc;
                    s += red[((wv * TN + tn) * 32 + l) * 2 + 0]; q += red[((wv * TN + tn) * 32 + l) * 2 + 1];
                }
                double* dst = p.stats + (size_t)bx * 2 * p.N;
                dst[col] = s; dst[p.N + col] = q;
            }
        }
    }
}
template <int AM, int EM, bool VEC, int TN, int GM, bool RAG, int TM = 2, bool SP = false>
__global__ __launch_bounds__(MLP_T, 2) void linear_nt_kernel(const LinArgs p_in) {
    linear_nt_body<AM, EM, VEC, TN, GM, RAG, TM, SP>(p_in, (int)blockIdx.x);
}

// ---- weight gradient: dW[I,J] = sum_p A'[p,I] * B'[p,J]  (A' = dy of this layer, B' = activation below) ----
struct DwArgs {
    const float* A; const float* A2;           // dU / Y of this layer            [P,I]
    const float* sc; const float* sh; const float* k2; const float* mu;     // a, k1, k2, mean   [I]
    const int32_t* arg; const float* gz; int ns;            // sparse max-pool gradient [P/ns, I]
    const float* Bsrc; const float* bsc; const float* bsh;  // Y_prev (or X) [P,J], folded BN of the layer below
    float bslope;
    float* part;                                // [gridDim.x][I][J] partial sums
    const int* p_dev;                           // optional device-resident row count (compacted rows)
    const int2* rmeta;                          // optional per-row meta of compacted rows (see row_meta)
    int P, I, J;
    int a_mode, b_mode;                         // a: A_DY / A_DY_SPARSE ; b: A_PLAIN / A_BNACT
    int gx, ti, tj;                             // logical grid: gx row-chunk workgroups x (ti x tj) output tiles
    int ldo;                                    // row stride of an output tile in `part` (J, or dW's own when gx == 1)
};

constexpr int DW_BP = 32;

// Staged [32 rows][W channels] operand tile of the dW GEMM.  Vector path: thread t owns channels
// c4 = (t % (W/4))*4 of rows t/(W/4) + RP*i; scalar path: channel t % W of rows t/W + RP*i.
template <bool VEC, int W>
struct DwStage {
    static constexpr int CPR = VEC ? W / 4 : W;          // threads per row
    static constexpr int RP = MLP_T / CPR;               // rows per pass
    static constexpr int NI = DW_BP / RP;
    float4 v[VEC ? NI : 1], v2[VEC ? NI : 1], vg[VEC ? NI : 1];
    int4 vi[VEC ? NI : 1];
    float s[VEC ? 1 : NI], s2[VEC ? 1 : NI], sg[VEC ? 1 : NI];
    int si[VEC ? 1 : NI];
    float rw[NI];
    int rs[NI];
    int2 raw[NI];                  // row records of the chunk AFTER the staged one (prefetched: the sparse loader needs
                                   // the group id before it can address (arg, gz) -- see RowInfo above)
    int p0;
};

template <bool VEC, int W, bool RAG>
__device__ __forceinline__ void dw_fetch_meta(const DwArgs& p, int p0, int tid, DwStage<VEC, W>& st) {
    using S = DwStage<VEC, W>;
    if constexpr (RAG) {
#pragma unroll
        for (int i = 0; i < S::NI; ++i) st.raw[i] = p.rmeta[min(p0 + tid / S::CPR + S::RP * i, p.P - 1)];
    }
}

template <int AM, bool VEC, int W, bool RAG>
__device__ __forceinline__ void dw_load_a(const DwArgs& p, int p0, int c0, int tid, DwStage<VEC, W>& st) {
    using S = DwStage<VEC, W>;
    st.p0 = p0;
    const int c = min(c0 + (tid % S::CPR) * (VEC ? 4 : 1), p.I - (VEC ? 4 : 1));     // clamped, always valid
#pragma unroll
    for (int i = 0; i < S::NI; ++i) {
        const int r = min(p0 + tid / S::CPR + S::RP * i, p.P - 1);
        const size_t off = (size_t)r * p.I + c;
        int g = 0, srow = 0;
        float w = 1.f;
        if constexpr (RAG) row_meta(st.raw[i], g, srow, w);          // records fetched one chunk ahead (dw_fetch_meta)
        else if constexpr (AM == A_DY_SPARSE) { g = r / p.ns; srow = r - g * p.ns; }
        st.rw[i] = w; st.rs[i] = srow;
        if constexpr (AM == A_PLAIN) {                  // dy already formed (few-row layers: bn_bwd_dy_kernel)
            if constexpr (VEC) st.v[i] = *reinterpret_cast<const float4*>(p.A + off);
            else st.s[i] = p.A[off];
        } else if constexpr (VEC) {
            st.v2[i] = *reinterpret_cast<const float4*>(p.A2 + off);
            if constexpr (AM == A_DY) st.v[i] = *reinterpret_cast<const float4*>(p.A + off);
            else {
                const size_t go = (size_t)g * p.I + c;
                st.vi[i] = *reinterpret_cast<const int4*>(p.arg + go);
                st.vg[i] = *reinterpret_cast<const float4*>(p.gz + go);
            }
        } else {
            st.s2[i] = p.A2[off];
            if constexpr (AM == A_DY) st.s[i] = p.A[off];
            else {
                const size_t go = (size_t)g * p.I + c;
                st.si[i] = p.arg[go];
                st.sg[i] = p.gz[go];
            }
        }
    }
}

template <int AM, bool VEC, int W>
__device__ __forceinline__ void dw_store_a(const DwArgs& p, float* sX, int c0, int tid, const DwStage<VEC, W>& st,
                                           float4 sc, float4 sh, float4 k2, float4 mu) {
    using S = DwStage<VEC, W>;
    const int cl = (tid % S::CPR) * (VEC ? 4 : 1);
    const bool cin = c0 + cl < p.I;
#pragma unroll
    for (int i = 0; i < S::NI; ++i) {
        const int rl = tid / S::CPR + S::RP * i;
        const int r = st.p0 + rl;
        const bool in = cin && r < p.P;
        if constexpr (AM == A_PLAIN) {
            if constexpr (VEC) *reinterpret_cast<float4*>(&sX[rl * (W + 4) + cl]) = in ? st.v[i] : make_float4(0, 0, 0, 0);
            else sX[rl * (W + 4) + cl] = in ? st.s[i] : 0.f;
        } else if constexpr (VEC) {
            float4 a = make_float4(0, 0, 0, 0);
            if (in) {
                float4 du;
                if constexpr (AM == A_DY) du = st.v[i];
                else {
                    const int srow = st.rs[i];
                    const int4 ar = st.vi[i];
                    const float4 gz = st.vg[i];
                    du.x = ar.x == srow ? gz.x : 0.f; du.y = ar.y == srow ? gz.y : 0.f;
                    du.z = ar.z == srow ? gz.z : 0.f; du.w = ar.w == srow ? gz.w : 0.f;
                }
                const float4 y = st.v2[i];
                const float w = st.rw[i];
                a.x = fmaf(sc.x, du.x, -w * fmaf(k2.x, y.x - mu.x, sh.x)); a.y = fmaf(sc.y, du.y, -w * fmaf(k2.y, y.y - mu.y, sh.y));
                a.z = fmaf(sc.z, du.z, -w * fmaf(k2.z, y.z - mu.z, sh.z)); a.w = fmaf(sc.w, du.w, -w * fmaf(k2.w, y.w - mu.w, sh.w));
            }
            *reinterpret_cast<float4*>(&sX[rl * (W + 4) + cl]) = a;
        } else {
            float a = 0.f;
            if (in) {
                float du;
                if constexpr (AM == A_DY) du = st.s[i];
                else du = st.si[i] == st.rs[i] ? st.sg[i] : 0.f;
                a = fmaf(sc.x, du, -st.rw[i] * fmaf(k2.x, st.s2[i] - mu.x, sh.x));
            }
            sX[rl * (W + 4) + cl] = a;
        }
    }
}

template <bool VEC, int W>
__device__ __forceinline__ void dw_load_b(const DwArgs& p, int p0, int c0, int tid, DwStage<VEC, W>& st) {
    using S = DwStage<VEC, W>;
    st.p0 = p0;
    const int c = min(c0 + (tid % S::CPR) * (VEC ? 4 : 1), p.J - (VEC ? 4 : 1));
#pragma unroll
    for (int i = 0; i < S::NI; ++i) {
        const int r = min(p0 + tid / S::CPR + S::RP * i, p.P - 1);
        const size_t off = (size_t)r * p.J + c;
        if constexpr (VEC) st.v[i] = *reinterpret_cast<const float4*>(p.Bsrc + off);
        else st.s[i] = p.Bsrc[off];
    }
}

template <bool VEC, int W>
__device__ __forceinline__ void dw_store_b(const DwArgs& p, float* sX, int c0, int tid, const DwStage<VEC, W>& st,
                                           float4 sc, float4 sh) {
    using S = DwStage<VEC, W>;
    const int cl = (tid % S::CPR) * (VEC ? 4 : 1);
    const bool cin = c0 + cl < p.J;
    const bool act = p.b_mode == A_BNACT;
#pragma unroll
    for (int i = 0; i < S::NI; ++i) {
        const int rl = tid / S::CPR + S::RP * i;
        const bool in = cin && st.p0 + rl < p.P;
        if constexpr (VEC) {
            float4 b = make_float4(0, 0, 0, 0);
            if (in) {
                b = st.v[i];
                if (act) {
                    b.x = lrelu(fmaf(sc.x, b.x, sh.x), p.bslope); b.y = lrelu(fmaf(sc.y, b.y, sh.y), p.bslope);
                    b.z = lrelu(fmaf(sc.z, b.z, sh.z), p.bslope); b.w = lrelu(fmaf(sc.w, b.w, sh.w), p.bslope);
                }
            }
            *reinterpret_cast<float4*>(&sX[rl * (W + 4) + cl]) = b;
        } else {
            float b = 0.f;
            if (in) { b = st.s[i]; if (act) b = lrelu(fmaf(sc.x, b, sh.x), p.bslope); }
            sX[rl * (W + 4) + cl] = b;
        }
    }
}

// dW tile (64*TM) x (64*TN) per workgroup (2 x 2 waves, wave tile (32*TM) x (32*TN)); persistent over
// 32-row chunks of P with register prefetch of the next chunk; partial tiles go to `part`.
template <int AM, bool VEC, int TM, int TN, bool RAG>
__device__ __forceinline__ void linear_dw_body(const DwArgs& p_in, const int block) {
    DwArgs p = p_in;
    if (p.p_dev) p.P = *p.p_dev;
    constexpr int WI = 64 * TM, WJ = 64 * TN;
    __shared__ __attribute__((aligned(16))) float sA[DW_BP * (WI + 4)];   // [p][i]
    __shared__ __attribute__((aligned(16))) float sB[DW_BP * (WJ + 4)];   // [p][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, lh = lane >> 5;
    // same XCD-aware decode as the forward / dX kernel: the ti*tj tiles fed by the same rows sit 8 ids apart (one XCD,
    // dispatched together), so the operand columns they share (all of X for tiles along i) are L2 hits
    int bx, bt;
    tile_of_block(block, p.gx, p.ti * p.tj, bx, bt);
    const int i0 = (bt % p.ti) * WI, j0 = (bt / p.ti) * WJ;
    const int chunks = (p.P + DW_BP - 1) / DW_BP;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-thread channel constants (the thread's channels are the same for every chunk)
    float4 asc = make_float4(0, 0, 0, 0), ash = asc, ak2 = asc, amu = asc, bsc = make_float4(1, 1, 1, 1), bsh = asc;
    {
        using SA = DwStage<VEC, WI>;
        using SB = DwStage<VEC, WJ>;
        const int ca = i0 + (tid % SA::CPR) * (VEC ? 4 : 1), cb = j0 + (tid % SB::CPR) * (VEC ? 4 : 1);
        const int cac = min(ca, p.I - (VEC ? 4 : 1)), cbc = min(cb, p.J - (VEC ? 4 : 1));
        if constexpr (VEC) {
            if constexpr (AM != A_PLAIN) {
                asc = *reinterpret_cast<const float4*>(p.sc + cac); ash = *reinterpret_cast<const float4*>(p.sh + cac);
                ak2 = *reinterpret_cast<const float4*>(p.k2 + cac); amu = *reinterpret_cast<const float4*>(p.mu + cac);
            }
            if (p.b_mode == A_BNACT) { bsc = *reinterpret_cast<const float4*>(p.bsc + cbc); bsh = *reinterpret_cast<const float4*>(p.bsh + cbc); }
        } else {
            if constexpr (AM != A_PLAIN) { asc.x = p.sc[cac]; ash.x = p.sh[cac]; ak2.x = p.k2[cac]; amu.x = p.mu[cac]; }
            if (p.b_mode == A_BNACT) { bsc.x = p.bsc[cbc]; bsh.x = p.bsh[cbc]; }
        }
    }

    DwStage<VEC, WI> ra;
    DwStage<VEC, WJ> rb;
    int ch = bx;
    if (ch < chunks) {
        dw_fetch_meta<VEC, WI, RAG>(p, ch * DW_BP, tid, ra);
        dw_load_a<AM, VEC, WI, RAG>(p, ch * DW_BP, i0, tid, ra);
        dw_fetch_meta<VEC, WI, RAG>(p, (ch + p.gx) * DW_BP, tid, ra);       // clamped addresses: always valid
        dw_load_b<VEC, WJ>(p, ch * DW_BP, j0, tid, rb);
    }
    for (; ch < chunks; ch += p.gx) {
        __syncthreads();
        dw_store_a<AM, VEC, WI>(p, sA, i0, tid, ra, asc, ash, ak2, amu);
        dw_store_b<VEC, WJ>(p, sB, j0, tid, rb, bsc, bsh);
        __syncthreads();
        const int nch = ch + p.gx;
        if (nch < chunks) {
            dw_load_a<AM, VEC, WI, RAG>(p, nch * DW_BP, i0, tid, ra);
            dw_fetch_meta<VEC, WI, RAG>(p, (nch + p.gx) * DW_BP, tid, ra);
            dw_load_b<VEC, WJ>(p, nch * DW_BP, j0, tid, rb);
        }
#pragma unroll 4
        for (int ks = 0; ks < DW_BP / 2; ++ks) {
            const int pr = ks * 2 + lh;                 // lanes 0-31: row 2ks, lanes 32-63: row 2ks+1
            // a wave's two tiles along i (j) INTERLEAVE: tile t holds channels 2 lr + t of the wave's 64 -- one 8-byte LDS read feeds both
            // (round 6: one read per tile before; on gfx950 LDS instructions are matrix-pipe time)
            float av[TM], bv[TN];
            if constexpr (TM == 2) { const float2 v = *reinterpret_cast<const float2*>(&sA[pr * (WI + 4) + wr * 64 + 2 * lr]); av[0] = v.x; av[1] = v.y; }
            else av[0] = sA[pr * (WI + 4) + wr * 32 + lr];
            if constexpr (TN == 2) { const float2 v = *reinterpret_cast<const float2*>(&sB[pr * (WJ + 4) + wc * 64 + 2 * lr]); bv[0] = v.x; bv[1] = v.y; }
            else bv[0] = sB[pr * (WJ + 4) + wc * 32 + lr];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
        }
    }
    // accumulator r of tile (tm, tn): row i0 + wr 32 TM + TM i + tm with i = (r & 3) + 8 (r >> 2) + 4 lh (the MFMA's row), column
    // j0 + wc 32 TN + TN lr + tn -- with TN = 2 a lane's two tiles are neighbouring columns: one 8-byte store where the row stride allows
    float* out = p.part + (size_t)bx * p.I * p.J;
    const bool pair = TN == 2 && (p.ldo & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wr * 32 * TM + TM * ((r & 3) + 8 * (r >> 2) + 4 * lh) + tm;
            const int col = j0 + wc * 32 * TN + TN * lr;
            if (row >= p.I) continue;
            if constexpr (TN == 2) {
                if (pair && col + 1 < p.J) { *reinterpret_cast<float2*>(out + (size_t)row * p.ldo + col) = make_float2(acc[tm][0][r], acc[tm][1][r]); continue; }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                if (col + tn < p.J) out[(size_t)row * p.ldo + col + tn] = acc[tm][tn][r];
        }
}
template <int AM, bool VEC, int TM, int TN, bool RAG>
__global__ __launch_bounds__(MLP_T, 2) void linear_dw_kernel(const DwArgs p_in) {
    linear_dw_body<AM, VEC, TM, TN, RAG>(p_in, (int)blockIdx.x);
}

// Both backward GEMMs of a few-row layer (the GroupAll level's 4 096 rows, the part-seg decoders, heads) in ONE launch (round 6): the
// first n_dw workgroups run the weight-gradient body on 128 x 128 tiles, the rest the input-gradient body on 64 x 64 tiles.  The two
// read the same (dU | arg, gz ; Y ; constants) and write disjoint outputs, so nothing orders them; on separate streams they co-ran at the
// SUM of their times, and that is what this launch takes as well -- what it saves is a dispatch + drain (~4.5 us on this
// chip, six times per GroupAll level with the reduce folded into the finish launch) and the idle tail of the first kernel, which the
// second one's workgroups fill.  Same bodies, same arithmetic: bit-identical to the two launches.
template <int AM, int EM, bool VEC>
__global__ __launch_bounds__(MLP_T, 2) void linear_bwd_pair_kernel(const DwArgs d, const LinArgs a, const int n_dw) {
    if ((int)blockIdx.x < n_dw) linear_dw_body<AM, VEC, 2, 2, false>(d, (int)blockIdx.x);
    else linear_nt_body<AM, EM, VEC, 1, 0, false, 1, false>(a, (int)blockIdx.x - n_dw);
}

static void dw_grid(int P, int I, int J, int& gx, int& ti, int& tj, int& tm, int& tn) {
    tm = I > 64 ? 2 : 1; tn = J > 64 ? 2 : 1;
    ti = (I + 64 * tm - 1) / (64 * tm); tj = (J + 64 * tn - 1) / (64 * tn);
    const int chunks = (P + DW_BP - 1) / DW_BP;
    gx = (512 + ti * tj - 1) / (ti * tj);        // 2 resident workgroups per CU in ONE round (768 measured 10-20 % slower)
    if (gx > chunks) gx = chunks;
    if (gx < 1) gx = 1;
}

template <int AM, bool VEC, bool RAG>
static void launch_dw_t2(const DwArgs& d, dim3 grid, int tm, int tn, hipStream_t st) {
    if (tm == 1 && tn == 1) PCL_LAUNCH_TIMED((linear_dw_kernel<AM, VEC, 1, 1, RAG>), grid, dim3(MLP_T), st, d);
    else if (tm == 2 && tn == 1) PCL_LAUNCH_TIMED((linear_dw_kernel<AM, VEC, 2, 1, RAG>), grid, dim3(MLP_T), st, d);
    else if (tm == 1 && tn == 2) PCL_LAUNCH_TIMED((linear_dw_kernel<AM, VEC, 1, 2, RAG>), grid, dim3(MLP_T), st, d);
    else PCL_LAUNCH_TIMED((linear_dw_kernel<AM, VEC, 2, 2, RAG>), grid, dim3(MLP_T), st, d);
}
template <int AM, bool VEC>
static void launch_dw_t(const DwArgs& d, dim3 grid, int tm, int tn, hipStream_t st) {
    if (d.rmeta) launch_dw_t2<AM, VEC, true>(d, grid, tm, tn, st);
    else launch_dw_t2<AM, VEC, false>(d, grid, tm, tn, st);
}

// out[e] = sum_r part[r][e] (out rows of ncols elements, ldo apart): 32 outputs x 8 row-lanes per 256-thread block (partials are L2-resident).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int rows, size_t n, int ncols, int ldo,
                                                          float* __restrict__ out) {
    __shared__ float red[8][33];
    const int el = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const size_t e = (size_t)blockIdx.x * 32 + el;
    float s = 0.f;
    if (e < n) {
#pragma unroll 8
        for (int r = ry; r < rows; r += 8) s += part[(size_t)r * n + e];
    }
    red[ry][el] = s;
    __syncthreads();
    if (ry == 0 && e < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j][el];
        out[ldo == ncols ? e : (e / ncols) * ldo + e % ncols] = t;
    }
}

// ---- forward of a hidden layer with the weight RESIDENT in LDS (round 3) ---------------------------------------------------------
// Y[R x NS] = lrelu(BN_prev(X))[R x CIN] . W[n0..n0+NS)[CIN]^T for the set-abstraction shapes (CIN in {64,128}, Cout in {64,128,256}):
// the plan of the fused backward's dX phase.  A persistent workgroup of 8 waves (two per SIMD) keeps its NS-column slab of W in
// LDS for the whole kernel and walks row tiles of R = 128 (CIN = 64) or 64 (CIN = 128) rows: the tile's rows are transformed
// ONCE on their way into a double-buffered LDS image (the next tile is deposited behind this tile's MFMAs and its raw rows were
// requested a tile earlier), the K loop is LDS reads and MFMAs only -- no B re-staging per row tile, no barrier inside, one
// barrier per tile -- and the epilogue stores the C/D layout as dword buffer stores with the row in the scalar offset and sums
// the BatchNorm statistics about a per-lane pivot.  Cout = 256 runs as two 128-column slabs in workgroups 8 ids apart (same
// XCD: the second reads the rows from L2).  linear_nt_kernel re-stages the weight tile for every 128-row tile behind two
// barriers per 32-wide k step.
constexpr int FR_T = 512;
struct FrArgs {
    const float* X; const float* W; const float* bias; const float* sc; const float* sh; float slope;
    float* Y; double* stats; const int2* rmeta; const int* m_dev;
    int M, N, gx, nt, stat_rows;
};
template <int CI, int NS, bool RAG, bool PLAIN>
__global__ __launch_bounds__(FR_T) void linear_fwd_res_kernel(const FrArgs p_in) {
    FrArgs p = p_in;
    if (p.m_dev) p.M = __builtin_amdgcn_readfirstlane(*p.m_dev);
    constexpr int CIN = 64 * CI, R = CIN == 64 ? 128 : 64;
    constexpr int ALD = CIN + 4, WLD = CIN + 4;
    constexpr int RB = R / 32, CBK = NS / 32, TM = RB * CBK / 8;            // 32x32 tiles per wave (same column block)
    static_assert(RB * CBK == 8 * TM && (TM == 1 || TM == 2), "wave tiling");
    constexpr int CPR = CIN / 4, RP = FR_T / CPR, NI = R / RP;               // staging: thread -> 16-byte piece k4 of rows row0 + RP*i
    constexpr int NWB = NS * CIN / 4 / FR_T;
    __shared__ __attribute__((aligned(16))) float lds[NS * WLD + 2 * R * ALD + 2 * R];
    float* const sW = lds;
    float* const sA = sW + NS * WLD;
    float* const sMult = sA + 2 * R * ALD;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    int bx, by;
    tile_of_block(blockIdx.x, p.gx, p.nt, bx, by);
    const int n0 = by * NS;
    const int tiles = (p.M + R - 1) / R;
    const unsigned irow = CIN * 4u, orow = (unsigned)p.N * 4u;
    const int k4 = (tid % CPR) * 4, row0 = tid / CPR;
    float4 csc = make_float4(1.f, 1.f, 1.f, 1.f), csh = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (!PLAIN) { csc = *reinterpret_cast<const float4*>(p.sc + k4); csh = *reinterpret_cast<const float4*>(p.sh + k4); }
    const int cb = wave % CBK, rb0 = wave / CBK;                             // tiles (rb0 + (8 / CBK) * t, cb), t < TM
    const int col = n0 + cb * 32 + lr;
    const float bias = p.bias ? p.bias[col] : 0.f;

    // weight slab -> LDS (once)
    {
        const rsrc_t rW = buf_rsrc(p.W, (size_t)n0 * irow, (size_t)p.N * irow);
#pragma unroll
        for (int i = 0; i < NWB; ++i) {
            const int e = tid + FR_T * i, n = e / CPR, kk = (e % CPR) * 4;
            const float4 w = buf_ld4(rW, (unsigned)n * irow + (unsigned)kk * 4, 0);
            *reinterpret_cast<float4*>(&sW[n * WLD + kk]) = w;
        }
    }
    float4 rX[NI];
    float rMu = 0.f;
    auto request = [&](int tile) {
        const bool live = tile < tiles;
        const int m0 = tile * R;
        const rsrc_t rA = buf_rsrc(p.X, (size_t)m0 * irow, live ? (size_t)p.M * irow : 0);
        const unsigned vo = (unsigned)row0 * irow + (unsigned)k4 * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) rX[i] = buf_ld4(rA, vo, (unsigned)(RP * i) * irow);
        const int row = m0 + tid;
        if constexpr (RAG) {
            const rsrc_t rR = buf_rsrc(p.rmeta, 0, live ? (size_t)p.M * 8 : 0);
            const int2 rec = buf_ld2i(rR, (tid < R && row < p.M) ? (unsigned)row * 8u : BUF_OOB);
            rMu = (float)(rec.y >> 16);
        } else {
            rMu = (live && tid < R && row < p.M) ? 1.f : 0.f;
        }
    };
    auto deposit = [&](int buf) {
        float* a = sA + buf * R * ALD;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float4 z = rX[i];
            if constexpr (!PLAIN) {             // the layer below's folded BatchNorm + activation, applied once per element
                float t;
                t = fmaf(csc.x, rX[i].x, csh.x); z.x = fmaxf(t, t * p.slope);
                t = fmaf(csc.y, rX[i].y, csh.y); z.y = fmaxf(t, t * p.slope);
                t = fmaf(csc.z, rX[i].z, csh.z); z.z = fmaxf(t, t * p.slope);
                t = fmaf(csc.w, rX[i].w, csh.w); z.w = fmaxf(t, t * p.slope);
            }
            *reinterpret_cast<float4*>(&a[(row0 + RP * i) * ALD + k4]) = z;
        }
        if (tid < R) sMult[buf * R + tid] = rMu;
    };
    int tile = bx;
    request(tile);
    deposit(0);
    request(tile + p.gx);
    __syncthreads();
    double st_s = 0.0, st_q = 0.0;
    for (int it = 0; tile < tiles; tile += p.gx, ++it) {
        const int buf = it & 1;
        const int m0 = tile * R;
        const float* a = sA + buf * R * ALD;
        f32x16 acc[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // operands of step k+1 are read from LDS before the MFMAs of step k are issued (pinned: hipcc sinks loads to their use)
        const float* pa = a + (rb0 * 32 + lr) * ALD + lh * 4;
        const float* pb = sW + (cb * 32 + lr) * WLD + lh * 4;
        struct Op { float4 a[TM]; float4 b; };
        auto ld = [&](int k8) -> Op {
            Op o;
#pragma unroll
            for (int t = 0; t < TM; ++t) o.a[t] = *reinterpret_cast<const float4*>(pa + t * (8 / CBK) * 32 * ALD + k8 * 8);
            o.b = *reinterpret_cast<const float4*>(pb + k8 * 8);
            return o;
        };
        Op cur = ld(0);
#pragma unroll
        for (int k8 = 0; k8 < CIN / 8; ++k8) {
            Op nxt = cur;
            if (k8 + 1 < CIN / 8) nxt = ld(k8 + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t].x, cur.b.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t].y, cur.b.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t].z, cur.b.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t].w, cur.b.w, acc[t], 0, 0, 0);
            }
            cur = nxt;
        }
        // ---- epilogue: store (rows past M are dropped by the range check), multiplicity-weighted sums about a pivot
        {
            const rsrc_t rY = buf_rsrc(p.Y, (size_t)m0 * orow, (size_t)p.M * orow);
            const unsigned v0 = (unsigned)(4 * lh) * orow + (unsigned)col * 4;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int rbase = (rb0 + t * (8 / CBK)) * 32;
                const float piv = acc[t][0] + bias;
                float ts = 0.f, tq = 0.f, tw = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = rbase + (r & 3) + 8 * (r >> 2);               // + 4*lh
                    const float c = acc[t][r] + bias;
                    const float w = sMult[buf * R + rl + 4 * lh];
                    const float dlt = c - piv;
                    tw += w; ts = fmaf(w, dlt, ts); tq = fmaf(w * dlt, dlt, tq);
                    buf_st1(rY, v0, (unsigned)rl * orow, c);
                }
                const double n = (double)tw, pv = (double)piv;
                st_s += (double)ts + n * pv; st_q += (double)tq + 2.0 * pv * (double)ts + n * pv * pv;
            }
        }
        // the next tile (requested a tile ago) goes into the other image: its readers finished before the last barrier
        deposit(buf ^ 1);
        request(tile + 2 * p.gx);
        __syncthreads();
    }
    // ---- this workgroup's row of the BatchNorm sums (its NS columns); the rows the 4-wave kernel's grid would have written beyond
    // gx are cleared so that the finalize sums the same workspace layout
    {
        double s = st_s, q = st_q;
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        double* red = reinterpret_cast<double*>(sA);             // [8 waves][32][2]
        __syncthreads();
        if (lh == 0) { red[(wave * 32 + lr) * 2] = s; red[(wave * 32 + lr) * 2 + 1] = q; }
        __syncthreads();
        if (tid < NS) {
            const int cb_ = tid / 32, l = tid & 31;
            double ss = 0.0, qq = 0.0;
#pragma unroll
            for (int w = 0; w < 8 / CBK; ++w) { ss += red[((w * CBK + cb_) * 32 + l) * 2]; qq += red[((w * CBK + cb_) * 32 + l) * 2 + 1]; }
            double* dst = p.stats + (size_t)bx * 2 * p.N;
            dst[n0 + tid] = ss; dst[p.N + n0 + tid] = qq;
            for (int r = bx + p.gx; r < p.stat_rows; r += p.gx) {
                double* z = p.stats + (size_t)r * 2 * p.N;
                z[n0 + tid] = 0.0; z[p.N + n0 + tid] = 0.0;
            }
        }
    }
}

// ---- the same forward on the bf16 matrix pipe, fp32 operands split three ways (round 4; OPT-IN: pcl_set_matrix_form) -------------------
// gfx950 runs v_mfma_f32_32x32x2_f32 at 157 TF/s -- the packed-FMA rate of the vector ALU, whose issue port it shares -- and
// v_mfma_f32_32x32x16_bf16 at 2.5 PF/s.  An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significant bits:
// h0 = bf16(v), h1 = bf16(v - h0), h2 = v - h0 - h1; both differences are exact in fp32 and h2 fits 8 bits), so an fp32 product
// a*b is the sum of the nine products ah_i * bh_j, each of them exact in the fp32 accumulator's format, and
//      acc += a*b     ==     nine bf16 MFMAs into the same fp32 accumulator (smallest terms first).
// Nothing is dropped: all 24 bits of both operands enter every product; what differs from the fp32 MFMA is the order in which the
// partial products are rounded into the accumulator (tests/test_mlp_hip.py::test_split_gemm_error_vs_fp64 measures both against
// fp64: equally close, mean error 0.7-1.08 x).  Nine 32-cycle MFMAs per 16 k instead of eight 64-cycle ones:
// 0.56 x the matrix time.  Rows are split ONCE on their way into LDS (three bf16 planes, row-major, 16 bytes of padding per row:
// conflict-free ds_read_b128 fragments); a wave's weight fragments (32 output columns x all of K) are split once per kernel and
// stay in registers (48 / 96 VGPRs for K = 64 / 128) -- no LDS copy of the weight, so ONE workgroup covers all Cout <= 256 columns
// of a row tile and the rows are read, transformed and split once instead of once per 128-column slab.
// MEASURED: 128 -> 256 runs 122 -> 100 us, 64 -> 128 and 128 -> 128 the same as the fp32 form, 64 -> 64 slower
// (57 -> 67 us), and the training step 1.91 -> 1.96 ms because a 155-160 KB workgroup per CU leaves no room for the side stream's
// sampling kernels beside it.  Per tile and SIMD the matrix pipe is busy 8.8 k of 14 k cycles: the epilogue (BatchNorm sums, 4 KB
// LDS transpose, stores) and the deposit (BatchNorm + activation + the split: ~30 VALU per float4) are ~420 vector instructions per
// wave and tile beside 144 MFMAs, and vector work beside the partner wave's bf16 MFMAs issues at about a third of its solo rate on
// this chip (late-wave epilogue 3.7 k cycles beside MFMAs, 1.1 k alone; s_setprio does not change it).  With the MFMAs removed the
// same kernel streams at 5.4-6.0 TB/s (full-line dwordx4 stores through the LDS transpose; 4.1 TB/s with 16 dword stores per
// block), with the stores removed it is 15-20 us faster whatever their shape -- the two halves do not overlap inside one
// workgroup's barrier cadence.  So the fp32 MFMA form stays the default; this kernel is kept selectable, with its tests.
// Inputs beyond bf16's range behave like fp32 except |v| > 3.39e38 (rounds to inf in h0, then NaN); no such activations exist here.
template <int CI, int CBK, bool RAG, bool PLAIN>
__global__ __launch_bounds__(FR_T) void linear_fwd_split_kernel(const FrArgs p_in) {
    FrArgs p = p_in;
    if (p.m_dev) p.M = __builtin_amdgcn_readfirstlane(*p.m_dev);
    constexpr int CIN = 64 * CI, N = 32 * CBK, R = CIN == 64 ? 128 : 64;
    constexpr int ALDB = CIN * 2 + 16;                                      // bytes per row of a plane
    constexpr int PSZ = R * ALDB;                                           // one plane of one image
    constexpr int RB = R / 32, WPC = CBK >= 8 ? 1 : 8 / CBK, TM = RB / WPC; // waves per column block; 32x32 tiles per wave
    static_assert(RB % WPC == 0 && TM >= 1 && TM <= 2, "wave tiling");
    constexpr int CPR = CIN / 4, RP = FR_T / CPR, NI = R / RP;               // staging: thread -> 4 k of rows row0 + RP*i
    constexpr int KS = CIN / 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * PSZ + 3 * R * 4 + 2 * FR_T * 16 + 8 * 4096 + 2 * CIN * 4];
    float* const sFold = reinterpret_cast<float*>(lds + 2 * 3 * PSZ + 3 * R * 4 + 2 * FR_T * 16 + 8 * 4096);        // [2][CIN] scale | shift
    float* const sMult = reinterpret_cast<float*>(lds + 2 * 3 * PSZ);
    uint4* const sPiv = reinterpret_cast<uint4*>(lds + 2 * 3 * PSZ + 3 * R * 4);        // [2][FR_T]: this lane's 16 pivots, packed bf16
    unsigned char* const sOut = lds + 2 * 3 * PSZ + 3 * R * 4 + 2 * FR_T * 16;           // [8 waves][32 rows][32 floats]: store staging
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int bx = blockIdx.x;
    const int tiles = (p.M + R - 1) / R;
    const unsigned irow = CIN * 4u, orow = (unsigned)N * 4u;
    const int k4 = (tid % CPR) * 4, row0 = tid / CPR;
    if constexpr (!PLAIN) {                                                  // the folded BatchNorm of the layer below, read per deposit
        if (tid < CIN) { sFold[tid] = p.sc[tid]; sFold[CIN + tid] = p.sh[tid]; }
    }
    const int cb = wave % CBK, rb0 = wave / CBK;                             // tiles (rb0 + WPC * t, cb), t < TM
    const int col = cb * 32 + lr;                                            // the weight row this lane's fragments hold

    float4 rX[NI];
    float rMu = 0.f;
    auto request = [&](int tile) {
        const bool live = tile < tiles;
        const int m0 = tile * R;
        const rsrc_t rA = buf_rsrc(p.X, (size_t)m0 * irow, live ? (size_t)p.M * irow : 0);
        const unsigned vo = (unsigned)row0 * irow + (unsigned)k4 * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) rX[i] = buf_ld4(rA, vo, (unsigned)(RP * i) * irow);
        const int row = m0 + tid;
        if constexpr (RAG) {
            const rsrc_t rR = buf_rsrc(p.rmeta, 0, live ? (size_t)p.M * 8 : 0);
            const int2 rec = buf_ld2i(rR, (tid < R && row < p.M) ? (unsigned)row * 8u : BUF_OOB);
            rMu = (float)(rec.y >> 16);
        } else {
            rMu = (live && tid < R && row < p.M) ? 1.f : 0.f;
        }
    };
    auto deposit = [&](int buf, int mb) {
        unsigned char* a = lds + buf * 3 * PSZ;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float4 z = rX[i];
            if constexpr (!PLAIN) {             // the layer below's folded BatchNorm + activation, applied once per element
                const float4 csc = *reinterpret_cast<const float4*>(sFold + k4), csh = *reinterpret_cast<const float4*>(sFold + CIN + k4);
                float t;
                t = fmaf(csc.x, rX[i].x, csh.x); z.x = fmaxf(t, t * p.slope);
                t = fmaf(csc.y, rX[i].y, csh.y); z.y = fmaxf(t, t * p.slope);
                t = fmaf(csc.z, rX[i].z, csh.z); z.z = fmaxf(t, t * p.slope);
                t = fmaf(csc.w, rX[i].w, csh.w); z.w = fmaxf(t, t * p.slope);
            }
            uint2 w0, w1, w2;
            split3(z.x, z.y, w0.x, w1.x, w2.x); split3(z.z, z.w, w0.y, w1.y, w2.y);
            unsigned char* q = a + (row0 + RP * i) * ALDB + k4 * 2;
            *reinterpret_cast<uint2*>(q) = w0; *reinterpret_cast<uint2*>(q + PSZ) = w1; *reinterpret_cast<uint2*>(q + 2 * PSZ) = w2;
        }
        if (tid < R) sMult[mb * R + tid] = rMu;
    };
    int tile = bx;
    request(tile);
    // this wave's weight fragments: W[col][16 ks + 8 lh .. + 8), split once
    bf16x8 bw[KS][3];
    {
        const float* wrow = p.W + (size_t)col * CIN + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const Split8 s = split8(*reinterpret_cast<const float4*>(wrow + ks * 16), *reinterpret_cast<const float4*>(wrow + ks * 16 + 4));
            bw[ks][0] = s.pl[0]; bw[ks][1] = s.pl[1]; bw[ks][2] = s.pl[2];
        }
    }
    if constexpr (!PLAIN) __syncthreads();                                   // sFold
    deposit(0, 0);
    request(tile + p.gx);
    __syncthreads();
    f32x16 acc;
    // The MFMAs run TRANSPOSED (A = weight fragment, B = row fragment): acc[r] is Y[row lr of the block][channel 8 (r>>2) + 4 lh + (r&3)
    // of column block cb], four consecutive channels per register quad.  A block leaves through a wave-private 4 KB LDS stage (four
    // ds_write_b128, four ds_read_b128, 16-byte chunks XOR-swizzled by the row) as four dwordx4 stores of 8 rows x 128 bytes: full
    // lines, 1 KB per instruction.  (The store path is what bounds this kernel: as 16 dword stores per block -- two 128-byte lines
    // per instruction -- the epilogue took 4-6 k cycles per tile against 4.4 k of MFMAs; as dwordx4 from the accumulator layout -- 32
    // rows x 32 bytes per instruction -- no less.)
    // BatchNorm sums: a lane keeps (sum, sum of squares) of ITS rows for its 16 channels in fp32 about its own pivots (its first
    // value per channel rounded to bf16, parked in LDS) and converts to fp64 once, after the last tile.
    float ss[16], qq[16], tw = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ss[r] = 0.f; qq[r] = 0.f; }
    bool have_piv = false;
    unsigned char* const stage = sOut + wave * 4096;
    auto epilogue = [&](int m0, int mb, int t) {    // block t of the tile at row m0
        const rsrc_t rY = buf_rsrc(p.Y, (size_t)m0 * orow, (size_t)p.M * orow);
        auto bias4 = [&](int g) { return p.bias ? *reinterpret_cast<const float4*>(p.bias + cb * 32 + 8 * g + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f); };
        uint4 pk[2];
        if (have_piv) { pk[0] = sPiv[tid]; pk[1] = sPiv[FR_T + tid]; }
        else {                                  // (pivots: the bias is left out -- any value near the channel's mean serves)
            pk[0] = make_uint4(bf16_pack(acc[0], acc[1]), bf16_pack(acc[2], acc[3]), bf16_pack(acc[4], acc[5]), bf16_pack(acc[6], acc[7]));
            pk[1] = make_uint4(bf16_pack(acc[8], acc[9]), bf16_pack(acc[10], acc[11]), bf16_pack(acc[12], acc[13]), bf16_pack(acc[14], acc[15]));
            sPiv[tid] = pk[0]; sPiv[FR_T + tid] = pk[1];
            have_piv = true;
        }
        const uint32_t pw[8] = {pk[0].x, pk[0].y, pk[0].z, pk[0].w, pk[1].x, pk[1].y, pk[1].z, pk[1].w};
        const int rbase = (rb0 + t * WPC) * 32;
        const float w = sMult[mb * R + rbase + lr];
        tw += w;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bz = bias4(g);
            const float4 c = make_float4(acc[4 * g] + bz.x, acc[4 * g + 1] + bz.y, acc[4 * g + 2] + bz.z, acc[4 * g + 3] + bz.w);
            *reinterpret_cast<float4*>(stage + (lr * 8 + ((2 * g + lh) ^ (lr & 7))) * 16) = c;
            float d, e;
            d = c.x - bf16_lo(pw[2 * g]); e = w * d; ss[4 * g] += e; qq[4 * g] = fmaf(e, d, qq[4 * g]);
            d = c.y - bf16_hi(pw[2 * g]); e = w * d; ss[4 * g + 1] += e; qq[4 * g + 1] = fmaf(e, d, qq[4 * g + 1]);
            d = c.z - bf16_lo(pw[2 * g + 1]); e = w * d; ss[4 * g + 2] += e; qq[4 * g + 2] = fmaf(e, d, qq[4 * g + 2]);
            d = c.w - bf16_hi(pw[2 * g + 1]); e = w * d; ss[4 * g + 3] += e; qq[4 * g + 3] = fmaf(e, d, qq[4 * g + 3]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (lane >> 3);
            const float4 v = *reinterpret_cast<const float4*>(stage + (row * 8 + ((lane & 7) ^ (row & 7))) * 16);
            buf_st4(rY, (unsigned)(rbase + row) * orow + (unsigned)(cb * 32 + 4 * (lane & 7)) * 4, 0, v);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // The two waves of a SIMD run out of phase: waves 0-3 store a tile right after its MFMAs, waves 4-7 keep the accumulators over
    // the barrier and store under the other wave's MFMAs of the NEXT tile (bf16 MFMAs leave the vector ALU and the store path free;
    // in phase, both waves' epilogues would run with the matrix pipe idle).  The multiplicities have three buffers for that.
    const bool late = wave >= 4;
    int it = 0, mb = 0;
#if PCL_EXP == 8                                     // lab build: cycles per phase of the tile loop
    long long tph[5] = {0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define FS_MARK(i) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tlast; tlast = t_; }
#else
#define FS_MARK(i)
#endif
    auto mma = [&](int buf, int t) {            // acc = block t of the tile in image buf
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char* pa = lds + buf * 3 * PSZ + ((rb0 + t * WPC) * 32 + lr) * ALDB + lh * 16;
        struct Op { bf16x8 a[3]; };
        auto ld = [&](int ks) -> Op {
            Op o;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) o.a[pl] = *reinterpret_cast<const bf16x8*>(pa + pl * PSZ + ks * 32);
            return o;
        };
        {
            Op cur = ld(0);                     // operands of step k+1 are read before the MFMAs of step k are issued (pinned: hipcc sinks loads)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Op nxt = cur;
                if (ks + 1 < KS) nxt = ld(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma_split9(bw[ks], cur.a, acc);
                cur = nxt;
            }
        }
    };
    auto refill = [&](int buf) {                // the next tile (requested a tile ago) into the other image, the one after it requested
        const int mn = mb == 2 ? 0 : mb + 1;
        deposit(buf ^ 1, mn);
        request(tile + 2 * p.gx);
        mb = mn;
    };
    for (; tile < tiles; tile += p.gx, ++it) {
        FS_MARK(0)
        const int mbt = mb;
        FS_MARK(1)
#pragma unroll
        for (int t = 0; t < TM; ++t) {          // a block's stores drain under the next block's MFMAs
            __builtin_amdgcn_s_setprio(0);
            mma(it & 1, t);
            __builtin_amdgcn_s_setprio(2);      // the vector / memory phases go first: the partner's MFMAs are paced by the pipe anyway
            if (t + 1 < TM || !late) epilogue(tile * R, mbt, t);
        }
        FS_MARK(2)
        refill(it & 1);                         // (late waves refilling FIRST, under the early waves' MFMAs, measured 10 % slower)
        FS_MARK(3)
        __syncthreads();
        FS_MARK(4)
        if (late) epilogue(tile * R, mbt, TM - 1);
    }
#if PCL_EXP == 8
    if (bx == 37 && (tid == 0 || tid == 448))
        printf("fs<%d,%d> wave %d tiles %d: late-epi %lld | - %lld | mma+epi %lld | refill %lld | barrier %lld  (cycles/tile)\n", CIN, N, wave, it,
               tph[0] / it, tph[1] / it, tph[2] / it, tph[3] / it, tph[4] / it);
#endif
    {
        double* red = reinterpret_cast<double*>(lds);             // [8 waves][32 channels][2]
        __syncthreads();
        const double n = (double)tw;
        uint4 pk[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (have_piv) { pk[0] = sPiv[tid]; pk[1] = sPiv[FR_T + tid]; }
        const uint32_t pw[8] = {pk[0].x, pk[0].y, pk[0].z, pk[0].w, pk[1].x, pk[1].y, pk[1].z, pk[1].w};
#pragma unroll
        for (int c16 = 0; c16 < 16; ++c16) {
            const double pvd = (double)((c16 & 1) ? bf16_hi(pw[c16 >> 1]) : bf16_lo(pw[c16 >> 1])), sd = (double)ss[c16];
            double s = sd + n * pvd, q = (double)qq[c16] + 2.0 * pvd * sd + n * pvd * pvd;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
            if (lr == 0) { const int ch = 8 * (c16 >> 2) + 4 * lh + (c16 & 3); red[(wave * 32 + ch) * 2] = s; red[(wave * 32 + ch) * 2 + 1] = q; }
        }
        __syncthreads();
        if (tid < N) {
            const int cb_ = tid / 32, l = tid & 31;
            double ss = 0.0, qq = 0.0;
#pragma unroll
            for (int w = 0; w < WPC; ++w) { ss += red[((w * CBK + cb_) * 32 + l) * 2]; qq += red[((w * CBK + cb_) * 32 + l) * 2 + 1]; }
            double* dst = p.stats + (size_t)bx * 2 * N;
            dst[tid] = ss; dst[N + tid] = qq;
            for (int r = bx + p.gx; r < p.stat_rows; r += p.gx) {
                double* z = p.stats + (size_t)r * 2 * N;
                z[tid] = 0.0; z[N + tid] = 0.0;
            }
        }
    }
}

// which forward launches take the resident-weight kernel: a hidden layer (folded BatchNorm + activation on the input) of a
// set-abstraction shape with enough rows to keep one workgroup per CU busy for several tiles
static bool fwd_res_eligible(const LinArgs& a) {
    if (!path_switches().fwd_resident || (a.a_mode != A_BNACT && a.a_mode != A_PLAIN) || a.e_mode != E_STORE_STATS || a.gmax || a.n_begin != 0 || a.ldc != a.N) return false;
    if (!((a.K == 64 && (a.N == 64 || a.N == 128)) || (a.K == 128 && (a.N == 128 || a.N == 256)))) return false;
    if (a.M < 32768 || (size_t)a.M * (size_t)(a.N > a.K ? a.N : a.K) * 4 >= 0xffffffffull) return false;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(a.bias)) return false;
    if (a.a_mode == A_PLAIN) return al16(a.A) && al16(a.B);
    return al16(a.A) && al16(a.B) && al16(a.sc) && al16(a.sh) && a.slope >= 0.f && a.slope <= 1.f;
}
template <int CI, int NS>
static int launch_fwd_res_t(const FrArgs& f, bool rag, hipStream_t st) {
    const dim3 grid(f.gx * f.nt), blk(FR_T);
    const bool plain = f.sc == nullptr;
    if (rag && plain) PCL_LAUNCH_TIMED((linear_fwd_res_kernel<CI, NS, true, true>), grid, blk, st, f);
    else if (rag) PCL_LAUNCH_TIMED((linear_fwd_res_kernel<CI, NS, true, false>), grid, blk, st, f);
    else if (plain) PCL_LAUNCH_TIMED((linear_fwd_res_kernel<CI, NS, false, true>), grid, blk, st, f);
    else PCL_LAUNCH_TIMED((linear_fwd_res_kernel<CI, NS, false, false>), grid, blk, st, f);
    return check_launch("pcl_linear_fwd(resident weight)");
}
template <int CI, int CBK>
static int launch_fwd_split_t(const FrArgs& f, bool rag, hipStream_t st) {
    const dim3 grid(f.gx), blk(FR_T);
    const bool plain = f.sc == nullptr;
    if (rag && plain) PCL_LAUNCH_TIMED((linear_fwd_split_kernel<CI, CBK, true, true>), grid, blk, st, f);
    else if (rag) PCL_LAUNCH_TIMED((linear_fwd_split_kernel<CI, CBK, true, false>), grid, blk, st, f);
    else if (plain) PCL_LAUNCH_TIMED((linear_fwd_split_kernel<CI, CBK, false, true>), grid, blk, st, f);
    else PCL_LAUNCH_TIMED((linear_fwd_split_kernel<CI, CBK, false, false>), grid, blk, st, f);
    return check_launch("pcl_linear_fwd(split bf16 planes)");
}
// the matrix-pipe form of the GEMM family: 0 (DEFAULT) = fp32 MFMA, bit 0 / 1 = fp32 operands as three bf16 planes on the bf16 MFMA (opt-in)
static int g_split_mfma = 0;               // bit 0: the resident-operand forward, bit 1: the staged GEMMs with K >= g_split_min_k
static int g_split_min_k = 128;
static bool fwd_split_on() { return (g_split_mfma & 1) != 0; }
static int fr_cu_count();
static int launch_fwd_res(const LinArgs& a, int stat_rows, hipStream_t st) {
    FrArgs f = {};
    f.X = a.A; f.W = a.B; f.bias = a.bias; f.sc = a.a_mode == A_PLAIN ? nullptr : a.sc; f.sh = a.a_mode == A_PLAIN ? nullptr : a.sh; f.slope = a.slope; f.Y = a.C; f.stats = a.stats;
    f.rmeta = a.rmeta; f.m_dev = a.m_dev; f.M = a.M; f.N = a.N; f.stat_rows = stat_rows;
    const int NS = a.N == 64 ? 64 : 128, R = a.K == 64 ? 128 : 64;
    const bool split = fwd_split_on() || ((g_split_mfma & 4) && a.K == 128 && a.N == 256);
    f.nt = split ? 1 : a.N / NS;
    const int tiles = (a.M + R - 1) / R;
    int gx = fr_cu_count() / f.nt;                       // one workgroup per CU in all
    if (gx > tiles) gx = tiles;
    if (gx > stat_rows) gx = stat_rows;
    if (gx < 1) gx = 1;
    f.gx = gx;
    const bool rag = a.rmeta != nullptr;
    if (split) {
        if (a.K == 64 && a.N == 64) return launch_fwd_split_t<1, 2>(f, rag, st);
        if (a.K == 64) return launch_fwd_split_t<1, 4>(f, rag, st);
        if (a.N == 128) return launch_fwd_split_t<2, 4>(f, rag, st);
        return launch_fwd_split_t<2, 8>(f, rag, st);
    }
    if (a.K == 64 && NS == 64) return launch_fwd_res_t<1, 64>(f, rag, st);
    if (a.K == 64) return launch_fwd_res_t<1, 128>(f, rag, st);
    return launch_fwd_res_t<2, 128>(f, rag, st);
}

// ---- fused backward of one layer: dX and dW from ONE pass over (dU | arg,gz ; Y ; Yprev) ---------------------------------
// The two backward GEMMs of a layer both need dy = a*du - w*(k1 + k2*(y - mean)): formed separately, dy costs its VALU twice
// (VALU time IS matrix time on gfx950, see the vector-path header) and (dU, Y) are read from HBM twice -- 2 of the 7
// P-sized streams of a layer's backward.  Here a persistent workgroup (8 waves, two per SIMD so that one wave's LDS and
// barrier waits are the other's issue slots) walks row tiles of R = 64 rows (128 when Cin = 64); per tile
//   1. dy[R x Cout] is formed once into LDS (row-major, 4 dwords of padding) and Yprev[R x Cin] is staged raw.  Sparse mode (the
//      layer under a max pool: du is gz[g][c] at ONE row per group and channel, the winner arg[g][c], and 0 elsewhere): the
//      dense part -w (k1 + k2 (y - mean)) is deposited from Y alone and the winners are added in a second short pass, one
//      (group, channel) pair per thread -- `dy[winner row][c] = fma(a, gz, dense)`, the same value the per-row form computes,
//      bit for bit.  (Round 4.  Before, every row's thread gathered its group's (arg, gz) rows -- 2/3 of the 112 prefetch
//      registers, 131 KB of L2 reads per tile for ~5 KB of information, and 8 compare / select per element.)
//   2. dX = dy W: eight 32x32 output tiles, one per wave, K = Cout; the weight (as stored, [k][n]) is LDS-resident for the
//      whole kernel when it fits (<= 64 KB) -- then this loop is LDS reads and MFMAs only, no VALU, no barrier.  The one
//      shape where it does not (256 x 128: 128 KB beside a 100 KB tile image) takes the B operand straight from L2 into
//      registers: a lane's MFMA operand is W[k][its column], 32 consecutive floats of a weight row per half-wave, requested
//      three 8-k steps ahead of the MFMAs that use them (round 4; before, 32-row chunks went through a double-buffered LDS
//      stage with a block barrier per chunk, which kept all eight waves in lockstep: no wave's epilogue could run under
//      another's MFMAs, and every chunk boundary drained the matrix pipe).  The epilogue masks with
//      relu'(BN(Yprev)) read from LDS, sums (du, du*Yprev) for the BatchNorm below and stores dU_prev in the C/D layout;
//   3. dW += dy^T z, z = lrelu(BN(Yprev)) applied while reading the staged Yprev: K = the tile's rows, A = dy columns, B = z
//      columns; the [Cout x Cin] accumulators stay in registers for the whole kernel and leave once, as this workgroup's
//      partial tile.  Round 6: a wave owns TMW = min(Cout / 32, 4) output-channel tiles whose channels INTERLEAVE -- tile a holds
//      channels TMW i + a of its 128- (64-) channel block -- so that the A operands of its TMW MFMAs are ONE 16- (8-) byte LDS read of
//      the dy row as stored (lane i takes dy[row][TMW i .. TMW i + TMW - 1]); one input-channel block per wave, so the z
//      transform runs once per TMW MFMAs; the waves that this leaves over split the tile's rows (KW row groups), and the KW
//      partial sums meet in LDS once, at the kernel's end.  Per MFMA of this loop: 0.5 LDS reads + 0.75 vector instructions
//      (128 x 64; before: 2 + 3) -- on gfx950 every one of them is matrix-pipe time.
// The next tile's raw operands are requested into registers where nothing that is waited for soon queues behind them in the
// wave's in-order memory counter: under the dX loop (resident weight), or under the FIRST HALF of the dW loop (weight from L2:
// a weight operand waited for behind a row-tile prefetch waits for all of it, so the prefetch has landed before dX starts).
// Supported: (Cout, Cin) in {64,128} x {64,128} and 256 x 128, a masked (non-first) layer.
#ifndef PCL_FB_TWO_PRIO
#define PCL_FB_TWO_PRIO 3
#endif
constexpr int FB_T = 512;
__host__ __device__ constexpr int fb_rows(int Cin) { return Cin == 64 ? 128 : 64; }
__host__ __device__ constexpr bool fb_resident(int Cout, int Cin) { return Cout * Cin * 4 <= 64 * 1024; }
// partial dW tiles a workgroup writes (round 6: always one -- the row groups of a tile, fb_wsplit, are summed in LDS at the kernel's end)
__host__ __device__ constexpr int fb_ksplit(int, int) { return 1; }
// 128 x 64 runs 128-row tiles (48 prefetch registers): with four tiles per wave (64 accumulator registers) hipcc spills 24-34 registers into
// the tile loop; two tiles per wave (8-byte A reads, 32 accumulators) spill nothing and measure faster -- 177 against 183 us, 186 before the
// interleaved tiles (gpurun_out/r06c, one box).  `make EXP=9` builds the four-tile form for A/B.
#if PCL_EXP == 9
#define PCL_FB_TMW_128x64 4
#else
#define PCL_FB_TMW_128x64 2
#endif
__host__ __device__ constexpr int fb_tmw(int Cout, int Cin) { return Cout == 128 && Cin == 64 ? PCL_FB_TMW_128x64 : Cout / 32 < 4 ? Cout / 32 : 4; }      // dW tiles per wave = dwords per A read
__host__ __device__ constexpr int fb_wsplit(int Cout, int Cin) { return 8 / ((Cout / 32 / fb_tmw(Cout, Cin)) * (Cin / 32)); }     // row groups of a tile in the dW phase

struct FbArgs {
    const float* dU; const float* Y;                            // [P,Cout] (dU null in the sparse mode)
    const float* a; const float* k1; const float* k2; const float* mu;       // [Cout]
    const int32_t* arg; const float* gz; int ns;                // sparse max-pool gradient [P/ns or groups, Cout]
    const float* W;                                             // [Cout][Cin]
    const float* Yprev; const float* psc; const float* psh; float pslope;     // layer below: pre-BN output [P,Cin], folded BN
    float* dUprev;                                              // [P,Cin]
    double* stats;                                              // [gx][2][Cin]
    float* part;                                                // [gx * ksplit][Cout][Cin] partial dW tiles
    const int* p_dev; const int2* rmeta;                        // compacted rows
    int P, gx;
#if PCL_EXP == 7
    int lab_slot;                                               // lab build: launch number of this shape (ring slot of the time stamps)
#endif
};

#if PCL_EXP == 7
// lab build only: per-workgroup time stamps of the last 8 launches of the four headline shapes, read back by pcl_lab_fbk_read
// (tools/fb_budget.py).  A device-global buffer instead of printf: 512 hostcall printfs per launch stretched the launch tenfold.
constexpr int FBK_SHAPES = 4, FBK_RING = 8, FBK_WORDS = 16;
__device__ long long g_fbk[FBK_SHAPES][FBK_RING][256][2][FBK_WORDS];
__host__ __device__ constexpr int fbk_shape(bool sparse, int co, int ci) {
    return sparse && co == 4 && ci == 2 ? 0 : sparse && co == 2 && ci == 1 ? 1 : !sparse && co == 2 && ci == 2 ? 2 : !sparse && co == 1 && ci == 1 ? 3 : -1;
}
#endif
template <bool SPARSE, bool RAG, int CO, int CI, bool TWO = false>
__global__ __launch_bounds__(FB_T) void linear_bwd_fused_kernel(const FbArgs p_in) {
    FbArgs p = p_in;
#if PCL_EXP == 7
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
    const long long cy_entry = __builtin_readcyclecounter();
#endif
    if (p.p_dev) p.P = __builtin_amdgcn_readfirstlane(*p.p_dev);
    // TWO (round 6, the 128 x 64 shape): TWO tile images of R = 64 rows beside the resident weight (2 x 51 + 34 KB) and the waves in two
    // roles -- waves 0-3 (one per SIMD) own the four dX tiles of a row tile, waves 4-7 its dW tiles (two each, all 64 rows: no row groups)
    // -- so that while tile t's MFMAs run on image t & 1, every wave deposits tile t + 1 into the other image and re-requests the freed
    // registers for tile t + 2 BETWEEN its MFMAs.  The deposit, its wait for the loads and the barrier behind it -- the 4.6 k of 24.5 k
    // cycles per 128 rows in which no wave of a SIMD had an MFMA to issue (profiles/r05_fused_backward_cycle_budget.txt) -- leave the
    // critical path; what stays serial per tile is the winners' pass (sparse mode) between two barriers.
    constexpr int COUT = 64 * CO, CIN = 64 * CI, R = TWO ? 64 : fb_rows(CIN);
    constexpr bool WRES = fb_resident(COUT, CIN);
    static_assert(!TWO || (WRES && COUT == 128 && CIN == 64), "two tile images: the 128 x 64 shape");
    // LDS row strides (dwords).  The resident weight is kept TRANSPOSED, sW[n][k] with WLD = COUT + 4: the B operands of four
    // consecutive dX MFMAs -- W[k .. k + 3][n] -- are then one 16-byte read like the A operands (round 6; as stored they were four
    // single-dword reads a row apart)
    constexpr int DLD = COUT + 4, YLD = CIN + 4, WLD = COUT + 4;
    constexpr int WROWS = WRES ? CIN : 0;                                // resident weight (else: B operand from L2, no LDS copy)
    // staging maps: thread -> (16-byte column piece, rows row0 + RP*i)
    constexpr int CPR_O = COUT / 4, RP_O = FB_T / CPR_O, NI_O = R / RP_O;
    constexpr int CPR_I = CIN / 4, RP_I = FB_T / CPR_I, NI_I = R / RP_I;
    constexpr int NWB = WRES ? COUT * CIN / 4 / FB_T : 1;                // 16-byte weight pieces per thread (resident weight)
    // dX: (R/32) x (CIN/32) = 8 output tiles, wave -> (rbx, cbx)
    constexpr int CB = CIN / 32;
    static_assert((R / 32) * CB == (TWO ? 4 : 8), "one dX tile per wave (TWO: per wave of the dX role)");
    // dW: NTI x NTJ accumulator tiles over KW row groups of WPG waves; wave (wa, wb) of a group owns the TMW channel-interleaved
    // tiles of output-channel block wa (32 TMW channels) x input-channel block wb (see the header, point 3)
    constexpr int NTI = COUT / 32, NTJ = CIN / 32, TMW = fb_tmw(COUT, CIN), TNW = 1, WA = NTI / TMW, WB = NTJ;
    constexpr int KW = TWO ? 1 : fb_wsplit(COUT, CIN), WPG = TWO ? 4 : 8 / KW;
    static_assert(WA * WB == WPG && TMW * WA == NTI && WPG * KW == (TWO ? 4 : 8) && (TMW == 2 || TMW == 4), "dW tiling");
    constexpr int KR = R / KW;                                           // rows of a tile one wave group accumulates
    static_assert(KW == 1 || (size_t)KW * COUT * CIN <= (size_t)R * DLD + R * YLD + WROWS * WLD, "the row groups' partial tiles meet in the tile image's LDS");
    constexpr int IMG = R * DLD + R * YLD, NIMG = TWO ? 2 : 1;          // one tile image: dy [R][DLD] | Yprev [R][YLD]
    __shared__ __attribute__((aligned(16))) float lds[NIMG * IMG + WROWS * WLD + 2 * R * 2 + 2 * 128 + 4 * COUT];
    float* sDY = lds;                                                    // (TWO: re-pointed at the image of the tile per iteration)
    float* sY = sDY + R * DLD;
    float* const sWb = lds + NIMG * IMG;
    int2* const sMeta = reinterpret_cast<int2*>(sWb + WROWS * WLD);      // [2][R] row records {group, row-in-group | mult << 16}
    // [2][128] sparse mode: (tile row) - (row-in-group) of the rows of group g, at slot g & 127 (a tile's <= R groups are consecutive)
    int* const sDelta = reinterpret_cast<int*>(sMeta + 2 * R);
    // [4][COUT] the dy transform's constants (a, k1, k2, mean): read back once per tile by the deposit -- held in registers for the
    // whole kernel they were 16 of the registers the 256 x 128 shape does not have
    float* const sCst = reinterpret_cast<float*>(sDelta + 2 * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int bx = blockIdx.x;
    const int tiles = (p.P + R - 1) / R;
    const unsigned orow = COUT * 4u, irow = CIN * 4u;                    // bytes per row
    // per-thread constants of the dy transform (this thread's 4 columns, fixed for the whole kernel)
    const int oc4 = (tid % CPR_O) * 4, orow0 = tid / CPR_O;
    const int ic4 = (tid % CPR_I) * 4, irow0 = tid / CPR_I;
    for (int c = tid; c < COUT; c += FB_T) { sCst[c] = p.a[c]; sCst[COUT + c] = p.k1[c]; sCst[2 * COUT + c] = p.k2[c]; sCst[3 * COUT + c] = p.mu[c]; }
    const rsrc_t rArg = buf_rsrc(p.arg, 0, SPARSE ? 0xfffffff0ull : 0), rGz = buf_rsrc(p.gz, 0, SPARSE ? 0xfffffff0ull : 0);      // (ends below BUF_OOB: the guard offset of a group past the tile must fail the range check)
    const rsrc_t rW = buf_rsrc(p.W, 0, (size_t)COUT * irow);
    // dX tile of this wave and the per-lane constants of its column (mask of the layer below); dW tiles and their z transform
    const int rbx = wave / CB, cbx = wave % CB;
    const int xcol = cbx * 32 + lr;
    const float xsc = p.psc[xcol], xsh = p.psh[xcol];
    const int kq = TWO ? 0 : wave / WPG, wq = wave % WPG, wa = wq % WA, wb = wq / WA;
    float zsc[TNW], zsh[TNW];
#pragma unroll
    for (int b = 0; b < TNW; ++b) { zsc[b] = p.psc[(wb * TNW + b) * 32 + lr]; zsh[b] = p.psh[(wb * TNW + b) * 32 + lr]; }
    f32x16 accw[TMW][TNW];
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b = 0; b < TNW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw[a][b][r] = 0.f;
    double st_s = 0.0, st_q = 0.0;

    // ---- raw operands of one tile in registers (256 x 128 sparse: 64 accumulator + 112 prefetch registers; they fit because the
    // requests are issued in slices through the dW loop -- all at once they spilled)
    float4 rY[NI_O], rU[SPARSE ? 1 : NI_O], rP[NI_I];
    // sparse mode: (arg, gz) of the first NFX * GP groups of the tile at this thread's channel (thread -> channel tid % COUT of
    // group tid / COUT of the round); further groups of a tile (rare: many groups of one or two rows) are read in the pass itself
    constexpr int GP = FB_T / COUT, NFX = COUT >= 128 ? 2 : 4;        // (the 128- and 256-wide shapes have no registers to spare: 8 | 4 groups per tile prefetched, the rest read in the pass)
    const int fxc = tid % COUT, fxg = tid / COUT;
    int fxA[SPARSE ? NFX : 1]; float fxG[SPARSE ? NFX : 1];
    const float fxa = SPARSE ? p.a[fxc] : 0.f;
    int2 rM = make_int2(0, 0);                       // row record of row tid (< R) of the tile after the requested one
    auto record_of = [&](int tile) -> int2 {         // row record of row tid of `tile`
        // a row past P gets multiplicity 0 and a row-in-group no `arg` entry can equal: du = 0 and dy = 0 exactly
        const int row = tile * R + tid;
        const bool in = tid < R && row < p.P;
        int2 rec;
        if constexpr (RAG) {
            const rsrc_t rR = buf_rsrc(p.rmeta, 0, (size_t)p.P * 8);
            rec = buf_ld2i(rR, in ? (unsigned)row * 8u : BUF_OOB);
        } else {
            const int g = SPARSE ? row / p.ns : 0;
            rec = make_int2(g, (SPARSE ? row - g * p.ns : 0) | (1 << 16));
        }
        return in ? rec : make_int2(0, 0xffff);
    };
    auto meta_store = [&](int mb, int2 rec) {        // row record of row tid -> LDS (+ the group's row offset, sparse mode)
        if (tid < R) {
            sMeta[mb * R + tid] = rec;
            if constexpr (SPARSE) {
                const int srow = rec.y & 0xffff;
                if (srow != 0xffff && (tid == 0 || srow == 0)) sDelta[mb * 128 + (rec.x & 127)] = tid - srow;
            }
        }
    };
    // The loads of a tile are a list of NREQ single requests (compile-time index n), issued a few at a time between the
    // MFMAs of the loop the prefetch flies under: a wave that issues 20-28 16-byte loads back to back sits in the memory
    // issue queue for thousands of cycles while the matrix pipe of its SIMD idles (both waves of a SIMD are in the same
    // phase).  Rows orow0 + RP*i: the row step goes into the scalar offset (one per-lane offset for all i, no VALU per load);
    // the hardware range check covers per-lane + scalar offset (measured: tools/ubench/bufcheck.hip), so rows past P --
    // and every row when there is no next tile (`len` = 0) -- read 0 without touching memory.
    constexpr int NREQ_O = NI_O * (SPARSE ? 1 : 2), NREQ = NREQ_O + NI_I + (SPARSE ? 2 * NFX : 0);
    struct ReqCtx { rsrc_t y, u, pr; unsigned vo, vi; int gf, gl; };
    auto tile_groups = [&](int tile, int mb, int& gf, int& gl) {      // first / last group of `tile` (its records are in sMeta[mb])
        int nv = p.P - tile * R; nv = nv > R ? R : nv;
        gf = 0; gl = -1;
        if (nv > 0) { gf = sMeta[mb * R].x; gl = sMeta[mb * R + nv - 1].x; }
    };
    auto req_open = [&](int tile, int mb, bool live) -> ReqCtx {
        const int m0 = tile * R;
        ReqCtx c;
        c.y = buf_rsrc(p.Y, (size_t)m0 * orow, live ? (size_t)p.P * orow : 0);
        c.u = buf_rsrc(p.dU, (size_t)m0 * orow, !SPARSE && live ? (size_t)p.P * orow : 0);
        c.pr = buf_rsrc(p.Yprev, (size_t)m0 * irow, live ? (size_t)p.P * irow : 0);
        c.vo = (unsigned)orow0 * orow + (unsigned)oc4 * 4; c.vi = (unsigned)irow0 * irow + (unsigned)ic4 * 4;
        c.gf = 0; c.gl = -1;
        if constexpr (SPARSE) { if (live) tile_groups(tile, mb, c.gf, c.gl); }
        return c;
    };
    auto req_one = [&](const ReqCtx& c, int n) {     // n: compile-time after unrolling
        if (n < NREQ_O) {
            if constexpr (SPARSE) rY[n] = buf_ld4(c.y, c.vo, (unsigned)(RP_O * n) * orow);
            else {
                const int i = n / 2;
                if (n % 2 == 0) rY[i] = buf_ld4(c.y, c.vo, (unsigned)(RP_O * i) * orow);
                else rU[i] = buf_ld4(c.u, c.vo, (unsigned)(RP_O * i) * orow);
            }
        } else if (n < NREQ_O + NI_I) {
            const int i = n - NREQ_O;
            rP[i] = buf_ld4(c.pr, c.vi, (unsigned)(RP_I * i) * irow);
        } else if constexpr (SPARSE) {
            const int j = (n - NREQ_O - NI_I) / 2, g = c.gf + j * GP + fxg;
            const unsigned go = g <= c.gl ? ((unsigned)g * COUT + (unsigned)fxc) * 4u : BUF_OOB;
            if ((n - NREQ_O - NI_I) % 2 == 0) fxA[j] = __builtin_bit_cast(int, buf_ld1(rArg, go, 0));
            else fxG[j] = buf_ld1(rGz, go, 0);
        }
    };
    auto req_slice = [&](const ReqCtx& c, int j, int J) {        // slice j of J of the list
#pragma unroll
        for (int n = j * NREQ / J; n < (j + 1) * NREQ / J; ++n) req_one(c, n);
    };
    auto deposit = [&](int mb) {                     // registers -> LDS image of the requested tile: dy (transformed), Yprev (raw)
        // (base pointer + compile-time step: the steps fold into the DS offset fields; indexed from the array start hipcc
        // hoists one address register per row out of the tile loop and spills them)
        const int2* mrow = sMeta + mb * R + orow0;
        float* const dst = sDY + orow0 * DLD + oc4;
        float* const dsty = sY + irow0 * YLD + ic4;
        const float4 ca = *reinterpret_cast<const float4*>(sCst + oc4), ck1 = *reinterpret_cast<const float4*>(sCst + COUT + oc4);
        const float4 ck2 = *reinterpret_cast<const float4*>(sCst + 2 * COUT + oc4), cmu = *reinterpret_cast<const float4*>(sCst + 3 * COUT + oc4);
        (void)ca;
#pragma unroll
        for (int i = 0; i < NI_O; ++i) {
            const float w = (float)(mrow[RP_O * i].y >> 16);
            const float4 y = rY[i];
            float4 d;            // rows past P: du = y = 0 (range-checked loads) and w = 0 -> dy = 0 exactly
            if constexpr (SPARSE) {      // the dense part; fma(a, du, this) with du = 0 is this value itself
                d.x = -w * fmaf(ck2.x, y.x - cmu.x, ck1.x); d.y = -w * fmaf(ck2.y, y.y - cmu.y, ck1.y);
                d.z = -w * fmaf(ck2.z, y.z - cmu.z, ck1.z); d.w = -w * fmaf(ck2.w, y.w - cmu.w, ck1.w);
            } else {
                const float4 du = rU[i];
                d.x = fmaf(ca.x, du.x, -w * fmaf(ck2.x, y.x - cmu.x, ck1.x)); d.y = fmaf(ca.y, du.y, -w * fmaf(ck2.y, y.y - cmu.y, ck1.y));
                d.z = fmaf(ca.z, du.z, -w * fmaf(ck2.z, y.z - cmu.z, ck1.z)); d.w = fmaf(ca.w, du.w, -w * fmaf(ck2.w, y.w - cmu.w, ck1.w));
            }
            *reinterpret_cast<float4*>(dst + RP_O * i * DLD) = d;
        }
#pragma unroll
        for (int i = 0; i < NI_I; ++i) *reinterpret_cast<float4*>(dsty + RP_I * i * YLD) = rP[i];
    };
    // weight rows (32*kc ...) of W[Cout][Cin] -> registers -> LDS rows (32*buf ...): one chunk of 32, or (kc = buf = 0) all
    float4 rWt[NWB];
    auto w_request = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NWB; ++i) {
            // 16-byte piece tid + FB_T * i: row tid / CPR_I + (FB_T / CPR_I) * i (the uniform part goes into the scalar offset)
            rWt[i] = buf_ld4(rW, (unsigned)(tid / CPR_I) * irow + (unsigned)(tid % CPR_I) * 16u, (unsigned)(kc * 32 + (FB_T / CPR_I) * i) * irow);
        }
    };
    auto w_deposit = [&](int) {                      // piece (row k, columns n .. n + 3) of W[Cout][Cin] -> sW[n + j][k] (once per kernel)
#pragma unroll
        for (int i = 0; i < NWB; ++i) {
            float* q = sWb + ((tid % CPR_I) * 4) * WLD + tid / CPR_I + (FB_T / CPR_I) * i;
            q[0] = rWt[i].x; q[WLD] = rWt[i].y; q[2 * WLD] = rWt[i].z; q[3 * WLD] = rWt[i].w;
        }
    };

#if PCL_EXP == 7                                     // lab build: cycles per phase of the tile loop, printed by two waves of one workgroup
    long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define FB_MARK(i) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tlast; tlast = t_; }
    // whole-kernel budget (round 5): shader-clock cycles (s_memtime) beside the constant 100 MHz counter (s_memrealtime) at the
    // kernel's entry, after the prologue, after the tile loop, after the partial-tile write-out has DRAINED and at the exit
    long long tk[5] = {cy_entry, 0, 0, 0, 0};
#else
#define FB_MARK(i)
#endif
    int tile = bx, it = 0;
    if constexpr (TWO) {
        static_assert(!TWO || TMW == 2, "two tile images: two dW tiles per wave of the dW role");
        // ================= two tile images, waves in two roles (see the head of the kernel) =================
        // Side work of a tile = NIT items (a 16-byte piece of Y [and dU] or of Yprev per thread and item): item n of tile t + 1 is
        // deposited from its registers into the other image, then the registers are re-requested for tile t + 2; in sparse mode the
        // (arg, gz) of tile t + 1's first groups follow.  One item per MFMA step (dX: 4 MFMAs, dW: every second step of 2 MFMAs).
        constexpr int NIT = NI_O + NI_I, NSIDE = NIT + (SPARSE ? 2 * NFX : 0);
        static_assert(NSIDE <= COUT / 8 && NSIDE <= R / 4, "side work fits the MFMA loops");
        struct RowCtx { rsrc_t y, u, pr; };
        auto rows_open = [&](int tile_, bool live) -> RowCtx {
            const int r0 = tile_ * R;
            RowCtx c;
            c.y = buf_rsrc(p.Y, (size_t)r0 * orow, live ? (size_t)p.P * orow : 0);
            c.u = buf_rsrc(p.dU, (size_t)r0 * orow, !SPARSE && live ? (size_t)p.P * orow : 0);
            c.pr = buf_rsrc(p.Yprev, (size_t)r0 * irow, live ? (size_t)p.P * irow : 0);
            return c;
        };
        const unsigned vo = (unsigned)orow0 * orow + (unsigned)oc4 * 4, vi = (unsigned)irow0 * irow + (unsigned)ic4 * 4;
        auto item_req = [&](const RowCtx& c, int n) {                       // n: compile-time after unrolling
            if (n < NI_O) {
                rY[n] = buf_ld4(c.y, vo, (unsigned)(RP_O * n) * orow);
                if constexpr (!SPARSE) rU[n] = buf_ld4(c.u, vo, (unsigned)(RP_O * n) * orow);
            } else rP[n - NI_O] = buf_ld4(c.pr, vi, (unsigned)(RP_I * (n - NI_O)) * irow);
        };
        // the dy transform's constants of this thread's four channels: registers for the whole kernel (this shape has them)
        const float4 ca = *reinterpret_cast<const float4*>(p.a + oc4), ck1 = *reinterpret_cast<const float4*>(p.k1 + oc4);
        const float4 ck2 = *reinterpret_cast<const float4*>(p.k2 + oc4), cmu = *reinterpret_cast<const float4*>(p.mu + oc4);
        (void)ca;
        float wrow[NI_O];                                                   // multiplicities of this thread's rows of the tile being deposited
        auto rows_w = [&](int mb_) {
#pragma unroll
            for (int n = 0; n < NI_O; ++n) wrow[n] = (float)(sMeta[mb_ * R + orow0 + RP_O * n].y >> 16);
        };
        auto item_dep = [&](int img, int n) {                               // registers of item n -> image `img`
            float* const base = lds + img * IMG;
            if (n < NI_O) {
                const float w = wrow[n];
                const float4 y = rY[n];
                float4 d;            // rows past P: du = y = 0 (range-checked loads) and w = 0 -> dy = 0 exactly
                if constexpr (SPARSE) {
                    d.x = -w * fmaf(ck2.x, y.x - cmu.x, ck1.x); d.y = -w * fmaf(ck2.y, y.y - cmu.y, ck1.y);
                    d.z = -w * fmaf(ck2.z, y.z - cmu.z, ck1.z); d.w = -w * fmaf(ck2.w, y.w - cmu.w, ck1.w);
                } else {
                    const float4 du = rU[n];
                    d.x = fmaf(ca.x, du.x, -w * fmaf(ck2.x, y.x - cmu.x, ck1.x)); d.y = fmaf(ca.y, du.y, -w * fmaf(ck2.y, y.y - cmu.y, ck1.y));
                    d.z = fmaf(ca.z, du.z, -w * fmaf(ck2.z, y.z - cmu.z, ck1.z)); d.w = fmaf(ca.w, du.w, -w * fmaf(ck2.w, y.w - cmu.w, ck1.w));
                }
                *reinterpret_cast<float4*>(base + (orow0 + RP_O * n) * DLD + oc4) = d;
            } else *reinterpret_cast<float4*>(base + R * DLD + (irow0 + RP_I * (n - NI_O)) * YLD + ic4) = rP[n - NI_O];
        };
        auto fx_req = [&](int gf, int gl, int j2) {                         // (arg, gz) of groups gf + (j2 / 2) GP + fxg of the next tile
            if constexpr (SPARSE) {
                const int j = j2 / 2, g = gf + j * GP + fxg;
                const unsigned go = g <= gl ? ((unsigned)g * COUT + (unsigned)fxc) * 4u : BUF_OOB;
                if (j2 % 2 == 0) fxA[j] = __builtin_bit_cast(int, buf_ld1(rArg, go, 0));
                else fxG[j] = buf_ld1(rGz, go, 0);
            }
        };
        auto winners = [&](int tile_, int mb_, int img) {                   // dy[row of slot arg[g][c] of group g][c] += a[c] gz[g][c]
            if constexpr (SPARSE) {
                float* const dyb = lds + img * IMG;
                int gf, gl;
                tile_groups(tile_, mb_, gf, gl);
                const int* dl = sDelta + mb_ * 128;
                auto fix = [&](int g, int srow, float gzv) {
                    const int r = srow + dl[g & 127];
                    if (g <= gl && (unsigned)r < (unsigned)R) {
                        float* q = dyb + r * DLD + fxc;
                        *q = fmaf(fxa, gzv, *q);
                    }
                };
#pragma unroll
                for (int j = 0; j < NFX; ++j)
                    if (gf + j * GP <= gl) fix(gf + j * GP + fxg, fxA[j], fxG[j]);
                for (int g0 = gf + NFX * GP; g0 <= gl; g0 += GP) {          // (rare) groups beyond the prefetched rounds
                    const int g = g0 + fxg;
                    const unsigned go = g <= gl ? ((unsigned)g * COUT + (unsigned)fxc) * 4u : BUF_OOB;
                    fix(g, __builtin_bit_cast(int, buf_ld1(rArg, go, 0)), buf_ld1(rGz, go, 0));
                }
            }
        };
        if (tile < tiles) {
            // prologue: image 0 <- tile 0 (dense part), registers <- tile 1, (arg, gz) registers <- tile 0
            meta_store(0, record_of(tile));
            rM = record_of(tile + p.gx);
            w_request(0); w_deposit(0);
            __syncthreads();
            const RowCtx c0 = rows_open(tile, true);
#pragma unroll
            for (int n = 0; n < NIT; ++n) item_req(c0, n);
            if constexpr (SPARSE) {
                int gf, gl;
                tile_groups(tile, 0, gf, gl);
#pragma unroll
                for (int j2 = 0; j2 < 2 * NFX; ++j2) fx_req(gf, gl, j2);
            }
            const RowCtx c1 = rows_open(tile + p.gx, tile + p.gx < tiles);
            rows_w(0);
#pragma unroll
            for (int n = 0; n < NIT; ++n) { item_dep(0, n); item_req(c1, n); }
            __syncthreads();                                                // the dense part of tile 0 is complete
        }
#if PCL_EXP == 7
        tk[1] = tlast = __builtin_readcyclecounter();
#endif
        for (; tile < tiles; tile += p.gx, ++it) {
            const int m0 = tile * R, cur = it & 1, nxt = cur ^ 1;
            sDY = lds + cur * IMG; sY = sDY + R * DLD;
            FB_MARK(6)
            meta_store(nxt, rM);                                            // records of tile t + 1 (the slot's last readers passed barrier B)
            winners(tile, cur, cur);
            rM = record_of(tile + 2 * p.gx);
            FB_MARK(0)
            __syncthreads();                                                // A: image `cur` and the records of tile t + 1 are complete
            FB_MARK(1)
            const RowCtx cq = rows_open(tile + 2 * p.gx, tile + 2 * p.gx < tiles);
            int gf1 = 0, gl1 = -1;
            if constexpr (SPARSE) { if (tile + p.gx < tiles) tile_groups(tile + p.gx, nxt, gf1, gl1); }
            rows_w(nxt);
            auto side = [&](int k) {                                        // slot k of the tile's side work (compile-time after unrolling)
                if (k < NIT) { item_dep(nxt, k); item_req(cq, k); }
                else if (k < NSIDE) fx_req(gf1, gl1, k - NIT);
            };
            if (wave < 4) {
                // ---- dX = dy W on this wave's 32 x 32 tile (K = COUT), then its epilogue.  The dX wave of a SIMD issues ahead of its dW wave
                // (s_setprio) so that its epilogue -- LDS reads, compares, 16 stores, no MFMA -- runs while the dW wave still has MFMAs to
                // issue; at equal priority both loops end together and the matrix pipe idles through every epilogue.
                __builtin_amdgcn_s_setprio(PCL_FB_TWO_PRIO);
                f32x16 accx;
#pragma unroll
                for (int r = 0; r < 16; ++r) accx[r] = 0.f;
                const float* sAx = sDY + (rbx * 32 + lr) * DLD + lh * 4;
                const float* sBx = sWb + xcol * WLD + lh * 4;
                float4 xa = *reinterpret_cast<const float4*>(sAx), xb = *reinterpret_cast<const float4*>(sBx);
#pragma unroll
                for (int k8 = 0; k8 < COUT / 8; ++k8) {
                    float4 na = xa, nb = xb;
                    if (k8 + 1 < COUT / 8) { na = *reinterpret_cast<const float4*>(sAx + (k8 + 1) * 8); nb = *reinterpret_cast<const float4*>(sBx + (k8 + 1) * 8); }
                    side(k8);
                    __builtin_amdgcn_sched_barrier(0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, xb.x, accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, xb.y, accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, xb.z, accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, xb.w, accx, 0, 0, 0);
                    xa = na; xb = nb;
                }
                FB_MARK(2)
                const rsrc_t rD = buf_rsrc(p.dUprev, (size_t)(m0 + rbx * 32) * irow, (size_t)p.P * irow);
                float ts = 0.f, tq = 0.f;
                const float* const yb = sY + (rbx * 32 + 4 * lh) * YLD + xcol;
                const float piv = yb[0];
                const unsigned v0 = (unsigned)(4 * lh) * irow + (unsigned)xcol * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (r & 3) + 8 * (r >> 2);                  // + 4*lh + 32*rbx
                    const float y = yb[rl * YLD];
                    float c = accx[r];
                    c = fmaf(xsc, y, xsh) > 0.f ? c : c * p.pslope;
                    ts += c; tq = fmaf(c, y - piv, tq);
                    buf_st1(rD, v0, (unsigned)rl * irow, c);
                }
                st_s += (double)ts; st_q += (double)tq + (double)piv * (double)ts;
                __builtin_amdgcn_s_setprio(0);
                FB_MARK(3)
            } else {
                // ---- dW += dy^T z over the tile's R rows: this wave's TMW channel-interleaved tiles
                const float* sAw = sDY + lh * DLD + wa * TMW * 32 + TMW * lr;
                const float* sBw = sY + lh * YLD + wb * 32 + lr;
                float2 wa_ = *reinterpret_cast<const float2*>(sAw);
                float wb_ = sBw[0];
#pragma unroll
                for (int ks = 0; ks < R / 2; ++ks) {
                    float2 na = wa_; float nb = wb_;
                    if (ks + 1 < R / 2) { na = *reinterpret_cast<const float2*>(sAw + 2 * (ks + 1) * DLD); nb = sBw[2 * (ks + 1) * YLD]; }
                    if (ks % 2 == 0) side(ks / 2);
                    __builtin_amdgcn_sched_barrier(0);
                    const float t = fmaf(zsc[0], wb_, zsh[0]);
                    const float bv = fmaxf(t, t * p.pslope);
                    accw[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa_.x, bv, accw[0][0], 0, 0, 0);
                    accw[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa_.y, bv, accw[1][0], 0, 0, 0);
                    wa_ = na; wb_ = nb;
                }
                FB_MARK(4)
            }
            __syncthreads();                                                // B: image `cur` is free, image `nxt` holds tile t + 1's dense part
            FB_MARK(5)
        }
    } else {
    if (tile < tiles) {
        meta_store(0, record_of(tile));
        rM = record_of(tile + p.gx);
        if constexpr (WRES) { w_request(0); w_deposit(0); }
        __syncthreads();
        const ReqCtx c0 = req_open(tile, 0, true);
        req_slice(c0, 0, 1);
    }
#if PCL_EXP == 7
    tk[1] = tlast = __builtin_readcyclecounter();
#endif
    for (; tile < tiles; tile += p.gx, ++it) {
        const int m0 = tile * R, mb = it & 1;
        const bool more = tile + p.gx < tiles;
        // ---- LDS image of this tile, row records of the next one
        FB_MARK(6)
        meta_store(mb ^ 1, rM);                                             // (its last readers passed barrier B of the tile before)
        deposit(mb);
        rM = record_of(tile + 2 * p.gx);
        if constexpr (SPARSE) {
            // ---- the winners: dy[row of slot arg[g][c] of group g][c] += a[c] gz[g][c], GP groups per round
            __syncthreads();                                                // A0: the dense part is complete
            int gf, gl;
            tile_groups(tile, mb, gf, gl);
            const int* dl = sDelta + mb * 128;
            auto fix = [&](int g, int srow, float gzv) {
                const int r = srow + dl[g & 127];
                if (g <= gl && (unsigned)r < (unsigned)R) {
                    float* q = sDY + r * DLD + fxc;
                    *q = fmaf(fxa, gzv, *q);
                }
            };
#pragma unroll
            for (int j = 0; j < NFX; ++j)
                if (gf + j * GP <= gl) fix(gf + j * GP + fxg, fxA[j], fxG[j]);
            for (int g0 = gf + NFX * GP; g0 <= gl; g0 += GP) {              // (rare) groups beyond the prefetched rounds
                const int g = g0 + fxg;
                const unsigned go = g <= gl ? ((unsigned)g * COUT + (unsigned)fxc) * 4u : BUF_OOB;
                fix(g, __builtin_bit_cast(int, buf_ld1(rArg, go, 0)), buf_ld1(rGz, go, 0));
            }
        }
        FB_MARK(0)
        __syncthreads();                                                    // A: sDY, sY, sMeta[next] complete
        FB_MARK(1)
        const ReqCtx cn = req_open(tile + p.gx, mb ^ 1, more);              // the next tile: requested under this tile's MFMAs
        // ---- dX = dy W
        auto phase_dx = [&]() {
        f32x16 accx;
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[r] = 0.f;
        // (operands of step k+1 are read from LDS before the MFMAs of step k are issued -- pinned with sched_barrier: left to
        // itself hipcc sinks every load down to its first use and the LDS / L2 latency of each step is exposed)
        const float* sAx = sDY + (rbx * 32 + lr) * DLD + lh * 4;
        struct XOp { float4 a, b; };
        auto x_load = [&](const float* pa, const float* pb) -> XOp {
            XOp o;
            o.a = *reinterpret_cast<const float4*>(pa);
            o.b = *reinterpret_cast<const float4*>(pb);
            return o;
        };
        auto x_mfma = [&](const XOp& o) {
            accx = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a.x, o.b.x, accx, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a.y, o.b.y, accx, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a.z, o.b.z, accx, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a.w, o.b.w, accx, 0, 0, 0);
        };
        if constexpr (WRES) {
            // lane (lr, lh) of MFMA j of step k8 contracts k = 8 k8 + 4 lh + j: A = dy[its row][k], B = W[k][its column] = sW[xcol][k]
            const float* sBx = sWb + xcol * WLD + lh * 4;
            XOp cur = x_load(sAx, sBx);
#pragma unroll
            for (int k8 = 0; k8 < COUT / 8; ++k8) {
                XOp nxt = cur;
                if (k8 + 1 < COUT / 8) nxt = x_load(sAx + (k8 + 1) * 8, sBx + (k8 + 1) * 8);
                req_slice(cn, k8, COUT / 8);
                __builtin_amdgcn_sched_barrier(0);
                x_mfma(cur);
                cur = nxt;
            }
        } else {
            // B from L2: lane (lr, lh) of MFMA j of step k8 contracts k = 8*k8 + 4*lh + j, i.e. needs W[8*k8 + 4*lh + j][xcol]
            // (the row goes into the scalar offset, one per-lane offset for the whole loop); ring of DB steps in flight
            constexpr int DB = 3, NK8 = COUT / 8;
            const unsigned vB = (unsigned)(lh * 4) * irow + (unsigned)xcol * 4u;
            auto b_load = [&](int k8) -> float4 {
                float4 b;
                b.x = buf_ld1(rW, vB, (unsigned)(k8 * 8 + 0) * irow); b.y = buf_ld1(rW, vB, (unsigned)(k8 * 8 + 1) * irow);
                b.z = buf_ld1(rW, vB, (unsigned)(k8 * 8 + 2) * irow); b.w = buf_ld1(rW, vB, (unsigned)(k8 * 8 + 3) * irow);
                return b;
            };
            float4 bq[DB];
#pragma unroll
            for (int d = 0; d < DB; ++d) bq[d] = b_load(d);
            float4 acur = *reinterpret_cast<const float4*>(sAx);
#pragma unroll
            for (int k8 = 0; k8 < NK8; ++k8) {
                float4 anxt = acur;
                if (k8 + 1 < NK8) anxt = *reinterpret_cast<const float4*>(sAx + (k8 + 1) * 8);
                const float4 b = bq[k8 % DB];
                __builtin_amdgcn_sched_barrier(0);
                accx = __builtin_amdgcn_mfma_f32_32x32x2f32(acur.x, b.x, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x2f32(acur.y, b.y, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x2f32(acur.z, b.z, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x2f32(acur.w, b.w, accx, 0, 0, 0);
                if (k8 + DB < NK8) bq[k8 % DB] = b_load(k8 + DB);
                acur = anxt;
            }
        }
        FB_MARK(2)
        // ---- dX epilogue: mask with relu'(BN(Yprev)), sums for the BatchNorm below, store in the C/D layout
        {
            // row block rbx goes into the descriptor's base and the row within it into the scalar offset (compile-time); rows past P
            // are dropped by the range check.  LDS: one base pointer + compile-time steps that fold into the DS offset field (sixteen
            // addresses computed from rbx were hoisted out of the tile loop as 64-bit values and spilled).
            const rsrc_t rD = buf_rsrc(p.dUprev, (size_t)(m0 + rbx * 32) * irow, (size_t)p.P * irow);
            float ts = 0.f, tq = 0.f;
            const float* const yb = sY + (rbx * 32 + 4 * lh) * YLD + xcol;
            const float piv = yb[0];
            const unsigned v0 = (unsigned)(4 * lh) * irow + (unsigned)xcol * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2);                      // + 4*lh + 32*rbx
                const float y = yb[rl * YLD];
                float c = accx[r];
                c = fmaf(xsc, y, xsh) > 0.f ? c : c * p.pslope;
                ts += c; tq = fmaf(c, y - piv, tq);
                buf_st1(rD, v0, (unsigned)rl * irow, c);
            }
            asm volatile("" : "+v"(ts), "+v"(tq));   // (else hipcc sinks these sums below the dW loop and carries 32 values through it)
            st_s += (double)ts; st_q += (double)tq + (double)piv * (double)ts;
        }
        FB_MARK(3)
        };
        // ---- dW += dy^T z over this wave group's rows of the tile (rows past P have dy = 0)
        auto phase_dw = [&]() {
        {
            // lane (lr, lh) of step ks contracts row kq KR + 2 ks + lh: A = dy[row][channels TMW lr .. + TMW - 1 of block wa] (one LDS
            // read for the TMW tiles), B = z[row][column lr of block wb]
            const float* sAw = sDY + (kq * KR + lh) * DLD + wa * TMW * 32 + TMW * lr;
            const float* sBw = sY + (kq * KR + lh) * YLD + wb * TNW * 32 + lr;
            struct WOp { float a[TMW], b[TNW]; };
            auto w_load = [&](int ks) -> WOp {
                WOp o;
                if constexpr (TMW == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(sAw + 2 * ks * DLD);
                    o.a[0] = v.x; o.a[1] = v.y; o.a[2] = v.z; o.a[3] = v.w;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(sAw + 2 * ks * DLD);
                    o.a[0] = v.x; o.a[1] = v.y;
                }
#pragma unroll
                for (int b = 0; b < TNW; ++b) o.b[b] = sBw[2 * ks * YLD + b * 32];
                return o;
            };
            WOp cur = w_load(0);
#pragma unroll
            for (int ks = 0; ks < KR / 2; ++ks) {
                WOp nxt = cur;
                if (ks + 1 < KR / 2) nxt = w_load(ks + 1);
                if constexpr (!WRES && SPARSE) { if (ks < KR / 4) req_slice(cn, ks, KR / 4); }    // (weight from L2: first half of the loop, see header)
                if constexpr (!WRES && !SPARSE) req_slice(cn, ks, KR / 2);
                __builtin_amdgcn_sched_barrier(0);
                float bv[TNW];
#pragma unroll
                for (int b = 0; b < TNW; ++b) {
                    const float t = fmaf(zsc[b], cur.b[b], zsh[b]);
                    bv[b] = fmaxf(t, t * p.pslope);
                }
#pragma unroll
                for (int a = 0; a < TMW; ++a)
#pragma unroll
                    for (int b = 0; b < TNW; ++b) accw[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[a], bv[b], accw[a][b], 0, 0, 0);
                cur = nxt;
            }
        }
        FB_MARK(4)
        };
        // The two phases only read the tile's LDS image, in any order (no barrier inside either): waves 4-7
        // -- the second wave of every SIMD -- take dW first, so that one wave's epilogue (VALU, LDS reads, stores; no MFMA)
        // runs under the other's MFMAs instead of both leaving the matrix pipe idle at the same time.
        // (dense 256 x 128 -- no BASELINE network has it -- keeps one order: with dU its prefetch is 96 registers, which the second
        //  order would hold across the dX loop; they do not exist)
        constexpr bool STAGGER = WRES || SPARSE;
        if (!STAGGER || wave < 4) { phase_dx(); phase_dw(); }
        else { phase_dw(); phase_dx(); }
        __syncthreads();                                                    // B: every wave is done with sDY / sY / sMeta[mb]
        FB_MARK(5)
    }
#if PCL_EXP == 7
    tk[2] = __builtin_readcyclecounter();
#endif
    }
    // ---- this workgroup's partial dW tile and its row of the BatchNorm sums
    // accumulator r of tile a: output channel wa 32 TMW + TMW i + a with i = (r & 3) + 8 (r >> 2) + 4 lh (the MFMA's row), column wb 32 + lr
    float* out = p.part + (size_t)bx * COUT * CIN;
    if constexpr (KW == 1) {
        if (!TWO || wave >= 4)                                  // (TWO: the waves of the dW role hold the accumulators)
#pragma unroll
        for (int a = 0; a < TMW; ++a) {
            const int col = wb * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(wa * TMW * 32 + TMW * ((r & 3) + 8 * (r >> 2) + 4 * lh) + a) * CIN + col] = accw[a][0][r];
        }
    } else {
        // the KW row groups' partial tiles meet in LDS (the tile image is dead: barrier B of the last tile was passed), summed in the order
        // of the groups and written once, 16 bytes per thread and store
        float* const sRed = lds;
#pragma unroll
        for (int a = 0; a < TMW; ++a) {
            const int col = wb * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sRed[(size_t)kq * COUT * CIN + (size_t)(wa * TMW * 32 + TMW * ((r & 3) + 8 * (r >> 2) + 4 * lh) + a) * CIN + col] = accw[a][0][r];
        }
        __syncthreads();
        for (int e = tid; e < COUT * CIN / 4; e += FB_T) {
            float4 v = reinterpret_cast<const float4*>(sRed)[e];
#pragma unroll
            for (int q = 1; q < KW; ++q) {
                const float4 u = reinterpret_cast<const float4*>(sRed + (size_t)q * COUT * CIN)[e];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            reinterpret_cast<float4*>(out)[e] = v;
        }
    }
    {
        double s = st_s, q = st_q;
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        double* red = reinterpret_cast<double*>(lds);           // [8 waves][32][2]
        __syncthreads();
        if (lh == 0) { red[(wave * 32 + lr) * 2] = s; red[(wave * 32 + lr) * 2 + 1] = q; }
        __syncthreads();
        if (tid < CIN) {
            const int cb_ = tid / 32, l = tid & 31;
            double ss = 0.0, qq = 0.0;
#pragma unroll
            for (int w = 0; w < R / 32; ++w) { ss += red[((w * CB + cb_) * 32 + l) * 2]; qq += red[((w * CB + cb_) * 32 + l) * 2 + 1]; }
            double* dst = p.stats + (size_t)bx * 2 * CIN;
            dst[tid] = ss; dst[CIN + tid] = qq;
        }
    }
#if PCL_EXP == 7
    tk[3] = __builtin_readcyclecounter();                     // (stores issued, not drained)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tk[4] = __builtin_readcyclecounter();
    const unsigned long long rt_exit = __builtin_amdgcn_s_memrealtime();
    constexpr int SH = fbk_shape(SPARSE, CO, CI);
    if (SH >= 0 && bx < 256 && (tid == 0 || tid == 448)) {
        long long* q = g_fbk[SH < 0 ? 0 : SH][p.lab_slot & (FBK_RING - 1)][bx][tid == 0 ? 0 : 1];
        q[0] = (long long)rt_entry; q[1] = (long long)rt_exit;
#pragma unroll
        for (int i = 0; i < 5; ++i) q[2 + i] = tk[i];
#pragma unroll
        for (int i = 0; i < 7; ++i) q[7 + i] = tph[i];
        q[14] = it; q[15] = ((long long)p.lab_slot << 8) | wave;
    }
#endif
}

// one persistent workgroup per CU (the device's CU count, asked once); pcl_set_fb_max_blocks caps it (tests: several row tiles
// per workgroup at sizes small enough for an fp64 comparison free of ReLU-mask flips; tuning).  The cap is a process-wide
// setting made BETWEEN calls: it sizes the statistics rows, the workspace and the finish reduction of every later call.
static int g_fb_cap = 0;
static int g_fb_two = 1;                 // the 128 x 64 shape on two tile images (pcl_set_fb_two_images)
static int fb_cu_count() {
    static const int n = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        (void)hipGetLastError();
        return cu > 256 ? 256 : cu;          // (the partial-tile workspace and the 256-row statistics are sized for <= 256)
    }();
    return n;
}
static int fr_cu_count() { return fb_cu_count(); }
static int fb_grid(int P, int Cin) {
    const int R = fb_rows(Cin), tiles = (P + R - 1) / R;
    const int cus = fb_cu_count();
    int gx = g_fb_cap >= 1 && g_fb_cap < cus ? g_fb_cap : cus;
    if (gx > tiles) gx = tiles;
    if (gx < 1) gx = 1;
    return gx;
}

// ---- BatchNorm bookkeeping (tiny kernels, one thread per channel) ---------------------------------------
// mean/var from the fp64 partials; folded scale/shift; Jittor-style running statistics (biased variance).
// column sums of a [rows][2][C] fp64 partial workspace: 4 channels x 64 row-lanes per 256-thread block
__device__ __forceinline__ void stat_colsum(const double* __restrict__ stats, int rows, int C, int c, int ry, double& s,
                                            double& q, double* red /*[2][4][4]*/) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    if (c < C) {
        const double* b = stats + c;
        const size_t st = (size_t)2 * C;
        int r = ry;
        for (; r + 192 < rows; r += 256) {          // four independent row streams in flight per lane
            s0 += b[(size_t)r * st]; q0 += b[(size_t)r * st + C];
            s1 += b[(size_t)(r + 64) * st]; q1 += b[(size_t)(r + 64) * st + C];
            s2 += b[(size_t)(r + 128) * st]; q2 += b[(size_t)(r + 128) * st + C];
            s3 += b[(size_t)(r + 192) * st]; q3 += b[(size_t)(r + 192) * st + C];
        }
        for (; r < rows; r += 64) { s0 += b[(size_t)r * st]; q0 += b[(size_t)r * st + C]; }
    }
    s = (s0 + s1) + (s2 + s3); q = (q0 + q1) + (q2 + q3);
    // thread = ry*4 + channel-in-block: a wave holds 16 row-lanes of each of the 4 channels (lane bits 2..5)
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    const int cl = threadIdx.x & 3, wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 4) { red[wave * 4 + cl] = s; red[16 + wave * 4 + cl] = q; }
    __syncthreads();
    s = (red[cl] + red[4 + cl]) + (red[8 + cl] + red[12 + cl]);
    q = (red[16 + cl] + red[20 + cl]) + (red[24 + cl] + red[28 + cl]);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ stats, int rows, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, int P, int C, float eps, float momentum,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
    __shared__ double red[2 * 64 * 4];
    const int c = blockIdx.x * 4 + (threadIdx.x & 3), ry = threadIdx.x >> 2;
    double s, q;
    stat_colsum(stats, rows, C, c, ry, s, q, red);
    if (ry != 0 || c >= C) return;
    const double mean = s / P;
    double var = q / P - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float a = g * invstd;
    scale[c] = a;
    shift[c] = b - a * (float)mean;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean) running_mean[c] += ((float)mean - running_mean[c]) * momentum;
    if (running_var) running_var[c] += ((float)var - running_var[c]) * momentum;
}

// BatchNorm backward constants from sum(du) and sum(du*y):
//   dbeta = S1, dgamma = (S2 - mean*S1)*invstd,  dy = a*du - k1 - k2*(y - mean) with a = gamma*invstd,
//   k2 = a*dgamma*invstd/P, k1 = a*dbeta/P   (y is centred where it is used: |mean| >> std must not cost digits).
struct BnConstsArgs {
    const double* stats; int rows; const float* gamma; const float* mean; const float* invstd; int P, C;
    float* dgamma; float* dbeta; float* a_out; float* k1; float* k2; float* dbias_zero;
};
__device__ __forceinline__ void bn_bwd_consts_block(const BnConstsArgs& q, int block, double* red /*[2*64*4]*/) {
    const double* __restrict__ stats = q.stats; const int rows = q.rows, P = q.P, C = q.C;
    const float* __restrict__ gamma = q.gamma; const float* __restrict__ mean = q.mean; const float* __restrict__ invstd = q.invstd;
    float* __restrict__ dgamma = q.dgamma; float* __restrict__ dbeta = q.dbeta; float* __restrict__ a_out = q.a_out;
    float* __restrict__ k1 = q.k1; float* __restrict__ k2 = q.k2; float* __restrict__ dbias_zero = q.dbias_zero;
    const int c = block * 4 + (threadIdx.x & 3), ry = threadIdx.x >> 2;
    double s1, s2;
    stat_colsum(stats, rows, C, c, ry, s1, s2, red);
    if (ry != 0 || c >= C) return;
    const double mu = mean[c], is = invstd[c];
    const double g = gamma ? gamma[c] : 1.0;
    const double dg = (s2 - mu * s1) * is;
    const double a = g * is;
    const double kk2 = a * dg * is / P;
    if (dgamma) dgamma[c] = (float)dg;
    if (dbeta) dbeta[c] = (float)s1;
    if (dbias_zero) dbias_zero[c] = 0.f;         // a conv bias under training-mode BatchNorm has an exactly zero gradient
    a_out[c] = (float)a;
    k2[c] = (float)kk2;
    k1[c] = (float)(a * s1 / P);
}
__global__ __launch_bounds__(256) void bn_bwd_consts_kernel(const BnConstsArgs q) {
    __shared__ double red[2 * 64 * 4];
    bn_bwd_consts_block(q, blockIdx.x, red);
}
// ---- few-row layers (round 5): BatchNorm-backward constants AND dy of a layer in one launch ---------------------------------------------
// On a few thousand rows (the GroupAll level: 4 096; the part-seg decoder: 2 048 .. 8 192) the two GEMMs of a layer's backward re-form
// dy = a du - k1 - k2 (y - mean) in their loaders once per output tile that reads it: 8 times for a 512-wide dX, again per 128-column
// slab of dW, and every one of those vector instructions is matrix time on gfx950 (fp32 MFMA and VALU share the issue port: co-running
// dW beside dX on a second stream changed nothing, 104.7 + 63.0 us against 51.8 and 52.3 alone).  Here dy is formed ONCE, by the launch
// that used to compute only the constants: a block owns 32 channels x a slab of rows, sums the partial rows of its channels
// (stat_colsum's order, four channels at a time: the constants are bit-identical to pcl_bn_bwd_consts_f32's) and writes dy; the row slab 0
// blocks also write dgamma / dbeta / the constants.  The GEMMs behind it read plain operands.
struct BnDyArgs {
    BnConstsArgs q;
    const float* dU; const float* Y; const int32_t* arg; const float* gz; int ns;      // dU dense, or (arg, gz) of the max pool
    float* dy; int M;
};
__global__ __launch_bounds__(256) void bn_bwd_dy_kernel(const BnDyArgs p) {
    __shared__ double red[8][32];
    __shared__ __attribute__((aligned(16))) float sk[4][32];      // a, k1, k2, mean of this block's 32 channels
    const BnConstsArgs& q = p.q;
    const int c0 = blockIdx.x * 32;
    {
        // stat_colsum's summation order for each channel (row lane ry sums rows ry, ry + 64, ... in four streams; 16 row lanes of a wave by
        // xor shuffles; four waves through LDS), with the partial rows of all eight channel quads requested together: one memory round trip
        // instead of eight
        const int cl = threadIdx.x & 3, ry = threadIdx.x >> 2, wave = threadIdx.x >> 6;
        const size_t st = (size_t)2 * q.C;
        double s[8], t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + 4 * j + cl;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
            if (c < q.C) {
                const double* b = q.stats + c;
                int r = ry;
                for (; r + 192 < q.rows; r += 256) {
                    s0 += b[(size_t)r * st]; q0 += b[(size_t)r * st + q.C];
                    s1 += b[(size_t)(r + 64) * st]; q1 += b[(size_t)(r + 64) * st + q.C];
                    s2 += b[(size_t)(r + 128) * st]; q2 += b[(size_t)(r + 128) * st + q.C];
                    s3 += b[(size_t)(r + 192) * st]; q3 += b[(size_t)(r + 192) * st + q.C];
                }
                for (; r < q.rows; r += 64) { s0 += b[(size_t)r * st]; q0 += b[(size_t)r * st + q.C]; }
            }
            s[j] = (s0 + s1) + (s2 + s3); t[j] = (q0 + q1) + (q2 + q3);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) { s[j] += __shfl_xor(s[j], off); t[j] += __shfl_xor(t[j], off); }
            if ((threadIdx.x & 63) < 4) { red[j][wave * 4 + cl] = s[j]; red[j][16 + wave * 4 + cl] = t[j]; }
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int j = threadIdx.x >> 2, c = c0 + threadIdx.x;              // (threadIdx.x = 4 j + cl)
            if (c < q.C) {
                const double* r = red[j];
                const double s1 = (r[cl] + r[4 + cl]) + (r[8 + cl] + r[12 + cl]);
                const double s2 = (r[16 + cl] + r[20 + cl]) + (r[24 + cl] + r[28 + cl]);
                const double mu = q.mean[c], is = q.invstd[c];
                const double g = q.gamma ? q.gamma[c] : 1.0;
                const double dg = (s2 - mu * s1) * is;
                const double a = g * is;
                const double kk2 = a * dg * is / q.P;
                const float fa = (float)a, fk2 = (float)kk2, fk1 = (float)(a * s1 / q.P);
                sk[0][threadIdx.x] = fa; sk[1][threadIdx.x] = fk1; sk[2][threadIdx.x] = fk2; sk[3][threadIdx.x] = q.mean[c];
                if (blockIdx.y == 0) {
                    if (q.dgamma) q.dgamma[c] = (float)dg;
                    if (q.dbeta) q.dbeta[c] = (float)s1;
                    if (q.dbias_zero) q.dbias_zero[c] = 0.f;
                    q.a_out[c] = fa; q.k2[c] = fk2; q.k1[c] = fk1;
                }
            }
        }
    }
    __syncthreads();
    // rows of this block's slab, 32 per pass: thread = (row-in-pass, channel quad)
    const int cq = (threadIdx.x & 7) * 4, c = c0 + cq;
    if (c >= q.C) return;                                          // (C % 4 == 0: a quad is inside or outside)
    const float4 a = *reinterpret_cast<const float4*>(&sk[0][cq]), k1 = *reinterpret_cast<const float4*>(&sk[1][cq]);
    const float4 k2 = *reinterpret_cast<const float4*>(&sk[2][cq]), mu = *reinterpret_cast<const float4*>(&sk[3][cq]);
    const int per = (p.M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(p.M, r0 + per);
#pragma unroll 2
    for (int m = r0 + (threadIdx.x >> 3); m < r1; m += 32) {
        const size_t off = (size_t)m * q.C + c;
        const float4 y = *reinterpret_cast<const float4*>(p.Y + off);
        float4 du;
        if (p.dU) du = *reinterpret_cast<const float4*>(p.dU + off);
        else {
            const int g = m / p.ns, srow = m - g * p.ns;
            const int4 ar = *reinterpret_cast<const int4*>(p.arg + (size_t)g * q.C + c);
            const float4 gz = *reinterpret_cast<const float4*>(p.gz + (size_t)g * q.C + c);
            du.x = ar.x == srow ? gz.x : 0.f; du.y = ar.y == srow ? gz.y : 0.f; du.z = ar.z == srow ? gz.z : 0.f; du.w = ar.w == srow ? gz.w : 0.f;
        }
        float4 d;        // (the arithmetic of the dy-forming loaders: the staged kernels and this path agree bit for bit on dy)
        d.x = fmaf(a.x, du.x, -fmaf(k2.x, y.x - mu.x, k1.x)); d.y = fmaf(a.y, du.y, -fmaf(k2.y, y.y - mu.y, k1.y));
        d.z = fmaf(a.z, du.z, -fmaf(k2.z, y.z - mu.z, k1.z)); d.w = fmaf(a.w, du.w, -fmaf(k2.w, y.w - mu.w, k1.w));
        *reinterpret_cast<float4*>(p.dy + off) = d;
    }
}

// second (and last) launch of the fused backward of a layer: blocks [0, nred) sum the workgroups' partial dW tiles (as
// reduce_rows_kernel), the blocks after them turn the BatchNorm sums the fused kernel left into the constants of the layer below.
// a third kind of block (round 6): the coordinate columns of a folded first layer's weight gradient, dW0[c, 0..2] = sum_r px[r][c][0..2] -- the
// reduce that followed the folded layer's backward as a launch of its own (csrc/compact.hip: group_linear_dw_kernel, same sum in the same order)
struct FinishXJob { const float* px; int rows, C1, ld; float* dW0; int first; };
__global__ __launch_bounds__(256) void fused_finish_kernel(const float* __restrict__ part, int rows, size_t n, int ncols, float* __restrict__ out,
                                                           int nred, const BnConstsArgs q, int ldo = 0, const FinishXJob xj = FinishXJob{nullptr, 0, 0, 0, nullptr, 0}) {
    __shared__ double red[2 * 64 * 4];
    if (xj.px && (int)blockIdx.x >= xj.first) {
        const int nx = xj.C1 * 3, e = ((int)blockIdx.x - xj.first) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (e >= nx) return;
        float s = 0.f;
        for (int r = lane; r < xj.rows; r += 64) s += xj.px[(size_t)r * nx + e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) xj.dW0[(size_t)(e / 3) * xj.ld + e % 3] = s;
        return;
    }
    if ((int)blockIdx.x >= nred) { bn_bwd_consts_block(q, blockIdx.x - nred, red); return; }
    float* redf = reinterpret_cast<float*>(red);            // [8][33]
    const int el = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const size_t e = (size_t)blockIdx.x * 32 + el;
    float acc = 0.f;
    if (e < n) {
#pragma unroll 8
        for (int r = ry; r < rows; r += 8) acc += part[(size_t)r * n + e];
    }
    redf[ry * 33 + el] = acc;
    __syncthreads();
    if (ry == 0 && e < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += redf[j * 33 + el];
        out[ldo == 0 || ldo == ncols ? e : (e / ncols) * ldo + e % ncols] = t;      // (ldo: the tile is a column block of a wider matrix)
    }
}

// out[g,c] = max_s lrelu(scale*y+shift); arg = first s attaining it; ymax = y at arg.  One thread per (g,c),
// lanes along c (coalesced 4 B x 64 = 256 B per row segment).
__global__ __launch_bounds__(256) void bn_act_max_kernel(const float* __restrict__ Y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float slope, int G, int ns,
                                                         int C, float* __restrict__ out, int32_t* __restrict__ arg,
                                                         float* __restrict__ ymax) {
    const size_t total = (size_t)G * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t g = e / C;
        const int c = (int)(e - g * C);
        const float a = scale[c], b = shift[c];
        const float* y = Y + g * ns * C + c;
        float best = -INFINITY, by = 0.f;
        int bi = 0;
        for (int s = 0; s < ns; ++s) {
            const float yy = y[(size_t)s * C];
            const float z = lrelu(fmaf(a, yy, b), slope);
            if (z > best) { best = z; bi = s; by = yy; }
        }
        out[e] = best;
        if (arg) arg[e] = bi;
        if (ymax) ymax[e] = by;
    }
}

// Same for few, long groups (PointNet: G = B, ns = N): 64 channels x SL row slices per workgroup (slice sl takes rows sl, sl+SL, ..),
// slices combined through LDS -- equal values: the lower row wins, as in the sequential scan.  The one-thread-per-(g,c)
// form above runs 8192 threads for B = 8, C = 1024 and took 288 us of PointNet's 0.84 ms step.
template <int SL>
__global__ __launch_bounds__(64 * SL) void bn_act_max_sliced_kernel(const float* __restrict__ Y, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, float slope, int ns, int C,
                                                                    float* __restrict__ out, int32_t* __restrict__ arg,
                                                                    float* __restrict__ ymax) {
    __shared__ float sz[SL][64], sy[SL][64];
    __shared__ int ss[SL][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, g = blockIdx.y;
    float best = -INFINITY, by = 0.f;
    int bi = 0;
    if (c < C) {
        const float a = scale[c], b = shift[c];
        const float* y = Y + (size_t)g * ns * C + c;
        for (int s = sl; s < ns; s += SL) {
            const float yy = y[(size_t)s * C];
            const float z = lrelu(fmaf(a, yy, b), slope);
            if (z > best) { best = z; bi = s; by = yy; }
        }
    }
    sz[sl][cl] = best; sy[sl][cl] = by; ss[sl][cl] = bi;
    __syncthreads();
    if (sl != 0 || c >= C) return;
#pragma unroll
    for (int j = 1; j < SL; ++j) {
        const float z = sz[j][cl];
        const int s = ss[j][cl];
        if (z > best || (z == best && s < bi)) { best = z; bi = s; by = sy[j][cl]; }
    }
    const size_t e = (size_t)g * C + c;
    out[e] = best;
    if (arg) arg[e] = bi;
    if (ymax) ymax[e] = by;
}

// DGCNN's global pooling (networks/cls/dgcnn.py:114-116): max AND mean over the N points of a cloud of z = lrelu(scale*y + shift),
// straight from the pre-BatchNorm conv output -- the [B,N,C] activation (134 MB at B = 32, N = 1024, C = 1024) is never written,
// and neither are the two reads, the zero-filled scatter target, the broadcast and the sum of the composite's backward.
// out_max / out_mean rows are ldo floats apart (both halves of the concatenated [B, 2C] vector in one tensor).
template <int SL>
__global__ __launch_bounds__(64 * SL) void bn_act_maxmean_sliced_kernel(const float* __restrict__ Y, const float* __restrict__ scale,
                                                                        const float* __restrict__ shift, float slope, int ns, int C, int ldo,
                                                                        float* __restrict__ out_max, float* __restrict__ out_mean,
                                                                        int32_t* __restrict__ arg) {
    __shared__ float sz[SL][64];
    __shared__ int ss[SL][64];
    __shared__ double sm[SL][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, g = blockIdx.y;
    float best = -INFINITY;
    int bi = 0;
    double sum = 0.0;
    if (c < C) {
        const float a = scale[c], b = shift[c];
        const float* y = Y + (size_t)g * ns * C + c;
        for (int s = sl; s < ns; s += SL) {
            const float z = lrelu(fmaf(a, y[(size_t)s * C], b), slope);
            if (z > best) { best = z; bi = s; }
            sum += (double)z;
        }
    }
    sz[sl][cl] = best; ss[sl][cl] = bi; sm[sl][cl] = sum;
    __syncthreads();
    if (sl != 0 || c >= C) return;
#pragma unroll
    for (int j = 1; j < SL; ++j) {
        const float z = sz[j][cl];
        const int s = ss[j][cl];
        if (z > best || (z == best && s < bi)) { best = z; bi = s; }
        sum += sm[j][cl];
    }
    out_max[(size_t)g * ldo + c] = best;
    out_mean[(size_t)g * ldo + c] = (float)(sum / ns);
    arg[(size_t)g * C + c] = bi;
}

// fused max-pool finish: pick the extreme that maximises lrelu(scale*y+shift) (max for scale >= 0, min otherwise)
__global__ __launch_bounds__(256) void group_minmax_finalize_kernel(const float* __restrict__ gmax, const float* __restrict__ gmin,
                                                                    const int32_t* __restrict__ gamax, const int32_t* __restrict__ gamin,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    float slope, int C, size_t total, float* __restrict__ out,
                                                                    int32_t* __restrict__ arg, float* __restrict__ ymax,
                                                                    float* __restrict__ out2 = nullptr, int ld2 = 0) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float a = scale[c];
        const bool up = a >= 0.f;
        const float y = up ? gmax[e] : gmin[e];
        const float o = lrelu(fmaf(a, y, shift[c]), slope);
        out[e] = o;
        if (out2) out2[(e / C) * (size_t)ld2 + c] = o;        // a second copy as a column slice of a wider matrix (a concatenation written in place)
        arg[e] = up ? gamax[e] : gamin[e];
        ymax[e] = y;
    }
}

// gz[g,c] = gout[g,c] * act'(out[g,c]); per-workgroup partial sums of gz and gz*ymax -> stats rows.
// Thread layout of the two elementwise backward kernels below: CW = min(256, next power of two >= C) lanes over the
// channels, 256/CW rows per pass, so that narrow layers (WeightNet: 8 and 16 channels) still use every lane and a wave
// touches contiguous memory; the row-lanes of a block are folded through LDS into its one partial row.
__device__ __forceinline__ void fold_row_lanes(double s1, double s2, int c, int rsub, int RS, int CW, int C, double* red,
                                               double* __restrict__ dst) {
    if (RS > 1) {
        red[(size_t)rsub * CW + (threadIdx.x % CW)] = s1;
        red[(size_t)(RS + rsub) * CW + (threadIdx.x % CW)] = s2;
        __syncthreads();
        if (rsub == 0) {
            for (int j = 1; j < RS; ++j) { s1 += red[(size_t)j * CW + threadIdx.x]; s2 += red[(size_t)(RS + j) * CW + threadIdx.x]; }
        }
    }
    if (rsub == 0 && c < C) { dst[c] = s1; dst[C + c] = s2; }
}

// `aux`: side jobs of the stack's backward that would otherwise be launches of their own (every dependent launch costs ~9 us
// of a 2 ms step): zero-fill [zero, zero + n_zero) -- the target of the folded first layer's atomics -- and write the
// (a, k1, k2, mu) = (1, 0, 0, 0) constants of its plain point GEMMs to [ones, ones + n_one) / [ones + n_one, .. + n_one0).
struct PrepAux { float4* zero; size_t n_zero4; float* ones; int n_one, n_one0; };
__global__ __launch_bounds__(256) void maxgrad_prep_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                           const float* __restrict__ ymax, float slope, int G, int C, int CW, int ldg,
                                                           float* __restrict__ gz, double* __restrict__ stats, const PrepAux aux) {
    // grid.x = channel blocks of CW, grid.y = row slices (<= STAT_ROWS); gout rows are ldg floats apart (a slice of a wider
    // gradient: the consumer concatenated several stacks' outputs), everything else is dense [G, C]
    __shared__ double red[2 * 256];
    if (aux.zero || aux.ones) {
        const size_t nb = (size_t)gridDim.x * gridDim.y, b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        for (size_t e = b * 256 + threadIdx.x; e < aux.n_zero4; e += nb * 256) aux.zero[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b == 0 && aux.ones)
            for (int e = threadIdx.x; e < aux.n_one + aux.n_one0; e += 256) aux.ones[e] = e < aux.n_one ? 1.f : 0.f;
    }
    const int RS = 256 / CW, rsub = threadIdx.x / CW;
    const int c = blockIdx.x * CW + threadIdx.x % CW;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int g = blockIdx.y * RS + rsub; g < G; g += gridDim.y * RS) {
            const size_t e = (size_t)g * C + c;
            const float v = gout[(size_t)g * ldg + c] * (out[e] > 0.f ? 1.f : slope);
            gz[e] = v;
            s1 += v; s2 += (double)v * ymax[e];
        }
    fold_row_lanes(s1, s2, c, rsub, RS, CW, C, red, stats + (size_t)blockIdx.y * 2 * C);
}

// z = lrelu(scale*y+shift) on [P,C] (used where the activation itself is a module output)
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ Y, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float slope, int C, size_t total,
                                                     float* __restrict__ out) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        out[e] = lrelu(fmaf(scale[c], Y[e], shift[c]), slope);
    }
}

// du = gz * act'(scale*y+shift) on [P,C] with per-workgroup partials of sum(du), sum(du*y)
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ Y,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float slope, int P, int C, int CW, float* __restrict__ du,
                                                         double* __restrict__ stats) {
    __shared__ double red[2 * 256];
    const int RS = 256 / CW, rsub = threadIdx.x / CW;
    const int c = blockIdx.x * CW + threadIdx.x % CW;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const float a = scale[c], b = shift[c];
        for (int r = blockIdx.y * RS + rsub; r < P; r += gridDim.y * RS) {
            const size_t e = (size_t)r * C + c;
            const float y = Y[e];
            const float v = gz[e] * (fmaf(a, y, b) > 0.f ? 1.f : slope);
            du[e] = v;
            s1 += v; s2 += (double)v * y;
        }
    }
    fold_row_lanes(s1, s2, c, rsub, RS, CW, C, red, stats + (size_t)blockIdx.y * 2 * C);
}

// backward of bn_act_maxmean_sliced_kernel (DGCNN's global max + mean pooling): du[g*ns + s, c] = act'(scale*y + shift) * ([s == arg[g,c]] gmax[g,c] + gmean[g,c] / ns) with the
// BatchNorm-backward sums (sum du, sum du*y) as fp64 partial rows (the layout pcl_bn_bwd_consts_f32 reads)
__global__ __launch_bounds__(256) void bn_act_maxmean_bwd_kernel(const float* __restrict__ gmax, const float* __restrict__ gmean, int ldg,
                                                                 const int32_t* __restrict__ arg, const float* __restrict__ Y,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 float slope, int G, int ns, int C, int CW, float* __restrict__ du,
                                                                 double* __restrict__ stats) {
    __shared__ double red[2 * 256];
    const int RS = 256 / CW, rsub = threadIdx.x / CW;
    const int c = blockIdx.x * CW + threadIdx.x % CW;
    const int P = G * ns;
    const float inv_ns = 1.f / (float)ns;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const float a = scale[c], b = shift[c];
        for (int r = blockIdx.y * RS + rsub; r < P; r += gridDim.y * RS) {
            const int g = r / ns, srow = r - g * ns;
            const size_t e = (size_t)r * C + c;
            const float y = Y[e];
            const float gm = gmean[(size_t)g * ldg + c] * inv_ns;
            const float gz = arg[(size_t)g * C + c] == srow ? gmax[(size_t)g * ldg + c] + gm : gm;
            const float v = gz * (fmaf(a, y, b) > 0.f ? 1.f : slope);
            du[e] = v;
            s1 += v; s2 += (double)v * y;
        }
    }
    fold_row_lanes(s1, s2, c, rsub, RS, CW, C, red, stats + (size_t)blockIdx.y * 2 * C);
}

// ---- stand-alone training-mode BatchNorm over the rows of x [P,C] (the BatchNorm AFTER an activation of PointCNN's Conv /
// SepConv / Dense blocks, misc/layers.py:151-169,:173-206): column sums as fp64 partial rows, the same finalize / constants
// kernels as the stacks, one apply pass each way.
//   forward : (sum x, sum x^2)   -> pcl_bn_finalize_f32 -> out = scale*x + shift          (pcl_bn_act_f32 with slope 1)
//   backward: (sum g, sum g*x)   -> pcl_bn_bwd_consts_f32 -> dx = a*g - k1 - k2*(x - mean)
__global__ __launch_bounds__(256) void bn_rows_stats_kernel(const float* __restrict__ x, const float* __restrict__ g, int P, int C, int CW,
                                                            double* __restrict__ stats) {
    __shared__ double red[2 * 256];
    const int RS = 256 / CW, rsub = threadIdx.x / CW;
    const int c = blockIdx.x * CW + threadIdx.x % CW;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        // four independent streams per thread: the fp64 adds are a dependent chain otherwise
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        const int step = gridDim.y * RS;
        int r = blockIdx.y * RS + rsub;
        for (; r + step < P; r += 2 * step) {
            const size_t e0 = (size_t)r * C + c, e1 = (size_t)(r + step) * C + c;
            const float x0 = x[e0], x1 = x[e1];
            const float u0 = g ? g[e0] : x0, u1 = g ? g[e1] : x1;
            a0 += u0; b0 += (double)u0 * x0;
            a1 += u1; b1 += (double)u1 * x1;
        }
        if (r < P) { const size_t e0 = (size_t)r * C + c; const float x0 = x[e0], u0 = g ? g[e0] : x0; a0 += u0; b0 += (double)u0 * x0; }
        s1 = a0 + a1; s2 = b0 + b1;
    }
    fold_row_lanes(s1, s2, c, rsub, RS, CW, C, red, stats + (size_t)blockIdx.y * 2 * C);
}
__global__ __launch_bounds__(256) void bn_rows_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                const float* __restrict__ a, const float* __restrict__ k1,
                                                                const float* __restrict__ k2, const float* __restrict__ mean, int C,
                                                                size_t total, float* __restrict__ dx) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        dx[e] = fmaf(a[c], g[e], -k1[c]) - k2[c] * (x[e] - mean[c]);
    }
}

static void linear_grid(int M, int N, bool bwd, bool rag, int& gx, int& n_tiles, bool& narrow, bool& low) {
    // 128x64 block tile instead of 128x128 for narrow outputs -- and for the backward (dy-forming) loaders, whose
    // three operand streams + 128x128 accumulators do not fit 256 VGPRs at 2 waves/SIMD (47-97 spilled VGPRs
    // measured); re-reading the A operand from L2 for the second column tile is cheaper than the spills.
    int m_tiles = (M + 127) / 128;
    narrow = N <= 64 || bwd || m_tiles * ((N + 127) / 128) < 384;   // also: too few 128x128 tiles to fill 256 CUs
    const int tbn = narrow ? 64 : 128;
    n_tiles = (N + tbn - 1) / tbn;
    // small M (the GroupAll level: 4096 rows): 64-row tiles double the workgroups, two per CU hide each other's latency
    low = narrow && !rag && m_tiles * n_tiles < 768;
    if (low) m_tiles = (M + 63) / 64;
    gx = m_tiles < STAT_ROWS ? m_tiles : STAT_ROWS;
    const int want = (1024 + n_tiles - 1) / n_tiles;   // ~4 workgroups per CU in total
    if (gx > want) gx = want;
}

template <int AM, int EM, int GM = 0, bool RAG = false>
static int launch_linear_t(const LinArgs& a_in, hipStream_t st) {
    LinArgs a = a_in;
    const bool vec = (a.K % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0) && (!a.A2 || (reinterpret_cast<uintptr_t>(a.A2) & 15) == 0) &&
                     (AM < A_DY || a.ldb % 4 == 0);
    int gx, n_tiles; bool narrow, low;
    linear_grid(a.M, a.N - a.n_begin, AM >= A_DY, RAG, gx, n_tiles, narrow, low);
    a.gx = gx; a.nt = n_tiles;
    dim3 grid(gx * n_tiles);
    if constexpr (GM == 0) {                   // matrix-bound shapes: fp32 operands as three bf16 planes on the bf16 matrix pipe
        if (vec && (g_split_mfma & 2) && a.K >= g_split_min_k) {
            if (!RAG && low) PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, true, 1, 0, false, 1, true>), grid, dim3(MLP_T), st, a);
            else if (narrow || AM >= A_DY) PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, true, 1, 0, RAG, 2, true>), grid, dim3(MLP_T), st, a);
            else { if constexpr (AM < A_DY) PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, true, 2, 0, RAG, 2, true>), grid, dim3(MLP_T), st, a); }
            return check_launch("pcl_linear(split)");
        }
    }
    if constexpr (!RAG && GM == 0) {
        if (low) {
            if (vec) PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, true, 1, 0, false, 1>), grid, dim3(MLP_T), st, a);
            else PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, false, 1, 0, false, 1>), grid, dim3(MLP_T), st, a);
            return check_launch("pcl_linear");
        }
    }
    // (the backward loaders always take 64-column tiles -- linear_grid: `narrow = ... || bwd` -- so their 128-column instantiations, which
    //  spilled up to 224 registers, are not compiled at all: round 5)
    if (narrow || AM >= A_DY) {
        if (vec) PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, true, 1, GM, RAG>), grid, dim3(MLP_T), st, a);
        else if constexpr (AM == A_DY_SPARSE && RAG && GM == 0) {
            // scalar staging of the sparse max gradient on compacted rows (Cout % 4 != 0): three operand streams per row for 128-row tiles
            // spilled 23 registers; 64-row tiles fit (the persistent loop and the statistics rows do not depend on the tile height)
            PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, false, 1, GM, RAG, 1>), grid, dim3(MLP_T), st, a);
        } else PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, false, 1, GM, RAG>), grid, dim3(MLP_T), st, a);
    } else {
        if constexpr (AM < A_DY) {
            if (vec) PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, true, 2, GM, RAG>), grid, dim3(MLP_T), st, a);
            else PCL_LAUNCH_TIMED((linear_nt_kernel<AM, EM, false, 2, GM, RAG>), grid, dim3(MLP_T), st, a);
        }
    }
    return check_launch("pcl_linear");
}

// number of stats rows a launch_linear_t call with M rows and N columns writes
static int linear_stat_rows(int M, int N, bool bwd, bool rag) {
    int gx, n_tiles; bool narrow, low;
    linear_grid(M, N, bwd, rag, gx, n_tiles, narrow, low);
    return gx;
}

static int launch_linear(const LinArgs& a, hipStream_t st) {
    if (fwd_res_eligible(a)) return launch_fwd_res(a, linear_stat_rows(a.M, a.N, false, a.rmeta != nullptr), st);
    if (a.rmeta) {                                     // duplicate-compacted rows
        if (a.a_mode == A_PLAIN && a.e_mode == E_STORE_STATS) return launch_linear_t<A_PLAIN, E_STORE_STATS, 0, true>(a, st);
        if (a.a_mode == A_BNACT && a.e_mode == E_STORE_STATS) return launch_linear_t<A_BNACT, E_STORE_STATS, 0, true>(a, st);
        if (a.a_mode == A_DY && a.e_mode == E_MASK_STORE_STATS) return launch_linear_t<A_DY, E_MASK_STORE_STATS, 0, true>(a, st);
        if (a.a_mode == A_DY && a.e_mode == E_STORE) return launch_linear_t<A_DY, E_STORE, 0, true>(a, st);
        if (a.a_mode == A_DY_SPARSE && a.e_mode == E_MASK_STORE_STATS) return launch_linear_t<A_DY_SPARSE, E_MASK_STORE_STATS, 0, true>(a, st);
        if (a.a_mode == A_DY_SPARSE && a.e_mode == E_STORE) return launch_linear_t<A_DY_SPARSE, E_STORE, 0, true>(a, st);
        return fail(PCL_EINVAL, "pcl_linear: unsupported ragged mode combination %d/%d", a.a_mode, a.e_mode);
    }
    if (a.e_mode == E_STORE_STATS && a.gmax) {
        if (a.a_mode == A_PLAIN && a.ns == 64) return launch_linear_t<A_PLAIN, E_STORE_STATS, 64>(a, st);
        if (a.a_mode == A_BNACT && a.ns == 64) return launch_linear_t<A_BNACT, E_STORE_STATS, 64>(a, st);
        if (a.a_mode == A_PLAIN && a.ns == 32) return launch_linear_t<A_PLAIN, E_STORE_STATS, 32>(a, st);
        if (a.a_mode == A_BNACT && a.ns == 32) return launch_linear_t<A_BNACT, E_STORE_STATS, 32>(a, st);
        return fail(PCL_EINVAL, "pcl_linear_fwd_gmax_f32: group size %d not supported (32 or 64)", a.ns);
    }
    if (a.a_mode == A_PLAIN && a.e_mode == E_STORE_STATS) return launch_linear_t<A_PLAIN, E_STORE_STATS>(a, st);
    if (a.a_mode == A_BNACT && a.e_mode == E_STORE_STATS) return launch_linear_t<A_BNACT, E_STORE_STATS>(a, st);
    if (a.a_mode == A_DY && a.e_mode == E_MASK_STORE_STATS) return launch_linear_t<A_DY, E_MASK_STORE_STATS>(a, st);
    if (a.a_mode == A_DY && a.e_mode == E_STORE) return launch_linear_t<A_DY, E_STORE>(a, st);
    if (a.a_mode == A_DY_SPARSE && a.e_mode == E_MASK_STORE_STATS) return launch_linear_t<A_DY_SPARSE, E_MASK_STORE_STATS>(a, st);
    if (a.a_mode == A_DY_SPARSE && a.e_mode == E_STORE) return launch_linear_t<A_DY_SPARSE, E_STORE>(a, st);
    return fail(PCL_EINVAL, "pcl_linear: unsupported mode combination %d/%d", a.a_mode, a.e_mode);
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_mlp_stat_rows(int P, int C, int flags) {
    if (P < 1 || C < 1) return 1;
    return linear_stat_rows(P, C, (flags & 1) != 0, (flags & 2) != 0);
}

extern "C" int pcl_linear_fwd_rows_f32(const float* X, const float* W, const float* bias, const float* in_scale,
                                       const float* in_shift, float in_slope, int P, int Cin, int Cout, float* Y,
                                       double* stats_ws, const int32_t* row_meta, const int32_t* n_rows_dev, void* stream) {
    PCL_REQUIRE(X && W && Y && stats_ws, "pcl_linear_fwd_f32: null pointer");
    PCL_REQUIRE(P >= 1 && Cin >= 1 && Cout >= 1, "pcl_linear_fwd_f32: bad sizes P=%d Cin=%d Cout=%d", P, Cin, Cout);
    PCL_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "pcl_linear_fwd_f32: in_scale/in_shift must come together");
    PCL_REQUIRE((row_meta == nullptr) == (n_rows_dev == nullptr), "pcl_linear_fwd_rows_f32: row_meta and n_rows_dev come together");
    LinArgs a = {};
    a.A = X; a.B = W; a.bias = bias; a.sc = in_scale; a.sh = in_shift; a.slope = in_slope;
    a.C = Y; a.stats = stats_ws; a.M = P; a.N = Cout; a.K = Cin; a.ldc = Cout;
    a.rmeta = reinterpret_cast<const int2*>(row_meta); a.m_dev = n_rows_dev;
    a.a_mode = in_scale ? A_BNACT : A_PLAIN; a.e_mode = E_STORE_STATS;
    return launch_linear(a, as_stream(stream));
}

extern "C" int pcl_linear_fwd_f32(const float* X, const float* W, const float* bias, const float* in_scale,
                                  const float* in_shift, float in_slope, int P, int Cin, int Cout, float* Y,
                                  double* stats_ws, void* stream) {
    return pcl_linear_fwd_rows_f32(X, W, bias, in_scale, in_shift, in_slope, P, Cin, Cout, Y, stats_ws, nullptr, nullptr, stream);
}

extern "C" int pcl_linear_fwd_gmax_f32(const float* X, const float* W, const float* bias, const float* in_scale,
                                       const float* in_shift, float in_slope, int P, int Cin, int Cout, int ns, float* Y,
                                       double* stats_ws, float* gmax, float* gmin, int32_t* gamax, int32_t* gamin,
                                       void* stream) {
    PCL_REQUIRE(X && W && Y && stats_ws && gmax && gmin && gamax && gamin, "pcl_linear_fwd_gmax_f32: null pointer");
    PCL_REQUIRE(P >= 1 && Cin >= 1 && Cout >= 1 && (ns == 32 || ns == 64) && P % ns == 0,
                "pcl_linear_fwd_gmax_f32: bad sizes P=%d Cin=%d Cout=%d ns=%d (ns must be 32 or 64 and divide P)", P, Cin, Cout, ns);
    PCL_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "pcl_linear_fwd_gmax_f32: in_scale/in_shift must come together");
    LinArgs a = {};
    a.A = X; a.B = W; a.bias = bias; a.sc = in_scale; a.sh = in_shift; a.slope = in_slope;
    a.C = Y; a.stats = stats_ws; a.M = P; a.N = Cout; a.K = Cin; a.ldc = Cout; a.ns = ns;
    a.gmax = gmax; a.gmin = gmin; a.gamax = gamax; a.gamin = gamin;
    a.a_mode = in_scale ? A_BNACT : A_PLAIN; a.e_mode = E_STORE_STATS;
    return launch_linear(a, as_stream(stream));
}

// the same, on 32 x 32 (group, channel) tiles, with a THIRD copy of `out` in the transposed layout out_t[b][c][n] (groups = the N points of
// B clouds, N % 32 == 0): the next EdgeConv stage's k-NN search reads [B, C, N] -- the transpose launch in front of it is gone (round 6)
__global__ __launch_bounds__(256) void group_minmax_finalize_t_kernel(const float* __restrict__ gmax, const float* __restrict__ gmin,
                                                                      const int32_t* __restrict__ gamax, const int32_t* __restrict__ gamin,
                                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                                      float slope, int C, int N, int tiles_c, size_t tiles, float* __restrict__ out,
                                                                      int32_t* __restrict__ arg, float* __restrict__ ymax,
                                                                      float* __restrict__ out2, int ld2, float* __restrict__ out_t) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t g0 = (t / tiles_c) * 32;
        const int c0 = (int)(t % tiles_c) * 32, c = c0 + tx;
        float a = 0.f, sh = 0.f;
        if (c < C) { a = scale[c]; sh = shift[c]; }
        const bool up = a >= 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ty + 8 * i;
            float o = 0.f;
            if (c < C) {
                const size_t e = (g0 + r) * C + c;
                const float y = up ? gmax[e] : gmin[e];
                o = lrelu(fmaf(a, y, sh), slope);
                out[e] = o;
                arg[e] = up ? gamax[e] : gamin[e];
                ymax[e] = y;
                if (out2) out2[(g0 + r) * (size_t)ld2 + c] = o;
            }
            tile[r][tx] = o;
        }
        __syncthreads();
        const size_t b = g0 / N, n0 = g0 % N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cl = ty + 8 * i;
            if (c0 + cl < C) out_t[(b * C + c0 + cl) * (size_t)N + n0 + tx] = tile[tx][cl];
        }
        __syncthreads();
    }
}

extern "C" int pcl_group_minmax_finalize2_f32(const float* gmax, const float* gmin, const int32_t* gamax, const int32_t* gamin,
                                              const float* scale, const float* shift, float slope, int G, int C, float* out,
                                              int32_t* arg, float* ymax, float* out2, int out2_ld, void* stream);
extern "C" int pcl_group_minmax_finalize_t_f32(const float* gmax, const float* gmin, const int32_t* gamax, const int32_t* gamin,
                                               const float* scale, const float* shift, float slope, int B, int N, int C, float* out,
                                               int32_t* arg, float* ymax, float* out2, int out2_ld, float* out_t, void* stream) {
    PCL_REQUIRE(gmax && gmin && gamax && gamin && scale && shift && out && arg && ymax && out_t && B >= 1 && N >= 32 && N % 32 == 0 && C >= 1 &&
                (!out2 || out2_ld >= C), "pcl_group_minmax_finalize_t_f32: bad arguments (N a multiple of 32)");
    const int tiles_c = (C + 31) / 32;
    const size_t tiles = (size_t)B * N / 32 * tiles_c;
    const int blocks = tiles < 8192 ? (int)tiles : 8192;
    hipLaunchKernelGGL(group_minmax_finalize_t_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gmax, gmin, gamax, gamin, scale, shift, slope, C, N,
                       tiles_c, tiles, out, arg, ymax, out2, out2_ld, out_t);
    return check_launch("pcl_group_minmax_finalize_t_f32");
}
extern "C" int pcl_group_minmax_finalize_f32(const float* gmax, const float* gmin, const int32_t* gamax, const int32_t* gamin,
                                             const float* scale, const float* shift, float slope, int G, int C, float* out,
                                             int32_t* arg, float* ymax, void* stream) {
    return pcl_group_minmax_finalize2_f32(gmax, gmin, gamax, gamin, scale, shift, slope, G, C, out, arg, ymax, nullptr, 0, stream);
}
extern "C" int pcl_group_minmax_finalize2_f32(const float* gmax, const float* gmin, const int32_t* gamax, const int32_t* gamin,
                                              const float* scale, const float* shift, float slope, int G, int C, float* out,
                                              int32_t* arg, float* ymax, float* out2, int out2_ld, void* stream) {
    PCL_REQUIRE(gmax && gmin && gamax && gamin && scale && shift && out && arg && ymax && G >= 1 && C >= 1 && (!out2 || out2_ld >= C),
                "pcl_group_minmax_finalize_f32: bad arguments");
    const size_t total = (size_t)G * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(group_minmax_finalize_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gmax, gmin, gamax, gamin, scale,
                       shift, slope, C, total, out, arg, ymax, out2, out2_ld);
    return check_launch("pcl_group_minmax_finalize_f32");
}

extern "C" int pcl_linear_bwd_dx_rows_f32(const float* dU, const float* Y, const float* a_, const float* k1, const float* k2, const float* mu,
                                          const int32_t* arg, const float* gz, int ns, const float* W, int P, int Cout,
                                          int Cin, const float* Yprev, const float* prev_scale, const float* prev_shift,
                                          float prev_slope, float* dUprev, double* stats_ws, const int32_t* row_meta,
                                          const int32_t* n_rows_dev, int first_col, int cin_stride, void* stream);

extern "C" int pcl_linear_bwd_dx_f32(const float* dU, const float* Y, const float* a_, const float* k1, const float* k2, const float* mu,
                                     const int32_t* arg, const float* gz, int ns, const float* W, int P, int Cout,
                                     int Cin, const float* Yprev, const float* prev_scale, const float* prev_shift,
                                     float prev_slope, float* dUprev, double* stats_ws, void* stream) {
    return pcl_linear_bwd_dx_rows_f32(dU, Y, a_, k1, k2, mu, arg, gz, ns, W, P, Cout, Cin, Yprev, prev_scale, prev_shift, prev_slope,
                                      dUprev, stats_ws, nullptr, nullptr, 0, 0, stream);
}

extern "C" int pcl_linear_bwd_dx_rows_f32(const float* dU, const float* Y, const float* a_, const float* k1, const float* k2, const float* mu,
                                          const int32_t* arg, const float* gz, int ns, const float* W, int P, int Cout,
                                          int Cin, const float* Yprev, const float* prev_scale, const float* prev_shift,
                                          float prev_slope, float* dUprev, double* stats_ws, const int32_t* row_meta,
                                          const int32_t* n_rows_dev, int first_col, int cin_stride, void* stream) {
    PCL_REQUIRE(Y && a_ && k1 && k2 && mu && W && dUprev, "pcl_linear_bwd_dx_f32: null pointer");
    PCL_REQUIRE((dU != nullptr) != (arg != nullptr && gz != nullptr), "pcl_linear_bwd_dx_f32: pass dU or (arg,gz)");
    PCL_REQUIRE(P >= 1 && Cin >= 1 && Cout >= 1 && (dU || ns >= 1), "pcl_linear_bwd_dx_f32: bad sizes");
    PCL_REQUIRE(!Yprev || (prev_scale && prev_shift && stats_ws), "pcl_linear_bwd_dx_f32: masked mode needs scale/shift/stats");
    LinArgs a = {};
    a.A = dU; a.A2 = Y; a.B = W; a.sc = a_; a.sh = k1; a.k2 = k2; a.mu = mu; a.arg = arg; a.gz = gz; a.ns = ns;
    a.C = dUprev; a.stats = stats_ws; a.Yprev = Yprev; a.esc = prev_scale; a.esh = prev_shift; a.eslope = prev_slope;
    PCL_REQUIRE(first_col >= 0 && first_col < Cin && (first_col == 0 || !Yprev), "pcl_linear_bwd_dx_rows_f32: first_col=%d only for the input gradient", first_col);
    a.M = P; a.N = Cin; a.K = Cout; a.n_begin = first_col; a.ldc = a.ldb = cin_stride > 0 ? cin_stride : Cin;
    PCL_REQUIRE(a.ldc >= Cin && (a.ldc == Cin || !Yprev), "pcl_linear_bwd_dx_rows_f32: cin_stride=%d", cin_stride);
    a.rmeta = reinterpret_cast<const int2*>(row_meta); a.m_dev = n_rows_dev;
    a.a_mode = dU ? A_DY : A_DY_SPARSE; a.e_mode = Yprev ? E_MASK_STORE_STATS : E_STORE;
    return launch_linear(a, as_stream(stream));
}

extern "C" size_t pcl_linear_bwd_dw_workspace_bytes(int P, int Cout, int Cin) {
    if (P < 1 || Cout < 1 || Cin < 1) return 0;
    int gx, ti, tj, tm, tn;
    dw_grid(P, Cout, Cin, gx, ti, tj, tm, tn);
    return sizeof(float) * (size_t)gx * Cout * Cin;
}

extern "C" int pcl_linear_bwd_dw_rows_f32(const float* dU, const float* Y, const float* a_, const float* k1, const float* k2, const float* mu,
                                          const int32_t* arg, const float* gz, int ns, const float* Xprev,
                                          const float* prev_scale, const float* prev_shift, float prev_slope, int P, int Cout,
                                          int Cin, float* dW, void* workspace, size_t workspace_bytes, const int32_t* row_meta,
                                          const int32_t* n_rows_dev, int dw_ld, void* stream);

extern "C" int pcl_linear_bwd_dw_f32(const float* dU, const float* Y, const float* a_, const float* k1, const float* k2, const float* mu,
                                     const int32_t* arg, const float* gz, int ns, const float* Xprev,
                                     const float* prev_scale, const float* prev_shift, float prev_slope, int P, int Cout,
                                     int Cin, float* dW, void* workspace, size_t workspace_bytes, void* stream) {
    return pcl_linear_bwd_dw_rows_f32(dU, Y, a_, k1, k2, mu, arg, gz, ns, Xprev, prev_scale, prev_shift, prev_slope, P, Cout, Cin, dW,
                                      workspace, workspace_bytes, nullptr, nullptr, 0, stream);
}

extern "C" int pcl_linear_bwd_dw_rows_f32(const float* dU, const float* Y, const float* a_, const float* k1, const float* k2, const float* mu,
                                          const int32_t* arg, const float* gz, int ns, const float* Xprev,
                                          const float* prev_scale, const float* prev_shift, float prev_slope, int P, int Cout,
                                          int Cin, float* dW, void* workspace, size_t workspace_bytes, const int32_t* row_meta,
                                          const int32_t* n_rows_dev, int dw_ld, void* stream) {
    PCL_REQUIRE(Y && a_ && k1 && k2 && mu && Xprev && dW, "pcl_linear_bwd_dw_f32: null pointer");
    PCL_REQUIRE(dw_ld == 0 || dw_ld >= Cin, "pcl_linear_bwd_dw_f32: dw_ld=%d < Cin=%d", dw_ld, Cin);
    PCL_REQUIRE((dU != nullptr) != (arg != nullptr && gz != nullptr), "pcl_linear_bwd_dw_f32: pass dU or (arg,gz)");
    PCL_REQUIRE((prev_scale == nullptr) == (prev_shift == nullptr), "pcl_linear_bwd_dw_f32: scale/shift together");
    PCL_REQUIRE(P >= 1 && Cin >= 1 && Cout >= 1 && (dU || ns >= 1), "pcl_linear_bwd_dw_f32: bad sizes");
    const size_t need = pcl_linear_bwd_dw_workspace_bytes(P, Cout, Cin);
    if (!workspace || workspace_bytes < need) return fail(PCL_EWS, "pcl_linear_bwd_dw_f32: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    int gx, ti, tj, tm, tn;
    dw_grid(P, Cout, Cin, gx, ti, tj, tm, tn);
    DwArgs d = {};
    d.A = dU; d.A2 = Y; d.sc = a_; d.sh = k1; d.k2 = k2; d.mu = mu; d.arg = arg; d.gz = gz; d.ns = ns;
    d.Bsrc = Xprev; d.bsc = prev_scale; d.bsh = prev_shift; d.bslope = prev_slope;
    d.part = static_cast<float*>(workspace); d.P = P; d.I = Cout; d.J = Cin;
    d.rmeta = reinterpret_cast<const int2*>(row_meta); d.p_dev = n_rows_dev;
    d.a_mode = dU ? A_DY : A_DY_SPARSE; d.b_mode = prev_scale ? A_BNACT : A_PLAIN;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = (Cout % 4 == 0) && (Cin % 4 == 0) && al16(Y) && al16(Xprev) && (!dU || al16(dU)) && (!gz || (al16(gz) && al16(arg)));
    d.gx = gx; d.ti = ti; d.tj = tj; d.ldo = Cin;
    // one row-chunk workgroup per output tile (few rows, a large weight: the per-point Linear of PointConv's GroupAll
    // level is 32 x 16384 -> 1024): nothing to reduce, the tiles go straight into dW instead of through the workspace
    const bool direct = gx == 1;
    if (direct) { d.part = dW; d.ldo = dw_ld ? dw_ld : Cin; }
    dim3 grid(gx * ti * tj);
    if (dU) { if (vec) launch_dw_t<A_DY, true>(d, grid, tm, tn, st); else launch_dw_t<A_DY, false>(d, grid, tm, tn, st); }
    else { if (vec) launch_dw_t<A_DY_SPARSE, true>(d, grid, tm, tn, st); else launch_dw_t<A_DY_SPARSE, false>(d, grid, tm, tn, st); }
    int rc = check_launch("pcl_linear_bwd_dw_f32");
    if (rc || direct) return rc;
    const size_t n = (size_t)Cout * Cin;
    const int blocks = (int)((n + 31) / 32);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(blocks), dim3(256), 0, st, d.part, gx, n, Cin, dw_ld ? dw_ld : Cin, dW);
    return check_launch("pcl_linear_bwd_dw_f32(reduce)");
}

// ---- both backward GEMMs of a few-row layer in one launch, the partial-tile sum in the finish launch (round 6) --------------------------
static int g_bwd_pair = 1;               // lab switch (pcl_set_bwd_pair): 0 = the two launches + reduce of rounds 1-5
extern "C" void pcl_set_bwd_pair(int on) { g_bwd_pair = on != 0; }
extern "C" int pcl_get_bwd_pair(void) { return g_bwd_pair; }

extern "C" int pcl_linear_bwd_pair_supported(int P, int Cout, int Cin, int first_col) {
    if (!g_bwd_pair || P < 1 || Cout <= 64 || Cin <= 64 || (g_split_mfma & 2)) return 0;
    int gx, ti, tj, tm, tn;
    dw_grid(P, Cout, Cin, gx, ti, tj, tm, tn);
    if (gx == 1 || tm != 2 || tn != 2) return 0;                 // (gx == 1: the direct form, nothing to reduce)
    int gxl, n_tiles; bool narrow, low;
    linear_grid(P, Cin - first_col, true, false, gxl, n_tiles, narrow, low);
    return low ? 1 : 0;
}

extern "C" int pcl_linear_bwd_pair_f32(const float* dU, const float* Y, const float* a_, const float* k1, const float* k2, const float* mu,
                                       const int32_t* arg, const float* gz, int ns, const float* W, const float* Xprev,
                                       const float* prev_scale, const float* prev_shift, float prev_slope, int masked, int P, int Cout, int Cin,
                                       float* dUprev, double* stats_ws, int first_col, void* workspace, size_t workspace_bytes, void* stream) {
    PCL_REQUIRE(Y && a_ && k1 && k2 && mu && W && Xprev && dUprev, "pcl_linear_bwd_pair_f32: null pointer");
    PCL_REQUIRE((dU != nullptr) != (arg != nullptr && gz != nullptr), "pcl_linear_bwd_pair_f32: pass dU or (arg,gz)");
    PCL_REQUIRE((prev_scale == nullptr) == (prev_shift == nullptr), "pcl_linear_bwd_pair_f32: scale/shift together");
    PCL_REQUIRE(!masked || (prev_scale && stats_ws), "pcl_linear_bwd_pair_f32: masked mode needs scale/shift/stats");
    PCL_REQUIRE(first_col >= 0 && first_col < Cin && (first_col == 0 || !masked), "pcl_linear_bwd_pair_f32: first_col=%d only for the input gradient", first_col);
    PCL_REQUIRE(pcl_linear_bwd_pair_supported(P, Cout, Cin, first_col), "pcl_linear_bwd_pair_f32: shape P=%d Cout=%d Cin=%d not eligible", P, Cout, Cin);
    const size_t need = pcl_linear_bwd_dw_workspace_bytes(P, Cout, Cin);
    if (!workspace || workspace_bytes < need) return fail(PCL_EWS, "pcl_linear_bwd_pair_f32: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    int gx, ti, tj, tm, tn;
    dw_grid(P, Cout, Cin, gx, ti, tj, tm, tn);
    DwArgs d = {};
    d.A = dU; d.A2 = Y; d.sc = a_; d.sh = k1; d.k2 = k2; d.mu = mu; d.arg = arg; d.gz = gz; d.ns = ns;
    d.Bsrc = Xprev; d.bsc = prev_scale; d.bsh = prev_shift; d.bslope = prev_slope;
    d.part = static_cast<float*>(workspace); d.P = P; d.I = Cout; d.J = Cin;
    d.a_mode = dU ? A_DY : A_DY_SPARSE; d.b_mode = prev_scale ? A_BNACT : A_PLAIN;
    d.gx = gx; d.ti = ti; d.tj = tj; d.ldo = Cin;
    LinArgs a = {};
    a.A = dU; a.A2 = Y; a.B = W; a.sc = a_; a.sh = k1; a.k2 = k2; a.mu = mu; a.arg = arg; a.gz = gz; a.ns = ns;
    a.C = dUprev; a.stats = stats_ws; a.Yprev = masked ? Xprev : nullptr; a.esc = prev_scale; a.esh = prev_shift; a.eslope = prev_slope;
    a.M = P; a.N = Cin; a.K = Cout; a.n_begin = first_col; a.ldc = a.ldb = Cin;
    a.a_mode = d.a_mode; a.e_mode = masked ? E_MASK_STORE_STATS : E_STORE;
    int gxl, n_tiles; bool narrow, low;
    linear_grid(P, Cin - first_col, true, false, gxl, n_tiles, narrow, low);
    a.gx = gxl; a.nt = n_tiles;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    // the two bodies' vector paths have different preconditions; the pair runs both on the same path (scalar when either needs it)
    const bool vec_dw = (Cout % 4 == 0) && (Cin % 4 == 0) && al16(Y) && al16(Xprev) && (!dU || al16(dU)) && (!gz || (al16(gz) && al16(arg)));
    const bool vec_nt = (a.K % 4 == 0) && al16(a.A) && al16(a.B) && al16(a.A2) && a.ldb % 4 == 0;
    const int n_dw = gx * ti * tj;
    if (vec_dw != vec_nt) {
        // (an operand aligned for one body's vector path only: the two plain launches, each on its own path -- the finish launch still sums the tiles)
        if (dU) { if (vec_dw) launch_dw_t<A_DY, true>(d, dim3(n_dw), tm, tn, st); else launch_dw_t<A_DY, false>(d, dim3(n_dw), tm, tn, st); }
        else { if (vec_dw) launch_dw_t<A_DY_SPARSE, true>(d, dim3(n_dw), tm, tn, st); else launch_dw_t<A_DY_SPARSE, false>(d, dim3(n_dw), tm, tn, st); }
        if (const int rc = check_launch("pcl_linear_bwd_pair_f32(dw)")) return rc;
        return launch_linear(a, st);
    }
    const bool vec = vec_dw;
    const dim3 grid(n_dw + gxl * n_tiles), blk(MLP_T);
#define PCL_PAIR(AMv, EMv, VECv) PCL_LAUNCH_TIMED((linear_bwd_pair_kernel<AMv, EMv, VECv>), grid, blk, st, d, a, n_dw)
    if (dU) {
        if (masked) { if (vec) PCL_PAIR(A_DY, E_MASK_STORE_STATS, true); else PCL_PAIR(A_DY, E_MASK_STORE_STATS, false); }
        else { if (vec) PCL_PAIR(A_DY, E_STORE, true); else PCL_PAIR(A_DY, E_STORE, false); }
    } else {
        if (masked) { if (vec) PCL_PAIR(A_DY_SPARSE, E_MASK_STORE_STATS, true); else PCL_PAIR(A_DY_SPARSE, E_MASK_STORE_STATS, false); }
        else { if (vec) PCL_PAIR(A_DY_SPARSE, E_STORE, true); else PCL_PAIR(A_DY_SPARSE, E_STORE, false); }
    }
#undef PCL_PAIR
    return check_launch("pcl_linear_bwd_pair_f32");
}

namespace pcl {
int pair_finish_impl(const void* workspace, size_t workspace_bytes, int P, int Cout, int Cin, float* dW, int dw_ld,
                     const double* stats_ws, int stat_rows, const float* gamma_prev, const float* mean_prev,
                     const float* invstd_prev, int P_bn, float* dgamma_prev, float* dbeta_prev,
                     float* a_prev, float* k1_prev, float* k2_prev, float* dbias_zero_prev, void* stream,
                     const float* xpart, int xrows, int xC1, int xld, float* xdW0);
}
extern "C" int pcl_linear_bwd_pair_finish_f32(const void* workspace, size_t workspace_bytes, int P, int Cout, int Cin, float* dW, int dw_ld,
                                              const double* stats_ws, int stat_rows, const float* gamma_prev, const float* mean_prev,
                                              const float* invstd_prev, int P_bn, float* dgamma_prev, float* dbeta_prev,
                                              float* a_prev, float* k1_prev, float* k2_prev, float* dbias_zero_prev, void* stream) {
    return pair_finish_impl(workspace, workspace_bytes, P, Cout, Cin, dW, dw_ld, stats_ws, stat_rows, gamma_prev, mean_prev, invstd_prev, P_bn, dgamma_prev,
                            dbeta_prev, a_prev, k1_prev, k2_prev, dbias_zero_prev, stream, nullptr, 0, 0, 0, nullptr);
}
// (xpart ..: the coordinate-weight partials of a folded first layer [xrows][xC1][3] summed into xdW0[c * xld + 0..2] by extra blocks of this launch)
int pcl::pair_finish_impl(const void* workspace, size_t workspace_bytes, int P, int Cout, int Cin, float* dW, int dw_ld,
                          const double* stats_ws, int stat_rows, const float* gamma_prev, const float* mean_prev,
                          const float* invstd_prev, int P_bn, float* dgamma_prev, float* dbeta_prev,
                          float* a_prev, float* k1_prev, float* k2_prev, float* dbias_zero_prev, void* stream,
                          const float* xpart, int xrows, int xC1, int xld, float* xdW0) {
    PCL_REQUIRE(workspace && dW && P >= 1 && Cout >= 1 && Cin >= 1 && (dw_ld == 0 || dw_ld >= Cin), "pcl_linear_bwd_pair_finish_f32: bad arguments");
    const size_t need = pcl_linear_bwd_dw_workspace_bytes(P, Cout, Cin);
    if (workspace_bytes < need) return fail(PCL_EWS, "pcl_linear_bwd_pair_finish_f32: workspace %zu < %zu", workspace_bytes, need);
    int gx, ti, tj, tm, tn;
    dw_grid(P, Cout, Cin, gx, ti, tj, tm, tn);
    BnConstsArgs q = {};
    int extra = 0;
    if (stats_ws) {
        PCL_REQUIRE(mean_prev && invstd_prev && a_prev && k1_prev && k2_prev && P_bn >= 1 && stat_rows >= 1, "pcl_linear_bwd_pair_finish_f32: null pointer");
        q = BnConstsArgs{stats_ws, stat_rows, gamma_prev, mean_prev, invstd_prev, P_bn, Cin, dgamma_prev, dbeta_prev, a_prev, k1_prev, k2_prev, dbias_zero_prev};
        extra = (Cin + 3) / 4;
    }
    const size_t n = (size_t)Cout * Cin;
    const int nred = (int)((n + 31) / 32);
    // (the same sum, in the same order, as reduce_rows_kernel's: the weight gradient is bit-identical to the two-launch path's)
    FinishXJob xj = {nullptr, 0, 0, 0, nullptr, 0};
    int xblocks = 0;
    if (xpart && xdW0) { xj = FinishXJob{xpart, xrows, xC1, xld, xdW0, nred + extra}; xblocks = (xC1 * 3 + 3) / 4; }
    hipLaunchKernelGGL(fused_finish_kernel, dim3(nred + extra + xblocks), dim3(256), 0, as_stream(stream), static_cast<const float*>(workspace), gx, n, Cin, dW, nred, q,
                       dw_ld, xj);
    return check_launch("pcl_linear_bwd_pair_finish_f32");
}

// ---- few-row layers: constants + dy in one launch, weight gradient on the formed dy (round 5) ----------------------------------------
static int g_dw_force_gx = 0;            // lab knob (pcl_set_dw_tuning): row-chunk workgroups per output tile of the plain-dy dW
extern "C" void pcl_set_dw_tuning(int gx) { g_dw_force_gx = gx > 0 ? gx : 0; }

extern "C" int pcl_bn_bwd_dy_supported(int P, int C) { return P >= 1 && C >= 32 && C % 32 == 0; }

extern "C" int pcl_bn_bwd_dy_f32(const double* stats_ws, int stat_rows, const float* gamma, const float* mean, const float* invstd, int P_bn, int C,
                                 float* dgamma, float* dbeta, float* a_out, float* k1, float* k2, float* dbias_zero, const float* dU, const float* Y,
                                 const int32_t* arg, const float* gz, int ns, int P, float* dy, void* stream) {
    PCL_REQUIRE(stats_ws && mean && invstd && a_out && k1 && k2 && Y && dy, "pcl_bn_bwd_dy_f32: null pointer");
    PCL_REQUIRE(stat_rows >= 1 && P_bn >= 1 && pcl_bn_bwd_dy_supported(P, C), "pcl_bn_bwd_dy_f32: bad sizes P=%d C=%d (C must be a multiple of 32) stat_rows=%d", P, C, stat_rows);
    PCL_REQUIRE((dU != nullptr) != (arg != nullptr && gz != nullptr), "pcl_bn_bwd_dy_f32: pass dU or (arg, gz)");
    PCL_REQUIRE(dU || ns >= 1, "pcl_bn_bwd_dy_f32: ns = %d", ns);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    PCL_REQUIRE(al16(Y) && al16(dy) && (!dU || al16(dU)) && (!gz || (al16(gz) && al16(arg))), "pcl_bn_bwd_dy_f32: operands must be 16-byte aligned");
    BnDyArgs a = {};
    a.q = BnConstsArgs{stats_ws, stat_rows, gamma, mean, invstd, P_bn, C, dgamma, dbeta, a_out, k1, k2, dbias_zero};
    a.dU = dU; a.Y = Y; a.arg = arg; a.gz = gz; a.ns = ns; a.dy = dy; a.M = P;
    const int cg = C / 32;
    int rs = (512 + cg - 1) / cg;                        // ~2 blocks per CU; a block's row slab >= 64 rows
    if (rs > (P + 63) / 64) rs = (P + 63) / 64;
    if (rs < 1) rs = 1;
    hipLaunchKernelGGL(bn_bwd_dy_kernel, dim3(cg, rs), dim3(256), 0, as_stream(stream), a);
    return check_launch("pcl_bn_bwd_dy_f32");
}

static void dw_plain_grid(int P, int I, int J, int& gx, int& ti, int& tj, int& tm, int& tn) {
    dw_grid(P, I, J, gx, ti, tj, tm, tn);
    if (g_dw_force_gx) {
        const int chunks = (P + DW_BP - 1) / DW_BP;
        gx = g_dw_force_gx < chunks ? g_dw_force_gx : chunks;
    }
}
extern "C" size_t pcl_linear_bwd_dw_plain_workspace_bytes(int P, int Cout, int Cin) {
    if (P < 1 || Cout < 1 || Cin < 1) return 0;
    int gx, ti, tj, tm, tn;
    dw_plain_grid(P, Cout, Cin, gx, ti, tj, tm, tn);
    return sizeof(float) * (size_t)gx * Cout * Cin;
}
/* dW[Cout][Cin] = dy^T z, z = lrelu(prev_scale x + prev_shift) or x: the staged dW kernel on an already formed dy */
extern "C" int pcl_linear_bwd_dw_plain_f32(const float* dy, const float* Xprev, const float* prev_scale, const float* prev_shift, float prev_slope, int P,
                                           int Cout, int Cin, float* dW, void* workspace, size_t workspace_bytes, int dw_ld, void* stream) {
    PCL_REQUIRE(dy && Xprev && dW, "pcl_linear_bwd_dw_plain_f32: null pointer");
    PCL_REQUIRE(dw_ld == 0 || dw_ld >= Cin, "pcl_linear_bwd_dw_plain_f32: dw_ld=%d < Cin=%d", dw_ld, Cin);
    PCL_REQUIRE((prev_scale == nullptr) == (prev_shift == nullptr), "pcl_linear_bwd_dw_plain_f32: scale/shift together");
    PCL_REQUIRE(P >= 1 && Cin >= 1 && Cout >= 1, "pcl_linear_bwd_dw_plain_f32: bad sizes");
    const size_t need = pcl_linear_bwd_dw_plain_workspace_bytes(P, Cout, Cin);
    if (!workspace || workspace_bytes < need) return fail(PCL_EWS, "pcl_linear_bwd_dw_plain_f32: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    int gx, ti, tj, tm, tn;
    dw_plain_grid(P, Cout, Cin, gx, ti, tj, tm, tn);
    DwArgs d = {};
    d.A = dy; d.Bsrc = Xprev; d.bsc = prev_scale; d.bsh = prev_shift; d.bslope = prev_slope;
    d.part = static_cast<float*>(workspace); d.P = P; d.I = Cout; d.J = Cin;
    d.a_mode = A_PLAIN; d.b_mode = prev_scale ? A_BNACT : A_PLAIN;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = (Cout % 4 == 0) && (Cin % 4 == 0) && al16(dy) && al16(Xprev);
    d.gx = gx; d.ti = ti; d.tj = tj; d.ldo = Cin;
    const bool direct = gx == 1;
    if (direct) { d.part = dW; d.ldo = dw_ld ? dw_ld : Cin; }
    dim3 grid(gx * ti * tj);
    if (vec) launch_dw_t2<A_PLAIN, true, false>(d, grid, tm, tn, st); else launch_dw_t2<A_PLAIN, false, false>(d, grid, tm, tn, st);
    int rc = check_launch("pcl_linear_bwd_dw_plain_f32");
    if (rc || direct) return rc;
    const size_t n = (size_t)Cout * Cin;
    const int blocks = (int)((n + 31) / 32);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(blocks), dim3(256), 0, st, d.part, gx, n, Cin, dw_ld ? dw_ld : Cin, dW);
    return check_launch("pcl_linear_bwd_dw_plain_f32(reduce)");
}

extern "C" void pcl_set_fb_max_blocks(int n) { g_fb_cap = n; }
// 1 (opt-in; the default is 0, the fp32 MFMA): the resident-operand GEMMs take fp32 operands as three bf16 planes on the bf16 matrix pipe (nine exact partial
// products per fp32 product); 0: the fp32 MFMA form of rounds 2-3 (kept for A/B measurements and the error comparison test)
extern "C" void pcl_set_matrix_form(int split) { g_split_mfma = split & 7; g_split_min_k = (split >> 8) > 0 ? (split >> 8) : 128; }
extern "C" int pcl_get_matrix_form(void) { return g_split_mfma; }

extern "C" void pcl_set_fb_two_images(int on) { g_fb_two = on != 0; }
extern "C" int pcl_get_fb_two_images(void) { return g_fb_two; }

extern "C" int pcl_linear_bwd_fused_supported(int Cout, int Cin) {
    return ((Cout == 64 || Cout == 128) && (Cin == 64 || Cin == 128)) || (Cout == 256 && Cin == 128);
}
extern "C" int pcl_linear_bwd_fused_stat_rows(int P, int Cin) { return P < 1 ? 1 : fb_grid(P, Cin); }
extern "C" size_t pcl_linear_bwd_fused_workspace_bytes(int P, int Cout, int Cin) {
    if (P < 1 || Cout < 1 || Cin < 1) return 0;
    return sizeof(float) * (size_t)fb_grid(P, Cin) * fb_ksplit(Cout, Cin) * Cout * Cin;
}

template <bool SPARSE, bool RAG>
static int launch_fb(const FbArgs& a_in, int Cout, int Cin, hipStream_t st) {
#if PCL_EXP == 7
    FbArgs a = a_in;
    {
        static int n_launch[2][8][8];
        a.lab_slot = n_launch[SPARSE][Cout / 64][Cin / 64]++;
    }
#else
    const FbArgs& a = a_in;
#endif
    const dim3 grid(a.gx), blk(FB_T);
#define PCL_FB(CO, CI) PCL_LAUNCH_TIMED((linear_bwd_fused_kernel<SPARSE, RAG, CO, CI>), grid, blk, st, a)
    if (Cout == 64 && Cin == 64) PCL_FB(1, 1);
    else if (Cout == 64 && Cin == 128) PCL_FB(1, 2);
    else if (Cout == 128 && Cin == 64) {
        // two tile images of 64 rows, waves in two roles (round 6); pcl_set_fb_two_images(0) keeps the one-image form for A/B runs
        if (g_fb_two) PCL_LAUNCH_TIMED((linear_bwd_fused_kernel<SPARSE, RAG, 2, 1, true>), grid, blk, st, a);
        else PCL_FB(2, 1);
    }
    else if (Cout == 128 && Cin == 128) PCL_FB(2, 2);
    else PCL_FB(4, 2);
#undef PCL_FB
    return check_launch("pcl_linear_bwd_fused_rows_f32");
}

extern "C" int pcl_linear_bwd_fused_rows_f32(const float* dU, const float* Y, const float* a_, const float* k1, const float* k2, const float* mu,
                                             const int32_t* arg, const float* gz, int ns, const float* W, int P, int Cout, int Cin,
                                             const float* Yprev, const float* prev_scale, const float* prev_shift, float prev_slope,
                                             float* dUprev, double* stats_ws, void* workspace, size_t workspace_bytes,
                                             const int32_t* row_meta, const int32_t* n_rows_dev, void* stream) {
    PCL_REQUIRE(Y && a_ && k1 && k2 && mu && W && Yprev && prev_scale && prev_shift && dUprev && stats_ws, "pcl_linear_bwd_fused_rows_f32: null pointer");
    PCL_REQUIRE((dU != nullptr) != (arg != nullptr && gz != nullptr), "pcl_linear_bwd_fused_rows_f32: pass dU or (arg,gz)");
    PCL_REQUIRE(P >= 1 && (dU || ns >= 1) && pcl_linear_bwd_fused_supported(Cout, Cin),
                "pcl_linear_bwd_fused_rows_f32: unsupported sizes P=%d Cout=%d Cin=%d ((Cout, Cin) in {64,128} x {64,128} or 256 x 128)", P, Cout, Cin);
    PCL_REQUIRE((row_meta == nullptr) == (n_rows_dev == nullptr), "pcl_linear_bwd_fused_rows_f32: row_meta and n_rows_dev come together");
    PCL_REQUIRE(prev_slope >= 0.f && prev_slope <= 1.f, "pcl_linear_bwd_fused_rows_f32: slope %f outside [0,1]", prev_slope);
    PCL_REQUIRE((size_t)P * (size_t)(Cout > Cin ? Cout : Cin) * 4 < 0xffffffffull, "pcl_linear_bwd_fused_rows_f32: tensors beyond 4 GiB need the two-kernel path");
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    PCL_REQUIRE(al16(Y) && al16(W) && al16(Yprev) && al16(a_) && al16(k1) && al16(k2) && al16(mu) && (!dU || al16(dU)) && (!gz || (al16(gz) && al16(arg))),
                "pcl_linear_bwd_fused_rows_f32: operands must be 16-byte aligned");
    const size_t need = pcl_linear_bwd_fused_workspace_bytes(P, Cout, Cin);
    if (!workspace || workspace_bytes < need) return fail(PCL_EWS, "pcl_linear_bwd_fused_rows_f32: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    FbArgs f = {};
    f.dU = dU; f.Y = Y; f.a = a_; f.k1 = k1; f.k2 = k2; f.mu = mu; f.arg = arg; f.gz = gz; f.ns = ns; f.W = W;
    f.Yprev = Yprev; f.psc = prev_scale; f.psh = prev_shift; f.pslope = prev_slope; f.dUprev = dUprev; f.stats = stats_ws;
    f.part = static_cast<float*>(workspace); f.p_dev = n_rows_dev; f.rmeta = reinterpret_cast<const int2*>(row_meta);
    f.P = P; f.gx = fb_grid(P, Cin);
    int rc;
    if (dU) rc = row_meta ? launch_fb<false, true>(f, Cout, Cin, st) : launch_fb<false, false>(f, Cout, Cin, st);
    else rc = row_meta ? launch_fb<true, true>(f, Cout, Cin, st) : launch_fb<true, false>(f, Cout, Cin, st);
    return rc;
}

#if PCL_EXP == 7
// lab build only (not declared in include/pcl_hip.h): the time stamps of the last launches, after a device synchronisation
extern "C" int pcl_lab_fbk_read(void* dst, size_t bytes) {
    if (bytes < sizeof(g_fbk)) return (int)sizeof(g_fbk);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fbk), sizeof(g_fbk), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int pcl_linear_bwd_fused_finish_f32(const void* workspace, size_t workspace_bytes, int P, int Cout, int Cin, float* dW,
                                               const double* stats_ws, const float* gamma_prev, const float* mean_prev,
                                               const float* invstd_prev, int P_bn, float* dgamma_prev, float* dbeta_prev,
                                               float* a_prev, float* k1_prev, float* k2_prev, float* dbias_zero_prev, void* stream) {
    PCL_REQUIRE(workspace && dW && P >= 1 && pcl_linear_bwd_fused_supported(Cout, Cin), "pcl_linear_bwd_fused_finish_f32: bad arguments");
    const size_t need = pcl_linear_bwd_fused_workspace_bytes(P, Cout, Cin);
    if (workspace_bytes < need) return fail(PCL_EWS, "pcl_linear_bwd_fused_finish_f32: workspace %zu < %zu", workspace_bytes, need);
    BnConstsArgs q = {};
    int extra = 0;
    if (stats_ws) {
        PCL_REQUIRE(mean_prev && invstd_prev && a_prev && k1_prev && k2_prev && P_bn >= 1, "pcl_linear_bwd_fused_finish_f32: null pointer");
        q = BnConstsArgs{stats_ws, fb_grid(P, Cin), gamma_prev, mean_prev, invstd_prev, P_bn, Cin, dgamma_prev, dbeta_prev, a_prev, k1_prev, k2_prev, dbias_zero_prev};
        extra = (Cin + 3) / 4;
    }
    const size_t n = (size_t)Cout * Cin;
    const int nred = (int)((n + 31) / 32);
    hipLaunchKernelGGL(fused_finish_kernel, dim3(nred + extra), dim3(256), 0, as_stream(stream), static_cast<const float*>(workspace),
                       fb_grid(P, Cin) * fb_ksplit(Cout, Cin), n, Cin, dW, nred, q);
    return check_launch("pcl_linear_bwd_fused_finish_f32");
}

extern "C" int pcl_bn_finalize_f32(const double* stats_ws, int stat_rows, const float* gamma, const float* beta, int P,
                                   int C, float eps, float momentum, float* scale, float* shift, float* mean_out,
                                   float* invstd_out, float* running_mean, float* running_var, void* stream) {
    PCL_REQUIRE(stats_ws && scale && shift && mean_out && invstd_out, "pcl_bn_finalize_f32: null pointer");
    PCL_REQUIRE(P >= 1 && C >= 1 && stat_rows >= 1, "pcl_bn_finalize_f32: bad sizes");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, as_stream(stream), stats_ws, stat_rows,
                       gamma, beta, P, C, eps, momentum, scale, shift, mean_out, invstd_out, running_mean, running_var);
    return check_launch("pcl_bn_finalize_f32");
}

extern "C" int pcl_bn_bwd_consts_f32(const double* stats_ws, int stat_rows, const float* gamma, const float* mean,
                                     const float* invstd, int P, int C, float* dgamma, float* dbeta, float* a_out,
                                     float* k1, float* k2, float* dbias_zero, void* stream) {
    PCL_REQUIRE(stats_ws && mean && invstd && a_out && k1 && k2, "pcl_bn_bwd_consts_f32: null pointer");
    PCL_REQUIRE(P >= 1 && C >= 1 && stat_rows >= 1, "pcl_bn_bwd_consts_f32: bad sizes");
    const BnConstsArgs q = {stats_ws, stat_rows, gamma, mean, invstd, P, C, dgamma, dbeta, a_out, k1, k2, dbias_zero};
    hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3((C + 3) / 4), dim3(256), 0, as_stream(stream), q);
    return check_launch("pcl_bn_bwd_consts_f32");
}

extern "C" int pcl_bn_act_max_f32(const float* Y, const float* scale, const float* shift, float slope, int G, int ns,
                                  int C, float* out, int32_t* arg, float* ymax, void* stream) {
    PCL_REQUIRE(Y && scale && shift && out, "pcl_bn_act_max_f32: null pointer");
    PCL_REQUIRE(G >= 1 && ns >= 1 && C >= 1, "pcl_bn_act_max_f32: bad sizes");
    const size_t total = (size_t)G * C;
    if (ns >= 64 && total <= (size_t)1 << 18 && G <= 65535) {    // few long groups: rows split over 4 or 16 slices
        const dim3 grid((C + 63) / 64, G);
        if (ns >= 256) hipLaunchKernelGGL(bn_act_max_sliced_kernel<16>, grid, dim3(1024), 0, as_stream(stream), Y, scale, shift, slope, ns, C, out, arg, ymax);
        else hipLaunchKernelGGL(bn_act_max_sliced_kernel<4>, grid, dim3(256), 0, as_stream(stream), Y, scale, shift, slope, ns, C, out, arg, ymax);
        return check_launch("pcl_bn_act_max_f32(sliced)");
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_act_max_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), Y, scale, shift, slope, G, ns, C,
                       out, arg, ymax);
    return check_launch("pcl_bn_act_max_f32");
}

extern "C" int pcl_bn_act_max_mean_f32(const float* Y, const float* scale, const float* shift, float slope, int G, int ns, int C, int ldo,
                                       float* out_max, float* out_mean, int32_t* arg, void* stream) {
    PCL_REQUIRE(Y && scale && shift && out_max && out_mean && arg, "pcl_bn_act_max_mean_f32: null pointer");
    PCL_REQUIRE(G >= 1 && G <= 65535 && ns >= 1 && C >= 1 && ldo >= C, "pcl_bn_act_max_mean_f32: bad sizes G=%d ns=%d C=%d ldo=%d", G, ns, C, ldo);
    const dim3 grid((C + 63) / 64, G);
    if (ns >= 256) hipLaunchKernelGGL(bn_act_maxmean_sliced_kernel<16>, grid, dim3(1024), 0, as_stream(stream), Y, scale, shift, slope, ns, C, ldo, out_max, out_mean, arg);
    else hipLaunchKernelGGL(bn_act_maxmean_sliced_kernel<4>, grid, dim3(256), 0, as_stream(stream), Y, scale, shift, slope, ns, C, ldo, out_max, out_mean, arg);
    return check_launch("pcl_bn_act_max_mean_f32");
}

extern "C" int pcl_bn_act_max_mean_bwd_f32(const float* gmax, const float* gmean, int ldg, const int32_t* arg, const float* Y, const float* scale,
                                           const float* shift, float slope, int G, int ns, int C, float* du, double* stats_ws,
                                           int* stat_rows_out, void* stream) {
    PCL_REQUIRE(gmax && gmean && arg && Y && scale && shift && du && stats_ws && stat_rows_out, "pcl_bn_act_max_mean_bwd_f32: null pointer");
    PCL_REQUIRE(G >= 1 && ns >= 1 && C >= 1 && ldg >= C && (size_t)G * ns < 0x7fffffffull, "pcl_bn_act_max_mean_bwd_f32: bad sizes G=%d ns=%d C=%d ldg=%d", G, ns, C, ldg);
    const int P = G * ns;
    int CW = 256;
    while (CW / 2 >= C && CW > 1) CW >>= 1;
    const int RS = 256 / CW;
    const int rows = (P + RS - 1) / RS < STAT_ROWS ? (P + RS - 1) / RS : STAT_ROWS;
    *stat_rows_out = rows;
    hipLaunchKernelGGL(bn_act_maxmean_bwd_kernel, dim3((C + CW - 1) / CW, rows), dim3(256), 0, as_stream(stream), gmax, gmean, ldg, arg, Y, scale, shift,
                       slope, G, ns, C, CW, du, stats_ws);
    return check_launch("pcl_bn_act_max_mean_bwd_f32");
}

namespace pcl {
int maxgrad_prep_impl(const float* gout, const float* out, const float* ymax, float slope, int G, int C, float* gz, double* stats_ws,
                      int* stat_rows_out, void* stream, float* zero, size_t n_zero, float* ones, int n_one, int n_one0, int ldg);
}
extern "C" int pcl_maxgrad_prep_f32(const float* gout, const float* out, const float* ymax, float slope, int G, int C,
                                    float* gz, double* stats_ws, int* stat_rows_out, void* stream) {
    return maxgrad_prep_impl(gout, out, ymax, slope, G, C, gz, stats_ws, stat_rows_out, stream, nullptr, 0, nullptr, 0, 0, C);
}
int pcl::maxgrad_prep_impl(const float* gout, const float* out, const float* ymax, float slope, int G, int C, float* gz, double* stats_ws,
                           int* stat_rows_out, void* stream, float* zero, size_t n_zero, float* ones, int n_one, int n_one0, int ldg) {
    PCL_REQUIRE(gout && out && ymax && gz && stats_ws && stat_rows_out, "pcl_maxgrad_prep_f32: null pointer");
    PCL_REQUIRE(ldg >= C, "pcl_maxgrad_prep_f32: gout row stride %d below the width %d", ldg, C);
    PCL_REQUIRE(!zero || ((reinterpret_cast<uintptr_t>(zero) & 15) == 0 && n_zero % 4 == 0), "pcl_maxgrad_prep_f32: the zero-fill region must be 16-byte aligned and a multiple of 4 floats");
    const PrepAux aux = {reinterpret_cast<float4*>(zero), n_zero / 4, ones, n_one, n_one0};
    PCL_REQUIRE(G >= 1 && C >= 1, "pcl_maxgrad_prep_f32: bad sizes");
    int CW = 256;
    while (CW / 2 >= C && CW > 1) CW >>= 1;            // lanes over channels: next power of two >= C, at most 256
    const int RS = 256 / CW;
    int rows = (G + RS - 1) / RS < 512 ? (G + RS - 1) / RS : 512;
    *stat_rows_out = rows;
    hipLaunchKernelGGL(maxgrad_prep_kernel, dim3((C + CW - 1) / CW, rows), dim3(256), 0, as_stream(stream), gout, out, ymax,
                       slope, G, C, CW, ldg, gz, stats_ws, aux);
    return check_launch("pcl_maxgrad_prep_f32");
}

extern "C" int pcl_bn_act_f32(const float* Y, const float* scale, const float* shift, float slope, int P, int C,
                              float* out, void* stream) {
    PCL_REQUIRE(Y && scale && shift && out && P >= 1 && C >= 1, "pcl_bn_act_f32: bad arguments");
    const size_t total = (size_t)P * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_act_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), Y, scale, shift, slope, C, total, out);
    return check_launch("pcl_bn_act_f32");
}

extern "C" int pcl_bn_act_bwd_f32(const float* gz, const float* Y, const float* scale, const float* shift, float slope,
                                  int P, int C, float* du, double* stats_ws, int* stat_rows_out, void* stream) {
    PCL_REQUIRE(gz && Y && scale && shift && du && stats_ws && stat_rows_out && P >= 1 && C >= 1, "pcl_bn_act_bwd_f32: bad arguments");
    int CW = 256;
    while (CW / 2 >= C && CW > 1) CW >>= 1;
    const int RS = 256 / CW;
    int rows = (P + RS - 1) / RS < STAT_ROWS ? (P + RS - 1) / RS : STAT_ROWS;
    *stat_rows_out = rows;
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3((C + CW - 1) / CW, rows), dim3(256), 0, as_stream(stream), gz, Y, scale, shift,
                       slope, P, C, CW, du, stats_ws);
    return check_launch("pcl_bn_act_bwd_f32");
}

// Column sums of a stand-alone BatchNorm over rows (see bn_rows_stats_kernel): g == NULL: (sum x, sum x^2); else (sum g, sum g*x).
extern "C" int pcl_bn_rows_stats_f32(const float* x, const float* g, int P, int C, double* stats_ws, int* stat_rows_out, void* stream) {
    PCL_REQUIRE(x && stats_ws && stat_rows_out && P >= 1 && C >= 1, "pcl_bn_rows_stats_f32: bad arguments");
    int CW = 256;
    while (CW / 2 >= C && CW > 1) CW >>= 1;
    const int RS = 256 / CW;
    int rows = (P + RS - 1) / RS < STAT_ROWS ? (P + RS - 1) / RS : STAT_ROWS;
    // enough workgroups to fill the chip, not more partial rows than that needs
    const int cb = (C + CW - 1) / CW;
    const int want = (2048 + cb - 1) / cb;
    if (rows > want) rows = want;
    *stat_rows_out = rows;
    hipLaunchKernelGGL(bn_rows_stats_kernel, dim3(cb, rows), dim3(256), 0, as_stream(stream), x, g, P, C, CW, stats_ws);
    return check_launch("pcl_bn_rows_stats_f32");
}

extern "C" int pcl_bn_rows_bwd_apply_f32(const float* g, const float* x, const float* a, const float* k1, const float* k2, const float* mean,
                                         int P, int C, float* dx, void* stream) {
    PCL_REQUIRE(g && x && a && k1 && k2 && mean && dx && P >= 1 && C >= 1, "pcl_bn_rows_bwd_apply_f32: bad arguments");
    const size_t total = (size_t)P * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), g, x, a, k1, k2, mean, C, total, dx);
    return check_launch("pcl_bn_rows_bwd_apply_f32");
}
