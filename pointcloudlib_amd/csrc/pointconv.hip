// pointconv.hip -- PointConv's density-weighted per-point contraction for gfx950.
//
// Semantics (/root/reference/misc/pointconv_utils.py:393-394 and :319-320):
//     new_points = new_points * grouped_density                       [B,S,ns,C] * [B,S,ns,1]
//     out        = matmul(new_points^T [C x ns], weights [ns x 16])   -> [B,S,C*16]
// i.e. out[g,c,m] = sum_s feat[g,s,c] * dens[g,s] * w[g,s,m] for every group g = (b, point).
// The reference runs an elementwise multiply, a transpose copy and a batched GEMM of tiny (C x ns)(ns x 16) problems:
// three round trips of the [G,ns,C] tensor.  Here: one streaming pass per direction.  The op is HBM-bound (2*C*ns*16 flops per
// group against 4*ns*C + 64*C bytes), but 16 output columns are exactly the N of v_mfma_f32_16x16x4_f32, so the forward and the
// weight-gradient kernels feed the matrix pipe (fp32 in, fp32 accumulate: the arithmetic of a vector-ALU version at twice its
// rate and a fifth of its LDS reads -- the vector versions were LDS-issue-bound, not HBM-bound); the feature-gradient kernel
// (K = 16) stays on the vector ALU.
//   backward:  d_feat[g,s,c] = dens[g,s] * sum_m dout[g,c,m] * w[g,s,m]
//              t[g,s,m]      = sum_c feat[g,s,c] * dout[g,c,m]
//              d_w[g,s,m]    = dens[g,s] * t[g,s,m],   d_dens[g,s] = sum_m w[g,s,m] * t[g,s,m]
#include "common.h"

namespace pcl {

constexpr int PC_M = 16;         // WeightNet's output width (pointconv_utils.py:236: WeightNet(3, 16))
constexpr int PC_SCH = 64;       // rows of a group staged per pass

// The feature operand may be given as the PRE-BatchNorm output of the feature MLP's last layer plus that layer's folded
// BatchNorm (fsc, fsh) and activation slope: z = lrelu(fsc*y + fsh) is then formed while loading -- the [G,ns,C] activation
// (268 MB at the first two levels) is never written or re-read (misc/pointconv_utils.py:384-389 feeding :393-394).
struct FeatBN { const float* sc; const float* sh; float slope; };
__device__ __forceinline__ float feat_act(float y, float a, float b, float slope) { const float t = fmaf(a, y, b); return fmaxf(t, t * slope); }

// d_feat[g,s,c] = dens[g,s] * sum_m dout[g,c,m] * w[g,s,m]
__global__ __launch_bounds__(256) void pointconv_contract_bwd_feat_kernel(const float* __restrict__ dout, const float* __restrict__ dens,
                                                                          const float* __restrict__ w, int ns, int C,
                                                                          float* __restrict__ dfeat) {
    __shared__ __attribute__((aligned(16))) float swd[PC_SCH * PC_M];
    const int g = blockIdx.x, tid = threadIdx.x;
    float* DF = dfeat + (size_t)g * ns * C;
    for (int c0 = blockIdx.y * blockDim.x; c0 < C; c0 += gridDim.y * blockDim.x) {
        const int c = c0 + tid;
        float d[PC_M];
        if (c < C) {
            const float4* q = reinterpret_cast<const float4*>(dout + ((size_t)g * C + c) * PC_M);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float4 v = q[j]; d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w; }
        }
        for (int s0 = 0; s0 < ns; s0 += PC_SCH) {
            const int len = min(PC_SCH, ns - s0);
            __syncthreads();
            for (int e = tid; e < len * PC_M; e += blockDim.x)
                swd[e] = w[((size_t)g * ns + s0) * PC_M + e] * dens[(size_t)g * ns + s0 + e / PC_M];
            __syncthreads();
            if (c < C) {
                for (int s = 0; s < len; ++s) {
                    float a = 0.f;
#pragma unroll
                    for (int m = 0; m < PC_M; ++m) a = fmaf(d[m], swd[s * PC_M + m], a);
                    DF[(size_t)(s0 + s) * C + c] = a;
                }
            }
        }
    }
}

// The same with the feature MLP's last BatchNorm + activation folded in (see FeatBN): what leaves is du = d_feat * lrelu'(fsc*y+fsh),
// the gradient w.r.t. that BatchNorm's output, and this workgroup's partial row of (sum du, sum du*y) -- the separate
// elementwise pass (read d_feat and y, write du: 805 MB at the first two levels) is gone.
// Round 5: on the matrix pipe.  Per group d_feat [ns x C] = wd [ns x 16] . dout^T [16 x C] with wd[s][m] = dens[s] w[s][m]: a K = 16
// product, eight v_mfma_f32_32x32x2_f32 per 32 rows x 32 channels.  A wave owns a 32-channel block (a lane one channel: its dout row is the
// B operand, 8 of the 16 m per half-wave, the contraction index permuted the same way on both operands) and walks the workgroup's groups:
// the A operand is two 16-byte loads of w per lane, the 32 x 32 results arrive in the C/D layout -- lane = channel, registers = rows --
// which is exactly the layout y is loaded and du stored in (128-byte row segments per half-wave), and the channel sums stay in the lane.
// The vector form (a lane per channel, 16 fma + four 16-byte LDS broadcasts per element, the group's wd staged behind two block barriers,
// 117 registers) ran 156 us per 268 MB level = 3.4 TB/s inside a training step; this form has no LDS, no barrier, and the next unit's y
// rows in flight behind the current unit's MFMAs.  A workgroup (4 waves = 4 channel blocks at a time) owns one partial row: no atomics.
typedef float pc_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned pc_u32x4 __attribute__((ext_vector_type(4)));
using pc_rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned PC_OOB = 0xfffffff0u;             // a per-lane offset no descriptor here covers: loads return 0, stores are dropped
// descriptor over [base + first_byte, + bytes): wave-uniform (tensors here are < 4 GiB; the range check does the row / column masking)
__device__ __forceinline__ pc_rsrc_t pc_rsrc(const void* base, size_t first_byte, unsigned bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base) + first_byte;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ float pc_ld1(pc_rsrc_t r, unsigned voff, unsigned soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ float4 pc_ld4(pc_rsrc_t r, unsigned voff, unsigned soff) {
    const pc_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void pc_st1(pc_rsrc_t r, unsigned voff, unsigned soff, float x) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0); }

// operands of one unit = (group, 32-row block) as a lane holds them: requested a unit ahead, untouched until consumed
struct PcUnit { float y[16]; float4 a0, a1, b0, b1; float dn; };

__global__ __launch_bounds__(256) void pointconv_contract_bwd_feat_bn_kernel(const float* __restrict__ dout, const float* __restrict__ dens,
                                                                             const float* __restrict__ w, const float* __restrict__ Y,
                                                                             const FeatBN bn, int G, int ns, int C, float* __restrict__ du,
                                                                             double* __restrict__ stats) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 31, lh = lane >> 5;
    const int ncb = (C + 31) >> 5;
    const int per = (ncb + gridDim.y - 1) / gridDim.y;                        // channel blocks of this grid.y slice
    const int cb_end = min(ncb, ((int)blockIdx.y + 1) * per);
    const int nmb = (ns + 31) >> 5;
    const int ng = (G - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // groups of this workgroup: blockIdx.x, + gridDim.x, ...
    const int units = ng * nmb;
    const unsigned Cb = (unsigned)C * 4u;
    const unsigned grp_bytes = (unsigned)ns * Cb;                             // one group's rows of Y / du
    for (int cb = blockIdx.y * per + wv; cb < cb_end; cb += 4) {
        const int c = cb * 32 + lr;
        const bool cok = c < C;
        const int cc = cok ? c : C - 1;
        const float ba = bn.sc[cc], bb = bn.sh[cc];
        // per-lane offsets: Y / du element (row 4 lh, channel c) of a group; the dout row of channel c, its half of the 16 m; the w row lr
        const unsigned vy = cok ? (unsigned)(4 * lh) * Cb + (unsigned)c * 4u : PC_OOB;
        const unsigned vb = cok ? (unsigned)c * (PC_M * 4u) + (unsigned)lh * 32u : PC_OOB;
        const unsigned va = (unsigned)lr * (PC_M * 4u) + (unsigned)lh * 32u;
        double s1 = 0.0, s2 = 0.0;
        float t1 = 0.f, t2 = 0.f;
        // unit u -> (group, 32-row block); a unit past the end reads through empty descriptors (zeros) so that the requests need no branch
        auto request = [&](int u, PcUnit& q) {
            const bool live = u < units;
            const int gi = u / nmb, mb = u - gi * nmb;
            const int g = blockIdx.x + gi * gridDim.x;
            const pc_rsrc_t rY = pc_rsrc(Y, (size_t)g * grp_bytes, live ? grp_bytes : 0u);
            const pc_rsrc_t rW = pc_rsrc(w, (size_t)g * ns * (PC_M * 4), live ? (unsigned)ns * (PC_M * 4u) : 0u);
            const pc_rsrc_t rD = pc_rsrc(dens, (size_t)g * ns * 4, live ? (unsigned)ns * 4u : 0u);
            const pc_rsrc_t rB = pc_rsrc(dout, (size_t)g * C * (PC_M * 4), live ? (unsigned)C * (PC_M * 4u) : 0u);
            const unsigned row0 = (unsigned)mb * 32u;
#pragma unroll
            for (int i = 0; i < 16; ++i) q.y[i] = pc_ld1(rY, vy, (row0 + (unsigned)((i & 3) + 8 * (i >> 2))) * Cb);       // rows past ns: 0
            q.a0 = pc_ld4(rW, va, row0 * (PC_M * 4u)); q.a1 = pc_ld4(rW, va + 16u, row0 * (PC_M * 4u));
            q.dn = pc_ld1(rD, (unsigned)lr * 4u, row0 * 4u);
            q.b0 = pc_ld4(rB, vb, 0u); q.b1 = pc_ld4(rB, vb + 16u, 0u);
        };
        auto consume = [&](int u, const PcUnit& q) {
            const int gi = u / nmb, mb = u - gi * nmb;
            const int g = blockIdx.x + gi * gridDim.x;
            if (mb == 0) { t1 = 0.f; t2 = 0.f; }
            const float dn = q.dn;
            pc_f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a0.x * dn, q.b0.x, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a0.y * dn, q.b0.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a0.z * dn, q.b0.z, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a0.w * dn, q.b0.w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a1.x * dn, q.b1.x, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a1.y * dn, q.b1.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a1.z * dn, q.b1.z, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a1.w * dn, q.b1.w, acc, 0, 0, 0);
            const pc_rsrc_t rU = pc_rsrc(du, (size_t)g * grp_bytes, grp_bytes);
            const unsigned row0 = (unsigned)mb * 32u;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // rows past ns: the A row read 0 -> acc = 0 -> v = 0, y = 0: nothing stored (range check), nothing summed; channels past C alike
                const float a = acc[i];
                const float v = fmaf(ba, q.y[i], bb) > 0.f ? a : a * bn.slope;
                pc_st1(rU, vy, (row0 + (unsigned)((i & 3) + 8 * (i >> 2))) * Cb, v);
                t1 += v; t2 = fmaf(v, q.y[i], t2);
            }
            if (mb == nmb - 1) { s1 += (double)t1; s2 += (double)t2; }
        };
        PcUnit q0, q1;
        request(0, q0);
        for (int u = 0; u < units; u += 2) {
            // (scheduling fences: the next unit's requests go out BEFORE this unit is consumed -- hipcc otherwise sinks them behind the MFMAs and
            //  the wait for them then also waits for this unit's 16 stores, which count in the same counter)
            request(u + 1, q1);
            __builtin_amdgcn_sched_barrier(0);
            consume(u, q0);
            __builtin_amdgcn_sched_barrier(0);
            request(u + 2, q0);
            __builtin_amdgcn_sched_barrier(0);
            if (u + 1 < units) consume(u + 1, q1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the two half-waves hold the same channels (rows 4 lh + ...): fold them, one partial row per workgroup
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (lh == 0 && cok) { stats[(size_t)blockIdx.x * 2 * C + c] = s1; stats[(size_t)blockIdx.x * 2 * C + C + c] = s2; }
    }
}

// t[s,m] = sum_c feat[g,s,c] * dout[g,c,m];  d_w = dens * t;  d_dens = sum_m w * t.
// Per group a (ns x C)(C x 16) product: 16 output columns are exactly the N of v_mfma_f32_16x16x4_f32, so this one runs on the
// matrix pipe (fp32 in, fp32 accumulate -- the arithmetic of the vector version, twice its rate, and a fifth of its LDS reads:
// the vector form read every dout value once per lane-row, 4 broadcast 16-byte reads per 16 FMAs, and was LDS-issue-bound).
// One wave per group (four groups per workgroup), 64 rows per pass = up to four 16-row M blocks.  The group's feature rows are
// staged 32 channels at a time through the wave's own LDS slab, channel-major with a row stride of 80 dwords (A operand
// feat[16mb + l%16][4kk + l/16]: the four k rows of a wave's read fall on two banks' worth of addresses = the 2-way minimum of a
// 64-lane dword read), dout[g, c, :] beside it (B operand: 64 consecutive dwords per k step); the next chunk's loads are in
// flight while the current one is consumed, and the waves (own slabs) never meet at a block barrier.  Results leave in the
// C/D layout (lane = (4-row block q, column j)): d_w rows as 64-byte pieces, d_dens by a DPP sum over the 16 lanes of a row.
// History per call (268 MB of features): 256 threads = 64 rows x 4 quarter-rows with block barriers 230 us; one wave per group
// on the vector ALU 150 us (181 us with the feature BatchNorm folded in); dout through SGPRs 245 us.
constexpr int PC_CCH = 32, PC_FLD = 80;
typedef float pc_f32x4 __attribute__((ext_vector_type(4)));
template <int CTRL>
__device__ __forceinline__ float pc_dpp_add(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
// NMB: 16-row blocks per 64-row pass; GPW: groups per wave -- groups of <= 32 rows go two to a wave (rows 0..31 | 32..63 of the
// pass, each half with its own dout slab), so that the staging (64 rows per pass whatever the group size) is not half idle.
template <int NMB, int GPW>
__global__ __launch_bounds__(256, 2) void pointconv_contract_bwd_w_kernel(const float* __restrict__ feat, const float* __restrict__ dout,
                                                                       const float* __restrict__ dens, const float* __restrict__ w,
                                                                       int G, int ns, int C, float* __restrict__ dw,
                                                                       float* __restrict__ ddens, const FeatBN bn) {
    constexpr int RG = 64 / GPW, MBG = NMB / GPW;                 // rows and 16-row blocks of a pass per group
    static_assert(MBG * GPW == NMB, "blocks per group");
    // channel-major feature image sf[c][(row + 8*((c >> 2) & 3)) & 63], row stride 80 dwords: the A-operand reads (k rows 4kk..4kk+3
    // x 16 rows) and the staging writes (lane = (row r, 4 channels 4*l7..)) both land on 32 distinct banks -- 2-way, the minimum
    // for 64 lanes (without the rotation the writes are 8-way; with a stride of 66 instead the reads are 4-way: 137 vs 95 us)
    __shared__ float sf[4][PC_CCH][PC_FLD];
    __shared__ __attribute__((aligned(16))) float sd[4][GPW][PC_CCH * PC_M];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool vec = (C & 3) == 0;
    const int q = lane >> 4, j = lane & 15;
    // a wave owns its LDS slabs and LDS executes a wave's instructions in order: compiler fences instead of block barriers
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // Persistent: a wave walks the units u = (GPW groups, one pass of RG rows) wave_id, wave_id + W, ... and inside a unit the
    // channel chunks; the NEXT (unit, chunk) is always in flight while the current one is consumed -- also across units, so a
    // wave pays the memory latency once, not once per group (one group per wave and exit: 176 us for the 268 MB level).
    const int npass = (ns + RG - 1) / RG;
    const int units = ((G + GPW - 1) / GPW) * npass;
    const int W = gridDim.x * 4;
    const int cfirst = blockIdx.y * PC_CCH, cstep = gridDim.y * PC_CCH;
    float4 pv[8], pd[2 * GPW];                            // the next chunk, in flight while this one is consumed: RAW loads --
    // masks and the folded BatchNorm are applied when the chunk is stored to LDS an iteration later (any arithmetic on the loaded
    // values here makes the wave wait for them before this iteration's MFMAs: that was the case until round 3 and cost 2x)
    // 64 rows of the pass, channels c0..c0+cl: 8 lanes x 16 B per row, 8 rows per load pass; addresses clamped, no branches
    auto load_chunk = [&](int u, int c0) {
        const int g0 = (u / npass) * GPW, s0 = (u % npass) * RG;
        const int len = min(RG, ns - s0);
        const int cl = min(PC_CCH, C - c0);
        const int cq = (lane & 7) * 4;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int r = pass * 8 + (lane >> 3);
            const int grp = r / RG, rr = r % RG;
            const int gg = min(g0 + grp, G - 1);
            const float* src = feat + ((size_t)gg * ns + s0 + min(rr, len - 1)) * C + c0;
            if (vec) pv[pass] = *reinterpret_cast<const float4*>(src + min(cq, cl - 4));
            else { pv[pass].x = src[min(cq, cl - 1)]; pv[pass].y = src[min(cq + 1, cl - 1)]; pv[pass].z = src[min(cq + 2, cl - 1)]; pv[pass].w = src[min(cq + 3, cl - 1)]; }
        }
#pragma unroll
        for (int pass = 0; pass < 2 * GPW; ++pass) {    // dout[g, c0..c0+cl, :] is cl*16 contiguous floats per group
            const int grp = pass >> 1, e = ((pass & 1) * 64 + lane) * 4;
            const float* D = dout + (size_t)min(g0 + grp, G - 1) * C * PC_M;
            pd[pass] = *reinterpret_cast<const float4*>(D + (size_t)c0 * PC_M + min(e, cl * PC_M - 4));
        }
    };
    // chunk (u, c0) from the registers into the wave's LDS slabs: folded BatchNorm + activation, zeros outside the chunk / group
    auto stash_chunk = [&](int u, int c0) {
        const int s0 = (u % npass) * RG;
        const int len = min(RG, ns - s0);
        const int cl = min(PC_CCH, C - c0);
        const int cq = (lane & 7) * 4;
        float4 ba4 = make_float4(1.f, 1.f, 1.f, 1.f), bb4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bn.sc) {                  // channels as the (clamped) loads addressed them
            if (vec) {
                const int cb0 = c0 + min(cq, cl - 4);
                ba4 = *reinterpret_cast<const float4*>(bn.sc + cb0); bb4 = *reinterpret_cast<const float4*>(bn.sh + cb0);
            } else {
                ba4 = make_float4(bn.sc[c0 + min(cq, cl - 1)], bn.sc[c0 + min(cq + 1, cl - 1)], bn.sc[c0 + min(cq + 2, cl - 1)], bn.sc[c0 + min(cq + 3, cl - 1)]);
                bb4 = make_float4(bn.sh[c0 + min(cq, cl - 1)], bn.sh[c0 + min(cq + 1, cl - 1)], bn.sh[c0 + min(cq + 2, cl - 1)], bn.sh[c0 + min(cq + 3, cl - 1)]);
            }
        }
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {            // channel-major image: sf[c][row of the pass, rotated]
            const int r0 = pass * 8 + (lane >> 3);
            float4 v = pv[pass];
            if (bn.sc) {              // the feature MLP's folded BatchNorm + activation (this lane's four channels, constants per chunk)
                v.x = feat_act(v.x, ba4.x, bb4.x, bn.slope); v.y = feat_act(v.y, ba4.y, bb4.y, bn.slope);
                v.z = feat_act(v.z, ba4.z, bb4.z, bn.slope); v.w = feat_act(v.w, ba4.w, bb4.w, bn.slope);
            }
            const bool okr = r0 % RG < len;
            v.x = okr && cq < cl ? v.x : 0.f; v.y = okr && cq + 1 < cl ? v.y : 0.f;
            v.z = okr && cq + 2 < cl ? v.z : 0.f; v.w = okr && cq + 3 < cl ? v.w : 0.f;
            const int r = (r0 + 8 * (lane & 3)) & 63;     // ((cq + i) >> 2) & 3 = lane & 3
            sf[wave][cq][r] = v.x; sf[wave][cq + 1][r] = v.y; sf[wave][cq + 2][r] = v.z; sf[wave][cq + 3][r] = v.w;
        }
#pragma unroll
        for (int pass = 0; pass < 2 * GPW; ++pass) {
            const int e = ((pass & 1) * 64 + lane) * 4;
            *reinterpret_cast<float4*>(&sd[wave][pass >> 1][e]) = e < cl * PC_M ? pd[pass] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int u = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave), c0 = cfirst;
    if (u >= units || cfirst >= C) return;
    load_chunk(u, c0);
    pc_f32x4 acc[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) acc[mb] = pc_f32x4{0.f, 0.f, 0.f, 0.f};
    while (u < units) {                                   // one (unit, chunk) per iteration; a single prefetch site
        wave_sync();
        stash_chunk(u, c0);
        const bool last = c0 + cstep >= C;                // last chunk of this unit
        const int nu = last ? u + W : u, nc = last ? cfirst : c0 + cstep;
        if (nu < units) load_chunk(nu, nc);
        wave_sync();
        // eight k steps of four channels (channels past cl are zeros on both sides); A = feat[16mb + j][4kk + q], B = dout[4kk + q][j]
        // (the operands of step k+1 are read from LDS before the MFMAs of step k are issued: left alone hipcc reads each step's
        // operands right before its MFMAs and the LDS latency is exposed eight times per chunk)
        struct WOps { float a[NMB], b[GPW]; };
        auto w_ld = [&](int kk) -> WOps {
            WOps o;
#pragma unroll
            for (int gi = 0; gi < GPW; ++gi) o.b[gi] = sd[wave][gi][(4 * kk + q) * PC_M + j];
            const float* arow = &sf[wave][4 * kk + q][0];
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) o.a[mb] = arow[((mb / MBG) * RG + (mb % MBG) * 16 + 8 * (kk & 3) + j) & 63];
            return o;
        };
        WOps cur = w_ld(0);
#pragma unroll
        for (int kk = 0; kk < PC_CCH / 4; ++kk) {
            WOps nxt = cur;
            if (kk + 1 < PC_CCH / 4) nxt = w_ld(kk + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[mb], cur.b[mb / MBG], acc[mb], 0, 0, 0);
            cur = nxt;
        }
        if (last) {
            // C/D layout: acc[mb][r] = t[group mb / MBG, row 16 (mb % MBG) + 4q + r][j]
            const int g0 = (u / npass) * GPW, s0 = (u % npass) * RG;
            const int len = min(RG, ns - s0);             // rows of a group in this pass
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grp = mb / MBG, rr = (mb % MBG) * 16 + 4 * q + r;
                    const bool ok = rr < len && g0 + grp < G;
                    const size_t row = ok ? (size_t)(g0 + grp) * ns + s0 + rr : 0;
                    const float tv = acc[mb][r];
                    float dd = ok ? w[row * PC_M + j] * tv : 0.f;
                    dd = pc_dpp_add<0xB1>(dd); dd = pc_dpp_add<0x4E>(dd); dd = pc_dpp_add<0x141>(dd); dd = pc_dpp_add<0x140>(dd);   // sum over the row's 16 lanes
                    if (ok) {
                        const float de = dens[row];
                        if (gridDim.y == 1) { dw[row * PC_M + j] = de * tv; if (j == 0) ddens[row] = dd; }
                        else { unsafeAtomicAdd(dw + row * PC_M + j, de * tv); if (j == 0) unsafeAtomicAdd(ddens + row, dd); }
                    }
                }
                acc[mb] = pc_f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        u = nu; c0 = nc;
    }
}

// ---- weight gradient with a ROW-major LDS image (round 5) ----------------------------------------------------------------------------
// t[s][m] = sum_c z[s][c] dout[c][m] as D = A B on v_mfma_f32_32x32x2_f32 with A = z (M = 32 rows of the group), B = dout (N = the 16 m, half
// of the tile idle), K = the channels.  The kernel above transposes every feature chunk on its way into LDS (32 ds_write_b32 per lane and
// chunk) because the 16x16x4 A operand wants lane = (row, channel); with the contraction index permuted -- lane (row s, h) supplies channels
// 8 kk + 4 h .. + 3 as the k slots of four consecutive MFMAs, the same permutation on the dout side -- the A operand is a 16-byte read of
// the row AS STORED: the rows go into LDS with 16-byte writes (row stride 36 dwords: conflict-free) and come back with 16-byte reads, a
// quarter of the LDS instructions, and the loads are the rows' own 128-byte segments.  A wave owns its slab (no block barrier), a unit =
// one group with its NSB blocks of 32 rows sharing the dout fragments (staged beside the rows, 2 KB per chunk), the next chunk's rows in
// registers while this one is consumed.  Results arrive as lane = row: d_w[s][4h ..] and [8 + 4h ..] are two 16-byte stores, d_dens one
// shuffle away.  C % 4 == 0; few groups (the GroupAll level) split the channel chunks over grid.y with atomics as above.
constexpr int PW_LD = 36;
template <int NSB>
__global__ __launch_bounds__(256) void pointconv_contract_bwd_w_rows_kernel(const float* __restrict__ feat, const float* __restrict__ dout,
                                                                            const float* __restrict__ dens, const float* __restrict__ w,
                                                                            int G, int ns, int C, float* __restrict__ dw,
                                                                            float* __restrict__ ddens, const FeatBN bn) {
    __shared__ __attribute__((aligned(16))) float sz[4][NSB * 32][PW_LD];     // the chunk's rows, row-major
    __shared__ __attribute__((aligned(16))) float sdo[4][PC_CCH * PC_M];      // dout[g][c0 .. c0 + 32][16]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    const int W = gridDim.x * 4;
    const int cfirst = blockIdx.y * PC_CCH, cstep = gridDim.y * PC_CCH;
    const unsigned Cb = (unsigned)C * 4u;
    const bool act = bn.sc != nullptr;
    // staging: lane = (row r8 of a pass of 8 rows, channel quad q4): 8 lanes x 16 B per row
    const int r8 = lane >> 3, q4 = (lane & 7) * 4;
    float4 pz[NSB * 4], pd[2];
    auto request = [&](int g, int c0) {                                       // raw loads; masks / BatchNorm when they are stored to LDS
        const bool live = g < G;
        const pc_rsrc_t rF = pc_rsrc(feat, (size_t)(live ? g : 0) * ns * Cb, live ? (unsigned)ns * Cb : 0u);
        const pc_rsrc_t rD = pc_rsrc(dout, (size_t)(live ? g : 0) * C * (PC_M * 4), live ? (unsigned)C * (PC_M * 4u) : 0u);
        const unsigned vz = (unsigned)r8 * Cb + (unsigned)(c0 + q4) * 4u;      // (channels past C: c0 + q4 >= C only in the last chunk -> masked below)
        const bool cok = c0 + q4 < C;
#pragma unroll
        for (int i = 0; i < NSB * 4; ++i) pz[i] = pc_ld4(rF, cok ? vz : PC_OOB, (unsigned)(8 * i) * Cb);        // rows past ns: 0
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned e = (unsigned)(i * 64 + lane) * 16u;                 // 512 floats of dout rows c0 .. c0 + 31
            pd[i] = pc_ld4(rD, e, (unsigned)c0 * (PC_M * 4u));                  // (rows past C: beyond the descriptor -> 0)
        }
    };
    auto stash = [&](int c0) {
        float4 ba = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool cok = c0 + q4 < C;
        if (act && cok) { ba = *reinterpret_cast<const float4*>(bn.sc + c0 + q4); bb = *reinterpret_cast<const float4*>(bn.sh + c0 + q4); }
#pragma unroll
        for (int i = 0; i < NSB * 4; ++i) {
            float4 v = pz[i];
            if (act) {
                v.x = feat_act(v.x, ba.x, bb.x, bn.slope); v.y = feat_act(v.y, ba.y, bb.y, bn.slope);
                v.z = feat_act(v.z, ba.z, bb.z, bn.slope); v.w = feat_act(v.w, ba.w, bb.w, bn.slope);
            }
            const bool ok = cok && (8 * i + r8) < ns;                            // (act(0) != 0: rows past the group and channels past C are zeros)
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&sz[wave][8 * i + r8][q4]) = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&sdo[wave][(i * 64 + lane) * 4]) = pd[i];
    };
    int g = blockIdx.x * 4 + wave, c0 = cfirst;
    if (cfirst >= C) return;
    request(g, c0);
    pc_f32x16 acc[NSB];
#pragma unroll
    for (int b = 0; b < NSB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    while (g < G) {
        wave_sync();
        stash(c0);
        const bool last = c0 + cstep >= C;
        const int ng = last ? g + W : g, nc = last ? cfirst : c0 + cstep;
        request(ng, nc);
        wave_sync();
#pragma unroll
        for (int kk = 0; kk < PC_CCH / 8; ++kk) {
            float bq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j] = lr < PC_M ? sdo[wave][(8 * kk + 4 * lh + j) * PC_M + lr] : 0.f;
#pragma unroll
            for (int b = 0; b < NSB; ++b) {
                const float4 a = *reinterpret_cast<const float4*>(&sz[wave][32 * b + lr][8 * kk + 4 * lh]);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[0], a.x, acc[b], 0, 0, 0); acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[1], a.y, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[2], a.z, acc[b], 0, 0, 0); acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[3], a.w, acc[b], 0, 0, 0);
            }
        }
        if (last) {
            // D = dout^T z^T: lane = row s of block b (column of D), registers = m: t[s][4h + i] = acc[b][i], t[s][8 + 4h + i] = acc[b][4 + i]
#pragma unroll
            for (int b = 0; b < NSB; ++b) {
                const int srow = 32 * b + lr;
                const bool ok = srow < ns;
                const size_t row = (size_t)g * ns + (ok ? srow : 0);
                const float4 w0 = *reinterpret_cast<const float4*>(w + row * PC_M + 4 * lh), w1 = *reinterpret_cast<const float4*>(w + row * PC_M + 8 + 4 * lh);
                const float de = dens[row];
                float dd = ((w0.x * acc[b][0] + w0.y * acc[b][1]) + (w0.z * acc[b][2] + w0.w * acc[b][3])) +
                           ((w1.x * acc[b][4] + w1.y * acc[b][5]) + (w1.z * acc[b][6] + w1.w * acc[b][7]));
                dd += __shfl_xor(dd, 32);
                if (ok) {
                    float* o = dw + row * PC_M + 4 * lh;
                    if (gridDim.y == 1) {
                        *reinterpret_cast<float4*>(o) = make_float4(de * acc[b][0], de * acc[b][1], de * acc[b][2], de * acc[b][3]);
                        *reinterpret_cast<float4*>(o + 8) = make_float4(de * acc[b][4], de * acc[b][5], de * acc[b][6], de * acc[b][7]);
                        if (lh == 0) ddens[row] = dd;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { unsafeAtomicAdd(o + i, de * acc[b][i]); unsafeAtomicAdd(o + 8 + i, de * acc[b][4 + i]); }
                        if (lh == 0) unsafeAtomicAdd(ddens + row, dd);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
            }
        }
        g = ng; c0 = nc;
    }
}

// ---- forward, fragment-direct (round 5) ---------------------------------------------------------------------------------------------
// out[g][c][m] = sum_s wd[s][m] z[s][c] as D = A B with A = wd^T (M = the 16 m, rows 16..31 of the 32 x 32 tile idle), B = z (N = 32
// channels), K = the group's rows: v_mfma_f32_32x32x2_f32, lane (c = lane & 31, h = lane >> 5) supplies z[row][c] -- 128-byte row
// segments per half-wave straight from memory, no LDS -- and holds the results for ITS channel: out[g][c][4h .. 4h+3] and [8+4h .. 8+4h+3],
// two 16-byte stores.  The contraction index of step kk is row 32 mb + 16 h + kk on both operands, so that a lane's 16 densities are four
// 16-byte loads.  A wave owns (group, 32-channel block) units, consecutive waves consecutive blocks of a group (a workgroup covers whole
// rows); the next unit's z rows are requested before the current unit's MFMAs (scheduling fences, as in the kernel above).  The LDS-staged
// 16x16x4 form above it: 116 / 99 us per level (3.8 / 3.6 TB/s).
__global__ __launch_bounds__(256) void pointconv_contract_fwd_frag_kernel(const float* __restrict__ feat, const float* __restrict__ dens,
                                                                          const float* __restrict__ w, int G, int ns, int C,
                                                                          float* __restrict__ out, const FeatBN bn) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 31, lh = lane >> 5;
    const int ncb = (C + 31) >> 5, nmb = (ns + 31) >> 5;
    const int units = G * ncb;                                                // (group, channel block); a wave walks a unit's 32-row blocks itself
    const int W = gridDim.x * 4;
    const unsigned Cb = (unsigned)C * 4u, grp_bytes = (unsigned)ns * Cb;
    const bool act = bn.sc != nullptr;
    // this lane's A rows: m = lr (lanes 16..31 of each half idle: their offset is out of range -> 0)
    const unsigned va = lr < PC_M ? (unsigned)(16 * lh) * (PC_M * 4u) + (unsigned)lr * 4u : PC_OOB;
    const int u0 = blockIdx.x * 4 + wv;
    const int steps = u0 < units ? ((units - u0 + W - 1) / W) * nmb : 0;     // step t -> unit u0 + (t / nmb) W, row block t % nmb
    float zq[2][16];
    auto request = [&](int t, float (&z)[16]) {
        const bool live = t < steps;
        const int mb = t % nmb, u = u0 + (t / nmb) * W;
        const int cb = u % ncb, g = u / ncb;
        const int c = cb * 32 + lr;
        const pc_rsrc_t rF = pc_rsrc(feat, (size_t)g * grp_bytes, live ? grp_bytes : 0u);
        const unsigned vz = c < C ? (unsigned)(16 * lh) * Cb + (unsigned)c * 4u : PC_OOB;
        const unsigned row0 = (unsigned)mb * 32u;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) z[kk] = pc_ld1(rF, vz, (row0 + (unsigned)kk) * Cb);              // rows past ns: 0
    };
    request(0, zq[0]);
    pc_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto consume = [&](int t, const float (&z)[16]) {
        const int mb = t % nmb, u = u0 + (t / nmb) * W;
        const int cb = u % ncb, g = u / ncb;
        const int c = cb * 32 + lr;
        const int cc = min(c, C - 1);
        // A operand of this (group, row block): wd[row][m] = w[row][m] dens[row], rows 32 mb + 16 h + kk (L1 / L2 hits after the group's first unit)
        const pc_rsrc_t rW = pc_rsrc(w, (size_t)g * ns * (PC_M * 4), (unsigned)ns * (PC_M * 4u));
        const pc_rsrc_t rD = pc_rsrc(dens, (size_t)g * ns * 4, (unsigned)ns * 4u);
        const unsigned row0 = (unsigned)mb * 32u;
        float a[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) a[kk] = pc_ld1(rW, va, (row0 + (unsigned)kk) * (PC_M * 4u));
        float dn[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            // (16-byte loads of 4-byte-aligned rows: ns is any number; the range check is per dword on gfx950)
            dn[4 * q4] = pc_ld1(rD, (unsigned)(16 * lh + 4 * q4) * 4u, row0 * 4u); dn[4 * q4 + 1] = pc_ld1(rD, (unsigned)(16 * lh + 4 * q4 + 1) * 4u, row0 * 4u);
            dn[4 * q4 + 2] = pc_ld1(rD, (unsigned)(16 * lh + 4 * q4 + 2) * 4u, row0 * 4u); dn[4 * q4 + 3] = pc_ld1(rD, (unsigned)(16 * lh + 4 * q4 + 3) * 4u, row0 * 4u);
        }
        float ba = 1.f, bb = 0.f;
        if (act) { ba = bn.sc[cc]; bb = bn.sh[cc]; }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float zz = act ? feat_act(z[kk], ba, bb, bn.slope) : z[kk];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk] * dn[kk], zz, acc, 0, 0, 0);
        }
        if (mb == nmb - 1) {
            if (c < C) {
                float* o = out + ((size_t)g * C + c) * PC_M + 4 * lh;
                *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4*>(o + 8) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        }
    };
    for (int t = 0; t < steps; t += 2) {
        request(t + 1, zq[1]);
        __builtin_amdgcn_sched_barrier(0);
        consume(t, zq[0]);
        __builtin_amdgcn_sched_barrier(0);
        request(t + 2, zq[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < steps) consume(t + 1, zq[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace pcl
using namespace pcl;

// groups per wave and 16-row blocks per pass of the weight-gradient kernel
static int bwd_w_gpw(int ns) { return ns <= 32 ? 2 : 1; }
static int bwd_w_max_wgs(int slices) {
    static const int cus = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        (void)hipGetLastError();
        return cu;
    }();
    const int n = 2 * cus / (slices < 1 ? 1 : slices);
    return n < 1 ? 1 : n;
}
static int g_bwd_w_rows = 1;             // lab switch (pcl_set_pointconv_paths)
extern "C" void pcl_set_pointconv_paths(int bwd_w_rows) { if (bwd_w_rows >= 0) g_bwd_w_rows = bwd_w_rows != 0; }
static void launch_bwd_w(int slices, hipStream_t st, const float* feat, const float* dout, const float* dens, const float* w, int G, int ns, int C,
                         float* dw, float* ddens, const FeatBN bn) {
    if (g_bwd_w_rows && C % 4 == 0 && ns <= 64 && (size_t)ns * C * 4 < 0x7fffffffull && ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(dout)) & 15) == 0) {
        // row-major LDS image, a wave per group (ns <= 64: the GroupAll level's 128-row groups keep the kernel above)
        int wgs = (G + 3) / 4;
        const int cap = 3 * bwd_w_max_wgs(slices) / 2;                   // persistent: three workgroups per CU (LDS: 26 / 45 KB each)
        if (wgs > cap) wgs = cap;
        const dim3 grid(wgs, slices), blk(256);
        if (ns <= 32) hipLaunchKernelGGL(pointconv_contract_bwd_w_rows_kernel<1>, grid, blk, 0, st, feat, dout, dens, w, G, ns, C, dw, ddens, bn);
        else hipLaunchKernelGGL(pointconv_contract_bwd_w_rows_kernel<2>, grid, blk, 0, st, feat, dout, dens, w, G, ns, C, dw, ddens, bn);
        return;
    }
    const int gpw = bwd_w_gpw(ns);
    const int RG = 64 / gpw, units = ((G + gpw - 1) / gpw) * ((ns + RG - 1) / RG);
    int wgs = (units + 3) / 4;
    const int cap = bwd_w_max_wgs(slices) * (gpw == 1 ? 3 : 2) / 2;      // persistent: as many workgroups as fit (LDS: 48 / 56 KB each)
    if (wgs > cap) wgs = cap;
    const dim3 grid(wgs, slices), blk(256);
    if (gpw == 2) {
        if (ns <= 16) hipLaunchKernelGGL((pointconv_contract_bwd_w_kernel<2, 2>), grid, blk, 0, st, feat, dout, dens, w, G, ns, C, dw, ddens, bn);
        else hipLaunchKernelGGL((pointconv_contract_bwd_w_kernel<4, 2>), grid, blk, 0, st, feat, dout, dens, w, G, ns, C, dw, ddens, bn);
    } else if (ns <= 48) hipLaunchKernelGGL((pointconv_contract_bwd_w_kernel<3, 1>), grid, blk, 0, st, feat, dout, dens, w, G, ns, C, dw, ddens, bn);
    else hipLaunchKernelGGL((pointconv_contract_bwd_w_kernel<4, 1>), grid, blk, 0, st, feat, dout, dens, w, G, ns, C, dw, ddens, bn);
}

static int pc_block(int C) { return C >= 256 ? 256 : (C + 63) / 64 * 64; }
// channel blocks on grid.y when there are few groups (GroupAll level: G = batch size), so that >= ~512 workgroups exist
static int pc_chan_blocks(int G, int C) {
    const int per = (C + pc_block(C) - 1) / pc_block(C);
    int want = G >= 512 ? 1 : (512 + G - 1) / G;
    return want < per ? want : per;
}

static void launch_contract_fwd(hipStream_t st, const float* feat, const float* dens, const float* w, int G, int ns, int C, float* out,
                                const FeatBN bn) {
    // fragment-direct kernel: a wave per (group, 32-channel block) unit, persistent over the units
    const long long units = (long long)G * ((C + 31) / 32);
    long long wgs = (units + 3) / 4;
    const long long cap = 2ll * bwd_w_max_wgs(1);                     // 4 workgroups per CU (128 registers: 4 waves per SIMD), all resident
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL(pointconv_contract_fwd_frag_kernel, dim3((unsigned)wgs), dim3(256), 0, st, feat, dens, w, G, ns, C, out, bn);
}

extern "C" int pcl_pointconv_contract_f32(const float* feat, const float* density, const float* weights, int G, int ns, int C,
                                          int M, float* out, void* stream) {
    PCL_REQUIRE(feat && density && weights && out, "pcl_pointconv_contract_f32: null pointer");
    PCL_REQUIRE(G >= 1 && ns >= 1 && C >= 1 && M == PC_M, "pcl_pointconv_contract_f32: bad sizes G=%d ns=%d C=%d M=%d (M must be 16)", G, ns, C, M);
    PCL_REQUIRE((long long)G * ((C + 31) / 32) < (1ll << 30) && (size_t)ns * C * 4 < 0xffffffffull, "pcl_pointconv_contract_f32: G=%d ns=%d C=%d beyond the kernel's 32-bit unit / group offsets", G, ns, C);
    launch_contract_fwd(as_stream(stream), feat, density, weights, G, ns, C, out, FeatBN{nullptr, nullptr, 1.f});
    return check_launch("pcl_pointconv_contract_f32");
}

extern "C" int pcl_pointconv_contract_bn_f32(const float* Y, const float* scale, const float* shift, float slope, const float* density,
                                             const float* weights, int G, int ns, int C, int M, float* out, void* stream) {
    PCL_REQUIRE(Y && scale && shift && density && weights && out, "pcl_pointconv_contract_bn_f32: null pointer");
    PCL_REQUIRE(G >= 1 && ns >= 1 && C >= 1 && M == PC_M && slope >= 0.f && slope <= 1.f, "pcl_pointconv_contract_bn_f32: bad sizes G=%d ns=%d C=%d M=%d slope=%g", G, ns, C, M, (double)slope);
    PCL_REQUIRE((long long)G * ((C + 31) / 32) < (1ll << 30) && (size_t)ns * C * 4 < 0xffffffffull, "pcl_pointconv_contract_bn_f32: G=%d ns=%d C=%d beyond the kernel's 32-bit unit / group offsets", G, ns, C);
    launch_contract_fwd(as_stream(stream), Y, density, weights, G, ns, C, out, FeatBN{scale, shift, slope});
    return check_launch("pcl_pointconv_contract_bn_f32");
}

extern "C" int pcl_pointconv_contract_bn_stat_rows(int G) { return G < 1 ? 0 : (G < 1024 ? G : 1024); }      // (<= G: a workgroup takes >= 1 group)

extern "C" int pcl_pointconv_contract_bn_bwd_f32(const float* dout, const float* Y, const float* scale, const float* shift, float slope,
                                                 const float* density, const float* weights, int G, int ns, int C, int M, float* du,
                                                 float* dweights, float* ddensity, double* stats_ws, void* stream) {
    PCL_REQUIRE(dout && Y && scale && shift && density && weights && du && dweights && ddensity && stats_ws, "pcl_pointconv_contract_bn_bwd_f32: null pointer");
    PCL_REQUIRE(G >= 1 && ns >= 1 && C >= 1 && M == PC_M && slope >= 0.f && slope <= 1.f, "pcl_pointconv_contract_bn_bwd_f32: bad sizes G=%d ns=%d C=%d M=%d", G, ns, C, M);
    PCL_REQUIRE((size_t)ns * C * 4 < 0x7fffffffull && (size_t)C * PC_M * 4 < 0x7fffffffull, "pcl_pointconv_contract_bn_bwd_f32: a group of ns=%d x C=%d floats is beyond the kernels' 32-bit group offsets", ns, C);
    PCL_REQUIRE(C % 4 != 0 || ((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0, "pcl_pointconv_contract_bn_bwd_f32: scale / shift must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const FeatBN bn = {scale, shift, slope};
    const int rows = pcl_pointconv_contract_bn_stat_rows(G);
    // few groups (the GroupAll level): channel blocks spread over grid.y, >= 4 per slice (one per wave)
    const int ncb = (C + 31) / 32;
    int cslices = rows >= 512 ? 1 : (512 + rows - 1) / rows;
    if (cslices > (ncb + 3) / 4) cslices = (ncb + 3) / 4;
    hipLaunchKernelGGL(pointconv_contract_bwd_feat_bn_kernel, dim3(rows, cslices), dim3(256), 0, st, dout, density, weights, Y, bn, G, ns, C, du, stats_ws);
    int rc = check_launch("pcl_pointconv_contract_bn_bwd_f32(feat)");
    if (rc) return rc;
    const int wgs = (G + 4 * bwd_w_gpw(ns) - 1) / (4 * bwd_w_gpw(ns));
    int slices = wgs >= 512 ? 1 : (512 + wgs - 1) / wgs;
    if (slices > (C + PC_CCH - 1) / PC_CCH) slices = (C + PC_CCH - 1) / PC_CCH;
    if (slices > 1) {
        hipError_t e = hipMemsetAsync(dweights, 0, sizeof(float) * (size_t)G * ns * PC_M, st);
        if (e == hipSuccess) e = hipMemsetAsync(ddensity, 0, sizeof(float) * (size_t)G * ns, st);
        if (e != hipSuccess) return fail(PCL_EHIP, "pcl_pointconv_contract_bn_bwd_f32: memset: %s", hipGetErrorString(e));
    }
    launch_bwd_w(slices, st, Y, dout, density, weights, G, ns, C, dweights, ddensity, bn);
    return check_launch("pcl_pointconv_contract_bn_bwd_f32(w)");
}

extern "C" int pcl_pointconv_contract_bwd_f32(const float* dout, const float* feat, const float* density, const float* weights,
                                              int G, int ns, int C, int M, float* dfeat, float* dweights, float* ddensity,
                                              void* stream) {
    PCL_REQUIRE(dout && feat && density && weights && dfeat && dweights && ddensity, "pcl_pointconv_contract_bwd_f32: null pointer");
    PCL_REQUIRE(G >= 1 && ns >= 1 && C >= 1 && M == PC_M, "pcl_pointconv_contract_bwd_f32: bad sizes G=%d ns=%d C=%d M=%d", G, ns, C, M);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(pointconv_contract_bwd_feat_kernel, dim3(G, pc_chan_blocks(G, C)), dim3(pc_block(C)), 0, st, dout, density, weights, ns, C, dfeat);
    int rc = check_launch("pcl_pointconv_contract_bwd_f32(feat)");
    if (rc) return rc;
    // few groups (the GroupAll level: G = batch size): slice the channels over grid.y so that the chip is busy; the slices'
    // partial sums meet through atomics in zero-filled outputs
    const int wgs = (G + 4 * bwd_w_gpw(ns) - 1) / (4 * bwd_w_gpw(ns));
    int slices = wgs >= 512 ? 1 : (512 + wgs - 1) / wgs;
    if (slices > (C + PC_CCH - 1) / PC_CCH) slices = (C + PC_CCH - 1) / PC_CCH;
    if (slices > 1) {
        hipError_t e = hipMemsetAsync(dweights, 0, sizeof(float) * (size_t)G * ns * PC_M, st);
        if (e == hipSuccess) e = hipMemsetAsync(ddensity, 0, sizeof(float) * (size_t)G * ns, st);
        if (e != hipSuccess) return fail(PCL_EHIP, "pcl_pointconv_contract_bwd_f32: memset: %s", hipGetErrorString(e));
    }
    launch_bwd_w(slices, st, feat, dout, density, weights, G, ns, C, dweights, ddensity, FeatBN{nullptr, nullptr, 1.f});
    return check_launch("pcl_pointconv_contract_bwd_f32(w)");
}
