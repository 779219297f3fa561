// xconv.hip -- the X-transform core of PointCNN's XConv for gfx950: per region r (a representative point and its K neighbours)
//
//     FX[r]   = X[r] (K x K)  .  F[r] (K x C)                                  jt.matmul(X, fts_cat), misc/layers.py:505
//     D[r, c*dm + j] = bias[c*dm + j] + sum_k wd[c, j, k] * FX[r, k, c]         the depthwise (1,K) conv of SepConv, :151
//
// in ONE pass: F is read once, FX never exists in memory, D goes straight to the pointwise 1x1 conv (the library's MFMA
// GEMM, pcl_linear_fwd_f32).  F = [F1 | F2] along the channels (the lifted coordinates and the gathered features of
// misc/layers.py:486-489) is taken as two tensors, so the concat is not materialised either.
//
// The reference runs this as a batched [K x K] . [K x C] matmul plus a grouped conv over NCHW permutes; here a thread owns one
// channel c of one region (lanes along c: every access to F / D is coalesced), holds that channel's dm x K depthwise taps in
// registers for the whole kernel, reads X[r] as LDS broadcasts, and does K*K + dm*K FMAs per region.  Regions are packed
// 256 / C to a workgroup pass (C = 36 at the first stage of the classifier).  HBM-bound: 4*(K*C + K*K + C*dm) bytes per region.
//
// Backward (same thread map) recomputes FX, forms dFX[k] = sum_j wd[c,j,k] dD[c*dm+j], writes dF = X^T dFX, accumulates the
// tap / bias gradients in registers over all regions of the thread (per-workgroup partials, summed by the caller -- no
// atomics), and leaves (dFX, F) of the pass in LDS so that the second half of the pass can form dX[k,k'] = sum_c dFX[k,c] F[k',c]
// with one thread per (region, k, k') pair.
#include "common.h"

namespace pcl {

constexpr int XC_T = 256;

struct XcArgs {
    const float* X;            // [R, K, K]
    const float* F1; int C1;   // [R, K, C1]
    const float* F2; int C2;   // [R, K, C2] or null (C2 = 0)
    const float* wd;           // [C, DM, K]
    const float* bias;         // [C * DM]
    float* D;                  // [R, C * DM]
    int R;
    // backward
    const float* dD;           // [R, C * DM]
    float* dX; float* dF1; float* dF2;
    float* dwd_part;           // [grid, C, DM, K]
    float* dbias_part;         // [grid, C * DM]
};

__device__ __forceinline__ float xc_load_f(const XcArgs& p, size_t r, int k, int K, int c) {
    return c < p.C1 ? p.F1[(r * K + k) * p.C1 + c] : p.F2[(r * K + k) * p.C2 + (c - p.C1)];
}

// thread -> (slot, channel): slots = 256 / C regions per pass when C <= 256; wider C: one region per pass, a thread takes
// channels tid and tid + 256 (C <= 512)
template <int K, int DM, bool WIDE>
__global__ __launch_bounds__(XC_T) void xconv_core_fwd_kernel(const XcArgs p) {
    constexpr int NCH = WIDE ? 2 : 1;
    __shared__ float sX[32 * K * K];                          // X of the regions of one pass (<= 32 slots)
    const int C = p.C1 + p.C2, tid = threadIdx.x;
    const int slots = WIDE ? 1 : (XC_T / C < 32 ? XC_T / C : 32);
    const int slot = WIDE ? 0 : tid / C, c0 = WIDE ? tid : tid % C;
    const bool active = WIDE ? true : slot < slots;
    float w[NCH][DM][K], b[NCH][DM];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        const int c = c0 + 256 * h;
#pragma unroll
        for (int j = 0; j < DM; ++j) {
            b[h][j] = (active && c < C) ? p.bias[c * DM + j] : 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) w[h][j][k] = (active && c < C) ? p.wd[(c * DM + j) * K + k] : 0.f;
        }
    }
    for (int r0 = blockIdx.x * slots; r0 < p.R; r0 += gridDim.x * slots) {
        __syncthreads();                                       // (the previous pass is done with sX)
        for (int e = tid; e < slots * K * K; e += XC_T) {
            const int s = e / (K * K);
            sX[e] = r0 + s < p.R ? p.X[(size_t)(r0 + s) * K * K + (e - s * K * K)] : 0.f;
        }
        __syncthreads();
        const int r = r0 + slot;
        if (!active || r >= p.R) continue;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            const int c = c0 + 256 * h;
            if (c >= C) continue;
            float f[K], fx[K];
#pragma unroll
            for (int k = 0; k < K; ++k) f[k] = xc_load_f(p, r, k, K, c);
            const float* x = sX + slot * K * K;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float s = 0.f;
#pragma unroll
                for (int kk = 0; kk < K; ++kk) s = fmaf(x[k * K + kk], f[kk], s);
                fx[k] = s;
                __builtin_amdgcn_sched_barrier(0);             // (else all K*K LDS values of X are hoisted into registers)
            }
            float* out = p.D + (size_t)r * C * DM + c * DM;
            float o[DM];
#pragma unroll
            for (int j = 0; j < DM; ++j) {
                float s = b[h][j];
#pragma unroll
                for (int k = 0; k < K; ++k) s = fmaf(w[h][j][k], fx[k], s);
                o[j] = s;
            }
            if constexpr (DM % 4 == 0) {
#pragma unroll
                for (int j = 0; j < DM; j += 4) *reinterpret_cast<float4*>(out + j) = make_float4(o[j], o[j + 1], o[j + 2], o[j + 3]);
            } else if constexpr (DM == 2) {
                *reinterpret_cast<float2*>(out) = make_float2(o[0], o[1]);
            } else {
#pragma unroll
                for (int j = 0; j < DM; ++j) out[j] = o[j];
            }
        }
    }
}

// backward over the channel window [cb, cb + cw) (cw <= 256; wider C: one launch per window, the later ones add to dX)
template <int K, int DM>
__global__ __launch_bounds__(XC_T) void xconv_core_bwd_kernel(const XcArgs p, int cb, int cw, int accum) {
    constexpr int CS = 256;                                    // channel slots of a pass in LDS (slot * cw + c)
    static_assert(K % 4 == 0, "16-byte rows of X");
    __shared__ __attribute__((aligned(16))) float sX[32 * K * K];
    __shared__ __attribute__((aligned(16))) float sA[K][CS + 4];      // dFX[k][slot*cw + c]
    __shared__ __attribute__((aligned(16))) float sB[K][CS + 4];      // F[k][slot*cw + c]
    const int C = p.C1 + p.C2, tid = threadIdx.x;
    const int slots = XC_T / cw < 32 ? XC_T / cw : 32;
    const int slot = tid / cw, c = cb + tid % cw;
    const bool active = slot < slots;
    float w[DM][K], gw[DM][K], gb[DM];
#pragma unroll
    for (int j = 0; j < DM; ++j) {
        gb[j] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            w[j][k] = active ? p.wd[(c * DM + j) * K + k] : 0.f;
            gw[j][k] = 0.f;
        }
    }
    for (int r0 = blockIdx.x * slots; r0 < p.R; r0 += gridDim.x * slots) {
        __syncthreads();                                       // (the previous pass is done with sX / sA / sB)
        for (int e = tid; e < slots * K * K; e += XC_T) {
            const int s = e / (K * K);
            sX[e] = r0 + s < p.R ? p.X[(size_t)(r0 + s) * K * K + (e - s * K * K)] : 0.f;
        }
        __syncthreads();
        const int r = r0 + slot;
        const bool on = active && r < p.R;
        float f[K], fx[K], dfx[K], g[DM];
        if (on) {
#pragma unroll
            for (int k = 0; k < K; ++k) f[k] = xc_load_f(p, r, k, K, c);
            const float* gin = p.dD + (size_t)r * C * DM + c * DM;
            if constexpr (DM % 4 == 0) {
#pragma unroll
                for (int j = 0; j < DM; j += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(gin + j);
                    g[j] = t.x; g[j + 1] = t.y; g[j + 2] = t.z; g[j + 3] = t.w;
                }
            } else if constexpr (DM == 2) {
                const float2 t = *reinterpret_cast<const float2*>(gin);
                g[0] = t.x; g[1] = t.y;
            } else {
#pragma unroll
                for (int j = 0; j < DM; ++j) g[j] = gin[j];
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) f[k] = 0.f;
#pragma unroll
            for (int j = 0; j < DM; ++j) g[j] = 0.f;
        }
        // ONE pass over the rows of X (16-byte LDS broadcasts): row k gives FX[k] = X[k,:] . F (a dot product) and adds X[k,:] dFX[k] to
        // dF[:] = sum_k X[k, :] dFX[k] -- the same sums in the same order as two passes (rows for FX, columns for dF), a quarter of the
        // LDS instructions, and nothing for hipcc to hoist: the column pass used to pull all K*K values of X into registers (424 VGPRs at
        // K = 16: one wave per SIMD).
        const float4* x4 = reinterpret_cast<const float4*>(sX + (active ? slot : 0) * K * K);
        float dF[K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) dF[kk] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float row[K];
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                const float4 t = x4[k * (K / 4) + q];
                row[4 * q] = t.x; row[4 * q + 1] = t.y; row[4 * q + 2] = t.z; row[4 * q + 3] = t.w;
            }
            float s = 0.f, d = 0.f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) s = fmaf(row[kk], f[kk], s);
            fx[k] = s;
#pragma unroll
            for (int j = 0; j < DM; ++j) {
                d = fmaf(w[j][k], g[j], d);
                gw[j][k] = fmaf(g[j], fx[k], gw[j][k]);
            }
            dfx[k] = d;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) dF[kk] = fmaf(row[kk], d, dF[kk]);
            // one row of X in registers at a time: the row's results are pinned (empty asm with the accumulators as in/out operands)
            // before the next row's LDS reads may issue -- a scheduling barrier or a memory clobber alone lets hipcc read all K*K
            // values first and do the arithmetic afterwards (256 registers at K = 16, one wave per SIMD)
#pragma unroll
            for (int q = 0; q < K / 4; ++q)
                asm volatile("" : "+v"(dF[4 * q]), "+v"(dF[4 * q + 1]), "+v"(dF[4 * q + 2]), "+v"(dF[4 * q + 3]) :: "memory");
            asm volatile("" : "+v"(fx[k]), "+v"(dfx[k]) :: "memory");
        }
#pragma unroll
        for (int j = 0; j < DM; ++j) gb[j] += g[j];
        if (on) {
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                if (c < p.C1) p.dF1[((size_t)r * K + kk) * p.C1 + c] = dF[kk];
                else p.dF2[((size_t)r * K + kk) * p.C2 + (c - p.C1)] = dF[kk];
            }
        }
        // (column tid = slot*cw + channel; threads past the last slot write zeros nobody reads)
#pragma unroll
        for (int k = 0; k < K; ++k) { sA[k][tid] = on ? dfx[k] : 0.f; sB[k][tid] = on ? f[k] : 0.f; }
        __syncthreads();
        // dX[slot][k][k'] (+)= sum_c dFX[k][slot*cw + c] * F[k'][slot*cw + c]
        for (int e = tid; e < slots * K * K; e += XC_T) {
            const int s = e / (K * K), kk2 = e - s * K * K, k = kk2 / K, kk = kk2 - k * K;
            if (r0 + s >= p.R) continue;
            const float* a = &sA[k][s * cw];
            const float* bb = &sB[kk][s * cw];
            float acc = 0.f;
            if ((cw & 3) == 0) {                               // (rows are 16-byte aligned and so is s*cw)
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
                for (int cc = 0; cc < cw; cc += 4) {
                    const float4 u = *reinterpret_cast<const float4*>(a + cc), v = *reinterpret_cast<const float4*>(bb + cc);
                    a0 = fmaf(u.x, v.x, a0); a1 = fmaf(u.y, v.y, a1); a2 = fmaf(u.z, v.z, a2); a3 = fmaf(u.w, v.w, a3);
                }
                acc = (a0 + a1) + (a2 + a3);
            } else {
                for (int cc = 0; cc < cw; ++cc) acc = fmaf(a[cc], bb[cc], acc);
            }
            float* dst = p.dX + (size_t)(r0 + s) * K * K + kk2;
            *dst = accum ? *dst + acc : acc;
        }
    }
    // per-workgroup partials of the tap / bias gradients: slots of the same channel are summed through LDS first
    float* red = &sA[0][0];                                     // K * (CS + 4) floats >= 256 values per round
#pragma unroll
    for (int j = 0; j < DM; ++j) {
#pragma unroll
        for (int k = 0; k <= K; ++k) {                         // k == K: the bias gradient
            const float v = k < K ? gw[j][k] : gb[j];
            __syncthreads();
            red[tid] = active ? v : 0.f;
            __syncthreads();
            if (tid < cw) {
                float s = 0.f;
#pragma unroll 1                                               // (unrolled, the reads of all DM*(K+1) rounds were hoisted: 424 VGPRs for the whole kernel)
                for (int sl = 0; sl < slots; ++sl) s += red[sl * cw + tid];
                const int cc = cb + tid;
                if (k < K) p.dwd_part[((size_t)blockIdx.x * C + cc) * DM * K + j * K + k] = s;
                else p.dbias_part[(size_t)blockIdx.x * C * DM + cc * DM + j] = s;
            }
        }
    }
}

static int xc_grid(int R, int C) {
    const int slots = C > 256 ? 1 : (256 / C < 32 ? 256 / C : 32);
    int g = (R + slots - 1) / slots;
    if (g > 512) g = 512;
    return g < 1 ? 1 : g;
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_xconv_core_supported(int K, int DM, int C) {
    const bool k_ok = K == 8 || K == 12 || K == 16;
    const bool dm_ok = DM == 1 || DM == 2 || DM == 4 || DM == 16;
    return k_ok && dm_ok && C >= 1 && C <= 512 && !(DM == 16 && (K != 8 || C > 256)) && !(DM == 4 && C > 256);
}
extern "C" int pcl_xconv_core_partials(int R, int C) { return xc_grid(R, C); }

static int xc_launch_fwd(const XcArgs& a, int K, int DM, hipStream_t st) {
    const int C = a.C1 + a.C2;
    const dim3 grid(xc_grid(a.R, C)), blk(XC_T);
#define PCL_XC(KK, DD, WW) do { hipLaunchKernelGGL((xconv_core_fwd_kernel<KK, DD, WW>), grid, blk, 0, st, a); return check_launch("pcl_xconv_core_fwd_f32"); } while (0)
#define PCL_XC_K(KK)                                                                                         \
    if (K == KK) {                                                                                           \
        if (C > 256) { if (DM == 1) PCL_XC(KK, 1, true); if (DM == 2) PCL_XC(KK, 2, true); }                 \
        else { if (DM == 1) PCL_XC(KK, 1, false); if (DM == 2) PCL_XC(KK, 2, false); if (DM == 4) PCL_XC(KK, 4, false); }  \
    }
    PCL_XC_K(8) PCL_XC_K(12) PCL_XC_K(16)
    if (K == 8 && DM == 16 && C <= 256) PCL_XC(8, 16, false);
#undef PCL_XC_K
#undef PCL_XC
    return fail(PCL_EINVAL, "pcl_xconv_core_fwd_f32: unsupported K=%d depth_multiplier=%d C=%d", K, DM, C);
}

static int xc_launch_bwd(const XcArgs& a, int K, int DM, hipStream_t st) {
    const int C = a.C1 + a.C2;
    const dim3 grid(xc_grid(a.R, C)), blk(XC_T);                // (the same grid for every window: the partials are indexed by it)
    for (int cb = 0; cb < C; cb += 256) {
        const int cw = C - cb < 256 ? C - cb : 256, accum = cb > 0;
#define PCL_XC(KK, DD) if (K == KK && DM == DD) hipLaunchKernelGGL((xconv_core_bwd_kernel<KK, DD>), grid, blk, 0, st, a, cb, cw, accum); else
        PCL_XC(8, 1) PCL_XC(8, 2) PCL_XC(8, 4) PCL_XC(8, 16) PCL_XC(12, 1) PCL_XC(12, 2) PCL_XC(12, 4) PCL_XC(16, 1) PCL_XC(16, 2) PCL_XC(16, 4)
            return fail(PCL_EINVAL, "pcl_xconv_core_bwd_f32: unsupported K=%d depth_multiplier=%d", K, DM);
#undef PCL_XC
        const int rc = check_launch("pcl_xconv_core_bwd_f32");
        if (rc) return rc;
    }
    return PCL_OK;
}

extern "C" int pcl_xconv_core_fwd_f32(const float* X, const float* F1, int C1, const float* F2, int C2, const float* wd,
                                      const float* bias, int R, int K, int DM, float* D, void* stream) {
    PCL_REQUIRE(X && F1 && wd && bias && D && (F2 != nullptr) == (C2 > 0), "pcl_xconv_core_fwd_f32: null pointer");
    PCL_REQUIRE(R >= 1 && C1 >= 1 && C2 >= 0 && pcl_xconv_core_supported(K, DM, C1 + C2),
                "pcl_xconv_core_fwd_f32: unsupported sizes R=%d K=%d dm=%d C=%d+%d", R, K, DM, C1, C2);
    XcArgs a = {};
    a.X = X; a.F1 = F1; a.C1 = C1; a.F2 = F2; a.C2 = C2; a.wd = wd; a.bias = bias; a.D = D; a.R = R;
    return xc_launch_fwd(a, K, DM, as_stream(stream));
}

extern "C" int pcl_xconv_core_bwd_f32(const float* X, const float* F1, int C1, const float* F2, int C2, const float* wd,
                                      const float* dD, int R, int K, int DM, float* dX, float* dF1, float* dF2,
                                      float* dwd_part, float* dbias_part, void* stream) {
    PCL_REQUIRE(X && F1 && wd && dD && dX && dF1 && dwd_part && dbias_part && (F2 != nullptr) == (C2 > 0) && (dF2 != nullptr) == (C2 > 0),
                "pcl_xconv_core_bwd_f32: null pointer");
    PCL_REQUIRE(R >= 1 && C1 >= 1 && C2 >= 0 && pcl_xconv_core_supported(K, DM, C1 + C2),
                "pcl_xconv_core_bwd_f32: unsupported sizes R=%d K=%d dm=%d C=%d+%d", R, K, DM, C1, C2);
    XcArgs a = {};
    a.X = X; a.F1 = F1; a.C1 = C1; a.F2 = F2; a.C2 = C2; a.wd = wd; a.R = R; a.dD = dD; a.dX = dX; a.dF1 = dF1; a.dF2 = dF2;
    a.dwd_part = dwd_part; a.dbias_part = dbias_part;
    return xc_launch_bwd(a, K, DM, as_stream(stream));
}
