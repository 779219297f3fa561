// head.hip -- the classification head on a handful of rows, gfx950.
//
// After the last set-abstraction level a cloud is ONE row: the FC head (Linear -> BatchNorm1d -> ReLU -> ... -> Linear,
// /root/reference/networks/cls/pointnet2.py:138-147, dgcnn.py:87-93,117-121, pointnet.py:22-38) runs on R = batch-size rows
// (32).  As library GEMM + BatchNorm + activation launches that is ~45 tiny kernels per step, each a few microseconds of
// latency for microseconds of work.  With R <= 64 a whole column of the output fits one wave, so a layer is one kernel:
//   forward : wave = one output column n; lanes split K (coalesced reads of W[n,:], X staged through LDS); after the
//             cross-lane reduction lane r holds y[r,n]; batch mean / variance over the R rows, normalise, activate, store.
//   backward: (1) wave = column n again: du = dout * act'(out), BatchNorm backward over the column, dy[:,n] stored;
//                 dW[n,:] = sum_r dy[r] X[r,:] written coalesced; dgamma, dbeta, dbias.
//             (2) dX[r,k] = sum_n dy[r,n] W[n,k]: lane = k, the n range split over waves, atomics into dX.
// BatchNorm1d follows torch (the head is torch's BatchNorm1d in the counterpart networks): biased variance for the
// normalisation, UNBIASED variance into running_var, momentum 0.1.
#include "common.h"

namespace pcl {

constexpr int HD_KC = 256;       // k per wave pass: 64 lanes x 4

// Dropout of the head (nn.Dropout between the FC layers, networks/cls/pointnet2.py:146, dgcnn.py:118-120): element idx of a
// layer's output is kept with probability 1 - p and scaled by 1/(1-p).  The keep decision is a counter-based hash of
// (seed, idx) -- recomputed in backward, no mask is stored.  p = 0: identity.
struct Drop { float p; unsigned lo, hi; };
__device__ __forceinline__ float drop_scale(const Drop d, unsigned idx) {
    if (d.p <= 0.f) return 1.f;
    unsigned h = (idx + d.hi) * 0x9E3779B1u ^ d.lo;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += d.hi; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    return u >= d.p ? 1.0f / (1.0f - d.p) : 0.f;
}

// One workgroup per output column n, its 4 waves split K in 256-wide chunks; X rows are read straight from global
// memory (L2-resident, 16-byte pieces), so a lane has RMAX + 1 independent loads in flight per chunk and no barrier
// until the four partial sums meet in LDS.
template <int RMAX>
__device__ __forceinline__ void head_col_dot(const float* __restrict__ X, const float* __restrict__ Wn, int R, int K, int wave,
                                             int lane, float (&acc)[RMAX]) {
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = 0.f;
    const bool vec = (K & 3) == 0;
    for (int k0 = wave * HD_KC; k0 < K; k0 += 4 * HD_KC) {
        const int k = k0 + 4 * lane;
        float w[4];
        if (vec && k + 3 < K) {
            const float4 t = *reinterpret_cast<const float4*>(Wn + k);
            w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = k + j < K ? Wn[k + j] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            if (r < R) {
                if (vec && k + 3 < K) {
                    const float4 t = *reinterpret_cast<const float4*>(X + (size_t)r * K + k);
                    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = k + j < K ? X[(size_t)r * K + k + j] : 0.f;
                }
            }
            acc[r] = fmaf(w[3], x[3], fmaf(w[2], x[2], fmaf(w[1], x[1], fmaf(w[0], x[0], acc[r]))));
        }
    }
}

// Column epilogue of a forward layer, executed by one full wave: lane r holds y[r,n] (pre-bias).  bn_mode: 0 none, 1 batch
// statistics (training), 2 running statistics (eval); +4: running_var receives the BIASED batch variance (the
// BatchNorm of the set-abstraction stacks, pcl_bn_finalize_f32) instead of torch's unbiased one.
__device__ __forceinline__ void head_epilogue(float y, int lane, int n, const float* __restrict__ bias, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                              int R, int N, int bn_mode_full, float eps, float momentum, float slope,
                                              float* __restrict__ Ypre, float* __restrict__ OUT, float* __restrict__ mean_out,
                                              float* __restrict__ invstd_out, const Drop drop = Drop{0.f, 0u, 0u}) {
    const int bn_mode = bn_mode_full & 3;
    const bool biased = (bn_mode_full & 4) != 0;
    const bool in = lane < R;
    y = in ? y + (bias ? bias[n] : 0.f) : 0.f;
    float out = y;
    if (bn_mode != 0) {
        float mean, invstd;
        if (bn_mode == 1) {
            float s = y, q = 0.f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            mean = s / (float)R;
            const float d = in ? y - mean : 0.f;
            q = d * d;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
            const float var = q / (float)R;
            invstd = 1.0f / sqrtf(var + eps);
            if (lane == 0) {
                if (rmean) rmean[n] += (mean - rmean[n]) * momentum;
                if (rvar) rvar[n] += (((R > 1 && !biased) ? q / (float)(R - 1) : var) - rvar[n]) * momentum;
            }
        } else {
            mean = rmean[n];
            invstd = 1.0f / sqrtf(rvar[n] + eps);
        }
        if (lane == 0) { mean_out[n] = mean; invstd_out[n] = invstd; }
        out = fmaf((y - mean) * invstd, gamma ? gamma[n] : 1.f, beta ? beta[n] : 0.f);
    }
    out = out > 0.f ? out : out * slope;
    out *= drop_scale(drop, (unsigned)(lane * N + n));
    if (in) { Ypre[(size_t)lane * N + n] = y; OUT[(size_t)lane * N + n] = out; }
}

// Y_pre [R,N] (pre-BatchNorm, kept for backward), OUT [R,N] = act(BN(Y_pre)); mean/invstd [N] saved.
template <int RMAX>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, int R, int K, int N, int bn_mode /*0 none,1 train,2 eval*/,
                                                       float eps, float momentum, float slope, float* __restrict__ Ypre,
                                                       float* __restrict__ OUT, float* __restrict__ mean_out,
                                                       float* __restrict__ invstd_out, const Drop drop) {
    __shared__ float part[4][RMAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x;
    float acc[RMAX];
    head_col_dot<RMAX>(X, W + (size_t)n * K, R, K, wave, lane, acc);
    // lane r of each wave gets that wave's partial of row r; the four partials meet in LDS
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        float v = acc[r];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == r) part[wave][r] = v;
    }
    __syncthreads();
    if (wave != 0) return;
    float y = 0.f;
    if (lane < RMAX) y = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    head_epilogue(y, lane, n, bias, gamma, beta, rmean, rvar, R, N, bn_mode, eps, momentum, slope, Ypre, OUT, mean_out, invstd_out, drop);
}

// backward, column part: dy[:,n], dW[n,:], dbias[n], dgamma[n], dbeta[n].  One workgroup per column; every wave forms
// the column's dy (R values, lane r) itself and writes its quarter of dW[n,:].
template <int RMAX>
__global__ __launch_bounds__(256) void head_bwd_col_kernel(const float* __restrict__ X, const float* __restrict__ dOUT,
                                                           const float* __restrict__ OUT, const float* __restrict__ Ypre,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, int R, int K, int N, int bn_mode,
                                                           float slope, float* __restrict__ dY, float* __restrict__ dW,
                                                           float* __restrict__ dbias, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dX_zero, const Drop drop) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x;
    if (dX_zero) {                         // the dX kernel that follows accumulates with atomics: clear its target here
        const int total = R * K, chunk = (total + N - 1) / N;
        for (int e = n * chunk + threadIdx.x; e < min((n + 1) * chunk, total); e += 256) dX_zero[e] = 0.f;
    }
    const bool in = lane < R;
    const size_t o = (size_t)min(lane, R - 1) * N + n;
    float du = in ? dOUT[o] * drop_scale(drop, (unsigned)o) : 0.f;     // (a dropped element: OUT = 0 and du = 0)
    const float outv = OUT[o];
    du = outv > 0.f ? du : du * slope;                      // act'(.) from the sign of the activation's output (slope >= 0)
    float dy = du;
    if (bn_mode != 0) {
        const float g = gamma ? gamma[n] : 1.f, mu = mean[n], is = invstd[n];
        const float xh = in ? (Ypre[o] - mu) * is : 0.f;
        float s1 = du, s2 = du * xh;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        if (lane == 0 && wave == 0) { if (dgamma) dgamma[n] = s2; if (dbeta) dbeta[n] = s1; }
        dy = bn_mode == 1 ? g * is * (du - s1 / (float)R - xh * s2 / (float)R) : g * is * du;
        if (!in) dy = 0.f;
    }
    if (in && wave == 0) dY[o] = dy;
    if (dbias) {
        float s = dy;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0 && wave == 0) dbias[n] = s;
    }
    // dW[n, k] = sum_r dy[r] * X[r, k]: lane r holds dy[r]; broadcast it row by row
    const bool vec = (K & 3) == 0;
    for (int k0 = wave * HD_KC; k0 < K; k0 += 4 * HD_KC) {
        const int k = k0 + 4 * lane;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const float d = __shfl(dy, r);
            if (r < R) {
                if (vec && k + 3 < K) {
                    const float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * K + k);
                    a[0] = fmaf(d, x.x, a[0]); a[1] = fmaf(d, x.y, a[1]); a[2] = fmaf(d, x.z, a[2]); a[3] = fmaf(d, x.w, a[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < K) a[j] = fmaf(d, X[(size_t)r * K + k + j], a[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k + j < K) dW[(size_t)n * K + k + j] = a[j];
    }
}

// Column part of a layer's backward, one full wave per column (lane r = batch row r): du = dout * act', BatchNorm backward over the
// column, dy stored; dbias / dgamma / dbeta.  Returns this lane's dy.
__device__ __forceinline__ float head_col_dy(int lane, int n, const float* __restrict__ dOUT, const float* __restrict__ OUT,
                                             const float* __restrict__ Ypre, const float* __restrict__ gamma,
                                             const float* __restrict__ mean, const float* __restrict__ invstd, int R, int N,
                                             int bn_mode, float slope, float* __restrict__ dY, float* __restrict__ dbias,
                                             float* __restrict__ dgamma, float* __restrict__ dbeta, const Drop drop) {
    const bool in = lane < R;
    const size_t o = (size_t)min(lane, R - 1) * N + n;
    float du = in ? dOUT[o] * drop_scale(drop, (unsigned)o) : 0.f;
    du = OUT[o] > 0.f ? du : du * slope;
    float dy = du;
    if (bn_mode != 0) {
        const float g = gamma ? gamma[n] : 1.f, mu = mean[n], is = invstd[n];
        const float xh = in ? (Ypre[o] - mu) * is : 0.f;
        float s1 = du, s2 = du * xh;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        if (lane == 0) { if (dgamma) dgamma[n] = s2; if (dbeta) dbeta[n] = s1; }
        dy = bn_mode == 1 ? g * is * (du - s1 / (float)R - xh * s2 / (float)R) : g * is * du;
        if (!in) dy = 0.f;
    }
    if (in) dY[o] = dy;
    if (dbias) {
        float sb = dy;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sb += __shfl_xor(sb, off);
        if (lane == 0) dbias[n] = sb;
    }
    return dy;
}

// dX[r,k] += sum_{n in this wave's range} dy[r,n] * W[n,k]; grid (ceil(K/64), N splits of HD_NS)
constexpr int HD_NS = 8;
template <int RMAX>
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ dY, const float* __restrict__ W, int R, int K,
                                                          int N, float* __restrict__ dX) {
    __shared__ float sD[4][RMAX][HD_NS + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 64 + lane;
    const int n0 = (blockIdx.y * 4 + wave) * HD_NS;
    for (int e = lane; e < RMAX * HD_NS; e += 64) {
        const int r = e / HD_NS, j = e - r * HD_NS;
        sD[wave][r][j] = (r < R && n0 + j < N) ? dY[(size_t)r * N + n0 + j] : 0.f;
    }
    __syncthreads();
    float acc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = 0.f;
    const int kc = min(k, K - 1);
    for (int j = 0; j < HD_NS; ++j) {
        if (n0 + j >= N) break;
        const float w = W[(size_t)(n0 + j) * K + kc];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[r] = fmaf(sD[wave][r][j], w, acc[r]);
    }
    if (k < K && n0 < N) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
            if (r < R) unsafeAtomicAdd(&dX[(size_t)r * K + k], acc[r]);
    }
}

typedef float hd_f32x16 __attribute__((ext_vector_type(16)));
// ---- wide layers, forward on the matrix pipe with split K (round 4) ------------------------------------------------------------------
// part[ks][32][N] = X[32, k range ks] . W[N, k range ks]^T: grid (N / 32, KS) workgroups, X and W chunks of 128 k staged through LDS
// (coalesced 512-byte row pieces), the four waves split every chunk's k, their tiles meet in LDS in a fixed order.  A second kernel (one
// wave per column) adds the KS partial sums in order and runs the layer's epilogue: deterministic, 61 + 6 us at 16384 x 1024 where the
// 8-columns-per-workgroup VALU kernel took 111.  The partial sums live in a workspace of the caller (pcl_head_layer_fwd_workspace_bytes).
constexpr int HM_KC = 128;
typedef unsigned hd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hd_rsrc(const void* base, size_t first_byte, size_t bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base) + first_byte;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
    const unsigned n = bytes > 0x7fffffffull ? 0x7fffffffu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
__device__ __forceinline__ float4 hd_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const hd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__global__ __launch_bounds__(256) void head_fwd_part_kernel(const float* __restrict__ X, const float* __restrict__ W, int R, int K, int N, int kper,
                                                            float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float sX[32][HM_KC + 4];
    __shared__ __attribute__((aligned(16))) float sW[32][HM_KC + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 32, kbeg = blockIdx.y * kper, kend = min(K, kbeg + kper);
    const int row = tid >> 3, kq = (tid & 7) * 4;
    hd_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // operands through buffer descriptors over this workgroup's k range of the 32 rows / columns: rows past R, columns past N and k past
    // kend read 0 by the range check -- no branch around a load (the branchy form kept px / pw in 160 B of scratch and waited for every
    // load right after issuing it: 85 us inside a training step for a kernel that moves 67 MB)
    float4 px[4], pw[4];
    const unsigned rowb = (unsigned)K * 4u;
    const __amdgpu_buffer_rsrc_t rX = hd_rsrc(X, 0, (size_t)R * rowb), rW = hd_rsrc(W, (size_t)n0 * rowb, (size_t)min(32, N - n0) * rowb);
    const unsigned vo = (unsigned)row * rowb + (unsigned)kq * 4u;
    auto load = [&](int k0) {                                  // (K % 4 == 0 and 16-byte rows: checked by the caller)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 32 * i;                         // wave-uniform part of the k index; k + kq < kend <=> k < kend (kend % 4 == 0 ... HM_KC)
            const unsigned dead = (k + kq < kend) ? 0u : 0x80000000u;
            px[i] = hd_ld4(rX, vo | dead, (unsigned)k * 4u);
            pw[i] = hd_ld4(rW, vo | dead, (unsigned)k * 4u);
        }
    };
    load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += HM_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&sX[row][kq + 32 * i]) = px[i];
            *reinterpret_cast<float4*>(&sW[row][kq + 32 * i]) = pw[i];
        }
        __syncthreads();
        if (k0 + HM_KC < kend) load(k0 + HM_KC);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {                        // this wave's 32 k of the chunk: lanes lh = 0 / 1 take k 0-3 / 4-7 of every 8
            const float4 a = *reinterpret_cast<const float4*>(&sX[lr][wave * 32 + kk * 8 + lh * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&sW[lr][wave * 32 + kk * 8 + lh * 4]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    __syncthreads();
    // the four waves' tiles meet in LDS: waves 0, 1 park theirs in sX, waves 2, 3 in sW (each array holds two [32][33] tiles)
    float* red = &sX[0][0];
    float* red2 = &sW[0][0];
    float* mine = (wave < 2 ? red : red2) + (wave & 1) * 32 * 33;
#pragma unroll
    for (int i = 0; i < 16; ++i) mine[((i & 3) + 8 * (i >> 2) + 4 * lh) * 33 + lr] = acc[i];
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const int r = e >> 5, c = e & 31, o = r * 33 + c;
        const float s = (red[o] + red[32 * 33 + o]) + (red2[o] + red2[32 * 33 + o]);
        if (n0 + c < N) part[((size_t)blockIdx.y * 32 + r) * N + n0 + c] = s;
    }
}
__global__ __launch_bounds__(256) void head_fwd_finish_kernel(const float* __restrict__ part, int KS, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ rmean, float* __restrict__ rvar, int R, int N, int bn_mode,
                                                              float eps, float momentum, float slope, float* __restrict__ Ypre,
                                                              float* __restrict__ OUT, float* __restrict__ mean_out,
                                                              float* __restrict__ invstd_out, const Drop drop) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float y = 0.f;
    if (lane < R) for (int ks = 0; ks < KS; ++ks) y += part[((size_t)ks * 32 + lane) * N + n];
    head_epilogue(y, lane, n, bias, gamma, beta, rmean, rvar, R, N, bn_mode, eps, momentum, slope, Ypre, OUT, mean_out, invstd_out, drop);
}

// ---- wide layers, backward on the matrix pipe (round 4) -----------------------------------------------------------------------------
// The VALU forms above give every workgroup 8 columns and all of K: 128 workgroups of one wave per SIMD, 98 / 79 us for the 67 MB of
// dW / W that a 16384 -> 1024 layer moves.  With R <= 32 a layer's backward is two GEMMs with a 32-wide dimension, which is one MFMA tile:
//   dW[n, k] = sum_r dy[r, n] X[r, k]   -- a 32 x 32 tile of dW is 16 v_mfma_f32_32x32x2_f32 (K = the 32 rows); A = dy (LDS, once per
//                                          workgroup), B = X rows read coalesced (L2-resident), the tile stored as full 128-byte lines;
//   dX[r, k] = sum_n dy[r, n] W[n, k]   -- a wave owns a 32-wide k block and streams W[n, k block] once (128-byte lines), dy from LDS.
// 30 / 37 us at 16384 x 1024 (tools/ubench/head_wide_mfma.hip, checked there against fp64).  The column part (du, BatchNorm backward,
// dbias / dgamma / dbeta, clearing dX) is its own small kernel, one wave per column.
__global__ __launch_bounds__(256) void head_dy_kernel(const float* __restrict__ dOUT, const float* __restrict__ OUT, const float* __restrict__ Ypre,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, int R, int K, int N, int bn_mode, float slope,
                                                      float* __restrict__ dY, float* __restrict__ dbias, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, float* __restrict__ dX_zero, const Drop drop) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (dX_zero) {
        const int total = R * K, nblk = (int)gridDim.x, chunk = (total + nblk - 1) / nblk;
        for (int e = blockIdx.x * chunk + tid; e < min(((int)blockIdx.x + 1) * chunk, total); e += 256) dX_zero[e] = 0.f;
    }
    const int n = blockIdx.x * 4 + wave;
    if (n < N) (void)head_col_dy(lane, n, dOUT, OUT, Ypre, gamma, mean, invstd, R, N, bn_mode & 3, slope, dY, dbias, dgamma, dbeta, drop);
}
__global__ __launch_bounds__(256) void head_dw_mfma_kernel(const float* __restrict__ dY, const float* __restrict__ X, int R, int K, int N,
                                                           int kb_per_wg, float* __restrict__ dW) {
    __shared__ float sD[32][33];                             // dy[r][n0 + n]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    for (int e = tid; e < 1024; e += 256) {
        const int r = e >> 5, c = e & 31;
        sD[r][c] = (r < R && n0 + c < N) ? dY[(size_t)r * N + n0 + c] : 0.f;
    }
    __syncthreads();
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = sD[2 * j + lh][lr];   // A operand: row = column n of the layer, k index = batch row r
    const int kblocks = (K + 31) / 32;
    for (int kb = blockIdx.y * kb_per_wg + wave; kb < min(kblocks, ((int)blockIdx.y + 1) * kb_per_wg); kb += 4) {
        const int k = kb * 32 + lr, kc = min(k, K - 1);
        float b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) b[j] = X[(size_t)min(2 * j + lh, R - 1) * K + kc];   // (rows past R: dy is 0 there)
        hd_f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
        if (k < K) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = n0 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                if (n < N) dW[(size_t)n * K + k] = acc[i];
            }
        }
    }
}
__global__ __launch_bounds__(256) void head_dx_mfma_kernel(const float* __restrict__ dY, const float* __restrict__ W, int R, int K, int N,
                                                           int nper, float* __restrict__ dX) {
    __shared__ float sD[32][128 + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int k = (blockIdx.x * 4 + wave) * 32 + lr, kc = min(k, K - 1);
    const int nbeg = blockIdx.y * nper, nend = min(N, nbeg + nper);
    hd_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int nb = nbeg; nb < nend; nb += 128) {
        __syncthreads();
        for (int e = tid; e < 32 * 128; e += 256) {
            const int r = e >> 7, c = e & 127;
            sD[r][c] = (r < R && nb + c < nend) ? dY[(size_t)r * N + nb + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 2
        for (int j0 = 0; j0 < 64; j0 += 16) {
            float b[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) b[j] = W[(size_t)min(nb + 2 * (j0 + j) + lh, N - 1) * K + kc];   // (past nend: dy is 0)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sD[lr][2 * (j0 + j) + lh], b[j], acc, 0, 0, 0);
        }
    }
    if (k < K) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (r < R) unsafeAtomicAdd(&dX[(size_t)r * K + k], acc[i]);
        }
    }
}

// the 8-columns-per-workgroup kernels: few rows, a long reduction, 16-byte rows
static bool head_wide(int R, int K, const float* X, const float* W) {
    return R <= 32 && K >= 2048 && K % 4 == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
}

}  // namespace pcl
using namespace pcl;

// The split-K forward's partial sums [KS][32][N] live in a workspace of the caller (round 5; before: a stream-ordered allocation of the
// call, which set a process-wide mempool attribute and could not be captured into a graph)
static inline int head_wide_ks(int K) { return K >= 8192 ? 16 : 8; }
static inline size_t head_fwd_ws_bytes(int R, int K, int N) { return (R <= 32 && K >= 2048) ? (size_t)head_wide_ks(K) * 32 * N * sizeof(float) : 0; }

static int head_layer_fwd_impl(const float* X, const float* W, const float* bias, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, int R, int K, int N, int bn_mode, float eps,
                                      float momentum, float slope, float* Ypre, float* OUT, float* mean_out, float* invstd_out,
                                      void* workspace, size_t workspace_bytes, void* stream, const Drop drop) {
    PCL_REQUIRE(X && W && Ypre && OUT, "pcl_head_layer_fwd_f32: null pointer");
    PCL_REQUIRE(R >= 1 && R <= 64 && K >= 1 && N >= 1, "pcl_head_layer_fwd_f32: bad sizes R=%d K=%d N=%d (R <= 64)", R, K, N);
    PCL_REQUIRE((bn_mode & ~7) == 0 && (bn_mode & 3) <= 2 && ((bn_mode & 3) == 0 || (mean_out && invstd_out)) &&
                ((bn_mode & 3) != 2 || (running_mean && running_var)), "pcl_head_layer_fwd_f32: bn_mode=%d", bn_mode);
    hipStream_t st = as_stream(stream);
    const dim3 grid(N), block(256);
    if (head_wide(R, K, X, W)) {
        const int KS = head_wide_ks(K), kper = ((K + KS - 1) / KS + HM_KC - 1) / HM_KC * HM_KC;
        const size_t need = (size_t)KS * 32 * N * sizeof(float);
        if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
            return fail(PCL_EWS, "pcl_head_layer_fwd_f32: workspace %zu < %zu (pcl_head_layer_fwd_workspace_bytes; 16-byte aligned)", workspace ? workspace_bytes : (size_t)0, need);
        float* part = static_cast<float*>(workspace);
        hipLaunchKernelGGL(head_fwd_part_kernel, dim3((N + 31) / 32, KS), block, 0, st, X, W, R, K, N, kper, part);
        hipLaunchKernelGGL(head_fwd_finish_kernel, dim3((N + 3) / 4), block, 0, st, part, KS, bias, gamma, beta, running_mean, running_var, R, N,
                           bn_mode, eps, momentum, slope, Ypre, OUT, mean_out, invstd_out, drop);
    } else if (R <= 32)
        hipLaunchKernelGGL(head_fwd_kernel<32>, grid, block, 0, st, X, W, bias, gamma, beta, running_mean, running_var, R, K, N, bn_mode, eps,
                           momentum, slope, Ypre, OUT, mean_out, invstd_out, drop);
    else
        hipLaunchKernelGGL(head_fwd_kernel<64>, grid, block, 0, st, X, W, bias, gamma, beta, running_mean, running_var, R, K, N, bn_mode, eps,
                           momentum, slope, Ypre, OUT, mean_out, invstd_out, drop);
    return check_launch("pcl_head_layer_fwd_f32");
}

extern "C" int pcl_head_layer_fwd_f32(const float* X, const float* W, const float* bias, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, int R, int K, int N, int bn_mode, float eps,
                                      float momentum, float slope, float* Ypre, float* OUT, float* mean_out, float* invstd_out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    return head_layer_fwd_impl(X, W, bias, gamma, beta, running_mean, running_var, R, K, N, bn_mode, eps, momentum, slope, Ypre, OUT, mean_out,
                               invstd_out, workspace, workspace_bytes, stream, Drop{0.f, 0u, 0u});
}
extern "C" size_t pcl_head_layer_fwd_workspace_bytes(int R, int K, int N) { return (R < 1 || K < 1 || N < 1) ? 0 : head_fwd_ws_bytes(R, K, N); }

static int head_layer_bwd_impl(const float* X, const float* W, const float* dOUT, const float* OUT, const float* Ypre,
                                      const float* gamma, const float* mean, const float* invstd, int R, int K, int N, int bn_mode,
                                      float slope, float* dY_ws, float* dW, float* dbias, float* dgamma, float* dbeta, float* dX,
                                      void* stream, const Drop drop) {
    PCL_REQUIRE(X && W && dOUT && OUT && Ypre && dY_ws && dW, "pcl_head_layer_bwd_f32: null pointer");
    PCL_REQUIRE(R >= 1 && R <= 64 && K >= 1 && N >= 1, "pcl_head_layer_bwd_f32: bad sizes R=%d K=%d N=%d (R <= 64)", R, K, N);
    PCL_REQUIRE((bn_mode & 3) == 0 || (mean && invstd), "pcl_head_layer_bwd_f32: BatchNorm needs mean / invstd");
    hipStream_t st = as_stream(stream);
    const dim3 grid(N), block(256);
    const bool wide = head_wide(R, K, X, W);
    if (wide) {                                   // column part, then dW as 32 x 32 MFMA tiles
        hipLaunchKernelGGL(head_dy_kernel, dim3((N + 3) / 4), block, 0, st, dOUT, OUT, Ypre, gamma, mean, invstd, R, K, N, bn_mode, slope, dY_ws,
                           dbias, dgamma, dbeta, dX, drop);
        const int kblocks = (K + 31) / 32, gy = kblocks >= 128 ? 32 : 8, kb_per_wg = (kblocks + gy - 1) / gy;
        hipLaunchKernelGGL(head_dw_mfma_kernel, dim3((N + 31) / 32, gy), block, 0, st, dY_ws, X, R, K, N, kb_per_wg, dW);
    } else if (R <= 32)
        hipLaunchKernelGGL(head_bwd_col_kernel<32>, grid, block, 0, st, X, dOUT, OUT, Ypre, gamma, mean, invstd, R, K, N, bn_mode & 3, slope, dY_ws,
                           dW, dbias, dgamma, dbeta, dX, drop);
    else
        hipLaunchKernelGGL(head_bwd_col_kernel<64>, grid, block, 0, st, X, dOUT, OUT, Ypre, gamma, mean, invstd, R, K, N, bn_mode & 3, slope, dY_ws,
                           dW, dbias, dgamma, dbeta, dX, drop);
    int rc = check_launch("pcl_head_layer_bwd_f32(col)");
    if (rc || !dX) return rc;
    const dim3 g2((K + 63) / 64, (N + 4 * HD_NS - 1) / (4 * HD_NS));
    if (R <= 32 && K >= 2048) {
        const int NS = N >= 512 ? 2 : 1, nper = ((N + NS - 1) / NS + 127) / 128 * 128;
        hipLaunchKernelGGL(head_dx_mfma_kernel, dim3((K + 127) / 128, NS), block, 0, st, dY_ws, W, R, K, N, nper, dX);
    }
    else if (R <= 32) hipLaunchKernelGGL(head_bwd_dx_kernel<32>, g2, block, 0, st, dY_ws, W, R, K, N, dX);
    else hipLaunchKernelGGL(head_bwd_dx_kernel<64>, g2, block, 0, st, dY_ws, W, R, K, N, dX);
    return check_launch("pcl_head_layer_bwd_f32(dx)");
}

extern "C" int pcl_head_layer_bwd_f32(const float* X, const float* W, const float* dOUT, const float* OUT, const float* Ypre,
                                      const float* gamma, const float* mean, const float* invstd, int R, int K, int N, int bn_mode,
                                      float slope, float* dY_ws, float* dW, float* dbias, float* dgamma, float* dbeta, float* dX,
                                      void* stream) {
    return head_layer_bwd_impl(X, W, dOUT, OUT, Ypre, gamma, mean, invstd, R, K, N, bn_mode, slope, dY_ws, dW, dbias, dgamma, dbeta, dX, stream,
                               Drop{0.f, 0u, 0u});
}

// ---- the whole FC head behind one entry point (networks/cls/pointnet2.py:138-147,:157-158; dgcnn.py:87-93,:117-121) --------
// Up to PCL_HEAD_MAX_LAYERS x [Linear (+bias) -> BatchNorm1d -> (Leaky)ReLU -> Dropout] on R <= 64 rows: the per-layer kernels
// above, launched from ONE call per direction with the buffers carved here.  Backward start is where a training step is
// host-bound (a dozen 5-10 us kernels, each behind its own autograd node): one node + one call instead.
namespace pcl {
struct HeadSave { float* Ypre[PCL_HEAD_MAX_LAYERS]; float* OUT[PCL_HEAD_MAX_LAYERS]; float* mean[PCL_HEAD_MAX_LAYERS]; float* invstd[PCL_HEAD_MAX_LAYERS]; float* fws; size_t fws_bytes; size_t bytes; };
static inline size_t hal(size_t n) { return (n + 255) & ~(size_t)255; }
static HeadSave head_save(const pcl_fc_head_t& d, void* base) {
    HeadSave h = {};
    char* b = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t n) { float* r = b ? reinterpret_cast<float*>(b + off) : nullptr; off += hal(n * sizeof(float)); return r; };
    for (int l = 0; l < d.n_layers; ++l) {
        const size_t rn = (size_t)d.R * d.layer[l].N;
        h.Ypre[l] = take(rn);
        h.OUT[l] = l == d.n_layers - 1 ? nullptr : take(rn);        // the last layer's output is the caller's `out`
        h.mean[l] = take(d.layer[l].N); h.invstd[l] = take(d.layer[l].N);
    }
    // the wide layers' split-K partial sums (forward only; one region, reused layer after layer in stream order)
    size_t wmax = 0;
    for (int l = 0; l < d.n_layers; ++l) { const size_t w = head_fwd_ws_bytes(d.R, d.layer[l].K, d.layer[l].N); wmax = w > wmax ? w : wmax; }
    h.fws_bytes = wmax;
    h.fws = wmax ? take(wmax / sizeof(float)) : nullptr;
    h.bytes = off;
    return h;
}
static size_t head_bwd_tmp(const pcl_fc_head_t& d, float** dY, float** dX0, float** dX1, void* base) {
    int nmax = 0, kmax = 0;
    for (int l = 0; l < d.n_layers; ++l) { nmax = d.layer[l].N > nmax ? d.layer[l].N : nmax; if (l > 0) kmax = d.layer[l].K > kmax ? d.layer[l].K : kmax; }
    char* b = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t n) { float* r = b ? reinterpret_cast<float*>(b + off) : nullptr; off += hal(n * sizeof(float)); return r; };
    float* a = take((size_t)d.R * nmax); float* x0 = take((size_t)d.R * (kmax ? kmax : 1)); float* x1 = take((size_t)d.R * (kmax ? kmax : 1));
    if (dY) *dY = a; if (dX0) *dX0 = x0; if (dX1) *dX1 = x1;
    return off;
}
static int head_validate(const pcl_fc_head_t* dp, const char* who) {
    PCL_REQUIRE(dp, "%s: null descriptor", who);
    const pcl_fc_head_t& d = *dp;
    PCL_REQUIRE(d.struct_bytes == (int32_t)sizeof(pcl_fc_head_t), "%s: descriptor is %d bytes, this library expects %zu", who, d.struct_bytes, sizeof(pcl_fc_head_t));
    PCL_REQUIRE(d.n_layers >= 1 && d.n_layers <= PCL_HEAD_MAX_LAYERS && d.R >= 1 && d.R <= 64, "%s: n_layers=%d R=%d", who, d.n_layers, d.R);
    for (int l = 0; l < d.n_layers; ++l) {
        const pcl_head_layer_t& y = d.layer[l];
        PCL_REQUIRE(y.W && y.K >= 1 && y.N >= 1 && (l == 0 || y.K == d.layer[l - 1].N), "%s: layer %d: K=%d N=%d", who, l, y.K, y.N);
        PCL_REQUIRE(y.drop_p >= 0.f && y.drop_p < 1.f, "%s: layer %d: drop_p=%g", who, l, (double)y.drop_p);
    }
    return PCL_OK;
}
static inline Drop head_drop(const pcl_fc_head_t& d, int l) {
    const unsigned lo = (unsigned)(d.seed & 0xffffffffull) ^ (0x9E3779B9u * (unsigned)(l + 1)), hi = (unsigned)(d.seed >> 32) + 0x7F4A7C15u * (unsigned)(l + 1);
    return Drop{d.layer[l].drop_p, lo, hi};
}
}  // namespace pcl

extern "C" int pcl_fc_head_sizes(const pcl_fc_head_t* d, size_t* save_bytes, size_t* bwd_tmp_bytes) {
    int rc = head_validate(d, "pcl_fc_head_sizes");
    if (rc) return rc;
    if (save_bytes) *save_bytes = head_save(*d, nullptr).bytes;
    if (bwd_tmp_bytes) *bwd_tmp_bytes = head_bwd_tmp(*d, nullptr, nullptr, nullptr, nullptr);
    return PCL_OK;
}

extern "C" int pcl_fc_head_fwd_f32(const pcl_fc_head_t* dp) {
    int rc = head_validate(dp, "pcl_fc_head_fwd_f32");
    if (rc) return rc;
    const pcl_fc_head_t& d = *dp;
    PCL_REQUIRE(d.x && d.out && d.save, "pcl_fc_head_fwd_f32: null x / out / save");
    const HeadSave h = head_save(d, d.save);
    if (d.save_bytes < h.bytes) return fail(PCL_EWS, "pcl_fc_head_fwd_f32: save %zu < %zu", d.save_bytes, h.bytes);
    const float* cur = d.x;
    for (int l = 0; l < d.n_layers; ++l) {
        const pcl_head_layer_t& y = d.layer[l];
        float* out = l == d.n_layers - 1 ? d.out : h.OUT[l];
        rc = head_layer_fwd_impl(cur, y.W, y.bias, y.gamma, y.beta, y.running_mean, y.running_var, d.R, y.K, y.N, y.bn_mode, y.eps, y.momentum,
                                 y.slope, h.Ypre[l], out, h.mean[l], h.invstd[l], h.fws, h.fws_bytes, d.stream, head_drop(d, l));
        if (rc) return rc;
        cur = out;
    }
    return PCL_OK;
}

extern "C" int pcl_fc_head_bwd_f32(const pcl_fc_head_t* dp) {
    int rc = head_validate(dp, "pcl_fc_head_bwd_f32");
    if (rc) return rc;
    const pcl_fc_head_t& d = *dp;
    PCL_REQUIRE(d.x && d.out && d.save && d.tmp && d.gout, "pcl_fc_head_bwd_f32: null x / out / save / tmp / gout");
    const HeadSave h = head_save(d, d.save);
    float *dY, *dXa, *dXb;
    const size_t need = head_bwd_tmp(d, &dY, &dXa, &dXb, d.tmp);
    if (d.save_bytes < h.bytes || d.tmp_bytes < need) return fail(PCL_EWS, "pcl_fc_head_bwd_f32: save %zu < %zu or tmp %zu < %zu", d.save_bytes, h.bytes, d.tmp_bytes, need);
    const float* g = d.gout;
    for (int l = d.n_layers - 1; l >= 0; --l) {
        const pcl_head_layer_t& y = d.layer[l];
        PCL_REQUIRE(y.dW && (!y.bias || y.dbias) && (!(y.bn_mode & 3) || !y.gamma || (y.dgamma && y.dbeta)), "pcl_fc_head_bwd_f32: layer %d: null gradient output", l);
        const float* X = l == 0 ? d.x : h.OUT[l - 1];
        const float* OUT = l == d.n_layers - 1 ? d.out : h.OUT[l];
        float* dX = l == 0 ? d.dx : ((d.n_layers - 1 - l) & 1 ? dXb : dXa);
        rc = head_layer_bwd_impl(X, y.W, g, OUT, h.Ypre[l], y.gamma, h.mean[l], h.invstd[l], d.R, y.K, y.N, y.bn_mode, y.slope, dY, y.dW, y.dbias,
                                 y.dgamma, y.dbeta, dX, d.stream, head_drop(d, l));
        if (rc) return rc;
        g = dX;
    }
    return PCL_OK;
}

// ---- label-smoothed cross entropy of the classifier (train_cls.py:31-51) --------------------------------------------
//   w = one_hot*(1-eps) + (1-one_hot)*eps/(C-1);  loss = -mean_r sum_c w[r,c] log_softmax(x)[r,c]
// One workgroup: a wave per row (lanes over classes), row losses folded through LDS.  Also writes the gradient of the
// mean loss, (softmax - w)/R, so that backward is a scale.  (The reference builds it from ~12 elementwise launches.)
namespace pcl {
__global__ __launch_bounds__(1024) void soft_ce_kernel(const float* __restrict__ x, const int64_t* __restrict__ target, float eps,
                                                       int R, int C, float* __restrict__ loss, float* __restrict__ dx) {
    __shared__ float part[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float lo = C > 1 ? eps / (float)(C - 1) : 0.f, hi = 1.f - eps, invR = 1.f / (float)R;
    float acc = 0.f;
    for (int r = wave; r < R; r += 16) {
        const float* xr = x + (size_t)r * C;
        const int t = (int)target[r];
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, xr[c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += expf(xr[c] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
        const float lse = logf(se);
        float l = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float lp = xr[c] - mx - lse, w = c == t ? hi : lo;
            l -= w * lp;
            if (dx) dx[(size_t)r * C + c] = (expf(lp) - w) * invR;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        acc += l;
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int j = 0; j < 16; ++j) s += part[j];
        *loss = s * invR;
    }
}
}  // namespace pcl

// ---- the same loss over tens of thousands of rows (round 6) --------------------------------------------------------------------------
// Part segmentation takes the cross entropy over every point: nn.cross_entropy_loss(pred [B*N, 50], seg) (train_partseg.py:116) = B * N =
// 32 768 rows.  PyTorch runs it as log-softmax forward / backward + three nll kernels + the layout copies around them (81 us and 7
// launches per PointNet++ part-seg step, profiles/r05_cfg4_kernel_stats.csv).  Here: a wave per row as above, rows strided over the
// grid's waves, the gradient of the MEAN loss written in the same pass; every workgroup leaves the sum of its rows' losses in
// partial[block] (waves in order) and a one-workgroup launch folds the partials in a fixed tree -- no atomics, run-to-run identical.
namespace pcl {
__global__ __launch_bounds__(256) void soft_ce_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ target, float eps,
                                                            int R, int C, float* __restrict__ partial, float* __restrict__ dx) {
    __shared__ float part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float lo = C > 1 ? eps / (float)(C - 1) : 0.f, hi = 1.f - eps, invR = 1.f / (float)R;
    float acc = 0.f;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float* xr = x + (size_t)r * C;
        const int t = (int)target[r];
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, xr[c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += expf(xr[c] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
        const float lse = logf(se);
        float l = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float lp = xr[c] - mx - lse, w = c == t ? hi : lo;
            l -= w * lp;
            if (dx) dx[(size_t)r * C + c] = (expf(lp) - w) * invR;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        acc += l;
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(1024) void soft_ce_finish_kernel(const float* __restrict__ partial, int n, float invR, float* __restrict__ loss) {
    __shared__ float s[1024];
    s[threadIdx.x] = threadIdx.x < n ? partial[threadIdx.x] : 0.f;            // n <= 1024
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = s[0] * invR;
}
}  // namespace pcl

extern "C" int pcl_soft_ce_rows_blocks(int R) { const int b = (R + 31) / 32; return b < 1 ? 1 : (b > 1024 ? 1024 : b); }

extern "C" int pcl_soft_ce_rows_f32(const float* logits, const int64_t* target, float eps, int R, int C, float* partial, float* loss,
                                    float* dlogits, void* stream) {
    PCL_REQUIRE(logits && target && loss && partial, "pcl_soft_ce_rows_f32: null pointer");
    PCL_REQUIRE(R >= 1 && C >= 1 && (size_t)R * C < 0x7fffffffull && eps >= 0.f && eps < 1.f, "pcl_soft_ce_rows_f32: bad sizes R=%d C=%d eps=%g", R, C, (double)eps);
    const int nb = pcl_soft_ce_rows_blocks(R);
    hipLaunchKernelGGL(pcl::soft_ce_rows_kernel, dim3(nb), dim3(256), 0, as_stream(stream), logits, target, eps, R, C, partial, dlogits);
    hipLaunchKernelGGL(pcl::soft_ce_finish_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partial, nb, 1.f / (float)R, loss);
    return check_launch("pcl_soft_ce_rows_f32");
}

extern "C" int pcl_soft_ce_f32(const float* logits, const int64_t* target, float eps, int R, int C, float* loss, float* dlogits,
                               void* stream) {
    PCL_REQUIRE(logits && target && loss, "pcl_soft_ce_f32: null pointer");
    PCL_REQUIRE(R >= 1 && R <= 65536 && C >= 1 && eps >= 0.f && eps < 1.f, "pcl_soft_ce_f32: bad sizes R=%d C=%d eps=%g", R, C, (double)eps);
    hipLaunchKernelGGL(pcl::soft_ce_kernel, dim3(1), dim3(1024), 0, as_stream(stream), logits, target, eps, R, C, loss, dlogits);
    return check_launch("pcl_soft_ce_f32");
}
