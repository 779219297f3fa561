// narrow.hip -- [1x1 conv -> BatchNorm(train) -> (Leaky)ReLU] x 3 stacks whose layers have at most 16 channels, gfx950.
//
// Reference: WeightNet (3 -> 8 -> 8 -> 16 on every grouped row) and DensityNet (1 -> 8 -> 8 -> 1 on every point) of PointConv,
// /root/reference/misc/pointconv_utils.py:186-250: nn.Conv + nn.BatchNorm + relu per layer.
// On the library's GEMM kernels such a stack is 22 launches forward and 22 backward (GEMM, statistics, BatchNorm+activation per
// layer; mask, constants, dW, partial reduce, dX per layer back), each of them launch-bound (a row is 4-64 bytes: 11-45 us per
// launch at 0.5-1.5 TB/s) -- 132 launches and 1.2 ms of a 6.1 ms PointConv step for a few MFLOP.
// A row's whole chain fits a lane's registers (<= 16 + 8 + 8 + 3 floats) and its input is 4-12 bytes, so nothing intermediate is
// stored at all: every pass RECOMPUTES the chain from x.  What forces several passes is BatchNorm: layer l's batch statistics
// need all rows of layer l-1's output.
//   forward : pass k = 0,1,2 runs the chain up to layer k's pre-BatchNorm output and sums (y, y^2) per channel (fp64 from the
//             first addition on: DensityNet sees |mean| >> std); pass 3 writes the stack's output.  Pass k+1 begins by reducing
//             pass k's per-workgroup partial rows (<= 256 rows x 32 doubles, every workgroup on its own: no launch for it).
//   backward: pass 0 sums (du, du*y) of the last layer; pass p = 1,2,3 forms dy of layers 2 .. 3-p (constants from the sums of
//             pass p-1, reduced in its prologue), accumulates dW of layer 3-p (fp32 per lane over its <= a few dozen rows,
//             fp64 from the workgroup reduction on) and the sums of the layer below; a one-workgroup launch reduces the last
//             partials.  dW of the first layer is summed about a pivot row (x - x[0]): sum(dy) = 0 under BatchNorm, and the
//             raw input of DensityNet is a near-constant density.
// 4 + 5 launches instead of 44, each a few microseconds of work.  Only `scale, shift, mean, invstd` per layer survive from
// forward to backward (768 bytes).  No input gradient (both nets read coordinates / densities).
// Called through the per-stack entry points (stack.hip: pcl_mlp_stack_fwd_f32 / _bwd_f32 take this path when narrow_supported).
#include "common.h"

namespace pcl {

constexpr int NW_T = 256;            // threads per workgroup, one row per lane and iteration
constexpr int NW_MAXB = 256;         // workgroups = rows of a partial table
constexpr int NW_C = 16;             // widest layer
constexpr int NW_SQ = 2 * NW_C;      // forward partial row: (sum y | sum y^2); backward: (sum du | sum du*y) ...
constexpr int NW_BP = NW_SQ + 128;   // ... followed by the dW partial of one layer (<= 16 x 8)

struct NwArgs {
    const float* x; int rows; int prev_grid;                  // prev_grid: rows of the partial table the prologue reduces
    const float* W[3]; const float* bias[3]; const float* gamma[3]; const float* beta[3];
    float* rmean[3]; float* rvar[3];
    float eps, momentum, slope, out_slope;
    float* vec;                       // [3][4][NW_C]: scale, shift, mean, invstd (forward -> backward)
    const double* part_in; double* part_out;
    float* out;                       // [rows, C3]
    const float* dz;                  // [rows, C3]
    float* bvec;                      // [3][3][NW_C]: a, k1, k2 of the layers whose sums are already reduced
    float* dW[3]; float* dbias[3]; float* dgamma[3]; float* dbeta[3];
};

template <int C0, int C1, int C2, int C3>
struct Nw {
    static constexpr int oW0 = 0, oB0 = oW0 + C1 * C0, oW1 = oB0 + C1, oB1 = oW1 + C2 * C1, oW2 = oB1 + C2, oB2 = oW2 + C3 * C2;
    static constexpr int oV = (oB2 + C3 + 3) & ~3;                       // per layer 6 vectors of NW_C: scale, beta, mean, a, k1, k2
    static constexpr int oPiv = oV + 3 * 6 * NW_C, nLds = oPiv + 4;
    static constexpr int vo(int l, int which) { return oV + (l * 6 + which) * NW_C; }
};

template <int CI, int CO>
__device__ __forceinline__ void nw_lin(const float (&in)[CI], float (&out)[CO], const float* __restrict__ sW, const float* __restrict__ sB) {
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        float acc = sB[o];
#pragma unroll
        for (int k = 0; k < CI; ++k) acc = fmaf(sW[o * CI + k], in[k], acc);
        out[o] = acc;
    }
}
// BatchNorm + activation of one layer.  sv = the layer's vectors (scale | beta | mean | ...): u = scale*(y - mean) + beta -- the
// subtraction first, so that the rounding is relative to the NORMALISED value (the one-fma form scale*y + shift of the GEMM
// operand loads rounds relative to |mean|/std, hundreds of times coarser on DensityNet's near-constant input: ReLU masks flip).
__device__ __forceinline__ float nw_u(const float* __restrict__ sv, int c, float y) { return fmaf(sv[c], y - sv[2 * NW_C + c], sv[NW_C + c]); }
template <int C>
__device__ __forceinline__ void nw_act(const float (&y)[C], float (&z)[C], const float* __restrict__ sv, float slope) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float u = nw_u(sv, c, y[c]);
        z[c] = u > 0.f ? u : u * slope;
    }
}
// dz -> dy of a layer: du = dz * act'(u); dy = a*du - (k1 + k2*(y - mean))   (the constants of bn_bwd_consts_kernel, mlp.hip)
template <int C>
__device__ __forceinline__ void nw_dy(const float (&dz)[C], const float (&y)[C], float (&dy)[C], const float* __restrict__ sv, float slope) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float u = nw_u(sv, c, y[c]);
        const float du = u > 0.f ? dz[c] : dz[c] * slope;
        dy[c] = fmaf(sv[3 * NW_C + c], du, -fmaf(sv[5 * NW_C + c], y[c] - sv[2 * NW_C + c], sv[4 * NW_C + c]));
    }
}
// gradient w.r.t. the layer's input: dzin[k] = sum_o dy[o] W[o][k]
template <int CI, int CO>
__device__ __forceinline__ void nw_back(const float (&dy)[CO], float (&dzin)[CI], const float* __restrict__ sW) {
#pragma unroll
    for (int k = 0; k < CI; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) acc = fmaf(dy[o], sW[o * CI + k], acc);
        dzin[k] = acc;
    }
}

// wave sums on the DPP network (no LDS traffic): four butterfly steps inside each row of 16 lanes, row_bcast:15 / row_bcast:31
// fold the rows -- lane 63 holds the total (the other lanes partial sums)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float nw_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double nw_dpp_add(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return v + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <typename T>
__device__ __forceinline__ T nw_wave_sum(T v) {
    v = nw_dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = nw_dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = nw_dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = nw_dpp_add<0x140, 0xf>(v);     // row_mirror: every lane holds its row's sum
    v = nw_dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1, 3
    v = nw_dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2, 3: lane 63 = the wave's sum
    return v;
}

// column sums of a partial table [rows <= NW_MAXB][stride] (doubles), columns [0, 32): tot[col], fixed order; all 256 threads.
// Every thread's loads are issued together (one memory round trip, not one per row).
__device__ __forceinline__ void nw_colsum(const double* __restrict__ part, int rows, int stride, double* red /*[8][32]*/, double* tot /*[32]*/) {
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    double v[NW_MAXB / 8];
#pragma unroll
    for (int i = 0; i < NW_MAXB / 8; ++i) {
        const int r = sl + 8 * i;
        v[i] = r < rows ? part[(size_t)r * stride + col] : 0.0;
    }
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < NW_MAXB / 8; ++i) a += v[i];
    red[sl * 32 + col] = a;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) t += red[s * 32 + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}

// ---- forward -----------------------------------------------------------------------------------------------------------
template <int C0, int C1, int C2, int C3, int PASS>
__global__ __launch_bounds__(NW_T) void nw_fwd_kernel(const NwArgs p) {
    using S = Nw<C0, C1, C2, C3>;
    constexpr int CO[3] = {C1, C2, C3};
    __shared__ float sm[S::nLds];
    __shared__ double red[8 * 32];
    __shared__ double tot[32];
    const int tid = threadIdx.x;
    // weights and biases of the layers this pass evaluates
    {
        constexpr int LW = PASS < 3 ? PASS : 2;
        for (int i = tid; i < C1 * C0; i += NW_T) sm[S::oW0 + i] = p.W[0][i];
        for (int i = tid; i < C1; i += NW_T) sm[S::oB0 + i] = p.bias[0] ? p.bias[0][i] : 0.f;
        if (LW >= 1) {
            for (int i = tid; i < C2 * C1; i += NW_T) sm[S::oW1 + i] = p.W[1][i];
            for (int i = tid; i < C2; i += NW_T) sm[S::oB1 + i] = p.bias[1] ? p.bias[1][i] : 0.f;
        }
        if (LW >= 2) {
            for (int i = tid; i < C3 * C2; i += NW_T) sm[S::oW2 + i] = p.W[2][i];
            for (int i = tid; i < C3; i += NW_T) sm[S::oB2 + i] = p.bias[2] ? p.bias[2][i] : 0.f;
        }
    }
    // BatchNorm of the layers below: layer PASS-1 from the partial sums of the pass before, the others from `vec`
    if constexpr (PASS >= 1) {
        constexpr int l = PASS - 1, C = CO[l];
        nw_colsum(p.part_in, p.prev_grid, NW_SQ, red, tot);
        if (tid < C) {
            const double mean = tot[tid] / p.rows;
            double var = tot[NW_C + tid] / p.rows - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)p.eps));
            const float a = p.gamma[l][tid] * invstd;
            const float sh = p.beta[l][tid] - a * (float)mean;
            sm[S::vo(l, 0) + tid] = a; sm[S::vo(l, 1) + tid] = p.beta[l][tid]; sm[S::vo(l, 2) + tid] = (float)mean;
            if (blockIdx.x == 0) {
                float* v = p.vec + l * 4 * NW_C;
                v[tid] = a; v[NW_C + tid] = sh; v[2 * NW_C + tid] = (float)mean; v[3 * NW_C + tid] = invstd;
                if (p.rmean[l]) p.rmean[l][tid] += ((float)mean - p.rmean[l][tid]) * p.momentum;
                if (p.rvar[l]) p.rvar[l][tid] += ((float)var - p.rvar[l][tid]) * p.momentum;
            }
        }
#pragma unroll
        for (int k = 0; k < l; ++k)
            if (tid < CO[k]) {
                sm[S::vo(k, 0) + tid] = p.vec[k * 4 * NW_C + tid]; sm[S::vo(k, 1) + tid] = p.beta[k][tid];
                sm[S::vo(k, 2) + tid] = p.vec[k * 4 * NW_C + 2 * NW_C + tid];
            }
    }
    __syncthreads();
    constexpr int CS = PASS < 3 ? CO[PASS] : 1;
    double s[CS], q[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) { s[c] = 0.0; q[c] = 0.0; }
    for (int row = blockIdx.x * NW_T + tid; row < p.rows; row += gridDim.x * NW_T) {
        asm volatile("" ::: "memory");        // weights are re-read from LDS (broadcast) every row: hoisted out of the loop they cost 200+ registers
        float x[C0];
#pragma unroll
        for (int k = 0; k < C0; ++k) x[k] = p.x[(size_t)row * C0 + k];
        float y0[C1];
        nw_lin<C0, C1>(x, y0, sm + S::oW0, sm + S::oB0);
        if constexpr (PASS == 0) {
#pragma unroll
            for (int c = 0; c < C1; ++c) { s[c] += (double)y0[c]; q[c] = fma((double)y0[c], (double)y0[c], q[c]); }
        } else {
            float z0[C1], y1[C2];
            nw_act<C1>(y0, z0, sm + S::vo(0, 0), p.slope);
            nw_lin<C1, C2>(z0, y1, sm + S::oW1, sm + S::oB1);
            if constexpr (PASS == 1) {
#pragma unroll
                for (int c = 0; c < C2; ++c) { s[c] += (double)y1[c]; q[c] = fma((double)y1[c], (double)y1[c], q[c]); }
            } else {
                float z1[C2], y2[C3];
                nw_act<C2>(y1, z1, sm + S::vo(1, 0), p.slope);
                nw_lin<C2, C3>(z1, y2, sm + S::oW2, sm + S::oB2);
                if constexpr (PASS == 2) {
#pragma unroll
                    for (int c = 0; c < C3; ++c) { s[c] += (double)y2[c]; q[c] = fma((double)y2[c], (double)y2[c], q[c]); }
                } else {
                    float z2[C3];
                    nw_act<C3>(y2, z2, sm + S::vo(2, 0), p.out_slope);
                    float* o = p.out + (size_t)row * C3;
                    if constexpr (C3 % 4 == 0) {
#pragma unroll
                        for (int c = 0; c < C3; c += 4) *reinterpret_cast<float4*>(o + c) = make_float4(z2[c], z2[c + 1], z2[c + 2], z2[c + 3]);
                    } else {
#pragma unroll
                        for (int c = 0; c < C3; ++c) o[c] = z2[c];
                    }
                }
            }
        }
    }
    if constexpr (PASS < 3) {
        // workgroup sums -> this workgroup's partial row (wave sums by shuffles, the four waves in a fixed order)
        __syncthreads();
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            const double a = nw_wave_sum(s[c]), b = nw_wave_sum(q[c]);
            if (lane == 63) { red[wave * 32 + c] = a; red[wave * 32 + NW_C + c] = b; }
        }
        __syncthreads();
        if (tid < NW_SQ) {
            const int c = tid & (NW_C - 1);
            double t = 0.0;
            if (c < CS) t = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
            p.part_out[(size_t)blockIdx.x * NW_SQ + tid] = t;
        }
    }
}

// dW of layer l (n entries) from the partial rows of the pass that accumulated it: workgroup b sums the columns b, b + gridDim.x,
// ..., one partial row per thread (prev_grid <= NW_MAXB = NW_T), wave sums by DPP, the four waves in a fixed order
__device__ __forceinline__ void nw_reduce_dw(const NwArgs& p, int l, int n, double* red4) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int col = blockIdx.x; col < n; col += gridDim.x) {
        const double v = tid < p.prev_grid ? p.part_in[(size_t)tid * NW_BP + NW_SQ + col] : 0.0;
        const double w = nw_wave_sum(v);
        __syncthreads();
        if (lane == 63) red4[wave] = w;
        __syncthreads();
        if (tid == 0) p.dW[l][col] = (float)((red4[0] + red4[1]) + (red4[2] + red4[3]));
    }
}
__global__ __launch_bounds__(NW_T) void nw_bwd_final_kernel(const NwArgs p, int n) {
    __shared__ double red4[4];
    nw_reduce_dw(p, 0, n, red4);
}

// ---- backward ----------------------------------------------------------------------------------------------------------
// PASS 0: sums of the last layer.  PASS p = 1..3: dW of layer L = 3 - p and the sums of layer L - 1; nw_bwd_final_kernel: the last reduce.
template <int C0, int C1, int C2, int C3, int PASS>
__global__ __launch_bounds__(NW_T) void nw_bwd_kernel(const NwArgs p) {
    using S = Nw<C0, C1, C2, C3>;
    constexpr int CI[3] = {C0, C1, C2}, CO[3] = {C1, C2, C3};
    __shared__ float sm[S::nLds];
    __shared__ double red[8 * 32];
    __shared__ double tot[32];
    __shared__ double redw[4 * 128];
    const int tid = threadIdx.x;
    const float slopes[3] = {p.slope, p.slope, p.out_slope};
    (void)slopes;
    {
        for (int i = tid; i < C1 * C0; i += NW_T) sm[S::oW0 + i] = p.W[0][i];
        for (int i = tid; i < C1; i += NW_T) sm[S::oB0 + i] = p.bias[0] ? p.bias[0][i] : 0.f;
        for (int i = tid; i < C2 * C1; i += NW_T) sm[S::oW1 + i] = p.W[1][i];
        for (int i = tid; i < C2; i += NW_T) sm[S::oB1 + i] = p.bias[1] ? p.bias[1][i] : 0.f;
        for (int i = tid; i < C3 * C2; i += NW_T) sm[S::oW2 + i] = p.W[2][i];
        for (int i = tid; i < C3; i += NW_T) sm[S::oB2 + i] = p.bias[2] ? p.bias[2][i] : 0.f;
#pragma unroll
        for (int l = 0; l < 3; ++l)
            if (tid < CO[l]) {
                sm[S::vo(l, 0) + tid] = p.vec[l * 4 * NW_C + tid]; sm[S::vo(l, 1) + tid] = p.beta[l][tid];
                sm[S::vo(l, 2) + tid] = p.vec[l * 4 * NW_C + 2 * NW_C + tid];
            }
        if (tid < C0) sm[S::oPiv + tid] = p.x[tid];
    }
    if constexpr (PASS >= 1) {
        // BatchNorm-backward constants of layer Lq from the sums the pass before left; the layers above it from bvec
        constexpr int Lq = 3 - PASS, C = CO[Lq];
        nw_colsum(p.part_in, p.prev_grid, NW_BP, red, tot);
        if (tid < C) {
            const double s1 = tot[tid], s2 = tot[NW_C + tid];
            const double mu = p.vec[Lq * 4 * NW_C + 2 * NW_C + tid], is = p.vec[Lq * 4 * NW_C + 3 * NW_C + tid];
            const double g = p.gamma[Lq][tid];
            const double dg = (s2 - mu * s1) * is;
            const double a = g * is;
            const float af = (float)a, k1 = (float)(a * s1 / p.rows), k2 = (float)(a * dg * is / p.rows);
            sm[S::vo(Lq, 3) + tid] = af; sm[S::vo(Lq, 4) + tid] = k1; sm[S::vo(Lq, 5) + tid] = k2;
            if (blockIdx.x == 0) {
                p.dgamma[Lq][tid] = (float)dg; p.dbeta[Lq][tid] = (float)s1;
                if (p.dbias[Lq]) p.dbias[Lq][tid] = 0.f;      // a conv bias under training-mode BatchNorm has an exactly zero gradient
                float* b = p.bvec + Lq * 3 * NW_C;
                b[tid] = af; b[NW_C + tid] = k1; b[2 * NW_C + tid] = k2;
            }
        }
#pragma unroll
        for (int l = Lq + 1; l < 3; ++l)
            if (tid < CO[l]) {
                const float* b = p.bvec + l * 3 * NW_C;
                sm[S::vo(l, 3) + tid] = b[tid]; sm[S::vo(l, 4) + tid] = b[NW_C + tid]; sm[S::vo(l, 5) + tid] = b[2 * NW_C + tid];
            }
    }
    if constexpr (PASS >= 2) {
        nw_reduce_dw(p, 4 - PASS, CO[4 - PASS] * CI[4 - PASS], tot);                       // dW of the layer the pass before accumulated
    }
    __syncthreads();
    constexpr int L = PASS == 0 ? 3 : 3 - PASS;                 // layer whose dW this pass accumulates (3: none)
    constexpr int CQ = PASS == 0 ? C3 : (L > 0 ? CO[L - 1] : 1);        // channels of the sums this pass leaves
    constexpr int NWA = L < 3 ? CO[L] * CI[L] : 1;
    double s[CQ], q[CQ];
    float aw[NWA];
#pragma unroll
    for (int c = 0; c < CQ; ++c) { s[c] = 0.0; q[c] = 0.0; }
#pragma unroll
    for (int i = 0; i < NWA; ++i) aw[i] = 0.f;
    for (int row = blockIdx.x * NW_T + tid; row < p.rows; row += gridDim.x * NW_T) {
        asm volatile("" ::: "memory");        // (as in the forward: no hoisting of the LDS-resident weights)
        float x[C0], y0[C1], z0[C1], y1[C2], z1[C2], y2[C3], dz2[C3];
#pragma unroll
        for (int k = 0; k < C0; ++k) x[k] = p.x[(size_t)row * C0 + k];
        nw_lin<C0, C1>(x, y0, sm + S::oW0, sm + S::oB0);
        nw_act<C1>(y0, z0, sm + S::vo(0, 0), p.slope);
        nw_lin<C1, C2>(z0, y1, sm + S::oW1, sm + S::oB1);
        nw_act<C2>(y1, z1, sm + S::vo(1, 0), p.slope);
        nw_lin<C2, C3>(z1, y2, sm + S::oW2, sm + S::oB2);
        {
            const float* g = p.dz + (size_t)row * C3;
            if constexpr (C3 % 4 == 0) {
#pragma unroll
                for (int c = 0; c < C3; c += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(g + c);
                    dz2[c] = t.x; dz2[c + 1] = t.y; dz2[c + 2] = t.z; dz2[c + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < C3; ++c) dz2[c] = g[c];
            }
        }
        if constexpr (PASS == 0) {
#pragma unroll
            for (int c = 0; c < C3; ++c) {
                const float u = nw_u(sm + S::vo(2, 0), c, y2[c]);
                const float du = u > 0.f ? dz2[c] : dz2[c] * p.out_slope;
                s[c] += (double)du; q[c] = fma((double)du, (double)y2[c], q[c]);
            }
        } else {
            float dy2[C3], dz1[C2];
            nw_dy<C3>(dz2, y2, dy2, sm + S::vo(2, 0), p.out_slope);
            if constexpr (L == 2) {
#pragma unroll
                for (int o = 0; o < C3; ++o)
#pragma unroll
                    for (int k = 0; k < C2; ++k) aw[o * C2 + k] = fmaf(dy2[o], z1[k], aw[o * C2 + k]);
            }
            nw_back<C2, C3>(dy2, dz1, sm + S::oW2);
            if constexpr (L == 2) {
#pragma unroll
                for (int c = 0; c < C2; ++c) {
                    const float u = nw_u(sm + S::vo(1, 0), c, y1[c]);
                    const float du = u > 0.f ? dz1[c] : dz1[c] * p.slope;
                    s[c] += (double)du; q[c] = fma((double)du, (double)y1[c], q[c]);
                }
            } else {
                float dy1[C2], dz0[C1];
                nw_dy<C2>(dz1, y1, dy1, sm + S::vo(1, 0), p.slope);
                if constexpr (L == 1) {
#pragma unroll
                    for (int o = 0; o < C2; ++o)
#pragma unroll
                        for (int k = 0; k < C1; ++k) aw[o * C1 + k] = fmaf(dy1[o], z0[k], aw[o * C1 + k]);
                }
                nw_back<C1, C2>(dy1, dz0, sm + S::oW1);
                if constexpr (L == 1) {
#pragma unroll
                    for (int c = 0; c < C1; ++c) {
                        const float u = nw_u(sm + S::vo(0, 0), c, y0[c]);
                        const float du = u > 0.f ? dz0[c] : dz0[c] * p.slope;
                        s[c] += (double)du; q[c] = fma((double)du, (double)y0[c], q[c]);
                    }
                } else {
                    float dy0[C1];
                    nw_dy<C1>(dz0, y0, dy0, sm + S::vo(0, 0), p.slope);
#pragma unroll
                    for (int o = 0; o < C1; ++o)
#pragma unroll
                        for (int k = 0; k < C0; ++k) aw[o * C0 + k] = fmaf(dy0[o], x[k] - sm[S::oPiv + k], aw[o * C0 + k]);
                }
            }
        }
    }
    // workgroup sums -> partial row [NW_BP]: (s | q | dW)
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    if constexpr (L > 0) {
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            const double a = nw_wave_sum(s[c]), b = nw_wave_sum(q[c]);
            if (lane == 63) { red[wave * 32 + c] = a; red[wave * 32 + NW_C + c] = b; }
        }
    }
    if constexpr (L < 3) {
#pragma unroll
        for (int i = 0; i < NWA; ++i) {
            const float a = nw_wave_sum(aw[i]);           // (fp32 inside the wave: 64 partial sums of a few rows each; fp64 from here on)
            if (lane == 63) redw[wave * 128 + i] = (double)a;
        }
    }
    __syncthreads();
    double* po = p.part_out + (size_t)blockIdx.x * NW_BP;
    if (tid < NW_SQ) {
        const int c = tid & (NW_C - 1);
        double t = 0.0;
        if (L > 0 && c < CQ) t = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
        po[tid] = t;
    }
    if (L < 3 && tid < NWA) po[NW_SQ + tid] = (redw[tid] + redw[128 + tid]) + (redw[256 + tid] + redw[384 + tid]);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static inline int nw_shape(const pcl_mlp_stack_t& d) {
    if (d.n_layers != 3) return 0;
    if (d.c[0] == 3 && d.c[1] == 8 && d.c[2] == 8 && d.c[3] == 16) return 1;
    if (d.c[0] == 1 && d.c[1] == 8 && d.c[2] == 8 && d.c[3] == 1) return 2;
    return 0;
}
bool narrow_supported(const pcl_mlp_stack_t& d) {
    return path_switches().narrow_stacks && nw_shape(d) && !d.grouped && d.pool == 0 && !d.need_dx && !d.defer_act && d.P >= 1;
}
size_t narrow_save_bytes() { return (size_t)3 * 4 * NW_C * sizeof(float); }
size_t narrow_fwd_tmp_bytes() { return (size_t)2 * NW_MAXB * NW_SQ * sizeof(double); }
size_t narrow_bwd_tmp_bytes() { return (size_t)2 * NW_MAXB * NW_BP * sizeof(double) + (size_t)3 * 3 * NW_C * sizeof(float); }

static NwArgs nw_args(const pcl_mlp_stack_t& d) {
    NwArgs a = {};
    a.x = d.x; a.rows = d.P;
    for (int l = 0; l < 3; ++l) {
        a.W[l] = d.layer[l].W; a.bias[l] = d.layer[l].bias; a.gamma[l] = d.layer[l].gamma; a.beta[l] = d.layer[l].beta;
        a.rmean[l] = d.layer[l].running_mean; a.rvar[l] = d.layer[l].running_var;
        a.dW[l] = d.layer[l].dW; a.dbias[l] = d.layer[l].dbias; a.dgamma[l] = d.layer[l].dgamma; a.dbeta[l] = d.layer[l].dbeta;
    }
    a.eps = d.eps; a.momentum = d.momentum; a.slope = d.slope; a.out_slope = d.out_slope;
    a.vec = static_cast<float*>(d.save);
    return a;
}
static inline int nw_grid(int rows) {
    int g = (rows + NW_T - 1) / NW_T;
    return g > NW_MAXB ? NW_MAXB : (g < 1 ? 1 : g);
}

template <int C0, int C1, int C2, int C3>
static int nw_fwd_t(NwArgs a, double* part, hipStream_t st) {
    const int g = nw_grid(a.rows);
    double* pa = part; double* pb = part + (size_t)NW_MAXB * NW_SQ;
    a.prev_grid = g;
    a.part_out = pa;
    hipLaunchKernelGGL((nw_fwd_kernel<C0, C1, C2, C3, 0>), dim3(g), dim3(NW_T), 0, st, a);
    a.part_in = pa; a.part_out = pb;
    hipLaunchKernelGGL((nw_fwd_kernel<C0, C1, C2, C3, 1>), dim3(g), dim3(NW_T), 0, st, a);
    a.part_in = pb; a.part_out = pa;
    hipLaunchKernelGGL((nw_fwd_kernel<C0, C1, C2, C3, 2>), dim3(g), dim3(NW_T), 0, st, a);
    a.part_in = pa; a.part_out = nullptr;
    hipLaunchKernelGGL((nw_fwd_kernel<C0, C1, C2, C3, 3>), dim3(g), dim3(NW_T), 0, st, a);
    return check_launch("pcl_mlp_stack_fwd_f32(narrow)");
}
template <int C0, int C1, int C2, int C3>
static int nw_bwd_t(NwArgs a, double* part, hipStream_t st) {
    const int g = nw_grid(a.rows);
    double* pa = part; double* pb = part + (size_t)NW_MAXB * NW_BP;
    a.prev_grid = g;
    a.part_out = pa;
    hipLaunchKernelGGL((nw_bwd_kernel<C0, C1, C2, C3, 0>), dim3(g), dim3(NW_T), 0, st, a);
    a.part_in = pa; a.part_out = pb;
    hipLaunchKernelGGL((nw_bwd_kernel<C0, C1, C2, C3, 1>), dim3(g), dim3(NW_T), 0, st, a);
    a.part_in = pb; a.part_out = pa;
    hipLaunchKernelGGL((nw_bwd_kernel<C0, C1, C2, C3, 2>), dim3(g), dim3(NW_T), 0, st, a);
    a.part_in = pa; a.part_out = pb;
    hipLaunchKernelGGL((nw_bwd_kernel<C0, C1, C2, C3, 3>), dim3(g), dim3(NW_T), 0, st, a);
    a.part_in = pb; a.part_out = nullptr;
    hipLaunchKernelGGL(nw_bwd_final_kernel, dim3(C1 * C0), dim3(NW_T), 0, st, a, C1 * C0);
    return check_launch("pcl_mlp_stack_bwd_f32(narrow)");
}

int narrow_fwd(const pcl_mlp_stack_t& d) {
    NwArgs a = nw_args(d);
    a.out = d.out;
    hipStream_t st = as_stream(d.stream);
    double* part = static_cast<double*>(d.tmp);
    return nw_shape(d) == 1 ? nw_fwd_t<3, 8, 8, 16>(a, part, st) : nw_fwd_t<1, 8, 8, 1>(a, part, st);
}
int narrow_bwd(const pcl_mlp_stack_t& d) {
    NwArgs a = nw_args(d);
    a.dz = d.gout;
    hipStream_t st = as_stream(d.stream);
    double* part = static_cast<double*>(d.tmp);
    a.bvec = reinterpret_cast<float*>(part + (size_t)2 * NW_MAXB * NW_BP);
    return nw_shape(d) == 1 ? nw_bwd_t<3, 8, 8, 16>(a, part, st) : nw_bwd_t<1, 8, 8, 1>(a, part, st);
}

}  // namespace pcl
