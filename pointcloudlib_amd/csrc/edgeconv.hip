// edgeconv.hip -- DGCNN's EdgeConv without the edge tensor, gfx950.
//
// Reference (/root/reference/networks/cls/dgcnn.py:29-50, :72-83, :100-111): gather the k nearest neighbours of every
// point, build e[i,j] = [x_nbr - x_i, x_i] (a [B,N,k,2C] tensor, 671 MB at the last stage), Conv2d 1x1 (no bias) +
// BatchNorm + LeakyReLU(0.2), max over the k neighbours.
// The 1x1 conv is linear, so with W = [Wa | Wb]
//     y[i,j] = Wa (x_nbr - x_i) + Wb x_i = U[nbr(i,j)] + V[i],   U = x Wa^T,  V = x (Wb - Wa)^T,
// one GEMM over the N points instead of the N*k edges (k = 20 times fewer flops, no edge tensor at all).  What is left
// is HBM/L2-bound streaming, done here:
//   * edgeconv_gather_kernel: y = U[nbr] + V on the fly; per channel the BatchNorm batch sums (fp64 partial rows, same
//     workspace layout as the GEMM epilogues in mlp.hip) and, per point, max/min of y over the neighbours with their
//     positions (the sign of the BatchNorm scale is not known yet; pcl_group_minmax_finalize_f32 picks afterwards);
//   * backward of BatchNorm + max for every edge, d(i,j) = [j == arg] a gz[i] - k1 - k2 (U[nbr] + V[i] - mean), summed into
//     dV[i] over the edges of i and into dU[n] over the edges that END in n.  Linear again, so neither sum needs the edges:
//         dV[i] = a gz[i] - k k1 - k2 (k (V[i] - mean) + SU[i]),         SU[i] = sum_j U[nbr(i,j)]   (from the forward gather)
//         dU[n] = hits[n] - deg(n) k1 - k2 (deg(n) (U[n] - mean) + SV[n]), SV[n] = sum_{i : n in nbr(i)} V[i]
//     hits[n] = sum of a gz[i] over the (i, channel) pairs whose arg neighbour is n: one atomic per point and channel
//     instead of one per edge and channel; SV walks the TRANSPOSED neighbour lists (knn_transpose_kernel: per cloud a
//     counting sort of the N*k edges by end point, lists sorted by source so that the sum order is fixed).
//     edgeconv_scatter_kernel (one atomic per edge and channel, 1.08 ms per DGCNN step) stays as the path without lists.
// Summation order differs from the reference's conv (y is formed from two fp32 products instead of one), within 1e-6.
#include "common.h"

namespace pcl {

constexpr int EC_PB = 32;        // points per workgroup == BatchNorm partial rows per 32 points

// UV [B*N, 2C] (U | V), idx [B*N, k] (neighbour index within the cloud).
// HILO (round 5): UVlo [B*N, 2C] holds the residuals of U | V (pcl_frag_linear_fwd_f32's Y_lo: U = UV + UVlo to ~2^-48).  y = U[nbr] + V[i]
// is a DIFFERENCE of two large products when neighbours are close in feature space (y = Wa (x_nbr - x_i) + Wb x_i with |Wa x| >> |Wa (x_nbr
// - x_i)| in the later stages): the fp32 rounding of U and V, each relative to |U|, is what an fp32 y is then uncertain by -- several times
// the rounding of a y formed from the edge itself, and enough to pick other max-pool winners than the fp64 evaluation does more often
// than the edge form does.  With the residuals y = (U + V) + (Ulo + Vlo) is the fp32 rounding of the exact edge value.
template <bool HILO>
__global__ __launch_bounds__(256) void edgeconv_gather_kernel(const float* __restrict__ UV, const float* __restrict__ UVlo, const int32_t* __restrict__ idx,
                                                              int N, int k, int C, size_t P /* B*N */, float* __restrict__ ymax,
                                                              float* __restrict__ ymin, int32_t* __restrict__ jmax,
                                                              int32_t* __restrict__ jmin, double* __restrict__ stats,
                                                              float* __restrict__ sumU) {
    extern __shared__ double ssum[];                 // [2][C]
    const int tid = threadIdx.x;
    for (int c = tid; c < 2 * C; c += 256) ssum[c] = 0.0;
    __syncthreads();
    // (blocks of a cloud on one XCD: the U rows they gather stay in that L2; stats rows keep the logical block order)
    const unsigned lb = (N % EC_PB == 0) ? xcd_cloud_block(blockIdx.x, gridDim.x, N / EC_PB) : blockIdx.x;
    const size_t p0 = (size_t)lb * EC_PB;
    const int items = EC_PB * C;
    for (int e = tid; e < items; e += 256) {
        const int pi = e / C, c = e - pi * C;
        const size_t p = p0 + pi;
        if (p >= P) break;
        const size_t base = (p / N) * N;             // first point of this cloud
        const float v = UV[p * 2 * C + C + c];
        const float vlo = HILO ? UVlo[p * 2 * C + C + c] : 0.f;
        const int32_t* I = idx + p * k;
        float vmax = -INFINITY, vmin = INFINITY;
        int imax = 0, imin = 0;
        double s = 0.0, q = 0.0;
        float su = 0.f;
        int j = 0;
        constexpr int GF = HILO ? 5 : 10;            // gathers in flight per lane (k = 20 / 40: whole rounds): 4 -> 10 took the C = 256 stage
        for (; j + GF <= k; j += GF) {               // from 167 to 137 us -- the kernel waits on L2 round trips, not on bandwidth
            float u[GF], ul[HILO ? GF : 1];
#pragma unroll
            for (int t = 0; t < GF; ++t) {
                const size_t o = (base + I[j + t]) * 2 * C + c;
                u[t] = UV[o];
                if constexpr (HILO) ul[t] = UVlo[o];
            }
#pragma unroll
            for (int t = 0; t < GF; ++t) {
                su += u[t];
                float y = u[t] + v;
                if constexpr (HILO) y += ul[t] + vlo;
                s += (double)y; q += (double)y * (double)y;
                if (y > vmax) { vmax = y; imax = j + t; }
                if (y < vmin) { vmin = y; imin = j + t; }
            }
        }
        for (; j < k; ++j) {
            const size_t o = (base + I[j]) * 2 * C + c;
            const float u = UV[o];
            float y = u + v;
            if constexpr (HILO) y += UVlo[o] + vlo;
            su += u;
            s += (double)y; q += (double)y * (double)y;
            if (y > vmax) { vmax = y; imax = j; }
            if (y < vmin) { vmin = y; imin = j; }
        }
        const size_t o = p * C + c;
        ymax[o] = vmax; ymin[o] = vmin; jmax[o] = imax; jmin[o] = imin;
        if (sumU) sumU[o] = su;
        atomicAdd(&ssum[c], s); atomicAdd(&ssum[C + c], q);
    }
    __syncthreads();
    double* dst = stats + (size_t)lb * 2 * C;
    for (int c = tid; c < 2 * C; c += 256) dst[c] = ssum[c];
}

// dUV [B*N, 2C]: the U half must be zero on entry (atomics), the V half is written.
__global__ __launch_bounds__(256) void edgeconv_scatter_kernel(const float* __restrict__ UV, const int32_t* __restrict__ idx,
                                                               const float* __restrict__ gz, const int32_t* __restrict__ arg,
                                                               const float* __restrict__ a_, const float* __restrict__ k1_,
                                                               const float* __restrict__ k2_, const float* __restrict__ mu_,
                                                               int N, int k, int C, size_t P, float* __restrict__ dUV) {
    const int tid = threadIdx.x;
    const size_t p0 = (size_t)blockIdx.x * EC_PB;
    const int items = EC_PB * C;
    for (int e = tid; e < items; e += 256) {
        const int pi = e / C, c = e - pi * C;
        const size_t p = p0 + pi;
        if (p >= P) break;
        const size_t base = (p / N) * N;
        const float v = UV[p * 2 * C + C + c];
        const float a = a_[c], k1 = k1_[c], k2 = k2_[c], mu = mu_[c];
        const float g = a * gz[p * C + c];
        const int ja = arg[p * C + c];
        const int32_t* I = idx + p * k;
        float dv = 0.f;
        int j = 0;
        for (; j + 4 <= k; j += 4) {
            int n[4];
            float u[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { n[t] = I[j + t]; u[t] = UV[(base + n[t]) * 2 * C + c]; }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float y = u[t] + v;
                const float d = (j + t == ja ? g : 0.f) - fmaf(k2, y - mu, k1);
                dv += d;
                unsafeAtomicAdd(&dUV[(base + n[t]) * 2 * C + c], d);
            }
        }
        for (; j < k; ++j) {
            const int n = I[j];
            const float y = UV[(base + n) * 2 * C + c] + v;
            const float d = (j == ja ? g : 0.f) - fmaf(k2, y - mu, k1);
            dv += d;
            unsafeAtomicAdd(&dUV[(base + n) * 2 * C + c], d);
        }
        dUV[p * 2 * C + C + c] = dv;
    }
}

// ---- transposed neighbour lists -------------------------------------------------------------------------------------
// idx [B,N,k] (neighbour n of point i, position j) -> for every point n of a cloud the SOURCES i of the edges ending in
// n: in_off [B*N+1] (offsets into in_src, global), in_src [B*N*k] (source point index within the cloud), each list
// ascending.  One workgroup per cloud: LDS histogram, LDS scan, fill through LDS cursors (list order as the atomics
// fall); knn_lists_sort_kernel puts every list in ascending order so that sums over a list have a fixed order.
constexpr int KT_MAXN = 8192;
__global__ __launch_bounds__(1024) void knn_transpose_kernel(const int32_t* __restrict__ idx, int N, int k, int B,
                                                             int32_t* __restrict__ in_off, int32_t* __restrict__ in_src) {
    __shared__ int cnt[KT_MAXN];
    __shared__ int part[1024];
    const int b = blockIdx.x, t = threadIdx.x;
    const int E = N * k;
    const int32_t* I = idx + (size_t)b * E;
    int32_t* S = in_src + (size_t)b * E;
    for (int n = t; n < N; n += 1024) cnt[n] = 0;
    __syncthreads();
    for (int e = t; e < E; e += 1024) atomicAdd(&cnt[I[e]], 1);
    __syncthreads();
    // exclusive scan of cnt[0..N): per-thread chunks + scan of the chunk sums
    const int per = (N + 1023) / 1024, lo = min(t * per, N), hi = min(lo + per, N);
    int s = 0;
    for (int n = lo; n < hi; ++n) s += cnt[n];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int n = lo; n < hi; ++n) {
        const int c = cnt[n];
        in_off[(size_t)b * N + n] = b * E + run;
        cnt[n] = run;                                  // becomes the fill cursor
        run += c;
    }
    if (b == B - 1 && t == 1023) in_off[(size_t)B * N] = B * E;
    __syncthreads();
    for (int e = t; e < E; e += 1024) {
        const int pos = atomicAdd(&cnt[I[e]], 1);
        S[pos] = e / k;
    }
}

// Round 5: the same lists without the sort.  A point's list ascends with the SOURCE index, and a source names a point at most once (k-NN
// neighbours are distinct), so the lists come out sorted if the sources are filed in order: each of the workgroup's waves owns a contiguous
// range of the cloud's sources, counts its edges per end point (cnt[w][n]), the counts become start positions (a scan over the points, then
// over the waves), and every wave files ITS sources one after the other (a wave's LDS operations execute in order; eight sources' index
// rows are requested ahead).  One launch instead of two (transpose 20.9 + sort 11.4 us per EdgeConv stage of DGCNN cls).  Should a row name a
// point twice, both edges are filed, in an unspecified mutual order.
__global__ __launch_bounds__(1024) void knn_transpose_ordered_kernel(const int32_t* __restrict__ idx, int N, int k, int B, int NWV,
                                                                     int32_t* __restrict__ in_off, int32_t* __restrict__ in_src) {
    extern __shared__ int kt_lds[];                        // cnt [NWV][N] | tot [N]
    __shared__ int wtot[16];
    const int b = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    int* cnt = kt_lds; int* tot = kt_lds + NWV * N;
    const int E = N * k;
    const int32_t* I = idx + (size_t)b * E;
    int32_t* S = in_src + (size_t)b * E;
    const int wv = min(wave, NWV - 1);                     // (waves beyond NWV idle in the counting / filing phases)
    const bool wact = wave < NWV;
    const int qpw = (N + NWV - 1) / NWV, q0 = min(N, wv * qpw), q1 = min(N, q0 + qpw);
    for (int i = t; i < NWV * N; i += 1024) cnt[i] = 0;
    __syncthreads();
    int* mycnt = cnt + wv * N;
    if (wact) for (int e = q0 * k + lane; e < q1 * k; e += 64) atomicAdd(&mycnt[I[e]], 1);
    __syncthreads();
    const int per = (N + 1023) / 1024, i0 = min(N, t * per), i1 = min(N, i0 + per);
    int s = 0;
    for (int i = i0; i < i1; ++i) { int v = 0; for (int w = 0; w < NWV; ++w) v += cnt[w * N + i]; tot[i] = v; s += v; }
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int run = inc - s;
    for (int w = 0; w < wave; ++w) run += wtot[w];
    for (int i = i0; i < i1; ++i) {
        in_off[(size_t)b * N + i] = b * E + run;
        int pos = run;
        for (int w = 0; w < NWV; ++w) { const int c = cnt[w * N + i]; cnt[w * N + i] = pos; pos += c; }
        run += tot[i];
    }
    if (b == B - 1 && t == 1023) in_off[(size_t)B * N] = B * E;
    __syncthreads();
    if (wact) {
        constexpr int QA = 8;                              // sources whose index rows are in flight
        for (int qb = q0; qb < q1; qb += QA) {
            for (int j0 = 0; j0 < k; j0 += 64) {           // (k > 64: the row in pieces; still one source after the other per piece ...)
                int nb[QA];
#pragma unroll
                for (int u = 0; u < QA; ++u) nb[u] = (qb + u < q1 && j0 + lane < k) ? I[(size_t)(qb + u) * k + j0 + lane] : -1;
#pragma unroll
                for (int u = 0; u < QA; ++u) {
                    if (nb[u] >= 0) S[atomicAdd(&mycnt[nb[u]], 1)] = qb + u;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
}

// Bitonic sort of 64*R ints held as v[r] of lane l = element r*64 + l (ascending).
template <int R>
__device__ __forceinline__ void wave_sort(int (&v)[R], int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64 * R; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int pr = r ^ (j >> 6);
                    if (pr > r) {
                        const bool up = (((r * 64 + lane) & k2) == 0);
                        const int lo = min(v[r], v[pr]), hi = max(v[r], v[pr]);
                        v[r] = up ? lo : hi;
                        v[pr] = up ? hi : lo;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int other = __shfl_xor(v[r], j);
                    const bool up = (((r * 64 + lane) & k2) == 0), lower = (lane & j) == 0;
                    v[r] = (lower == up) ? min(v[r], other) : max(v[r], other);
                }
            }
        }
    }
}

// Every list ascending (the fill order above depends on the LDS atomics): one wave per list, in registers up to 256
// entries, one lane by insertion beyond that (in-degrees of a kNN graph average k).
__global__ __launch_bounds__(256) void knn_lists_sort_kernel(const int32_t* __restrict__ in_off, size_t P, int32_t* __restrict__ in_src) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t p = (size_t)blockIdx.x * 4 + wave;
    if (p >= P) return;
    const int beg = in_off[p], len = in_off[p + 1] - beg;
    int32_t* S = in_src + beg;
    if (len <= 1) return;
    if (len <= 64) {
        int v[1] = {lane < len ? S[lane] : INT_MAX};
        wave_sort<1>(v, lane);
        if (lane < len) S[lane] = v[0];
    } else if (len <= 256) {
        int v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = r * 64 + lane < len ? S[r * 64 + lane] : INT_MAX;
        wave_sort<4>(v, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r * 64 + lane < len) S[r * 64 + lane] = v[r];
    } else if (lane == 0) {
        for (int a = 1; a < len; ++a) {
            const int x = S[a];
            int q = a - 1;
            while (q >= 0 && S[q] > x) { S[q + 1] = S[q]; --q; }
            S[q + 1] = x;
        }
    }
}

// dUV[:, :C] += a gz[i] at the arg neighbour of (i, c); the U half must be zero on entry.  The V half is written:
//   dV[i] = a gz[i] - k k1 - k2 (k (V[i] - mean) + SU[i])
__global__ __launch_bounds__(256) void edgeconv_hits_kernel(const float* __restrict__ UV, const int32_t* __restrict__ idx,
                                                            const float* __restrict__ gz, const int32_t* __restrict__ arg,
                                                            const float* __restrict__ sumU, const float* __restrict__ a_,
                                                            const float* __restrict__ k1_, const float* __restrict__ k2_,
                                                            const float* __restrict__ mu_, int N, int k, int C, size_t P,
                                                            float* __restrict__ dUV) {
    const size_t total = P * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t p = e / C;
        const int c = (int)(e - p * C);
        const size_t base = (p / N) * N;
        const float g = a_[c] * gz[e];
        const int n = idx[p * k + arg[e]];
        if (g != 0.f) unsafeAtomicAdd(&dUV[(base + n) * 2 * C + c], g);
        const float v = UV[p * 2 * C + C + c];
        dUV[p * 2 * C + C + c] = g - (float)k * k1_[c] - k2_[c] * fmaf((float)k, v - mu_[c], sumU[e]);
    }
}

// The same through LDS: workgroup (cloud b, chunk of CH channels) owns hits[:, c0..c0+CH) of its cloud -- LDS float
// atomics instead of 64 scattered global ones per wave instruction (arg differs per channel, so a wave's 64 lanes end in
// 64 different rows: 520 us per DGCNN step with global atomics), then plain coalesced stores: no memset either.
__global__ __launch_bounds__(1024) void edgeconv_hits_lds_kernel(const float* __restrict__ UV, const int32_t* __restrict__ idx,
                                                                const float* __restrict__ gz, const int32_t* __restrict__ arg,
                                                                const float* __restrict__ sumU, const float* __restrict__ a_,
                                                                const float* __restrict__ k1_, const float* __restrict__ k2_,
                                                                const float* __restrict__ mu_, int N, int k, int C, int CH,
                                                                float* __restrict__ dUV) {
    extern __shared__ float hits[];                  // [N][CH]
    const int b = blockIdx.x, c0 = blockIdx.y * CH, tid = threadIdx.x;
    const int n_el = N * CH;
    for (int e = tid; e < n_el; e += 1024) hits[e] = 0.f;
    __syncthreads();
    constexpr int HU = 4;                             // (eight: 98 -> 103 us per stage)
    for (int e0 = tid; e0 < n_el; e0 += HU * 1024) {   // HU independent (point, channel) pairs in flight per lane
        float g[HU], v[HU], su[HU];
        int n[HU], cc[HU];
        size_t o2[HU];
        bool ok[HU];
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int e = e0 + u * 1024;
            const int i = min(e, n_el - 1) / CH;
            cc[u] = min(e, n_el - 1) - i * CH;
            const int c = c0 + cc[u];
            ok[u] = e < n_el && c < C;
            const size_t p = (size_t)b * N + i, o = p * C + min(c, C - 1);
            o2[u] = p * 2 * C + C + min(c, C - 1);
            g[u] = a_[min(c, C - 1)] * gz[o];
            n[u] = idx[p * k + arg[o]];
            v[u] = UV[o2[u]];
            su[u] = sumU[o];
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            if (!ok[u]) continue;
            const int c = c0 + cc[u];
            if (g[u] != 0.f) atomicAdd(&hits[n[u] * CH + cc[u]], g[u]);
            dUV[o2[u]] = g[u] - (float)k * k1_[c] - k2_[c] * fmaf((float)k, v[u] - mu_[c], su[u]);
        }
    }
    __syncthreads();
    for (int e = tid; e < n_el; e += 1024) {
        const int n = e / CH, cc = e - n * CH;
        if (c0 + cc < C) dUV[((size_t)b * N + n) * 2 * C + c0 + cc] = hits[e];
    }
}

// dU[n] = hits[n] (already in place) - deg k1 - k2 (deg (U[n] - mean) + sum over the list of n of V[src]); one wave per
// point, lanes over channels, four list entries in flight.
__global__ __launch_bounds__(256) void edgeconv_insum_kernel(const float* __restrict__ UV, const int32_t* __restrict__ in_off,
                                                             const int32_t* __restrict__ in_src, const float* __restrict__ k1_,
                                                             const float* __restrict__ k2_, const float* __restrict__ mu_,
                                                             int N, int C, size_t P, float* __restrict__ dUV) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t p = (size_t)((N & 3) == 0 ? xcd_cloud_block(blockIdx.x, gridDim.x, N / 4) : blockIdx.x) * 4 + wave;
    if (p >= P) return;
    const size_t base = (p / N) * N;
    const int beg = in_off[p], end = in_off[p + 1];
    const float deg = (float)(end - beg);
    for (int c = lane; c < C; c += 64) {
        float sv = 0.f;
        int e = beg;
        for (; e + 8 <= end; e += 8) {               // eight gathers in flight, summed as the two groups of four below would be
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = UV[(base + in_src[e + t]) * 2 * C + C + c];
            sv += (v[0] + v[1]) + (v[2] + v[3]);
            sv += (v[4] + v[5]) + (v[6] + v[7]);
        }
        for (; e + 4 <= end; e += 4) {
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = UV[(base + in_src[e + t]) * 2 * C + C + c];
            sv += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (; e < end; ++e) sv += UV[(base + in_src[e]) * 2 * C + C + c];
        const size_t o = p * 2 * C + c;
        dUV[o] = dUV[o] - deg * k1_[c] - k2_[c] * fmaf(deg, UV[o] - mu_[c], sv);
    }
}

// W = [Wa | Wb] [Co][2C] of the stage's conv -> the point GEMM's weight Wcat = [Wa ; Wb - Wa] [2Co][C] (back = 0), and the
// gradient of Wcat -> the gradient of W: dWa = dWcat_top - dWcat_bottom, dWb = dWcat_bottom (back = 1).  One launch each
// instead of a slice, a subtraction and a concatenation in PyTorch per stage and direction.
__global__ __launch_bounds__(256) void edgeconv_wcat_kernel(const float* __restrict__ src, int Co, int C, int back, float* __restrict__ dst) {
    const int n = Co * C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int o = i / C, c = i - o * C;
        if (!back) {
            const float wa = src[(size_t)o * 2 * C + c], wb = src[(size_t)o * 2 * C + C + c];
            dst[i] = wa; dst[(size_t)n + i] = wb - wa;
        } else {
            const float top = src[i], bot = src[(size_t)n + i];
            dst[(size_t)o * 2 * C + c] = top - bot; dst[(size_t)o * 2 * C + C + c] = bot;
        }
    }
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_edgeconv_wcat_f32(const float* src, int Co, int C, int backward, float* dst, void* stream) {
    PCL_REQUIRE(src && dst && Co >= 1 && C >= 1, "pcl_edgeconv_wcat_f32: bad arguments");
    int blocks = (Co * C + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(edgeconv_wcat_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, Co, C, backward ? 1 : 0, dst);
    return check_launch("pcl_edgeconv_wcat_f32");
}

extern "C" int pcl_edgeconv_stat_rows(int B, int N) {
    if (B < 1 || N < 1) return 0;
    return (int)(((size_t)B * N + EC_PB - 1) / EC_PB);
}

extern "C" int pcl_edgeconv_gather_f32(const float* UV, const int32_t* idx, int B, int N, int k, int C, float* ymax, float* ymin,
                                       int32_t* jmax, int32_t* jmin, double* stats_ws, float* sumU, void* stream) {
    PCL_REQUIRE(UV && idx && ymax && ymin && jmax && jmin && stats_ws, "pcl_edgeconv_gather_f32: null pointer");
    PCL_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= N && C >= 1 && C <= 2048, "pcl_edgeconv_gather_f32: bad sizes B=%d N=%d k=%d C=%d", B, N, k, C);
    const size_t P = (size_t)B * N;
    const int blocks = pcl_edgeconv_stat_rows(B, N);
    hipLaunchKernelGGL(edgeconv_gather_kernel<false>, dim3(blocks), dim3(256), sizeof(double) * 2 * C, as_stream(stream), UV, (const float*)nullptr, idx, N, k, C, P,
                       ymax, ymin, jmax, jmin, stats_ws, sumU);
    return check_launch("pcl_edgeconv_gather_f32");
}

extern "C" int pcl_edgeconv_gather_hilo_f32(const float* UV, const float* UVlo, const int32_t* idx, int B, int N, int k, int C, float* ymax, float* ymin,
                                            int32_t* jmax, int32_t* jmin, double* stats_ws, float* sumU, void* stream) {
    PCL_REQUIRE(UV && UVlo && idx && ymax && ymin && jmax && jmin && stats_ws, "pcl_edgeconv_gather_hilo_f32: null pointer");
    PCL_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= N && C >= 1 && C <= 2048, "pcl_edgeconv_gather_hilo_f32: bad sizes B=%d N=%d k=%d C=%d", B, N, k, C);
    const size_t P = (size_t)B * N;
    const int blocks = pcl_edgeconv_stat_rows(B, N);
    hipLaunchKernelGGL(edgeconv_gather_kernel<true>, dim3(blocks), dim3(256), sizeof(double) * 2 * C, as_stream(stream), UV, UVlo, idx, N, k, C, P,
                       ymax, ymin, jmax, jmin, stats_ws, sumU);
    return check_launch("pcl_edgeconv_gather_hilo_f32");
}

extern "C" int pcl_knn_transpose_i32(const int32_t* idx, int B, int N, int k, int32_t* in_off, int32_t* in_src, void* stream) {
    PCL_REQUIRE(idx && in_off && in_src, "pcl_knn_transpose_i32: null pointer");
    PCL_REQUIRE(B >= 1 && N >= 1 && N <= KT_MAXN && k >= 1 && (size_t)B * N * k < (size_t)1 << 31,
                "pcl_knn_transpose_i32: bad sizes B=%d N=%d k=%d (N <= %d)", B, N, k, KT_MAXN);
    hipStream_t st = as_stream(stream);
    if (k <= 64 && N <= 8192) {
        // the sort-free form: waves x N counters in LDS (k > 64 would interleave two sources' pieces: the sorted form below)
        int nwv = 16;
        while (nwv > 1 && (size_t)(nwv + 1) * N * sizeof(int) > 128 * 1024) nwv >>= 1;
        const size_t lds = (size_t)(nwv + 1) * N * sizeof(int);
        auto kern = knn_transpose_ordered_kernel;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(PCL_EHIP, "pcl_knn_transpose_i32: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, dim3(B), dim3(1024), lds, st, idx, N, k, B, nwv, in_off, in_src);
        return check_launch("pcl_knn_transpose_i32");
    }
    hipLaunchKernelGGL(knn_transpose_kernel, dim3(B), dim3(1024), 0, st, idx, N, k, B, in_off, in_src);
    int rc = check_launch("pcl_knn_transpose_i32");
    if (rc) return rc;
    const size_t P = (size_t)B * N;
    hipLaunchKernelGGL(knn_lists_sort_kernel, dim3((int)((P + 3) / 4)), dim3(256), 0, st, in_off, P, in_src);
    return check_launch("pcl_knn_transpose_i32(sort)");
}

extern "C" int pcl_edgeconv_scatter_f32(const float* UV, const int32_t* idx, const float* gz, const int32_t* arg, const float* a,
                                        const float* k1, const float* k2, const float* mu, int B, int N, int k, int C,
                                        const int32_t* in_off, const int32_t* in_src, const float* sumU, float* dUV, void* stream) {
    PCL_REQUIRE(UV && idx && gz && arg && a && k1 && k2 && mu && dUV, "pcl_edgeconv_scatter_f32: null pointer");
    PCL_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= N && C >= 1, "pcl_edgeconv_scatter_f32: bad sizes B=%d N=%d k=%d C=%d", B, N, k, C);
    PCL_REQUIRE((in_off == nullptr) == (in_src == nullptr) && (in_off == nullptr) == (sumU == nullptr),
                "pcl_edgeconv_scatter_f32: in_off, in_src and sumU go together");
    const size_t P = (size_t)B * N;
    hipStream_t st = as_stream(stream);
    int CH = 16;                                      // channels per workgroup of the LDS path: N*CH floats <= 64 KiB
    while (CH > 1 && (size_t)N * CH * sizeof(float) > 65536) CH >>= 1;
    const bool lds = in_off && (size_t)N * CH * sizeof(float) <= 65536;
    if (!lds) {
        hipError_t e = hipMemsetAsync(dUV, 0, sizeof(float) * P * 2 * C, st);
        if (e != hipSuccess) return fail(PCL_EHIP, "pcl_edgeconv_scatter_f32: memset: %s", hipGetErrorString(e));
    }
    if (!in_off) {
        hipLaunchKernelGGL(edgeconv_scatter_kernel, dim3((int)((P + EC_PB - 1) / EC_PB)), dim3(256), 0, st, UV, idx, gz, arg, a, k1, k2, mu,
                           N, k, C, P, dUV);
        return check_launch("pcl_edgeconv_scatter_f32");
    }
    if (lds) {
        hipLaunchKernelGGL(edgeconv_hits_lds_kernel, dim3(B, (C + CH - 1) / CH), dim3(1024), sizeof(float) * N * CH, st, UV, idx, gz, arg, sumU,
                           a, k1, k2, mu, N, k, C, CH, dUV);
    } else {
        size_t blocks = (P * C + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(edgeconv_hits_kernel, dim3((int)blocks), dim3(256), 0, st, UV, idx, gz, arg, sumU, a, k1, k2, mu, N, k, C, P, dUV);
    }
    int rc = check_launch("pcl_edgeconv_scatter_f32(hits)");
    if (rc) return rc;
    hipLaunchKernelGGL(edgeconv_insum_kernel, dim3((int)((P + 3) / 4)), dim3(256), 0, st, UV, in_off, in_src, k1, k2, mu, N, C, P, dUV);
    return check_launch("pcl_edgeconv_scatter_f32(insum)");
}
