// compact.hip -- duplicate-compacted ("ragged") grouping for ball-query groups, gfx950.
//
// query_ball_point pads every group to nsample slots with copies of its FIRST hit
// (/root/reference/misc/ops.py:321-324): at BASELINE config 2 the SA1 groups hold on average 30 distinct points
// in 64 slots.  Identical rows stay identical through conv/BN/ReLU, so the per-group MLP
// only needs the DISTINCT rows plus each row's multiplicity w (first hit: nsample - cnt + 1, others 1):
//   BatchNorm batch sums = sum_rows w*y, w*y^2 (identical to summing the padded rows), the max over the group is
//   unchanged, and in backward the dense BatchNorm term of a row counts w times.
// This file builds the compacted rows [P_eff, D] in (group, slot) order with their metadata, the ragged max-pool
// and the scatter-add of the input gradient.  P_eff stays on the device (no host sync); kernels take a capacity.
#include "common.h"

namespace pcl {

// goff[g] = sum_{h<g} max(cnt[h],1); goff[G] = P_eff.  One workgroup, 1024 lanes, sequential chunks + LDS scan.
struct GoffMulti { const int32_t* cnt[4]; int32_t* goff[4]; };
__device__ __forceinline__ void group_offsets_body(const int32_t* __restrict__ cnt, int G, int32_t* __restrict__ goff);
__global__ __launch_bounds__(1024) void group_offsets_kernel(const int32_t* __restrict__ cnt, int G, int32_t* __restrict__ goff) {
    group_offsets_body(cnt, G, goff);
}
// the same scan for up to four count arrays of one size (the scales of a multi-scale level): workgroup i takes array i
__global__ __launch_bounds__(1024) void group_offsets_multi_kernel(const GoffMulti a, int G) {
    group_offsets_body(a.cnt[blockIdx.x], G, a.goff[blockIdx.x]);
}
__device__ __forceinline__ void group_offsets_body(const int32_t* __restrict__ cnt, int G, int32_t* __restrict__ goff) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (G + 1023) / 1024;
    const int lo = min(t * per, G), hi = min(lo + per, G);
    int s = 0;
    for (int g = lo; g < hi; ++g) s += max(cnt[g], 1);
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;                       // exclusive prefix of this thread's chunk
    for (int g = lo; g < hi; ++g) { goff[g] = run; run += max(cnt[g], 1); }
    if (t == 1023) goff[G] = part[1023];
}

// One workgroup per group, wave w takes the slots [w*chunk, (w+1)*chunk) of the group's c distinct ones.
// rows off..off+c-1 = concat(xyz[idx]-new_xyz, feat[idx]).  Wide rows (C >= 32): a slot's feature row is copied by the
// whole wave (coalesced 256 B pieces), four slots in flight; narrow rows: one lane per slot.
__global__ __launch_bounds__(256) void group_compact_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                            const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ cnt, const int32_t* __restrict__ goff,
                                                            int G, int N, int m, int ns, int C, int use_xyz, int S,
                                                            float* __restrict__ rows, int2* __restrict__ rmeta,
                                                            int32_t* __restrict__ rsrc) {
    const int g = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D = (use_xyz ? 3 : 0) + C, off3 = use_xyz ? 3 : 0;
    const int c = max(cnt[g], 1), base = goff[g];
    const int b = g / m;
    const int32_t* I = idx + (size_t)g * ns;
    const int chunk = (ns + 3) >> 2;
    const int s_lo = wave * chunk, s_hi = min(s_lo + chunk, c);
    if (s_lo >= s_hi) return;
    float* o = rows + (size_t)base * S;               // row stride S >= D; columns D..S-1 are zero
    const float* fb = feat ? feat + (size_t)b * N * C : nullptr;
    if (C >= 32) {
        float q[3] = {0.f, 0.f, 0.f};
        if (use_xyz && lane < 3) q[0] = new_xyz[(size_t)g * 3 + lane];
        for (int s = s_lo; s < s_hi; s += 4) {
            int k[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) k[j] = I[min(s + j, s_hi - 1)];
            for (int c0 = 0; c0 < C; c0 += 128) {
                float v[4][2];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ch = min(c0 + h * 64 + lane, C - 1);
                        v[j][h] = fb[(size_t)k[j] * C + ch];
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ch = c0 + h * 64 + lane;
                        if (s + j < s_hi && ch < C) o[(size_t)(s + j) * S + off3 + ch] = v[j][h];
                    }
            }
            if (use_xyz && lane < 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (s + j < s_hi) o[(size_t)(s + j) * S + lane] = __fsub_rn(xyz[((size_t)b * N + k[j]) * 3 + lane], q[0]);
            }
            if (lane < S - D) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (s + j < s_hi) o[(size_t)(s + j) * S + D + lane] = 0.f;
            }
        }
    } else {
        for (int s = s_lo + lane; s < s_hi; s += 64) {
            const int k = I[s];
            float* oo = o + (size_t)s * S;
            if (use_xyz) {
#pragma unroll
                for (int d = 0; d < 3; ++d) oo[d] = __fsub_rn(xyz[((size_t)b * N + k) * 3 + d], new_xyz[(size_t)g * 3 + d]);
            }
            for (int d = 0; d < C; ++d) oo[off3 + d] = fb[(size_t)k * C + d];
            for (int d = D; d < S; ++d) oo[d] = 0.f;
        }
    }
    for (int s = s_lo + lane; s < s_hi; s += 64) {
        const int mult = s == 0 ? ns - c + 1 : 1;
        rmeta[base + s] = make_int2(g, s | (mult << 16));
        rsrc[base + s] = b * N + I[s];
    }
}

// out[g,c] = max over the group's rows of lrelu(scale*y+shift); arg = compact row-in-group; ymax = y there.
// A lane owns V consecutive channels of one group; four rows are in flight per lane.
template <int V>
__global__ __launch_bounds__(256) void bn_act_max_rows_kernel(const float* __restrict__ Y, const int32_t* __restrict__ goff,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float slope, int G, int C, float* __restrict__ out,
                                                              int32_t* __restrict__ arg, float* __restrict__ ymax) {
    const int CV = C / V;
    const size_t total = (size_t)G * CV;
    // (round 6, measured and dropped: the groups in DESCENDING order, so that the rows the GEMM in front of this kernel wrote last -- the ones
    //  most likely still in the 256 MB Infinity Cache -- are read first: 55.6 / 42.7 us ascending, 56.3 / 43.3 descending, gpurun_out/r06g)
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t g = e / CV;
        const int c = (int)(e - g * CV) * V;
        float a[V], bsh[V], best[V], by[V];
        int bi[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { a[v] = scale[c + v]; bsh[v] = shift[c + v]; best[v] = -INFINITY; by[v] = 0.f; bi[v] = 0; }
        const int r0 = goff[g], r1 = goff[g + 1];
        for (int r = r0; r < r1; r += 4) {
            float yy[4][V];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* src = Y + (size_t)min(r + j, r1 - 1) * C + c;
                if constexpr (V == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src);
                    yy[j][0] = t.x; yy[j][1] = t.y; yy[j][2] = t.z; yy[j][3] = t.w;
                } else {
                    yy[j][0] = *src;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (r + j < r1) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float u = fmaf(a[v], yy[j][v], bsh[v]);
                        const float z = u > 0.f ? u : u * slope;
                        if (z > best[v]) { best[v] = z; bi[v] = r + j - r0; by[v] = yy[j][v]; }
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) { out[g * C + c + v] = best[v]; arg[g * C + c + v] = bi[v]; ymax[g * C + c + v] = by[v]; }
    }
}

// gfeat[rsrc[r], c] += grows[r, off+c] for r < *n_rows.  One wave per row (two rows in flight), lanes over channels.
__global__ __launch_bounds__(256) void scatter_rows_add_kernel(const float* __restrict__ grows, const int32_t* __restrict__ rsrc,
                                                               const int32_t* __restrict__ n_rows, int D, int off, int C,
                                                               float* __restrict__ gfeat) {
    const int n = *n_rows, lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int r = wv * 2; r < n; r += nw * 2) {
        const int r2 = min(r + 1, n - 1);
        const size_t d0 = (size_t)rsrc[r] * C, d1 = (size_t)rsrc[r2] * C;
        for (int c = lane; c < C; c += 64) {
            const float v0 = grows[(size_t)r * D + off + c], v1 = grows[(size_t)r2 * D + off + c];
            unsafeAtomicAdd(&gfeat[d0 + c], v0);
            if (r + 1 < n) unsafeAtomicAdd(&gfeat[d1 + c], v1);
        }
    }
}

// ---- first MLP layer folded into the grouping ------------------------------------------------------------------
// The first 1x1 conv of a set-abstraction MLP is linear in the grouped row [xyz_nbr - centre | feat_nbr]:
//     y[g,s] = Wx (xyz[nbr] - centre[g]) + Wf feat[nbr] = Wx (xyz[nbr] - centre[g]) + Uf[nbr],   Uf = feat Wf^T,
// and Uf is one GEMM over the N points of a cloud instead of over its m*ns grouped rows (16x fewer rows at SA2).  The
// xyz part (3 FMAs per output) is evaluated here on the exact local coordinates, as the reference forms them.
// This kernel writes the layer's pre-BatchNorm output for the DISTINCT rows of every group (see the header of this
// file), the row metadata, and the multiplicity-weighted BatchNorm batch sums as fp64 partial rows [gridDim.x][2][C1].
// Persistent: workgroup b walks groups b, b+gridDim.x, ...; wave w takes the slots [w*chunk, (w+1)*chunk) of a group.
constexpr int GL_MAXH = 4;       // channels per lane: C1 <= 256
constexpr int GL_RPI = 4;        // rows in flight per wave
constexpr int GL_CF = 4;         // features folded inline (e.g. the 3 normals of the first level), wider ones come as Uf

// side job of the metadata kernel: the feature columns of the folded layer's weight as a dense matrix for the point GEMM
struct DenseJob { const float* src; float* dst; int rows, cols, ldw, first, blocks; };

// metadata of the distinct rows: one wave per group
// Also per row: rloc = (xyz[nbr] - centre, multiplicity) and rfeat = the (<= 4) inline feature columns, so that the
// streaming kernels below read one or two 16-byte records per row instead of a dozen scalars.
__global__ __launch_bounds__(256) void group_rows_meta_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                              const float* __restrict__ fs, int CF,
                                                              const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt,
                                                              const int32_t* __restrict__ goff, int G, int N, int m, int ns,
                                                              int2* __restrict__ rmeta, int32_t* __restrict__ rsrc,
                                                              float4* __restrict__ rloc, float4* __restrict__ rfeat, DenseJob dj) {
    if ((int)blockIdx.x >= dj.first) {       // side job (stack.hip): columns [off, off + cols) of W[rows][ldw] -> dense [rows][cols]
        const int n = dj.rows * dj.cols;
        for (int i = ((int)blockIdx.x - dj.first) * 256 + (int)threadIdx.x; i < n; i += dj.blocks * 256) {
            const int r = i / dj.cols;
            dj.dst[i] = dj.src[(size_t)r * dj.ldw + (i - r * dj.cols)];
        }
        return;
    }
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    const int c = max(cnt[g], 1), base = goff[g], b = g / m;
    const int32_t* I = idx + (size_t)g * ns;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    if (xyz) { q0 = new_xyz[(size_t)g * 3]; q1 = new_xyz[(size_t)g * 3 + 1]; q2 = new_xyz[(size_t)g * 3 + 2]; }
    for (int s = lane; s < c; s += 64) {
        const int mult = s == 0 ? ns - c + 1 : 1;
        const int src = b * N + I[s];
        rmeta[base + s] = make_int2(g, s | (mult << 16));
        rsrc[base + s] = src;
        float4 L = make_float4(0.f, 0.f, 0.f, (float)mult);
        if (xyz) {
            const float* pk = xyz + (size_t)src * 3;
            L.x = __fsub_rn(pk[0], q0); L.y = __fsub_rn(pk[1], q1); L.z = __fsub_rn(pk[2], q2);
        }
        rloc[base + s] = L;
        if (rfeat) {
            float f[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < CF; ++j) f[j] = fs[(size_t)src * CF + j];
            rfeat[base + s] = make_float4(f[0], f[1], f[2], f[3]);
        }
    }
}

// Row-parallel: wave w walks rows 4w.., 4w + 4*nwaves.., four rows in flight, lanes over the C1 channels; the next
// iteration's row metadata is requested before the current rows are processed (two dependent gathers per row otherwise).
template <int NH>
__global__ __launch_bounds__(256) void group_linear_kernel(const float* __restrict__ Uf, const float* __restrict__ Wx,
                                                           const float* __restrict__ Wfs, int CF, int ldw,
                                                           const float4* __restrict__ rloc, const float4* __restrict__ rfeat,
                                                           const int32_t* __restrict__ rsrc,
                                                           const int32_t* __restrict__ n_rows, int C1, float* __restrict__ Y,
                                                           double* __restrict__ stats) {
    __shared__ double red[2][4][64 * NH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int R = *n_rows;
    float wx[NH][3], wf[NH][GL_CF];          // coordinate weights and the weights of up to GL_CF inline features
    double ss[NH], qq[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int ch = lane + 64 * h;
#pragma unroll
        for (int d = 0; d < 3; ++d) wx[h][d] = (Wx && ch < C1) ? Wx[(size_t)ch * ldw + d] : 0.f;
#pragma unroll
        for (int f = 0; f < GL_CF; ++f) wf[h][f] = (f < CF && ch < C1) ? Wfs[(size_t)ch * ldw + f] : 0.f;
        ss[h] = 0.0; qq[h] = 0.0;
    }
    const int stride = gridDim.x * 4 * GL_RPI;
    int r0 = (blockIdx.x * 4 + wave) * GL_RPI;
    float4 nl[GL_RPI], nf[GL_RPI]; int sr[GL_RPI];
#pragma unroll
    for (int j = 0; j < GL_RPI; ++j) {
        const int r = min(r0 + j, R - 1);
        nl[j] = rloc[r]; sr[j] = Uf ? rsrc[r] : 0; nf[j] = CF ? rfeat[r] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (; r0 < R; r0 += stride) {
        float4 L[GL_RPI], F[GL_RPI]; int cs[GL_RPI];
#pragma unroll
        for (int j = 0; j < GL_RPI; ++j) { L[j] = nl[j]; F[j] = nf[j]; cs[j] = sr[j]; }
        {
            const int rn = r0 + stride;
#pragma unroll
            for (int j = 0; j < GL_RPI; ++j) {
                const int r = min(rn + j, R - 1);
                nl[j] = rloc[r]; sr[j] = Uf ? rsrc[r] : 0; nf[j] = CF ? rfeat[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int ch = min(lane + 64 * h, C1 - 1);
            float u[GL_RPI];
#pragma unroll
            for (int j = 0; j < GL_RPI; ++j) u[j] = Uf ? Uf[(size_t)cs[j] * C1 + ch] : 0.f;
#pragma unroll
            for (int j = 0; j < GL_RPI; ++j) {
                if (r0 + j < R && lane + 64 * h < C1) {
                    float y = fmaf(wx[h][2], L[j].z, fmaf(wx[h][1], L[j].y, fmaf(wx[h][0], L[j].x, u[j])));
                    y = fmaf(wf[h][3], F[j].w, fmaf(wf[h][2], F[j].z, fmaf(wf[h][1], F[j].y, fmaf(wf[h][0], F[j].x, y))));
                    Y[(size_t)(r0 + j) * C1 + ch] = y;
                    const double w = (double)L[j].w;
                    ss[h] += w * (double)y; qq[h] += w * (double)y * (double)y;
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) { red[0][wave][lane + 64 * h] = ss[h]; red[1][wave][lane + 64 * h] = qq[h]; }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C1; ch += 256) {
        double* dst = stats + (size_t)blockIdx.x * 2 * C1;
        dst[ch] = (red[0][0][ch] + red[0][1][ch]) + (red[0][2][ch] + red[0][3][ch]);
        dst[C1 + ch] = (red[1][0][ch] + red[1][1][ch]) + (red[1][2][ch] + red[1][3][ch]);
    }
}

// backward: dy = a*du - w*(k1 + k2*(y - mu)) per distinct row; dUf[point] += dy (atomics), dWx partial sums per workgroup.
template <int NH>
__global__ __launch_bounds__(256) void group_linear_bwd_kernel(const float4* __restrict__ rloc, const float4* __restrict__ rfeat,
                                                               int CF, const float* __restrict__ dU, const float* __restrict__ Y,
                                                               const float* __restrict__ a_, const float* __restrict__ k1_,
                                                               const float* __restrict__ k2_, const float* __restrict__ mu_,
                                                               const int32_t* __restrict__ rsrc,
                                                               const int32_t* __restrict__ n_rows, int C1, float* __restrict__ dUf,
                                                               float* __restrict__ dWx_part, float* __restrict__ dWf_part) {
    __shared__ float red[3 + GL_CF][4][64 * NH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int R = *n_rows;
    float a[NH], k1[NH], k2[NH], mu[NH], gw[NH][3 + GL_CF];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int ch = min(lane + 64 * h, C1 - 1);
        a[h] = a_[ch]; k1[h] = k1_[ch]; k2[h] = k2_[ch]; mu[h] = mu_[ch];
#pragma unroll
        for (int d = 0; d < 3 + GL_CF; ++d) gw[h][d] = 0.f;
    }
    const int stride = gridDim.x * 4 * GL_RPI;
    int r0 = (blockIdx.x * 4 + wave) * GL_RPI;
    float4 nl[GL_RPI], nf[GL_RPI]; int sr[GL_RPI];
#pragma unroll
    for (int j = 0; j < GL_RPI; ++j) {
        const int r = min(r0 + j, R - 1);
        nl[j] = rloc[r]; sr[j] = dUf ? rsrc[r] : 0; nf[j] = dWf_part ? rfeat[r] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (; r0 < R; r0 += stride) {
        float4 L[GL_RPI], F[GL_RPI]; int cs[GL_RPI];
#pragma unroll
        for (int j = 0; j < GL_RPI; ++j) { L[j] = nl[j]; F[j] = nf[j]; cs[j] = sr[j]; }
        {
            const int rn = r0 + stride;
#pragma unroll
            for (int j = 0; j < GL_RPI; ++j) {
                const int r = min(rn + j, R - 1);
                nl[j] = rloc[r]; sr[j] = dUf ? rsrc[r] : 0; nf[j] = dWf_part ? rfeat[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int ch = min(lane + 64 * h, C1 - 1);
            float du[GL_RPI], y[GL_RPI];
#pragma unroll
            for (int j = 0; j < GL_RPI; ++j) {
                const size_t o = (size_t)min(r0 + j, R - 1) * C1 + ch;
                du[j] = dU[o]; y[j] = Y[o];
            }
#pragma unroll
            for (int j = 0; j < GL_RPI; ++j) {
                if (r0 + j < R && lane + 64 * h < C1) {
                    const float dy = fmaf(a[h], du[j], -L[j].w * fmaf(k2[h], y[j] - mu[h], k1[h]));
                    if (dUf) unsafeAtomicAdd(&dUf[(size_t)cs[j] * C1 + ch], dy);
                    gw[h][0] = fmaf(dy, L[j].x, gw[h][0]); gw[h][1] = fmaf(dy, L[j].y, gw[h][1]);
                    gw[h][2] = fmaf(dy, L[j].z, gw[h][2]);
                    gw[h][3] = fmaf(dy, F[j].x, gw[h][3]); gw[h][4] = fmaf(dy, F[j].y, gw[h][4]);
                    gw[h][5] = fmaf(dy, F[j].z, gw[h][5]); gw[h][6] = fmaf(dy, F[j].w, gw[h][6]);
                }
            }
        }
    }
    if (dWx_part || dWf_part) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int d = 0; d < 3 + GL_CF; ++d) red[d][wave][lane + 64 * h] = gw[h][d];
        __syncthreads();
        if (dWx_part)
            for (int e = threadIdx.x; e < C1 * 3; e += 256) {
                const int ch = e / 3, d = e - ch * 3;
                dWx_part[(size_t)blockIdx.x * C1 * 3 + e] = (red[d][0][ch] + red[d][1][ch]) + (red[d][2][ch] + red[d][3][ch]);
            }
        if (dWf_part)
            for (int e = threadIdx.x; e < C1 * CF; e += 256) {
                const int ch = e / CF, d = 3 + (e - ch * CF);
                dWf_part[(size_t)blockIdx.x * C1 * CF + e] = (red[d][0][ch] + red[d][1][ch]) + (red[d][2][ch] + red[d][3][ch]);
            }
    }
}

// ---- 16-byte-channel variants of the two kernels above (C1 = 64 / 128 / 256) -----------------------------------------
// A lane owns 4 consecutive channels, LPR = C1/4 lanes cover a row, so one wave instruction moves 64/LPR consecutive
// rows = 1 KiB of contiguous Y (or dU) instead of 256 B: a quarter of the memory instructions for the same bytes.
template <int LPR>     // lanes per row: 16, 32 or 64
__global__ __launch_bounds__(256) void group_linear_v4_kernel(const float* __restrict__ Uf, const float* __restrict__ Wx,
                                                              const float* __restrict__ Wfs, int CF, int ldw,
                                                              const float4* __restrict__ rloc, const float4* __restrict__ rfeat,
                                                              const int32_t* __restrict__ rsrc,
                                                              const int32_t* __restrict__ n_rows, float* __restrict__ Y,
                                                              double* __restrict__ stats) {
    constexpr int C1 = 4 * LPR, RPW = 64 / LPR, U = 4;          // rows per wave instruction; instructions in flight
    __shared__ double red[2][4][C1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cl = lane % LPR, rsub = lane / LPR, c4 = 4 * cl;
    const int R = *n_rows;
    float wx[3][4], wf[GL_CF][4];
    double ss[4], qq[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int d = 0; d < 3; ++d) wx[d][v] = Wx ? Wx[(size_t)(c4 + v) * ldw + d] : 0.f;
#pragma unroll
        for (int f = 0; f < GL_CF; ++f) wf[f][v] = f < CF ? Wfs[(size_t)(c4 + v) * ldw + f] : 0.f;
        ss[v] = 0.0; qq[v] = 0.0;
    }
    const int stride = gridDim.x * 4 * RPW * U;
    for (int r0 = (blockIdx.x * 4 + wave) * RPW * U; r0 < R; r0 += stride) {
        float4 L[U], F[U], u[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int r = min(r0 + j * RPW + rsub, R - 1);
            L[j] = rloc[r];
            F[j] = CF ? rfeat[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            u[j] = Uf ? *reinterpret_cast<const float4*>(Uf + (size_t)rsrc[r] * C1 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int r = r0 + j * RPW + rsub;
            if (r < R) {
                float y[4] = {u[j].x, u[j].y, u[j].z, u[j].w};
                const double w = (double)L[j].w;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float t = fmaf(wx[2][v], L[j].z, fmaf(wx[1][v], L[j].y, fmaf(wx[0][v], L[j].x, y[v])));
                    t = fmaf(wf[3][v], F[j].w, fmaf(wf[2][v], F[j].z, fmaf(wf[1][v], F[j].y, fmaf(wf[0][v], F[j].x, t))));
                    y[v] = t;
                    ss[v] += w * (double)t; qq[v] += w * (double)t * (double)t;
                }
                *reinterpret_cast<float4*>(Y + (size_t)r * C1 + c4) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
    }
    // fold the RPW row-lanes of a wave, then the four waves
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) { ss[v] += __shfl_xor(ss[v], off); qq[v] += __shfl_xor(qq[v], off); }
        if (rsub == 0) { red[0][wave][c4 + v] = ss[v]; red[1][wave][c4 + v] = qq[v]; }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C1; ch += 256) {
        double* dst = stats + (size_t)blockIdx.x * 2 * C1;
        dst[ch] = (red[0][0][ch] + red[0][1][ch]) + (red[0][2][ch] + red[0][3][ch]);
        dst[C1 + ch] = (red[1][0][ch] + red[1][1][ch]) + (red[1][2][ch] + red[1][3][ch]);
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void group_linear_bwd_v4_kernel(const float4* __restrict__ rloc, const float4* __restrict__ rfeat,
                                                                  int CF, const float* __restrict__ dU, const float* __restrict__ Y,
                                                                  const float* __restrict__ a_, const float* __restrict__ k1_,
                                                                  const float* __restrict__ k2_, const float* __restrict__ mu_,
                                                                  const int32_t* __restrict__ rsrc,
                                                                  const int32_t* __restrict__ n_rows, float* __restrict__ dUf,
                                                                  float* __restrict__ dWx_part, float* __restrict__ dWf_part) {
    constexpr int C1 = 4 * LPR, RPW = 64 / LPR, U = 4;
    __shared__ float red[3 + GL_CF][4][C1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cl = lane % LPR, rsub = lane / LPR, c4 = 4 * cl;
    const int R = *n_rows;
    float a[4], k1[4], k2[4], mu[4], gw[3 + GL_CF][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        a[v] = a_[c4 + v]; k1[v] = k1_[c4 + v]; k2[v] = k2_[c4 + v]; mu[v] = mu_[c4 + v];
#pragma unroll
        for (int d = 0; d < 3 + GL_CF; ++d) gw[d][v] = 0.f;
    }
    const int stride = gridDim.x * 4 * RPW * U;
    for (int r0 = (blockIdx.x * 4 + wave) * RPW * U; r0 < R; r0 += stride) {
        float4 L[U], F[U], du[U], yv[U];
        int src[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int r = min(r0 + j * RPW + rsub, R - 1);
            L[j] = rloc[r];
            F[j] = dWf_part ? rfeat[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            src[j] = dUf ? rsrc[r] : 0;
            du[j] = *reinterpret_cast<const float4*>(dU + (size_t)r * C1 + c4);
            yv[j] = *reinterpret_cast<const float4*>(Y + (size_t)r * C1 + c4);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int r = r0 + j * RPW + rsub;
            if (r < R) {
                const float dd[4] = {du[j].x, du[j].y, du[j].z, du[j].w}, yy[4] = {yv[j].x, yv[j].y, yv[j].z, yv[j].w};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float dy = fmaf(a[v], dd[v], -L[j].w * fmaf(k2[v], yy[v] - mu[v], k1[v]));
                    if (dUf) unsafeAtomicAdd(&dUf[(size_t)src[j] * C1 + c4 + v], dy);
                    gw[0][v] = fmaf(dy, L[j].x, gw[0][v]); gw[1][v] = fmaf(dy, L[j].y, gw[1][v]); gw[2][v] = fmaf(dy, L[j].z, gw[2][v]);
                    gw[3][v] = fmaf(dy, F[j].x, gw[3][v]); gw[4][v] = fmaf(dy, F[j].y, gw[4][v]);
                    gw[5][v] = fmaf(dy, F[j].z, gw[5][v]); gw[6][v] = fmaf(dy, F[j].w, gw[6][v]);
                }
            }
        }
    }
    if (dWx_part || dWf_part) {
#pragma unroll
        for (int d = 0; d < 3 + GL_CF; ++d)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float t = gw[d][v];
#pragma unroll
                for (int off = LPR; off < 64; off <<= 1) t += __shfl_xor(t, off);
                if (rsub == 0) red[d][wave][c4 + v] = t;
            }
        __syncthreads();
        if (dWx_part)
            for (int e = threadIdx.x; e < C1 * 3; e += 256) {
                const int ch = e / 3, d = e - ch * 3;
                dWx_part[(size_t)blockIdx.x * C1 * 3 + e] = (red[d][0][ch] + red[d][1][ch]) + (red[d][2][ch] + red[d][3][ch]);
            }
        if (dWf_part)
            for (int e = threadIdx.x; e < C1 * CF; e += 256) {
                const int ch = e / CF, d = 3 + (e - ch * CF);
                dWf_part[(size_t)blockIdx.x * C1 * CF + e] = (red[d][0][ch] + red[d][1][ch]) + (red[d][2][ch] + red[d][3][ch]);
            }
    }
}

// ---- the scatter to the points as a gather over the points' row lists (round 5) ---------------------------------------------------------
// With wide point features every row's dy [C1] went to dUf[source point] by one fp32 atomic per element: 22.5 M of them at the second
// level of PointNet++, 52 of the kernel's 82 us (tools/dbg/glinbwd_atomics.py; the reads alone run at 6 TB/s), in an order the hardware
// picks.  Here the rows are walked BY SOURCE POINT: in_off [B*N + 1] / in_rows [rows] list each point's rows in ascending order
// (rows_transpose_kernel, built once per forward from row_src), a wave per point, a lane owns four channels: every row is still read
// exactly once (512-byte rows), the sums run in a fixed order -- run-to-run identical -- and dUf is written once, not zero-filled and
// atomically updated.  dWx partial sums as before (per workgroup, fixed order).
template <int LPR>
__global__ __launch_bounds__(256) void group_linear_bwd_gather_kernel(const float4* __restrict__ rloc, const float* __restrict__ dU,
                                                                      const float* __restrict__ Y, const float* __restrict__ a_,
                                                                      const float* __restrict__ k1_, const float* __restrict__ k2_,
                                                                      const float* __restrict__ mu_, const int32_t* __restrict__ in_off,
                                                                      const int32_t* __restrict__ in_rows, int P, float* __restrict__ dUf,
                                                                      float* __restrict__ dWx_part) {
    constexpr int C1 = 4 * LPR, PPW = 64 / LPR, U = 4;
    __shared__ float red[3][4][C1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cl = lane % LPR, psub = lane / LPR, c4 = 4 * cl;
    const float4 a = *reinterpret_cast<const float4*>(a_ + c4), k1 = *reinterpret_cast<const float4*>(k1_ + c4);
    const float4 k2 = *reinterpret_cast<const float4*>(k2_ + c4), mu = *reinterpret_cast<const float4*>(mu_ + c4);
    float gw[3][4];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int v = 0; v < 4; ++v) gw[d][v] = 0.f;
    // a wave takes ONE point at a time and its 64 / LPR row-lanes alternate over the point's list in chunks of U rows: the lists of popular
    // points (the centre of a cloud is in every group: 100+ rows) are the kernel's tail, and this halves / quarters their dependent chain;
    // the row-lanes' sums meet by shuffles in a fixed order.  The next chunk's row indices are requested before this chunk's rows are used.
    const int stride = gridDim.x * 4;
    // (a point in no group has o0 == o1, and for the trailing points of the last cloud that is in_off[P] = the number of rows: the
    // prefetch index is clamped to the last row so that the -- unused -- value is never read past a full `in_rows`)
    const int last = max(in_off[P] - 1, 0);
    for (int p = blockIdx.x * 4 + wave; p < P; p += stride) {
        const int o0 = in_off[p], o1 = in_off[p + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int o = o0 + psub * U;
        int r[U];
#pragma unroll
        for (int j = 0; j < U; ++j) r[j] = in_rows[min(max(o0, min(o + j, o1 - 1)), last)];
        for (; o < o1; o += PPW * U) {
            float4 L[U], du[U], yv[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                L[j] = rloc[r[j]];
                du[j] = *reinterpret_cast<const float4*>(dU + (size_t)r[j] * C1 + c4);
                yv[j] = *reinterpret_cast<const float4*>(Y + (size_t)r[j] * C1 + c4);
            }
            int rn[U];
#pragma unroll
            for (int j = 0; j < U; ++j) rn[j] = in_rows[min(max(o0, min(o + PPW * U + j, o1 - 1)), last)];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (o + j < o1) {
                    const float w = L[j].w;
                    const float d0 = fmaf(a.x, du[j].x, -w * fmaf(k2.x, yv[j].x - mu.x, k1.x)), d1 = fmaf(a.y, du[j].y, -w * fmaf(k2.y, yv[j].y - mu.y, k1.y));
                    const float d2 = fmaf(a.z, du[j].z, -w * fmaf(k2.z, yv[j].z - mu.z, k1.z)), d3 = fmaf(a.w, du[j].w, -w * fmaf(k2.w, yv[j].w - mu.w, k1.w));
                    acc.x += d0; acc.y += d1; acc.z += d2; acc.w += d3;
                    gw[0][0] = fmaf(d0, L[j].x, gw[0][0]); gw[0][1] = fmaf(d1, L[j].x, gw[0][1]); gw[0][2] = fmaf(d2, L[j].x, gw[0][2]); gw[0][3] = fmaf(d3, L[j].x, gw[0][3]);
                    gw[1][0] = fmaf(d0, L[j].y, gw[1][0]); gw[1][1] = fmaf(d1, L[j].y, gw[1][1]); gw[1][2] = fmaf(d2, L[j].y, gw[1][2]); gw[1][3] = fmaf(d3, L[j].y, gw[1][3]);
                    gw[2][0] = fmaf(d0, L[j].z, gw[2][0]); gw[2][1] = fmaf(d1, L[j].z, gw[2][1]); gw[2][2] = fmaf(d2, L[j].z, gw[2][2]); gw[2][3] = fmaf(d3, L[j].z, gw[2][3]);
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) r[j] = rn[j];
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
        }
        if (psub == 0) *reinterpret_cast<float4*>(dUf + (size_t)p * C1 + c4) = acc;
    }
    if (dWx_part) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float t = gw[d][v];
#pragma unroll
                for (int off = LPR; off < 64; off <<= 1) t += __shfl_xor(t, off);
                if (psub == 0) red[d][wave][c4 + v] = t;
            }
        __syncthreads();
        for (int e = threadIdx.x; e < C1 * 3; e += 256) {
            const int ch = e / 3, d = e - ch * 3;
            dWx_part[(size_t)blockIdx.x * C1 * 3 + e] = (red[d][0][ch] + red[d][1][ch]) + (red[d][2][ch] + red[d][3][ch]);
        }
    }
}

// in_off / in_rows from row_src: one workgroup per cloud (a cloud's rows are contiguous: groups are ordered by cloud).  A point's rows
// ascend with the group index -- a group holds a point at most once (ball query / k-NN return distinct indices) -- so the lists come out
// sorted if the groups are filled in order: each of the 16 waves owns a contiguous range of the cloud's groups, counts its rows per point
// (cnt[w][p]), the counts become start positions (a scan over the points, then over the waves), and every wave files its groups ONE AFTER
// THE OTHER (a wave's LDS operations execute in order; no sort: popular points have lists of 100+ rows and a per-point insertion sort took
// 53 us inside a training step).  Should a group name a point twice, both rows are filed (LDS atomics) in an unspecified mutual order.
constexpr int RT_T = 1024, RT_W = RT_T / 64;
__global__ __launch_bounds__(RT_T) void rows_transpose_kernel(const int32_t* __restrict__ rsrc, const int32_t* __restrict__ group_off, int m, int N,
                                                              int B, int32_t* __restrict__ in_off, int32_t* __restrict__ in_rows) {
    extern __shared__ int rt_lds[];                        // cnt [RT_W][N] | off [N + 1] | lst [rows of the cloud]
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int* cnt = rt_lds; int* off = rt_lds + RT_W * N; int* lst = off + N + 1;
    const int g_first = b * m, pbase = b * N;
    const int rb = group_off[g_first], re = group_off[g_first + m];
    const int gpw = (m + RT_W - 1) / RT_W;
    const int g0 = min(m, wave * gpw), g1 = min(m, g0 + gpw);       // this wave's groups (relative to the cloud)
    for (int i = tid; i < RT_W * N; i += RT_T) cnt[i] = 0;
    __syncthreads();
    int* mycnt = cnt + wave * N;
    {
        const int w0 = group_off[g_first + g0], w1 = group_off[g_first + g1];
        for (int r = w0 + lane; r < w1; r += 64) atomicAdd(&mycnt[rsrc[r] - pbase], 1);
    }
    __syncthreads();
    // totals per point, exclusive scan over the points (thread t owns a contiguous piece), then the waves' start positions
    __shared__ int wtot[RT_W];
    const int per = (N + RT_T - 1) / RT_T, i0 = tid * per, i1 = min(N, i0 + per);
    int s = 0;
    for (int i = i0; i < i1; ++i) { int t = 0; for (int w = 0; w < RT_W; ++w) t += cnt[w * N + i]; off[i] = t; s += t; }
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int run = inc - s;
    for (int w = 0; w < wave; ++w) run += wtot[w];
    for (int i = i0; i < i1; ++i) {
        const int t = off[i];
        off[i] = run;
        int pos = run;
        for (int w = 0; w < RT_W; ++w) { const int c = cnt[w * N + i]; cnt[w * N + i] = pos; pos += c; }
        run += t;
    }
    if (tid == 0) off[N] = re - rb;
    __syncthreads();
    for (int g = g0; g < g1; ++g) {                        // one group after the other: ascending rows per point
        const int q0 = group_off[g_first + g], q1 = group_off[g_first + g + 1];
        for (int r = q0 + lane; r < q1; r += 64) lst[atomicAdd(&mycnt[rsrc[r] - pbase], 1)] = r;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int pl = tid; pl < N; pl += RT_T) in_off[pbase + pl] = rb + off[pl];
    if (b == B - 1 && tid == 0) in_off[pbase + N] = re;
    for (int i = tid; i < re - rb; i += RT_T) in_rows[rb + i] = lst[i];
}

// dW0[c, 0..2] = sum_r dWx_part[r][c][0..2];  dW0[c, off + f] = sum_r dWf_part[r][c][f]  (dW0 has leading dimension ld):
// one wave per output element group -- 64 row-lanes per element, partials are L2-resident.
__global__ __launch_bounds__(256) void group_linear_dw_kernel(const float* __restrict__ px, const float* __restrict__ pf, int rows,
                                                              int C1, int CF, int off, int ld, float* __restrict__ dW0) {
    const int nx = px ? C1 * 3 : 0, nf = pf ? C1 * CF : 0;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= nx + nf) return;
    const bool isx = e < nx;
    const float* src = isx ? px + e : pf + (e - nx);
    const int n = isx ? nx : nf;
    float s = 0.f;
    for (int r = lane; r < rows; r += 64) s += src[(size_t)r * n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        if (isx) dW0[(size_t)(e / 3) * ld + e % 3] = s;
        else dW0[(size_t)((e - nx) / CF) * ld + off + (e - nx) % CF] = s;
    }
}

}  // namespace pcl
using namespace pcl;

extern "C" int pcl_group_offsets_i32(const int32_t* cnt, int G, int32_t* group_off, void* stream) {
    PCL_REQUIRE(cnt && group_off && G >= 1, "pcl_group_offsets_i32: bad arguments");
    hipLaunchKernelGGL(group_offsets_kernel, dim3(1), dim3(1024), 0, as_stream(stream), cnt, G, group_off);
    return check_launch("pcl_group_offsets_i32");
}

extern "C" int pcl_group_offsets_multi_i32(int n, const int32_t* const* cnt, int G, int32_t* const* group_off, void* stream) {
    PCL_REQUIRE(cnt && group_off && G >= 1 && n >= 1 && n <= 4, "pcl_group_offsets_multi_i32: bad arguments (n = %d: 1..4)", n);
    GoffMulti a = {};
    for (int i = 0; i < n; ++i) {
        PCL_REQUIRE(cnt[i] && group_off[i], "pcl_group_offsets_multi_i32: null array %d", i);
        a.cnt[i] = cnt[i]; a.goff[i] = group_off[i];
    }
    hipLaunchKernelGGL(group_offsets_multi_kernel, dim3(n), dim3(1024), 0, as_stream(stream), a, G);
    return check_launch("pcl_group_offsets_multi_i32");
}

extern "C" int pcl_group_compact_f32(const float* xyz, const float* new_xyz, const float* feat, const int32_t* idx,
                                     const int32_t* cnt, const int32_t* group_off, int B, int N, int m, int ns, int C,
                                     int use_xyz, int row_stride, float* rows, int32_t* row_meta, int32_t* row_src,
                                     void* stream) {
    PCL_REQUIRE(idx && cnt && rows && row_meta && row_src && group_off, "pcl_group_compact_f32: null pointer");
    PCL_REQUIRE(!use_xyz || (xyz && new_xyz), "pcl_group_compact_f32: use_xyz needs xyz and new_xyz");
    PCL_REQUIRE(C == 0 || feat, "pcl_group_compact_f32: C=%d needs feat", C);
    PCL_REQUIRE(B >= 1 && N >= 1 && m >= 1 && ns >= 1 && ns < 32768 && C >= 0 && (use_xyz || C > 0), "pcl_group_compact_f32: bad sizes");
    const int D = (use_xyz ? 3 : 0) + C;
    PCL_REQUIRE(row_stride >= D && row_stride < D + 64, "pcl_group_compact_f32: row_stride=%d for %d columns", row_stride, D);
    const int G = B * m;
    hipLaunchKernelGGL(group_compact_kernel, dim3(G), dim3(256), 0, as_stream(stream), xyz, new_xyz, feat, idx, cnt, group_off, G, N, m,
                       ns, C, use_xyz, row_stride, rows, reinterpret_cast<int2*>(row_meta), row_src);
    return check_launch("pcl_group_compact_f32");
}

extern "C" int pcl_bn_act_max_rows_f32(const float* Y, const int32_t* group_off, const float* scale, const float* shift,
                                       float slope, int G, int C, float* out, int32_t* arg, float* ymax, void* stream) {
    PCL_REQUIRE(Y && group_off && scale && shift && out && arg && ymax && G >= 1 && C >= 1, "pcl_bn_act_max_rows_f32: bad arguments");
    const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
    const size_t total = (size_t)G * (vec ? C / 4 : C);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    if (vec)
        hipLaunchKernelGGL(bn_act_max_rows_kernel<4>, dim3(blocks), dim3(256), 0, as_stream(stream), Y, group_off, scale, shift, slope,
                           G, C, out, arg, ymax);
    else
        hipLaunchKernelGGL(bn_act_max_rows_kernel<1>, dim3(blocks), dim3(256), 0, as_stream(stream), Y, group_off, scale, shift, slope,
                           G, C, out, arg, ymax);
    return check_launch("pcl_bn_act_max_rows_f32");
}

extern "C" int pcl_scatter_rows_add_f32(const float* grows, const int32_t* row_src, const int32_t* n_rows_dev, int rows_cap,
                                        int D, int off, int C, int n_dst_rows, float* gfeat, void* stream) {
    PCL_REQUIRE(grows && row_src && n_rows_dev && gfeat && rows_cap >= 1 && D >= 1 && off >= 0 && C >= 1 && off + C <= D && n_dst_rows >= 1,
                "pcl_scatter_rows_add_f32: bad arguments");
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)n_dst_rows * C, st);
    if (e != hipSuccess) return fail(PCL_EHIP, "pcl_scatter_rows_add_f32: memset: %s", hipGetErrorString(e));
    size_t blocks = ((size_t)rows_cap + 7) / 8;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scatter_rows_add_kernel, dim3((int)blocks), dim3(256), 0, st, grows, row_src, n_rows_dev, D, off, C, gfeat);
    return check_launch("pcl_scatter_rows_add_f32");
}

constexpr int GL_BLOCKS = 1024;     // 4 workgroups of 4 waves per CU = one resident round at 4 waves/SIMD; also the number of
                                     // BatchNorm / dWx partial rows (2048: step +10 us, 512: same step, forward of the 128-wide level slower)

extern "C" int pcl_group_linear_stat_rows(int B, int m) { return (B < 1 || m < 1) ? 0 : GL_BLOCKS; }

namespace pcl {
int group_linear_fwd_impl(const float* xyz, const float* new_xyz, const float* Uf, const float* Wx, const float* feat_small,
                          const float* Wf_small, int CF, int ldw, const int32_t* idx, const int32_t* cnt, const int32_t* group_off, int B,
                          int N, int m, int ns, int C1, float* Y, int32_t* row_meta, int32_t* row_src, float* row_loc, float* row_feat,
                          double* stats_ws, void* stream, int phase, const float* dense_src, float* dense_dst, int dense_cols);
}
extern "C" int pcl_group_linear_f32(const float* xyz, const float* new_xyz, const float* Uf, const float* Wx, const float* feat_small,
                                    const float* Wf_small, int CF, int ldw, const int32_t* idx, const int32_t* cnt,
                                    const int32_t* group_off, int B, int N, int m, int ns, int C1, float* Y, int32_t* row_meta,
                                    int32_t* row_src, float* row_loc, float* row_feat, double* stats_ws, void* stream) {
    return group_linear_fwd_impl(xyz, new_xyz, Uf, Wx, feat_small, Wf_small, CF, ldw, idx, cnt, group_off, B, N, m, ns, C1, Y, row_meta, row_src,
                                 row_loc, row_feat, stats_ws, stream, 3, nullptr, nullptr, 0);
}
// phase: 1 = the row metadata only (+ the densify side job: dense_dst[C1][dense_cols] = dense_src[C1][ldw] columns, stack.hip
// runs the point GEMM that needs it between the two phases), 2 = the streaming kernel only, 3 = both
int pcl::group_linear_fwd_impl(const float* xyz, const float* new_xyz, const float* Uf, const float* Wx, const float* feat_small,
                               const float* Wf_small, int CF, int ldw, const int32_t* idx, const int32_t* cnt, const int32_t* group_off,
                               int B, int N, int m, int ns, int C1, float* Y, int32_t* row_meta, int32_t* row_src, float* row_loc,
                               float* row_feat, double* stats_ws, void* stream, int phase, const float* dense_src, float* dense_dst,
                               int dense_cols) {
    PCL_REQUIRE(idx && cnt && group_off && Y && row_meta && row_src && row_loc && stats_ws, "pcl_group_linear_f32: null pointer");
    PCL_REQUIRE(CF == 0 || row_feat, "pcl_group_linear_f32: inline features need row_feat");
    PCL_REQUIRE(phase == 1 || Uf || Wx || CF > 0, "pcl_group_linear_f32: need features (Uf or feat_small) and/or coordinates (Wx)");
    PCL_REQUIRE(CF >= 0 && CF <= GL_CF && (CF == 0 || (feat_small && Wf_small)), "pcl_group_linear_f32: CF=%d inline features (<= %d)", CF, GL_CF);
    PCL_REQUIRE(!Wx || (xyz && new_xyz), "pcl_group_linear_f32: Wx needs xyz and new_xyz");
    PCL_REQUIRE(ldw >= (Wx ? 3 : 0) && ldw >= CF, "pcl_group_linear_f32: ldw=%d", ldw);
    PCL_REQUIRE(B >= 1 && N >= 1 && m >= 1 && ns >= 1 && ns < 32768 && C1 >= 1 && C1 <= 64 * GL_MAXH,
                "pcl_group_linear_f32: bad sizes B=%d N=%d m=%d ns=%d C1=%d (C1 <= %d)", B, N, m, ns, C1, 64 * GL_MAXH);
    const int G = B * m;
    hipStream_t st = as_stream(stream);
    if (phase & 1) {
        const int gb = (G + 3) / 4;
        DenseJob dj = {dense_src, dense_dst, C1, dense_cols, ldw, gb, 0};
        if (dense_dst) dj.blocks = (C1 * dense_cols + 1023) / 1024 < 64 ? (C1 * dense_cols + 1023) / 1024 : 64;
        hipLaunchKernelGGL(group_rows_meta_kernel, dim3(gb + dj.blocks), dim3(256), 0, st, Wx ? xyz : nullptr, new_xyz, feat_small, CF, idx, cnt,
                           group_off, G, N, m, ns, reinterpret_cast<int2*>(row_meta), row_src, reinterpret_cast<float4*>(row_loc),
                           reinterpret_cast<float4*>(CF ? row_feat : nullptr), dj);
        int rc = check_launch("pcl_group_linear_f32(meta)");
        if (rc || phase == 1) return rc;
    }
    const float4* rl = reinterpret_cast<const float4*>(row_loc);
    const float4* rf = reinterpret_cast<const float4*>(row_feat);
    const dim3 grid(GL_BLOCKS), block(256);
    const bool al16 = ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Uf)) & 15) == 0;
    if (al16 && (C1 == 64 || C1 == 128 || C1 == 256)) {
        if (C1 == 64) hipLaunchKernelGGL(group_linear_v4_kernel<16>, grid, block, 0, st, Uf, Wx, Wf_small, CF, ldw, rl, rf, row_src, group_off + G, Y, stats_ws);
        else if (C1 == 128) hipLaunchKernelGGL(group_linear_v4_kernel<32>, grid, block, 0, st, Uf, Wx, Wf_small, CF, ldw, rl, rf, row_src, group_off + G, Y, stats_ws);
        else hipLaunchKernelGGL(group_linear_v4_kernel<64>, grid, block, 0, st, Uf, Wx, Wf_small, CF, ldw, rl, rf, row_src, group_off + G, Y, stats_ws);
        return check_launch("pcl_group_linear_f32");
    }
    if (C1 <= 64) hipLaunchKernelGGL(group_linear_kernel<1>, grid, block, 0, st, Uf, Wx, Wf_small, CF, ldw, rl, rf, row_src, group_off + G, C1, Y, stats_ws);
    else if (C1 <= 128) hipLaunchKernelGGL(group_linear_kernel<2>, grid, block, 0, st, Uf, Wx, Wf_small, CF, ldw, rl, rf, row_src, group_off + G, C1, Y, stats_ws);
    else hipLaunchKernelGGL(group_linear_kernel<4>, grid, block, 0, st, Uf, Wx, Wf_small, CF, ldw, rl, rf, row_src, group_off + G, C1, Y, stats_ws);
    return check_launch("pcl_group_linear_f32");
}

namespace pcl {
int group_linear_bwd_impl(const float* row_loc, const float* row_feat, int CF, const float* dU, const float* Y, const float* a,
                          const float* k1, const float* k2, const float* mu, const int32_t* row_src, const int32_t* n_rows_dev, int B, int N,
                          int C1, float* dUf, float* dWx_part, float* dWf_part, float* dW0, int ldw, int off, void* stream, bool duf_is_zero);
}
extern "C" int pcl_group_linear_bwd_f32(const float* row_loc, const float* row_feat, int CF, const float* dU, const float* Y,
                                        const float* a, const float* k1, const float* k2, const float* mu, const int32_t* row_src,
                                        const int32_t* n_rows_dev, int B, int N, int C1, float* dUf, float* dWx_part,
                                        float* dWf_part, float* dW0, int ldw, int off, void* stream) {
    return group_linear_bwd_impl(row_loc, row_feat, CF, dU, Y, a, k1, k2, mu, row_src, n_rows_dev, B, N, C1, dUf, dWx_part, dWf_part, dW0, ldw,
                                 off, stream, false);
}
// duf_is_zero: an earlier launch of the same stack call already cleared dUf (stack.hip: the max-gradient kernel's side job)
int pcl::group_linear_bwd_impl(const float* row_loc, const float* row_feat, int CF, const float* dU, const float* Y, const float* a,
                               const float* k1, const float* k2, const float* mu, const int32_t* row_src, const int32_t* n_rows_dev, int B,
                               int N, int C1, float* dUf, float* dWx_part, float* dWf_part, float* dW0, int ldw, int off, void* stream,
                               bool duf_is_zero) {
    PCL_REQUIRE(row_loc && dU && Y && a && k1 && k2 && mu && row_src && n_rows_dev, "pcl_group_linear_bwd_f32: null pointer");
    PCL_REQUIRE(dUf || dWx_part || dWf_part, "pcl_group_linear_bwd_f32: nothing to compute");
    PCL_REQUIRE(!dWf_part || (row_feat && CF >= 1 && CF <= GL_CF), "pcl_group_linear_bwd_f32: dWf needs row_feat, CF=%d", CF);
    PCL_REQUIRE(B >= 1 && N >= 1 && C1 >= 1 && C1 <= 64 * GL_MAXH, "pcl_group_linear_bwd_f32: bad sizes");
    PCL_REQUIRE(!dW0 || ((dWx_part || dWf_part) && ldw >= (dWx_part ? 3 : 0) && off >= 0 && ldw >= off + (dWf_part ? CF : 0)),
                "pcl_group_linear_bwd_f32: dW0 needs the partial buffers and ldw=%d >= off=%d + CF", ldw, off);
    hipStream_t st = as_stream(stream);
    if (dUf && !duf_is_zero) {
        hipError_t e = hipMemsetAsync(dUf, 0, sizeof(float) * (size_t)B * N * C1, st);
        if (e != hipSuccess) return fail(PCL_EHIP, "pcl_group_linear_bwd_f32: memset: %s", hipGetErrorString(e));
    }
    const float4* rl = reinterpret_cast<const float4*>(row_loc);
    const float4* rf = reinterpret_cast<const float4*>(row_feat);
    const dim3 grid(GL_BLOCKS), block(256);
    // with dUf the scatter is one atomic per element either way, and a lane owning 4 consecutive channels spreads every
    // atomic instruction over 4x the cache lines (measured 187 vs 77 us on SA2): the 16-byte layout only when there is none
    const bool al16 = !dUf && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(dU)) & 15) == 0;
    if (al16 && (C1 == 64 || C1 == 128 || C1 == 256)) {
        if (C1 == 64) hipLaunchKernelGGL(group_linear_bwd_v4_kernel<16>, grid, block, 0, st, rl, rf, CF, dU, Y, a, k1, k2, mu, row_src, n_rows_dev, dUf, dWx_part, dWf_part);
        else if (C1 == 128) hipLaunchKernelGGL(group_linear_bwd_v4_kernel<32>, grid, block, 0, st, rl, rf, CF, dU, Y, a, k1, k2, mu, row_src, n_rows_dev, dUf, dWx_part, dWf_part);
        else hipLaunchKernelGGL(group_linear_bwd_v4_kernel<64>, grid, block, 0, st, rl, rf, CF, dU, Y, a, k1, k2, mu, row_src, n_rows_dev, dUf, dWx_part, dWf_part);
    } else if (C1 <= 64) hipLaunchKernelGGL(group_linear_bwd_kernel<1>, grid, block, 0, st, rl, rf, CF, dU, Y, a, k1, k2, mu, row_src, n_rows_dev, C1, dUf, dWx_part, dWf_part);
    else if (C1 <= 128) hipLaunchKernelGGL(group_linear_bwd_kernel<2>, grid, block, 0, st, rl, rf, CF, dU, Y, a, k1, k2, mu, row_src, n_rows_dev, C1, dUf, dWx_part, dWf_part);
    else hipLaunchKernelGGL(group_linear_bwd_kernel<4>, grid, block, 0, st, rl, rf, CF, dU, Y, a, k1, k2, mu, row_src, n_rows_dev, C1, dUf, dWx_part, dWf_part);
    int rc = check_launch("pcl_group_linear_bwd_f32");
    if (rc || !dW0) return rc;
    const int n = (dWx_part ? C1 * 3 : 0) + (dWf_part ? C1 * CF : 0);
    hipLaunchKernelGGL(group_linear_dw_kernel, dim3((n + 3) / 4), block, 0, st, dWx_part, dWf_part, GL_BLOCKS, C1, CF, off, ldw, dW0);
    return check_launch("pcl_group_linear_bwd_f32(dW)");
}

/* in_off [B*N + 1], in_rows [group_off[B*m]]: every source point's rows (indices into the compacted row table), ascending */
extern "C" int pcl_group_rows_transpose_supported(int N, int m, int ns) {
    return N >= 1 && m >= 1 && ns >= 1 && ((size_t)(RT_W + 1) * N + 1 + (size_t)m * ns) * 4 <= 150 * 1024;
}
extern "C" int pcl_group_rows_transpose_i32(const int32_t* row_src, const int32_t* group_off, int B, int N, int m, int ns, int32_t* in_off,
                                            int32_t* in_rows, void* stream) {
    PCL_REQUIRE(row_src && group_off && in_off && in_rows && B >= 1 && m >= 1 && ns >= 1, "pcl_group_rows_transpose_i32: bad arguments");
    PCL_REQUIRE(pcl_group_rows_transpose_supported(N, m, ns), "pcl_group_rows_transpose_i32: N=%d m=%d ns=%d beyond the LDS-resident form", N, m, ns);
    const size_t lds = ((size_t)(RT_W + 1) * N + 1 + (size_t)m * ns) * 4;
    auto kern = rows_transpose_kernel;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(PCL_EHIP, "pcl_group_rows_transpose_i32: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(RT_T), lds, as_stream(stream), row_src, group_off, m, N, B, in_off, in_rows);
    return check_launch("pcl_group_rows_transpose_i32");
}
/* pcl_group_linear_bwd_f32 with the scatter as a gather over the points' row lists: dUf [B*N][C1] is WRITTEN (no zero-fill, no atomics),
 * dWx_part as there; C1 in {64, 128, 256}, 16-byte aligned rows */
static int g_scatter_gather = 1;        // lab switch (pcl_set_scatter_form): 0 = the fp32-atomic scatter everywhere
extern "C" void pcl_set_scatter_form(int gather) { if (gather >= 0) g_scatter_gather = gather != 0; }
extern "C" int pcl_group_linear_bwd_gather_supported(int C1) { return g_scatter_gather && (C1 == 64 || C1 == 128 || C1 == 256); }
extern "C" int pcl_group_linear_bwd_gather_f32(const float* row_loc, const float* dU, const float* Y, const float* a, const float* k1,
                                               const float* k2, const float* mu, const int32_t* in_off, const int32_t* in_rows, int B, int N,
                                               int C1, float* dUf, float* dWx_part, float* dW0, int ldw, void* stream) {
    PCL_REQUIRE(row_loc && dU && Y && a && k1 && k2 && mu && in_off && in_rows && dUf, "pcl_group_linear_bwd_gather_f32: null pointer");
    PCL_REQUIRE(B >= 1 && N >= 1 && pcl_group_linear_bwd_gather_supported(C1), "pcl_group_linear_bwd_gather_f32: bad sizes (C1 = %d: 64, 128 or 256)", C1);
    PCL_REQUIRE(((reinterpret_cast<uintptr_t>(dU) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(dUf)) & 15) == 0, "pcl_group_linear_bwd_gather_f32: 16-byte aligned rows");
    PCL_REQUIRE(!dW0 || (dWx_part && ldw >= 3), "pcl_group_linear_bwd_gather_f32: dW0 needs dWx_part and ldw >= 3");
    hipStream_t st = as_stream(stream);
    const float4* rl = reinterpret_cast<const float4*>(row_loc);
    const dim3 grid(GL_BLOCKS), block(256);
    const int P = B * N;
    if (C1 == 64) hipLaunchKernelGGL(group_linear_bwd_gather_kernel<16>, grid, block, 0, st, rl, dU, Y, a, k1, k2, mu, in_off, in_rows, P, dUf, dWx_part);
    else if (C1 == 128) hipLaunchKernelGGL(group_linear_bwd_gather_kernel<32>, grid, block, 0, st, rl, dU, Y, a, k1, k2, mu, in_off, in_rows, P, dUf, dWx_part);
    else hipLaunchKernelGGL(group_linear_bwd_gather_kernel<64>, grid, block, 0, st, rl, dU, Y, a, k1, k2, mu, in_off, in_rows, P, dUf, dWx_part);
    int rc = check_launch("pcl_group_linear_bwd_gather_f32");
    if (rc || !dW0) return rc;
    hipLaunchKernelGGL(group_linear_dw_kernel, dim3((C1 * 3 + 3) / 4), block, 0, st, dWx_part, (const float*)nullptr, GL_BLOCKS, C1, 0, 0, ldw, dW0);
    return check_launch("pcl_group_linear_bwd_gather_f32(dW)");
}
