// sgd.hip -- the optimiser step of the training loops as ONE launch per <= 96 parameter tensors (round 5).
//
// Reference: nn.SGD(net.parameters(), lr, momentum) of /root/reference/train_cls.py:404 (train_partseg.py: weight_decay = 1e-4):
//     g += wd * p;  v = mu * v + (1 - dampening) * g;  p -= lr * v          (no Nesterov)
// The host loop used torch's multi-tensor fused SGD: 29.5 us for the 1.47 M parameters of PointNet++ SSG cls, 3 x 45 us for PointConv's
// 19.6 M (335 MB of parameter, gradient and momentum traffic at 2.4 TB/s) -- its chunk table travels as kernel arguments 110 tensors at a
// time and a block handles one 64 KB chunk.  Here: the same arithmetic (the products and sums in fp64 of the fp32 operands, rounded once per
// statement -- what torch's kernel computes with its `double` hyper-parameters, so the two agree bit for bit), a table of
// (p, g, v, first block) per tensor as kernel arguments, 16-byte accesses where the three pointers allow (gradients are slices of a
// per-stack flat buffer: any 4-byte alignment), a block = 4 096 elements.
#include "common.h"

namespace pcl {

constexpr int SGD_MAXT = 96;                  // tensors per launch (kernel arguments: 96 x 24 B of pointers + 97 x 4 B of block offsets)
constexpr int SGD_CHUNK = 4096;               // elements per block: 256 threads x 4 x float4
struct SgdTable {
    float* p[SGD_MAXT]; const float* g[SGD_MAXT]; float* v[SGD_MAXT];
    unsigned first[SGD_MAXT + 1];             // first block of tensor t; first[nt] = grid size
    unsigned n[SGD_MAXT];                     // elements (< 2^32)
    int nt;
    double lr, mu, wd, one_minus_damp;
};

__device__ __forceinline__ void sgd_one(float& p, float g, float& v, const SgdTable& t) {
    if (t.wd != 0.0) g = (float)((double)g + t.wd * (double)p);
    v = (float)(t.mu * (double)v + t.one_minus_damp * (double)g);
    p = (float)((double)p - t.lr * (double)v);
}

__global__ __launch_bounds__(256) void sgd_momentum_kernel(const SgdTable t) {
    // tensor of this block: binary search over the block offsets (wave-uniform)
    int lo = 0, hi = t.nt - 1;
    const unsigned b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.first[mid] <= b) lo = mid; else hi = mid - 1;
    }
    float* __restrict__ p = t.p[lo];
    const float* __restrict__ g = t.g[lo];
    float* __restrict__ v = t.v[lo];
    const unsigned n = t.n[lo];
    const unsigned e0 = (b - t.first[lo]) * SGD_CHUNK;
    const bool al16 = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (al16 && e0 + SGD_CHUNK <= n) {
        float4 P[4], Gr[4], V[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned e = e0 + (i * 256 + threadIdx.x) * 4;
            P[i] = *reinterpret_cast<const float4*>(p + e); Gr[i] = *reinterpret_cast<const float4*>(g + e); V[i] = *reinterpret_cast<const float4*>(v + e);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sgd_one(P[i].x, Gr[i].x, V[i].x, t); sgd_one(P[i].y, Gr[i].y, V[i].y, t); sgd_one(P[i].z, Gr[i].z, V[i].z, t); sgd_one(P[i].w, Gr[i].w, V[i].w, t);
            const unsigned e = e0 + (i * 256 + threadIdx.x) * 4;
            *reinterpret_cast<float4*>(p + e) = P[i]; *reinterpret_cast<float4*>(v + e) = V[i];
        }
    } else {
        const unsigned e1 = min(n, e0 + SGD_CHUNK);
        for (unsigned e = e0 + threadIdx.x; e < e1; e += 256) {
            float pp = p[e], vv = v[e];
            sgd_one(pp, g[e], vv, t);
            p[e] = pp; v[e] = vv;
        }
    }
}

}  // namespace pcl
using namespace pcl;

/* params / grads / bufs: HOST arrays of n_tensors device pointers; numel: host array of element counts.  Every tensor has a gradient and a
 * momentum buffer (the first step of an optimiser, which creates the buffers as v = g, stays with the caller). */
extern "C" int pcl_sgd_momentum_f32(const uint64_t* params, const uint64_t* grads, const uint64_t* bufs, const int64_t* numel, int n_tensors, double lr,
                                    double momentum, double weight_decay, double dampening, void* stream) {
    PCL_REQUIRE(params && grads && bufs && numel && n_tensors >= 0, "pcl_sgd_momentum_f32: null table");
    hipStream_t st = as_stream(stream);
    for (int t0 = 0; t0 < n_tensors; t0 += SGD_MAXT) {
        SgdTable t = {};
        t.lr = lr; t.mu = momentum; t.wd = weight_decay; t.one_minus_damp = 1.0 - dampening;
        unsigned blocks = 0;
        int k = 0;
        for (int i = t0; i < n_tensors && i < t0 + SGD_MAXT; ++i) {
            if (numel[i] == 0) continue;
            PCL_REQUIRE(params[i] && grads[i] && bufs[i], "pcl_sgd_momentum_f32: tensor %d: null pointer", i);
            PCL_REQUIRE(numel[i] > 0 && numel[i] < (int64_t)0xffffffffll - SGD_CHUNK, "pcl_sgd_momentum_f32: tensor %d: %lld elements", i, (long long)numel[i]);
            t.p[k] = reinterpret_cast<float*>(params[i]); t.g[k] = reinterpret_cast<const float*>(grads[i]); t.v[k] = reinterpret_cast<float*>(bufs[i]);
            t.n[k] = (unsigned)numel[i]; t.first[k] = blocks;
            blocks += (unsigned)((numel[i] + SGD_CHUNK - 1) / SGD_CHUNK);
            ++k;
        }
        t.nt = k; t.first[k] = blocks;
        if (k == 0) continue;
        hipLaunchKernelGGL(sgd_momentum_kernel, dim3(blocks), dim3(256), 0, st, t);
        const int rc = check_launch("pcl_sgd_momentum_f32");
        if (rc) return rc;
    }
    return PCL_OK;
}
