// The wide FC layer (R <= 32 rows, K = 16384 -> N = 1024: PointConv's per-point Linear on the GroupAll level; 2048 -> 512: DGCNN's first
// FC layer) on the fp32 MFMA with split K -- candidates for csrc/head.hip's *_wide_kernel (VALU, 8 columns per workgroup: 111 / 98 / 79 us
// at 16384 x 1024 for 67 MB of weight traffic each).  Self-checking against a CPU fp64 reference.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hw tools/ubench/head_wide_mfma.hip && /tmp/hw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 128;                                   // k per staged chunk

// ---- forward: part[ks][32][N] = X[32, k-range ks] . W[N, k-range ks]^T; grid (N/32, KS), 4 waves split every chunk's k
__global__ __launch_bounds__(256) void fwd_part(const float* __restrict__ X, const float* __restrict__ W, int R, int K, int N, int kper,
                                                float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float sX[32][KC + 4];
    __shared__ __attribute__((aligned(16))) float sW[32][KC + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 32, kbeg = blockIdx.y * kper, kend = min(K, kbeg + kper);
    const int row = tid >> 3, kq = (tid & 7) * 4;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 px[4], pw[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kq + 32 * i;
            const bool kin = k < kend;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            px[i] = (kin && row < R) ? *reinterpret_cast<const float4*>(X + (size_t)row * K + k) : z;
            pw[i] = (kin && n0 + row < N) ? *reinterpret_cast<const float4*>(W + (size_t)(n0 + row) * K + k) : z;
        }
    };
    load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&sX[row][kq + 32 * i]) = px[i];
            *reinterpret_cast<float4*>(&sW[row][kq + 32 * i]) = pw[i];
        }
        __syncthreads();
        if (k0 + KC < kend) load(k0 + KC);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {                    // this wave's 32 k of the chunk: lanes lh = 0 / 1 take k 0-3 / 4-7 of every 8
            const float4 a = *reinterpret_cast<const float4*>(&sX[lr][wave * 32 + kk * 8 + lh * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&sW[lr][wave * 32 + kk * 8 + lh * 4]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    // the four waves' partial tiles meet in LDS (fixed order), one partial tile per workgroup
    __syncthreads();
    float* red = &sX[0][0];                                  // [4][32][33] floats = 16.9 KB <= sX + sW
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh) * 33 + lr] = acc[i];
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const int r = e >> 5, c = e & 31;
        const float s = (red[(0 * 32 + r) * 33 + c] + red[(1 * 32 + r) * 33 + c]) + (red[(2 * 32 + r) * 33 + c] + red[(3 * 32 + r) * 33 + c]);
        if (n0 + c < N) part[((size_t)blockIdx.y * 32 + r) * N + n0 + c] = s;
    }
}

// ---- dW[n, k] = sum_r dy[r, n] X[r, k]: grid (N/32, k groups); a wave walks 32-wide k blocks, one 32 x 32 tile of dW per block (16 MFMAs)
__global__ __launch_bounds__(256) void dw_tiles(const float* __restrict__ dY, const float* __restrict__ X, int R, int K, int N, int kb_per_wg,
                                                float* __restrict__ dW) {
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
    for (int j = 0; j < 16; ++j) a[j] = sD[2 * j + lh][lr];   // A operand: row = n (lr), k index = r
    const int kblocks = (K + 31) / 32;
    for (int kb = blockIdx.y * kb_per_wg + wave; kb < min(kblocks, (int)(blockIdx.y + 1) * kb_per_wg); kb += 4) {
        const int k = kb * 32 + lr;
        const int kc = min(k, K - 1);
        float b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) b[j] = X[(size_t)min(2 * j + lh, R - 1) * K + kc];   // (rows past R: dy is 0 there)
        f32x16 acc;
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

// ---- dX[r, k] += sum_{n in split} dy[r, n] W[n, k]: grid (K/128, NS); wave = one 32-wide k block, dy staged 128 columns at a time
__global__ __launch_bounds__(256) void dx_tiles(const float* __restrict__ dY, const float* __restrict__ W, int R, int K, int N, int nper,
                                                float* __restrict__ dX) {
    __shared__ float sD[32][128 + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int k = (blockIdx.x * 4 + wave) * 32 + lr, kc = min(k, K - 1);
    const int nbeg = blockIdx.y * nper, nend = min(N, nbeg + nper);
    f32x16 acc;
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

static double relerr(const std::vector<float>& got, const std::vector<double>& want) {
    double e = 0, m = 0;
    for (size_t i = 0; i < want.size(); ++i) { e = fmax(e, fabs(got[i] - want[i])); m = fmax(m, fabs(want[i])); }
    return e / m;
}
int main() {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct Case { int R, K, N, KS; } cases[] = {{32, 16384, 1024, 8}, {32, 16384, 1024, 16}, {32, 16384, 1024, 32}, {32, 2048, 512, 8}, {32, 2048, 512, 16}, {29, 4096, 130, 4}, {7, 2048, 20, 8}};
    for (auto c : cases) {
        const int R = c.R, K = c.K, N = c.N, KS = c.KS;
        std::vector<float> hX((size_t)R * K), hW((size_t)N * K), hdY((size_t)R * N);
        srand(1);
        for (auto& v : hX) v = (rand() % 2001 - 1000) * 1e-3f;
        for (auto& v : hW) v = (rand() % 2001 - 1000) * 1e-3f / sqrtf((float)K);
        for (auto& v : hdY) v = (rand() % 2001 - 1000) * 1e-3f;
        float *X, *W, *dY, *part, *dW, *dX;
        CHK(hipMalloc(&X, hX.size() * 4)); CHK(hipMalloc(&W, hW.size() * 4)); CHK(hipMalloc(&dY, hdY.size() * 4));
        CHK(hipMalloc(&part, (size_t)KS * 32 * N * 4)); CHK(hipMalloc(&dW, hW.size() * 4)); CHK(hipMalloc(&dX, hX.size() * 4));
        CHK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(dY, hdY.data(), hdY.size() * 4, hipMemcpyHostToDevice));
        const int kper = ((K + KS - 1) / KS + KC - 1) / KC * KC;
        const int kblocks = (K + 31) / 32, gy = kblocks >= 128 ? 32 : 8, kb_per_wg = (kblocks + gy - 1) / gy;
        const int NS = N >= 512 ? 2 : 1, nper = ((N + NS - 1) / NS + 127) / 128 * 128;
        float t[3] = {1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 5; ++rep) {
            float ms;
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(fwd_part, dim3((N + 31) / 32, KS), dim3(256), 0, 0, X, W, R, K, N, kper, part);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); t[0] = fminf(t[0], ms);
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(dw_tiles, dim3((N + 31) / 32, gy), dim3(256), 0, 0, dY, X, R, K, N, kb_per_wg, dW);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); t[1] = fminf(t[1], ms);
            CHK(hipMemset(dX, 0, hX.size() * 4));
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(dx_tiles, dim3((K + 127) / 128, NS), dim3(256), 0, 0, dY, W, R, K, N, nper, dX);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); t[2] = fminf(t[2], ms);
        }
        std::vector<float> hp((size_t)KS * 32 * N), gW(hW.size()), gX(hX.size()), gY((size_t)R * N);
        CHK(hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(gW.data(), dW, gW.size() * 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(gX.data(), dX, gX.size() * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < R; ++r) for (int n = 0; n < N; ++n) { float s = 0; for (int ks = 0; ks < KS; ++ks) s += hp[((size_t)ks * 32 + r) * N + n]; gY[(size_t)r * N + n] = s; }
        // CPU reference on a sample of outputs
        std::vector<double> wY, wW, wX; std::vector<float> sY, sW, sX;
        for (int q = 0; q < 400; ++q) {
            const int r = rand() % R, n = rand() % N, k = rand() % K;
            double a = 0; for (int kk = 0; kk < K; ++kk) a += (double)hX[(size_t)r * K + kk] * hW[(size_t)n * K + kk];
            wY.push_back(a); sY.push_back(gY[(size_t)r * N + n]);
            double b = 0; for (int rr = 0; rr < R; ++rr) b += (double)hdY[(size_t)rr * N + n] * hX[(size_t)rr * K + k];
            wW.push_back(b); sW.push_back(gW[(size_t)n * K + k]);
            double d = 0; for (int nn = 0; nn < N; ++nn) d += (double)hdY[(size_t)r * N + nn] * hW[(size_t)nn * K + k];
            wX.push_back(d); sX.push_back(gX[(size_t)r * K + k]);
        }
        const double mb = (double)N * K * 4e-6;
        printf("R=%2d K=%5d N=%4d: fwd %6.1f us  dW %6.1f us  dX %6.1f us  (weight %5.1f MB: %4.1f us at 5 TB/s)   rel err fwd %.1e dW %.1e dX %.1e\n",
               R, K, N, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, mb, mb / 5.0, relerr(sY, wY), relerr(sW, wW), relerr(sX, wX));
        hipFree(X); hipFree(W); hipFree(dY); hipFree(part); hipFree(dW); hipFree(dX);
    }
    return 0;
}
