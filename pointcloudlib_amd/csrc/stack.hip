// stack.hip -- per-stack entry points: one C-ABI call runs a whole [1x1 conv -> BatchNorm(train) -> (Leaky)ReLU] x L
// stack (+ the ball-query grouping folded into its first layer, + the max over the group), forward or backward.
//
// The reference runs one `execute` per module (networks/cls/pointnet2.py:33-62: group -> mlp -> max): that is the unit
// these entry points take.  Nothing new is computed here -- the kernels are the library's own (mlp.hip, compact.hip),
// launched in the order misc/mlp_hip.py launches them one ctypes call at a time; what changes is the host side: one call,
// one descriptor, one persistent + one transient buffer carved here instead of ~25 allocations and ~15 Python-level
// calls per stack and direction (the host needed 1.4-1.7 ms to enqueue a 2.1 ms PointNet++ step).
//
// Supported (everything else stays on the per-kernel entry points, see misc/mlp_hip.py):
//   training-mode BatchNorm on every layer, statistics local to this process; first layer either plain (x [P, c0] given)
//   or folded into the grouping of a set-abstraction level (`grouped`: pcl_group_linear_f32, duplicate-compacted rows).
#include "common.h"
#include <string.h>

namespace pcl {

static inline size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

struct Carver {
    char* base; size_t off;
    explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <typename T> T* take(size_t count) {
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += al256(count * sizeof(T));
        return r;
    }
};

// ---- layout of the persistent buffer (`save`: lives from forward to backward) --------------------------------------
struct SaveLayout {
    float* Y[PCL_STACK_MAX_LAYERS];        // pre-BatchNorm outputs [P, c[l+1]]
    float* vec[PCL_STACK_MAX_LAYERS];      // [4][cout]: scale, shift, mean, invstd
    int32_t* arg; float* ymax;             // pooled stacks: [G, cL]
    int32_t* row_meta; int32_t* row_src; float* row_loc; float* row_feat;     // grouped stacks
    int32_t* in_off; int32_t* in_rows;     // grouped, wide features: every source point's rows (the backward's scatter as a gather)
    size_t bytes;
};

static inline bool grouped_inline(const pcl_mlp_stack_t& d) { return d.grouped && d.Cf > 0 && d.Cf <= 4 && !d.need_dx; }
static inline bool grouped_wide(const pcl_mlp_stack_t& d) { return d.grouped && d.Cf > 0 && !grouped_inline(d); }
static inline int pooled_groups(const pcl_mlp_stack_t& d) { return d.grouped ? d.B * d.m : (d.pool ? d.P / d.pool : 0); }
// the folded first layer's backward walks the rows by source point (compact.hip: group_linear_bwd_gather_kernel) where the lists can be built
static inline bool gather_scatter(const pcl_mlp_stack_t& d) {
    return grouped_wide(d) && pcl_group_linear_bwd_gather_supported(d.c[1]) && pcl_group_rows_transpose_supported(d.N, d.m, d.pool);
}

static SaveLayout save_layout(const pcl_mlp_stack_t& d, void* base) {
    SaveLayout s = {};
    Carver c(base);
    const int L = d.n_layers;
    for (int l = 0; l < L; ++l) {
        s.Y[l] = c.take<float>((size_t)d.P * d.c[l + 1]);
        s.vec[l] = c.take<float>((size_t)4 * d.c[l + 1]);
    }
    const int G = pooled_groups(d);
    if (G) { s.arg = c.take<int32_t>((size_t)G * d.c[L]); s.ymax = c.take<float>((size_t)G * d.c[L]); }
    if (d.grouped) {
        s.row_meta = c.take<int32_t>((size_t)d.P * 2);
        s.row_src = c.take<int32_t>((size_t)d.P);
        s.row_loc = c.take<float>((size_t)d.P * 4);
        if (grouped_inline(d)) s.row_feat = c.take<float>((size_t)d.P * 4);
        if (gather_scatter(d)) { s.in_off = c.take<int32_t>((size_t)d.B * d.N + 1); s.in_rows = c.take<int32_t>((size_t)d.P); }
    }
    s.bytes = c.off;
    return s;
}

// forward GEMMs of a PLAIN stack through the fragment kernels with fp64-flushed accumulation (descriptor field flush_k)
static inline bool frag_fwd(const pcl_mlp_stack_t& d) { return !d.grouped && d.flush_k != 0; }
static inline bool use_gmax(const pcl_mlp_stack_t& d) { return !d.grouped && !d.flush_k && (d.pool == 32 || d.pool == 64); }
static inline bool fused_bwd_enabled() {
    return path_switches().fused_backward != 0;
}

// ---- transient buffer of the forward -------------------------------------------------------------------------------
struct FwdTmp {
    double* stats;                  // one region, reused layer after layer (stream order: finalize reads it before the next GEMM writes)
    float* Uf; double* pt_stats;    // grouped, wide features: the per-point product feat Wf^T and its (unused) sums
    float *gmax, *gmin; int32_t *gamax, *gamin;
    size_t bytes;
};
static FwdTmp fwd_tmp(const pcl_mlp_stack_t& d, void* base) {
    FwdTmp t = {};
    Carver c(base);
    const int L = d.n_layers;
    size_t rows_max = 0;
    for (int l = 0; l < L; ++l) {
        size_t r;
        if (l == 0 && d.grouped) r = (size_t)pcl_group_linear_stat_rows(d.B, d.m) * 2 * d.c[1];
        else if (frag_fwd(d)) r = (size_t)pcl_frag_stat_rows(d.P) * 2 * d.c[l + 1];
        else r = (size_t)pcl_mlp_stat_rows(d.P, d.c[l + 1], d.grouped ? 2 : 0) * 2 * d.c[l + 1];
        if (r > rows_max) rows_max = r;
    }
    t.stats = c.take<double>(rows_max);
    if (grouped_wide(d)) {
        t.Uf = c.take<float>((size_t)d.B * d.N * d.c[1]);
        if (!d.flush_k) t.pt_stats = c.take<double>((size_t)pcl_mlp_stat_rows(d.B * d.N, d.c[1], 0) * 2 * d.c[1]);     // (the flushed product writes no sums)
    }
    if (use_gmax(d)) {
        const size_t n = (size_t)(d.P / d.pool) * d.c[L];
        t.gmax = c.take<float>(n); t.gmin = c.take<float>(n); t.gamax = c.take<int32_t>(n); t.gamin = c.take<int32_t>(n);
    }
    t.bytes = c.off;
    return t;
}

// ---- transient buffer of the backward ------------------------------------------------------------------------------
struct BwdTmp {
    float* gz;                      // [G, cL] sparse max gradient
    double* stats[2];               // BatchNorm-backward sums, ping-pong
    float* dU[2];                   // [P, max c] gradient w.r.t. a BatchNorm output, ping-pong
    float* consts[2];               // [3][max c]: a, k1, k2 (ping-pong: the fused finish writes the next layer's while this one's is in use)
    float* ws;                      // partial dW tiles (fused kernel / dW kernel)
    size_t ws_bytes;
    float* dUf; float* dWxp; float* dWfp;      // grouped
    float* unit;                               // grouped wide: [C1] ones then [2*C1] zeros: (a, k1, k2, mu) of the plain point GEMMs
    float* ptws; size_t ptws_bytes;            // grouped wide: workspace of the point dW GEMM
    // weight gradients on the side stream (side_dw): what a dW launch reads must outlive the main chain's next layers, so the
    // BatchNorm-backward constants and the dU tensors are per layer instead of ping-pong
    float* consts_l[PCL_STACK_MAX_LAYERS];
    float* dU_l[PCL_STACK_MAX_LAYERS];         // dU_l[l] = gradient w.r.t. the BatchNorm output of layer l (written by layer l + 1's dX)
    float* dy_l[PCL_STACK_MAX_LAYERS];         // few-row layers: dy formed once per layer (pcl_bn_bwd_dy_f32), read by its dX and dW
    size_t bytes;
};
// Few-row plain stacks (the GroupAll level: 4 096 rows; the part-seg decoder: 2 048 .. 8 192): the weight gradient of a layer is not
// on the critical path of the backward -- the chain is consts(l) -> dX(l) -> consts(l - 1) -> ... -- and every GEMM of these layers is
// too short to fill the chip (fill / drain, a split-K tail and a reduce launch around 20-50 us of matrix work).  So the dW launches
// (+ their reduces) go to a second stream of the library, forked after consts(l) and joined at the end of the call: they run in the
// holes of the dX chain.  Results do not depend on the interleaving (same kernels, same grids, per-layer buffers).
static int g_side_dw = 0, g_side_dw_max_rows = 8192;
// Few-row layers (same stacks): BatchNorm-backward constants + dy in ONE launch, then dX on the fragment-direct kernel and dW on the staged
// kernel, both reading the formed dy (csrc/mlp.hip: bn_bwd_dy_kernel).  Per layer: no fused dX + dW kernel for the shape, Cout % 32 == 0.
static int g_fewrow = 0;
bool frag_rows_eligible(int P);
int frag_stat_rows(int P);
static inline bool fewrow_layer(const pcl_mlp_stack_t& d, int l) {
    if (!g_fewrow || d.grouped || !frag_rows_eligible(d.P)) return false;
    if (l > 0 && fused_bwd_enabled() && pcl_linear_bwd_fused_supported(d.c[l + 1], d.c[l])) return false;
    return pcl_bn_bwd_dy_supported(d.P, d.c[l + 1]) != 0;
}
static inline bool side_dw(const pcl_mlp_stack_t& d) {
    if (!g_side_dw || d.grouped || d.P > g_side_dw_max_rows || d.n_layers < 2) return false;
    for (int l = 1; l < d.n_layers; ++l) if (fused_bwd_enabled() && pcl_linear_bwd_fused_supported(d.c[l + 1], d.c[l])) return false;
    return true;
}

// the library's second stream and the two events of a fork / join, per host thread and device (created on first use, never destroyed:
// process lifetime).  Non-blocking stream: it synchronises with the caller's stream through the events only.
struct SideCtx { hipStream_t st; hipEvent_t fork, join; int state; };       // state: 0 = not created, 1 = ready, -1 = creation failed
static SideCtx* side_ctx() {
    static thread_local SideCtx ctx[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
    SideCtx& c = ctx[dev];
    if (c.state == 0) {
        const bool ok = hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&c.join, hipEventDisableTiming) == hipSuccess;
        c.state = ok ? 1 : -1;
        if (!ok) (void)hipGetLastError();
    }
    return c.state == 1 ? &c : nullptr;
}

static BwdTmp bwd_tmp(const pcl_mlp_stack_t& d, void* base) {
    BwdTmp t = {};
    Carver c(base);
    const int L = d.n_layers;
    int cmax = 0;
    for (int l = 1; l <= L; ++l) cmax = d.c[l] > cmax ? d.c[l] : cmax;
    const int cin_max = d.c[0] > cmax ? d.c[0] : cmax;
    const int G = pooled_groups(d);
    if (G) t.gz = c.take<float>((size_t)G * d.c[L]);
    for (int i = 0; i < 2; ++i) t.stats[i] = c.take<double>((size_t)1024 * 2 * cin_max);
    // dU buffers: the widest gradient that is ever materialised: layers' outputs (not the last one when it is pooled) and,
    // for a plain stack, nothing for the input (dx is the caller's)
    int cdu = 0;
    for (int l = 1; l <= L; ++l) if (!(l == L && G)) cdu = d.c[l] > cdu ? d.c[l] : cdu;
    for (int i = 0; i < 2; ++i) t.dU[i] = c.take<float>((size_t)d.P * (cdu ? cdu : 1));
    for (int i = 0; i < 2; ++i) t.consts[i] = c.take<float>((size_t)3 * cmax);
    size_t wsb = 0;
    for (int l = 0; l < L; ++l) {
        if (l == 0 && d.grouped) continue;
        const int cin = d.c[l], cout = d.c[l + 1];
        size_t b = pcl_linear_bwd_dw_workspace_bytes(d.P, cout, cin);
        if (fewrow_layer(d, l)) {
            const size_t f = pcl_linear_bwd_dw_plain_workspace_bytes(d.P, cout, cin);
            if (f > b) b = f;
        }
        if (l > 0 && pcl_linear_bwd_fused_supported(cout, cin)) {
            const size_t f = pcl_linear_bwd_fused_workspace_bytes(d.P, cout, cin);
            if (f > b) b = f;
        }
        if (b > wsb) wsb = b;
    }
    t.ws_bytes = wsb;
    t.ws = c.take<float>((wsb + 3) / 4);
    for (int l = 0; l < L; ++l)
        if (fewrow_layer(d, l)) t.dy_l[l] = c.take<float>((size_t)d.P * d.c[l + 1]);
    if (side_dw(d)) {
        for (int l = 0; l < L; ++l) {
            t.consts_l[l] = c.take<float>((size_t)3 * d.c[l + 1]);
            if (l < L - 1 || !G) t.dU_l[l] = c.take<float>((size_t)d.P * d.c[l + 1]);
        }
    }
    if (d.grouped) {
        const int rows = pcl_group_linear_stat_rows(d.B, d.m), C1 = d.c[1];
        if (grouped_wide(d)) {
            t.dUf = c.take<float>(((size_t)d.B * d.N * C1 + 3) / 4 * 4);
            t.unit = c.take<float>((size_t)3 * C1);
            t.ptws_bytes = pcl_linear_bwd_dw_workspace_bytes(d.B * d.N, C1, d.Cf);
            t.ptws = c.take<float>((t.ptws_bytes + 3) / 4);
        }
        if (d.use_xyz) t.dWxp = c.take<float>((size_t)rows * C1 * 3);
        if (grouped_inline(d)) t.dWfp = c.take<float>((size_t)rows * C1 * d.Cf);
    }
    t.bytes = c.off;
    return t;
}

bool narrow_supported(const pcl_mlp_stack_t& d);            // narrow.hip: three layers of <= 16 channels, recomputed per pass
size_t narrow_save_bytes();
size_t narrow_fwd_tmp_bytes();
size_t narrow_bwd_tmp_bytes();
int narrow_fwd(const pcl_mlp_stack_t& d);
int narrow_bwd(const pcl_mlp_stack_t& d);
int maxgrad_prep_impl(const float* gout, const float* out, const float* ymax, float slope, int G, int C, float* gz, double* stats_ws,
                      int* stat_rows_out, void* stream, float* zero, size_t n_zero, float* ones, int n_one, int n_one0, int ldg);
int pair_finish_impl(const void* workspace, size_t workspace_bytes, int P, int Cout, int Cin, float* dW, int dw_ld,
                     const double* stats_ws, int stat_rows, const float* gamma_prev, const float* mean_prev,
                     const float* invstd_prev, int P_bn, float* dgamma_prev, float* dbeta_prev,
                     float* a_prev, float* k1_prev, float* k2_prev, float* dbias_zero_prev, void* stream,
                     const float* xpart, int xrows, int xC1, int xld, float* xdW0);
int group_linear_fwd_impl(const float* xyz, const float* new_xyz, const float* Uf, const float* Wx, const float* feat_small,
                          const float* Wf_small, int CF, int ldw, const int32_t* idx, const int32_t* cnt, const int32_t* group_off, int B,
                          int N, int m, int ns, int C1, float* Y, int32_t* row_meta, int32_t* row_src, float* row_loc, float* row_feat,
                          double* stats_ws, void* stream, int phase, const float* dense_src, float* dense_dst, int dense_cols);
int group_linear_bwd_impl(const float* row_loc, const float* row_feat, int CF, const float* dU, const float* Y, const float* a,
                          const float* k1, const float* k2, const float* mu, const int32_t* row_src, const int32_t* n_rows_dev, int B, int N,
                          int C1, float* dUf, float* dWx_part, float* dWf_part, float* dW0, int ldw, int off, void* stream, bool duf_is_zero);

__global__ __launch_bounds__(256) void aux_fill_kernel(float4* zero, size_t n4, float* ones, int n_one, int n_zero) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) zero[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && ones)
        for (int e = threadIdx.x; e < n_one + n_zero; e += 256) ones[e] = e < n_one ? 1.f : 0.f;
}

static int validate(const pcl_mlp_stack_t* dp, const char* who) {
    PCL_REQUIRE(dp, "%s: null descriptor", who);
    const pcl_mlp_stack_t& d = *dp;
    PCL_REQUIRE(d.struct_bytes == (int32_t)sizeof(pcl_mlp_stack_t), "%s: descriptor is %d bytes, this library expects %zu (header mismatch)",
                who, d.struct_bytes, sizeof(pcl_mlp_stack_t));
    PCL_REQUIRE(d.n_layers >= 1 && d.n_layers <= PCL_STACK_MAX_LAYERS, "%s: n_layers=%d (1..%d)", who, d.n_layers, PCL_STACK_MAX_LAYERS);
    PCL_REQUIRE(d.P >= 1, "%s: P=%d", who, d.P);
    for (int l = 0; l <= d.n_layers; ++l) PCL_REQUIRE(d.c[l] >= 1, "%s: c[%d]=%d", who, l, d.c[l]);
    PCL_REQUIRE(d.pool >= 0 && (d.pool == 0 || d.grouped || d.P % d.pool == 0), "%s: pool=%d does not divide P=%d", who, d.pool, d.P);
    PCL_REQUIRE(!d.defer_act || d.pool == 0 || d.grouped, "%s: defer_act: a plain stack must have no max (a grouped one keeps its rows: no pooling)", who);
    if (d.grouped) {
        PCL_REQUIRE(d.n_layers >= 2, "%s: a grouped stack needs >= 2 layers (the folded layer's gradient arrives dense from the second)", who);
        PCL_REQUIRE(d.B >= 1 && d.N >= 1 && d.m >= 1 && d.pool >= 1 && d.P == d.B * d.m * d.pool, "%s: grouped: P=%d must be B*m*ns = %d*%d*%d", who, d.P, d.B, d.m, d.pool);
        PCL_REQUIRE(d.idx && d.cnt && d.group_off && d.xyz && d.new_xyz, "%s: grouped: null index / coordinate pointer", who);
        PCL_REQUIRE(d.Cf >= 0 && (d.Cf == 0 || d.feature), "%s: grouped: Cf=%d without features", who, d.Cf);
        PCL_REQUIRE(d.use_xyz || d.Cf > 0, "%s: grouped: neither coordinates nor features", who);
        PCL_REQUIRE(d.c[0] == (d.use_xyz ? 3 : 0) + d.Cf, "%s: grouped: c[0]=%d must be the first layer's fan-in %d", who, d.c[0], (d.use_xyz ? 3 : 0) + d.Cf);
        PCL_REQUIRE(!grouped_wide(d) || d.Wf_dense, "%s: grouped: wide features need Wf_dense", who);
        PCL_REQUIRE(!d.layer[0].bias || grouped_wide(d), "%s: grouped: a conv bias of the folded first layer rides in the per-point product (wide features only)", who);
    } else {
        PCL_REQUIRE(d.x, "%s: null input", who);
    }
    for (int l = 0; l < d.n_layers; ++l)
        PCL_REQUIRE(d.layer[l].W && d.layer[l].gamma && d.layer[l].beta, "%s: layer %d: W / gamma / beta must be given (training-mode BatchNorm on every layer)", who, l);
    PCL_REQUIRE(d.flush_k == 0 || d.flush_k == 8 || d.flush_k == 32, "%s: flush_k = %d (0, 8 or 32)", who, d.flush_k);
    return PCL_OK;
}

#define PCL_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

static void tagf(const char* fmt, int a, int b) {
    char buf[32];
    snprintf(buf, sizeof buf, fmt, a, b);
    set_launch_tag(buf);
}

// ---- hidden widths the fast kernels do not have: run the stack zero-padded (round 5) ----------------------------------------------------
// The MSG part-seg encoder has one stack [3, 64, 96, 128] on B*512*128 grouped rows (networks/seg/pointnet2_partseg.py:185-192).  The
// resident-weight forward and the fused backward exist for widths 64 / 128 / 256; a 96-wide layer took the staged kernels (fwd64x96 +
// fwd96x128 434 us, dx128x96 464, dw128x96 339, dx96x64 216, dw96x64 209 per step: 1.66 of cfg4's 6.1 ms, profiles/r04_cfg4_kernel_stats.csv).
// A 96-wide layer IS a 128-wide layer whose last 32 output channels have zero weights, gamma = beta = 0: their pre-BatchNorm output is
// exactly 0, scale = 0 * invstd = 0 and shift = 0, so the activation is exactly 0, the next layer's extra 32 weight columns multiply zeros,
// every gradient of the pad is exactly 0, and the 96 real channels see the same fma chains with 32 zero terms appended -- the results
// are those of the 96-wide stack, bit for bit per kernel.  One tiny launch builds the padded parameters (kept in `save` for the
// backward), one copies the 96 real entries of the running statistics / the gradients back.  +33 % flops on two layers for kernels that
// run 2-3 x faster.
struct PadItem { const float* src; float* dst; int rows, cols, lds, ldd, rows_p, cols_p; };      // dst [rows_p][ldd] <- src [rows][lds], zero elsewhere (pad) or dst [rows][ldd] <- src (unpad: rows_p = cols_p = 0)
// Capacity = the most a legal descriptor can ask for: per layer one padded weight + five parameter rows (forward), PCL_STACK_MAX_LAYERS layers
// (1 928 bytes of kernel argument).  push() refuses instead of writing past the table (the check used to run after the writes).
constexpr int PAD_ITEMS = PCL_STACK_MAX_LAYERS * 6;
struct PadTable {
    PadItem it[PAD_ITEMS]; int n;
    bool push(const PadItem& q) { if (n >= PAD_ITEMS) return false; it[n++] = q; return true; }
};
__global__ __launch_bounds__(256) void pad_copy_kernel(const PadTable t) {
    const PadItem q = t.it[blockIdx.y];
    const bool unpad = q.rows_p == 0;
    const int R = unpad ? q.rows : q.rows_p, C = unpad ? q.cols : q.cols_p;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < R * C; e += gridDim.x * 256) {
        const int r = e / C, c = e - r * C;
        q.dst[(size_t)r * q.ldd + c] = (r < q.rows && c < q.cols) ? q.src[(size_t)r * q.lds + c] : 0.f;
    }
}
static int launch_pad(const PadTable& t, void* st) {
    if (t.n == 0) return PCL_OK;
    hipLaunchKernelGGL(pad_copy_kernel, dim3(16, t.n), dim3(256), 0, as_stream(st), t);
    return check_launch("pcl_mlp_stack(pad)");
}
static inline int padded_width(const pcl_mlp_stack_t& d, int l) {
    // hidden widths only (the input is the caller's, the output goes to the caller): 96 -> 128 where the row count makes the fast kernels matter
    if (l >= 1 && l < d.n_layers && d.c[l] == 96 && d.P >= 32768 && !d.defer_act) return 128;
    return d.c[l];
}
struct PadPlan {
    bool on;
    int cp[PCL_STACK_MAX_LAYERS + 1];
    // scratch (floats), per layer: padded W, gamma, beta, running mean / var (in `save`, behind the padded stack's own layout) and padded
    // dW, dgamma, dbeta, dbias (in the backward's `tmp`, behind the padded stack's own)
    size_t w_off[PCL_STACK_MAX_LAYERS], v_off[PCL_STACK_MAX_LAYERS], save_floats;
    size_t dw_off[PCL_STACK_MAX_LAYERS], dv_off[PCL_STACK_MAX_LAYERS], bwd_floats;
    bool wpad[PCL_STACK_MAX_LAYERS], opad[PCL_STACK_MAX_LAYERS];      // layer l has a padded weight / padded output channels
};
static PadPlan pad_plan(const pcl_mlp_stack_t& d) {
    PadPlan p = {};
    const int L = d.n_layers;
    for (int l = 0; l <= L; ++l) { p.cp[l] = padded_width(d, l); p.on = p.on || p.cp[l] != d.c[l]; }
    if (!p.on) return p;
    size_t so = 0, bo = 0;
    auto take = [](size_t& o, size_t n) { const size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    for (int l = 0; l < L; ++l) {
        p.opad[l] = p.cp[l + 1] != d.c[l + 1];
        p.wpad[l] = p.opad[l] || p.cp[l] != d.c[l];
        if (p.wpad[l]) { p.w_off[l] = take(so, (size_t)p.cp[l + 1] * p.cp[l]); p.dw_off[l] = take(bo, (size_t)p.cp[l + 1] * p.cp[l]); }
        if (p.opad[l]) { p.v_off[l] = take(so, (size_t)5 * p.cp[l + 1]); p.dv_off[l] = take(bo, (size_t)3 * p.cp[l + 1]); }
    }
    p.save_floats = so; p.bwd_floats = bo;
    return p;
}
// the padded descriptor: widths, and the padded layers' parameter / gradient pointers into the scratch regions
static pcl_mlp_stack_t padded_desc(const pcl_mlp_stack_t& d, const PadPlan& p, float* save_x, float* bwd_x) {
    pcl_mlp_stack_t q = d;
    const int L = d.n_layers;
    for (int l = 0; l <= L; ++l) q.c[l] = p.cp[l];
    for (int l = 0; l < L; ++l) {
        pcl_stack_layer_t& y = q.layer[l];
        if (p.wpad[l]) { y.W = save_x + p.w_off[l]; y.dW = bwd_x ? bwd_x + p.dw_off[l] : nullptr; }
        if (p.opad[l]) {
            float* v = save_x + p.v_off[l];
            const int cp = p.cp[l + 1];
            y.gamma = v; y.beta = v + cp;
            y.running_mean = d.layer[l].running_mean ? v + 2 * cp : nullptr; y.running_var = d.layer[l].running_var ? v + 3 * cp : nullptr;
            y.bias = d.layer[l].bias ? v + 4 * cp : nullptr;
            if (bwd_x) { float* g = bwd_x + p.dv_off[l]; y.dgamma = g; y.dbeta = g + cp; y.dbias = d.layer[l].bias ? g + 2 * cp : nullptr; }
            else { y.dgamma = y.dbeta = y.dbias = nullptr; }
        }
    }
    return q;
}

}  // namespace pcl
using namespace pcl;

static int stack_sizes_impl(const pcl_mlp_stack_t* d, size_t* save_bytes, size_t* fwd_tmp_bytes, size_t* bwd_tmp_bytes);
static int stack_fwd_impl(const pcl_mlp_stack_t* dp);
static int stack_bwd_impl(const pcl_mlp_stack_t* dp);

extern "C" int pcl_mlp_stack_sizes(const pcl_mlp_stack_t* d, size_t* save_bytes, size_t* fwd_tmp_bytes, size_t* bwd_tmp_bytes) {
    PCL_TRY(validate(d, "pcl_mlp_stack_sizes"));
    const PadPlan p = narrow_supported(*d) ? PadPlan{} : pad_plan(*d);
    if (!p.on) return stack_sizes_impl(d, save_bytes, fwd_tmp_bytes, bwd_tmp_bytes);
    const pcl_mlp_stack_t q = padded_desc(*d, p, reinterpret_cast<float*>(uintptr_t(4096)), reinterpret_cast<float*>(uintptr_t(4096)));
    size_t sv = 0, ft = 0, bt = 0;
    PCL_TRY(stack_sizes_impl(&q, &sv, &ft, &bt));
    if (save_bytes) *save_bytes = al256(sv) + p.save_floats * 4;
    if (fwd_tmp_bytes) *fwd_tmp_bytes = ft;
    if (bwd_tmp_bytes) *bwd_tmp_bytes = al256(bt) + p.bwd_floats * 4;
    return PCL_OK;
}

extern "C" int pcl_mlp_stack_fwd_f32(const pcl_mlp_stack_t* dp) {
    PCL_TRY(validate(dp, "pcl_mlp_stack_fwd_f32"));
    const pcl_mlp_stack_t& d = *dp;
    const PadPlan p = narrow_supported(d) ? PadPlan{} : pad_plan(d);
    if (!p.on) return stack_fwd_impl(dp);
    PCL_REQUIRE(d.save && d.tmp, "pcl_mlp_stack_fwd_f32: null save / tmp");
    size_t sv = 0, ft = 0, bt = 0;
    pcl_mlp_stack_t q = padded_desc(d, p, nullptr, nullptr);
    {   // (sizes of the padded stack proper; its parameter pointers are patched below)
        pcl_mlp_stack_t probe = padded_desc(d, p, reinterpret_cast<float*>(uintptr_t(4096)), nullptr);
        PCL_TRY(stack_sizes_impl(&probe, &sv, &ft, &bt));
    }
    if (d.save_bytes < al256(sv) + p.save_floats * 4) return fail(PCL_EWS, "pcl_mlp_stack_fwd_f32: save %zu < %zu", d.save_bytes, al256(sv) + p.save_floats * 4);
    float* sx = reinterpret_cast<float*>(static_cast<char*>(d.save) + al256(sv));
    q = padded_desc(d, p, sx, nullptr);
    q.save_bytes = sv;
    PadTable t = {};
    const int L = d.n_layers;
    for (int l = 0; l < L; ++l) {
        const pcl_stack_layer_t& y = d.layer[l];
        bool ok = true;
        if (p.wpad[l]) ok = t.push(PadItem{y.W, sx + p.w_off[l], d.c[l + 1], d.c[l], d.c[l], p.cp[l], p.cp[l + 1], p.cp[l]}) && ok;
        if (p.opad[l]) {
            float* v = sx + p.v_off[l];
            const int c = d.c[l + 1], cp = p.cp[l + 1];
            // gamma | beta | running mean | running var | bias: five rows of one [5][cp] block; absent ones are copied from gamma (never read)
            const float* srcs[5] = {y.gamma, y.beta, y.running_mean, y.running_var, y.bias};
            for (int k = 0; k < 5; ++k) if (srcs[k]) ok = t.push(PadItem{srcs[k], v + (size_t)k * cp, 1, c, c, cp, 1, cp}) && ok;
        }
        PCL_REQUIRE(ok, "pcl_mlp_stack_fwd_f32: more padded layers than the pad table holds");
    }
    PCL_TRY(launch_pad(t, d.stream));
    PCL_TRY(stack_fwd_impl(&q));
    // the running statistics of the padded layers: their real entries back to the caller's buffers
    PadTable u = {};
    for (int l = 0; l < L; ++l) {
        if (!p.opad[l]) continue;
        const pcl_stack_layer_t& y = d.layer[l];
        float* v = sx + p.v_off[l];
        const int c = d.c[l + 1], cp = p.cp[l + 1];
        bool ok = true;
        if (y.running_mean) ok = u.push(PadItem{v + 2 * (size_t)cp, y.running_mean, 1, c, cp, c, 0, 0}) && ok;
        if (y.running_var) ok = u.push(PadItem{v + 3 * (size_t)cp, y.running_var, 1, c, cp, c, 0, 0}) && ok;
        PCL_REQUIRE(ok, "pcl_mlp_stack_fwd_f32: more padded layers than the pad table holds");
    }
    return launch_pad(u, d.stream);
}

extern "C" int pcl_mlp_stack_bwd_f32(const pcl_mlp_stack_t* dp) {
    PCL_TRY(validate(dp, "pcl_mlp_stack_bwd_f32"));
    const pcl_mlp_stack_t& d = *dp;
    const PadPlan p = narrow_supported(d) ? PadPlan{} : pad_plan(d);
    if (!p.on) return stack_bwd_impl(dp);
    PCL_REQUIRE(d.save && d.tmp, "pcl_mlp_stack_bwd_f32: null save / tmp");
    const int L = d.n_layers;
    for (int l = 0; l < L; ++l)
        PCL_REQUIRE(d.layer[l].dW && d.layer[l].dgamma && d.layer[l].dbeta && (!d.layer[l].bias || d.layer[l].dbias), "pcl_mlp_stack_bwd_f32: layer %d: null gradient output", l);
    size_t sv = 0, ft = 0, bt = 0;
    {
        pcl_mlp_stack_t probe = padded_desc(d, p, reinterpret_cast<float*>(uintptr_t(4096)), reinterpret_cast<float*>(uintptr_t(4096)));
        PCL_TRY(stack_sizes_impl(&probe, &sv, &ft, &bt));
    }
    if (d.save_bytes < al256(sv) + p.save_floats * 4 || d.tmp_bytes < al256(bt) + p.bwd_floats * 4)
        return fail(PCL_EWS, "pcl_mlp_stack_bwd_f32: save %zu < %zu or tmp %zu < %zu", d.save_bytes, al256(sv) + p.save_floats * 4, d.tmp_bytes, al256(bt) + p.bwd_floats * 4);
    float* sx = reinterpret_cast<float*>(static_cast<char*>(d.save) + al256(sv));      // padded parameters: written by the forward
    float* bx = reinterpret_cast<float*>(static_cast<char*>(d.tmp) + al256(bt));
    pcl_mlp_stack_t q = padded_desc(d, p, sx, bx);
    q.save_bytes = sv; q.tmp_bytes = bt;
    PCL_TRY(stack_bwd_impl(&q));
    PadTable u = {};
    for (int l = 0; l < L; ++l) {
        const pcl_stack_layer_t& y = d.layer[l];
        bool ok = true;
        if (p.wpad[l]) ok = u.push(PadItem{bx + p.dw_off[l], y.dW, d.c[l + 1], d.c[l], p.cp[l], d.c[l], 0, 0}) && ok;
        if (p.opad[l]) {
            float* g = bx + p.dv_off[l];
            const int c = d.c[l + 1], cp = p.cp[l + 1];
            ok = u.push(PadItem{g, y.dgamma, 1, c, cp, c, 0, 0}) && ok;
            ok = u.push(PadItem{g + cp, y.dbeta, 1, c, cp, c, 0, 0}) && ok;
            if (y.bias) ok = u.push(PadItem{g + 2 * (size_t)cp, y.dbias, 1, c, cp, c, 0, 0}) && ok;
        }
        PCL_REQUIRE(ok, "pcl_mlp_stack_bwd_f32: more padded layers than the pad table holds");
    }
    return launch_pad(u, d.stream);
}

static int stack_sizes_impl(const pcl_mlp_stack_t* d, size_t* save_bytes, size_t* fwd_tmp_bytes, size_t* bwd_tmp_bytes) {
    PCL_TRY(validate(d, "pcl_mlp_stack_sizes"));
    if (narrow_supported(*d)) {
        if (save_bytes) *save_bytes = narrow_save_bytes();
        if (fwd_tmp_bytes) *fwd_tmp_bytes = narrow_fwd_tmp_bytes();
        if (bwd_tmp_bytes) *bwd_tmp_bytes = narrow_bwd_tmp_bytes();
        return PCL_OK;
    }
    if (save_bytes) *save_bytes = save_layout(*d, nullptr).bytes;
    if (fwd_tmp_bytes) *fwd_tmp_bytes = fwd_tmp(*d, nullptr).bytes;
    if (bwd_tmp_bytes) *bwd_tmp_bytes = bwd_tmp(*d, nullptr).bytes;
    return PCL_OK;
}

extern "C" int pcl_mlp_stack_last(const pcl_mlp_stack_t* d, size_t* y_offset, size_t* scale_offset, size_t* shift_offset) {
    PCL_TRY(validate(d, "pcl_mlp_stack_last"));
    char* const base = reinterpret_cast<char*>(uintptr_t(4096));          // any non-null base: only differences are used
    const SaveLayout s = save_layout(*d, base);
    const int L = d->n_layers;
    if (y_offset) *y_offset = (size_t)(reinterpret_cast<char*>(s.Y[L - 1]) - base);
    if (scale_offset) *scale_offset = (size_t)(reinterpret_cast<char*>(s.vec[L - 1]) - base);
    if (shift_offset) *shift_offset = (size_t)(reinterpret_cast<char*>(s.vec[L - 1] + d->c[L]) - base);
    return PCL_OK;
}

static int stack_fwd_impl(const pcl_mlp_stack_t* dp) {
    PCL_TRY(validate(dp, "pcl_mlp_stack_fwd_f32"));
    const pcl_mlp_stack_t& d = *dp;
    PCL_REQUIRE((d.out || d.defer_act) && d.save && d.tmp, "pcl_mlp_stack_fwd_f32: null out / save / tmp");
    if (narrow_supported(d)) {
        if (d.save_bytes < narrow_save_bytes() || d.tmp_bytes < narrow_fwd_tmp_bytes())
            return fail(PCL_EWS, "pcl_mlp_stack_fwd_f32: save %zu < %zu or tmp %zu < %zu", d.save_bytes, narrow_save_bytes(), d.tmp_bytes, narrow_fwd_tmp_bytes());
        return narrow_fwd(d);
    }
    const SaveLayout s = save_layout(d, d.save);
    const FwdTmp t = fwd_tmp(d, d.tmp);
    if (d.save_bytes < s.bytes || d.tmp_bytes < t.bytes)
        return fail(PCL_EWS, "pcl_mlp_stack_fwd_f32: save %zu < %zu or tmp %zu < %zu", d.save_bytes, s.bytes, d.tmp_bytes, t.bytes);
    const int L = d.n_layers, P = d.P;
    const int G = pooled_groups(d);
    const int32_t* rmeta = d.grouped ? s.row_meta : nullptr;
    const int32_t* nrows = d.grouped ? d.group_off + G : nullptr;
    void* st = d.stream;
    const float* cur = d.x;
    const float *in_scale = nullptr, *in_shift = nullptr;
    for (int l = 0; l < L; ++l) {
        const pcl_stack_layer_t& ly = d.layer[l];
        const int cin = d.c[l], cout = d.c[l + 1];
        float* Y = s.Y[l];
        int rows;
        if (l == 0 && d.grouped) {
            const int off = d.use_xyz ? 3 : 0, ldw = d.c[0];
            const bool inl = grouped_inline(d), wide = grouped_wide(d);
            if (wide) {
                // row metadata first: its kernel also copies the feature columns of W (row stride c0, rows not 16-byte aligned)
                // into the dense matrix the point GEMM reads -- no separate copy launch per step
                PCL_TRY(group_linear_fwd_impl(d.xyz, d.new_xyz, nullptr, d.use_xyz ? ly.W : nullptr, nullptr, nullptr, 0, ldw, d.idx, d.cnt,
                                              d.group_off, d.B, d.N, d.m, d.pool, cout, Y, s.row_meta, s.row_src, s.row_loc, s.row_feat, t.stats,
                                              st, 1, ly.W + off, d.Wf_dense, d.Cf));
                tagf("pt%dx%d", d.Cf, cout);
                // (a conv bias of the folded layer is added here, once per point: every row gathers exactly one Uf row)
                if (d.flush_k) PCL_TRY(pcl_frag_linear_fwd_f32(d.feature, d.Cf, d.Wf_dense, d.Cf, ly.bias, nullptr, nullptr, 0.f, d.B * d.N, d.Cf, cout, t.Uf, cout,
                                                               nullptr, nullptr, d.flush_k, st));
                else PCL_TRY(pcl_linear_fwd_rows_f32(d.feature, d.Wf_dense, ly.bias, nullptr, nullptr, 0.f, d.B * d.N, d.Cf, cout, t.Uf,
                                                     t.pt_stats, nullptr, nullptr, st));
            }
            if (gather_scatter(d))         // the rows by source point, for the backward (row_src is complete after the metadata launch)
                PCL_TRY(pcl_group_rows_transpose_i32(s.row_src, d.group_off, d.B, d.N, d.m, d.pool, s.in_off, s.in_rows, st));
            tagf("glin%d", cout, 0);
            PCL_TRY(group_linear_fwd_impl(d.xyz, d.new_xyz, wide ? t.Uf : nullptr, d.use_xyz ? ly.W : nullptr, inl ? d.feature : nullptr,
                                          inl ? ly.W + off : nullptr, inl ? d.Cf : 0, ldw, d.idx, d.cnt, d.group_off, d.B, d.N, d.m,
                                          d.pool, cout, Y, s.row_meta, s.row_src, s.row_loc, s.row_feat, t.stats, st, wide ? 2 : 3, nullptr,
                                          nullptr, 0));
            rows = pcl_group_linear_stat_rows(d.B, d.m);
        } else if (l == L - 1 && use_gmax(d)) {
            tagf("fwd%dx%d", cin, cout);
            PCL_TRY(pcl_linear_fwd_gmax_f32(cur, ly.W, ly.bias, in_scale, in_shift, d.slope, P, cin, cout, d.pool, Y, t.stats, t.gmax,
                                            t.gmin, t.gamax, t.gamin, st));
            rows = pcl_mlp_stat_rows(P, cout, 0);
        } else if (frag_fwd(d)) {
            tagf("fwd%dx%d", cin, cout);
            PCL_TRY(pcl_frag_linear_fwd_f32(cur, cin, ly.W, cin, ly.bias, in_scale, in_shift, d.slope, P, cin, cout, Y, cout, nullptr, t.stats, d.flush_k, st));
            rows = pcl_frag_stat_rows(P);
        } else {
            tagf("fwd%dx%d", cin, cout);
            PCL_TRY(pcl_linear_fwd_rows_f32(cur, ly.W, ly.bias, in_scale, in_shift, d.slope, P, cin, cout, Y, t.stats, rmeta, nrows, st));
            rows = pcl_mlp_stat_rows(P, cout, d.grouped ? 2 : 0);
        }
        float* v = s.vec[l];
        PCL_TRY(pcl_bn_finalize_f32(t.stats, rows, ly.gamma, ly.beta, P, cout, d.eps, d.momentum, v, v + cout, v + 2 * cout, v + 3 * cout,
                                    ly.running_mean, ly.running_var, st));
        cur = Y; in_scale = v; in_shift = v + cout;
    }
    const int cl = d.c[L];
    if (G && !d.defer_act) {
        if (use_gmax(d)) PCL_TRY(pcl_group_minmax_finalize_f32(t.gmax, t.gmin, t.gamax, t.gamin, in_scale, in_shift, d.out_slope, G, cl, d.out, s.arg, s.ymax, st));
        else if (d.grouped) { tagf("maxrows%d", cl, 0); PCL_TRY(pcl_bn_act_max_rows_f32(cur, d.group_off, in_scale, in_shift, d.out_slope, G, cl, d.out, s.arg, s.ymax, st)); }
        else { tagf("max%d", cl, 0); PCL_TRY(pcl_bn_act_max_f32(cur, in_scale, in_shift, d.out_slope, G, d.pool, cl, d.out, s.arg, s.ymax, st)); }
    } else if (!d.defer_act) {
        PCL_TRY(pcl_bn_act_f32(cur, in_scale, in_shift, d.out_slope, P, cl, d.out, st));
    }                                   // defer_act: the consumer forms lrelu(scale*y + shift) itself while loading Y[L-1]
    set_launch_tag("");
    return PCL_OK;
}

static int stack_bwd_impl(const pcl_mlp_stack_t* dp) {
    PCL_TRY(validate(dp, "pcl_mlp_stack_bwd_f32"));
    const pcl_mlp_stack_t& d = *dp;
    PCL_REQUIRE((d.out || d.defer_act) && d.save && d.tmp && d.gout, "pcl_mlp_stack_bwd_f32: null out / save / tmp / gout");
    PCL_REQUIRE(d.gout_ld == 0 || d.gout_ld == d.c[d.n_layers] || (d.gout_ld > d.c[d.n_layers] && !d.defer_act && pooled_groups(d) > 0),
                "pcl_mlp_stack_bwd_f32: gout_ld %d (a row stride for gout is taken by pooled stacks only, and is at least the width %d)", d.gout_ld, d.c[d.n_layers]);
    PCL_REQUIRE(!d.defer_act || (d.ext_stats && d.ext_stat_rows >= 1), "pcl_mlp_stack_bwd_f32: defer_act needs ext_stats / ext_stat_rows");
    if (narrow_supported(d)) {
        if (d.save_bytes < narrow_save_bytes() || d.tmp_bytes < narrow_bwd_tmp_bytes())
            return fail(PCL_EWS, "pcl_mlp_stack_bwd_f32: save %zu < %zu or tmp %zu < %zu", d.save_bytes, narrow_save_bytes(), d.tmp_bytes, narrow_bwd_tmp_bytes());
        for (int l = 0; l < d.n_layers; ++l)
            PCL_REQUIRE(d.layer[l].dW && d.layer[l].dgamma && d.layer[l].dbeta && (!d.layer[l].bias || d.layer[l].dbias),
                        "pcl_mlp_stack_bwd_f32: layer %d: null gradient output", l);
        return narrow_bwd(d);
    }
    const SaveLayout s = save_layout(d, d.save);
    const BwdTmp t = bwd_tmp(d, d.tmp);
    if (d.save_bytes < s.bytes || d.tmp_bytes < t.bytes)
        return fail(PCL_EWS, "pcl_mlp_stack_bwd_f32: save %zu < %zu or tmp %zu < %zu", d.save_bytes, s.bytes, d.tmp_bytes, t.bytes);
    const int L = d.n_layers, P = d.P;
    const int G = pooled_groups(d);
    const int ns = d.pool ? d.pool : 1;
    for (int l = 0; l < L; ++l)
        PCL_REQUIRE(d.layer[l].dW && d.layer[l].dgamma && d.layer[l].dbeta && (!d.layer[l].bias || d.layer[l].dbias),
                    "pcl_mlp_stack_bwd_f32: layer %d: null gradient output", l);
    const int32_t* rmeta = d.grouped ? s.row_meta : nullptr;
    const int32_t* nrows = d.grouped ? d.group_off + G : nullptr;
    void* st = d.stream;
    const int cl = d.c[L];
    int rows = 0, cur_stats = 0, cur_du = 0, cur_c = 0;
    const float* dU = nullptr;          // dense gradient w.r.t. the current layer's BatchNorm output (null: sparse (arg, gz))
    const double* ext_stats = nullptr;  // defer_act: the first layer's sums come from the consumer
    bool sparse;
    const float* vL = s.vec[L - 1];
    const bool sd = side_dw(d) && !d.defer_act;
    SideCtx* const sc = sd ? side_ctx() : nullptr;
    if (sd && !sc) return fail(PCL_EHIP, "pcl_mlp_stack_bwd_f32: could not create the side stream / events");
    // join on every way out once forked: the caller's stream continues after the last weight gradient (the caller may release tmp /
    // save right after this call, also when it failed half way)
    struct Join {
        SideCtx* sc; hipStream_t main; bool forked;
        int now() {
            if (!forked) return PCL_OK;
            forked = false;
            if (hipEventRecord(sc->join, sc->st) != hipSuccess || hipStreamWaitEvent(main, sc->join, 0) != hipSuccess)
                return fail(PCL_EHIP, "pcl_mlp_stack_bwd_f32: join of the side stream failed");
            return PCL_OK;
        }
        ~Join() { (void)now(); }
    } join = {sc, reinterpret_cast<hipStream_t>(st), false};
    if (d.defer_act) {
        // the consumer already formed du (masked by the activation) and its two channel sums
        dU = d.gout; ext_stats = d.ext_stats; rows = d.ext_stat_rows; cur_du = 0; sparse = false;
        if (t.dUf) {                    // (grouped, wide features: what the max-gradient launch does on the side in the pooled case)
            const size_t nz4 = gather_scatter(d) ? 0 : ((size_t)d.B * d.N * d.c[1] + 3) / 4;      // (the gather WRITES dUf: only the unit constants then)
            int blocks = (int)((nz4 + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(aux_fill_kernel, dim3(blocks), dim3(256), 0, as_stream(st), reinterpret_cast<float4*>(t.dUf), nz4, t.unit, d.c[1], 2 * d.c[1]);
            PCL_TRY(check_launch("pcl_mlp_stack_bwd_f32(aux)"));
        }
    } else if (G) {
        // side jobs of this first launch (grouped stacks with wide features): clear the target of the folded layer's atomics and
        // write the unit constants of its point GEMMs -- a memset and a fill launch less
        const size_t nz = (t.dUf && !gather_scatter(d)) ? ((size_t)d.B * d.N * d.c[1] + 3) / 4 * 4 : 0;      // (the gather WRITES dUf)
        PCL_TRY(maxgrad_prep_impl(d.gout, d.out, s.ymax, d.out_slope, G, cl, t.gz, t.stats[0], &rows, st, t.dUf, nz, t.unit, t.unit ? d.c[1] : 0,
                                  t.unit ? 2 * d.c[1] : 0, d.gout_ld ? d.gout_ld : cl));
        sparse = true;
    } else {
        float* du0 = sd ? t.dU_l[L - 1] : t.dU[0];
        PCL_TRY(pcl_bn_act_bwd_f32(d.gout, s.Y[L - 1], vL, vL + cl, d.out_slope, P, cl, du0, t.stats[0], &rows, st));
        dU = du0; cur_du = 1; sparse = false;
    }
    bool have_pre = false;              // constants of layer l already computed by the fused finish of layer l + 1
    for (int l = L - 1; l >= 0; --l) {
        const pcl_stack_layer_t& ly = d.layer[l];
        const int cin = d.c[l], cout = d.c[l + 1];
        const float* v = s.vec[l];
        const float *mean = v + 2 * cout, *invstd = v + 3 * cout;
        float* k = sd ? t.consts_l[l] : t.consts[cur_c];
        float *a = k, *k1 = k + cout, *k2 = k + 2 * cout;
        const bool fr = fewrow_layer(d, l) && !have_pre;
        if (fr)             // the constants AND dy = a du - k1 - k2 (y - mean), formed once for both GEMMs of the layer
            PCL_TRY(pcl_bn_bwd_dy_f32(ext_stats ? ext_stats : t.stats[cur_stats], rows, ly.gamma, mean, invstd, P, cout, ly.dgamma, ly.dbeta, a, k1, k2,
                                      ly.dbias, sparse ? nullptr : dU, s.Y[l], sparse ? s.arg : nullptr, sparse ? t.gz : nullptr, ns, P, t.dy_l[l], st));
        else if (!have_pre)
            PCL_TRY(pcl_bn_bwd_consts_f32(ext_stats ? ext_stats : t.stats[cur_stats], rows, ly.gamma, mean, invstd, P, cout, ly.dgamma, ly.dbeta, a, k1,
                                          k2, ly.dbias, st));
        if (ext_stats) { ext_stats = nullptr; cur_stats = 1 - cur_stats; }      // (the layer below writes stats[1 - cur_stats]: keep the ping-pong consistent)
        have_pre = false;
        if (l == 0 && d.grouped) {
            // the folded first layer: dy = a*du - w*(k1 + k2*(y - mean)) per distinct row, scattered to the points / summed into dWx
            const int off = d.use_xyz ? 3 : 0, fan_in = d.c[0], C1 = cout;
            const bool inl = grouped_inline(d), wide = grouped_wide(d);
            // (the point GEMMs as a pair + finish: the finish launch also sums the coordinate-weight partials -- one launch fewer)
            const bool pt_pair = wide && d.need_dx && pcl_linear_bwd_pair_supported(d.B * d.N, C1, d.Cf, 0);
            const bool x_in_finish = pt_pair && gather_scatter(d) && t.dWxp;
            tagf("glinbwd%d", C1, 0);
            if (gather_scatter(d))
                PCL_TRY(pcl_group_linear_bwd_gather_f32(s.row_loc, dU, s.Y[0], a, k1, k2, mean, s.in_off, s.in_rows, d.B, d.N, C1, t.dUf, t.dWxp,
                                                        (t.dWxp && !x_in_finish) ? ly.dW : nullptr, fan_in, st));
            else
            PCL_TRY(group_linear_bwd_impl(s.row_loc, s.row_feat, inl ? d.Cf : 0, dU, s.Y[0], a, k1, k2, mean, s.row_src, nrows, d.B, d.N, C1,
                                          t.dUf, t.dWxp, t.dWfp, (t.dWxp || t.dWfp) ? ly.dW : nullptr, fan_in, off, st, /*duf_is_zero=*/true));
            if (wide) {
                // plain GEMMs through the BatchNorm-backward entry points with a = 1, k1 = k2 = 0 (dy == dUf): constants in consts[1 - cur_c]
                float* one = t.unit;                 // written by the max-gradient kernel at the head of this call
                float* zero = one + C1;
                const int Pp = d.B * d.N;
                if (pt_pair) {
                    // both point GEMMs from one launch, the tile sum in the next (see the few-row layers below)
                    PCL_REQUIRE(d.dx, "pcl_mlp_stack_bwd_f32: need_dx without dx");
                    tagf("ptpair%dx%d", C1, d.Cf);
                    PCL_TRY(pcl_linear_bwd_pair_f32(t.dUf, t.dUf, one, zero, zero, zero, nullptr, nullptr, 1, d.Wf_dense, d.feature, nullptr, nullptr, 0.f, 0, Pp,
                                                    C1, d.Cf, d.dx, nullptr, 0, t.ptws, t.ptws_bytes, st));
                    PCL_TRY(pair_finish_impl(t.ptws, t.ptws_bytes, Pp, C1, d.Cf, ly.dW + off, fan_in, nullptr, 0, nullptr, nullptr, nullptr, 0,
                                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st,
                                             x_in_finish ? t.dWxp : nullptr, pcl_group_linear_stat_rows(d.B, d.m), C1, fan_in, ly.dW));
                    break;
                }
                tagf("ptdw%dx%d", C1, d.Cf);
                PCL_TRY(pcl_linear_bwd_dw_rows_f32(t.dUf, t.dUf, one, zero, zero, zero, nullptr, nullptr, 1, d.feature, nullptr, nullptr, 0.f, Pp,
                                                   C1, d.Cf, ly.dW + off, t.ptws, t.ptws_bytes, nullptr, nullptr, fan_in, st));
                if (d.need_dx) {
                    PCL_REQUIRE(d.dx, "pcl_mlp_stack_bwd_f32: need_dx without dx");
                    tagf("ptdx%dx%d", C1, d.Cf);
                    PCL_TRY(pcl_linear_bwd_dx_rows_f32(t.dUf, t.dUf, one, zero, zero, zero, nullptr, nullptr, 1, d.Wf_dense, Pp, C1, d.Cf, nullptr,
                                                       nullptr, nullptr, 0.f, d.dx, nullptr, nullptr, nullptr, 0, 0, st));
                }
            }
            break;
        }
        const float* Xprev = l > 0 ? s.Y[l - 1] : d.x;
        const float* psc = l > 0 ? s.vec[l - 1] : nullptr;
        const float* psh = l > 0 ? s.vec[l - 1] + cin : nullptr;
        const float* Yl = s.Y[l];
        if (l > 0 && fused_bwd_enabled() && pcl_linear_bwd_fused_supported(cout, cin)) {
            // one pass forms dy once and produces BOTH the layer below's du (+ its BatchNorm-backward sums) and dW; the second
            // launch sums the partial tiles and turns those sums into the constants of layer l - 1
            float* dUp = t.dU[cur_du];
            double* stn = t.stats[1 - cur_stats];
            tagf("fb%dx%d", cout, cin);
            PCL_TRY(pcl_linear_bwd_fused_rows_f32(dU, Yl, a, k1, k2, mean, sparse ? s.arg : nullptr, sparse ? t.gz : nullptr, ns, ly.W, P, cout, cin,
                                                  Xprev, psc, psh, d.slope, dUp, stn, t.ws, t.ws_bytes, rmeta, nrows, st));
            const pcl_stack_layer_t& lp = d.layer[l - 1];
            float* kp = t.consts[1 - cur_c];
            const float* vp = s.vec[l - 1];
            PCL_TRY(pcl_linear_bwd_fused_finish_f32(t.ws, t.ws_bytes, P, cout, cin, ly.dW, stn, lp.gamma, vp + 2 * cin, vp + 3 * cin, P, lp.dgamma,
                                                    lp.dbeta, kp, kp + cin, kp + 2 * cin, lp.dbias, st));
            have_pre = true;
            dU = dUp; sparse = false; cur_du = 1 - cur_du; cur_stats = 1 - cur_stats; cur_c = 1 - cur_c;
            rows = pcl_linear_bwd_fused_stat_rows(P, cin);
            continue;
        }
        if (!sd && !fr && !rmeta && (l > 0 || d.need_dx) && pcl_linear_bwd_pair_supported(P, cout, cin, l == 0 ? d.x_grad_from : 0)) {
            // few-row layer on plain rows (the GroupAll level, decoders): dW's partial tiles and dX from ONE launch; the second launch sums
            // the tiles and -- the layer below's sums being complete by then -- forms ITS constants as well (round 6: four launches per
            // layer were consts, dW, reduce, dX).  Same kernels' arithmetic: bit-identical to the per-kernel path.
            float* dUp = l > 0 ? t.dU[cur_du] : d.dx;
            PCL_REQUIRE(dUp, "pcl_mlp_stack_bwd_f32: need_dx without dx");
            double* stn = l > 0 ? t.stats[1 - cur_stats] : nullptr;
            tagf("pair%dx%d", cout, cin);
            PCL_TRY(pcl_linear_bwd_pair_f32(dU, Yl, a, k1, k2, mean, sparse ? s.arg : nullptr, sparse ? t.gz : nullptr, ns, ly.W, Xprev, psc, psh, d.slope,
                                            l > 0 ? 1 : 0, P, cout, cin, dUp, stn, l == 0 ? d.x_grad_from : 0, t.ws, t.ws_bytes, st));
            if (l > 0) {
                const pcl_stack_layer_t& lp = d.layer[l - 1];
                float* kp = t.consts[1 - cur_c];
                const float* vp = s.vec[l - 1];
                const int rows_below = pcl_mlp_stat_rows(P, cin, 1);
                PCL_TRY(pcl_linear_bwd_pair_finish_f32(t.ws, t.ws_bytes, P, cout, cin, ly.dW, 0, stn, rows_below, lp.gamma, vp + 2 * cin, vp + 3 * cin, P,
                                                       lp.dgamma, lp.dbeta, kp, kp + cin, kp + 2 * cin, lp.dbias, st));
                have_pre = true;
                rows = rows_below;
                dU = dUp; sparse = false; cur_du = 1 - cur_du; cur_stats = 1 - cur_stats; cur_c = 1 - cur_c;
            } else
                PCL_TRY(pcl_linear_bwd_pair_finish_f32(t.ws, t.ws_bytes, P, cout, cin, ly.dW, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                                       nullptr, nullptr, nullptr, nullptr, st));
            continue;
        }
        void* dw_st = st;
        if (sd) {
            // fork: everything the dW launch reads is complete at this point of the caller's stream (consts(l) was the last to be written)
            if (hipEventRecord(sc->fork, as_stream(st)) != hipSuccess || hipStreamWaitEvent(sc->st, sc->fork, 0) != hipSuccess)
                return fail(PCL_EHIP, "pcl_mlp_stack_bwd_f32: fork to the side stream failed");
            dw_st = sc->st; join.forked = true;
        }
        tagf("dw%dx%d", cout, cin);
        if (fr) {
            PCL_TRY(pcl_linear_bwd_dw_plain_f32(t.dy_l[l], Xprev, psc, psh, d.slope, P, cout, cin, ly.dW, t.ws, t.ws_bytes, 0, dw_st));
            if (l > 0 || d.need_dx) {
                float* dUp = l > 0 ? (sd ? t.dU_l[l - 1] : t.dU[cur_du]) : d.dx;
                PCL_REQUIRE(dUp, "pcl_mlp_stack_bwd_f32: need_dx without dx");
                double* stn = l > 0 ? t.stats[1 - cur_stats] : nullptr;
                tagf("dx%dx%d", cout, cin);
                PCL_TRY(pcl_frag_linear_bwd_dx_f32(t.dy_l[l], ly.W, cin, P, cout, cin, l > 0 ? Xprev : nullptr, cin, l > 0 ? psc : nullptr, l > 0 ? psh : nullptr,
                                                   d.slope, dUp, cin, stn, l == 0 ? d.x_grad_from : 0, st));
                if (l > 0) {
                    rows = frag_stat_rows(P);
                    dU = dUp; sparse = false; cur_du = 1 - cur_du; cur_stats = 1 - cur_stats; cur_c = 1 - cur_c;
                }
            }
            continue;
        }
        PCL_TRY(pcl_linear_bwd_dw_rows_f32(dU, Yl, a, k1, k2, mean, sparse ? s.arg : nullptr, sparse ? t.gz : nullptr, ns, Xprev, psc, psh, d.slope,
                                           P, cout, cin, ly.dW, t.ws, t.ws_bytes, rmeta, nrows, 0, dw_st));
        if (l > 0 || d.need_dx) {
            float* dUp = l > 0 ? (sd ? t.dU_l[l - 1] : t.dU[cur_du]) : d.dx;
            PCL_REQUIRE(dUp, "pcl_mlp_stack_bwd_f32: need_dx without dx");
            double* stn = l > 0 ? t.stats[1 - cur_stats] : nullptr;
            tagf("dx%dx%d", cout, cin);
            PCL_TRY(pcl_linear_bwd_dx_rows_f32(dU, Yl, a, k1, k2, mean, sparse ? s.arg : nullptr, sparse ? t.gz : nullptr, ns, ly.W, P, cout, cin,
                                               l > 0 ? Xprev : nullptr, psc, psh, d.slope, dUp, stn, rmeta, nrows, l == 0 ? d.x_grad_from : 0, 0, st));
            if (l > 0) {
                rows = pcl_mlp_stat_rows(P, cin, 1 | (rmeta ? 2 : 0));
                dU = dUp; sparse = false; cur_du = 1 - cur_du; cur_stats = 1 - cur_stats; cur_c = 1 - cur_c;
            }
        }
    }
    PCL_TRY(join.now());
    set_launch_tag("");
    return PCL_OK;
}

extern "C" void pcl_set_stack_overlap(int side_dw_on, int max_rows) {
    if (side_dw_on >= 0) g_side_dw = side_dw_on != 0;
    if (max_rows > 0) g_side_dw_max_rows = max_rows;
}
extern "C" int pcl_get_stack_overlap(void) { return g_side_dw; }
extern "C" void pcl_set_fewrow_backward(int on) { if (on >= 0) g_fewrow = on != 0; }
extern "C" int pcl_get_fewrow_backward(void) { return g_fewrow; }
/* does layer (Cout <- Cin) of a plain stack on P rows take the few-row backward (constants + dy in one launch, dX and dW on the formed dy)?
 * The per-kernel host path asks the same question so that both paths launch the same kernels. */
extern "C" int pcl_mlp_fewrow_layer(int P, int Cout, int Cin, int first_layer) {
    if (!g_fewrow || !frag_rows_eligible(P)) return 0;
    if (!first_layer && fused_bwd_enabled() && pcl_linear_bwd_fused_supported(Cout, Cin)) return 0;
    return pcl_bn_bwd_dy_supported(P, Cout);
}
