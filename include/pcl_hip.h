/*
 * pcl_hip.h -- C ABI of libpcl_hip.so: the MI355X (gfx950) implementation of the
 * point-cloud hot path of Jittor/PointCloudLib.
 *
 * Boundary replaced: the reference reaches its three native kernels through Jittor's inline-op
 * FFI  jt.code(out_shapes, out_dtypes, inputs, cuda_src=...)  (misc/ops.py:278, :376-381,
 * :656-662), which hands the CUDA text raw device pointers (in0_p, out0_p, ...) and shapes
 * (in0_shape0, ...).  The entry points below are what a `jt.code`-style binding (or any FFI:
 * ctypes, cgo, JNI) would call instead: plain device pointers + sizes + a HIP stream, no
 * framework types.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless said otherwise; fp32 data, int32 indices,
 *    row-major, channel-last ([B,N,3], [B,N,C]) exactly as the reference's modules pass them;
 *  - the caller owns all buffers; the library never allocates, frees or synchronises; work is
 *    enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *  - every output element is defined on return (the reference leaves some rows uninitialised);
 *  - return value: PCL_OK (0) or a negative PCL_E* code; pcl_last_error() gives a thread-local
 *    message for the last failure.
 */
#ifndef PCL_HIP_H
#define PCL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCL_OK 0
#define PCL_EINVAL (-1)   /* bad argument (null pointer, size out of range, ...) */
#define PCL_EHIP (-2)     /* HIP runtime / launch error                           */
#define PCL_ENOSUP (-3)   /* valid request this build cannot serve (size limits)  */
#define PCL_EWS (-4)      /* workspace too small                                  */

int pcl_version(void);
/* As pcl_time_next_launch, for the next GEMM-family launch whose launch tag equals `tag` ("fb256x128", "fwd128x256", ...:
 * the per-stack entry points launch many such kernels from one call).  No reference counterpart (measurement hook). */
void pcl_time_tagged_launch(void* start_event, void* stop_event, const char* tag);
/* Measurement hook: arm two hipEvent_t (created with timing enabled) for the NEXT GEMM-family kernel this thread launches
 * (linear forward / dX / dW / fused backward); they receive that kernel's own begin and end timestamps.  No reference
 * counterpart (bench.py's roofline leg). */
void pcl_time_next_launch(void* start_event, void* stop_event);
/* Measurement hook: the source-level name of the GEMM-family / k-NN kernel this thread launched last ("(linear_nt_kernel<AM, EM,
 * true, 2, GM, RAG>)", ...): lets the profiling tools match an entry point's launch with the kernel name a rocprofv3 trace
 * shows (tools/pmc_traffic.py).  Static storage; "" before the first launch.  No reference counterpart. */
const char* pcl_last_launch_kernel(void);
/* reference: no reference counterpart: the reference has no error channel (CUDA errors surface at the next Jittor sync, misc/ops.py:269-271 are Python asserts) */
const char* pcl_last_error(void);

/* misc/ops.py:110-111  optimal_block(): 2 ** int(ln(batch_size)).  The reference launches FPS and
 * ball query with this many threads per block; for FPS it decides how exact distance ties are
 * broken, so callers that want reference-identical sampling pass it as `tie_stride`. */
int pcl_optimal_block(int batch_size);

/* test / tuning hook (process-wide, set between calls; no reference counterpart): threads per cloud of pcl_fps_f32 (0 = chosen from N;
 * one of 64, 128, 256, 512, 1024 -- any count gives the same indices) and the hardware issue priority of its waves (0..3, s_setprio;
 * default 3: inline in a forward pass the serial chain is the critical path; 0 when the sampling of the NEXT batch runs on a side
 * stream beside kernels the step waits for). */
void pcl_set_fps_tuning(int threads_per_cloud, int issue_priority);
/* ---- farthest point sampling -----------------------------------------------------------------
 * Replaces FurthestPointSampler's jt.code kernel, misc/ops.py:124-234 (launch :236-251), and the
 * gather of :280-284.
 *   xyz [B,N,3] -> idx_out [B,m] int32, new_xyz_out [B,m,3] (nullable).
 *   tie_stride     power of two in [1,512]: exact ties go to the smallest
 *                  (bitreverse_{log2 S}(k mod S), k) -- the order the reference's S-thread block
 *                  reduction produces (S = pcl_optimal_block(B)); 1 = lowest index wins.
 *   skip_sqnorm_le points with (double)(x*x+y*y+z*z) <= this are never sampled and never updated
 *                  (misc/ops.py:162-163 uses 1e-3, a double literal); negative disables the rule
 *                  (misc/pointconv_utils.py:74-116 has none).
 *   start_idx      [B] first sample per cloud, nullable = 0 (misc/ops.py:143; pointconv draws it
 *                  at random, pointconv_utils.py:88).
 * Requires 1 <= m <= N. */
int pcl_fps_f32(const float* xyz, int B, int N, int m, int tie_stride, double skip_sqnorm_le,
                const int32_t* start_idx, int32_t* idx_out, float* new_xyz_out, void* stream);

/* ---- ball query -------------------------------------------------------------------------------
 * Replaces query_ball_point_kernel, misc/ops.py:291-330 (launch :332-337).
 *   new_xyz [B,m,3], xyz [B,N,3] -> idx_out [B,m,nsample], cnt_out [B,m] (nullable).
 * First `nsample` indices k (ascending) with d2 < fl(radius*radius), padded with the first hit;
 * rows without any hit are zero-filled with cnt 0 (undefined in the reference). */
int pcl_ball_query_f32(const float* new_xyz, const float* xyz, int B, int m, int N, float radius,
                       int nsample, int32_t* idx_out, int32_t* cnt_out, void* stream);
/* n_radii (1..4) ball queries around the SAME centres in one scan of the cloud (round 5): replaces the per-scale BallQueryGrouper calls of
 * PointnetModuleMSG (networks/seg/pointnet2_partseg.py:93-103, :39-41 of the base class' loop over groupers; networks/cls/pointnet2.py:83-93).
 * radii / nsamples / idx_out / cnt_out are HOST arrays of n_radii entries (cnt_out or any of its entries may be NULL); idx_out[r] is
 * [B,m,nsamples[r]], cnt_out[r] [B,m] on the device.  Every list is identical to pcl_ball_query_f32's for that radius. */
int pcl_ball_query_multi_f32(const float* new_xyz, const float* xyz, int B, int m, int N, int n_radii, const float* radii,
                             const int32_t* nsamples, int32_t* const* idx_out, int32_t* const* cnt_out, void* stream);

/* ---- grouping ---------------------------------------------------------------------------------
 * Replaces the three Var.reindex gathers + subtract + concat of BallQueryGrouper.execute,
 * misc/ops.py:383-407.   out [B,m,ns,D], D = (use_xyz?3:0)+C, channel order [xyz-new_xyz, feat].
 * feat nullable when C == 0.  pcl_group_bwd_f32 is the gradient w.r.t. feat (scatter-add, the
 * gradient of reindex); it zero-fills gfeat [B,N,C] itself. */
int pcl_group_f32(const float* xyz, const float* new_xyz, const float* feat, const int32_t* idx,
                  int B, int N, int m, int ns, int C, int use_xyz, float* out, void* stream);
/* reference: gradient of the Var.reindex gathers of misc/ops.py:384-396 (Jittor derives it as reindex_reduce add) */
int pcl_group_bwd_f32(const float* gout, const int32_t* idx, int B, int N, int m, int ns, int C,
                      int use_xyz, float* gfeat, void* stream);
/* GroupAll.execute, misc/ops.py:415-419: out [B,1,N,D] = concat(xyz, feat), xyz not re-centred. */
int pcl_group_all_f32(const float* xyz, const float* feat, int B, int N, int C, int use_xyz,
                      float* out, void* stream);
/* reference: gradient of GroupAll's concat, misc/ops.py:415-419 */
int pcl_group_all_bwd_f32(const float* gout, int B, int N, int C, int use_xyz, float* gfeat,
                          void* stream);
/* Plain row gather out[b,i,:] = src[b,idx[b,i],:]  (index_points, misc/ops.py:12-27) and its
 * scatter-add gradient (zero-fills gsrc). */
int pcl_gather_rows_f32(const float* src, const int32_t* idx, int B, int N, int M, int C, float* out,
                        void* stream);
/* reference: gradient of index_points, misc/ops.py:12-27 */
int pcl_gather_rows_bwd_f32(const float* gout, const int32_t* idx, int B, int N, int M, int C,
                            float* gsrc, void* stream);

/* DGCNN edge features, get_graph_feature (networks/cls/dgcnn.py:29-50):
 *   x [B,N,C] channel-last, idx [B,N,k] -> out [B,N,k,2C] = concat(x[idx]-x[n], x[n]);  bwd defines all of gx. */
int pcl_edge_feature_f32(const float* x, const int32_t* idx, int B, int N, int k, int C, float* out, void* stream);
/* reference: gradient of get_graph_feature, networks/cls/dgcnn.py:29-50 */
int pcl_edge_feature_bwd_f32(const float* gout, const int32_t* idx, int B, int N, int k, int C, float* gx,
                             void* stream);

/* ---- brute-force k-NN -------------------------------------------------------------------------
 * Replaces knn_cuda_global (compute_distances + modified_insertion_sort), misc/ops.py:429-638.
 *   ref [B,C,Nr] (the reference's in0 = x_r), qry [B,C,Nq] (in1 = x_q), channel-major
 *   -> idx_out [B,k,Nq] int32: the k nearest refs of each query, ascending by
 *   (sum_c (ref-qry)^2 accumulated in ascending c, index).
 * The reference takes an uninitialised [B,Nr,Nq] scratch as a third input (misc/ops.py:655); here
 * the caller provides `workspace` of at least pcl_knn_workspace_bytes(...) bytes. */
size_t pcl_knn_workspace_bytes(int B, int C, int Nr, int Nq, int k);
/* reference: replaces compute_distances + modified_insertion_sort + the host glue, misc/ops.py:429-552, :562-663 */
int pcl_knn_f32(const float* ref, const float* qry, int B, int C, int Nr, int Nq, int k,
                int32_t* idx_out, void* workspace, size_t workspace_bytes, void* stream);
/* pcl_knn_f32 with the lists written as [B, Nq, k] rows (the layout the EdgeConv gathers read; the reference permutes KNN's [B, k, Nq] right
 * after the call, networks/cls/dgcnn.py:34-35): same search, same order within a list.  Fused kernel only: pcl_knn_nk_supported(Nr). */
int pcl_knn_nk_supported(int Nr);
int pcl_knn_nk_f32(const float* ref, const float* qry, int B, int C, int Nr, int Nq, int k, int32_t* idx_out, void* stream);
/* The NAMED SECOND DEFINITION of the same search: distances accumulated as ssd = fma(tmp, tmp, ssd) -- what nvcc's default
 * -fmad=true makes of `ssd += tmp*tmp` (misc/ops.py:488-491) -- instead of a separately rounded product and sum.  2 VALU
 * operations per (query, reference, channel) instead of 3.  Bit-exact against the oracle's "fma" reading
 * (oracle.contract("fma")); on the BASELINE inputs it changes no neighbour SET and 1.2e-4 of the ordered lists
 * (profiles/r03_contraction_sensitivity.txt).  Opt-in (PCL_KNN_CONTRACT=fma on the Python side); workspace as pcl_knn_f32
 * (0 bytes up to 4096 references). */
int pcl_knn_fma_f32(const float* ref, const float* qry, int B, int C, int Nr, int Nq, int k,
                    int32_t* idx_out, void* workspace, size_t workspace_bytes, void* stream);
/* PointConv's knn_point in the reference's own arithmetic, a NAMED SECOND DEFINITION like pcl_knn_fma_f32 (round 4): squared
 * distances in matmul form, -2 (src . dst) + |src|^2 + |dst|^2 with the operation order of misc/pointconv_utils.py:50-52, dot product
 * over c = 0, 1, 2 ascending (fma_dot = 1: fma chain; 0: every product and sum rounded), then the first k of a stable ascending
 * order (distance, index) -- bit for bit oracle/pcl_oracle.c::pclo_knn_point_matmul_f32.  The library's own definition of these
 * groups stays the direct form (pcl_knn_f32 on the coordinates); the two differ on 2.4e-4 / 7.3e-4 of PointConv's ordered lists.
 *   xyz [B,N,3] (16-byte aligned, N a multiple of 4, N <= 4096), new_xyz [B,S,3] -> idx_out [B,S,k] int32.  Opt-in on the Python
 *   side (PCL_KNN_POINT=matmul). */
/* reference: misc/pointconv_utils.py:34-53, :120-131 */
int pcl_knn_point_matmul_f32(const float* xyz, const float* new_xyz, int B, int N, int S, int k, int fma_dot, int32_t* idx_out,
                             void* stream);

/* ---- 3-NN inverse-distance interpolation (PointNetFeaturePropagation, misc/ops.py:83-93) -----
 *   xyz1 [B,N,3] (targets), xyz2 [B,S,3] (sources) -> idx3 [B,N,3], w3 [B,N,3]
 *   (3 nearest by (direct-form d2, index); w = 1/(d2+1e-8) normalised; S==1 -> idx 0, w (1,0,0)).
 *   interp fwd: out[b,n,:] = sum_j w3[b,n,j] * points2[b,idx3[b,n,j],:];  bwd zero-fills gpoints2. */
int pcl_three_nn_f32(const float* xyz1, const float* xyz2, int B, int N, int S, int32_t* idx3,
                     float* w3, void* stream);
/* reference: replaces the weighted index_points sum of PointNetFeaturePropagation, misc/ops.py:90-93 */
int pcl_three_interp_f32(const float* points2, const int32_t* idx3, const float* w3, int B, int N,
                         int S, int D, float* out, void* stream);
/* reference: gradient of misc/ops.py:90-93 w.r.t. points2 */
int pcl_three_interp_bwd_f32(const float* gout, const int32_t* idx3, const float* w3, int B, int N,
                             int S, int D, float* gpoints2, void* stream);

/* ---- Gaussian kernel density (PointConv compute_density, misc/pointconv_utils.py:174-184) ----------------
 *   density[b,i] = mean_j exp(-|x_i-x_j|^2 / (2 bw^2)) / (2.5 bw), without the [B,N,N] matrix (direct-form d2). */
int pcl_density_f32(const float* xyz, int B, int N, float bandwidth, float* density_out, void* stream);

/* First MLP layer folded into the grouping.  The first 1x1 conv of a set-abstraction MLP is linear in the grouped row
 * [xyz_nbr - centre | feat_nbr] (networks/cls/pointnet2.py:18-57, misc/ops.py:383-403):
 *     y[g,s] = Wx (xyz[nbr] - centre[g]) + Uf[nbr],   Uf [B*N, C1] = feat Wf^T  (one GEMM over the points, not the rows)
 * pcl_group_linear_f32 writes that pre-BatchNorm output for the DISTINCT rows of every ball-query group (same row order
 * and metadata as pcl_group_compact_f32) and the multiplicity-weighted BatchNorm sums as pcl_group_linear_stat_rows(B,m)
 * fp64 partial rows [rows][2][C1].  Wx [C1,3] or NULL (use_xyz = 0), Uf or NULL; C1 <= 256.  Narrow point features
 * (CF <= 4 columns, e.g. the normals of the first level) are folded inline instead: feat_small [B*N,CF], Wf_small [C1,CF].
 * pcl_group_linear_bwd_f32: dy = a*du - w*(k1 + k2*(y - mu)) per row (du: gradient w.r.t. the BatchNorm output as the dX
 * GEMM of the next layer leaves it), accumulated into dUf [B*N, C1] (zero-filled here) and into dWx_part
 * [pcl_group_linear_stat_rows][C1][3] / dWf_part [..][C1][CF] partial sums (sum over the first axis = dWx, dWf_small). */
int pcl_group_linear_stat_rows(int B, int m);
/* reference: replaces BallQueryGrouper's gathers (misc/ops.py:383-407) + the first Conv2d 1x1 of build_mlps (networks/cls/pointnet2.py:25-26, execute :51-54) */
int pcl_group_linear_f32(const float* xyz, const float* new_xyz, const float* Uf, const float* Wx, const float* feat_small,
                         const float* Wf_small, int CF, int ldw, const int32_t* idx, const int32_t* cnt, const int32_t* group_off,
                         int B, int N, int m, int ns, int C1, float* Y, int32_t* row_meta, int32_t* row_src,
                         float* row_loc /* [cap,4]: xyz - centre, multiplicity */, float* row_feat /* [cap,4] or NULL */,
                         double* stats_ws, void* stream);
/* reference: gradient of the same pair (misc/ops.py:383-407 + networks/cls/pointnet2.py:25-26) */
int pcl_group_linear_bwd_f32(const float* row_loc, const float* row_feat, int CF, const float* dU, const float* Y,
                             const float* a, const float* k1, const float* k2, const float* mu, const int32_t* row_src,
                             const int32_t* n_rows_dev /* &group_off[B*m] */, int B, int N, int C1, float* dUf,
                             float* dWx_part, float* dWf_part, float* dW0, int ldw, int off, void* stream);
/* The same scatter as a GATHER over the points' row lists (round 5): pcl_group_rows_transpose_i32 turns row_src into
 * in_off [B*N + 1] / in_rows [group_off[B*m]] -- every source point's rows, ascending (one workgroup per cloud; 17 N + 1 + m ns ints in
 * LDS: pcl_group_rows_transpose_supported; a group names a point at most once, as ball query / k-NN groups do) -- and pcl_group_linear_bwd_gather_f32 walks the rows by source point: dUf [B*N][C1]
 * is WRITTEN (no zero-fill, no atomics; sums in list order: run-to-run identical), dWx_part [pcl_group_linear_stat_rows][C1][3] as in
 * pcl_group_linear_bwd_f32, dW0[c][0..2] (leading dimension ldw) optionally reduced in the same call.  C1 in {64, 128, 256}. */
int pcl_group_rows_transpose_supported(int N, int m, int ns);
int pcl_group_rows_transpose_i32(const int32_t* row_src, const int32_t* group_off, int B, int N, int m, int ns, int32_t* in_off,
                                 int32_t* in_rows, void* stream);
int pcl_group_linear_bwd_gather_supported(int C1);
void pcl_set_pointconv_paths(int bwd_w_rows);  /* lab switch: the weight-gradient contraction with the row-major LDS image (1, default) or the transposed one (0) */
void pcl_set_scatter_form(int gather);        /* lab switch: 1 (default) the gather where supported, 0 the fp32-atomic scatter everywhere, < 0 leave as is */
int pcl_group_linear_bwd_gather_f32(const float* row_loc, const float* dU, const float* Y, const float* a, const float* k1,
                                    const float* k2, const float* mu, const int32_t* in_off, const int32_t* in_rows, int B, int N,
                                    int C1, float* dUf, float* dWx_part, float* dW0, int ldw, void* stream);

/* The classification head on R <= 64 rows (one row per cloud): Linear (+bias) -> BatchNorm1d -> (Leaky)ReLU as ONE kernel
 * per layer (networks/cls/pointnet2.py:138-147, dgcnn.py:87-93, pointnet.py:22-38).  X [R,K], W [N,K] (nn.Linear layout).
 * bn_mode: 0 none, 1 training (batch statistics over the R rows; running_mean/var updated in place torch-style, unbiased
 * running variance), 2 evaluation (running statistics); +4: running_var receives the BIASED batch variance, as
 * pcl_bn_finalize_f32 does for the set-abstraction stacks (PointConv's per-point Linear + BatchNorm1d on the GroupAll level,
 * misc/pointconv_utils.py:395-397, is such a layer: 32 rows x 16384 -> 1024; layers with K >= 2048 and R <= 32 run as fp32 MFMA
 * tiles: the forward as split-K partial tiles summed in a fixed order -- its partial sums live in `workspace`
 * (pcl_head_layer_fwd_workspace_bytes(R, K, N) bytes, 16-byte aligned; 0 bytes / NULL for every other shape) -- dW and dX as
 * 32 x 32 tiles).  slope: 1 = no activation, 0 = ReLU, 0.2 = LeakyReLU.
 * Forward keeps Ypre (pre-BatchNorm) and mean/invstd for backward; backward returns dW, dbias/dgamma/dbeta (nullable) and,
 * when dX != NULL, the input gradient (dY_ws: [R,N] scratch). */
int pcl_head_layer_fwd_f32(const float* X, const float* W, const float* bias, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int R, int K, int N, int bn_mode, float eps, float momentum,
                           float slope, float* Ypre, float* OUT, float* mean_out, float* invstd_out, void* workspace,
                           size_t workspace_bytes, void* stream);
size_t pcl_head_layer_fwd_workspace_bytes(int R, int K, int N);
/* reference: gradient of Linear + BatchNorm1d + ReLU of the heads, networks/cls/pointnet2.py:138-147, :155-158 */
int pcl_head_layer_bwd_f32(const float* X, const float* W, const float* dOUT, const float* OUT, const float* Ypre,
                           const float* gamma, const float* mean, const float* invstd, int R, int K, int N, int bn_mode,
                           float slope, float* dY_ws, float* dW, float* dbias, float* dgamma, float* dbeta, float* dX,
                           void* stream);

/* The whole FC head behind one entry point per direction: up to PCL_HEAD_MAX_LAYERS x [Linear (+bias) -> BatchNorm1d ->
 * (Leaky)ReLU -> Dropout(drop_p)] on R <= 64 rows (networks/cls/pointnet2.py:138-147 + :157-158; dgcnn.py:87-93,:117-121;
 * pointnet.py:22-38; pointconv.py:14-33).  Per layer: the kernels of pcl_head_layer_*_f32 with the same bn_mode / slope meaning;
 * drop_p > 0 applies inverted dropout to the layer's OUTPUT (keep decision = counter-based hash of (seed, layer, element), the
 * same in forward and backward -- no mask is stored; pass drop_p = 0 in evaluation mode).  `save` (forward -> backward) and the
 * backward's `tmp` are carved here (pcl_fc_head_sizes).  out [R, N_last]; backward: gout [R, N_last] -> layer[l].dW / dbias /
 * dgamma / dbeta and dx [R, K_0] (nullable). */
#define PCL_HEAD_MAX_LAYERS 4
typedef struct pcl_head_layer_t {
    const float* W;              /* [N, K] (nn.Linear layout) */
    const float* bias;           /* nullable */
    const float* gamma;          /* nullable (no affine) */
    const float* beta;
    float* running_mean;         /* nullable */
    float* running_var;
    float* dW;                   /* backward outputs */
    float* dbias;
    float* dgamma;
    float* dbeta;
    int32_t K, N, bn_mode;       /* bn_mode as pcl_head_layer_fwd_f32 */
    float eps, momentum, slope, drop_p;
    int32_t pad_;
} pcl_head_layer_t;
typedef struct pcl_fc_head_t {
    int32_t struct_bytes;
    int32_t n_layers;
    int32_t R;
    int32_t pad_;
    uint64_t seed;               /* dropout */
    const float* x;              /* [R, K_0] */
    pcl_head_layer_t layer[PCL_HEAD_MAX_LAYERS];
    float* out;
    void* save;
    size_t save_bytes;
    void* tmp;                   /* backward only */
    size_t tmp_bytes;
    const float* gout;
    float* dx;
    void* stream;
} pcl_fc_head_t;
int pcl_fc_head_sizes(const pcl_fc_head_t* desc, size_t* save_bytes, size_t* bwd_tmp_bytes);
/* reference: the fc_layer of PointNet2_cls.execute, networks/cls/pointnet2.py:157-158 (dgcnn.py:117-121, pointnet.py:37-39) */
int pcl_fc_head_fwd_f32(const pcl_fc_head_t* desc);
/* reference: its autograd backward */
int pcl_fc_head_bwd_f32(const pcl_fc_head_t* desc);

/* Label-smoothed cross entropy of the classification drivers, soft_cross_entropy_loss of train_cls.py:31-51:
 *   w = one_hot*(1-eps) + (1-one_hot)*eps/(C-1);  *loss = -mean_r sum_c w[r,c] log_softmax(logits)[r,c]
 * logits [R,C], target [R] int64 class ids; dlogits [R,C] (nullable) = d loss / d logits = (softmax - w)/R.  One launch
 * (one workgroup; R <= 65536) instead of the ~12 elementwise launches of the composite. */
int pcl_soft_ce_f32(const float* logits, const int64_t* target, float eps, int R, int C, float* loss, float* dlogits, void* stream);
/* The same loss over many rows -- the part-segmentation loss nn.cross_entropy_loss(pred [B*N, part_num], seg) of train_partseg.py:116 is
 * eps = 0 on B * N = 32 768 rows: rows strided over pcl_soft_ce_rows_blocks(R) workgroups, partial [that many floats] = the workgroups'
 * loss sums, folded in a fixed order by a second one-workgroup launch (deterministic); dlogits [R][C] (nullable) = the gradient of the
 * MEAN loss.  R * C < 2^31. */
int pcl_soft_ce_rows_blocks(int R);
int pcl_soft_ce_rows_f32(const float* logits, const int64_t* target, float eps, int R, int C, float* partial, float* loss, float* dlogits,
                         void* stream);

/* DGCNN EdgeConv without the edge tensor (networks/cls/dgcnn.py:29-50,:72-83,:100-111).  With the 1x1 conv weight split
 * W = [Wa | Wb], y[i,j] = U[nbr(i,j)] + V[i] where UV [B*N, 2C] = x [Wa ; Wb-Wa]^T is ONE plain GEMM over the points.
 *   pcl_edgeconv_gather_f32: per point and channel max/min of y over the k neighbours and their positions (the sign of
 *     the BatchNorm scale is not known yet; pcl_group_minmax_finalize_f32 picks), plus the BatchNorm batch sums of y over
 *     all B*N*k edges as pcl_edgeconv_stat_rows(B,N) fp64 partial rows [rows][2][C] (pcl_bn_finalize_f32 consumes them).
 *     sumU [B*N,C] (nullable): SU[i] = sum_j U[nbr(i,j)], which the backward uses.
 *   pcl_edgeconv_scatter_f32: dy = [j == arg] a*gz - k1 - k2*(y - mu) for every edge, summed into dUV (U half: over the
 *     edges pointing at a point; V half: over a point's own edges).  idx [B*N,k] int32 neighbour index within the cloud.
 *     With the transposed lists of pcl_knn_transpose_i32 and sumU the sums are formed without touching the edges:
 *     dV[i] = a gz[i] - k k1 - k2 (k (V[i]-mu) + SU[i]),  dU[n] = hits[n] - deg k1 - k2 (deg (U[n]-mu) + sum_{i->n} V[i]),
 *     hits by one atomic per (point, channel); with in_off = in_src = sumU = NULL: one atomic per (edge, channel).
 *   pcl_knn_transpose_i32: idx [B,N,k] -> in_off [B*N+1] (global offsets into in_src), in_src [B*N*k] (for every point the
 *     sources i, index within the cloud, of the edges i->n, ascending).  N <= 8192. */
int pcl_edgeconv_stat_rows(int B, int N);
/* reference: the same stage (networks/cls/dgcnn.py:29-50,:100-111); UVlo [B*N, 2C] = the residuals of U | V from pcl_frag_linear_fwd_f32's
 * Y_lo: y = (U[nbr] + V[i]) + (Ulo[nbr] + Vlo[i]) is the fp32 rounding of the exact edge value -- U[nbr] + V[i] alone is a difference of
 * two large products once neighbours are close in feature space, and picks other max-pool winners than the fp64 evaluation */
int pcl_edgeconv_gather_hilo_f32(const float* UV, const float* UVlo, const int32_t* idx, int B, int N, int k, int C, float* ymax, float* ymin,
                                 int32_t* jmax, int32_t* jmin, double* stats_ws, float* sumU, void* stream);
/* backward = 0: W = [Wa | Wb] [Co][2C] -> Wcat = [Wa ; Wb - Wa] [2Co][C] (the weight of the per-point GEMM UV = x Wcat^T);
 * backward = 1: dWcat [2Co][C] -> dW [Co][2C] (dWa = top - bottom, dWb = bottom). */
/* reference: the 1x1 conv over [x_nbr - x_i, x_i] of get_graph_feature + conv, networks/cls/dgcnn.py:29-50,:100-111 */
int pcl_edgeconv_wcat_f32(const float* src, int Co, int C, int backward, float* dst, void* stream);
/* reference: replaces get_graph_feature + conv + max over k, networks/cls/dgcnn.py:29-50, :72-83, :100-111 */
int pcl_edgeconv_gather_f32(const float* UV, const int32_t* idx, int B, int N, int k, int C, float* ymax, float* ymin,
                            int32_t* jmax, int32_t* jmin, double* stats_ws, float* sumU, void* stream);
/* reference: gradient of the same composition (networks/cls/dgcnn.py:100-111) */
int pcl_edgeconv_scatter_f32(const float* UV, const int32_t* idx, const float* gz, const int32_t* arg, const float* a,
                             const float* k1, const float* k2, const float* mu, int B, int N, int k, int C,
                             const int32_t* in_off, const int32_t* in_src, const float* sumU, float* dUV, void* stream);
/* reference: no counterpart (the transpose of the kNN graph of networks/cls/dgcnn.py:29-35, needed only by the backward) */
int pcl_knn_transpose_i32(const int32_t* idx, int B, int N, int k, int32_t* in_off, int32_t* in_src, void* stream);

/* PointConv's density-weighted contraction (misc/pointconv_utils.py:393-394, :319-320):
 *   out[g,c,m] = sum_s feat[g,s,c] * density[g,s] * weights[g,s,m]      feat [G,ns,C], density [G,ns], weights [G,ns,M],
 * out [G,C,M] (= the reference's `matmul((new_points*density)^T, weights).reshape(B,S,-1)` rows).  M must be 16
 * (WeightNet(3,16)).  The backward entry point returns all three input gradients. */
int pcl_pointconv_contract_f32(const float* feat, const float* density, const float* weights, int G, int ns, int C, int M,
                               float* out, void* stream);
/* The same contraction with the feature MLP's last BatchNorm + activation folded into the feature load (misc/pointconv_utils.py:384-389
 * feeding :393-394): Y [G,ns,C] is that layer's PRE-BatchNorm output, z = lrelu(scale*y + shift, slope) is formed on the fly.  The
 * backward returns du = d_feat * lrelu'(.) (the gradient w.r.t. the BatchNorm's output) and pcl_pointconv_contract_bn_stat_rows(G)
 * fp64 partial rows [rows][2][C] of (sum du, sum du*y) for pcl_bn_bwd_consts_f32 / pcl_mlp_stack_bwd_f32 (defer_act). */
int pcl_pointconv_contract_bn_f32(const float* Y, const float* scale, const float* shift, float slope, const float* density,
                                  const float* weights, int G, int ns, int C, int M, float* out, void* stream);
int pcl_pointconv_contract_bn_stat_rows(int G);
int pcl_pointconv_contract_bn_bwd_f32(const float* dout, const float* Y, const float* scale, const float* shift, float slope,
                                      const float* density, const float* weights, int G, int ns, int C, int M, float* du,
                                      float* dweights, float* ddensity, double* stats_ws, void* stream);
/* reference: gradient of misc/pointconv_utils.py:393-394 */
int pcl_pointconv_contract_bwd_f32(const float* dout, const float* feat, const float* density, const float* weights, int G,
                                   int ns, int C, int M, float* dfeat, float* dweights, float* ddensity, void* stream);

/* ---- per-group pointwise MLP: 1x1 conv + BatchNorm(train) + (Leaky)ReLU [+ max over the group] ---------
 * Replaces the nn.Conv(k=1)+nn.BatchNorm+nn.ReLU stacks of build_mlps (networks/cls/pointnet2.py:18-31;
 * DGCNN conv1-4 networks/cls/dgcnn.py:72-83; FP stacks misc/ops.py:54-64) and the max over the group
 * (pointnet2.py:57, dgcnn.py:102), forward and backward, on channel-last rows [P,C] (P = B*m*ns).
 * fp32-input MFMA kernels; BatchNorm+activation of the layer below is folded into the operand staging, the
 * batch statistics come out of the GEMM epilogues.  Notation per layer l: y = z_prev W^T (+bias) (stored),
 * u = scale*y + shift (BatchNorm, scale = gamma*invstd, shift = beta - scale*mean), z = lrelu(u).
 *
 * Statistics workspaces `stats_ws` are [rows][2][C] doubles with rows <= 1024; a GEMM epilogue producing C
 * channels over P rows writes pcl_mlp_stat_rows(P, C, flags) rows (flags: bit 0 = the producer is pcl_linear_bwd_dx*
 * rather than pcl_linear_fwd*, bit 1 = it runs on duplicate-compacted rows, i.e. row_meta is passed), the elementwise
 * producers report the count through *stat_rows_out. */
int pcl_mlp_stat_rows(int P, int C, int flags);
/* Y[P,Cout] = act_in(X[P,Cin]) W[Cout,Cin]^T (+bias);  act_in = identity (in_scale NULL) or
 * lrelu(in_scale*x+in_shift, in_slope).  stats_ws rows: (sum Y, sum Y^2) per channel. */
/* reference: replaces nn.Conv 1x1 (+ the BatchNorm/ReLU of the layer below), networks/cls/pointnet2.py:25-29, execute :53-54 */
int pcl_linear_fwd_f32(const float* X, const float* W, const float* bias, const float* in_scale,
                       const float* in_shift, float in_slope, int P, int Cin, int Cout, float* Y,
                       double* stats_ws, void* stream);
/* Same, for the LAST layer of a stack that is max-pooled over groups of ns (32 or 64) consecutive rows: the
 * epilogue also emits per-group max / min of Y and the row-in-group attaining them (first occurrence);
 * pcl_group_minmax_finalize_f32 then yields out = max_s lrelu(scale*y+shift) (max for scale >= 0, min otherwise),
 * arg and ymax without re-reading Y. */
/* reference: replaces the last nn.Conv 1x1 + argmax over nsample, networks/cls/pointnet2.py:25-29 and :57 */
int pcl_linear_fwd_gmax_f32(const float* X, const float* W, const float* bias, const float* in_scale,
                            const float* in_shift, float in_slope, int P, int Cin, int Cout, int ns, float* Y,
                            double* stats_ws, float* gmax, float* gmin, int32_t* gamax, int32_t* gamin, void* stream);
/* reference: replaces new_feature.argmax(dim=2)[1], networks/cls/pointnet2.py:57 (dgcnn.py:102-111: x.max(dim=-1)) */
int pcl_group_minmax_finalize_f32(const float* gmax, const float* gmin, const int32_t* gamax, const int32_t* gamin,
                                  const float* scale, const float* shift, float slope, int G, int C, float* out,
                                  int32_t* arg, float* ymax, void* stream);
/* ... and a second copy of `out` as a column slice of a wider matrix, out2[g * out2_ld + c] (round 6: DGCNN's concat(x1..x4),
 * networks/cls/dgcnn.py:112, is written by its producers; out2 may be null) */
int pcl_group_minmax_finalize2_f32(const float* gmax, const float* gmin, const int32_t* gamax, const int32_t* gamin,
                                   const float* scale, const float* shift, float slope, int G, int C, float* out,
                                   int32_t* arg, float* ymax, float* out2, int out2_ld, void* stream);
/* ... and a third copy transposed per cloud, out_t[b][c][n] for G = B * N groups (N % 32 == 0): the layout the next EdgeConv stage's KNN
 * reads (networks/cls/dgcnn.py:34: the reference transposes nothing because its tensors are [B, C, N] throughout; here activations are
 * channel-last and the stage hands both layouts on) */
int pcl_group_minmax_finalize_t_f32(const float* gmax, const float* gmin, const int32_t* gamax, const int32_t* gamin,
                                    const float* scale, const float* shift, float slope, int B, int N, int C, float* out,
                                    int32_t* arg, float* ymax, float* out2, int out2_ld, float* out_t, void* stream);
/* mean/var (biased, max(E[y^2]-E[y]^2,0)) from the partials -> scale, shift, mean, invstd; running stats
 * r += (batch - r)*momentum with the biased variance (nullable). */
/* reference: replaces nn.BatchNorm (training mode) statistics, networks/cls/pointnet2.py:28 (dgcnn.py:66-70) */
int pcl_bn_finalize_f32(const double* stats_ws, int stat_rows, const float* gamma, const float* beta, int P,
                        int C, float eps, float momentum, float* scale, float* shift, float* mean_out,
                        float* invstd_out, float* running_mean, float* running_var, void* stream);
/* Global max AND mean pooling of z = lrelu(scale*Y + shift) over the ns rows of each of G groups, straight from the pre-BatchNorm
 * rows Y [G*ns, C] (the activation is never materialised): out_max[g*ldo + c], out_mean[g*ldo + c] (ldo >= C: both may be halves
 * of one [G, 2C] tensor), arg [G,C] = first row attaining the max.  Backward: du [G*ns, C] = act'(.) * ([s == arg] gmax + gmean / ns)
 * (gmax / gmean rows ldg floats apart) and its BatchNorm-backward sums as fp64 partial rows [stat_rows][2][C] (<= 1024 rows;
 * feed pcl_bn_bwd_consts_f32, or pcl_mlp_stack_bwd_f32 of a defer_act stack as ext_stats).  Sizes: G <= 65535 (the groups are the
 * launch grid's y axis; PCL_EINVAL beyond), ns and C any positive value. */
/* reference: replaces BatchNorm + LeakyReLU of conv5 and x.max(dim=-1) / x.mean(dim=-1) + concat, networks/cls/dgcnn.py:113-116 */
int pcl_bn_act_max_mean_f32(const float* Y, const float* scale, const float* shift, float slope, int G, int ns, int C, int ldo,
                            float* out_max, float* out_mean, int32_t* arg, void* stream);
int pcl_bn_act_max_mean_bwd_f32(const float* gmax, const float* gmean, int ldg, const int32_t* arg, const float* Y, const float* scale,
                                const float* shift, float slope, int G, int ns, int C, float* du, double* stats_ws,
                                int* stat_rows_out, void* stream);
/* out[g,c] = max_s lrelu(scale*Y[g*ns+s,c]+shift); arg = first s attaining it; ymax = Y there (nullable). */
/* reference: replaces BatchNorm + ReLU + argmax over nsample, networks/cls/pointnet2.py:28-29, :57 */
int pcl_bn_act_max_f32(const float* Y, const float* scale, const float* shift, float slope, int G, int ns,
                       int C, float* out, int32_t* arg, float* ymax, void* stream);
/* out = lrelu(scale*Y+shift) on [P,C];  bwd: du = gz*act'(u) plus partial (sum du, sum du*y). */
/* reference: replaces BatchNorm + ReLU of the last layer when nothing is pooled (PointNetFeaturePropagation's mlp, misc/ops.py:100-105) */
int pcl_bn_act_f32(const float* Y, const float* scale, const float* shift, float slope, int P, int C,
                   float* out, void* stream);
/* reference: gradient of that BatchNorm + ReLU (misc/ops.py:100-105) */
int pcl_bn_act_bwd_f32(const float* gz, const float* Y, const float* scale, const float* shift, float slope,
                       int P, int C, float* du, double* stats_ws, int* stat_rows_out, void* stream);
/* Stand-alone training-mode BatchNorm over the rows of x [P,C] -- the BatchNorm AFTER an activation of PointCNN's Conv / SepConv /
 * Dense blocks (misc/layers.py:151-169,:173-206).  pcl_bn_rows_stats_f32 leaves fp64 partial rows [*stat_rows_out <= 1024][2][C] of
 * (sum x, sum x^2) (g == NULL: forward; then pcl_bn_finalize_f32 and pcl_bn_act_f32 with slope 1 apply it) or of (sum g, sum g*x)
 * (backward; then pcl_bn_bwd_consts_f32 and pcl_bn_rows_bwd_apply_f32: dx = a*g - k1 - k2*(x - mean)). */
int pcl_bn_rows_stats_f32(const float* x, const float* g, int P, int C, double* stats_ws, int* stat_rows_out, void* stream);
/* reference: gradient of nn.BatchNorm (training mode) w.r.t. its input, misc/layers.py:156,:192 */
int pcl_bn_rows_bwd_apply_f32(const float* g, const float* x, const float* a, const float* k1, const float* k2, const float* mean,
                              int P, int C, float* dx, void* stream);
/* backward of the max: gz[g,c] = gout*act'(out) plus partial (sum gz, sum gz*ymax). */
/* reference: gradient of the max over nsample, networks/cls/pointnet2.py:57 */
int pcl_maxgrad_prep_f32(const float* gout, const float* out, const float* ymax, float slope, int G, int C,
                         float* gz, double* stats_ws, int* stat_rows_out, void* stream);
/* BatchNorm backward constants from (sum du, sum du*y): dgamma, dbeta (nullable) and a, k1, k2 with
 * dy = a*du - k1 - k2*(y - mean)   (k1 = a*dbeta/P, k2 = a*dgamma*invstd/P; y is centred where it is used).
 * dbias_zero [C] (nullable) is cleared: the gradient of a conv bias that feeds training-mode BatchNorm is exactly 0. */
/* reference: gradient of training-mode nn.BatchNorm, networks/cls/pointnet2.py:28 */
int pcl_bn_bwd_consts_f32(const double* stats_ws, int stat_rows, const float* gamma, const float* mean,
                          const float* invstd, int P, int C, float* dgamma, float* dbeta, float* a_out,
                          float* k1, float* k2, float* dbias_zero, void* stream);
/* dUprev[P,Cin] = act'_prev(.) * (dy[P,Cout] W[Cout,Cin]), dy = a*du - k1 - k2*(Y - mu) formed on the fly from
 * dU (dense) or from (arg, gz, ns) (sparse max gradient; pass dU = NULL).  With Yprev: masked by the layer
 * below's activation and stats_ws gets (sum dUprev, sum dUprev*Yprev); Yprev NULL: plain store (input grad).
 * W is the layer's weight as stored, [Cout,Cin] row-major (no transposed copy is needed). */
/* reference: gradient of nn.Conv 1x1 w.r.t. its input fused with BatchNorm/ReLU backward, networks/cls/pointnet2.py:25-29 */
int pcl_linear_bwd_dx_f32(const float* dU, const float* Y, const float* a, const float* k1, const float* k2, const float* mu,
                          const int32_t* arg, const float* gz, int ns, const float* W, int P, int Cout,
                          int Cin, const float* Yprev, const float* prev_scale, const float* prev_shift,
                          float prev_slope, float* dUprev, double* stats_ws, void* stream);
/* dW[Cout,Cin] = dy^T act_prev(Xprev);  workspace of pcl_linear_bwd_dw_workspace_bytes() bytes. */
/* Fused backward of one hidden layer: ONE pass over the rows forms dy = a*du - w*(k1 + k2*(y - mu)) once and produces BOTH
 * dU_prev = (dy W) * lrelu'(BN_prev(Yprev)) (+ its BatchNorm-backward sums, [pcl_linear_bwd_fused_stat_rows(P, Cin)][2][Cin]
 * fp64) and this workgroup's partial of dW = dy^T lrelu(BN_prev(Yprev)) in `workspace`
 * (pcl_linear_bwd_fused_workspace_bytes(P, Cout, Cin)); Yprev is read once.  Arguments as pcl_linear_bwd_dx_rows_f32.
 * pcl_linear_bwd_fused_finish_f32 (second launch) sums the partial tiles into dW[Cout][Cin] and -- when stats_ws is given --
 * computes the BatchNorm-backward constants of the layer below from those sums (what pcl_bn_bwd_consts_f32 does).
 * Precondition of the sparse mode (arg, gz) with row_meta: the compacted rows of a group are CONTIGUOUS and their row-in-group
 * ascends by one from the group's first row (what pcl_group_compact_f32 / pcl_group_linear_f32 produce): the kernel locates a
 * group's winner as (tile row of the group's first row) + arg - (row-in-group of that first row). */
/* reference: the autograd backward of nn.Conv 1x1 + nn.BatchNorm + nn.ReLU, networks/cls/pointnet2.py:25-29 */
/* test / tuning hook: cap the persistent grid of the fused backward at n workgroups (0 = one per CU); process-wide, set between calls */
void pcl_set_fb_max_blocks(int n);
/* lab switch (A/B runs, tests): 1 (default) = the 128 x 64 fused backward walks 64-row tiles over TWO LDS images with the waves in two roles
 * (dX | dW), the next tile deposited between the MFMAs of the current one; 0 = the one-image form of rounds 2-5 (128-row tiles).  Same
 * arithmetic per element; dW / BatchNorm partial sums differ in summation order only.  Process-wide, set between calls. */
void pcl_set_fb_two_images(int on);
int pcl_get_fb_two_images(void);
/* The optimiser step (reference: nn.SGD(net.parameters(), lr, momentum), train_cls.py:404; train_partseg.py: weight_decay 1e-4):
 *   g += weight_decay * p;  v = momentum * v + (1 - dampening) * g;  p -= lr * v        (no Nesterov, every tensor has its buffer v)
 * for n_tensors fp32 tensors in one launch per 96 tensors.  params / grads / bufs / numel are HOST arrays (device pointers, element
 * counts); pointers need 4-byte alignment only.  The arithmetic is torch.optim.SGD(fused=True)'s: products and sums in fp64 of the fp32
 * operands, one rounding per statement (tests/test_networks_gpu.py::test_lean_sgd_is_torch_fused_sgd: bit-identical). */
int pcl_sgd_momentum_f32(const uint64_t* params, const uint64_t* grads, const uint64_t* bufs, const int64_t* numel, int n_tensors, double lr,
                         double momentum, double weight_decay, double dampening, void* stream);
/* Lab switches of the kernel selection (no reference counterpart), 1 = on (default), 0 = off, negative = leave as is: the resident-weight
 * forward (off: linear_nt_kernel), the recompute-per-pass narrow stacks of PointConv's WeightNet / DensityNet (off: the GEMM kernels), the
 * fused dX + dW backward inside pcl_mlp_stack_bwd_f32 (off: separate kernels).  A C call: the library reads no environment variables. */
void pcl_set_kernel_paths(int fwd_resident, int narrow_stacks, int fused_backward);
/* Weight gradients of few-row plain stacks on a second stream (no reference counterpart; round 5).  In pcl_mlp_stack_bwd_f32 the chain
 * that later layers wait for is consts(l) -> dX(l) -> consts(l - 1) -> ...; dW(l) (+ its split-K reduce) only has to be done when the call
 * returns.  For plain stacks of <= max_rows rows (default 8192: the GroupAll level, the part-seg decoder) whose layers have no fused
 * dX + dW kernel, every dW launch is forked to a stream the library owns (event after consts(l)) and joined into the caller's stream at
 * the end of the call, so it runs in the holes of the dX chain; same kernels, same grids, per-layer buffers: results are bit-identical
 * to the serial order.  Measured not faster: OFF by default.  side_dw: 1 on, 0 off (default), negative = leave as is; max_rows <= 0 = leave as is.  Capturable in a HIP graph
 * (event fork / join from the capturing stream). */
void pcl_set_stack_overlap(int side_dw, int max_rows);
int pcl_get_stack_overlap(void);
/* Few-row layers (round 5; the reference's autograd of nn.Conv 1x1 + nn.BatchNorm + nn.ReLU on the GroupAll level,
 * networks/cls/pointnet2.py:131-136: [259, 256, 512, 1024] on B * 128 rows, and the part-seg decoder's FP stacks,
 * networks/seg/pointnet2_partseg.py:146-156).  On <= pcl_frag_max_rows() rows the staged backward GEMMs re-form
 * dy = a du - k1 - k2 (y - mean) once per output tile that reads it, and on gfx950 those vector instructions are matrix time.
 * pcl_bn_bwd_dy_f32 = pcl_bn_bwd_consts_f32 (same arguments, bit-identical constants / dgamma / dbeta) + dy[P][C] formed ONCE from
 * dU [P][C] or the max pool's (arg, gz) [P / ns][C] and Y [P][C]; C % 32 == 0.  pcl_linear_bwd_dw_plain_f32: dW[Cout][Cin] = dy^T z with
 * z = lrelu(prev_scale x + prev_shift) (or x when prev_scale is NULL) on the staged dW kernel reading dy as a plain operand (workspace:
 * pcl_linear_bwd_dw_plain_workspace_bytes); the input gradient of such a layer is pcl_frag_linear_bwd_dx_f32 on the same dy.
 * pcl_mlp_fewrow_layer: does layer (Cout <- Cin) of a plain stack on P rows take this path inside pcl_mlp_stack_bwd_f32 (the host's
 * per-kernel path asks the same question).  pcl_set_fewrow_backward: lab switch, 1 on / 0 off (default: measured a wash) / negative leave as is.
 * pcl_set_dw_tuning: lab knob, row-chunk workgroups per output tile of the plain-dy dW (0 = the library's choice). */
int pcl_bn_bwd_dy_supported(int P, int C);
int pcl_bn_bwd_dy_f32(const double* stats_ws, int stat_rows, const float* gamma, const float* mean, const float* invstd, int P_bn, int C,
                      float* dgamma, float* dbeta, float* a_out, float* k1, float* k2, float* dbias_zero, const float* dU, const float* Y,
                      const int32_t* arg, const float* gz, int ns, int P, float* dy, void* stream);
size_t pcl_linear_bwd_dw_plain_workspace_bytes(int P, int Cout, int Cin);
int pcl_linear_bwd_dw_plain_f32(const float* dy, const float* Xprev, const float* prev_scale, const float* prev_shift, float prev_slope, int P,
                                int Cout, int Cin, float* dW, void* workspace, size_t workspace_bytes, int dw_ld, void* stream);
int pcl_mlp_fewrow_layer(int P, int Cout, int Cin, int first_layer);
void pcl_set_fewrow_backward(int on);
int pcl_get_fewrow_backward(void);
void pcl_set_dw_tuning(int gx);
/* Both backward GEMMs of a few-row layer in ONE launch (round 6): dW partial tiles (the body of pcl_linear_bwd_dw_rows_f32) and
 * dUprev = mask(dy W) + the BatchNorm-backward sums of the layer below (the body of pcl_linear_bwd_dx_rows_f32), plain rows.  Arguments as
 * in those two; `masked` != 0: Xprev is the pre-BN output of the layer below (mask + sums), 0: Xprev is the stack's input (first_col as
 * in pcl_linear_bwd_dx_rows_f32).  pcl_linear_bwd_pair_finish_f32 then sums the partial tiles into dW and, when stats_ws is given, turns
 * the sums into the constants of the layer below in the same launch (the arguments of pcl_bn_bwd_consts_f32).  Results are bit-identical
 * to the separate calls.  pcl_linear_bwd_pair_supported: 1 where the pair applies (Cout, Cin > 64, few enough rows for the 64-row dX
 * tiles); workspace = pcl_linear_bwd_dw_workspace_bytes(P, Cout, Cin); dw_ld: row stride of dW (0: Cin -- the folded first layer's feature
 * columns are a column block of its [C1][3 + Cf] weight gradient).  pcl_set_bwd_pair(0): lab switch, `supported` answers 0.
 * reference: the autograd backward of nn.Conv 1x1 + nn.BatchNorm + nn.ReLU, networks/cls/pointnet2.py:25-29, at the GroupAll level :131-136 */
int pcl_linear_bwd_pair_supported(int P, int Cout, int Cin, int first_col);
int pcl_linear_bwd_pair_f32(const float* dU, const float* Y, const float* a, const float* k1, const float* k2, const float* mean,
                            const int32_t* arg, const float* gz, int ns, const float* W, const float* Xprev,
                            const float* prev_scale, const float* prev_shift, float prev_slope, int masked, int P, int Cout, int Cin,
                            float* dUprev, double* stats_ws, int first_col, void* workspace, size_t workspace_bytes, void* stream);
int pcl_linear_bwd_pair_finish_f32(const void* workspace, size_t workspace_bytes, int P, int Cout, int Cin, float* dW, int dw_ld,
                                   const double* stats_ws, int stat_rows, const float* gamma_prev, const float* mean_prev,
                                   const float* invstd_prev, int P_bn, float* dgamma_prev, float* dbeta_prev,
                                   float* a_prev, float* k1_prev, float* k2_prev, float* dbias_zero_prev, void* stream);
void pcl_set_bwd_pair(int on);
int pcl_get_bwd_pair(void);
/* Matrix-pipe form of the GEMM family (no reference counterpart: the reference calls cuDNN / cuBLAS fp32 through jittor's nn.Conv /
 * nn.Linear, misc/layers.py:60-75).  Default 0: the fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere.  Opt-in, measured and not faster
 * as a whole: every fp32 operand split EXACTLY into three bf16 values, an fp32 product = nine exact bf16 products
 * accumulated in fp32 on v_mfma_f32_32x32x16_bf16.  form bit 0: the resident-operand forward (set-abstraction shapes); bit 1: the
 * staged GEMMs (forward and dX through linear_nt_kernel) with K >= (form >> 8), 128 if that field is 0; bit 2: the resident forward
 * for 128 -> 256 only.  Process-wide; takes effect
 * for launches enqueued after the call. */
void pcl_set_matrix_form(int form);
int pcl_get_matrix_form(void);
int pcl_linear_bwd_fused_supported(int Cout, int Cin);
int pcl_linear_bwd_fused_stat_rows(int P, int Cin);
size_t pcl_linear_bwd_fused_workspace_bytes(int P, int Cout, int Cin);
int pcl_linear_bwd_fused_rows_f32(const float* dU, const float* Y, const float* a, const float* k1, const float* k2, const float* mu,
                                  const int32_t* arg, const float* gz, int ns, const float* W, int P, int Cout, int Cin,
                                  const float* Yprev, const float* prev_scale, const float* prev_shift, float prev_slope,
                                  float* dUprev, double* stats_ws, void* workspace, size_t workspace_bytes,
                                  const int32_t* row_meta, const int32_t* n_rows_dev, void* stream);
int pcl_linear_bwd_fused_finish_f32(const void* workspace, size_t workspace_bytes, int P, int Cout, int Cin, float* dW,
                                    const double* stats_ws, const float* gamma_prev, const float* mean_prev,
                                    const float* invstd_prev, int P_bn, float* dgamma_prev, float* dbeta_prev, float* a_prev,
                                    float* k1_prev, float* k2_prev, float* dbias_zero_prev, void* stream);
/* reference: workspace of pcl_linear_bwd_dw_f32 (the reference lets Jittor allocate, networks/cls/pointnet2.py:25-29) */
size_t pcl_linear_bwd_dw_workspace_bytes(int P, int Cout, int Cin);
/* reference: gradient of nn.Conv 1x1 w.r.t. its weight, networks/cls/pointnet2.py:25-26 */
int pcl_linear_bwd_dw_f32(const float* dU, const float* Y, const float* a, const float* k1, const float* k2, const float* mu,
                          const int32_t* arg, const float* gz, int ns, const float* Xprev,
                          const float* prev_scale, const float* prev_shift, float prev_slope, int P, int Cout,
                          int Cin, float* dW, void* workspace, size_t workspace_bytes, void* stream);

/* ---- per-stack entry points -------------------------------------------------------------------------------------------
 * One call runs what the reference runs per module `execute` (networks/cls/pointnet2.py:33-62: group -> conv/bn/relu xL ->
 * max; PointNetFeaturePropagation's mlp, misc/ops.py:100-106): a whole [1x1 conv -> BatchNorm(training) -> (Leaky)ReLU] x L
 * stack, optionally with the ball-query grouping folded into its first layer (`grouped`, pcl_group_linear_f32 on
 * duplicate-compacted rows) and the max over each group.  The kernels are the ones behind the per-kernel entry points
 * above, launched in the same order; the caller passes ONE descriptor, one persistent buffer (`save`, forward -> backward:
 * pre-BatchNorm outputs, per-layer scale/shift/mean/invstd, arg/ymax, row records) and one transient buffer (`tmp`), both
 * carved here (pcl_mlp_stack_sizes).  Training-mode BatchNorm on every layer with process-local statistics only.
 *   plain stack:   x [P, c[0]] -> out [P, c[L]] (pool = 0) or [P/pool, c[L]] (max over groups of `pool` consecutive rows)
 *   grouped stack: (xyz [B,N,3], new_xyz [B,m,3], feature [B,N,Cf] | NULL, idx [B,m,ns], cnt [B,m], group_off [B*m+1]) ->
 *                  out [B*m, c[L]]; c[0] = 3*use_xyz + Cf, P = B*m*ns (row capacity), pool = ns; layer[0].W is the stored
 *                  first-layer weight [c[1], c[0]] (no bias); Wf_dense = a caller-owned BUFFER [c[1], Cf] that the
 *                  forward call fills with the feature columns of that weight (dense: the point GEMM wants 16-byte aligned
 *                  rows) and the backward call reads -- keep it with `save`; needed when Cf > 4 or the features require
 *                  a gradient.
 * A plain three-layer stack of widths 3-8-8-16 or 1-8-8-1 without pooling or input gradient (PointConv's WeightNet / DensityNet,
 * misc/pointconv_utils.py:186-250) runs on kernels that exist only behind these entry points (csrc/narrow.hip: every BatchNorm
 * pass recomputes a row's chain from x, `save` is 768 bytes); same results to fp32 rounding, PCL_NARROW=0 keeps the GEMM kernels.
 * Backward: gout [as out]; writes layer[l].dW / dgamma / dbeta (/ dbias) and, with need_dx, dx = the input gradient
 * ([P, c[0]], columns below x_grad_from unwritten) of a plain stack or dfeature [B*N, Cf] of a grouped one. */
#define PCL_STACK_MAX_LAYERS 8
typedef struct pcl_stack_layer_t {
    const float* W;              /* [c[l+1], c[l]] */
    const float* bias;           /* nullable */
    const float* gamma;
    const float* beta;
    float* running_mean;         /* nullable, updated by the forward */
    float* running_var;
    float* dW;                   /* backward outputs */
    float* dbias;                /* nullable (must be given when bias is): exactly zero under BatchNorm */
    float* dgamma;
    float* dbeta;
} pcl_stack_layer_t;
typedef struct pcl_mlp_stack_t {
    int32_t struct_bytes;        /* sizeof(pcl_mlp_stack_t) as the caller sees it */
    int32_t n_layers;
    int32_t c[PCL_STACK_MAX_LAYERS + 1];
    int32_t P;
    int32_t pool;
    int32_t grouped;
    int32_t x_grad_from;
    int32_t need_dx;
    int32_t B, N, m, Cf, use_xyz;            /* grouped only */
    float slope, out_slope, eps, momentum;
    const float* x;                          /* plain only */
    const float* xyz;                        /* grouped only ... */
    const float* new_xyz;
    const float* feature;
    float* Wf_dense;
    const int32_t* idx;
    const int32_t* cnt;
    const int32_t* group_off;
    pcl_stack_layer_t layer[PCL_STACK_MAX_LAYERS];
    float* out;
    void* save;
    size_t save_bytes;
    void* tmp;
    size_t tmp_bytes;
    const float* gout;                       /* backward only */
    float* dx;
    void* stream;
    int32_t defer_act;                       /* 1 (pool == 0 only): the consumer applies the last layer's BatchNorm + activation itself while loading
                                              * (PointConv's contraction, pcl_pointconv_contract_bn_f32): the forward stops at the last pre-BatchNorm
                                              * output (in `save`, see pcl_mlp_stack_last) and writes no `out`; the backward takes gout = the gradient
                                              * w.r.t. that BatchNorm's OUTPUT already masked by the activation, with its (sum du, sum du*y) partial
                                              * rows in ext_stats [ext_stat_rows][2][c[L]] (pcl_pointconv_contract_bn_bwd_f32 leaves both) */
    int32_t ext_stat_rows;
    const double* ext_stats;
    int32_t flush_k;                         /* 0: the default GEMM kernels.  8 | 32: every forward GEMM of the stack that runs on plain rows (all layers
                                              * of a plain stack, the per-point product of a grouped one) goes through pcl_frag_linear_fwd_f32 with that
                                              * flush interval: fp32 fma chains of at most flush_k terms summed in fp64 (the part-seg decoder, where the
                                              * distance from the fp64 evaluation is accumulation error: DESIGN.md section 10).  Backward unchanged. */
    int32_t gout_ld;                         /* backward of a pooled stack: gout rows are gout_ld floats apart (0: dense, c[L]) -- the consumer concatenated
                                              * several stacks' outputs (multi-scale grouping) and hands each its column slice of the wide gradient */
} pcl_mlp_stack_t;
/* where, inside `save`, the last layer's pre-BatchNorm output [P, c[L]] and its folded BatchNorm (scale [c[L]], shift [c[L]]) live (byte
 * offsets): what a deferring consumer reads (defer_act) */
int pcl_mlp_stack_last(const pcl_mlp_stack_t* desc, size_t* y_offset, size_t* scale_offset, size_t* shift_offset);
/* reference: no counterpart (Jittor allocates); sizes of `save`, of the forward's `tmp` and of the backward's `tmp` */
int pcl_mlp_stack_sizes(const pcl_mlp_stack_t* desc, size_t* save_bytes, size_t* fwd_tmp_bytes, size_t* bwd_tmp_bytes);
/* reference: PointNetModuleBase.execute, networks/cls/pointnet2.py:33-62 (grouper :51, mlp :54, argmax :57) */
int pcl_mlp_stack_fwd_f32(const pcl_mlp_stack_t* desc);
/* reference: the autograd backward of the same module (Jittor derives it; networks/cls/pointnet2.py:33-62) */
int pcl_mlp_stack_bwd_f32(const pcl_mlp_stack_t* desc);

/* ---- layers with FEW ROWS: fragment-direct GEMMs (csrc/frag.hip, round 5) -----------------------------------------
 * The GroupAll level (networks/cls/pointnet2.py:131-136: [259, 256, 512, 1024] on B*128 rows), the part-seg decoder's feature
 * propagation stacks (networks/seg/pointnet2_partseg.py:146-148, misc/ops.py:54-64,:103-106) and DGCNN's per-point product
 * (networks/cls/dgcnn.py:72-83) have 2 048 .. 32 768 rows: fewer output tiles than the chip has SIMDs.  These entry points run the
 * same arithmetic as pcl_linear_fwd_f32 / _bwd_dx_f32 / _bwd_dw_f32 (reference: nn.Conv(k=1) + nn.BatchNorm + ReLU and its autograd
 * backward) with a wave per 32 x 32 output tile reading MFMA fragments straight from L2, K split over the waves of a workgroup and
 * reduced in LDS in fixed order.  Rows may have any stride and any 4-byte alignment (no padding of K = 259).
 *   flush_k = 0: fp32 accumulation (one fma chain per output); 8 | 32: the chain is cut every flush_k terms and summed in fp64 --
 *   the error of a K-term dot product drops by ~sqrt(flush_k / K); Y is the fp32 rounding of that sum.
 *   stats_ws: [pcl_frag_stat_rows(P)][2][Cout] fp64 (sum y, sum y^2) for pcl_bn_finalize_f32, or NULL (no BatchNorm behind it).
 *   pcl_frag_dy_f32 forms dy = a*du - (k1 + k2*(y - mu)) of a layer ONCE (du dense, or the sparse max gradient (arg, gz, ns));
 *   it also clears `n_zero` words at `zero_words` (the dW workspace's counters: pcl_frag_dw_counter_words) when given.
 *   pcl_frag_linear_bwd_dx_f32: dUprev[:, first_col:] = (dy W)[:, first_col:] * lrelu'(BN_prev(Yprev)) + its BatchNorm-backward sums
 *   ([pcl_frag_stat_rows(P)][2][Cin]: sum du, sum du*yprev); Yprev = NULL: plain product (the stack's input gradient).
 *   pcl_frag_linear_bwd_dw_f32: dW[Cout][ldo] = dy^T lrelu(BN_prev(Xprev)) (prev_scale = NULL: plain Xprev); rows split over the waves of
 *   a workgroup and over several workgroups per 64 x 64 tile, whose partial tiles the LAST workgroup to arrive sums in fixed order:
 *   deterministic, no second launch.  workspace: pcl_frag_dw_workspace_bytes; its first pcl_frag_dw_counter_words words must be zero
 *   at launch (counters_cleared = 0: the entry point clears them itself with a memset node).
 *   pcl_frag_set_tuning: max_rows = largest P for which the per-stack entry points choose these kernels (default 8192; < 0 keeps
 *   it); the other arguments force launch shapes (0 = automatic) -- tests and timing only. */
int pcl_frag_stat_rows(int P);
int pcl_frag_max_rows(void);
void pcl_frag_set_tuning(int max_rows, int force_tn, int force_ksw, int dw_tm, int dw_tn, int dw_ksw, int dw_ksg);
int pcl_frag_linear_fwd_f32(const float* X, int ldx, const float* W, int ldw, const float* bias, const float* in_scale,
                            const float* in_shift, float in_slope, int P, int Cin, int Cout, float* Y, int ldy,
                            float* Y_lo /* NULL, or (flush_k != 0) [P][ldy]: Y + Y_lo = the fp64 sum to ~2^-48 */, double* stats_ws,
                            int flush_k, void* stream);
int pcl_frag_dy_f32(const float* dU, const float* Y, const float* a, const float* k1, const float* k2, const float* mu,
                    const int32_t* arg, const float* gz, int ns, int P, int C, float* dy, uint32_t* zero_words, int n_zero, void* stream);
int pcl_frag_linear_bwd_dx_f32(const float* dy, const float* W, int ldw, int P, int Cout, int Cin, const float* Yprev, int ldyp,
                               const float* prev_scale, const float* prev_shift, float prev_slope, float* dUprev, int ldu,
                               double* stats_ws, int first_col, void* stream);
size_t pcl_frag_dw_workspace_bytes(int P, int Cout, int Cin);
int pcl_frag_dw_counter_words(int P, int Cout, int Cin);
int pcl_frag_linear_bwd_dw_f32(const float* dy, const float* X, int ldx, const float* prev_scale, const float* prev_shift, float prev_slope,
                               int P, int Cout, int Cin, float* dW, int ldo, void* workspace, size_t workspace_bytes, int counters_cleared,
                               void* stream);

/* ---- duplicate-compacted ("ragged") groups --------------------------------------------------------------
 * query_ball_point pads each group with copies of its first hit (misc/ops.py:321-324).  Identical rows stay
 * identical through conv/BN/ReLU, so the MLP can run on the DISTINCT rows only, carrying a multiplicity per row:
 * BatchNorm sums weight each row by it, the max over the group is unchanged, and in backward the dense BatchNorm
 * term of a row counts `multiplicity` times.  Results equal the padded computation up to fp32 summation order.
 *   pcl_group_compact_f32: rows [cap = B*m*ns, D] (first group_off[B*m] rows valid, (group, slot) order),
 *     row_meta [cap,2] int32 = {group id, slot | multiplicity << 16}, row_src [cap] = b*N + point index,
 *     group_off [B*m+1] (group_off[B*m] = number of valid rows, stays on the device: no host sync).
 *   The *_rows_f32 GEMM entry points are the pcl_linear_* ones with that metadata (row_meta, n_rows_dev =
 *   &group_off[B*m]); P is the capacity; with both NULL they are identical to the plain entry points.
 *   group_off depends on the ball-query counts only (pcl_group_offsets_i32), so it can be produced with the indices.
 *   pcl_bn_act_max_rows_f32: max over each group's valid rows (arg = slot in the compacted group).
 *   pcl_scatter_rows_add_f32: gfeat[row_src[r], c] += grows[r, off+c]  (zero-fills gfeat [n_dst_rows, C]). */
int pcl_group_offsets_i32(const int32_t* cnt, int G, int32_t* group_off, void* stream);   /* exclusive scan of max(cnt,1) */
/* the same scan for n (1..4) count arrays of G entries in one launch (the scales of a PointnetModuleMSG level, round 5); cnt / group_off are
 * HOST arrays of n device pointers */
int pcl_group_offsets_multi_i32(int n, const int32_t* const* cnt, int G, int32_t* const* group_off, void* stream);
/* reference: replaces BallQueryGrouper.execute's gathers + concat, misc/ops.py:383-407, on rows without the padding duplicates of misc/ops.py:321-324 */
int pcl_group_compact_f32(const float* xyz, const float* new_xyz, const float* feat, const int32_t* idx,
                          const int32_t* cnt, const int32_t* group_off, int B, int N, int m, int ns, int C, int use_xyz,
                          int row_stride /* >= 3*use_xyz + C; extra columns are written as zeros */, float* rows,
                          int32_t* row_meta, int32_t* row_src, void* stream);
/* reference: as pcl_linear_fwd_f32 (networks/cls/pointnet2.py:25-29) on duplicate-compacted rows */
int pcl_linear_fwd_rows_f32(const float* X, const float* W, const float* bias, const float* in_scale,
                            const float* in_shift, float in_slope, int P, int Cin, int Cout, float* Y,
                            double* stats_ws, const int32_t* row_meta, const int32_t* n_rows_dev, void* stream);
/* reference: as pcl_bn_act_max_f32 (networks/cls/pointnet2.py:28-29, :57) on duplicate-compacted rows */
int pcl_bn_act_max_rows_f32(const float* Y, const int32_t* group_off, const float* scale, const float* shift,
                            float slope, int G, int C, float* out, int32_t* arg, float* ymax, void* stream);
/* reference: as pcl_linear_bwd_dx_f32 (networks/cls/pointnet2.py:25-29) on duplicate-compacted rows */
int pcl_linear_bwd_dx_rows_f32(const float* dU, const float* Y, const float* a, const float* k1, const float* k2, const float* mu,
                               const int32_t* arg, const float* gz, int ns, const float* W, int P, int Cout,
                               int Cin, const float* Yprev, const float* prev_scale, const float* prev_shift,
                               float prev_slope, float* dUprev, double* stats_ws, const int32_t* row_meta,
                               const int32_t* n_rows_dev, int first_col /* input gradient only: columns below it
                               (the xyz part of a grouped tensor) are skipped and left unwritten */,
                               int cin_stride /* row stride, in floats, of W and of dUprev when the layer's input rows are zero-padded
                               beyond Cin (0 = Cin) */, void* stream);
/* reference: as pcl_linear_bwd_dw_f32 (networks/cls/pointnet2.py:25-26) on duplicate-compacted rows */
int pcl_linear_bwd_dw_rows_f32(const float* dU, const float* Y, const float* a, const float* k1, const float* k2, const float* mu,
                               const int32_t* arg, const float* gz, int ns, const float* Xprev,
                               const float* prev_scale, const float* prev_shift, float prev_slope, int P, int Cout,
                               int Cin, float* dW, void* workspace, size_t workspace_bytes, const int32_t* row_meta,
                               const int32_t* n_rows_dev, int dw_ld, void* stream);
/* reference: gradient of the gathers of misc/ops.py:384-396 on duplicate-compacted rows */
int pcl_scatter_rows_add_f32(const float* grows, const int32_t* row_src, const int32_t* n_rows_dev, int rows_cap,
                             int D, int off, int C, int n_dst_rows, float* gfeat, void* stream);

/* ---- PointCNN: the X-transform core of XConv --------------------------------------------------------------------------
 * Per region r (R = B*P of them): FX = X[r] (K x K) . [F1[r] | F2[r]] (K x (C1+C2)) -- jt.matmul(X, fts_cat),
 * misc/layers.py:505 with the concat of :486-489 -- then the depthwise (1,K) conv of SepConv (:151, groups = in_channels):
 * D[r, c*dm + j] = bias[c*dm + j] + sum_k wd[c, j, k] * FX[k, c].  One pass, FX and the concat never exist in memory; D feeds
 * the pointwise 1x1 conv (pcl_linear_fwd_f32).  Layouts: X [R,K,K], F1 [R,K,C1], F2 [R,K,C2] (or NULL, C2 = 0),
 * wd [C, dm, K] (reference weight wd_ref[c*dm+j, 0, 0, k]), bias [C*dm], D [R, C*dm].
 * Backward: dX [R,K,K], dF1, dF2 as the inputs; the tap / bias gradients come as per-workgroup partials
 * dwd_part [pcl_xconv_core_partials(R, C)][C*dm*K], dbias_part [..][C*dm] that the caller sums (no atomics).
 * Supported: K in {8, 12, 16}, dm in {1, 2, 4, 16} (pcl_xconv_core_supported). */
int pcl_xconv_core_supported(int K, int dm, int C);
int pcl_xconv_core_partials(int R, int C);
int pcl_xconv_core_fwd_f32(const float* X, const float* F1, int C1, const float* F2, int C2, const float* wd, const float* bias,
                           int R, int K, int dm, float* D, void* stream);
int pcl_xconv_core_bwd_f32(const float* X, const float* F1, int C1, const float* F2, int C2, const float* wd, const float* dD,
                           int R, int K, int dm, float* dX, float* dF1, float* dF2, float* dwd_part, float* dbias_part,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCL_HIP_H */
