/* se3et_hip.h -- C ABI of libse3et_hip.so: MI355X (gfx950) kernels for the SE3ET hot path.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no framework types; pointers are DEVICE pointers unless the name ends in _host;
 *   - tensors are dense, row-major, float32 / int64 / int32 / uint8 as typed below;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are asynchronous;
 *   - outputs and workspaces are caller-allocated; the library never allocates or frees device memory
 *     and keeps no global state (one stream per process is enough, several are safe: one workspace per stream);
 *   - three workspaces begin with arrival counters of an in-kernel reduction (se3_dense_norm_fwd, se3_group_norm_stats,
 *     se3_kpconv_so3_fused's split form): they must be ZERO before their first use; every completed call leaves them zero;
 *   - return value: 0 = ok, otherwise an SE3_ERR_* code (the Python host raises RuntimeError, mirroring the
 *     TORCH_CHECK failures of the reference extension, geotransformer/extensions/common/torch_helper.h:6-35).
 *
 * Hard limits (requests beyond them return SE3_ERR_UNSUPPORTED, nothing is truncated silently):
 *   - stacked calls take at most SE3_MAX_BATCH = 32 clouds (16 registration pairs per forward);
 *   - radius search keeps at most SE3_MAX_NEIGHBOR_LIMIT = 64 neighbours per query (the reference's limits are 36 / 38);
 *   - point_to_node_partition: point_limit <= 128 (the KITTI configuration's patch size);
 *   - attention: anchors * heads <= 32, head dimension in {8, 16, 32, 64}, channels of the relative-position kernel in {32, 64, 128, 256};
 *   - KPConv matrix-core path: input channels a multiple of 8, output channels a multiple of 32, num_support * 6 * in_channels < 2^31, the
 *     SE3ET slot tables (kanchor 6, 15 kernel points), |orbit sums| < 65504; se3_linear_f16: in_features a multiple of 32, |x| < 65504;
 *   - se3_dense_norm_fwd: in_features a power of two 32..1024, out_features 32 / 64 / 128 or a multiple of 256 up to 4096, channels per group
 *     a power of two <= 32, at most 16 segments; the f16 attention form (kv_pieces_workspace): head dimension 64.
 *
 * Each entry point names the reference interface it replaces (paths under the reference repository).
 */
#ifndef SE3ET_HIP_H_
#define SE3ET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE3_OK 0
#define SE3_ERR_INVALID_ARG 1
#define SE3_ERR_UNSUPPORTED 2
#define SE3_ERR_LAUNCH 3
#define SE3_ERR_WORKSPACE 4

#define SE3_MAX_BATCH 32          /* clouds per stacked call = 16 registration pairs per forward (the reference always stacks 2: ref, src) */
#define SE3_MAX_NEIGHBOR_LIMIT 64 /* radius search keeps at most this many nearest neighbours */

/* Library / build identification. */
const char* se3_version(void);
const char* se3_last_error(void);   /* text of the last failure on the calling thread */
/* Benchmark tuning / profiling hooks (process-global, defaults 0; every variant computes the same result): kernel variant + workgroups per
 * CU of the relative-position kernel (3 = exact f32 MFMAs instead of the f16 hi / lo split); variant of the attention kernel (5 = f16-split
 * flash loop, 6 / 7 two waves per workgroup, 8 / 10 one wave); variant 9 writes 32 clock64() stamps per wave to `stamps`. */
void se3_debug_set_bias_variant(int variant, int split);
void se3_debug_set_attention_variant(int variant);
void se3_debug_set_attention_profile(long long* stamps);
/* Variant bits of the fused KPConv kernel (se3_kpconv_so3_fused): 1 = consecutive 16-point tiles on ONE XCD (workgroup i runs on XCD i mod 8:
 * tile = start of that XCD's contiguous tile range + i / 8) instead of tile = workgroup index -- for row orders with spatial locality. */
void se3_debug_set_kpconv_variant(int variant);
/* Diagnostic bits of se3_kpconv_so3_union (timing only: results are wrong with any bit set): 1 producers skip the gather product, 2 the row loads,
 * 4 the A fragments; 8 consumers skip their MFMAs. */
void se3_debug_set_kpconv_union_variant(int variant);
/* se3_log_sinkhorn_fwd: 0 = the iteration in base 2 with the previous iteration's logsumexp as the shift (default), 1 = natural base with the exact
 * maximum in every pass (the reference's order of operations) -- A/B runs. */
void se3_debug_set_sinkhorn_variant(int variant);
/* Rows of dense launches (se3_linear_stream*, se3_dense_norm_fwd, se3_dense_residual_fwd, se3_linear_f16) whose values left the headroom of the
 * row's f16-split scale -- more than 2^8 times the largest magnitude of the row's first 32 values -- or held NaN / Inf, since the last
 * reset: such values are clamped to the f16 range (finite, wrong) and counted here (events: once per row, K-step and column block).  Synchronises the device; reset != 0 zeroes the count. */
unsigned long long se3_debug_dense_saturated_rows(int reset);
/* NaN / Inf values of K / V^T (and of the equivariant cross attention's q) that the operand split of the f16 attention kernels clamped since the
 * last reset.  Finite values of any size are not clamped: a query row of one head, 8 key rows of one head and a value channel of one cloud are
 * each scaled by their own power of two before the f16 hi / lo split and the kernels take the scales out of the f32 logits / output
 * (csrc/attention.hip: x6_split_kernel).  A non-zero count says the input already held NaN / Inf. */
unsigned long long se3_debug_attention_saturated(int reset);
/* Per-launch timing of the two RPE self-attention kernels (bench.py): while enabled every launch carries its own start / stop
 * HIP event pair (hipExtLaunchKernelGGL) on the launch stream; collect() waits for them, returns the count and fills the
 * durations (us) and tags (1 = relative-position logits kernel, 2 = attention kernel) in launch order. */
void se3_debug_kernel_timing(int enable);
int se3_debug_kernel_timing_collect(float* microseconds, int* tags, int capacity);
/* The same, plus per record the algorithmic bytes of the call (negative = equivariant call) for the logits launches (tag 1) of
 * se3_rpe_self_attention_stack*_fwd and 0 for every other launch. */
int se3_debug_kernel_timing_collect_ex(float* microseconds, int* tags, double* aux, int capacity);

/* ---- A2: stack-mode radius neighbour search ---------------------------------------------------------------
 * Replaces geotransformer.ext.radius_neighbors (geotransformer/extensions/pybind.cpp:6-11,
 * cpu/radius_neighbors/radius_neighbors.cpp:5-76, radius_neighbors_cpu.cpp:3-91) together with the column
 * truncation of modules/ops/radius_search.py:25-27.
 * For every query the support points of the same batch element with d2 = (dx*dx + dy*dy) + dz*dz < radius*radius
 * (float32, unfused) are ranked by (d2, index); the first `limit` indices go to neighbors[q * limit + j], unused
 * entries are filled with ns (the total support count).  max_count (`batch` int32, DEVICE) receives per batch element the
 * max over its queries of the number of in-radius points (the caller keeps min(limit, max over the elements) columns; with
 * several registration pairs stacked, a pair's own width is the max over its two clouds).  q_lengths_host / s_lengths_host are HOST
 * arrays of `batch` int64.  limit <= SE3_MAX_NEIGHBOR_LIMIT. */
int se3_radius_neighbors(const float* q_points, int64_t nq, const float* s_points, int64_t ns,
                         const int64_t* q_lengths_host, const int64_t* s_lengths_host, int batch, float radius,
                         int limit, int64_t* neighbors, int32_t* max_count, void* stream);

/* Several pairs stacked in one table (se3et_amd/data.py): out (rows, width) = the first `width` columns of full (rows, full_width), with the
 * columns at and beyond a PAIR's own width set to -1 (pair p = rows [pair_row_ends[p-1], pair_row_ends[p]); a pair run alone keeps
 * min(limit, its largest neighbour count) columns -- the reference's collate, geotransformer/utils/data.py precompute_data_stack_mode).
 * HOST arrays, at most 64 pairs. */
int se3_neighbor_table_trim(const int64_t* full, int64_t rows, int full_width, int width, const int64_t* pair_row_ends_host,
                            const int* pair_widths_host, int num_pairs, int64_t* out, void* stream);
/* Uniform-grid variant of se3_radius_neighbors for large supports (identical results).  se3_radius_grid_build bins the
 * support cloud (cells of edge >= radius) into a caller-owned workspace; se3_radius_neighbors_grid then searches any query
 * set against it with the SAME radius.  One grid serves every search sharing support and radius (a stage's neighbour and
 * sub-sampling tables and the previous stage's up-sampling table in geotransformer/utils/data.py:48-84). */
size_t se3_radius_grid_workspace_bytes(int64_t ns, int batch);
int se3_radius_grid_build(const float* s_points, int64_t ns, const int64_t* s_lengths_host, int batch, float radius,
                          void* workspace, size_t workspace_bytes, void* stream);
int se3_radius_neighbors_grid(const float* q_points, int64_t nq, const int64_t* q_lengths_host, const int64_t* s_lengths_host,
                              int64_t ns, int batch, const void* grid_workspace, float radius, int limit, int64_t* neighbors,
                              int32_t* max_count, int max_count_is_zero, void* stream);
/* (max_count_is_zero != 0: the caller cleared max_count itself -- e.g. one fill for the counters of all searches of a pyramid -- and the
 * call does not clear it again; the search only ever raises the values.) */

/* ---- A2, exact ties: the reference's order of exactly tied distances (round 6; csrc/radius_ties.hip) ----------------------------------
 * The reference's row is the head of ALL in-radius matches in the order of its k-d tree walk, std::sort-ed (unstable) on the distance alone
 * (extensions/extra/nanoflann/nanoflann.hpp:857-1002,1286-1287,1348-1407; cpu/radius_neighbors/radius_neighbors_cpu.cpp:29-90): which
 * member of a group of EXACTLY equal float32 distances comes first -- and which ones a neighbour limit keeps -- depends on that walk.  The
 * searches above order such groups by index.  Their `_ties` forms additionally append the number of every row whose kept columns hold an
 * exact tie (among themselves or with the first column cut) to tie_rows (DEVICE int32, room for nq entries) behind the DEVICE counter
 * tie_count (int32, cleared by the caller; both NULL: the plain search).  For those rows se3_radius_neighbors_tie_order reproduces the
 * reference: one lane per row walks the reference's tree -- built on the HOST by se3_kdtree_build_host from a host copy of the support
 * clouds (se3_kdtree_max_bytes: size of the buffer; *used_bytes: what to upload) -- with the reference's float arithmetic, collects the
 * matches in walk order and sorts them with a restatement of libstdc++'s std::sort.  max_hits >= the search's largest max_count;
 * scratch: DEVICE, se3_radius_tie_scratch_bytes(num_tie_rows, max_hits).  Clouds without ties never pay for any of this. */
int se3_radius_neighbors_ties(const float* q_points, int64_t nq, const float* s_points, int64_t ns, const int64_t* q_lengths_host,
                              const int64_t* s_lengths_host, int batch, float radius, int limit, int64_t* neighbors, int32_t* max_count,
                              int32_t* tie_rows, int32_t* tie_count, void* stream);
int se3_radius_neighbors_grid_ties(const float* q_points, int64_t nq, const int64_t* q_lengths_host, const int64_t* s_lengths_host,
                                   int64_t ns, int batch, const void* grid_workspace, float radius, int limit, int64_t* neighbors,
                                   int32_t* max_count, int max_count_is_zero, int32_t* tie_rows, int32_t* tie_count, void* stream);
size_t se3_kdtree_max_bytes(int64_t ns, int batch);
int se3_kdtree_build_host(const float* s_points_host, int64_t ns, const int64_t* s_lengths_host, int batch, void* tree_host, size_t capacity,
                          size_t* used_bytes);
size_t se3_radius_tie_scratch_bytes(int64_t num_tie_rows, int max_hits);
int se3_radius_neighbors_tie_order(const float* q_points, int64_t nq, const float* s_points, int64_t ns, const int64_t* q_lengths_host,
                                   const int64_t* s_lengths_host, int batch, const void* tree_dev, float radius, int limit,
                                   const int32_t* tie_rows, int64_t num_tie_rows, int max_hits, void* scratch, size_t scratch_bytes,
                                   int64_t* neighbors, void* stream);
/* The code of se3_radius_neighbors_tie_order's kernel run on HOST memory (all pointers host, tree_host as built): lets the walk and the
 * std::sort restatement be checked where there is no GPU.  *overflowed: rows that did not fit max_hits (left untouched). */
int se3_debug_radius_tie_order_host(const float* q_points, int64_t nq, const float* s_points, int64_t ns, const int64_t* q_lengths,
                                    const int64_t* s_lengths, int batch, const void* tree_host, float radius, int limit,
                                    const int32_t* tie_rows, int64_t num_tie_rows, int max_hits, int64_t* neighbors, int* overflowed);
/* The std::sort restatement of that kernel against libstdc++ on the host: mode 0 the whole sort against std::sort, mode 1 its heap sort against
 * std::partial_sort(first, last, last); keys compare on their high 32 bits alone.  Returns the number of differing positions (0: identical). */
int64_t se3_debug_std_sort_host(const unsigned long long* keys, int64_t n, int mode);

/* ---- A1: stack-mode grid subsampling ----------------------------------------------------------------------
 * Replaces geotransformer.ext.grid_subsampling (pybind.cpp:13-17, cpu/grid_subsampling/grid_subsampling.cpp:5-83,
 * grid_subsampling_cpu.cpp:3-109, grid_subsampling_cpu.h:24-74).  Per batch element: voxel hash, per voxel the
 * input point closest to the voxel mean (float32 accumulation in input order, first minimum), emitted in the
 * iteration order of libstdc++'s std::unordered_map<size_t,...> (emulated on the device).
 * s_points / s_normals must hold n rows; s_lengths (DEVICE, `batch` int64) receives the per-cloud counts and the
 * sampled clouds are written back to back.  `normals` may be NULL (then s_normals is not written). */
size_t se3_grid_subsample_workspace_bytes(int64_t n, int batch);
int se3_grid_subsample(const float* points, const float* normals, int64_t n, const int64_t* lengths_host, int batch,
                       float voxel_size, float* s_points, float* s_normals, int64_t* s_lengths, void* workspace,
                       size_t workspace_bytes, void* stream);
/* The same with the clouds' sizes in DEVICE memory (the s_lengths of a previous call): a pyramid stage that follows another without a
 * host synchronisation in between (the reference subsamples stage after stage on the CPU, geotransformer/utils/data.py:33-52).  n_rows =
 * the rows of `points` that exist, an upper bound of the sum of lengths_dev (only the first sum rows are read); outputs and the
 * workspace (se3_grid_subsample_workspace_bytes(n_rows, batch)) are sized by n_rows.  Results are those of se3_grid_subsample. */
int se3_grid_subsample_dev(const float* points, const float* normals, int64_t n_rows, const int64_t* lengths_dev, int batch,
                           float voxel_size, float* s_points, float* s_normals, int64_t* s_lengths, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- E4: log-domain Sinkhorn with dustbin (LearnableLogOptimalTransport.forward) ---------------------------------
 * Replaces geotransformer/modules/sinkhorn/learnable_sinkhorn.py:13-66.  scores (batch, rows, cols) float32,
 * row_masks (batch, rows) / col_masks (batch, cols) uint8 (1 = valid point), alpha: device pointer to the learnable
 * dustbin score, inf: the finite "minus infinity" (1e12 in the reference).  out (batch, rows+1, cols+1).
 * rows, cols <= 143. */
int se3_log_sinkhorn_fwd(const float* scores, const uint8_t* row_masks, const uint8_t* col_masks, const float* alpha,
                         int batch, int rows, int cols, int iterations, float inf, float* out, void* stream);
/* Backward of se3_log_sinkhorn_fwd (training step: autograd through learnable_sinkhorn.py:13-66): grad_out (batch, rows+1, cols+1) ->
 * grad_scores (batch, rows, cols) (0 at masked entries) and grad_alpha_partial (batch): the gradient of the dustbin score per patch pair,
 * summed by the caller.  One workgroup per patch pair: the forward iterations again with every (u_t, v_t) kept in LDS, then the reverse
 * sweep.  iterations * (rows + cols + 2) * 4 bytes of LDS (<= 150 KB). */
int se3_log_sinkhorn_bwd(const float* scores, const uint8_t* row_masks, const uint8_t* col_masks, const float* alpha,
                         const float* grad_out, int batch, int rows, int cols, int iterations, float inf, float* grad_scores,
                         float* grad_alpha_partial, void* stream);

/* ---- B2: GroupNorm over stacked points, fused with "+ residual" and LeakyReLU ------------------------------------
 * Replaces GroupNormEPN (geotransformer/modules/e2pn/blocks_epn.py:684-701) and kpconv GroupNorm
 * (geotransformer/modules/kpconv/modules.py:34-51) plus the activation / shortcut add that follows them
 * (blocks_epn.py:660-664, 741-742, 838-852).  x (rows, channels): every leading dim (points, anchors) is a row; the
 * statistics of a group span all rows.  residual may be NULL.  out = act(norm(x + x_bias) * weight + bias + residual);
 * x_bias (channels, may be NULL) is the bias of the linear layer that produced x (the UnaryBlock mlp): folding it in here lets
 * that GEMM run bias-free. */
size_t se3_group_norm_workspace_bytes(int64_t rows, int channels, int groups);
int se3_group_norm_fwd(const float* x, const float* x_bias, const float* residual, const float* weight, const float* bias,
                       int64_t rows, int channels, int groups, float eps, int apply_leaky_relu, float slope, float* out,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Segmented form: segment_row_offsets_host (num_segments + 1 entries, HOST, first 0, last rows) cuts the rows into independent
 * ranges with their own statistics -- one registration pair each when several pairs share a launch (the reference normalises
 * per pair because it runs one pair per forward).  num_segments <= 16. */
int se3_group_norm_segments_fwd(const float* x, const float* x_bias, const float* residual, const float* weight,
                                const float* bias, int64_t rows, int channels, int groups,
                                const int64_t* segment_row_offsets_host, int num_segments, float eps, int apply_leaky_relu,
                                float slope, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* Backward of se3_group_norm_segments_fwd (training step: autograd of GroupNormEPN / F.leaky_relu, blocks_epn.py:684-701): grad_x (rows,
 * channels); grad_residual (rows, channels) or NULL; grad_params (num_segments, 3, channels) = every segment's contribution to (d weight,
 * d bias, d x_bias), summed over segments by the caller.  workspace: se3_group_norm_bwd_workspace_bytes(channels) bytes. */
size_t se3_group_norm_bwd_workspace_bytes(int channels);
int se3_group_norm_segments_bwd(const float* x, const float* x_bias, const float* residual, const float* weight, const float* bias,
                                const float* grad_out, int64_t rows, int channels, int groups, const int64_t* segment_row_offsets_host,
                                int num_segments, float eps, int apply_leaky_relu, float slope, float* grad_x, float* grad_residual,
                                float* grad_params, void* workspace, size_t workspace_bytes, void* stream);

/* Pending forms (inference path of the bottleneck blocks, blocks_epn.py:798-852): a GroupNorm [+ LeakyReLU] that has been reduced to its
 * affine table [segment][2][channels] (scale, shift: norm(x + x_bias) * weight + bias == x * scale + shift) but not applied -- the consumer
 * applies it while it loads x, so the normalised activation never makes a round trip through HBM.
 *   se3_group_norm_stats   the table of GroupNorm over T(x), T = a pending stage on x itself (in_affine NULL: T = identity).  Workspace:
 *                          se3_group_norm_stats_workspace_bytes(channels) bytes, ZERO before the first call (arrival counters of the
 *                          in-kernel finalize at its start; every call leaves them zero), one per stream.
 *   se3_group_norm_apply   out = lrelu_f( Tb(Ta(x)) + R ),  T.(v) = lrelu_slope(v * scale + shift);  R = residual * scale_r + shift_r
 *                          (residual_affine: the shortcut branch's own pending GroupNorm), the plain residual, or nothing.  A slope of 1
 *                          is "no LeakyReLU".  channels % 4 == 0.  blocked_layout = 1: x is (points, 6, channels) and out is written as
 *                          [point][channels / 16][anchor pair][16][2], the layout se3_kpconv_so3_fused gathers whole cache lines from
 *                          (x_blocked = 1; channels % 16 == 0).  blocked_layout = 2: out as [point][channels / 8][6 anchors][8], the
 *                          layout se3_kpconv_so3_union loads whole 192-byte row chunks from (x_chunked = 1; channels % 8 == 0).
 *   se3_dense_norm_fwd     UnaryBlockEPN (blocks_epn.py:639-665) = mlp + GroupNormEPN with both of the above folded into the GEMM
 *                          (csrc/dense_norm.hip): out = T_b(T_a(x)) W^T WITHOUT the bias (raw), affine_out = the table of
 *                          GroupNorm(out + linear_bias) computed from the accumulators.  weight_pieces: se3_linear_split_weights_f16.
 *                          in_features a power of two 32..1024; out_features 32, 64, 128 or a multiple of 256; f16 hi/lo split arithmetic
 *                          (f32 accuracy, |T(x)| < 65504); out_features / groups a power of two <= 32.  Workspace:
 *                          se3_dense_norm_workspace_bytes(groups) bytes, ZERO before the first call (it starts with the arrival counters
 *                          of the in-kernel finalize; every call leaves them zero), one per stream.  out == NULL: statistics only (the
 *                          product is reduced to affine_out and never stored). */
size_t se3_group_norm_stats_workspace_bytes(int channels);
int se3_group_norm_stats(const float* x, const float* in_affine, float in_slope, const float* x_bias, const float* weight, const float* bias,
                         int64_t rows, int channels, int groups, const int64_t* segment_row_offsets_host, int num_segments, float eps,
                         float* affine_out, void* workspace, size_t workspace_bytes, void* stream);
int se3_group_norm_apply(const float* x, const float* affine_a, float slope_a, const float* affine_b, float slope_b, const float* residual,
                         const float* residual_affine, float final_slope, int64_t rows, int channels,
                         const int64_t* segment_row_offsets_host, int num_segments, int blocked_layout, float* out, void* stream);
/* ... the same with the largest |out| of the call atomicMax-ed (as a bit pattern) into *amax_out (blocked layouts 1 / 2 only; the caller zeroes
 * the word): se3_kpconv_so3_fused_scaled / se3_kpconv_so3_union scale their input by a power of two from it before the f16 split. */
int se3_group_norm_apply_amax(const float* x, const float* affine_a, float slope_a, const float* affine_b, float slope_b, const float* residual,
                              const float* residual_affine, float final_slope, int64_t rows, int channels,
                              const int64_t* segment_row_offsets_host, int num_segments, int blocked_layout, float* out, float* amax_out,
                              void* stream);
size_t se3_dense_norm_workspace_bytes(int groups);
int se3_dense_norm_fwd(const float* x, int64_t rows, int in_features, const float* in_affine_a, float in_slope_a, const float* in_affine_b,
                       float in_slope_b, const void* weight_pieces, int out_features, const float* linear_bias, const float* norm_weight,
                       const float* norm_bias, int groups, float eps, const int64_t* segment_row_offsets_host, int num_segments, float* out,
                       float* affine_out, void* workspace, size_t workspace_bytes, void* stream);
/* Round 4 -- the tail of ResnetBottleneckBlockEPN (blocks_epn.py:838-852: `x = self.unary2(x); return leaky_relu(x + shortcut)`) without the
 * raw output of the expanding layers ever reaching HBM.  (1) se3_dense_norm_fwd with out == NULL: the GEMM of unary2 (and of skip_conv) is run
 * for its GroupNorm statistics only -> affine table.  (2) se3_dense_residual_fwd runs the GEMM again and writes
 *     out = lrelu( (T_b(T_a(x)) W^T) * scale + shift + R, final_slope ),   (scale, shift) = affine[segment] of step (1),
 * R = residual (rows, out_features), or the shortcut layer (x2 W2^T) * scale2 + shift2 (x2 (rows, in_features2) concrete, weight_pieces2,
 * affine2 from its own step (1); its norm weight must be non-zero: the two products share one accumulator set through the ratio
 * scale / scale2), or nothing (both NULL).  Same shape limits as se3_dense_norm_fwd; no workspace. */
int se3_dense_residual_fwd(const float* x, int64_t rows, int in_features, const float* in_affine_a, float in_slope_a, const float* in_affine_b,
                           float in_slope_b, const void* weight_pieces, const float* affine, const float* x2, int in_features2,
                           const void* weight_pieces2, const float* affine2, const float* residual, int out_features, float final_slope,
                           const int64_t* segment_row_offsets_host, int num_segments, float* out, void* stream);
/* Round 4 -- the streaming kernel as a plain dense layer for ANY row count and row strides: out = act(x W^T + bias), weight_pieces from
 * se3_linear_split_weights_f16 (in_features % 32 == 0, |x| < 65504, f32 accuracy).  Replaces the library GEMMs of the transformer's nn.Linear
 * layers (rpe_transformer.py:56-73,134-165; vanilla_transformer.py:22-37; output_layer.py:7-47; geotransformer.py:213-317).
 * se3_linear_stream_transposed stores the product of every block of `block_rows` rows transposed, out_t (rows / block_rows, out_features, ld):
 * the value projection in the attention kernels' operand layout V^T (A, C, Rp) straight from the packed rows (A, R, C). */
int se3_linear_stream(const float* x, int64_t rows, int in_features, int64_t x_row_stride, const void* weight_pieces, const float* bias,
                      int out_features, int apply_relu, float* out, int64_t out_row_stride, void* stream);
int se3_linear_stream_transposed(const float* x, int64_t rows, int in_features, int64_t x_row_stride, const void* weight_pieces,
                                 const float* bias, int out_features, int block_rows, float* out_t, int64_t ld, void* stream);
/* Tuning hook (tools/micro): workgroups se3_dense_norm_fwd aims at (default 768 = 3 per compute unit, all resident at once). */
void se3_dense_norm_set_target_chunks(int workgroups);

/* ---- D6: LayerNorm(hidden + residual) ---------------------------------------------------------------------------
 * Replaces the residual + nn.LayerNorm tails of geotransformer/modules/transformer/rpe_transformer.py:163-164,
 * vanilla_transformer.py:910-911 and output_layer.py:21,46.  hidden (rows, channels); residual (residual_rows,
 * channels) is broadcast with row % residual_rows (anchor broadcast).  hidden_bias (channels, may be NULL) is added to hidden
 * first: the bias of the output / FFN linear that produced it. */
int se3_add_layer_norm_fwd(const float* hidden, const float* hidden_bias, const float* residual, const float* weight,
                           const float* bias, int64_t rows, int64_t residual_rows, int channels, float eps, float* out,
                           void* stream);
/* Backward (training step: autograd through the residual + nn.LayerNorm tails): grad_hidden (rows, channels), which is also the gradient
 * of the residual before its anchor broadcast is summed; grad_params (3, channels) = (d weight, d bias, d hidden_bias), zero-initialised
 * by the caller and accumulated with float atomics. */
int se3_add_layer_norm_bwd(const float* hidden, const float* hidden_bias, const float* residual, const float* weight,
                           const float* grad_out, int64_t rows, int64_t residual_rows, int channels, float eps, float* grad_hidden,
                           float* grad_params, void* stream);

/* ---- B2 / D: dense layers on the f16 matrix cores at f32 accuracy (csrc/linear_f16.hip) ------------------------------------------------------
 * y = x W^T [+ bias] [ReLU]: UnaryBlockEPN.mlp / LastUnaryBlockEPN.mlp (blocks_epn.py:639-665,798-852), the kpconv UnaryBlock
 * (modules/kpconv/modules.py:54-93) and the transformer's nn.Linear layers (rpe_transformer.py:56-73, vanilla_transformer.py:22-37,
 * output_layer.py).  weight (out_features, in_features) row-major f32, in_features a multiple of 32; se3_linear_split_weights_f16 turns it
 * into f16 hi / lo MFMA fragments scaled by a power of two (`pieces`: se3_linear_weight_pieces_bytes bytes, once per weight version);
 * se3_linear_f16: x (rows, in_features) with row stride x_row_stride floats (16-byte aligned rows), out (rows, out_features) with row
 * stride out_row_stride; three products hi hi + hi lo + lo hi in f32 (2^-22 per term).  Range |x| < 65504. */
size_t se3_linear_weight_pieces_bytes(int out_features, int in_features);
int se3_linear_split_weights_f16(const float* weight, int out_features, int in_features, void* pieces, void* stream);
int se3_linear_f16(const float* x, int64_t rows, int in_features, int64_t x_row_stride, const void* weight_pieces, const float* bias,
                   int out_features, int apply_relu, float* out, int64_t out_row_stride, void* stream);

/* ---- B2/B3: padded row gather and neighbour max pooling ------------------------------------------------------------
 * Replace nearest_upsample (geotransformer/modules/kpconv/functional.py:6-22), the zero-padded patch gathers of
 * experiments/se3ete.3dmatch/model.py:108-111,190-193 and max_pool (geotransformer/modules/e2pn/blocks.py:93-110).
 * x (n, width); idx entries equal to n address an all-zero row. */
int se3_gather_rows_padded(const float* x, const int64_t* idx, int64_t n, int64_t m, int64_t width, float* out,
                           void* stream);
int se3_neighbor_max_pool(const float* x, const int64_t* idx, int64_t n, int64_t m, int nn, int64_t width, float* out,
                          void* stream);
/* Backward of se3_neighbor_max_pool (autograd of torch.max over the gathered rows, blocks.py:93-110): dx (n, width) += dout of every row
 * at its arg-max neighbour (first in table order on ties; nothing for the zero row of padded entries).  dx zero-initialised by the caller. */
int se3_neighbor_max_pool_bwd(const float* x, const int64_t* idx, const float* dout, int64_t n, int64_t m, int nn, int64_t width,
                              float* dx, void* stream);
/* Maximum over the anchor axis (InvOutBlockEPN, geotransformer/modules/e2pn/blocks_epn.py:908-926; the anchor 'amax' between
 * equivariant and invariant transformer blocks, transformer/conditional_transformer.py:282-283,299-302):
 * out[r, c] = max_a x[a * anchor_stride + r * row_stride + c], out (rows, channels) contiguous.  Serves (A, R, C)
 * (anchor_stride = R*C, row_stride = C) and (R, A, C) (anchor_stride = C, row_stride = A*C).  num_anchors = 6, channels % 4 == 0. */
int se3_anchor_max(const float* x, int num_anchors, int64_t rows, int channels, int64_t anchor_stride, int64_t row_stride,
                   float* out, void* stream);

/* ---- B1: E2PN anchor-group KPConv (KPConvInterSO3), neighbour-gather stage ---------------------------------------
 * Replaces feat_gather_by_perm + the (k, a) part of the weight contraction of
 * geotransformer/modules/e2pn/blocks_epn.py:334-390,454-546.  x (num_support, 6, Cin), idx (num_queries, NN) int64
 * padded with num_support; kernel_points_host (15, 3) float32, kidx_host (15, 6) int64 [k][r] -> weight slot,
 * ridx_host (6, 6) int64 [a][r] -> anchor slot: HOST arrays (tiny constant tables).  Writes
 * G (num_queries, 6[r], 6[s], 6[t], Cin) so that out[(p, r), :] = G[(p, r), (s, t, c)] @ weights[(s, t, c), :]. */
int se3_kpconv_so3_gather(const float* q_pts, const float* s_pts, const int64_t* idx, const float* x,
                          const float* kernel_points_host, const int64_t* kidx_host, const int64_t* ridx_host, float sigma,
                          int64_t num_queries, int64_t num_support, int num_neighbors, int in_channels, float* G,
                          void* stream);
/* Backward of that stage with respect to x (training step, experiments/se3ete.3dmatch: autograd through blocks_epn.py:454-546): with
 * dG = dout @ weights^T (num_queries * 6, 36 Cin) from a library GEMM, dx (num_support, 6, Cin) += gather^T(dG).  dx zero-initialised by
 * the caller; accumulated with float atomics (summation order = arrival order).  The weight gradient is the library GEMM G^T @ dout. */
int se3_kpconv_so3_gather_bwd(const float* q_pts, const float* s_pts, const int64_t* idx, const float* dG,
                              const float* kernel_points_host, const int64_t* kidx_host, const int64_t* ridx_host, float sigma,
                              int64_t num_queries, int64_t num_support, int num_neighbors, int in_channels, float* dx, void* stream);
/* The same with an ORDER-INDEPENDENT sum (round 5): the contributions are added as 64-bit fixed-point integers at a scale taken from max |dG|
 * (max_abs_dG: DEVICE word, any upper bound), so two runs are bit-identical; dx_fixed (num_support, 6, in_channels) int64, ZERO on entry;
 * se3_kpconv_fixed_to_float(dx_fixed, num_support * 6 * in_channels, max_abs_dG, num_queries, dx) converts the sums to float32. */
int se3_kpconv_so3_gather_bwd_fixed(const float* q_pts, const float* s_pts, const int64_t* idx, const float* dG,
                                    const float* kernel_points_host, const int64_t* kidx_host, const int64_t* ridx_host, float sigma,
                                    int64_t num_queries, int64_t num_support, int num_neighbors, int in_channels, const float* max_abs_dG,
                                    long long* dx_fixed, void* stream);
int se3_kpconv_fixed_to_float(const long long* dx_fixed, int64_t count, const float* max_abs_dG, int64_t num_queries, float* dx, void* stream);
/* Order-independent (bit-identical) forms of the other scatter-adds of the training step, the same 64-bit fixed-point scheme (csrc/common.h:
 * se3_fixed_scale_exp): se3_fixed_to_float(fixed, count, bound, terms, growth, out) converts sums accumulated with the same (bound, terms, growth).
 *   se3_neighbor_max_pool_bwd_fixed   backward of se3_neighbor_max_pool (terms = rows of idx, growth 0)
 *   se3_scatter_add_rows_fixed        backward of se3_gather_rows_padded (terms = gathered rows, growth 0)
 *   se3_add_layer_norm_bwd_partials   backward of se3_add_layer_norm_fwd with per-workgroup partial parameter sums
 *                                     (se3_add_layer_norm_bwd_blocks(rows), 3, channels) that the caller adds in order */
int se3_fixed_to_float(const long long* fixed, int64_t count, const float* bound, int64_t terms, int growth, float* out, void* stream);
int se3_neighbor_max_pool_bwd_fixed(const float* x, const int64_t* idx, const float* dout, int64_t n, int64_t m, int nn, int64_t width,
                                    const float* max_abs_dout, long long* dx_fixed, void* stream);
int se3_scatter_add_rows_fixed(const float* g, const int64_t* idx, int64_t n, int64_t m, int64_t width, const float* max_abs_g,
                               long long* dx_fixed, void* stream);
int64_t se3_add_layer_norm_bwd_blocks(int64_t rows);
int se3_add_layer_norm_bwd_partials(const float* hidden, const float* hidden_bias, const float* residual, const float* weight,
                                    const float* grad_out, int64_t rows, int64_t residual_rows, int channels, float eps, float* grad_hidden,
                                    float* grad_params_partials, void* stream);

/* Matrix-core form of the same convolution (csrc/kpconv_sums.h, csrc/kpconv_mfma.hip; input channels a multiple of 8, output channels a
 * multiple of 32, SE3ET slot tables compiled in; num_support * 6 * in_channels < 2^31).  Over all (weight slot s, output anchor r) only 16
 * distinct kernel-point sums occur (6 vertices, centre, 3 equators, 6 face quadruples): the "orbits".
 *   se3_kpconv_neighbor_table: per query point its valid neighbours (compacted; shadow / padded entries carry weight 0 in the reference,
 *     blocks_epn.py:471,377) and their 16 orbit weights hw[o] = sum_{k in o} max(0, 1 - |s - q - kp_k| / sigma) (blocks_epn.py:520-533);
 *     kernel_points_dev (15, 3) DEVICE; table: se3_kpconv_neighbor_table_bytes bytes.  Depends on the geometry only: layers that share
 *     (q_pts, s_pts, idx, kernel points, sigma) may share it.
 *   se3_kpconv_split_weights_f16: weights (36 Cin, Cout) -- KPConvInterSO3.weights (6, 6, Cin, Cout) flattened, blocks_epn.py:105-106 -- as f16
 *     hi / lo MFMA fragments scaled by a power of two (`pieces`: se3_kpconv_weight_pieces_bytes bytes incl. a 256-byte header with 1 / scale).
 *   se3_kpconv_so3_fused: KPConvInterSO3.forward (blocks_epn.py:454-546) in ONE kernel: producer waves form the orbit sums
 *     H[p, o, a, c] = sum_n hw[p, n, o] x[idx[p, n], a, c] of a 16-point tile on the f32 matrix cores and leave them, split into f16 hi + lo
 *     pieces, in LDS; consumer waves multiply them with v_mfma_f32_32x32x16_f16 (hi hi + hi lo + lo hi in f32: 2^-22 per term), reading the
 *     slot sums of blocks_epn.py:503-546 in place: out (P, 6, Cout).  The operand never exists in HBM.  With few tiles (one pair per
 *     forward, the coarse stages) the input channels of a tile are split over several workgroups whose partial outputs the last one to
 *     arrive adds in a fixed order: split_workspace = se3_kpconv_fused_split_workspace_bytes bytes (0: this shape does not split), ZERO
 *     before the first call (arrival counters at its start; every call leaves them zero), one per stream; NULL = never split.
 *     x_blocked = 1: x in the blocked layout written by se3_group_norm_apply(blocked_layout = 1) (in_channels % 16 == 0).
 *   se3_kpconv_so3_gather_sums + se3_kpconv_so3_contract_f16: the same two stages as two launches, H as tile images
 *     [Cin / 8][ceil(P / 16)][piece][point][97 x 16 B] in HBM (se3_kpconv_sums_bytes bytes).
 * Range: |H| < 65504 (f16 hi piece); values below 2^-3 keep an absolute error of 2^-25. */
size_t se3_kpconv_neighbor_table_bytes(int64_t num_queries, int num_neighbors);
int se3_kpconv_neighbor_table(const float* q_pts, const float* s_pts, const int64_t* idx, const float* kernel_points_dev, float sigma,
                              int64_t num_queries, int64_t num_support, int num_neighbors, void* table, size_t table_bytes, void* stream);
size_t se3_kpconv_weight_pieces_bytes(int in_channels, int out_channels);
int se3_kpconv_split_weights_f16(const float* weights, int in_channels, int out_channels, void* pieces, void* stream);
size_t se3_kpconv_fused_split_workspace_bytes(int64_t num_queries, int in_channels, int out_channels);
int se3_kpconv_so3_fused(const float* x, const void* table, int64_t num_queries, int64_t num_support, int num_neighbors, int in_channels,
                         int out_channels, const void* weight_pieces, float* out, void* split_workspace, size_t split_workspace_bytes,
                         int x_blocked, void* stream);
/* ... x_amax: DEVICE word holding the largest |x| (se3_group_norm_apply_amax) or NULL.  Outside [2^-4, 2^7) the features are scaled by the power
 * of two that brings it to [2^6, 2^7) before their f16 split and the output is scaled back: no magnitude window on x (NULL: |H| < 65504 as before). */
int se3_kpconv_so3_fused_scaled(const float* x, const void* table, int64_t num_queries, int64_t num_support, int num_neighbors, int in_channels,
                                int out_channels, const void* weight_pieces, float* out, void* split_workspace, size_t split_workspace_bytes,
                                int x_blocked, const float* x_amax, void* stream);
/* Union-staged form of se3_kpconv_so3_fused (csrc/kpconv_union.hip, round 5): a workgroup owns 16 points that are spatial neighbours, reads
 * the DISTINCT support rows of their neighbour lists once (whole rows, 16 B per lane) and forms every point's orbit sums as a product of
 * its orbit weights, scattered over the tile's row list, with the shared rows.  Tile membership only: no tensor is reordered.
 *   se3_point_order_keys / _place: a spatial order of one stage's stacked points: key = cloud << 32 | 30-bit Morton code of floor(p / cell);
 *     the caller sorts the keys (any stable sort; torch.sort) and _place writes `order` (se3_point_order_groups(lengths) * 16 int32: the point
 *     at every position, -1 = padding; every cloud starts a new group of 16, so a cloud's groups do not depend on what else is stacked).
 *   se3_kpconv_union_plan: per group of 16 order positions the distinct rows of the neighbour lists (se3_kpconv_neighbor_table) sorted by row
 *     number and every list slot's index into them; groups with more than 128 distinct rows are cut into halves (sub-tiles) until they
 *     fit.  plan: se3_kpconv_union_plan_bytes bytes; a function of (order, table) only.
 *   se3_kpconv_so3_union: KPConvInterSO3.forward (blocks_epn.py:454-546) as se3_kpconv_so3_fused; x_chunked = 1: x in the layout written by
 *     se3_group_norm_apply(blocked_layout = 2).  split_workspace: se3_kpconv_union_split_workspace_bytes bytes, same contract as the fused form.
 *     Summation order inside a point's neighbourhood: ascending support row; results agree with se3_kpconv_so3_fused to f32 rounding. */
int64_t se3_point_order_groups(const int64_t* cloud_lengths_host, int num_clouds);
/* (the same order in one launch, one workgroup per cloud sorting in LDS: clouds of at most 8192 points, SE3_ERR_UNSUPPORTED beyond) */
int se3_point_order(const float* points, int64_t num_points, const int64_t* cloud_lengths_host, int num_clouds, float cell, int32_t* order,
                   void* stream);
/* (up to four stages of one pyramid in one launch: arrays of the single-stage arguments) */
int se3_point_order_stages(const float* const* points, const int64_t* num_points, const int64_t* const* cloud_lengths_host, const int* num_clouds,
                           const float* cell, int32_t* const* order, int num_stages, void* stream);
int se3_point_order_keys(const float* points, int64_t num_points, const int64_t* cloud_lengths_host, int num_clouds, float cell, int64_t* keys,
                         void* stream);
int se3_point_order_place(const int64_t* sorted_keys, const int64_t* sorted_index, int64_t num_points, const int64_t* cloud_lengths_host,
                          int num_clouds, int32_t* order, void* stream);
size_t se3_kpconv_union_plan_bytes(int64_t num_groups, int num_neighbors);
int se3_kpconv_union_plan(const void* table, int64_t num_queries, int num_neighbors, const int32_t* order, int64_t num_groups, void* plan,
                          size_t plan_bytes, void* stream);
size_t se3_kpconv_union_split_workspace_bytes(int64_t num_groups, int in_channels, int out_channels);
int se3_kpconv_so3_union(const float* x, const void* table, const void* plan, int64_t num_groups, int64_t num_queries, int64_t num_support,
                         int num_neighbors, int in_channels, int out_channels, const void* weight_pieces, float* out, void* split_workspace,
                         size_t split_workspace_bytes, int x_chunked, const float* x_amax, void* stream);
size_t se3_kpconv_sums_bytes(int64_t num_queries, int in_channels);
int se3_kpconv_so3_gather_sums(const float* x, const void* table, int64_t num_queries, int64_t num_support, int num_neighbors,
                               int in_channels, void* sums, void* stream);
int se3_kpconv_so3_contract_f16(const void* sums, const void* weight_pieces, int64_t num_queries, int in_channels, int out_channels,
                                float* out, void* stream);

/* ---- D1/D2: RPE self attention (RPEMultiHeadAttention.forward) ------------------------------------------------------
 * Replaces geotransformer/modules/transformer/rpe_transformer.py:39-131 in two launches.
 * (1) se3_rpe_bias_fwd streams the (N, M, C) geometric embedding once and writes the relative-position logits
 *     bias[a*H+h, n, m] = qp[a, n, h, :] . emb[n, m, :] (+ qe[a, n, h, :] . eq_emb[a, n, m, :]), where qp = W_p^T q (C values
 *     per head) and qe = W_eq^T q (4 values per head) are the position projections folded onto the query side
 *     (anchors * heads <= 32).  qp and qe are column blocks of one projection output: element (a, n, h, c) of qp lives at
 *     qp[a * anchor_stride + n * row_stride + h * C + c], element (a, n, h, e) of qe at qe[a * anchor_stride + n * row_stride
 *     + 4 h + e] (strides in floats, multiples of 4).
 *     eq_emb (A, N, M, 4) and qe are NULL for non-equivariant layers.  bias has row stride bias_row_stride >= M.
 * (2) se3_attention_fwd: out[a, n, h*d:(h+1)*d] = softmax_m((q_a[n,h] . k_a[m,h] + bias[a*H+h, n, m]) * scale) v_a[m, h].
 *     q/k are (anchors, rows, row_stride >= C) -- column blocks of a wider projection are fine --, out (anchors, N, C);
 *     the values are passed TRANSPOSED, vt (anchors, C, v_row_stride) with v_row_stride a multiple of 4 >= ceil32(M)
 *     (entries beyond M must be finite), so that the P.V operand loads are contiguous; bias_row_stride likewise.  Anchor strides are in floats (0 = the same tensor for every anchor, which is how plain cross attention
 *     vanilla_transformer.py:39-85 with per-anchor values is expressed); bias may be NULL.  C / H in {8, 16, 32, 64}; C in {32,64,128,256} for (1). */
int se3_rpe_bias_fwd(const float* qp, const float* qe, int row_stride, int64_t anchor_stride, const float* emb,
                     const float* eq_emb, int N, int M, int C, int AH, int H, int bias_row_stride, float* bias, void* stream);
int se3_attention_fwd(const float* q, const float* k, const float* vt, const float* bias, int num_anchors, int N, int M, int C,
                      int H, int q_row_stride, int k_row_stride, int v_row_stride, int64_t q_anchor_stride,
                      int64_t k_anchor_stride, int64_t v_anchor_stride, int64_t out_anchor_stride, int bias_row_stride,
                      float scale, float* out, void* stream);

/* Stack mode of the two calls above (the reference runs the same RPETransformerLayer on the ref and the src cloud one after
 * the other, rpe_conditional_transformer.py:49-53; here the clouds of a pair share ONE launch per kernel).  The clouds' rows
 * are packed in one (anchors, rows, row_stride) projection: cloud c owns the query rows q_starts[c] .. + q_lengths[c] and the
 * key rows k_starts[c] .. + k_lengths[c] (self attention: the same rows; k_starts multiples of 32 because the transposed
 * values vt (anchors, C, v_row_stride) are addressed by key column).  emb_ptrs[c] -> (N_c, M_c, C), eq_ptrs[c] -> (A, N_c, M_c, 4)
 * or NULL (all clouds alike).  The logits of cloud c are the block bias + bias_offsets[c] of shape (A*H, N_c, ceil32(M_c)).
 * All arrays are HOST arrays of num_clouds (<= 16) entries.  out is (anchors, rows, C) in the packed row order. */
int se3_rpe_bias_stack_fwd(const float* qp, const float* qe, int row_stride, int64_t anchor_stride,
                           const float* const* emb_ptrs, const float* const* eq_ptrs, const int64_t* q_starts,
                           const int64_t* q_lengths, const int64_t* k_lengths, const int64_t* bias_offsets, int num_clouds,
                           int C, int AH, int H, float* bias, void* stream);
int se3_attention_stack_fwd(const float* q, const float* k, const float* vt, const float* bias, const int64_t* q_starts,
                            const int64_t* q_lengths, const int64_t* k_starts, const int64_t* k_lengths,
                            const int64_t* bias_offsets, int num_clouds, int num_anchors, int C, int H, int q_row_stride,
                            int k_row_stride, int v_row_stride, int64_t q_anchor_stride, int64_t k_anchor_stride,
                            int64_t v_anchor_stride, int64_t out_anchor_stride, float scale, float* out, void* kv_pieces_workspace,
                            size_t kv_pieces_bytes, void* stream);
/* kv_pieces_workspace (may be NULL): se3_attention_kv_pieces_bytes(anchors, max_c(k_starts[c] + k_lengths[c]), C, v_row_stride) bytes, 16-byte
 * aligned, one per stream.  With it (head dimension 64, k_starts multiples of 16, v_row_stride a multiple of 16) k and vt are split once
 * into f16 hi / lo pieces and both products run on the f16 matrix cores (three products, f32 accumulation: the error of an f32 product);
 * without it, or for other shapes, on the f32 matrix cores. */
size_t se3_attention_kv_pieces_bytes(int num_anchors, int64_t key_rows, int C, int v_row_stride);

/* The whole stack-mode RPE self-attention call (the reference's RPEMultiHeadAttention.forward up to the output projection,
 * rpe_transformer.py:39-131, for all clouds of the pair): se3_rpe_bias_stack_fwd into `logits_workspace` followed by
 * se3_attention_stack_fwd, launched back to back.  Queries and keys of cloud c are the packed rows starts[c] .. + lengths[c]
 * (starts multiples of 4, ceil32(lengths[c]) value columns readable); logits_workspace holds
 * sum_c A*H * lengths[c] * ceil32(lengths[c]) floats. */
int se3_rpe_self_attention_stack_fwd(const float* q, const float* k, const float* vt, const float* qp, const float* qe,
                                     int row_stride, int64_t anchor_stride, int v_row_stride, int64_t v_anchor_stride,
                                     const float* const* emb_ptrs, const float* const* eq_ptrs, const int64_t* starts,
                                     const int64_t* lengths, int num_clouds, int num_anchors, int C, int H,
                                     float* logits_workspace, int64_t out_anchor_stride, float* out, void* kv_pieces_workspace,
                                     size_t kv_pieces_bytes, void* stream);

/* bf16 geometric embedding (BASELINE.json configs[2]: "bf16 attention"): the same two calls with emb_ptrs[c] -> (N_c, M_c, C)
 * bfloat16 (16-byte aligned, C a multiple of 32), which halves the N*M*C term of the call's HBM bytes.  Queries, keys, values,
 * the equivariant embedding, the logits and the output stay float32; the folded queries are split into bf16 hi + lo parts
 * inside the kernel, so the only rounding is the stored embedding itself (2^-9 relative per element). */
int se3_rpe_bias_stack_bf16_fwd(const float* qp, const float* qe, int row_stride, int64_t anchor_stride,
                                const uint16_t* const* emb_ptrs, const float* const* eq_ptrs, const int64_t* q_starts,
                                const int64_t* q_lengths, const int64_t* k_lengths, const int64_t* bias_offsets, int num_clouds,
                                int C, int AH, int H, float* bias, void* stream);
int se3_rpe_self_attention_stack_bf16_fwd(const float* q, const float* k, const float* vt, const float* qp, const float* qe,
                                          int row_stride, int64_t anchor_stride, int v_row_stride, int64_t v_anchor_stride,
                                          const uint16_t* const* emb_ptrs, const float* const* eq_ptrs, const int64_t* starts,
                                          const int64_t* lengths, int num_clouds, int num_anchors, int C, int H,
                                          float* logits_workspace, int64_t out_anchor_stride, float* out, void* kv_pieces_workspace,
                                          size_t kv_pieces_bytes, void* stream);

/* ---- D4/D5: anchor-equivariant cross attention (MultiHeadAttentionEQ, 'a_soft' / 'r_soft') ----------------------------
 * Replaces geotransformer/modules/transformer/vanilla_transformer.py:247-476,506-577,751-870.  q (A, N, C), k/v (A, M, C).
 * se3_cross_eq_stats writes partial[(a*A+e) * P + i] whose sum over i is sum_{n,m} (mean_h q_a.k_e * scale)^2
 * (*num_partials_per_pair = P = ceil(N/32)); the caller turns g = sum / (N M) into the (A, A) mixing weights `mix`
 * (a_soft: g / sum_e g; r_soft: the 24 rotation weights collapsed onto anchor pairs) and se3_cross_eq_apply computes
 * out[a] = sum_e mix[a, e] softmax_m(q_a.k_e * scale) v_e (vt: transposed key-padded values (A, C, key_stride)). */
int se3_cross_eq_stats(const float* q, const float* k, int A, int N, int M, int C, int H, float scale, float* partial,
                       int* num_partials_per_pair, void* stream);
/* mode 0 = a_soft (weights: A*A values = mix), mode 1 = r_soft (weights: num_rotations values; trace_idx (R, A) int64) */
int se3_cross_eq_mix(const float* partial, int num_partials_per_pair, int A, int N, int M, int mode, const int64_t* trace_idx,
                     int num_rotations, float* mix, float* weights, void* stream);
int se3_cross_eq_apply(const float* q, const float* k, const float* vt, const float* mix, int A, int N, int M, int C, int H,
                       int key_stride, float scale, float* out, void* stream);
/* Stack mode: statistics, mixing weights and weighted attention of ALL pairs of a batch in three launches.  q (A, Rq, C) and
 * k (A, Rk, C) hold the packed rows of all pairs (pair p: query rows q_starts[p] .. + q_lengths[p], key rows k_starts[p] .. +
 * k_lengths[p]; row stride C, anchor strides given), vt (A, C, v_row_stride) the transposed values addressed by key column
 * (k_starts multiples of 4, ceil32(k_lengths[p]) columns readable).  partial_workspace: num_pairs * A*A * max_p ceil(N_p/32)
 * floats; with sums_given != 0 its first num_pairs * A*A floats already hold sum_{n,m} (mean_h S[a,e,h,n,m])^2 of every
 * (pair, a, e) and the statistics launch is skipped (mean_h S = (scale/H) q_a[n].k_e[m] over all C channels, so the caller can
 * get the sums from two Gram matrices per pair: (scale/H)^2 <Q_a^T Q_a, K_e^T K_e>_F -- 3.5x fewer flops, library GEMMs); mix (num_pairs, A, A); weights (num_pairs, A*A) for mode 0 / (num_pairs, num_rotations) for mode 1; out (A, Rq, C) in
 * the packing of q.  num_pairs <= 16. */
/* Gram matrices of packed rows: out (num_anchors, num_pairs, C, C) = X^T X over the rows [starts[p], starts[p] + lengths[p]) of x[a]
 * (x (num_anchors, rows, C), row stride C, anchor_stride floats between anchors); C = 128 or 256, HOST arrays, at most 16 pairs. */
int se3_gram_stack(const float* x, int num_anchors, int C, int64_t anchor_stride, const int64_t* starts, const int64_t* lengths, int num_pairs,
                   float* out, void* stream);
/* The sums_given statistics of the stack mode from per-pair Gram matrices: out (num_pairs, A, A) = factor * <gq[a, p], gk[e, p]>_F, gq / gk
 * (A, num_pairs, elements) contiguous (elements = C * C, a multiple of 4). */
int se3_gram_frobenius(const float* gq, const float* gk, int A, int num_pairs, int64_t elements, float factor, float* out, void* stream);
int se3_cross_eq_stack_fwd(const float* q, const float* k, const float* vt, const int64_t* q_starts, const int64_t* q_lengths,
                           const int64_t* k_starts, const int64_t* k_lengths, int num_pairs, int A, int C, int H,
                           int64_t q_anchor_stride, int64_t k_anchor_stride, int v_row_stride, int64_t v_anchor_stride, int mode,
                           const int64_t* trace_idx, int num_rotations, int sums_given, float* partial_workspace, float* mix,
                           float* weights, float* out, void* stream);
/* The same on the f16 matrix cores at f32 accuracy (head dimension 64, A <= 6, key starts and v_row_stride multiples of 16, sums_given
 * != 0; anything else is forwarded to se3_cross_eq_stack_fwd): q, k and vt are split once into f16 hi / lo pieces in `workspace`
 * (se3_cross_eq_x6_workspace_bytes bytes, 16-byte aligned; q (A, q_rows, C), k (A, k_rows, C) packed rows) and the three piece products
 * above 2^-22 are accumulated in f32.
 * out (A, q_rows, out_groups * C) with anchor stride out_anchor_stride floats: out_groups = 1 is the result itself (out_anchor_stride =
 * q_anchor_stride for an output in the layout of q).  out_groups = G > 1 (a divisor of A; few pairs, where one workgroup per (query tile,
 * head, anchor) leaves most of the chip idle): workgroup group g sums the key anchors [g, g + 1) * A / G and writes its partial result at
 * channel offset g * C; the result is the sum of the G channel blocks, which the caller gets for free from the output projection that
 * follows by stacking its weights G times along the input dimension.  (G > 1 is refused when the call has to take the f32 kernels.) */
size_t se3_cross_eq_x6_workspace_bytes(int A, int64_t q_rows, int64_t k_rows, int C, int v_row_stride);
int se3_cross_eq_stack_x6_fwd(const float* q, const float* k, const float* vt, const int64_t* q_starts, const int64_t* q_lengths,
                              const int64_t* k_starts, const int64_t* k_lengths, int num_pairs, int A, int C, int H, int64_t q_rows,
                              int64_t k_rows, int64_t q_anchor_stride, int64_t k_anchor_stride, int v_row_stride,
                              int64_t v_anchor_stride, int mode, const int64_t* trace_idx, int num_rotations, int sums_given,
                              float* partial_workspace, float* mix, float* weights, float* out, int out_groups,
                              int64_t out_anchor_stride, void* workspace, size_t workspace_bytes, void* stream);


/* ---- G1/G2: geometric structure embedding --------------------------------------------------------------------------
 * Replaces GeometricStructureEmbedding.forward (geotransformer/modules/geotransformer/geotransformer.py:57-121 with
 * transformer/positional_embedding.py:8-34).  points (N, 3); knn (N, 3) int64: the 3 nearest other points of each point.
 * table_d / table_a: (entries, C, 2) float32 = (f, df/dx) of f(x) = W emb(x) + b sampled at x = j / entries_per_unit
 * (built by the caller with two small GEMMs); indices beyond a table fall back to the exact sinusoid sum with
 * w_* (C, C), b_* (C,), div_term (C/2,).  emb (N, N, C); eq_emb (num_anchors, N, N, 4) or NULL, wigner_d1 (num_anchors, 3, 3).
 * workspace: se3_geo_embedding_workspace_bytes(N) bytes of device scratch (per-pair index records), or NULL: the single-kernel form that
 * needs none (2-3x slower: every table read then comes from L2). */
size_t se3_geo_embedding_workspace_bytes(int N);
int se3_geo_embedding_fwd(const float* points, const int64_t* knn, int N, int C, const float* table_d, int d_entries,
                          float d_entries_per_unit, const float* table_a, int a_entries, float a_entries_per_unit,
                          float sigma_d, float sigma_a, const float* w_d, const float* b_d, const float* w_a, const float* b_a,
                          const float* div_term, const float* wigner_d1, int num_anchors, float* emb, float* eq_emb,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Operands of the embedding's weight gradients (training step: autograd through proj_d / proj_a and the max over the 3 angles,
 * geotransformer.py:92-121): S (4, N, N, C) = the sinusoid embeddings of the distance index and of the three angle indices; dEk (3, N, N, C)
 * = grad_emb masked to the channels where angle k holds the maximum.  d proj_d.weight = grad_emb^T S[0], d proj_a.weight = sum_k
 * dEk[k]^T S[1 + k] (library GEMMs), both bias gradients = sum grad_emb. */
int se3_geo_embedding_bwd_operands(const float* points, const int64_t* knn, int N, int C, const float* table_a, int a_entries,
                                   float a_entries_per_unit, float sigma_d, float sigma_a, const float* w_a, const float* b_a,
                                   const float* div_term, const float* grad_emb, float* S, float* dEk, void* stream);
/* Builds / validates such a table on the device (replaces the caller-side GEMMs): table (entries, C, 2) float32 = (f, df/dx) of
 * f(x) = weight emb(x) + bias at x = j / entries_per_unit (float64 accumulation), emb = SinusoidalPositionalEmbedding
 * (transformer/positional_embedding.py:8-34) with div_term (C/2,).  state: se3_embedding_table_state_bytes() zero-initialised bytes owned by the caller next to
 * the table; the kernel keeps a content hash of (weight, bias, div_term, sizes) there and returns immediately when the table
 * already belongs to the current values -- call it in front of every se3_geo_embedding_*_fwd; no host synchronisation.  A (table, state)
 * pair has ONE writer: calls for it must be ordered (one stream); concurrent streams keep their own pair. */
size_t se3_embedding_table_state_bytes(void);
int se3_embedding_table_refresh(const float* weight, const float* bias, const float* div_term, int C, int entries,
                                float entries_per_unit, float* table, void* state, void* stream);
/* Same, emb (N, N, C) written as bfloat16 (round to nearest even of the float32 value); eq_emb stays float32. */
int se3_geo_embedding_bf16_fwd(const float* points, const int64_t* knn, int N, int C, const float* table_d, int d_entries,
                               float d_entries_per_unit, const float* table_a, int a_entries, float a_entries_per_unit,
                               float sigma_d, float sigma_a, const float* w_d, const float* b_d, const float* w_a,
                               const float* b_a, const float* div_term, const float* wigner_d1, int num_anchors, uint16_t* emb,
                               float* eq_emb, void* workspace, size_t workspace_bytes, void* stream);

/* ---- G1 / E3: nearest-neighbour selections on the superpoint level ----------------------------------------------------
 * se3_knn3: knn (N, 3) int64 = the 3 nearest OTHER points of every point (get_embedding_indices,
 * geotransformer/modules/geotransformer/geotransformer.py:69-90: topk(k + 1) of the distance map, first column dropped).
 * se3_point_to_node_partition replaces point_to_node_partition (geotransformer/modules/ops/pointcloud_partition.py:60-107):
 * point_to_node (N) int64 = nearest node of every point; node_masks (M) uint8 = node owns at least one point;
 * node_knn_indices (M, limit) int64 = the `limit` nearest of the node's OWN points in ascending distance, padded with N;
 * node_knn_masks (M, limit) uint8.  Distances as pairwise_distance (modules/ops/pairwise_distance.py:4-30), ties by index.
 * limit <= 128. */
int se3_knn3(const float* points, int N, int64_t* knn, void* stream);
/* se3_knn3 for num_clouds (<= 16) stacked clouds in one launch: points (sum lengths, 3), lengths HOST array; knn (sum lengths, 3)
 * holds indices LOCAL to the point's own cloud. */
int se3_knn3_stack(const float* points, const int64_t* lengths, int num_clouds, int64_t* knn, void* stream);
int se3_point_to_node_partition(const float* points, const float* nodes, int N, int M, int limit, int64_t* point_to_node,
                                uint8_t* node_masks, int64_t* node_knn_indices, uint8_t* node_knn_masks, void* stream);
/* Stack mode: the same partition for num_clouds (<= 16) clouds in one launch per kernel.  points / nodes are the stacked fine
 * points / superpoints of all clouds (cloud c: point_lengths[c] points, node_lengths[c] nodes; HOST arrays).  All outputs use
 * GLOBAL indices into the stacked arrays: point_to_node (total points) = nearest node of the point's own cloud;
 * node_knn_indices (total nodes, limit) padded with the total point count; masks as above. */
int se3_point_to_node_partition_stack(const float* points, const float* nodes, const int64_t* point_lengths,
                                      const int64_t* node_lengths, int num_clouds, int limit, int64_t* point_to_node,
                                      uint8_t* node_masks, int64_t* node_knn_indices, uint8_t* node_knn_masks, void* stream);

/* ---- E1: pairwise squared distances ---------------------------------------------------------------------------------
 * Replaces pairwise_distance (geotransformer/modules/ops/pairwise_distance.py:4-30, channel-last form): x (batch, N, C), y (batch, M, C)
 * with unit channel stride, row stride C and the given batch strides (floats; 0 = the same rows for every batch); out (batch, N, M)
 * contiguous = max(|x_n|^2 - 2 x_n.y_m + |y_m|^2, 0), or max(2 - 2 x_n.y_m, 0) with normalized != 0 (unit vectors).  Any C >= 1 (C = 3:
 * point coordinates), batch <= 65535; f32 matrix cores.  (The superpoint scores and the point-to-node partition have the distance
 * matrix fused into their own kernels: E2, E3.) */
int se3_pairwise_distance(const float* x, const float* y, int64_t batch, int N, int M, int C, int64_t x_batch_stride,
                          int64_t y_batch_stride, int normalized, float* out, void* stream);

/* ---- E2: superpoint matching scores --------------------------------------------------------------------------------
 * Replaces the score part of SuperPointMatching.forward (geotransformer/modules/geotransformer/superpoint_matching.py:31-39):
 * scores[n, m] = exp(-clamp(2 - 2 ref[n].src[m], 0)), optionally dual-normalised (S / rowsum * S / colsum).
 * ref (N, C), src (M, C) L2-normalised; workspace: N + M floats. */
int se3_superpoint_scores(const float* ref_feats, const float* src_feats, int N, int M, int C, int dual_normalization,
                          float* scores, float* workspace, void* stream);
/* Stack mode: the score matrices of num_pairs (<= 16) registration pairs in one launch per kernel.  feats (rows, C): the
 * L2-normalised superpoint features of all clouds; pair p: ref rows ref_rows[p] .. + ref_lengths[p], src rows src_rows[p] .. +
 * src_lengths[p]; node_masks (uint8): entries ref_mask_offsets[p] + n / src_mask_offsets[p] + m tell whether the node owns a
 * fine point -- nodes with mask 0 are absent (superpoint_matching.py:24-29 drops them before scoring): they do not enter the
 * normalisation sums and their scores are -1.  scores (num_pairs, score_stride): pair p's (N_p, M_p) matrix row-major at the
 * start of row p, -1 beyond it, so that ONE top-k over the rows selects per pair.  workspace: sum(N_p) + sum(M_p) floats.
 * All arrays are HOST arrays. */
int se3_superpoint_scores_stack(const float* feats, const uint8_t* node_masks, const int64_t* ref_rows, const int64_t* src_rows,
                                const int64_t* ref_lengths, const int64_t* src_lengths, const int64_t* ref_mask_offsets,
                                const int64_t* src_mask_offsets, int num_pairs, int C, int dual_normalization,
                                int64_t score_stride, float* scores, float* workspace, void* stream);

/* ---- F1: weighted Procrustes / inlier voting for local-to-global registration -------------------------------------------
 * Replace weighted_procrustes (geotransformer/modules/registration/procrustes.py:6-73, SVD on the CPU in the reference) and
 * the hypothesis voting of LocalGlobalRegistration (geotransformer/modules/geotransformer/local_global_registration.py:139-194).
 * Correspondences are stacked: src/ref (total, 3), scores (total,); problem s uses rows [segment_offsets[s],
 * segment_offsets[s+1]) (int64, DEVICE, num_segments + 1 entries).  If gate_transform (4x4 row-major, DEVICE) is not NULL the
 * weight of a correspondence is score * [ |ref - T src| < gate_radius ] (the refinement re-weighting).  Weights are
 * normalised by (sum + eps).  transforms: (num_segments, 4, 4).  se3_count_inliers: votes[b] = #{i : |ref_i - T_b src_i| < radius}. */
int se3_weighted_procrustes(const float* src_points, const float* ref_points, const float* scores,
                            const int64_t* segment_offsets, int num_segments, const float* gate_transform, float gate_radius,
                            float eps, float* transforms, void* stream);
int se3_count_inliers(const float* src_points, const float* ref_points, int64_t num_points, const float* transforms,
                      int num_transforms, float radius, int32_t* votes, void* stream);
/* Several registration pairs at once: gate_per_segment != 0 -> gate_transforms holds one (4, 4) per segment (the pair's
 * current estimate); range_begin / range_end (DEVICE int64, one per transform) restrict hypothesis t to the correspondences
 * [range_begin[t], range_end[t]) of its own pair. */
int se3_weighted_procrustes_segments(const float* src_points, const float* ref_points, const float* scores,
                                     const int64_t* segment_offsets, int num_segments, const float* gate_transforms,
                                     int gate_per_segment, float gate_radius, float eps, float* transforms, void* stream);
int se3_count_inliers_ranges(const float* src_points, const float* ref_points, int64_t num_points, const float* transforms,
                             int num_transforms, const int64_t* range_begin, const int64_t* range_end, float radius,
                             int32_t* votes, void* stream);

/* Mutual top-k correspondence mask (local_global_registration.py:104-131): mask[b, i, j] = 1 iff scores[b, i, j] is among the k
 * largest of row i AND of column j of patch pair b (ties by index), exceeds `threshold`, and row_masks[b, i] & col_masks[b, j].
 * scores (batch, rows, cols) float32, masks uint8; rows * cols <= 16384. */
int se3_mutual_topk_mask(const float* scores, const uint8_t* row_masks, const uint8_t* col_masks, int batch, int rows, int cols,
                         int k, float threshold, uint8_t* mask, void* stream);



/* grouping.anchor_query (grouping_cuda_kernel.cu:166-233, grouping_cuda.cpp:88-108): anchor_weights (b, np, na, ks, nn) =
 * (kw_k - r)^2 + ((kh_k - theta) r)^2 for the local neighbour coordinates grouped_xyz (b, 3, np, nn), r = |x| + 1e-6,
 * theta = acos(x . anchors[a] / r), kernel_points (ks, 2) = (radial, angular) positions.  (sample_idx, grouped_indices and nq of the
 * reference's signature are not read by its kernel.) */
int se3_vgtk_anchor_query(const float* grouped_xyz, const float* anchors, const float* kernel_points, int batch, int num_points,
                          int num_neighbors, int num_anchors, int kernel_size, float* anchor_weights, void* stream);
/* grouping.initial_anchor_query (grouping_cuda_kernel.cu:102-152, grouping_cuda.cpp:138-158): centers (b, 3, nc), xyz (m, 3) shared by the
 * batch, kernel_points (ks, na, 3): anchor_weights / anchor_counts (b, ks, nc, na) = sum / number over the points within `radius` of the
 * centre of the positive part of 1 - |kernel point + centre - point|^2 / sigma.  Summed in point order (the reference: atomicAdd). */
int se3_vgtk_initial_anchor_query(const float* centers, const float* xyz, const float* kernel_points, int batch, int num_centers,
                                  int num_points, int num_anchors, int kernel_size, float radius, float sigma, float* anchor_weights,
                                  float* anchor_counts, void* stream);
/* ---- EPN toolkit (vgtk) equivalents: SURVEY section 8f row 4 -------------------------------------------------------------------------
 * Replace the CUDA extensions vgtk.cuda.{gathering, grouping, zpconv} (geotransformer/modules/e2pn/vgtk/vgtk/cuda: gathering_cuda.cpp,
 * grouping_cuda.cpp, zpconv_cuda.cpp and their *_kernel.cu) entry for entry: same tensor layouts (channel-first, int32 indices), same
 * edge cases; float sums are gathers in a fixed order (the reference scatters with atomicAdd), errors come back as status codes, every
 * launch takes a stream.  Called from se3et_amd/vgtk.py (mirror of vgtk/spconv/functional.py, vgtk/pc/sample.py).
 *   gather_points: out[b, c, j] = points[b, c, idx[b, j]]; backward scatter-adds in ascending j           (gathering_cuda_kernel.cu:42-98)
 *   ball_query: the first nsample support indices (ascending) within `radius` of each query, a short list repeats itself, a list short
 *     by exactly one keeps a trailing 0, an empty one is all 0                                            (grouping_cuda_kernel.cu:52-99)
 *   furthest_point_sampling: idxs[b, 0] = 0, then iteratively the point furthest from the chosen set; points with |p|^2 <= 1e-3 are
 *     never chosen; exact ties resolve as in the reference's block reduction; temp_workspace (b, n) floats (grouping_cuda_kernel.cu:337-452)
 *   inter_zpconv: out[b, c, k, p, a] = sum_n w[b, p, a, k, n] feats[b, c, nbr[b, p, a, k, n], a]; backward = its transpose
 *     (workspace: se3_vgtk_inter_zpconv_bwd_workspace_bytes)                                               (zpconv_cuda_kernel.cu:32-116)
 *   intra_zpconv: out[b, c, k, p, a] = sum_n w[a, k, n] feats[b, c, p, nbr[a, n]]; backward = its transpose (zpconv_cuda_kernel.cu:119-195) */
int se3_vgtk_gather_points_fwd(const float* points, const int32_t* idx, int batch, int channels, int num_points, int num_indices,
                               float* out, void* stream);
int se3_vgtk_gather_points_bwd(const float* grad_out, const int32_t* idx, int batch, int channels, int num_points, int num_indices,
                               float* grad_points, void* stream);
int se3_vgtk_ball_query(const float* new_xyz, const float* xyz, int batch, int num_support, int num_queries, float radius, int nsample,
                        int32_t* idx, void* stream);
int se3_vgtk_furthest_point_sampling(const float* dataset, int batch, int num_points, int num_samples, float* temp_workspace,
                                     int32_t* idxs, void* stream);
int se3_vgtk_inter_zpconv_fwd(const int32_t* neighbors, const float* weights, const float* feats, int batch, int num_samples,
                              int num_support, int num_anchors, int kernel_size, int num_nn, int channels, float* out, void* stream);
size_t se3_vgtk_inter_zpconv_bwd_workspace_bytes(int batch, int num_samples, int num_support, int num_anchors, int kernel_size, int num_nn);
int se3_vgtk_inter_zpconv_bwd(const int32_t* neighbors, const float* weights, const float* grad_out, int batch, int num_samples,
                              int num_support, int num_anchors, int kernel_size, int num_nn, int channels, float* grad_feats,
                              void* workspace, size_t workspace_bytes, void* stream);
int se3_vgtk_intra_zpconv_fwd(const int32_t* neighbors, const float* weights, const float* feats, int batch, int num_points,
                              int anchors_in, int anchors_out, int kernel_size, int num_nn, int channels, float* out, void* stream);
int se3_vgtk_intra_zpconv_bwd(const int32_t* neighbors, const float* weights, const float* grad_out, int batch, int num_points,
                              int anchors_in, int anchors_out, int kernel_size, int num_nn, int channels, float* grad_feats, void* stream);


/* ---- host-pointer variants of geotransformer.ext for DataLoader worker processes -----------------------------------------
 * Same contract as ext.grid_subsampling / ext.radius_neighbors (extensions/pybind.cpp:6-18; CPU, contiguous float32 / int64 tensors):
 * plain host memory in and out, no GPU, no global state, callable from forked workers.  grid: s_points / s_normals hold n rows, the first
 * sum(s_lengths) are valid, in the reference's emission order.  radius: out == NULL computes *max_count only (the width the reference's
 * result would have); otherwise out is (nq, limit) int64, rows ascending in distance, padded with ns.  Bound by se3et_amd/ext.py. */
int se3_grid_subsample_host(const float* points, const float* normals, int64_t n, const int64_t* lengths, int batch, float voxel_size,
                            float* s_points, float* s_normals, int64_t* s_lengths);
int se3_radius_neighbors_host(const float* q_points, int64_t nq, const float* s_points, int64_t ns, const int64_t* q_lengths,
                              const int64_t* s_lengths, int batch, float radius, int64_t limit, int64_t* out, int64_t* max_count);

/* ---- round 4: the remaining library products of the inference forward as kernels --------------------------------------------------------
 * se3_patch_scores   experiments/se3ete.3dmatch/model.py:186-203 -- the fine-matching score matrices of all patch pairs with the two feature
 *                    gathers fused: out (num_patches, K, K) = <feats[ref_idx[b, n]], feats[src_idx[b, m]]> * scale; an index outside
 *                    [0, num_rows) selects a zero row (the reference's padded feature row); K = patch_points in {64, 128}, C % 64 == 0.
 * se3_anchor_mix_stack  conditional_transformer.py:209-249 (eq2inv_soft) for all pairs of a batch: out[a, r, :] = sum_e mix[p(r)][a, e]
 *                    x[e, r, :] for the packed rows r of pair p (rows of no pair: zero); x, out (6, rows, channels), mix (pairs, 6, 6). */
int se3_patch_scores(const float* feats, const int64_t* ref_idx, const int64_t* src_idx, int64_t num_patches, int patch_points,
                     int64_t num_rows, int C, float scale, float* out, void* stream);
int se3_anchor_mix_stack(const float* x, int64_t rows, int channels, const float* mix, const int64_t* starts_host, const int64_t* lengths_host,
                         int num_pairs, float* out, void* stream);

/* ---- D7 / D8: the coarse transformer of a batch of pairs, every launch issued from C ---------------------------------------------------------
 * RPEConditionalTransformer.forward (conditional_transformer.py:251-390) with the layers it schedules (rpe_transformer.py:134-194,
 * vanilla_transformer.py:872-946, output_layer.py:7-47) and GeometricTransformer's out_proj (geotransformer.py:310-317) as ONE host call:
 * about 130 launches on the caller's stream, no interpreter between them (one pair per forward was bound by ~480 Python-issued launches).
 * The plan is a HOST structure of device pointers: weights as se3_linear_split_weights_f16 pieces (+ f32 biases), the per-cloud geometric
 * embeddings, the packed-row layout (refs of all pairs first, then srcs; cloud starts multiples of 32).  x_in (A, rows, C) = in_proj of the
 * superpoint features in that layout, out (rows, out_proj.out_features).  Block types: 0 'self', 1 'self_eq', 2 'cross', 3 'cross_a_soft',
 * 4 'cross_r_soft' (followed by eq2inv_soft + RotCompressOutput when the next block is invariant).  Supported: the SE3ET-E / -E2 and
 * SE3ET-I / -I2 / KITTI block lists, A = 6, head dimension 64 for the key-anchor groups of the equivariant cross attention. */
#define SE3_MAX_BLOCKS 16
typedef struct {
  const void* pieces;       /* se3_linear_split_weights_f16 of the (out_features, in_features) weight */
  const float* bias;        /* (out_features) or NULL */
  int in_features, out_features;
} se3_linear_t;
typedef struct {
  int type;
  int off_q, off_k, off_qp, off_qe;        /* self blocks: column offsets of [q | k | W_p^T q | W_eq^T q] in the stacked projection (off_qe < 0: none) */
  se3_linear_t stack;                      /* self blocks: the composed projection */
  se3_linear_t q, k, v;                    /* cross blocks: proj_q, proj_k; all blocks: proj_v */
  se3_linear_t out;                        /* attention.linear (its bias is applied by the LayerNorm kernel) */
  const void* out_pieces_g2;               /* equivariant cross blocks: the same weight repeated 2 / 3 times along in_features (key-anchor groups) */
  const void* out_pieces_g3;
  const float* ln1_w;
  const float* ln1_b;
  float ln1_eps;
  se3_linear_t expand, squeeze;            /* AttentionOutput (the squeeze bias is applied by the LayerNorm kernel) */
  const float* ln2_w;
  const float* ln2_b;
  float ln2_eps;
  const int64_t* trace_idx;                /* equivariant cross blocks: (num_rotations, 6) int64 on the device */
  int num_rotations;
} se3_layer_t;
typedef struct {
  int A, C, H, num_blocks, num_pairs, emb_bf16;
  se3_layer_t layers[SE3_MAX_BLOCKS];
  se3_linear_t rc_expand, rc_squeeze;      /* RotCompressOutput (only read behind a cross_r_soft block) */
  const float* rc_ln_w;
  const float* rc_ln_b;
  float rc_ln_eps;
  se3_linear_t out_proj;
  int64_t starts[SE3_MAX_BATCH], lengths[SE3_MAX_BATCH];   /* packed row start / length of every cloud: ref0 .. ref(B-1), src0 .. src(B-1) */
  int64_t rows0, rows;                     /* packed rows of all refs; of all clouds */
  const float* emb[SE3_MAX_BATCH];         /* (N_c, N_c, C) geometric embedding of cloud c (bfloat16 when emb_bf16) */
  const float* eq[SE3_MAX_BATCH];          /* (A, N_c, N_c, 4) equivariant embedding or NULL */
} se3_transformer_plan_t;
size_t se3_transformer_workspace_bytes(const se3_transformer_plan_t* plan);
/* layout[6] = sizeof(se3_linear_t), sizeof(se3_layer_t), sizeof(se3_transformer_plan_t), offsetof(plan, layers), offsetof(plan, starts),
 * offsetof(plan, emb): a binding in another language checks its mirror of the structs against the library it loaded. */
void se3_transformer_plan_layout(size_t* layout);
int se3_transformer_forward(const se3_transformer_plan_t* plan, const float* x_in, float* out, void* workspace, size_t workspace_bytes,
                            void* stream);
int se3_linear_stream_segments(const float* x, int64_t rows, int in_features, int seg_channels, int64_t seg_stride, const void* weight_pieces,
                               const float* bias, int out_features, int apply_relu, float* out, int64_t out_row_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SE3ET_HIP_H_ */
