/* presight_hip.h — C ABI of libpresight_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the NeRF prior-builder hot path of PreSight (ray sampling -> multires hash
 * grid -> tiny MLP -> volumetric rendering, forward and backward, plus the dense field query used
 * for prior extraction).  The reference reaches this functionality through nerfstudio's operator
 * seam: the `tcnn_encoding` backend object of HashEncoding / SHEncoding / MLP selected by the
 * `implementation` string (ns/field_components/encodings.py:288-310,694-706, mlp.py:100-136), and
 * through plain torch ops for samplers/renderers.  A maintainer binds these symbols with ctypes
 * (see INTEGRATION.md); no torch types cross the boundary.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; buffers are caller-allocated
 *     (torch caching allocator), contiguous, fp32 unless stated otherwise; no ownership transfer
 *   - `stream` is a hipStream_t; all work is asynchronous on it, no host synchronisation inside
 *   - return value 0 = ok, non-zero = error (ps_last_error() gives the text); callers raise
 *   - "ns/" = /root/reference/nerfstudio-0.3.3/nerfstudio
 */
#ifndef PRESIGHT_HIP_H
#define PRESIGHT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int ps_abi_version(void);
const char* ps_last_error(void);
int ps_device_info(int* cu_count, int* wave_size, char* arch /*host*/, int arch_len);

/* ---- a7 HashEncoding (torch-fallback semantics), ns/field_components/encodings.py:324-384 ------
 * x [N,3]; table [L*2^log2T, F] level-major; scalings [L] (floor(min_res*g^l), fp32); out [N, L*F]. */
int ps_hashgrid_fwd(const float* x, const float* table, const float* scalings, int L, int F, int log2T, int64_t N,
                    float* out, void* stream);
/* dtable += d(out)/d(table)^T dout.  Operator-level reference kernel: global fp32 atomics (dtable must be pre-zeroed or hold the
 * running gradient); the training path uses the binned fixed-point scatter ps_grid_scatter_binned below instead. */
int ps_hashgrid_bwd(const float* x, const float* dout, const float* scalings, int L, int F, int log2T, int64_t N,
                    float* dtable, void* stream);
/* idx [N, L, 8] int64: the 8 corner rows per level in the reference's corner order (encodings.py:354-361) */
int ps_hashgrid_indices(const float* x, const float* scalings, int L, int log2T, int64_t N, int64_t* idx, void* stream);

/* ---- a8 MLP, ns/field_components/mlp.py:138-174 ------------------------------------------------
 * Supported shapes: see ps_mlp_shape_supported.  Weights are used in a packed "fragment order":
 *   packed = [forward block of each layer][transposed block of each layer]   (ps_mlp_sizes)
 * built from the torch-layout tensors by one ps_mlp_pack_layer call per layer. */
int ps_mlp_shape_supported(int in_dim, int hidden, int out_dim, int num_layers);
int ps_mlp_sizes(int in_dim, int hidden, int out_dim, int num_layers, int64_t N, int64_t* packed_floats /*host*/,
                 int64_t* grad_floats /*host*/, int* n_parts /*host*/);
/* W [out,in], b [out]; colmap int32 [KS*4] (input column per (k-step, lane group), -1 = pad) */
int ps_mlp_pack_layer(const float* W, const float* b, int out_dim, int in_dim, const int* colmap, int KS, int NB,
                      float* fw_block, float* wt_block, void* stream);
/* gW [out,in] += sum_parts dW ; gb [out] += sum_parts db */
int ps_mlp_unpack_grad_layer(const float* gpart, int n_parts, int64_t part_stride, int out_dim, int in_dim,
                             const int* colmap, int KS, int NB, float* gW, float* gb, void* stream);
/* The same two operations for all layers (<= 8) of one fused stack in ONE launch.  Every array argument is a HOST
 * array of n_layers entries (device pointers / ints); gpart[i] points at layer i's block inside the first partial. */
int ps_mlp_pack_layers(int n_layers, const float* const* W, const float* const* b, const int* out_dim, const int* in_dim,
                       const int* const* colmap, const int* KS, const int* NB, float* const* fw_block,
                       float* const* wt_block, void* stream);
int ps_mlp_unpack_grad_layers(int n_layers, const float* const* gpart, int n_parts, int64_t part_stride,
                              const int* out_dim, const int* in_dim, const int* const* colmap, const int* KS,
                              const int* NB, float* const* gW, float* const* gb, void* stream);
/* y [N,out] = MLP(x [N,in]); out_act: 0 none, 1 sigmoid */
int ps_mlp_fwd(const float* x, const float* packed, float* y, int64_t N, int in_dim, int hidden, int out_dim,
               int num_layers, int out_act, void* stream);
/* recomputes the hidden activations from x; dx may be NULL; gpart [n_parts, grad_floats] is overwritten */
int ps_mlp_bwd(const float* x, const float* dy, const float* packed, float* dx, float* gpart, int64_t N, int in_dim,
               int hidden, int out_dim, int num_layers, int out_act, void* stream);

/* ---- a6 / a10 / a5 / a4 point-wise operators --------------------------------------------------- */
/* ns/fields/PreSight/ingp_field.py:169-177: aabb [2,3]; u [M,3]; sel uint8 [M] (may be NULL) */
int ps_contract(const float* p, const float* aabb, int64_t M, int contract, float* u, uint8_t* sel, void* stream);
/* SH degree 4 of (d+1)/2, ns/utils/math.py:27-79 via ns/fields/base_field.py:136-142; out [M,16] */
int ps_sh4(const float* dirs, int64_t M, float* out, void* stream);
/* SHEncoding.forward (ns/field_components/encodings.py:711-719): real SH of `levels` (1..4) levels evaluated on x [M,3] AS
 * GIVEN (the fields pass (d+1)/2); out [M, levels^2] */
int ps_sh_encode(const float* x, int64_t M, int levels, float* out, void* stream);
/* argmin_k ||p - c_k||, ns/fields/PreSight/ingp_field_ms.py:97; assign int32 [M] */
int ps_route(const float* p, int64_t M, const float* centroids, int K, int32_t* assign, void* stream);
/* o + d*(start+end)/2, ns/cameras/rays.py:49-58; ebins [R,S+1]; pos [R*S,3] */
int ps_sample_positions(const float* origins, const float* dirs, const float* ebins, int64_t R, int S, float* pos,
                        void* stream);

/* ---- a1 / a3 / a12 / a13 / a15 per-ray operators ----------------------------------------------- */
/* ns/cameras/cameras.py:497-880 (pinhole): ray_indices int64 [R,3] = (cam,row,col); c2w [C,3,4] */
int ps_generate_rays(const int64_t* ray_indices, const float* c2w, const float* fx, const float* fy, const float* cx,
                     const float* cy, int64_t R, float* origins, float* dirs, float* pixel_area, float* dir_norm,
                     void* stream);
/* ns/model_components/ray_samplers.py:78-128 with the piecewise spacing of nerfacto_nusc_ms.py:312-317.
 * jitter [R] (training, single jitter) or NULL (eval); sbins/ebins [R,S+1] */
int ps_spaced_bins(const float* jitter, int64_t R, int S, float near, float far, float thr, float* sbins, float* ebins,
                   void* stream);
/* ns/cameras/rays.py:128-150; sigma/weights [R,S] */
int ps_weights_fwd(const float* ebins, const float* sigma, int64_t R, int S, float* weights, void* stream);
int ps_weights_bwd(const float* ebins, const float* sigma, const float* dweights, int64_t R, int S, float* dsigma,
                   void* stream);
/* ns/model_components/ray_samplers.py:305-372 (include_original=False, single jitter); weights are raised
 * to `anneal` first (ray_samplers.py:597).  new_* [R, n_new+1] */
int ps_pdf_resample(const float* weights, const float* sbins, const float* jitter, int64_t R, int S, int n_new,
                    float anneal, float pad, float eps, float near, float far, float thr, float* new_sbins,
                    float* new_ebins, void* stream);
/* ns/model_components/renderers.py:70-117,286-383 + nerfacto_nusc_ms.py:530.  rgb_s [R,S,3], sem_s [R,S,C];
 * outputs may be NULL; minmax[2] (pre-set to {+inf, 0}) receives the batch-global min/max sample midpoint */
int ps_composite_fwd(const float* weights, const float* ebins, const float* rgb_s, const float* sem_s, int64_t R, int S,
                     int C, float threshold, float* rgb, float* acc, float* depth, float* exp_depth, float* sem,
                     float* minmax, void* stream);
int ps_clip(float* v, int64_t n, const float* minmax, float* keep /* nullable [n]: 1 where the value was inside the range, else 0 =
                                                                     the clip's derivative for the backward */, void* stream);
int ps_composite_bwd(const float* weights, const float* ebins, const float* rgb_s, const float* sem_s, const float* d_rgb,
                     const float* d_acc, const float* d_sem, const float* d_exp, int64_t R, int S, int C,
                     float* d_weights, float* d_rgb_s, float* d_sem_s,
                     const float* d_weights_add0 /* nullable [R,S]: gradients that reach the weights from elsewhere (the semantic
                                                    branch of the factored node, losses that act on the weights: distortion, line of
                                                    sight -- ns/models/PreSight/nerfacto_nusc_ms.py:586-612) are added to d_weights
                                                    here, add0 first, instead of by two element-wise launches */,
                     const float* d_weights_add1 /* nullable */, void* stream);

/* ---- a16 per-ray losses: value per ray + gradient w.r.t. the weights in one pass ---------------------
 * distortion (ns/model_components/losses.py:130-149): sbins [R,S+1], w [R,S] -> per_ray [R], dw [R,S] */
int ps_distortion_loss(const float* sbins, const float* w, int64_t R, int S, float* per_ray, float* dw, void* stream);
/* z-anti-aliased interlevel loss (ns/model_components/PreSight/losses.py:127-206): final-level histogram
 * (c [R,S+1], w [R,S]) blurred with `pulse_width`, resampled on the proposal edges cp [R,Sp+1] and compared with
 * wp [R,Sp]; per_ray [R] = sum_k max(ws-wp,0)^2/(wp+1e-5), dwp [R,Sp] its gradient */
int ps_interlevel_loss(const float* c, const float* w, const float* cp, const float* wp, int64_t R, int S, int Sp,
                       float pulse_width, float* per_ray, float* dwp, void* stream);

/* depth supervision of the lidar / monodepth configs (ns/model_components/PreSight/losses.py:28-103, called from
 * nerfacto_nusc_ms.py:576-629).  depth [R] metres, sky [R] or NULL (1 = sky), ebins [R,S+1] / pred [R] in scene
 * units (divided by pose_scale inside).  keep[r] = 1 < depth < upper_bound (and not sky); per_ray, dw, dpred are zero
 * for the other rays; the reference's masked mean is sum(per_ray) / sum(keep) (NaN when no ray is kept). */
int ps_line_of_sight_loss(const float* w, const float* ebins, const float* depth, const float* sky, int64_t R, int S,
                          float sigma, float upper_bound, float pose_scale, float* per_ray, float* dw, float* keep,
                          void* stream);
int ps_expected_depth_loss(const float* depth, const float* pred, const float* sky, int64_t R, float upper_bound,
                           int inverse, float pose_scale, float* per_ray, float* dpred, float* keep, void* stream);

/* ---- per-ray tail of the training step (one launch per operator and direction) ---------------------------
 * nn.Embedding lookup into a column block of out [R, out_stride] and its scatter-add backward
 * (ns/field_components/embedding.py:27-55; idx int64 [R], table [rows, D]) */
int ps_embedding_fwd(const int64_t* idx, const float* table, int64_t R, int D, int out_stride, int col0, float* out,
                     void* stream);
int ps_embedding_bwd(const int64_t* idx, const float* dout, int64_t R, int D, int rows, int out_stride, int col0,
                     float* dtable /* += */, void* stream);
/* two nn.Embedding lookups concatenated (ns/field_components/embedding.py:27-55; the model's appearance + video codes,
 * nerfacto_nusc_ms.py:472-485) in ONE launch per direction: out [R, D0+D1] = [table0[idx0[r*stride0]] | table1[idx1[r*stride1]]],
 * indices int64 read through an element stride (the camera index is column 0 of ray_indices [R,3]); bwd: dtable_k += scatter of
 * dout's columns (small tables pre-reduced in LDS per workgroup, as ps_embedding_bwd) */
int ps_embedding_pair_fwd(const int64_t* idx0, int64_t stride0, const float* table0, int D0, const int64_t* idx1, int64_t stride1,
                          const float* table1, int D1, int64_t R, float* out, void* stream);
int ps_embedding_pair_bwd(const int64_t* idx0, int64_t stride0, int rows0, int D0, float* dtable0, const int64_t* idx1, int64_t stride1,
                          int rows1, int D1, float* dtable1, const float* dout, int64_t R, void* stream);
/* sky blending, ns/models/PreSight/nerfacto_nusc_ms.py:512-533: acc = clamp(acc_raw, 0, 1),
 * rgb = rgb_f + (1-acc) sky_rgb, sem = sem_f + (1-acc) sky_sem (sky_* / sem may be NULL).  Backward: d(rgb_f) = d(rgb)
 * and d(sem_f) = d(sem) are identities (not written); d_acc_raw [R], d_sky_rgb [R,3], d_sky_sem [R,C] are. */
int ps_sky_blend_fwd(const float* rgb_f, const float* acc_raw, const float* sem_f, const float* sky_rgb,
                     const float* sky_sem, int64_t R, int C, float* rgb, float* acc, float* sem, void* stream);
int ps_sky_blend_bwd(const float* acc_raw, const float* sky_rgb, const float* sky_sem, const float* d_rgb,
                     const float* d_acc, const float* d_sem, int64_t R, int C, float* d_acc_raw, float* d_sky_rgb,
                     float* d_sky_sem, void* stream);
/* scalar losses: partial [ps_loss_partials(n)] receives per-workgroup sums (the caller adds them and divides by n),
 * dpred / dacc the gradient of the MEAN.  MSE: nn.MSELoss (nerfacto_nusc_ms.py:568) and semantic_loss with
 * clip_target=1 (ns/model_components/PreSight/losses.py:117-125); sky BCE: losses.py:106-115 */
int ps_loss_partials(int64_t n);
/* the scalar arithmetic around a loss term in one launch each (the reference spends ~6 element-wise launches per term on it:
 * .mean() / sum()/count, the *_loss_mult scaling of nerfacto_nusc_ms.py:558-645 and their backward nodes):
 * ps_loss_finish: out[0] = scale * sum(terms[0..n)) / D, D = sum(keep[0..n_keep)) if keep != NULL else denom; inv[0] = scale / D
 * (nullable).  ps_scale_grad: out[i] = grad[i] * g[0] * (factor ? factor[0] : 1) * host_scale -- g is the device scalar
 * autograd passes to the loss node, factor the inv[] of ps_loss_finish. */
int ps_loss_finish(const float* terms, int64_t n, const float* keep, int64_t n_keep, float denom, float scale, float* out,
                   float* inv, void* stream);
int ps_scale_grad(const float* grad, int64_t n, const float* g, const float* factor, float host_scale, float* out, void* stream);
int ps_mse_loss(const float* pred, const float* target, int64_t n, int clip_target, float* partial, float* dpred,
                void* stream);
int ps_sky_bce_loss(const float* acc, const float* sky_mask, int64_t R, float eps, float* partial, float* dacc,
                    void* stream);

/* data feed: collate one batch from a device-resident ImageChunk (ns/data/PreSight/my_dataset.py:28-73): pick int64 [R]
 * = drawn pixel slots; chunk arrays rgbs [P,3], skies/depths [P] (nullable), features [P,C] (nullable), pixel_indices /
 * image_indices / video_ids / widths int64 [P]; ray_indices [R,3] = (image, pixel // width, pixel % width) */
int ps_gather_batch(const int64_t* pick, int64_t R, const float* rgbs, const float* skies, const float* depths,
                    const float* features, int C, const int64_t* pixel_indices, const int64_t* image_indices,
                    const int64_t* video_ids, const int64_t* widths, int64_t* ray_indices, float* o_rgb, float* o_sky,
                    float* o_depth, float* o_feat, int64_t* o_video, void* stream);

/* ---- field level (fused) -------------------------------------------------------------------------
 * A field evaluation is:  ps_field_points -> ps_grid_encode -> ps_{prop,main}_field_fwd, and backward
 * ps_{prop,main}_field_bwd -> ps_grid_scatter.  Features travel as level planes feat[l][n][f]
 * (plane_stride floats between levels).  Semantics: ns/fields/PreSight/ingp_field.py:168-237,
 * prop_density_field.py:129-153, ns/cameras/rays.py:49-58, ns/field_components/encodings.py:343-384. */
/* positions (pos [N,3]) or rays (origins/dirs [R,3], ebins [R,S+1], point n = ray n/S sample n%S);
 * u [N,3] normalised+contracted+masked, sel [N] (1.0 / 0.0) */
int ps_field_points(const float* pos, const float* origins, const float* dirs, const float* ebins, int S,
                    const float* aabb, int contract, int64_t N, float* u, float* sel, void* stream);
/* slice_counts (nullable): uint32 [L, ps_grid_scatter_slices(F, log2T)], zeroed and filled with the number of records
 * the binned table backward will emit per (level, table slice) for these points (upper bound) -- hand it to
 * ps_grid_scatter_binned to save the backward its counting pass (training forward only) */
int ps_grid_encode(const float* u, const float* table, const float* scalings, int L, int F, int log2T, int64_t N,
                   int64_t plane_stride, float* feat, uint32_t* slice_counts, void* stream);
int ps_grid_scatter_slices(int F, int log2T);
/* table gradient without HBM atomics (LDS slice owners); accumulate=0 overwrites dtable, 1 adds to it */
int ps_grid_scatter(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t N,
                    int64_t plane_stride, float* dtable, int accumulate, void* stream);
/* Fast path of the same gradient (used by the fused fields): every corner contribution is computed once, binned
 * through `workspace` (ps_grid_scatter_workspace bytes, caller-allocated device memory) into the record stream of the
 * table slice that owns its row, and reduced with fixed-point int64 LDS atomics -> bit-reproducible;
 * accumulate=0 overwrites dtable, 1 adds to it, 2 adds to a dtable the caller guarantees to be all zero (first contribution
 * of a step into a zeroed gradient buffer: written without reading; slices without records are left untouched). */
int64_t ps_grid_scatter_workspace(int L, int F, int log2T, int64_t N);
int ps_grid_scatter_binned(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t N,
                           int64_t plane_stride, float* dtable, int accumulate, const uint32_t* slice_counts /*nullable*/,
                           int absmax_ready /* workspace[0..L) already holds the level maxima, see ps_prop_field_bwd */,
                           void* workspace, void* stream);
/* proposal field: MLP (L*F -> hidden -> 1), packed with ps_mlp_pack_layer (LINEAR first-layer colmap) */
int ps_prop_field_sizes(int LF, int hidden, int64_t N, int64_t* packed_floats /*host*/, int64_t* grad_floats /*host*/,
                        int* n_parts /*host*/);
int ps_prop_field_fwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel,
                      const float* packed, int64_t N, float* sigma, void* stream);
int ps_prop_field_bwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel,
                      const float* packed, const float* dsigma, int64_t N, float* dfeat, float* gpart,
                      uint32_t* level_absmax /* nullable: [L] float bits of max |dfeat| per level, zeroed + filled */, void* stream);
/* main field: packed = [base | semantic head | colour head]; offsets[6] = packed offsets of the three MLPs then
 * their gradient-block offsets.  dirs [R,3], app [R,A] (A <= 16, may be NULL), point n belongs to ray n/S.
 * Outputs sigma [N], rgb [N,3], sem [N,64]; any of them may be NULL to skip that head. */
int ps_main_field_sizes(int LF, int hidden, int hidden_color, int64_t N, int64_t* packed_floats /*host*/,
                        int64_t* grad_floats /*host*/, int* n_parts /*host*/, int64_t* offsets /*host [6]*/);
/* inference forward with a gated semantic head (prior extraction: ns/scripts/extract_priors.py:133-150 drops the points whose mean
 * density stays below the threshold): sigma [N] for every point; sem [N,64] only for the 32-point tiles in which some point has
 * (gate_a[n] + gate_b[n] + sigma[n]) / 3 >= gate_threshold -- the other rows are left UNWRITTEN.  gate_a / gate_b [N]: the two
 * proposal fields' densities.  Pass the caller's threshold lowered by a few ulp.  The kernel evaluates the MERGED network: base
 * output rows 16..79 folded into the semantic head's first layer (ps_merge_linear_fwd), packed = [base (L*F -> hidden -> 16) |
 * semantic head (hidden -> 64 merged, 64 -> 64, 64 -> 64) | colour head (unused)] with the sizes of ps_main_field_gated_sizes.
 * Densities bit-identical to ps_main_field_fwd, semantics equal to fp32 rounding. */
int ps_main_field_gated_sizes(int LF, int hidden, int hidden_color, int64_t* packed_floats /*host*/, int64_t* offsets /*host [3]*/);
int ps_main_field_fwd_gated(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color, const float* sel,
                            const float* packed, int64_t N, const float* gate_a, const float* gate_b, float gate_threshold,
                            float* sigma, float* sem,
                            unsigned long long* gate_stats /* nullable, device [2]: += (32-point tiles whose semantic head ran, tiles visited) */,
                            void* stream);
/* the same gated query for the K routed sub-fields of a production tile (ns/fields/PreSight/ingp_field_ms.py:97-126 evaluated by
 * ns/scripts/extract_priors.py:133-138): feat / sel in the sorted layout of ps_ms_route (n_slots), packed = K blocks of the
 * ps_main_field_gated_sizes layout back to back, gate_a / gate_b / sigma / sem in the CALLER's point order (reached through perm). */
int ps_main_field_fwd_gated_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color, const float* sel,
                               const float* packed, int64_t n_slots, const float* gate_a, const float* gate_b, float gate_threshold,
                               float* sigma, float* sem, unsigned long long* gate_stats /* as above */, const int32_t* perm,
                               const int32_t* field_start, int K, void* stream);
int ps_main_field_fwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                      const float* sel, const float* dirs, const float* app, int S, int A, const float* packed, int64_t N,
                      float* sigma, float* rgb, float* sem, float* acts /* nullable: [ceil(N/16)*16, ps_main_field_act_width], register order */,
                      void* stream);
/* acts: the hidden activations of the three MLPs, written by the training forward and read by ps_main_field_bwd instead of
 * recomputing the forward (a third of its matrix ops, half of its weight-fragment traffic; 1.6 KB per point at cfg 2) */
int ps_main_field_act_width(int LF, int hidden, int hidden_color);
/* weights == NULL: drgb [N,3] and dsem [N,64] are per-sample gradients.  weights [N] (the compositing weights of
 * RaySamples.get_weights, point n = ray n/S): drgb [R,3] / dsem [R,64] are the gradients of the COMPOSITED per-ray outputs
 * and the kernel forms weights[n] * d[n/S] itself -- the d_rgb_s / d_sem_s outputs of ps_composite_bwd are then not
 * needed (pass NULL there). */
int ps_main_field_bwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                      const float* sel, const float* dirs, const float* app, int S, int A, const float* packed,
                      const float* dsigma, const float* drgb, const float* dsem, const float* weights /*nullable*/, int64_t N,
                      float* dfeat, float* dapp, float* gpart, const float* acts /* nullable: recompute */,
                      float* dzb_scratch /* nullable; with acts: [ceil(N/16)*16, 80] workspace -> the three-kernel backward
                                            (semantic head, colour head, base MLP: one stack's weight gradients stay in
                                            registers and its transposed weights in LDS); NULL: one fused kernel */,
                      int stages /* three-kernel backward: which kernels THIS call launches -- bit 0 semantic head, bit 1 colour
                                    head, bit 2 base MLP; 7 = all.  A measurement aid (bench.py times the kernels one by one from
                                    three calls); the stages of one backward must run in this order */,
                      void* stream);


/* ---- a5 sub-field router: all K sub-fields of a tile in ONE launch per kernel, no host synchronisation ---------------
 * Reference: iNGPFieldMS / PropNetDensityFieldMS / SkyFieldMS (ns/fields/PreSight/ingp_field_ms.py:97-126,
 * prop_density_field_ms.py:90-102, sky_field_ms.py:97-114) route with cdist().argmin() and loop over the sub-fields with
 * boolean masks (4*K masked index ops and K host syncs per call).  Here ps_ms_route sorts the points by sub-field (stable)
 * into a PADDED layout — every sub-field's group starts on a chunk boundary of ps_ms_chunk() points — described by
 *   field_start [K+1]  first chunk of every group,  chunk_field [chunks]  sub-field of a chunk (-1 unused),
 *   perm [slots]       sorted slot -> point index in the caller's order (-1 padding),
 * and the *_ms entry points below evaluate every group with ITS sub-field's table / AABB / MLP weights.  Their per-point
 * working arrays (u, sel, feature planes, kept activations, d(features)) are in sorted order (`n_slots` rows); densities,
 * colours, semantics and their gradients stay in the caller's order (reached through perm). */
int ps_ms_chunk(void);
/* out[0] = int32 words of the plan buffer, out[1] / out[2] = word offsets of field_start / chunk_field inside it,
 * out[3] = slots of the sorted layout (= length of perm), out[4] = chunks */
int ps_ms_layout(int64_t N, int K, int64_t* out /*host[5]*/);
/* positions (pos [N,3]) or rays (origins/dirs [R,3], ebins [R,S+1], point n = ray n/S sample n%S); centroids [K,3] */
int ps_ms_route(const float* pos, const float* origins, const float* dirs, const float* ebins, int S, int64_t N,
                const float* centroids, int K, int32_t* plan, int32_t* perm, void* stream);
/* ns/fields/PreSight/ingp_field.py:169-177 per slot with the AABB of the slot's sub-field: aabbs [K,2,3]; u [slots,3], sel [slots] */
int ps_ms_field_points(const float* pos, const float* origins, const float* dirs, const float* ebins, int S, const float* aabbs,
                       int contract, int64_t N, int K, const int32_t* plan, const int32_t* perm, float* u, float* sel,
                       void* stream);
/* out[perm[i], :] = sorted[i, :] for the occupied slots */
int ps_ms_unsort(const float* sorted, const int32_t* perm, int64_t n_slots, int width, float* out, void* stream);
/* hash grid: tables / dtables = DEVICE arrays of K pointers; slice_counts [K, L, slices]; the gradient is ADDED to dtables[k] */
int ps_grid_encode_ms(const float* u, const float* const* tables, const float* scalings, int L, int F, int log2T, int64_t n_slots,
                      int64_t plane_stride, float* feat, uint32_t* slice_counts, int K, const int32_t* chunk_field, void* stream);
int64_t ps_grid_scatter_workspace_ms(int L, int F, int log2T, int64_t n_slots, int K);
int ps_grid_scatter_binned_ms(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t n_slots,
                              int64_t plane_stride, float* const* dtables, int K, const int32_t* chunk_field,
                              const uint32_t* slice_counts, int absmax_ready, void* workspace,
                              int dst_is_zero /* the caller guarantees dtables[*] hold zeros (first contribution of the step): the
                                                 flush then writes instead of read-modify-writing;
                                                 slices without records stay untouched */,
                              void* stream);
/* The binned table backward in PIECES, for a gradient that is exchanged bucket by bucket while the backward is still running
 * (the reference's DDP overlaps its bucketed all-reduce with backward: ns/pipelines/PreSight/my_pipeline.py:121-124): phase 1 = prepare
 * (counts, stream offsets, record write pass), phase 2 = accumulate the items [item_begin, item_end) of ps_grid_scatter_items()
 * = K * L * slices, ordered (sub-field, level, slice) like the gradient in memory; the same arguments in every call of one scatter.
 * phase | 8 (in every call of the scatter): the coarsest level of a single table is summed per cell corner into a dense int64 histogram
 * in the workspace instead of travelling as records (its (ceil(scalings[0]) + 1)^3 cells fit on chip: 1 / L of the record traffic and
 * the accumulate pass's hot rows gone; same fixed-point terms, bit-identical gradient).  phase | 4 overrides it: EVERY level as records
 * (a caller that consumes the records themselves -- the sparse exchange below). */
int ps_grid_scatter_items(int L, int F, int log2T, int K);
int ps_grid_scatter_binned_part(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t N,
                                int64_t plane_stride, float* dtable, int accumulate, const uint32_t* slice_counts, int absmax_ready,
                                void* workspace, int phase, int item_begin, int item_end, void* stream);
int ps_grid_scatter_binned_ms_part(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t n_slots,
                                   int64_t plane_stride, float* const* dtables, int K, const int32_t* chunk_field,
                                   const uint32_t* slice_counts, int absmax_ready, void* workspace, int dst_is_zero, int phase,
                                   int item_begin, int item_end, void* stream);
/* Sparse gradient exchange of a data-parallel run (round 6; replaces DDP's dense all-reduce of the table gradients,
 * ns/pipelines/PreSight/my_pipeline.py:121-124, for the hash tables): after a phase-1 call the binned backward's record streams lie in
 * the workspace at ps_grid_scatter_layout's offsets; the ranks exchange the streams of every table slice with the slice's owner
 * (presight_amd/dist.py) and the owner runs ps_grid_accumulate_runs over all ranks' runs: int64 fixed point with a scale from the
 * MAX-reduced per-level maximum -> exact, independent of the order of the runs.  out[0..8]: see csrc/encode.hip. */
int ps_grid_scatter_layout(int L, int F, int log2T, int64_t N, int K, int64_t* out /*host [9]*/);
int ps_grid_accumulate_runs(const uint32_t* run_starts, const uint32_t* run_counts, int n_runs, const uint32_t* rec_idx, const float* rec_val,
                            int64_t plane_stride, const uint32_t* gmax_bits, int L, int F, int log2T, int K, int64_t n_points_total,
                            float* dtable /*K = 1*/, float* const* dtables /*device [K] or NULL*/, float out_scale, int item_begin, int item_end,
                            void* stream);
/* Table backward + Adam in ONE pass, for training that exchanges no gradients (a single process, or one tile per GPU:
 * docs/building_priors.md:7-44).  The reference runs loss.backward() and then torch.optim.Adam over every table
 * (ns/engine/trainer.py:470-486, ns/engine/optimizers.py:133-140): the table gradient is written, read back by the optimizer and zeroed
 * at the next step.  A hash table receives exactly one gradient contribution per step, so here the accumulate pass applies the SAME
 * element update as ps_adam_step_ranges (one shared device function: equal bits) to every slice it finishes and the gradient never
 * reaches memory.  dtable / dtables[k] point INTO the flat gradient buffer grad_base and only locate the entries (element offset
 * dtable - grad_base into param_base / exp_avg_base / exp_avg_sq_base, all of one layout); they must hold zeros and stay zero.
 * Every slice of the items [item_begin, item_end) is updated, records or not (weight decay and moment decay reach every entry, as in
 * torch).  phase / items as in ps_grid_scatter_binned_part.  step >= 1: the tables' torch-style step count for this update.
 * Routed tile: group_of_field [K] (device int32, nullable) names the device-decided group of every sub-field (>= 0: group_flags /
 * group_steps as in ps_adam_step_ranges -- a sub-field whose flag is down is left untouched; its step count is advanced by the
 * ps_adam_step_ranges call that updates the group's other parameters, after this one in stream order); < 0: updated at `step`. */
int ps_grid_scatter_binned_adam(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t N,
                                int64_t plane_stride, float* dtable, const uint32_t* slice_counts, int absmax_ready, void* workspace,
                                int phase, int item_begin, int item_end, const float* grad_base, float* param_base,
                                float* exp_avg_base, float* exp_avg_sq_base, float lr, float beta1, float beta2, float eps,
                                float weight_decay, float grad_scale, int step, void* stream);
int ps_grid_scatter_binned_ms_adam(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t n_slots,
                                   int64_t plane_stride, float* const* dtables, int K, const int32_t* chunk_field,
                                   const uint32_t* slice_counts, int absmax_ready, void* workspace, int phase, int item_begin,
                                   int item_end, const float* grad_base, float* param_base, float* exp_avg_base,
                                   float* exp_avg_sq_base, float lr, float beta1, float beta2, float eps, float weight_decay,
                                   float grad_scale, int step, const int32_t* group_of_field, const int32_t* group_flags,
                                   const int32_t* group_steps, void* stream);
/* fused fields: packed = K packed parameter blocks back to back (ps_*_field_sizes packed_floats each); gpart receives
 * ps_*_field_parts_ms(n_slots, K) partial gradient blocks, reduced per sub-field by ps_mlp_unpack_table_ms */
int ps_prop_field_parts_ms(int64_t n_slots, int K);
int ps_main_field_parts_ms(int64_t n_slots, int K);
int ps_prop_field_fwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel, const float* packed,
                         int64_t n_slots, float* sigma, const int32_t* perm, const int32_t* field_start, int K, void* stream);
int ps_prop_field_bwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel, const float* packed,
                         const float* dsigma, int64_t n_slots, float* dfeat, float* gpart, uint32_t* level_absmax /* [K, L] */,
                         const int32_t* perm, const int32_t* field_start, int K, void* stream);
int ps_main_field_fwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color, const float* sel,
                         const float* dirs, const float* app, int S, int A, const float* packed, int64_t n_slots, float* sigma,
                         float* rgb, float* sem, float* acts, const int32_t* perm, const int32_t* field_start, int K, void* stream);
int ps_main_field_bwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color, const float* sel,
                         const float* dirs, const float* app, int S, int A, const float* packed, const float* dsigma,
                         const float* drgb, const float* dsem, const float* weights, int64_t n_slots, float* dfeat, float* dapp,
                         float* gpart, const float* acts, float* dzb_scratch /* as ps_main_field_bwd, [n_slots, 80] */,
                         float* dapp_points /* nullable, with dzb_scratch: d(appearance) per point [N, A] in the caller's order,
                                               WRITTEN; the caller sums it over the samples of a ray and dapp is left untouched
                                               (the sorted layout would otherwise cost 16 float atomics per point) */,
                         const int32_t* perm, const int32_t* field_start, int K, int stages /* as ps_main_field_bwd */, void* stream);
/* The MERGED network as the TRAINING path of routed tiles (the production shape: K = 8..16 sub-fields,
 * ns/configs/method_configs.py:87,141): per sub-field, base output rows 16..79 (the 64-d semantic embedding, no activation:
 * ns/fields/PreSight/ingp_field.py:130-151) are folded into the semantic head's first layer -- W' = W_sem0 W_base1[16:],
 * b' = W_sem0 b_base1[16:] + b_sem0 (ps_merge_linear_fwd_batch: K maps in one launch) -- so the base MLP ends in 16 outputs and the
 * three-layer head reads the base hidden layer: 4096 of the 27 264 forward MACs and 8192 of the backward MACs per sample less, same
 * function / parameters / gradients (fp32 sums re-associated).  Arguments as ps_main_field_fwd_ms / ps_main_field_bwd_ms; packed =
 * K blocks [base (L*F -> hidden -> 16) | semantic head (hidden -> 64 merged, 64 -> 64, 64 -> 64) | colour head] of
 * ps_main_field_m_sizes; acts [n_slots, act_width], dzb_scratch [n_slots, dzb_width]; three-kernel backward only (acts and
 * dzb_scratch required).  The gradient block's first semantic slot holds d(W'), d(b'): ps_merge_linear_bwd_batch adds the chain-rule
 * gradients to the four tensors of every sub-field.  ptrs: device table of K rows of addresses -- forward [W0, b0, We, be],
 * backward [W0, We, be, dW0, db0, dWe, dbe]; Wm / dWm [K, O, I], bm / dbm [K, O] contiguous. */
int ps_main_field_m_sizes(int LF, int hidden, int hidden_color, int64_t* packed_floats /*host*/, int64_t* grad_floats /*host*/,
                          int64_t* offsets /*host [6]*/, int* act_width /*host*/, int* dzb_width /*host*/);
int ps_main_field_m_fwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color, const float* sel,
                           const float* dirs, const float* app, int S, int A, const float* packed, int64_t n_slots, float* sigma,
                           float* rgb, float* sem, float* acts, const int32_t* perm, const int32_t* field_start, int K, void* stream);
int ps_main_field_m_bwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color, const float* sel,
                           const float* dirs, const float* app, int S, int A, const float* packed, const float* dsigma,
                           const float* drgb, const float* dsem, const float* weights, int64_t n_slots, float* dfeat, float* dapp,
                           float* gpart, const float* acts, float* dzb_scratch, float* dapp_points, const int32_t* perm,
                           const int32_t* field_start, int K, int stages, void* stream);
int ps_merge_linear_fwd_batch(const int64_t* ptrs /*device [K,4]*/, int n_fields, int O, int K, int I, float* Wm, float* bm, void* stream);
int ps_merge_linear_bwd_batch(const int64_t* ptrs /*device [K,7]*/, int n_fields, const float* dWm, const float* dbm, int O, int K, int I,
                              void* stream);
/* fused sky field (ns/fields/PreSight/sky_field.py:95-110): per ray SH4((dir+1)/2) -> semantic head (16 -> 32 -> 32 -> 64) and
 * [SH | appearance] -> colour head (16+A -> 32 -> 32 -> 3, sigmoid), one kernel per direction; packed = [colour | semantic]
 * packed stacks (K of them back to back for the routed sky model, perm / field_start from ps_ms_route on the ray ORIGINS,
 * sky_field_ms.py:97-114; NULL / NULL / 1 for a single field).  rgb / sem / drgb / dsem / dapp are in the caller's ray order;
 * sem / dsem may be NULL (no semantic head); dapp [R,A] is written (not accumulated). */
int ps_sky_field_supported(int A, int width, int num_layers, int semantic_dim);
int ps_sky_field_sizes(int A, int64_t N, int K, int ms, int64_t* packed_floats /*host*/, int64_t* grad_floats /*host*/,
                       int* n_parts /*host*/, int64_t* offsets /*host [4]: packed colour, semantic; gradient colour, semantic*/);
int ps_sky_field_fwd(const float* dirs, const float* app, int A, const float* packed, int64_t N, float* rgb, float* sem,
                     const int32_t* perm, const int32_t* field_start, int K, void* stream);
int ps_sky_field_bwd(const float* dirs, const float* app, int A, const float* packed, const float* drgb, const float* dsem,
                     int64_t N, float* dapp, float* gpart, const int32_t* perm, const int32_t* field_start, int K, void* stream);
/* MLP pack / gradient unpack driven by a descriptor TABLE in device memory (one 56-byte record per layer:
 * {W | partial block, b, colmap, dst0, dst1 : pointers; out_dim, in_dim, KS, NB : int32}, ps_mlp_layer_desc_bytes()).
 * unpack: record i belongs to sub-field i / layers_per_field; B = workgroups of the backward launch
 * (ps_*_field_parts_ms / parts_per_block), parts_per_block = 1 (main field) or 4 (proposal field). */
int ps_mlp_layer_desc_bytes(void);
int ps_mlp_pack_table(const void* table, int n_layers, int max_elems, void* stream);
int ps_mlp_unpack_table_ms(const void* table, int n_layers, int layers_per_field, const int32_t* field_start, int K, int B,
                           int parts_per_block, int64_t part_stride, int max_elems, void* stream);

/* ---- a18 prior extraction ---------------------------------------------------------------------------
 * voxel index of Open3D's voxel_down_sample_and_trace as called by ns/scripts/extract_priors.py:216-245:
 * idx = floor((p - (min_bound - voxel/2)) / voxel), fp64 arithmetic, int64 [n,3] (bit exact); min_bound is a HOST array */
int ps_voxel_index(const float* pts, int64_t n, double voxel, const double* min_bound /*host[3]*/, int64_t* idx,
                   void* stream);
/* cell centres of the dense res^3 query lattice over aabb (HOST array min xyz, max xyz), z fastest;
 * writes points [start, start+count) -> pts [count,3] */
int ps_lattice_points(const float* aabb /*host[6]*/, int res, int64_t start, int64_t count, float* pts, void* stream);
/* (a + b + c) / 3: mean of proposal-net and main-field densities, extract_priors.py:133-137 */
int ps_mean_density(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream);
/* The kept rows of a lattice chunk without a host round trip (round 6): what ns/scripts/extract_priors.py:133-152 does with
 * `mask = density > threshold` + masked selects, as three launches that APPEND to tile-wide arrays at a position kept in device memory
 * -- no nonzero(), no host synchronisation per chunk.  dens [n], pts [n,3], sem [n,64] fp32; cursor: int64[2] device memory, zeroed
 * before the first chunk of a tile ({rows so far, rows that did not fit `capacity`}); rows appear in ascending lattice order;
 * out_pts = pts / pose_scale (torch's scalar division: multiplication with the fp32 reciprocal), out_feat = clamp(sem, 0, 1) as fp16,
 * out_vox = ps_voxel_index(out_pts), out_row (nullable) = row0 + the point's index.  workspace: ps_emit_kept_workspace(n) bytes. */
int64_t ps_emit_kept_workspace(int64_t n);
int ps_emit_kept(const float* dens, int64_t n, float threshold, const float* pts, const float* sem, int C, float pose_scale, double voxel,
                 const double* min_bound /*host[3]*/, int64_t row0, int64_t* cursor, int64_t capacity, void* workspace, float* out_pts,
                 float* out_dens, void* out_feat_f16, int64_t* out_vox, int64_t* out_row /*nullable*/, void* stream);
/* out[i, :] = fp16(clamp(src[idx[i], :], lo, hi)): the features of the points above the density threshold, clipped to [0, 1] and
 * stored as fp16 (feat.clip(0, 1).astype(np.float16) of the selected rows, extract_priors.py:136-138); src [n,C] fp32, C <= 64,
 * idx [m] int64 row numbers, out [m,C] fp16 */
int ps_gather_clip_f16(const float* src, const int64_t* idx, int64_t m, int C, float lo, float hi, void* out_f16, void* stream);

/* voxel down-sampling of the extracted points (extract_priors.py:151-191, 216-245; Open3D voxel_down_sample_and_trace):
 * keys[i] = (ix * ny + iy) * nz + iz of the ps_voxel_index triple; after a stable sort of the keys (order [n] = point of
 * sorted position, starts / counts [V] = run of every voxel) ps_voxel_reduce writes per voxel the mean point (f32), the
 * mean colour (f32, colors may be NULL) and the mean feature (fp16 members summed in fp64 -> fp16, C <= 64 channels);
 * sums (nullable) [V, 6 + C] receives the raw fp64 sums {point, colour, feature} for merging partial results of ranks. */
int ps_voxel_keys(const float* pts, int64_t n, double voxel, const double* min_bound /*host[3]*/, int64_t ny, int64_t nz,
                  int64_t* keys, void* stream);
int ps_voxel_reduce(const int64_t* order, const int64_t* starts, const int64_t* counts, int64_t V, const float* pts,
                    const void* feats_f16, const float* colors, int C, float* o_pts, void* o_feat_f16, float* o_col,
                    double* sums, void* stream);

/* ---- f1 optimizer: torch.optim.Adam semantics (L2 weight decay added to the gradient, bias-corrected), in place.
 * Reference configuration: ns/configs/method_configs.py:158-168 (lr 1e-2, eps 1e-15, weight_decay 1e-5).
 * p, g, m, v: [n] fp32, 16-byte aligned; step counts from 1.  grad_scale multiplies g before the weight decay is added:
 * 1.0 = the reference's default (update_grad_scaler=False: optimizer.step() on the 2**10-scaled gradients, the weight decay is
 * added to THOSE -- ns/engine/trainer.py:481-486, ns/engine/optimizers.py:133-140); 1 / loss scale = its update_grad_scaler=True
 * branch (GradScaler.step unscales before the optimizer runs, ns/engine/optimizers.py:118-131). */
int ps_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, float grad_scale, void* stream);
/* The same update over n_ranges disjoint, non-empty ranges [start[i], start[i]+count[i]) (floats; start multiples of 4) of
 * flat p / g / m / v buffers, range i at its OWN step count (torch.optim.Adam advances state["step"] per parameter, only when
 * the parameter has a gradient: ns/engine/optimizers.py:133-140 after zero_grad(set_to_none=True)).  start / count / step /
 * group are HOST arrays; the range table travels as a kernel argument (one launch per 32 ranges, no host->device copy).
 * group (nullable) / group[i] >= 0: whether range i received a gradient this step is decided ON THE DEVICE --
 * group_flags[group[i]] != 0 (device int32 [n_groups], set by ps_ms_mark_groups for the routed sub-fields that received
 * samples: ingp_field_ms.py:97-126 never calls an empty sub-field, so its gradients stay None and Adam skips it); its step
 * count lives in group_steps[group[i]] (device int32) and is advanced after the update (the groups this call references: one
 * optimizer step may be issued as several calls over disjoint ranges, on different streams).  A range whose flag is 0 is left
 * untouched: parameters, both moments and the step count keep their bits.  n_groups <= 256. */
/* optimizer.zero_grad (ns/engine/trainer.py:470) for the ranges of the flat gradient buffer the previous step wrote: up to 16
 * (start, count) ranges of floats, HOST arrays, one launch */
int ps_zero_ranges(float* p, int n_ranges, const int64_t* start /*host*/, const int64_t* count /*host*/, void* stream);
int ps_adam_step_ranges(float* p, const float* g, float* m, float* v, int n_ranges, const int64_t* start /*host*/,
                        const int64_t* count /*host*/, const int* step /*host*/, const int* group /*host, nullable*/,
                        const int32_t* group_flags /*device, nullable*/, int32_t* group_steps /*device, nullable*/, int n_groups,
                        float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);
/* flags[group_of_field[k]] = 1 for every sub-field k of a routed layout (field_start [K+1] from ps_ms_route) that received
 * points; group_of_field [K] device int32, < 0 = no group.  Called by the routed backward nodes; no host synchronisation. */
int ps_ms_mark_groups(const int32_t* field_start, int K, const int32_t* group_of_field, int32_t* flags, void* stream);

/* ---- factored semantic path of the training render node (one sub-field).  The semantic head's input is a linear function of the
 * base MLP's hidden layer (no activation on the base output, ns/fields/PreSight/ingp_field.py:130-151) and its output is
 * composited linearly over the ray (ns/models/PreSight/nerfacto_nusc_ms.py:530), so (1) base layer 1 rows 16..79 and semantic
 * layer 0 are merged into one layer on the hidden activations, (2) the semantic output layer is applied once per ray to the
 * composited last hidden activations.  Same function / parameters / gradients as ps_main_field_fwd + ps_composite_fwd, 8192 of
 * the 26752 MACs per sample less in the forward and 16384 less in the backward (fp32 sums re-associated).
 * ps_merge_linear_fwd: Wm [O,I] = W0 [O,K] We [K,I], bm = W0 be + b0; _bwd adds the chain-rule gradients of the four tensors.
 * ps_ray_colour_fwd: the colour head's first layer split over its inputs [SH16(dir) | geo15 | appearance]
 *   (ns/fields/PreSight/ingp_field.py:239-262: directions and the camera's embedding are constant along a ray):
 *   ray_colour [R,HC] = W0[:, 0:16] SH16(dirs) + W0[:, 31:31+A] app, W0 [HC, 31+A] torch layout, HC = 32 | 64, A <= 16.
 * ps_ray_colour_bwd: dray_part [R*S/16, HC] (from ps_main_field_f_bwd) reduced per ray; dW0's SH / appearance columns += , dapp [R,A] written.
 * ps_main_field_f_*: as ps_main_field_fwd / _bwd with packed = [base (L*F -> hidden -> 16) | semantic (hidden -> 64 merged, 64 -> 64)
 *   | colour head whose first layer takes the 16 base outputs only (column map: geo15 -> torch columns 16..30)]; ray_colour is added
 *   to that layer's pre-activations; S % 32 == 0 and N % S == 0.  The forward renders the semantic branch itself: ebins [R,S+1] in,
 *   weights [N] (ps_weights_fwd's formula; ns/model_components/ray_samplers.py get_weights) and sem_hidden_ray [R,64] = sum_n w_n *
 *   (last hidden activations of the semantic head) out; sigma [N], rgb [N,3] per sample; acts [N, act_width], dzb_scratch
 *   [N, dzb_width] (ps_main_field_f_sizes).  Backward: dsem_hidden [R,64] = v of ps_sem_out_bwd (per ray), drgb [R,3] per ray,
 *   weights required; stage 1 writes dweights_sem [N] = <v_ray, last hidden activations> (add to d(weights) before
 *   ps_weights_bwd) and needs no dsigma; stages 2 | 4 need dsigma; dray_part [N/16, HC] = per-16-sample-block sums of d(ray_colour).
 * ps_sem_out_fwd: sem [R,64] = H W^T + b acc (H [R,64] = composited hidden activations, acc [R] = sum of the weights, unclamped).
 * ps_sem_out_bwd: v [R,64] = dsem W, cray [R] = <dsem, b> (joins d(acc)), dW [64,64] += dsem^T H, db [64] += dsem^T acc. */
int ps_merge_linear_fwd(const float* W0, const float* b0, const float* We, const float* be, int O, int K, int I, float* Wm, float* bm,
                        void* stream);
int ps_merge_linear_bwd(const float* dWm, const float* dbm, const float* W0, const float* We, const float* be, int O, int K, int I,
                        float* dW0, float* db0, float* dWe, float* dbe, void* stream);
int ps_main_field_f_sizes(int LF, int hidden, int hidden_color, int64_t N, int64_t* packed_floats /*host*/, int64_t* grad_floats /*host*/,
                          int* n_parts /*host*/, int64_t* offsets /*host [6]*/, int* act_width /*host*/, int* dzb_width /*host*/);
int ps_main_field_f_fwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color, const float* sel,
                        const float* ray_colour, const float* ebins, int S, const float* packed, int64_t N, float* sigma, float* rgb,
                        float* weights, float* sem_hidden_ray, float* acts, void* stream);
int ps_main_field_f_bwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color, const float* sel, int S,
                        const float* packed, const float* dsigma, const float* drgb, const float* dsem_hidden, const float* weights,
                        int64_t N, float* dfeat, float* dray_part, float* dweights_sem, float* gpart, const float* acts,
                        float* dzb_scratch, int stages, void* stream);
int ps_ray_colour_fwd(const float* dirs, const float* app, const float* W0, int64_t R, int A, int HC, float* ray_colour, void* stream);
int ps_ray_colour_bwd(const float* dray_part, const float* dirs, const float* app, const float* W0, int64_t R, int S, int A, int HC,
                      float* dW0, float* dapp, void* stream);
int ps_sem_out_fwd(const float* H, const float* acc, const float* W, const float* b, int64_t R, int C, float* sem, void* stream);
int ps_sem_out_bwd(const float* dsem, const float* H, const float* acc, const float* W, const float* b, int64_t R, int C, float* v,
                   float* cray, float* dW, float* db, void* stream);

/* ---- BASELINE cfg 4: dynamic branch of the dual (static + dynamic) field -- csrc/dynamic.hip.  The reference has no dynamic /
 * flow field ("no counterpart"); these entry points implement the model DEFINED by
 * oracle/dual_oracle.py, whose conventions are those of the static stack (ns/fields/PreSight/ingp_field.py:168-267 for the MLP
 * stack and trunc_exp * selector, ns/field_components/encodings.py:324-384 for the hash grid, extended by a time axis with
 * Instant-NGP's fourth prime 3674653429).  A maintainer would bind them from a new field class next to iNGPField. */
/* x4[n] = (u[n], times[n / S]): the static field's normalised positions u [N,3] + the rays' normalised timestamps [N / S] */
int ps_dyn_points(const float* u, const float* times, int S, int64_t N, float* x4 /*[N,4]*/, void* stream);
/* 4-D multiresolution hash grid on level planes feat[l][n][f].  e0 == NULL: feat = H4(x), x [N,4].
 * e0 != NULL (temporal aggregation): x [2N,4] = forward-warped then backward-warped positions, feat = (e0 + H4(x[n]) + H4(x[N+n])) / 3.
 * slice_counts (nullable, [L * ps_grid_scatter_slices(F, log2T)] uint32, NOT cleared here): += the records the binned table
 * backward will emit for the encoded positions, so that ps_grid4_scatter_binned needs no counting pass. */
int ps_grid4_encode(const float* x, const float* table, const float* scalings, int L, int F, int log2T, int64_t N,
                    int64_t plane_stride, const float* e0 /*nullable*/, float* feat, uint32_t* slice_counts /*nullable*/, void* stream);
/* dx[m][0..3) = g_scale * sum_l scalings[l] * <dfeat[l][m mod period], d H4[l] / d x_a> for the spatial axes (the hash grid is
 * multilinear inside a cell; exactly integer coordinates have zero derivative, like autograd through ceil / floor).
 * x [M,4], M <= 2 * period when period > 0 (two position sets share one gradient plane), dx [M,3]. */
int ps_grid4_input_grad(const float* x, const float* dfeat, const float* table, const float* scalings, int L, int F, int log2T,
                        int64_t M, int64_t period, int64_t plane_stride, float g_scale, float* dx, void* stream);
/* the same gradient, level-parallel (a workgroup = 2048 positions of ONE level, dealt to the XCDs like the encode, + an ordered sum
 * over the levels): bit-identical result, workspace = ps_grid4_input_grad_workspace(L, M) bytes */
int64_t ps_grid4_input_grad_workspace(int L, int64_t M);
int ps_grid4_input_grad_levels(const float* x, const float* dfeat, const float* table, const float* scalings, int L, int F, int log2T,
                               int64_t M, int64_t period, int64_t plane_stride, float g_scale, float* dx, float* workspace, void* stream);
/* table gradient of the 4-D grid: the binned fixed-point scatter of ps_grid_scatter_binned with 8 x-pair records per (point,
 * level); dtable (+)= out_scale * scatter(d(features)); accumulate as ps_grid_scatter_binned; workspace of
 * ps_grid4_scatter_workspace.  x [M,4].  period > 0: up to three position sets of `period` points in ONE launch -- point m < period
 * takes row m of dfeat, m >= period row (m - period) mod period of dfeat_b (NULL: of dfeat): the unwarped set with d(e0) and both
 * warped sets with d(aggregated features).  slice_counts (nullable): record counts of ps_grid4_encode for exactly these M points. */
int64_t ps_grid4_scatter_workspace(int L, int F, int log2T, int64_t M);
int ps_grid4_scatter_binned(const float* x, const float* dfeat, const float* dfeat_b /*nullable*/, const float* scalings, int L, int F,
                            int log2T, int64_t M, int64_t period, int64_t plane_stride, float out_scale, float* dtable, int accumulate,
                            const uint32_t* slice_counts /*nullable*/, void* workspace, void* stream);
/* flow MLP Linear(L*F, H) ReLU Linear(H, H) ReLU Linear(H, 6) on the fp32 matrix cores; packed as ps_mlp_pack_layers packs a
 * 3-layer stack (ps_flow_sizes: packed / gradient block sizes, partial blocks ps_flow_bwd writes).
 * fwd: xw [2N,4] <- (u + s*flow[0:3], t + dt) for n < N, (u + s*flow[3:6], t - dt) for N + n (x4 [N,4] = (u, t)).
 * bwd: dxw [2N,3] = gradient w.r.t. the warped positions (ps_grid4_input_grad); de0 = dagg + 3 * d(e0 via the flow MLP), i.e.
 * THREE TIMES d(e0): the factor 1/3 of the aggregation is applied once, for all three position sets, by the table scatter. */
int ps_flow_sizes(int LF, int hidden, int64_t N, int64_t* packed_floats /*host*/, int64_t* grad_floats /*host*/, int* n_parts /*host*/);
int ps_flow_fwd(const float* e0, int64_t plane_stride, int LF, int F, int hidden, const float* packed, const float* x4, int64_t N,
                float flow_scale, float dt, float* xw, void* stream);
int ps_flow_bwd(const float* e0, int64_t plane_stride, int LF, int F, int hidden, const float* packed, const float* dxw,
                const float* dagg, int64_t N, float flow_scale, float* de0, float* gpart, void* stream);
/* density-weighted blend of the two branches per sample: sigma = ss + sd, wd = sd / max(sigma, 1e-6), c = cs + wd (cd - cs) for
 * the colour [N,3] and the semantics [N,C] (C a multiple of 4).  bwd: any of dsigma / drgb / dsem may be NULL (no gradient);
 * extra_dsigma_d (nullable) is added to d(sigma_d): gradients that reach the dynamic density directly (its regulariser). */
int ps_blend_fwd(const float* sigma_s, const float* rgb_s, const float* sem_s, const float* sigma_d, const float* rgb_d,
                 const float* sem_d, int64_t N, int C, float* sigma, float* rgb, float* sem, void* stream);
int ps_blend_bwd(const float* sigma_s, const float* rgb_s, const float* sem_s, const float* sigma_d, const float* rgb_d,
                 const float* sem_d, const float* dsigma, const float* drgb, const float* dsem, int64_t N, int C,
                 const float* extra_dsigma_d, float* dsigma_s, float* drgb_s, float* dsem_s, float* dsigma_d, float* drgb_d,
                 float* dsem_d, void* stream);

/* ---- round 6: fused per-ray tail of the training step (csrc/raytail.hip, csrc/losses.hip) --------------------------
 * The same operators as above, fewer launches.  Replaces, for the factored training render node (one sub-field, S <= 64):
 *   ps_ray_out_fwd        ps_composite_fwd's rgb / accumulation (ns/model_components/renderers.py:70-117,286-314) + ps_sem_out_fwd (the
 *                         semantic head's output layer per ray, ns/fields/PreSight/ingp_field.py:143-151), same arithmetic and order;
 *                         weights [R,S], rgb_s [R,S,3], sem_hidden_ray [R,64], W [64,64], b [64] -> rgb [R,3], acc [R] (unclamped), sem [R,64]
 *   ps_ray_dsigma_bwd     d(weights) = <rgb_s, d_rgb> + d_acc + d_acc2 + add0 + add1 (ps_composite_bwd without the depth terms) handed
 *                         straight to RaySamples.get_weights' backward (ns/cameras/rays.py:128-150 = ps_weights_bwd): dsigma [R,S]
 *   ps_blend_losses       rgb += (1 - clamp(acc)) sky_rgb etc. (ns/models/PreSight/nerfacto_nusc_ms.py:512-533) + MSELoss(rgb)
 *                         (nerfacto_nusc_ms.py:568) + sky_loss + semantic_loss (ns/model_components/PreSight/losses.py:106-125) in one
 *                         launch: blended outputs, per-workgroup partial sums of the three terms (partial [3][ps_blend_losses_partials(R)],
 *                         finished by ps_finish_losses), AND the gradients of the k-weighted sum of the terms w.r.t. every input of
 *                         the blend (k_t = loss_mult_t * seed of the backward pass * d(mean)/d(sum)); targets nullable (term absent)
 *   ps_*_loss_scaled      ps_distortion_loss / ps_interlevel_loss with the gradient scaling of ps_scale_grad folded in:
 *                         dw = grad_scale * d(per_ray) (grad_scale = loss_mult / count * seed; the fp32 product ps_scale_grad formed)
 *   ps_finish_losses      ALL scalar loss values of a step and their sum in one launch: descriptor d = (terms[d] device pointer, n[d],
 *                         denom[d], scale[d], out_of[d]) -- HOST arrays of n_desc <= 16 entries, out_of non-decreasing;
 *                         *out[o] = sum over the descriptors of output o, in order, of scale * (sum(terms) / denom) -- each term with
 *                         the bits of ps_loss_finish, several descriptors per output = term_0 + term_1 as the reference's python sum
 *                         of the interlevel levels forms it; *total (nullable) = ((*out[0] + *out[1]) + *out[2]) + ... =
 *                         functools.reduce(torch.add, loss_dict.values()) of ns/engine/trainer.py:478.  out: HOST array of n_out device
 *                         pointers; ticket: one zeroed uint32 in device memory that the launch leaves zeroed */
/*   ps_spaced_bins_points ps_spaced_bins + ps_field_points (the first proposal field's points from these edges), one launch
 *   ps_weights_resample   RaySamples.get_weights (ps_weights_fwd) + PDFSampler (ps_pdf_resample) of the NEXT level's edges + (u != NULL)
 *                         ps_field_points on those edges for the next field: one launch per proposal level instead of three, same
 *                         arithmetic in the same order (ns/cameras/rays.py:128-150, ns/model_components/ray_samplers.py:305-372,
 *                         ns/cameras/rays.py:49-58 + ns/fields/PreSight/ingp_field.py:169-177); weights [R,S] is written */
int ps_spaced_bins_points(const float* jitter, int64_t R, int S, float near, float far, float thr, float* sbins, float* ebins,
                          const float* origins, const float* dirs, const float* aabb, int contract, float* u, float* sel, void* stream);
int ps_weights_resample(const float* ebins, const float* sigma, const float* sbins, const float* jitter /*nullable*/, int64_t R, int S,
                        int n_new, float anneal, float pad, float eps, float near, float far, float thr, float* weights,
                        float* new_sbins, float* new_ebins, const float* origins /*nullable*/, const float* dirs /*nullable*/,
                        const float* aabb /*nullable*/, int contract, float* u /*nullable*/, float* sel /*nullable*/, void* stream);
int ps_ray_out_fwd(const float* weights, const float* rgb_s, const float* sem_hidden_ray, const float* W, const float* b, int64_t R,
                   int S, int C, float* rgb, float* acc, float* sem, void* stream);
int ps_ray_dsigma_bwd(const float* ebins, const float* sigma, const float* rgb_s, const float* d_rgb, const float* d_acc /*nullable*/,
                      const float* d_acc2 /*nullable*/, const float* d_weights_add0 /*nullable*/, const float* d_weights_add1 /*nullable*/,
                      int64_t R, int S, float* dsigma, void* stream);
int ps_blend_losses_partials(int64_t R);
int ps_blend_losses(const float* rgb_f, const float* acc_raw, const float* sem_f, const float* sky_rgb, const float* sky_sem,
                    const float* rgb_t, const float* sky_t, const float* sem_t, int64_t R, int C, int clip_sem_target,
                    float bce_eps, float k_rgb, float k_sky, float k_sem, float* rgb, float* acc, float* sem, float* d_rgb,
                    float* d_sem, float* d_acc_raw, float* d_sky_rgb, float* d_sky_sem, float* partial, void* stream);
int ps_distortion_loss_scaled(const float* sbins, const float* w, int64_t R, int S, float* per_ray, float* dw, float grad_scale,
                              void* stream);
int ps_interlevel_loss_scaled(const float* c, const float* w, const float* cp, const float* wp, int64_t R, int S, int Sp,
                              float pulse_width, float* per_ray, float* dwp, float grad_scale, void* stream);
int ps_finish_losses(const float* const* terms, const int64_t* n, const float* denom, const float* scale, const int* out_of,
                     int n_desc, float* const* out, int n_out, float* total, uint32_t* ticket, void* stream);

#ifdef __cplusplus
}
#endif
#endif
