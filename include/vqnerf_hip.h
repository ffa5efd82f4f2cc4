/* vqnerf_hip.h -- C ABI of libvqnerf_hip.so (gfx950 / MI355X kernels for the VQ-NeRF hot path).
 *
 * The reference (JiuTongBro/vqnerf_release) has no FFI layer: its hot path is sequences of
 * framework ops behind two Python classes.  Each entry point below replaces one such sequence
 * (cited as reference file:line) and is what a reference-side ctypes binding would call
 * (INTEGRATION.md shows that stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr());
 *     the library allocates nothing persistent except the weight-pack handles of vqn_{neus,chain}_pack_create / _destroy
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued asynchronously
 *   - return 0 on success; -1 bad argument, -2 unsupported shape, -3 HIP runtime error;
 *     vqn_last_error() returns the thread-local message of the last failure
 *   - re-entrant; no global mutable state besides that thread-local string
 *   - all floating point is fp32, indices are int64, tensors are dense row-major
 */
#ifndef VQNERF_HIP_H_
#define VQNERF_HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int vqn_version(void);
const char* vqn_last_error(void);

/* ---- VQ codebook (decomp/nerfvq_nfr3/nerfactor/networks/vq_layers.py) --------------------- */

/* Replaces vq_layers.py:277-301,346-349: distances = |x|^2 - 2 x C + |C|^2, optional code-dropout
 * mask (dropped codes get max(distances)), idx = argmax(-distances) (lowest index on ties),
 * quant = C^T[idx].
 *   x [N,D], codebook [D,K]; sel_mask [K] (1 keep / 0 drop) or NULL; ws: >= 4 bytes of device
 *   scratch, required iff sel_mask != NULL; idx [N] int64; quant [N,D] or NULL; dist [N,K] or NULL.
 * Summation order is fixed (see oracle/vq_strict.c); D % 4 == 0, K <= 128. */
int vqn_vq_assign(const float* x, int64_t N, int D, const float* codebook, int K, const float* sel_mask,
                  float* ws, int64_t* idx, float* quant, float* dist, void* stream);

/* Which kernel a vqn_vq_assign / vqn_vq_quantize_rows call of this shape launches: 0 = the f32 kernel (all distances on the f32
 * matrix pipe), 1 = the prefiltered kernel (K in 17..64, D <= 256, no mask, no distance output: distances first from f16 pairs, the
 * codes within the error margin of the minimum then evaluated in the defined f32 order -- the same indices bit for bit, see vq.hip).
 * The environment variable VQN_VQ_SPLIT=0 forces 0. */
int vqn_vq_assign_variant(int D, int K, int has_sel_mask, int has_dist);

/* Replaces vq_layers.py:304-309 (the two reductions that feed the EMAs):
 *   counts[k] = #{n : idx[n] == k}   (as float),   dw[d,k] = sum_n x[n,d] [idx[n] == k].
 * counts [K], dw [D,K] are overwritten.  dw == NULL: counts only (x is not read). */
int vqn_vq_ema_stats(const float* x, const int64_t* idx, int64_t N, int D, int K, float* counts, float* dw,
                     float* ws, int64_t ws_bytes, void* stream);
/* Bytes of device scratch `ws` that make vqn_vq_ema_stats use a deterministic form with fixed-order reductions: the
 * matrix-pipe form x^T . onehot(idx) for D % 64 == 0, D <= 256, K <= 64, else per-wave LDS images for K*D <= 4096;
 * 0 = not applicable.  With ws == NULL (or too small) the single-pass LDS-atomic form runs.  Rows whose idx is outside
 * [0, K) contribute nothing in every form. */
int64_t vqn_vq_ema_stats_ws_bytes(int64_t N, int D, int K);

/* Replaces vq_layers.py:302 + :327 (and their autograd-free arithmetic) in one pass over the rows:
 *   ste[i]  = x[i] + (quant[i] - x[i])          (the straight-through output as the reference writes it; NULL = skip)
 *   *loss   = scale * sum_i (quant[i] - x[i])^2 (scale = 1 / numel gives mean((sg(q) - x)^2), the e_latent term)
 * summed in a fixed order (bit-reproducible for a given numel).  numel = N * D, a multiple of 4; ws: VQN_STE_WS_FLOATS
 * floats of device scratch.  numel == 0 writes NaN (the reference's mean over nothing). */
#define VQN_STE_WS_FLOATS 1024
int vqn_vq_ste_loss(const float* x, const float* quant, int64_t numel, float scale, float* ste, float* loss, float* ws,
                    void* stream);

/* The EMA codebook move of VectorQuantizerEMA in training (vq_layers.py:304-325 with Sonnet's ExponentialMovingAverage for the
 * cluster sizes and for dw, third party: hidden -= (hidden - v)(1 - decay); counter += 1; average = hidden / (1 - decay^counter)) as ONE
 * launch (round 3): updates both averages' state in place (hidden / average / int64 counter each), then
 *   n = sum_k cs_k;  cs'_k = (cs_k + eps) / (n + K eps) n;  update[d,k] = counts[k] > 0 ? average_dw[d,k] / cs'_k : codebook[d,k].
 * counts [K] and dw [D,K] are the (already all-reduced) statistics of vqn_vq_ema_stats; codebook [D,K] the clipped, normalised one the
 * assignment used.  The debias factor is taken in float64.  K <= 1024. */
int vqn_vq_ema_update(const float* counts, const float* dw, const float* codebook, int D, int K, double decay, float eps,
                      float* hidden_cs, float* average_cs, int64_t* counter_cs, float* hidden_dw, float* average_dw,
                      int64_t* counter_dw, float* update, void* stream);

/* Per-example loss terms of the VQ reflectance stage in TRAIN mode and their gradients, one launch each (round 3).  Replaces the
 * framework-op chains of vq_nfr.Model.compute_loss (decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:876-986, train branch) and their
 * autograd for the terms that are per surface point; N points, rows 2j / 2j+1 are a pixel pair (train_nfr.py:447-448):
 *   terms[i] = { w_rgb mean_c (lin_gt - rgb_pred)^2,  mean_c (lin_gt - vq_rgb)^2,  w_chr mean_c (chr(lin_gt) - chr(vq_rgb))^2,
 *                w_smooth exp(-chr_alpha e_j) (1 - <z_2j, z_2j+1>),  w_lambert max_c spec r' }
 * with lin_gt = srgb2linear(rgb_gt) if nerf != 0 else rgb_gt (util/img.py:166-186), chr(v) = v / |v| (0 at |v| = 0, :869-874),
 * e_j = |chr(rgb_gt_2j) - chr(rgb_gt_2j+1)| where that exceeds chr_thres else 0 (:927-953), r' = 0 for rough < 0.5 else 2 rough - 1
 * (:970-981).  z [N,D] / spec [N,3] / rough [N,1] may be NULL (their terms are 0).  The scalar terms of the loss (commitment loss,
 * code-separation term) stay with the caller.  _bwd: g_terms [N,5] = upstream gradient of every term -> d/d rgb_pred, d/d vq_rgb
 * [N,3], d/d z [N,D] (g_z may be NULL), d/d spec [N,3] (may be NULL); gradients are 0 (as TensorFlow's divide_no_nan / SqrtGrad
 * give) where |vq_rgb| = 0. */
int vqn_decomp_loss_fwd(const float* rgb_pred, const float* vq_rgb, const float* rgb_gt, const float* z, const float* spec,
                        const float* rough, int64_t N, int D, int nerf, float w_rgb, float w_chr, float w_smooth, float chr_alpha,
                        float chr_thres, float w_lambert, float* terms, void* stream);
int vqn_decomp_loss_bwd(const float* rgb_pred, const float* vq_rgb, const float* rgb_gt, const float* z, const float* spec,
                        const float* rough, int64_t N, int D, int nerf, float w_rgb, float w_chr, float w_smooth, float chr_alpha,
                        float chr_thres, float w_lambert, const float* g_terms, float* g_rgb_pred, float* g_vq_rgb, float* g_z,
                        float* g_spec, void* stream);

/* Replaces `mathutil.safe_l2_normalize(z, axis=1)` (util/math.py:63-64 = tf.linalg.l2_normalize: x * rsqrt(max(sum x^2, eps))) as
 * called at vq_nfr.py:575: y [N,D] = x / sqrt(max(sum_d x^2, eps)) row by row, the sum in the defined order of vqn_vq_assign's
 * |x|^2 and a correctly rounded sqrt / division (oracle/vq_strict.c states it in C).  D % 4 == 0. */
int vqn_l2_normalize_rows(const float* x, int64_t N, int D, float eps, float* y, void* stream);

/* Backward of vqn_l2_normalize_rows: gx = g s - [sum x^2 > eps] x s^3 (x . g), s = max(sum x^2, eps)^(-1/2) (the gradient of
 * tf.linalg.l2_normalize, util/math.py:63-64, under the reference's autograd), one pass over [N, D].  D % 4 == 0, D <= 1024. */
int vqn_l2_normalize_rows_bwd(const float* x, const float* g, int64_t N, int D, float eps, float* gx, void* stream);

/* Backward of the straight-through estimator + commitment loss (networks/vq_layers.py:302, :327): gx = g_ste + (x - quant) *
 * (g_loss[0] * 2 / numel); g_ste may be NULL, g_loss is a device scalar; each operation separately rounded. */
int vqn_vq_ste_loss_bwd(const float* x, const float* quant, const float* g_ste, const float* g_loss, int64_t numel, float* gx, void* stream);

/* The codebook as the model uses it (vq_nfr.py: tfp clip_by_value_preserve_gradient to [0, 1], then l2-normalise every code = a
 * column of the [D, K] variable): out = c s, c = x + (clip(x) - x), s_k = max(sum_d c^2, eps)^(-1/2).  g == NULL: forward;
 * g != NULL: out = the gradient wrt the raw variable for the incoming g [D, K] (identity through the clip). */
int vqn_codebook_prep(const float* raw, const float* g, int D, int K, float eps, float* out, void* stream);

/* The code-separation term (vq_nfr.py:955-968): out4[0] = -weight log(min_{i != j} |c_i - c_j|) over the columns of codebook [D, K]
 * (out4[1..3]: the minimum and its pair, for the backward); _bwd: g_codebook = g_loss[0] times its gradient (two non-zero columns). */
int vqn_sim_smooth_fwd(const float* codebook, int D, int K, float weight, float* out4, void* stream);
int vqn_sim_smooth_bwd(const float* codebook, const float* fwd4, const float* g_loss, int D, int K, float weight, float* g_codebook,
                       void* stream);

/* The inference path of vq_nfr.Model.call / fast_embed / vq_test (vq_nfr.py:575-578 -> vq_layers.py:277-302, :327-330) in ONE pass
 * over the rows: z [N,D] un-normalised encoder output -> l2-normalise (as vqn_l2_normalize_rows) -> nearest code (as vqn_vq_assign,
 * incl. the code-dropout mask) -> idx [N], ste [N,D] = z^ + (q - z^) (or NULL), *loss = loss_scale * sum (q - z^)^2 (fixed order
 * for a given N), counts [K] = code usage.  Bit-identical idx and ste to the sequence vqn_l2_normalize_rows -> vqn_vq_assign ->
 * vqn_vq_ste_loss; reads N D floats, writes N D floats + N indices (the sequence moves 7 N D).  D <= 256; ws: VQN_QUANT_WS_FLOATS
 * floats of device scratch. */
#define VQN_QUANT_WS_FLOATS 4096
int vqn_vq_quantize_rows(const float* z, int64_t N, int D, const float* codebook, int K, const float* sel_mask, float eps,
                         float loss_scale, float* ws, int64_t* idx, float* ste, float* loss, float* counts, void* stream);

/* Backward of that chain in one pass per row (round 4): g_xn = g_ste + (xnorm - quant) (g_loss 2 / (N D)) -- the straight-through
 * identity plus the commitment term's gradient (vq_layers.py:302, :327) -- then the l2-normalise backward of z at g_xn
 * (util/math.py:63-64).  g_ste may be NULL (zero), g_loss a device scalar.  = vqn_vq_ste_loss_bwd followed by
 * vqn_l2_normalize_rows_bwd, same roundings.  loss_post: the forward's second loss factor (below), applied to g_loss first. */
int vqn_vq_train_bwd(const float* z, const float* xnorm, const float* quant, const float* g_ste, const float* g_loss, float loss_post,
                     int64_t N, int D, float eps, float* g_z, void* stream);

/* vqn_vq_quantize_rows for the TRAINING path (round 4; vq_nfr.py:575-578 with is_training = True): the same single pass, and
 * xnorm [N, D] additionally receives the l2-normalised rows -- the x of the EMA statistics (vq_layers.py:304-309, vqn_vq_ema_stats)
 * and of the backward (vqn_vq_ste_loss_bwd, vqn_l2_normalize_rows_bwd).  Same indices / straight-through rows / loss / counts.
 * loss = (sum * loss_scale) * loss_post: the mean, then the commitment cost as its own rounded multiplication (vq_layers.py:327-330:
 * `commitment_cost * mean(...)`); 1.0f leaves the mean. */
int vqn_vq_quantize_rows_train(const float* z, int64_t N, int D, const float* codebook, int K, const float* sel_mask, float eps,
                               float loss_scale, float loss_post, float* ws, int64_t* idx, float* ste, float* loss, float* counts,
                               float* xnorm, void* stream);

/* ---- reflectance MLP stacks + shading (decomp/nerfvq_nfr3/nerfactor) ------------------------- */

/* Generic fused Dense-stack evaluator: [posenc ->] Dense -> Dense ... with skip-concats and several
 * heads per launch, driven by a layer program (flat int32 array in HOST memory, layout
 * include/vqn_chain_desc.h, built by vqnerf_release_amd/decomp/packing.py together with `wbuf`, device).
 * Replaces networks/embedder.py:23-47 + networks/mlp.py:24-50 + networks/seq.py:24-38 as evaluated by
 * models/vq_nfr.py:771-784 (_pred_enc_at: xyz [N,3] -> z [N,z_dim]) and :786-828 (_pred_diff_at,
 * _pred_spec_at, _pred_rough_at: z -> [N,3], [N,1|3], [N,1]); models/shape.py:169-179 (chunk_apply)
 * disappears.  `in` is [N, in_stride] floats; out_i is [N, ld_i] (NULL for unused slots). */
int vqn_mlp_chain_fwd(const int32_t* desc, const float* wbuf, const float* in, int64_t N, float* out0, int ld0,
                      float* out1, int ld1, float* out2, int ld2, float* out3, int ld3, void* stream);

/* The same evaluator on the split-precision engine ("fp16 MFMA path"): every f32 value is carried as an f16 hi/lo
 * pair and every product as hi*hi + 2^-11 (hi*lo + lo*hi) on v_mfma_f32_32x32x16_f16 with f32 accumulation.  Same
 * arguments; `desc` / `wbuf` must be built for it (ChainBuilder(mode='f16s'): even row counts, f16 hi/lo weight
 * fragments).  Opt-in: agrees with vqn_mlp_chain_fwd to ~1e-6 relative, not bitwise; |weights| < 6e4. */
int vqn_mlp_chain_fwd_f16s(const int32_t* desc, const float* wbuf, const float* in, int64_t N, float* out0, int ld0,
                           float* out1, int ld1, float* out2, int ld2, float* out3, int ld3, void* stream);

/* Fused shading: light / view directions (models/shape.py:103-119), camera-facing normal
 * (models/vq_nfr.py:830-833), GGX microfacet BRDF (util/microfacet.py:9-89) and the rendering-equation
 * sum over L lights with front-lit test, visibility, optional gamma and the [0,1] clip
 * (models/vq_nfr.py:694-723), for one or two material sets (the continuous and the VQ branch of
 * vq_nfr.Model.call) in one pass.
 *   xyz, normal, rayo [N,3]; lvis [N,L] or NULL; lxyz [L,3], lareas [L], light [L,3] (already >= 0);
 *   albedo_s, spec_s [N,3], rough_s [N]; gamma: device [2] = (bias, index) or NULL (data_type nerf);
 *   normal_out [N,3] or NULL; rgb_s [N,3]; rgb0_diff / rgb0_spec [N,3] or both NULL (vali mode,
 *   vq_nfr.py:605-610).  L in {256, 512, 1024}.  raw != 0: rgb_s receive the plain sums over lights
 *   (no gamma, no clip): the training path applies those in the host framework so that autograd sees them; raw == 2: the plain sums
 *   through x + (clip(x, 0, 1) - x) -- tfp's clip_by_value_preserve_gradient, whose gradient is the identity (so vqn_brdf_shade_bwd
 *   is the reverse of raw = 1 and raw = 2 alike), as the training path of data_type 'nerf' applies it (no gamma curve there).
 *   raw must be 0, 1 or 2, and gamma must be NULL unless raw == 0 (-1 otherwise: a gamma that would be silently ignored).
 *   probes [P,L,3] (or NULL): material set 0 is additionally re-lit by every probe in the same pass
 *   (vq_nfr.py:724-733, the per-probe Python loop of the reference) -> rgb0_probes [N,P,3]. */
int vqn_brdf_shade_fwd(const float* xyz, const float* normal, const float* rayo, const float* lvis,
                       const float* lxyz, const float* lareas, const float* light, int64_t N, int L, int n_sets,
                       const float* albedo0, const float* spec0, const float* rough0, const float* albedo1,
                       const float* spec1, const float* rough1, const float* gamma, float* normal_out, float* rgb0,
                       float* rgb1, float* rgb0_diff, float* rgb0_spec, int raw, const float* probes, int n_probes,
                       float* rgb0_probes, void* stream);

/* The same with the foreground gather of the visibility buffer (vq_nfr.py:558-559, tf.boolean_mask(lvis, alpha > 0)) folded in:
 * lvis_rows [N] int64 (or NULL = identity) names, for each of the N shaded points, its row of the FULL-view lvis [n_view, L];
 * everything else stays compact [N, .].  Saves the copy the gather makes: 2 KB per foreground point read and written once more. */
int vqn_brdf_shade_fwd_rows(const int64_t* lvis_rows, const float* xyz, const float* normal, const float* rayo, const float* lvis,
                            const float* lxyz, const float* lareas, const float* light, int64_t N, int L, int n_sets,
                            const float* albedo0, const float* spec0, const float* rough0, const float* albedo1,
                            const float* spec1, const float* rough1, const float* gamma, float* normal_out, float* rgb0,
                            float* rgb1, float* rgb0_diff, float* rgb0_spec, int raw, const float* probes, int n_probes,
                            float* rgb0_probes, void* stream);

/* Reverse of vqn_brdf_shade_fwd(raw = 1) (autograd in the reference: tape.gradient through microfacet.py:9-89 and
 * vq_nfr.py:694-723): given g_sum_s = d loss / d (plain sum over lights) [N,3] per material set, returns
 * d loss / d albedo_s, spec_s [N,3], rough_s [N], and per-wave partials of d loss / d light
 * [vqn_brdf_shade_bwd_partials(N)][L][3] which the caller sums in order (deterministic, no float atomics).
 * Geometry (xyz, normal, rayo, lvis) is data and gets no gradient. */
int64_t vqn_brdf_shade_bwd_partials(int64_t N);
int vqn_brdf_shade_bwd(const float* xyz, const float* normal, const float* rayo, const float* lvis, const float* lxyz,
                       const float* lareas, const float* light, int64_t N, int L, int n_sets, const float* albedo0,
                       const float* spec0, const float* rough0, const float* g_sum0, const float* albedo1,
                       const float* spec1, const float* rough1, const float* g_sum1, float* g_albedo0, float* g_spec0,
                       float* g_rough0, float* g_albedo1, float* g_spec1, float* g_rough1, float* g_light_partials,
                       void* stream);

/* ---- fused NeuS networks (geo/NeuS-ours2/models/{fields,renderer}.py) ---------------------- */

/* Network descriptors are flat int32 arrays in HOST memory (layout: include/vqn_neus_desc.h, built by
 * vqnerf_release_amd/geo/packing.py together with the packed weight buffers `wbuf_*`, which are
 * device memory).  Points are either explicit (`pts` [P,3], `dirs` [P,3]) or ray samples
 * (rays_o, rays_d [B,3], z [B,S], P = B*S, point = o + d*z) -- pass NULL for the unused form. */

/* Replaces SDFNetwork.sdf(pts) as called at renderer.py:337-338 (coarse samples) and :180-185
 * (new samples of each up-sampling step): posenc (embedder.py:16-34) + the whole weight-normed
 * softplus(beta=100) MLP with its skip connection (fields.py:72-91), sdf only.  out_sdf [P]. */
int vqn_neus_sdf_points(const int32_t* sdf_desc, const float* wbuf_sdf, const float* rays_o, const float* rays_d,
                        const float* z, const float* pts, int64_t P, int S, float* out_sdf, void* stream);

/* Bytes of device scratch vqn_neus_fine_points wants for full occupancy (it runs, slower, with less). */
int64_t vqn_neus_fine_scratch_bytes(const int32_t* sdf_desc);

/* Replaces renderer.py:216-227: sdf_network(pts) (fields.py:72-91), sdf_network.gradient(pts)
 * (fields.py:96-107, autograd wrt the input -> here an explicit reverse sweep) and
 * color_network(pts, gradients, dirs, feature) (fields.py:147-172).
 * out_sdf [P], out_grad [P,3], out_rgb [P,3].  col_desc with n_lin == 0 skips the colour net
 * (out_rgb may then be NULL): that is SDFNetwork.gradient(). */
int vqn_neus_fine_points(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc,
                         const float* wbuf_col, const float* rays_o, const float* rays_d, const float* z,
                         const float* pts, const float* dirs, int64_t P, int S, void* scratch,
                         int64_t scratch_bytes, float* out_sdf, float* out_grad, float* out_rgb, void* stream);

/* Training forward of the same path: vqn_neus_fine_points at explicit (pts, dirs) that also leaves, in the tile format of
 * vqn_tile_program ([point tile][feature tile][32 features][32 points] f32), what the backward of
 * geo/NeuS-ours2/models/renderer.py:216-227 under exp_runner.py:153-168 (loss.backward()) needs saved: tensors[] =
 * [E, OUTF, EXTR, U_1..U_nL, GH_0..GH_{nL-1}, C_1..C_nC] (embedding; [sdf ; features]; colour-net extras; SDF hidden
 * activations; the adjoints of the d sdf / d x sweep; colour hidden activations) with nL / nC the hidden layer counts of the
 * descriptors, n_tensors = 3 + 2 nL + nC, and e_tiles / outf_tiles / extr_tiles the feature tiles of the first three.
 * Networks of 5..8 feature tiles (the two-image kernel).  out_sdf [P], out_n [P,3], out_rgb [P,3]. */
int vqn_neus_train_fwd(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc, const float* wbuf_col,
                       const float* pts, const float* dirs, int64_t P, void* scratch, int64_t scratch_bytes,
                       float* const* tensors, int n_tensors, int e_tiles, int outf_tiles, int extr_tiles, float* out_sdf,
                       float* out_n, float* out_rgb, void* stream);

/* Backward of the same path in one launch (colour-network backward, tangent pass and reverse sweep of the SDF network with the
 * second-order terms of the normals): what loss.backward() does to renderer.py:216-227 under exp_runner.py:153-168, down to the
 * per-point adjoints the weight-gradient contraction (vqn_wgrad_partials*) sums over points.  desc: int32[76] (feature tiles and
 * pack offsets, geo/train_programs.py NeusTrainEngine._bwd_static); wbuf: the transposed / forward matrices in A-fragment order.
 * g_rgb [P,3], rgb [P,3] (the forward's colours when the colour net ends in a sigmoid, else NULL), g_n [P,3] or NULL, g_sdf [P]
 * or NULL.  saved = [U_1..U_nL, GH_0..GH_{nL-1}, C_1..C_nC] as vqn_neus_train_fwd left them; outs = [DC_0..DC_nC, GOUTF, ED,
 * UD_1..UD_nL, AB_0..AB_{nL-1}] in the same tile format.  Networks of 5..9 feature tiles. */
int64_t vqn_neus_train_bwd_scratch_bytes(const int32_t* desc);
int vqn_neus_train_bwd(const int32_t* desc, const float* wbuf, const float* pts, const float* g_rgb, const float* rgb,
                       const float* g_n, const float* g_sdf, int64_t P, void* scratch, int64_t scratch_bytes,
                       const float* const* saved, int n_saved, float* const* outs, int n_outs, void* stream);

/* vqn_neus_train_bwd on the exact-split engine (csrc/neus_train_bwd_x3.hip): the same pass, saved tensors, adjoints and outputs;
 * every layer GEMM as six bf16 MFMAs per product over bf16 piece triples, the layers in place (networks of at most 256-wide
 * layers).  desc as for vqn_neus_train_bwd with emb_rows in x3 rows (3 per 16 features); wbuf_pieces: the matrices as piece
 * triples in the x3 A-fragment order (geo/packing.py gemm_index_x3 + split_pack_x3 -- vqn_pack_x3_gather builds it in one launch);
 * wbuf_f32: the two thin f32 images (the last layer's sdf row in accumulator order, the normals' row dots). */
int64_t vqn_neus_train_bwd_x3_scratch_bytes(const int32_t* desc);
int vqn_neus_train_bwd_x3(const int32_t* desc, const void* wbuf_pieces, const float* wbuf_f32, const float* pts, const float* g_rgb,
                          const float* rgb, const float* g_n, const float* g_sdf, int64_t P, void* scratch, int64_t scratch_bytes,
                          const float* const* saved, int n_saved, float* const* outs, int n_outs, void* stream);

/* out[t][q][lane][k] (bf16, q = 0..2) = piece q of flat[gidx[t][lane][k]] for t < n_steps (64 lanes x 8 slots per K step): a
 * gather from a flat parameter vector fused with the exact three-way split x = p0 + p1 + p2 into bf16 pieces (truncation of the
 * word, twice on the exact remainders) -- a whole x3 weight pack in one launch. */
int vqn_pack_x3_gather(const float* flat, const int32_t* gidx, int64_t n_steps, void* out, void* stream);
/* ... and, in the same launch, the plain gather out_f32[i] = flat[fidx[i]], i < n_f32, of the thin f32 images the same kernels read
 * (biases in accumulator order, row-dot images: `wbuf_f32` below); n_f32 = 0: the call above. */
int vqn_pack_x3_gather2(const float* flat, const int32_t* gidx, int64_t n_steps, void* out, const int32_t* fidx, int64_t n_f32,
                        float* out_f32, void* stream);

/* Split-precision twins of vqn_neus_sdf_points / vqn_neus_fine_points ("fp16 MFMA path"): same arguments, same outputs,
 * every product taken as hi*hi + 2^-11 (hi*lo + lo*hi) over f16 hi/lo operand pairs on v_mfma_f32_32x32x16_f16 with f32
 * accumulation.  Descriptors and packs must be built for it (SdfPackPlan(mode='f16s'), ColPackPlan(matrix_mode='f16s'):
 * even row counts, f16 hi/lo weight fragments padded to whole 64-feature blocks).  Opt-in: agrees with the f32 entry
 * points to ~1e-6 relative, not bitwise; |weights| < 6e4.  Scratch size: vqn_neus_fine_scratch_bytes. */
int vqn_neus_sdf_points_f16s(const int32_t* sdf_desc, const float* wbuf_sdf, const float* rays_o, const float* rays_d,
                             const float* z, const float* pts, int64_t P, int S, float* out_sdf, void* stream);
int vqn_neus_fine_points_f16s(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc,
                              const float* wbuf_col, const float* rays_o, const float* rays_d, const float* z,
                              const float* pts, const float* dirs, int64_t P, int S, void* scratch,
                              int64_t scratch_bytes, float* out_sdf, float* out_grad, float* out_rgb, void* stream);

/* Exact-split twins of vqn_neus_sdf_points / vqn_neus_fine_points (round 3): same arguments, same outputs, f32-LEVEL results on the
 * bf16 matrix pipe.  Every f32 operand is split exactly into three bf16 pieces (8 + 8 + 8 significant bits, f32 exponent range: no
 * scaling, no range limit) and a product keeps the six cross terms down to 2^-24 -- what an f32 multiply rounds away -- on
 * v_mfma_f32_32x32x16_bf16 with f32 accumulation: 2.7x less matrix-pipe time than the f32-input MFMA of the plain entry points.
 * Replaces the same reference ops (geo/NeuS-ours2/models/renderer.py:337-338,180-185,216-227; fields.py:72-107,147-172).
 * Descriptors and packs must be built for it (SdfPackPlan(mode='x3'), ColPackPlan(matrix_mode='x3'): 3 LDS rows per 16
 * features, bf16 piece triples padded to whole 32-feature K blocks).  Layers of at most 256 outputs (the layers run in place in LDS,
 * one output tile per wave).  Not bitwise the f32 entry points' results (the sums associate differently); held to the f32 tolerances
 * against the reference's goldens (tests/test_gpu_neus_x3.py).  Scratch size: vqn_neus_fine_scratch_bytes. */
int vqn_neus_sdf_points_x3(const int32_t* sdf_desc, const float* wbuf_sdf, const float* rays_o, const float* rays_d,
                           const float* z, const float* pts, int64_t P, int S, float* out_sdf, void* stream);
int vqn_neus_fine_points_x3(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc,
                            const float* wbuf_col, const float* rays_o, const float* rays_d, const float* z,
                            const float* pts, const float* dirs, int64_t P, int S, void* scratch,
                            int64_t scratch_bytes, float* out_sdf, float* out_grad, float* out_rgb, void* stream);

/* vqn_neus_train_fwd on the exact-split engine: x3 packs and descriptors (vqn_neus_pack_create(..., f16s = 2)), the same outputs
 * and saved tensors (the f32 values the epilogues hold before the split into bf16 pieces).  Layers of at most 256 outputs. */
int vqn_neus_train_fwd_x3(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc, const float* wbuf_col,
                          const float* pts, const float* dirs, int64_t P, void* scratch, int64_t scratch_bytes,
                          float* const* tensors, int n_tensors, int e_tiles, int outf_tiles, int extr_tiles, float* out_sdf,
                          float* out_n, float* out_rgb, void* stream);

/* ---- weight packs of the fused NeuS kernels, built in C ------------------------------------------------ */

/* A pack handle owns the device memory of the two weight buffers + their gather tables and the two host descriptors
 * (include/vqn_neus_desc.h) for one (SDFNetwork, RenderingNetwork) shape; these are the only persistent allocations the library
 * makes (explicit create / destroy).  What it replaces in the reference: nothing computes there -- the networks' parameters
 * (fields.py:24-66, :121-140) are consumed by nn.Linear; here they are re-laid as MFMA A-fragments.
 *   sdf_dims [sdf_n_lin + 1] = [3 + 6 multires, hidden..., d_out] (fields.py:24); sdf_skip: the layer whose input is
 *   [h, embedding] / sqrt(2) (fields.py:81-82; -1 none; one skip, not the last layer); col_mode 0 idr, 1 no_view_dir, 2 no_normal
 *   (fields.py:147-158); col_n_layers hidden layers of col_d_hidden (0: no colour network, SDF-only packs); the colour net's
 *   d_feature is d_out - 1.  `f16s` selects the engine the packs are for: 0 = the f32 entry points, 1 = the split-precision
 *   (*_f16s) ones, 2 = the exact-split (*_x3) ones. */
typedef struct vqn_neus_pack vqn_neus_pack;
int vqn_neus_pack_create(const int32_t* sdf_dims, int sdf_n_lin, int sdf_skip, int multires, float scale, int col_mode,
                         int col_d_hidden, int col_n_layers, int multires_view, int squeeze_out, int f16s, vqn_neus_pack** out);
/* Gather the current EFFECTIVE weights (weight-norm applied: vqn_weight_norm_fwd) into the packs: one launch per network.
 * sdf_w[l] [out_l, in_l] row-major, sdf_b[l] [out_l], device pointers in host arrays of sdf_n_lin (col_n_layers + 1) entries;
 * out_l of the layer before the skip is dims[l + 1] - dims[0] (fields.py:38-41).  Call again after every optimiser step. */
int vqn_neus_pack_update(vqn_neus_pack* pack, const float* const* sdf_w, const float* const* sdf_b, const float* const* col_w,
                         const float* const* col_b, void* stream);
const int32_t* vqn_neus_pack_sdf_desc(const vqn_neus_pack* pack);   /* host, 108 ints: `sdf_desc` of the entry points below */
const int32_t* vqn_neus_pack_col_desc(const vqn_neus_pack* pack);   /* host, 80 ints (all zero without a colour network) */
const float* vqn_neus_pack_sdf_wbuf(const vqn_neus_pack* pack);     /* device: `wbuf_sdf` */
const float* vqn_neus_pack_col_wbuf(const vqn_neus_pack* pack);     /* device: `wbuf_col` (NULL without a colour network) */
int64_t vqn_neus_pack_sdf_floats(const vqn_neus_pack* pack);
int64_t vqn_neus_pack_col_floats(const vqn_neus_pack* pack);
void vqn_neus_pack_destroy(vqn_neus_pack* pack);

/* Host-only halves of the above (no device needed): the descriptor and the gather table of one network's pack.  A table entry
 * is 4 int32 per f32 word of the pack: (src, i0, i1, kind) -- src -1: zero, 2 l: element i0 of W_l, 2 l + 1: of bias_l, bit 30:
 * times 1/sqrt(2); kind 0 copy, 1 / 2: the f16 hi / lo halves (hi = f16(w), lo = f16((w - hi) 2^11)) of elements (i0, i1).
 * Return the number of words (entries are written only if words_cap is large enough; desc_out / words_out may be NULL), or a
 * negative error code.  with_reverse: include the transposed packs of the input-gradient sweep. */
int64_t vqn_neus_sdf_pack_plan(const int32_t* dims, int n_lin, int skip, int multires, float scale, int max_tiles, int with_reverse,
                               int f16s, int32_t* desc_out, int32_t* words_out, int64_t words_cap);
int64_t vqn_neus_col_pack_plan(int d_feature, int mode, int d_hidden, int n_layers, int d_out, int multires_view, int squeeze_out,
                               int feat_tiles, int f16s, int32_t* desc_out, int32_t* words_out, int64_t words_cap);

/* The whole inference path of vq_nfr.Model.call up to the shading (vq_nfr.py:534-692: encoder -> z -> continuous heads; l2-normalise
 * -> nearest code -> straight-through rows -> VQ heads) in ONE launch for K <= 64, z_dim = 256: program A (desc_a / wbuf_a: positional
 * encoding -> fine_enc -> bottleneck -> heads, z through output slot 0) leaves z in LDS, the VQ step runs on it with the arithmetic
 * of vqn_vq_quantize_rows (bit-identical indices and straight-through rows), program B (desc_b / wbuf_b: a head family on a raw
 * 256-feature input kept resident) reads the straight-through rows from LDS.  z and the quantised rows never reach HBM
 * (outs_a[0] may be NULL; pass a pointer to get z as well).  outs_* / ld_*: VQN_CHAIN_MAX_OUTS (4) pointers / leading dimensions per
 * program; cb_frags: vqn_vq_codebook_frags of the codebook; idx [N] int64, ste [N, 256] (or NULL: the straight-through rows stay on
 * the chip), *loss = loss_scale * sum (q - z^)^2, counts [K] as vqn_vq_quantize_rows; ws: VQN_QUANT_WS_FLOATS floats of device scratch. */
int vqn_mlp_chain_vq_fwd(const int32_t* desc_a, const float* wbuf_a, const int32_t* desc_b, const float* wbuf_b, const float* in,
                         int64_t N, float* const* outs_a, const int32_t* ld_a, float* const* outs_b, const int32_t* ld_b,
                         const float* cb_frags, int K, float eps, float loss_scale, int64_t* idx, float* ste, float* loss,
                         float* counts, float* ws, void* stream);
/* Codebook [256, K] (K <= 64: 1, 2 or 4 tiles of 16 codes) -> the MFMA B fragments + |c|^2 the fused kernel reads
 * (VQN_VQ_FRAGS_FLOATS(K) floats, device); call again whenever the codebook changes. */
#define VQN_VQ_FRAGS_FLOATS(K) (((K) <= 16 ? 1 : ((K) <= 32 ? 2 : 4)) * (16 * 64 * 4 + 16))
int vqn_vq_codebook_frags(const float* codebook, int D, int K, float* frags, void* stream);

/* Replaces `linear2srgb` (nerfactor/util/img.py:142-186) as applied to the rendered colours (vq_nfr.py:736-745, nfr_unit.py:302-306):
 * y = clip(x, 0, 1) through the piecewise sRGB curve, one pass over n floats. */
int vqn_linear2srgb(const float* x, int64_t n, float* y, void* stream);

/* ---- layer programs + weight packs of the Dense-stack evaluator, built in C ---------------------------------- */

/* vqn_mlp_chain_fwd takes a layer program (include/vqn_chain_desc.h) and a weight pack.  These entries build both from a
 * declarative description of the networks -- what vqnerf_release_amd/decomp/packing.py does for the Python host (f32 kernels):
 *   kind 0: a Dense chain (networks/mlp.py:24-50): `n_layers` layers of `widths[i]` with activations `acts[i]` (0 none, 1 relu,
 *           2 softplus(beta = 100), 3 sigmoid); after layer `skip_at` (or -1) the output is concat(y, stack input) (mlp.py:47-48)
 *   kind 1: a reflectance head (nfr_unit.py:110-129 / vq_nfr.py:135-164): three Dense layers, the stack input concatenated into the
 *           last one (skip_at = [1]), <= 4 outputs, middle width <= 128; built with its input resident in LDS
 * `input`: -1 = the program input ([N, in_feats] raw rows, or the positional encoding of [N, 3] points, embedder.py:23-47), or the
 * index of an earlier kind-0 stack whose output it reads from LDS; `out_slot` (0..3, or -1): which output of vqn_mlp_chain_fwd the
 * stack's result leaves through.  The reference's encoder is {kind 0 fine_enc, kind 0 bottleneck (input 0, slot 0)} on in_mode 1;
 * a head family is three kind-1 stacks on in_mode 0 -- or stacked after the encoder in the same program (input 1, slots 1..3).
 * Weights are passed layer by layer in stack order, Keras layout: kernel [in, out], bias [out] (device pointers). */
#define VQN_CHAIN_MAX_STACKS 8
typedef struct vqn_chain_stack {
  int32_t kind, n_layers;
  int32_t widths[8], acts[8];
  int32_t skip_at, input, out_slot;
} vqn_chain_stack;
typedef struct vqn_chain_pack vqn_chain_pack;
int vqn_chain_pack_create(int in_mode, int in_feats, int n_freqs, int n_stacks, const vqn_chain_stack* stacks, vqn_chain_pack** out);
/* Gather the current weights into the pack (one launch) and refresh the descriptor (the <= 4-output layers keep their biases in
 * it: one small device -> host copy, the call synchronises `stream`).  kernels / biases: host arrays of vqn_chain_pack_n_weights
 * device pointers. */
int vqn_chain_pack_update(vqn_chain_pack* pack, const float* const* kernels, const float* const* biases, void* stream);
int vqn_chain_pack_n_weights(const vqn_chain_pack* pack);
const int32_t* vqn_chain_pack_desc(const vqn_chain_pack* pack);     /* host, 272 ints: `desc` of vqn_mlp_chain_fwd */
const float* vqn_chain_pack_wbuf(const vqn_chain_pack* pack);       /* device: `wbuf` */
int64_t vqn_chain_pack_floats(const vqn_chain_pack* pack);
void vqn_chain_pack_destroy(vqn_chain_pack* pack);
/* Host-only half (no device needed): descriptor (without the small layers' biases) and gather table, as vqn_neus_*_pack_plan. */
int64_t vqn_chain_pack_plan(int in_mode, int in_feats, int n_freqs, int n_stacks, const vqn_chain_stack* stacks, int32_t* desc_out,
                            int32_t* words_out, int64_t words_cap);

/* ---- per-ray NeuS kernels (geo/NeuS-ours2/models/renderer.py) ------------------------------ */

/* Replaces NeuSRenderer.up_sample (renderer.py:131-175) incl. sample_pdf(det=True) (:39-69):
 * importance weights at a fixed inv_s from (z, sdf) [B,n], then n_new inverse-CDF samples per ray.
 * u [n_new] = the deterministic quantiles (torch.linspace(.5/n_new, 1-.5/n_new, n_new)).
 * z_new [B,n_new].  2 <= n <= 256, n_new <= 64. */
int vqn_neus_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int64_t B, int n,
                      float r_limit, float inv_s, const float* u, int n_new, float* z_new, void* stream);

/* Replaces NeuSRenderer.cat_z_vals (renderer.py:177-191): cat + sort + index-gather, as a stable
 * merge of the (sorted) old samples with the new ones, old first on equal keys; sdf follows z.
 * z_out [B,n+n_new]; sdf/sdf_new/sdf_out may all be NULL (last up-sampling step). */
int vqn_neus_merge(const float* z, const float* sdf, const float* z_new, const float* sdf_new, int64_t B, int n,
                   int n_new, float* z_out, float* sdf_out, void* stream);

/* Replaces renderer.py:209-213: dists = diff(z) ++ sample_dist, mid_z = z + dists/2.
 * sample_dist_per_ray [B] (the to_light variant, :211) or NULL to use the scalar. dists may be NULL. */
int vqn_neus_section_mids(const float* z, int64_t B, int n, float sample_dist, const float* sample_dist_per_ray,
                          float* mid_z, float* dists, void* stream);

/* Replaces renderer.py:229-282 (n_outside == 0): iter_cos, estimated prev/next sdf, sigmoids, alpha
 * clip, inside/relax sphere tests, transmittance cumprod, colour / surf / depth / weight sums, and
 * the per-ray eikonal partial sums.  Per-sample inputs are [B,n(,3)] at the section mid-points;
 * inv_s is a DEVICE scalar (exp(10*variance), clipped here to [1e-6,1e6]); background_rgb [3] or NULL.
 * Outputs: color [B,3], weights [B,n], cdf [B,n], inside [B,n], surf [B,3], depth [B],
 * weight_sum [B], weight_max [B], gerr [B,2] = (sum relax*(|g|-1)^2, sum relax); alpha [B,n] or NULL. */
int vqn_neus_composite_fwd(const float* rays_o, const float* rays_d, const float* mid_z, const float* dists,
                           const float* sdf, const float* grad, const float* rgb, const float* inv_s,
                           const float* background_rgb, int64_t B, int n, float radius, float cos_anneal_ratio,
                           float* color, float* weights, float* cdf, float* inside, float* surf, float* depth,
                           float* weight_sum, float* weight_max, float* gerr, float* alpha, void* stream);

/* Reverse of vqn_neus_composite_fwd (autograd in the reference: loss.backward() through renderer.py:229-282)
 * for the outputs a loss touches: d loss / d color [B,3] (required), d loss / d weight_sum [B] (or NULL),
 * d loss / d weights [B,n] (or NULL), d loss / d gradient_error (device scalar or NULL; gerr_den = device scalar
 * sum of gerr[:,1] over the rays of the batch).  Outputs: g_sdf [B,n], g_grad [B,n,3] (w.r.t. the SDF
 * gradients fed to the forward), g_rgb [B,n,3], g_inv_s [B] (per-ray partials of d loss / d inv_s; 0 outside
 * the [1e-6,1e6] clip). */
int vqn_neus_composite_bwd(const float* rays_o, const float* rays_d, const float* mid_z, const float* dists,
                           const float* sdf, const float* grad, const float* rgb, const float* inv_s,
                           const float* background_rgb, int64_t B, int n, float radius, float cos_anneal_ratio,
                           const float* g_color, const float* g_weight_sum, const float* g_weights,
                           const float* g_gradient_error, const float* gerr_den, float* g_sdf, float* g_grad,
                           float* g_rgb, float* g_inv_s, void* stream);

/* ---- training engine of the fused NeuS networks ------------------------------------------------- */

/* Tile-program interpreter (csrc/tile_vm.hip, descriptor include/vqn_vm_desc.h): runs a host-built op list over
 * the LDS activation image of 32-point tiles.  The training passes the reference gets from autograd
 * (loss.backward() through fields.py:72-107,147-172 with create_graph=True for the eikonal term) are such
 * programs (vqnerf_release_amd/geo/train_programs.py): forward with saved activations, colour-net reverse
 * sweep, SDF tangent pass + reverse sweep with second-order source terms.
 *   desc_dev / desc_host: the same descriptor in device and host memory (the host copy is validated);
 *   tensors[i] / tensor_ld[i]: device pointers and leading dims (VEC: row stride, TFMT: feature tiles; negative: see
 *   vqn_tile_program_grid). */
int vqn_tile_program(const void* desc_dev, const int32_t* desc_host, const float* wbuf, float* const* tensors,
                     const int32_t* tensor_ld, int n_tensors, int64_t N, void* stream);
/* Number of workgroups vqn_tile_program launches for this program and N points.  A tensor passed with a NEGATIVE tensor_ld
 * (-feature_tiles) is a per-workgroup temporary of the program: written and read back for the same tile by the same workgroup
 * (GEMM aux / store operands only), it is addressed by workgroup instead of by tile and needs only [grid][tiles][32][32] floats --
 * small enough to stay in L2 / Infinity Cache instead of streaming [n_tiles] images to HBM and back. */
int64_t vqn_tile_program_grid(const int32_t* desc_host, int64_t N);

/* Weight normalisation of all layers of a network in one launch and its backward in another (the geo trainer's per-step
 * chain rule through nn.utils.weight_norm, fields.py:65-66 / :139-140: w = g * v / ||v||_row).  Host arrays of n_layers
 * (<= 24) device pointers: v [rows_l, cols_l], g [rows_l], w / dw / dv [rows_l, cols_l], dg [rows_l]; contiguous fp32. */
int vqn_weight_norm_fwd(int n_layers, const float* const* v, const float* const* g, float* const* w, const int32_t* rows,
                        const int32_t* cols, void* stream);
int vqn_weight_norm_bwd(int n_layers, const float* const* v, const float* const* g, const float* const* dw, float* const* dv,
                        float* const* dg, const int32_t* rows, const int32_t* cols, void* stream);

/* Row-major [N, F] (row stride ldx floats) <-> the feature-major tile format of the training programs
 * (TFMT [ceil(N/32)][tiles_f][32 features][32 points], include/vqn_vm_desc.h), zero padded on pack.  In the reference these
 * hand-offs are implicit (autograd passes dense [N, F] tensors between the Keras layers, the VQ layer and the renderer). */
int vqn_tfmt_pack(const float* x, int64_t N, int F, int64_t ldx, float* t, int tiles_f, void* stream);

/* The top-layer delta of a backward program in one launch: t = g act'(y) with y the layer's saved TFMT output (act 0: t = g,
 * 1 ReLU: g [y > 0], 3 sigmoid: (g y) (1 - y), each product separately rounded); g [N, F] rows with row stride ldg, or NULL for
 * zeros (an unused head: autograd hands no gradient).  Same tile format and padding rule as vqn_tfmt_pack. */
int vqn_tfmt_pack_delta(const float* g, int64_t N, int F, int64_t ldg, const float* y_tfmt, int act, float* t, int tiles_f, void* stream);
int vqn_tfmt_unpack(const float* t, int tiles_f, int64_t N, int F, float* x, int64_t ldx, void* stream);

/* Weight-gradient contraction over points (autograd's grad_weight GEMMs): partial blocks
 * ws[s][a_nt*32][b_nt*32] = sum over the point tiles of split s of A[o][p] * B[i][p], A/B in TFMT with
 * a_tiles/b_tiles feature tiles of which [t0, t0+nt) are used.  Returns the number of partial blocks written
 * (> 0; the caller sums them in order) or a negative error code.  rowsum_ws (optional): [n_split][a_nt*32] partial sums over
 * points of A itself -- the bias gradients, for free while A streams through the registers. */
int vqn_wgrad_partials(const float* A, int a_tiles, int a_t0, int a_nt, const float* B, int b_tiles, int b_t0, int b_nt,
                       int64_t n_point_tiles, int n_split, float* ws, float* rowsum_ws, void* stream);

/* The same contraction with every f32 operand split exactly into three bf16 pieces and the six cross terms down to 2^-24 of |a||b|
 * kept (csrc/wgrad_x3.hip): f32-level products on the bf16 matrix pipe, 2.7x less matrix time for the 256 x 256 blocks (blocks of
 * <= 4 output tiles run the f32 kernel).  Same arguments, workspace and return value. */
int vqn_wgrad_partials_x3(const float* A, int a_tiles, int a_t0, int a_nt, const float* B, int b_tiles, int b_t0, int b_nt,
                          int64_t n_point_tiles, int n_split, float* ws, float* rowsum_ws, void* stream);

/* The contractions of a whole backward pass in as few launches as they have kernel shapes: problem i is vqn_wgrad_partials
 * (x3 == 0) / vqn_wgrad_partials_x3 (x3 != 0) of (A[i], a_tiles[i], a_t0[i], a_nt[i], B[i], ..., ws[i], rowsum_ws[i]) over the same
 * n_point_tiles -- the same kernels, hence the same partial blocks bit for bit -- a launch per kernel variant and 24 problems
 * (the reference's 2048-point steps are bound by their launch count, trainvali.py:443-486).  Returns the number of partial
 * blocks per problem (>= 1) or a negative error code. */
int vqn_wgrad_partials_batched(int count, const float* const* A, const int32_t* a_tiles, const int32_t* a_t0, const int32_t* a_nt,
                               const float* const* B, const int32_t* b_tiles, const int32_t* b_t0, const int32_t* b_nt,
                               int64_t n_point_tiles, int n_split, float* const* ws, float* const* rowsum_ws, int x3, void* stream);

/* The ordered sum of those partial blocks: out[r][c] (+)= sum_s ws[s][r][c], s = 0 .. n-1 (fixed order: deterministic),
 * written into a [rows, cols] window of a matrix with row stride out_ld.  cols, out_ld multiples of 4. */
int vqn_reduce_partials(const float* ws, int n, int rows, int cols, float* out, int64_t out_ld, int accumulate, void* stream);

/* The same ordered sums for MANY contractions in one launch, each result written where it belongs: entry i sums the n[i] partial
 * blocks [src_rows[i], src_cols[i]] at ws[i] (and, if ws2[i] != NULL, the n2[i] blocks at ws2[i], added: the first- and second-order
 * terms of an SDF layer, renderer.py:216-227 under autograd) in vqn_reduce_partials' order (bit-identical), multiplies by scale[i]
 * and stores element (r < rows_valid[i], col_first[i] <= c < cols_valid[i]) at dst[i][r * dst_row_stride[i] + c * dst_col_stride[i]] -- a slice
 * of a concatenated matrix, a transposed ([in, out], the layout of the reflectance nets' kernels) one, a bias row.  Replaces the
 * per-weight reduce / transpose / cat / scale kernel sequences of the training steps (launch-count bound at the reference's
 * 2048-point batches, trainvali.py:443-486). */
int vqn_wgrad_finalize(int count, const float* const* ws, const int32_t* n, const float* const* ws2, const int32_t* n2,
                       const int32_t* src_rows, const int32_t* src_cols, const int32_t* rows_valid, const int32_t* col_first,
                       const int32_t* cols_valid, float* const* dst, const int64_t* dst_row_stride, const int64_t* dst_col_stride, const float* scale,
                       void* stream);

/* Element-wise pieces of the reflectance training graph, one launch each (round 4; csrc/elementwise.hip).
 * vqn_clip_preserve: y = x + (clip(x, lo, hi) - x), every step rounded -- tfp.math.clip_by_value_preserve_gradient in the
 *   reference's own arithmetic (vq_nfr.py:731, :735-745); the gradient is the identity.
 * vqn_ks_split_fwd: spec = ks basecolor, albedo = (1 - ks) basecolor (vq_nfr.py:590-592; basecolor [n, 3], ks [n, 1 | 3]);
 * vqn_ks_split_bwd: their adjoints (g_albedo / g_spec may be NULL = zero). */
int vqn_clip_preserve(const float* x, int64_t n, float lo, float hi, float* y, void* stream);
/* out[i] = the per-point training loss summed in the reference's order (vq_nfr.py:906-981): terms [n, 5] = rgb, vqrgb, chromaticity,
 * chr_smooth, lambert; vqloss / sim device scalars (sim may be NULL); (((rgb + vqrgb) + vqloss) [+ chr] [+ smooth] [+ sim] [+ lambert]). */
int vqn_loss_total(const float* terms, int64_t n, const float* vqloss, const float* sim, int use_chr, int use_smooth, int use_lambert,
                   float* out, void* stream);
int vqn_ks_split_fwd(const float* basecolor, const float* ks, int ks_channels, int64_t n, float* albedo, float* spec, void* stream);
int vqn_ks_split_bwd(const float* basecolor, const float* ks, int ks_channels, int64_t n, const float* g_albedo, const float* g_spec,
                     float* g_basecolor, float* g_ks, void* stream);

/* Thin contractions (round 4; csrc/wgrad_thin.hip): out[r][f] = sum_p A[a_row0 + r][p] B[f][p] for r < a_rows[i] <= 8 rows of ONE
 * feature tile (a_t0[i]) of A against b_nt[i] feature tiles of B -- the weight gradient of a reflectance head's 1..3-output last layer
 * (nfr_unit.py:110-129; the heads of a family keep their rows in one tile: a_row0).  A stream over B on the vector ALU (f32 FMA chains
 * in point order), no matrix pipe.  Partial blocks ws[i], TRANSPOSED: [n][32 b_nt][8] (columns >= a_rows are zero), row sums
 * rowsum_ws[i] (may be NULL): [n][32] (entries >= a_rows zero); n = the return value = min(n_split, n_point_tiles).
 * vqn_wgrad_finalize sums them with src_rows = 32 b_nt, src_cols = 8 (a column range picks a head's rows). */
int vqn_wgrad_thin_batched(int count, const float* const* A, const int32_t* a_tiles, const int32_t* a_t0, const int32_t* a_row0,
                           const int32_t* a_rows, const float* const* B, const int32_t* b_tiles, const int32_t* b_t0, const int32_t* b_nt,
                           int64_t n_point_tiles, int n_split, float* const* ws, float* const* rowsum_ws, void* stream);

/* count contiguous f32 copies dst[i][0..n[i]) = src[i][0..n[i]) in one launch (the per-parameter gradients into the flat
 * gradient bucket of the data-parallel step, trainvali.py:469-477). */
int vqn_multi_copy(int count, const float* const* src, float* const* dst, const int64_t* n, void* stream);

/* ---- training passes of the reflectance Dense stacks on the exact-split engine (round 4; csrc/refl_train_x3.hip) ----
 * Replace `tape.gradient` through `_pred_enc_at` / `_pred_{diff,spec,rough}_at` (decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:771-828,
 * nfr_unit.py:329-391 over networks/{embedder,mlp,seq}.py) under train_nfr.py:562-576: the forward keeping what the backward needs, and
 * the backward down to every Dense layer's per-point adjoint (the weight gradients are vqn_wgrad_partials_batched over those tensors).
 * One stack: [posenc -> n_enc Dense layers with one skip-concat of the encoding -> z] (n_enc = 0: the input is z rows) -> up to three
 * heads Dense(w0) relu -> Dense(w1) relu -> Dense(c <= 3) sigmoid over [y1 ; z].  Layers of at most 256 outputs; every f32 operand as
 * three bf16 pieces, six bf16 MFMAs per product, f32 accumulation (f32-level results).
 * desc: vqn_refl_train_desc_ints() int32 words (decomp/train_programs.py: ReflTrainEngine builds it; csrc/refl_train_x3.hip: ReflDesc);
 * wbuf_pieces: the GEMM matrices as bf16 piece triples in x3 A-fragment order (vqn_pack_x3_gather); wbuf_f32: biases in accumulator
 * order, the heads' last layers as row-dot images and accumulator-order columns.  Tensors in the tile format are
 * [ceil(P/32)][feature tiles][32 features][32 points] f32.
 * forward -- saved (written): with an encoder [E, Y_0 .. Y_{n_enc-1}] (Y_{n_enc-1} = z), without [ZT = the input rows' tile-format
 *   copy]; then [H0_k, H1_k] per head.  z_rows_out [P, z_feats] (optional, with an encoder); head_out[k] [P, c_k].  save_tensors = 0:
 *   INFERENCE on the same kernel (the exact-split reflectance chain, `model.matrix_mode = 'x3'`): nothing is kept but z's tile-format
 *   tensor (Y_{n_enc-1} / ZT: heads 2 and 3 re-read it); the other `saved` entries may be NULL.
 * backward -- g_out[k] / head_out[k] [P, c_k]; g_z_rows: up to four [P, z_feats] adjoints of z from outside this launch's heads (summed
 *   in order); saved [Y_0 .. Y_{n_enc-1}] then [H0_k, H1_k]; outs (written) [D_0 .. D_{n_enc-1}] then [D0_k, D1_k, D2_k] per head.
 *   run_heads / run_enc select the part of the stack the launch walks: both = the whole backward; heads only = d / d z rows into
 *   gz_rows_out; encoder only = from g_z_rows down.  Adjoints of points past P are zero.  d2_row0 (may be NULL): head k writes its
 *   c_k rows of D2_k at row d2_row0[k] of the tile and nothing else -- the heads of a family then share ONE D2 tile tensor (all D2_k
 *   the same pointer), which vqn_wgrad_thin_batched reads by row range; NULL: rows 0..c_k-1, the rest of the tile zeroed.
 * Small batches (the reference's 2048 points are 64 tiles, a quarter of the CUs): the kernels switch by themselves to one 32-point
 *   image per workgroup when tile pairs would fill less than half of the chip, and split_heads != 0 gives every head its own
 *   workgroup row (forward: each row evaluates the encoder, row 0 writes its tensors; backward: heads only, gz_rows_out then holds
 *   n_heads slices [P, z_feats], one per head, for the caller -- or the encoder-only launch's g_z_rows -- to sum). */
int vqn_refl_train_desc_ints(void);
int vqn_refl_train_fwd_x3(const int32_t* desc, const void* wbuf_pieces, const float* wbuf_f32, const float* pts, const float* z_rows,
                          int64_t P, float* const* saved, int n_saved, float* z_rows_out, float* const* head_out, int split_heads,
                          int save_tensors, void* stream);
/* Stage 3 (ref_nfr.py:148-152,203-213: the diffuse / roughness heads read [z_xyz ; z_ref], 2 z_feats wide): descriptor field zx_tiles =
 * z_tiles gives the heads a SECOND input of z_feats features -- zx_rows [P, z_feats], 16-byte aligned; with zx_tiles = 0 it must be NULL
 * and the call is vqn_refl_train_fwd_x3.  The heads' first-layer pack then holds the K segments [z ; zx] (2 z_feats input rows), the last
 * layer a second row-dot image at offW2zx[k].  zx_tiles_out (may be NULL): tile-format copy of zx for the weight-gradient contractions.
 * No adjoint flows into zx (the backward entry is unchanged: d / d z of the z segment only). */
int vqn_refl_train_fwd_x3_zx(const int32_t* desc, const void* wbuf_pieces, const float* wbuf_f32, const float* pts, const float* z_rows,
                             const float* zx_rows, int64_t P, float* const* saved, int n_saved, float* z_rows_out, float* zx_tiles_out,
                             float* const* head_out, int split_heads, int save_tensors, void* stream);
int64_t vqn_refl_train_bwd_x3_scratch_bytes(const int32_t* desc);
int vqn_refl_train_bwd_x3(const int32_t* desc, const void* wbuf_pieces, const float* wbuf_f32, int64_t P, const float* const* g_out,
                          const float* const* head_out, const float* const* g_z_rows, int n_gz, const float* const* saved, int n_saved,
                          float* const* outs, int n_outs, float* gz_rows_out, const int32_t* d2_row0, int run_heads, int run_enc,
                          int split_heads, void* scratch, int64_t scratch_bytes, void* stream);

/* The Adam / AMSGrad update of the reference's optimisers for `count` f32 tensors in one launch per 56 tensors: steps[i] a device
 * float holding tensor i's step count AFTER this step's increment, lr_dev a device scalar (NULL: the host value lr),
 * max_exp_avg_sq NULL without AMSGrad.  exp_avg = b1 m + (1 - b1) g, exp_avg_sq = b2 v + (1 - b2) g^2, then
 *   eps_mode 0 -- torch.optim.Adam (geo/NeuS-ours2/nerf_runner.py:72): p -= (lr / (1 - b1^t)) m / (sqrt(max v) / sqrt(1 - b2^t) + eps)
 *   eps_mode 1 -- Keras Adam(amsgrad=True) (decomp/nerfvq_nfr3/nerfactor/train_nfr.py:121-139; TensorFlow 2.4.1
 *                 ResourceApplyAdamWithAmsgrad): p -= (lr sqrt(1 - b2^t) / (1 - b1^t)) m / (sqrt(max v) + eps)
 * weight_decay is L2 (added to g), maximize negates g.  Both are restated in oracle/optim.py. */
int vqn_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  float* const* max_exp_avg_sq, const float* const* steps, const int64_t* numel, const float* lr_dev, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int maximize, int eps_mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VQNERF_HIP_H_ */
