/* libdanet_hip.so -- C ABI of the MI355X-native DaNet hot path.
 *
 * The reference (HongwenZhang/DaNet-DensePose2SMPL) is pure Python and has no FFI: its
 * de-facto boundary is a set of Python call signatures.  Each entry point
 * below replaces the arithmetic behind one of them; the Python host in
 * danet-densepose2smpl_amd/ keeps the reference's class / method names and calls these
 * through ctypes (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted "host";
 *  - the caller owns all memory (tensors, workspaces); nothing is allocated or freed here;
 *  - every call only enqueues work on `stream` (a hipStream_t passed as void*); no call
 *    synchronises the device, so everything is hipGraph-capturable;
 *  - return value 0 = ok, negative = error; the message is in danet_last_error()
 *    (thread-local).  No C++ exceptions cross the boundary.
 */
#ifndef DANET_HIP_H
#define DANET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DANET_OK 0
#define DANET_ERR_ARG (-1)       /* invalid argument / unsupported shape */
#define DANET_ERR_HIP (-2)       /* a HIP runtime call failed */
#define DANET_ERR_WORKSPACE (-3) /* workspace too small */

int danet_version(void);
const char* danet_last_error(void);

/* ---------------------------------------------------------------------------------------
 * SMPL layer.  Replaces the arithmetic behind SMPL.forward
 * (/root/reference/models/smpl.py:27-46 -> smplx.SMPL.forward / smplx.lbs.lbs, and
 * vertices2joints(J_regressor_extra, vertices) at models/smpl.py:30).
 *
 * Inputs  betas [B,NB] f32, rotmats [B,24,3,3] f32 (row-major; global_orient first).
 * Model   v_template [V,3], shapedirs [V*3,NB], posedirs [207,V*3], J_template [24,3] and
 *         J_shapedirs [24*3,NB] (= J_regressor . v_template / shapedirs, folded once at model
 *         load), lbs_weights [V,24], parents [24] i32 (parents[0] < 0),
 *         J_regressor_extra [NE,V], landmark_verts [NL] i32.
 * Outputs verts [B,V,3]; joints54 [B,24+NL+NE,3] = 24 posed joints, NL landmark vertices,
 *         NE extra regressed joints; ctx [danet_smpl_lbs_ctx_floats(B)] and
 *         v_posed [B,V,3] are saved for the backward (v_posed may be NULL for inference).
 * ws      scratch of danet_smpl_lbs_fwd_ws_floats(B,V,NE) floats.
 * ticket  danet_smpl_lbs_ticket_words(B) uints of device memory, zeroed ONCE by the caller and from then on owned by these
 *         launches (arrival counters, reset by the kernel itself), which must not overlap -- one buffer per stream.  With a
 *         ticket the forward is ONE kernel launch (smpl_fused_fwd_kernel: chain, blend shapes, skinning, landmarks and the
 *         regressed joints, the last workgroup of a batch group to arrive sums the per-tile partials in a fixed order);
 *         NULL runs the three-launch form (prep -> main -> finalize).  Both give the same results.
 */
size_t danet_smpl_lbs_ctx_floats(int B);
size_t danet_smpl_lbs_fwd_ws_floats(int B, int V, int NE);
size_t danet_smpl_lbs_bwd_ws_floats(int B, int V, int NB);
size_t danet_smpl_lbs_ticket_words(int B);

int danet_smpl_lbs_forward(const float* betas, const float* rotmats, int B,
                           const float* v_template, const float* shapedirs, const float* posedirs,
                           const float* J_template, const float* J_shapedirs,
                           const float* lbs_weights, const int32_t* parents,
                           const float* J_regressor_extra, const int32_t* landmark_verts,
                           int V, int NB, int NL, int NE,
                           float* verts, float* joints54, float* ctx, float* v_posed,
                           float* ws, size_t ws_floats, void* ticket, void* stream);

/* Gradient w.r.t. betas and rotmats (the reference's differentiable call site is
 * /root/reference/models/danet/smpl_regressor.py:176).  g_verts [B,V,3] and
 * g_joints54 [B,24+NL+NE,3] may each be NULL (= zero).  Outputs g_betas [B,NB],
 * g_rotmats [B,24,3,3].
 * bar (NULL = none): the barrier state of danet_bn_backward_onepass (danet_bn_backward_onepass_bar_words() uints, zeroed once,
 * owned by the launches of ONE stream).  With it, and when the ntiles x ceil(B / 8) workgroups fit the co-residency budget
 * (danet_smpl_lbs_backward_fused_ok; max_blocks as for danet_bn_backward_onepass, <= 0 = the whole device), the backward pass is
 * ONE kernel launch (smpl_fused_bwd_kernel: its three phases separated by grid-wide barriers; a barrier that is not met sets
 * bar[2], as there); otherwise three launches.  Both forms produce bit-identical results. */
int danet_smpl_lbs_backward(const float* betas, const float* rotmats, int B,
                            const float* shapedirs, const float* posedirs, const float* J_shapedirs,
                            const float* lbs_weights, const int32_t* parents,
                            const float* J_regressor_extra, const int32_t* landmark_verts,
                            int V, int NB, int NL, int NE,
                            const float* ctx, const float* v_posed,
                            const float* g_verts, const float* g_joints54,
                            float* g_betas, float* g_rotmats,
                            float* ws, size_t ws_floats, void* bar, int max_blocks, void* stream);
int danet_smpl_lbs_backward_fused_ok(int B, int V, int max_blocks);
/* profiling aid: clock64 phase stamps of workgroup (0,0) of the last backward launch (16 values) */
int danet_smpl_lbs_debug(long long* out16);

/* ---------------------------------------------------------------------------------------
 * IUV renderer.  Replaces IUV_Renderer.verts2uvimg
 * (/root/reference/utils/renderer.py:256-278 -> neural_renderer.Renderer, 'projection'
 * camera, flat per-face colour, no anti-aliasing, fill_back=False).  Forward only: the
 * reference renders detached label meshes (models/danet/danet.py:163-165).
 *
 * verts [B,NV,3] f32, cam [B,3] f32 (s,tx,ty), vert_mapping [NDV] i32 (DensePose vertex ->
 * SMPL vertex), faces [F,3] i32 (into the NDV vertices), tex [F,3] f32; focal (5000),
 * orig (INIMG_SIZE), S (HEATMAP_SIZE).  out [B,3,S,S] f32; face_idx [B,S,S] i32 (-1 = bg)
 * and depth [B,S,S] f32 (+inf = bg) may be NULL.  ws: 8-byte-aligned scratch of
 * danet_iuv_raster_ws_bytes(B,NDV,S) bytes (projected vertices + 64-bit depth/id buffer).
 */
size_t danet_iuv_raster_ws_bytes(int B, int NDV, int S);
int danet_iuv_raster_forward(const float* verts, const float* cam, int B, int NV,
                             const int32_t* vert_mapping, int NDV,
                             const int32_t* faces, const float* tex, int F,
                             float focal, float orig, int S,
                             float* out, int32_t* face_idx, float* depth,
                             void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Geometry helpers (/root/reference/utils/geometry.py).
 *  danet_batch_rodrigues          geometry.py:9-45   theta [N,3] -> R [N,3,3] (via quaternion)
 *  danet_rodrigues_smplx          smplx.lbs.batch_rodrigues (pose2rot=True inside SMPL.forward)
 *  danet_rot6d_to_rotmat_*        geometry.py:47-61  x [N,6] (viewed [N,3,2]) -> R [N,3,3]
 */
int danet_batch_rodrigues(const float* theta, int N, float* R, void* stream);
int danet_rodrigues_smplx(const float* theta, int N, float* R, void* stream);
int danet_rot6d_to_rotmat_forward(const float* x, int N, float* R, void* stream);
int danet_rot6d_to_rotmat_backward(const float* x, const float* gR, int N, float* gx, void* stream);

/* ---------------------------------------------------------------------------------------
 * The regressor's graph tail as ONE launch per direction (csrc/gcn_tail.hip).  Replaces the torch operations of
 * /root/reference/models/danet/smpl_regressor.py:846-900 (training mode, REFINE_STRATEGY 'gcn', REFINE_ON, POS_INTERSUPV) with
 * models/module/GCN.py:12-92 and utils/geometry.py:47-61 behind them: pose head 0 -> rot6d; r2p graph convolution; coordinate head 0;
 * three refinement graph convolutions on D^-1/2 (I + A_mask relu(edge_importance)) D^-1/2 with the residual; coordinate head 1; p2r
 * graph convolution; pose head 1 -> rot6d.  Every graph convolution is (adjacency x features) x W + b -> BatchNorm1d(24) (training
 * mode: batch statistics, running statistics updated) -> ReLU.  All tensors fp32, contiguous.
 * `args`: struct danet_gcn_tail_args below.  forward reads x .. mean_pose, writes ws, jr0 [B,216], jp0 / jp1 [B,24,3], pose [B,216]
 * and the running statistics; backward reads the same inputs, ws as the forward left it and g_* (each may be NULL = zero), and
 * writes every g* buffer completely.  bar: DANET grid-barrier state (danet_bn_backward_onepass_bar_words() uints, zeroed once; shared
 * with the other barrier kernels of the SAME stream).  1 <= B <= danet_gcn_tail_max_batch() (a caller with more rows uses its
 * unfused operations).  24 workgroups, one per joint. */
struct danet_gcn_layer { const float* W; const float* bias; const float* gamma; const float* beta; float* running_mean; float* running_var; };
struct danet_gcn_tail_args {
    const float* x;                                   /* [B,24,128] limb features */
    struct danet_gcn_layer L[5];                      /* r2p 128->128, refine 128->256->256->128, p2r 128->128; W [in][out] */
    const float* A_r2p; const float* A_p2r; const float* A_mask; const float* edge;    /* [24,24] each */
    const float* Wp[2]; const float* bp[2];           /* pose heads [24,6,128], [144] */
    const float* Wc[2]; const float* bc[2];           /* coordinate heads [24,3,128], [72] */
    const float* mean_pose;                           /* [144] */
    float* ws;                                        /* danet_gcn_tail_ws_floats(B) floats: forward -> backward */
    float* jr0; float* jp0; float* jp1; float* pose;
    const float* g_jr0; const float* g_jp0; const float* g_jp1; const float* g_pose;
    float* gx; float* gW[5]; float* gb[5]; float* ggamma[5]; float* gbeta[5]; float* gedge;
    float* gWp[2]; float* gbp[2]; float* gWc[2]; float* gbc[2];
    float* scratch;                                   /* danet_gcn_tail_scratch_floats(B) floats (backward only) */
    unsigned* bar;
    int B; float momentum, eps;
};
size_t danet_gcn_tail_ws_floats(int B);
size_t danet_gcn_tail_scratch_floats(int B);
int danet_gcn_tail_max_batch(void);
int danet_gcn_tail_debug(long long* out32);          /* diagnostic: phase time stamps of the last launches (host array of 32) */
int danet_gcn_tail_forward(const void* args, void* stream);
int danet_gcn_tail_backward(const void* args, void* stream);

/* ---------------------------------------------------------------------------------------
 * Global (25-class) IUV glue (csrc/iuv_ops.hip).  Replaces utils/iuvmap.py:6-38,103-147 (iuvmap_clean, iuv_img2map),
 * models/danet/iuv_estimator.py:304-341 (body_uv_losses) and danet.py:194-205,247 (part drop, clean, concat), and
 * utils/keypoints.py:334-394 (soft-argmax of the joint heat-maps).
 *  u, v, ix: fp32 [B*H*W][ld] (25 valid channels, 28 <= ld <= 32, ld % 4 == 0); an: [B*H*W][lda = 16] (15 valid);
 *  gt: rendered IUV image [B,3,H,W] fp32 NCHW (want_loss only); w: [B] per-sample weights or NULL; keep: [B,25] part-drop
 *  mask or NULL.  forward: map = bf16 [B*H*W][80] (U*onehot | V*onehot | onehot | 5 zeros; one-hot of argmax(ix*keep)),
 *  am_raw / am_drop = uint8 argmax of ix / ix*keep, sums[4] (DOUBLES: order-independent accumulation) += (sum smooth-L1 U, V at the ground-truth part's channel,
 *  sum CE of the 25-way index, sum CE of the 15-way Ann logits), weighted by w.  backward: coef[4] = dL/dsums (device),
 *  dmap = gradient of map or NULL; du, dv, di ([..][ld]) and da ([..][lda]) are fully written.
 *  softargmax: hm fp32 [B*H*W][ld] (J valid) -> out [B,J,2] = E[(x, y)] under softmax(scale*hm); saved [B,J,4] feeds the
 *  backward, which writes dhm as dense [B*H*W][J]. */
int danet_iuv_global_forward(const float* u, const float* v, const float* ix, const float* an, int ld, int lda,
                             const float* gt, const float* w, const float* keep, int B, int H, int W, int want_loss,
                             void* map, unsigned char* am_raw, unsigned char* am_drop, double* sums, void* stream);
int danet_iuv_global_backward(const float* u, const float* v, const float* ix, const float* an, int ld, int lda,
                              const float* gt, const float* w, const float* keep, const unsigned char* am_drop,
                              const void* dmap, const float* coef, int B, int H, int W, int want_loss,
                              float* du, float* dv, float* di, float* da, void* stream);
int danet_softargmax_forward(const float* hm, int ld, int B, int J, int H, int W, float scale, float* out, float* saved, void* stream);
int danet_softargmax_backward(const float* hm, int ld, int B, int J, int H, int W, float scale, const float* saved,
                              const float* gout, float* dhm, void* stream);

/* ---------------------------------------------------------------------------------------
 * Launch-count glue (csrc/glue.hip).
 * danet_pad_multi: n (<= 16) zero-pad / crop copies of small dense fp32 tensors (<= 4-d, shapes given with leading ones) in one
 *   launch: dst[i] = src[i] inside the source's shape, 0 elsewhere.  Replaces F.pad on the parameters of layers whose widths are no
 *   multiple of 8 (the reference runs those widths as they are: models/module/res_module.py:364 Bottleneck(48, 12), the 25 / 15 /
 *   21-channel heads of models/module/hr_module.py:447-470).
 * danet_stn_theta_forward: models/danet/iuv_estimator.py:262-301 (affine_para) together with the visibility score of :176-186 (single-point
 *   bilinear sample of the per-joint part-membership of the arg-max index plane): centres [B,24,2] -> thetas [B,24,2,3].  No
 *   gradient (theta is detached before affine_grid, iuv_estimator.py:197).
 */
int danet_pad_multi(const void* const* src, void* const* dst, const int* sdims, const int* ddims, int n, void* stream);
/* Loss bookkeeping of the estimator (reference models/danet/iuv_estimator.py:325-339, 233-256) as one launch per pass:
 * forward  (sums != NULL): out[i] = (sum over rows r of sums[r][i], DOUBLES) * a[i] / (b[i] > 0 ? max(sum(w[0..nw)), 1) * b[i] : 1);
 * backward (sums == NULL): out[i] = (grads[i] ? *grads[i] : 0) * a[i] / (the same divisor) -- the coefficient vector of the loss kernels'
 * backward.  n <= 8; a, b, grads are HOST arrays; w (device, or NULL = nw ones) are the per-sample weights. */
int danet_loss_finalize(const void* sums, int rows, int n, const float* a, const float* b, const float* w, int nw,
                        const void* const* grads, float* out, void* stream);
/* danet_regroup_parts: the 24 part crops of an image as 24 channel groups of one map (models/danet/smpl_regressor.py:826
 *   `limb_feat.view(nbs, -1, h, w)`) on NHWC tensors: x [NB * J][HW][row_bytes] -> y [NB][HW][J][row_bytes]; inverse != 0: the other way
 *   (its gradient).  row_bytes (a pixel's channels of one crop) % 16 == 0; any element type.
 * danet_pack_image: x [B, C, H, W] fp32 NCHW (C <= 8) -> y [B, H, W, 8] bf16 NHWC, channels C .. 7 zero: the first convolution's operand. */
int danet_regroup_parts(const void* x, void* y, int NB, int J, int HW, int row_bytes, int inverse, void* stream);
int danet_pack_image(const float* x, void* y, int B, int C, int H, int W, void* stream);
/* /root/reference/models/smpl.py:31-37 (joints = joints54[:, JOINT_MAP]; smpl_joints = joints54[:, :24]; joints_J19 = joints[:, -24:][:, J24_TO_J19]):
 * one launch forward, one backward (the three gradients scattered and summed into g54; NULL = zero). */
int danet_smpl_joints_forward(const float* j54, const long* map49, const long* map19, int B, int NJ54, int N49, int N19,
                              float* j49, float* j19, float* j24, void* stream);
int danet_smpl_joints_backward(const float* g49, const float* g19, const float* g24, const long* map49, const long* map19,
                               int B, int NJ54, int N49, int N19, float* g54, void* stream);
int danet_stn_theta_forward(const float* centers, const unsigned char* am, const float* member, const float* ratio,
                            const float* offset, const float* rnd, const long* child, const long* parent, int B, int H, int W,
                            int align, float jitter, float vis_score, float* theta, void* stream);

/* ---------------------------------------------------------------------------------------
 * SMPL-side losses of the regressor (csrc/loss_ops.hip; models/danet/smpl_regressor.py:141-218,233-298): joint_rotation{0,1},
 * joint_position{0,1}, keypoints_2d (weak-perspective camera -> translation -> pin-hole projection), keypoints_3d (pelvis-
 * centred), smpl_pose, smpl_betas, smpl_verts, cam -- masked means over the rows selected by has_smpl / has_kp3d, times
 * the yaml weights.  `params` / `grads` are HOST structs of device pointers and scalars:
 *   params { const float* para, *target [B,229]; const float* jrot[2] [B,216]; const float* jpos[2] [B,72]; const float* gt_pts [B,72];
 *            const float* joints [B,49,3], *verts, *tverts [B,V,3] (verts NULL when its weight is 0); const float* kps2d [B,49,3],
 *            *kps3d [B,24,4], *has_smpl, *has_kp3d [B]; int B, V; float focal, img, op_w, gt_w; float w[10], cnt[10]; }
 *   grads  { float* dpara; float* djrot[2]; float* djpos[2]; float* djoints; float* dverts; }   (shapes of the inputs; NULL = skip)
 * forward: ps [B,10] scratch, out [10] losses, norm [10] (feeds the backward); backward: gout [10]. */
size_t danet_smpl_loss_param_bytes(void);
size_t danet_smpl_loss_grad_bytes(void);
int danet_smpl_loss_forward(const void* params, float* ps, float* out, float* norm, void* stream);
int danet_smpl_loss_backward(const void* params, const float* gout, const float* norm, const void* grads, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer (replaces torch.optim.Adam at /root/reference/train/trainer.py:42-44): one launch over a device table of
 * <= 32768-element chunks { float* p; const float* g (NULL = skip); int64 off (into m, v); int32 n; int32 param (2 * parameter
 * index + 1 for the parameter's first chunk) }.  lr and step (1-based GLOBAL count, float) are read from device memory; p, g,
 * m+off, v+off 16-byte aligned.  Per-parameter step counts as torch.optim.Adam keeps them: used (NULL = all): float per
 * parameter, the number of ranks in which it received a gradient this step (summed with the gradients), 0 -> the parameter is
 * skipped (moments untouched) and idle[param] (float per parameter, NULL = none, maintained by the kernel) counts it; bias
 * corrections use step - idle[param].  grad_scale multiplies every gradient (1 / world size: all-reduced sums become the
 * average without a pass of its own).  poison (NULL = none): device int, non-zero = the step's gradients are invalid (this
 * device's one-pass BatchNorm backward barrier error word, danet_bn_backward_onepass): every parameter is skipped and counted
 * idle.  poison_sum (NULL = none): device float, the SUM of that word over the data-parallel ranks (all-reduced with the last
 * gradient bucket): > 0 skips the step the same way -- on every rank, so the replicas cannot diverge. */
size_t danet_adam_chunk_bytes(void);
int danet_adam_step(const void* table, int nchunks, float* m, float* v, const float* lr, const float* step,
                    const float* used, float* idle, float beta1, float beta2, float eps, float grad_scale,
                    const int* poison, const float* poison_sum, void* stream);

/* ---------------------------------------------------------------------------------------
 * Partial-IUV ("limb") path glue (replaces /root/reference/models/danet/danet.py:264-283 and
 * /root/reference/models/danet/iuv_estimator.py:206-246, ~40 tensor ops on [B,24,3,7,H,W] fp32).
 * pred: the grouped conv's output, NHWC bf16 [B,H,W,24*cpj], channel = joint*cpj + {u,v,index}*7 + class; cpj = 21, or 24
 * when the conv's group-padded output is consumed as it is (the 3 padding channels per joint get zero gradients).
 *  danet_part_clean_*   x24 [B*24,H,W,24] bf16 = iuvmap_clean(keep[B,24,7] * pred) (+3 zero channels);
 *                       backward: d pred from d x24 (U,V channels only).
 *  danet_part_loss_*    sums [32][3] DOUBLES (order-independent accumulation; zeroed by the caller; column sums = smooth-L1 U, smooth-L1 V,
 *                       index cross-entropy) against the [B,3,H,W] IUV image resampled per joint by
 *                       theta [B,24,2,3]; sel [24][6] int; sample_w [B] (NULL = 1).  backward: d pred for
 *                       scale[0..2] * the three sums (scale on the device).
 */
int danet_part_clean_forward(const void* pred, const float* keep, int B, int H, int W, int cpj, void* x24, void* stream);
int danet_part_clean_backward(const void* g24, const void* pred, const float* keep, int B, int H, int W, int cpj, void* gpred, void* stream);
int danet_part_loss_forward(const void* pred, const float* iuv_img, const float* theta, const float* sample_w,
                            const int* sel, int B, int H, int W, int align, int cpj, double* sums, void* stream);
int danet_part_loss_backward(const void* pred, const float* iuv_img, const float* theta, const float* sample_w,
                             const int* sel, const float* scale, int B, int H, int W, int align, int cpj, void* gpred, void* stream);
/* d pred of BOTH consumers of the prediction in one pass (round 6): danet_part_loss_backward's three terms plus danet_part_clean_backward's
 * (g24 = d x24, keep as there); cpj == 24 only.  What autograd did as two kernels and an add over three 151 MB tensors. */
int danet_part_backward_fused(const void* pred, const float* iuv_img, const float* theta, const float* sample_w,
                              const int* sel, const float* scale, const void* g24, const float* keep,
                              int B, int H, int W, int align, int cpj, void* gpred, void* stream);

/* ---------------------------------------------------------------------------------------
 * Convolution (replaces the cuDNN/ATen kernels behind every nn.Conv2d on the hot path:
 * /root/reference/models/module/hr_module.py:15-378, res_module.py:21-535; shapes SURVEY.md A.2).
 * Activations are NHWC bf16 (torch channels_last), accumulation fp32 on MFMA.
 *
 *  danet_conv_pack_weights  fp32 W[Cout][Cin/groups][R][S] (torch layout) -> packed bf16
 *      mode 0: forward operand      [G][rows_pad][Kp], rows = cout, k = (r*S+s)*Cin_g + cin
 *      mode 1: data-gradient operand [G][rows_pad][Kp], rows = cin,  k = (r*S+s)*Cout_g + cout
 *      (danet_conv_packed_elems gives the element count; rows_pad = roundup(rows, 16*danet_conv_nt(rows)))
 *      chunk must be 0 (reserved).
 *  3x3 / stride-1 / pad-1 layers with groups = 1 and Cin % 16 == 0 (forward and data gradient) run on a persistent
 *      LDS-tile kernel (csrc/conv3x3.hip: halo tile staged once in LDS, taps = LDS address offsets, K-split across
 *      the waves of a workgroup for small-M layers); danet_conv_forward / danet_conv_forward_multi pick it by
 *      themselves, danet_conv_forward_kernel reports it (last digit 2).
 *      addend (optional; bf16 outputs on the 3x3 LDS kernels, or on the lean gather kernel without fused statistics): a bf16 tensor shaped like y that is added before the
 *      result is rounded -- a data-gradient launch thereby accumulates the residual branch's gradient (the `out += residual`
 *      of res_module.py:39-56 in backward) instead of leaving the sum to a separate pass.
 *  danet_conv_forward       y = conv(x, wp) (+bias[Cout])(ReLU); y is bf16 or fp32 NHWC.
 *      transposed = 1 gathers x at (o + pad - r*dil)/stride when divisible: with mode-1 weights
 *      this is the data gradient (x := dY, (H,W) := dY size, Cin := Cout of the layer, (OH,OW),
 *      Cout := size / channels of dX) and also ConvTranspose2d.
 *  danet_conv_pack_job_* / danet_conv_pack_weights_batched   two launches that repack a table of weights
 *      (a training step repacks ~600 of them after every optimizer step): fill a host table with
 *      danet_conv_pack_job_fill (entry i at byte i*danet_conv_pack_job_bytes()), copy it to the device, launch.
 *      A job with danet_conv_pack_job_bricks(...) > 0 belongs to the brick launch (coalesced reads through LDS):
 *      bstart = running sum of the brick counts of the jobs before it, and it does not advance `start`; the
 *      others (chunked K order, channel counts that are not multiples of 8) belong to the per-element launch:
 *      start = running sum of the element counts danet_conv_pack_job_fill returned for the per-element jobs
 *      before it.  total_elems / total_bricks are the two final sums.  The destination buffers must be zeroed
 *      once (the brick launch does not write padding).
 *      bn_sums (optional, danet_bn_ws_floats(Cout) floats = [32][2][Cout] accumulators, see danet_bn_acc_bytes; zeroed by the caller): per-channel sum and sum of squares of the
 *      bf16 output, accumulated by the epilogue; pass it to danet_bn_forward with ws_is_zero = 2 to skip
 *      the separate statistics pass.
 *      bn_x / bn_y / bn_saved / bn_red (optional, data-gradient launches on the fast kernel only): the
 *      BatchNorm that produced this conv's input -- its input bn_x, its output bn_y (NULL = no ReLU), its
 *      saved [mean | invstd] -- for which the epilogue accumulates sum(dy') and sum(dy'*xhat) into bn_red
 *      ([32][2][C] accumulators, zeroed); danet_bn_backward with ws_is_zero = 2 then skips its reduction pass.
 *      bn_gate (LDS-tile 3x3 kernel only; 0 elsewhere) says where that reduction takes the ReLU gate from: 0 = bn_y
 *      as above, 2 = bn_y points at the byte mask danet_bn_forward wrote (relu_mask): one byte per lane instead of eight.
 *  danet_conv_wgrad         dW (fp32, torch layout) = beta*dW + sum_pixels dY (x) X.
 *  Scratch buffers that must start zeroed (BN sums, wgrad accumulator) are cleared by the call unless
 *  ws_is_zero != 0 (the host then zeroes one arena per step instead of ~800 small memsets).
 */
int danet_conv_nt(int rows_per_group);
int danet_conv_kernel_id(int B, int OH, int OW, int Cin, int Cout, int groups);   /* MT*100 + NT*10 + vec8 */
int danet_conv_wgrad_kernel_id(int Cin, int Cout, int groups, int taps);            /* CT*100 + NI*10 + TG */
size_t danet_conv_packed_elems(int Cout_g, int Cin_g, int R, int S, int groups, int mode, int chunk);
int danet_conv_pack_weights(const float* w, void* wp, int Cout, int Cin_g, int R, int S, int groups,
                            int mode, int chunk, void* stream);
/* ... packed straight from a parameter whose per-group extent (src_Cout_g x src_Cin_g) is smaller than the padded widths the kernels run at
 * (Cout / groups x Cin_g, multiples of 8): the missing channels pack as zeros (the reference runs 3 / 12 / 21 / 25 / 15-channel layers as they
 * are, models/module/hr_module.py:447-470, res_module.py:364; here they are zero-padded and the parameters keep the reference's shapes) */
int danet_conv_pack_weights_padded(const float* w, void* wp, int Cout, int Cin_g, int R, int S, int groups,
                                   int mode, int chunk, int src_Cout_g, int src_Cin_g, void* stream);
long danet_conv_pack_job_fill_padded(void* job_host, const float* w, void* wp, long start, long bstart,
                                     int Cout, int Cin_g, int R, int S, int groups, int mode, int chunk, int src_Cout_g, int src_Cin_g);
size_t danet_conv_pack_job_bytes(void);
long danet_conv_pack_job_bricks(int Cout, int Cin_g, int R, int S, int groups, int mode, int chunk);
long danet_conv_pack_job_fill(void* job_host, const float* w, void* wp, long start, long bstart,
                              int Cout, int Cin_g, int R, int S, int groups, int mode, int chunk);
int danet_conv_pack_weights_batched(const void* jobs_dev, int njobs, long total_elems, long total_bricks, void* stream);
/* Up to 12 independent convolutions (forward or data gradient) in one launch (4 on the LDS-tile 3x3 kernel) -- HRNet branches
 * and fuse-layer exchange paths in lockstep.
 * job = { const void* x, *wp; void* y; float* bn_sums; const void* bn_x, *bn_y; const float* bn_saved; float* bn_red; const void* addend;
 *         int B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, bn_gate; }  (no bias / ReLU / fp32 output);
 * all problems must run on the fast kernel with the same danet_conv_nt(Cout/groups): query danet_conv_forward_multi_ok. */
/* fp32 verification convolution (csrc/conv_f32.hip; BASELINE config C4's arithmetic type, slow by design): NHWC fp32
 * tensors, weights in torch's [Cout][Cin/groups][R][S] layout.  mode 0: out = conv(a = x, b = w) + bias; mode 1: out = dX
 * from a = dY, b = w; mode 2: out = dW from a = x, b = dY.  (H, W, Cin) / (OH, OW, Cout) always describe x / y. */
int danet_conv_f32(int mode, const float* a, const float* b, const float* bias, float* out,
                   int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups,
                   void* stream);
/* fp32 convolutions on the matrix cores (csrc/conv_f32m.hip; v_mfma_f32_16x16x4_f32, exact fp32 fmaf chains): the reference's
 * own arithmetic type (nn.Conv2d in fp32, /root/reference/models/module/hr_module.py:188-378, res_module.py:27-97) as a
 * performance path.  fp32 NHWC activations; weights repacked by danet_conv_f32m_pack_weights (mode 0 forward operand, mode 1
 * data-gradient operand) with the channel counts the kernel runs with (Cout_gp >= Cout/groups, Cin_gp >= Cin_g, both
 * multiples of 4; padding packs as zeros; groups > 1 allows no padding).  danet_conv_f32m_ok: 1 when _forward takes the
 * problem.  _forward with transposed = 1 is the data gradient: (H, W, Cin) describe the tensor gathered FROM (dY).
 * _wgrad: dW[Cout_real][Cin_g_real][R][S] in torch's layout; per-pixel-chunk partial blocks go to ws (_wgrad_ws_floats floats, 0 =
 * unsupported) and are summed in a fixed order (deterministic, no atomics). */
size_t danet_conv_f32m_packed_elems(int Cout_gp, int Cin_gp, int R, int S, int groups, int mode);
int danet_conv_f32m_pack_weights(const float* w, float* wp, int Cout, int Cin_g, int R, int S, int groups, int mode,
                                 int Cout_gp, int Cin_gp, void* stream);
int danet_conv_f32m_ok(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil,
                       int groups, int transposed);
int danet_conv_f32m_forward(const float* x, const float* wp, const float* bias, float* y,
                            int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil,
                            int groups, int transposed, int relu, void* stream);
size_t danet_conv_f32m_wgrad_ws_floats(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad,
                                       int dil, int groups, int Cout_real, int Cin_g_real);
int danet_conv_f32m_wgrad(const float* x, const float* dy, float* dw, float* ws,
                          int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups,
                          int Cout_real, int Cin_g_real, void* stream);
int danet_conv_forward_multi_ok(const void* jobs, int n);          /* 0 no, 1 conv_fast_multi_kernel, 2 conv3x3_tile_kernel */
/* The convolutions of danet_conv_forward_multi AND the training-mode BatchNorm (+ residual) (+ ReLU) that follows each of them
 * (/root/reference/models/module/res_module.py:39-56 conv -> bn -> relu / conv -> bn -> + residual -> relu; hr_module.py:155-177):
 * bn_jobs[i] is a forward job of danet_bn_forward_multi whose x is conv job i's y, whose sums is its bn_sums (sums_state 2) and
 * whose M, C are that output's.  When the streamed 3x3 kernel takes the set (<= 4 plain forward problems) it is ONE launch: the
 * workgroups cross a grid-wide barrier after their last tile and normalise the tiles they wrote themselves (conv3x3s.hip
 * s3_bn_tail) -- results (y, saved, running statistics, mask) bit-identical to the two launches, which is what the call runs
 * otherwise.  *fused (optional) says which.  bar: the grid-barrier state of danet_bn_backward_onepass (same contract: zeroed once,
 * launches that use it must not overlap, the device otherwise idle so that all workgroups are resident; a barrier that times out
 * sets the same error word).  NULL bar: always two launches.  _ok: 1 when the set would be one launch. */
int danet_conv_bn_forward_multi_ok(const void* jobs, int n, const void* bn_jobs, void* bar);
int danet_conv_bn_forward_multi(const void* jobs, int n, const void* bn_jobs, float momentum, float eps, void* bar, int* fused, void* stream);
int danet_conv_forward_multi_kernel(const void* jobs, int n);      /* the kernel the set runs on: 0 none, 1 conv_fast_multi_kernel, 2 conv3x3_tile_kernel,
                                                                       3 conv3x3_stream_kernel (csrc/conv3x3s.hip) */
/* ---------------------------------------------------------------------------------------
 * The library's ONE run-time switch board, for A-B timing and tests (production code never calls it; the defaults come from
 * the environment variables the kernels' files name).  danet_knob(id, value): value < 0 only queries; returns the previous value
 * (-1: unknown id).  The switches are process-wide -- callers that flip one restore it (the Python host side offers a context
 * manager that cannot leak a flipped knob: _lib.knobs(...)).
 *   C3_*   the LDS-tile 3x3 kernel (csrc/conv3x3.hip): ENABLE 0/1; MT, KW = register tiling forced on every problem (0, 0 = the
 *          planner's choice); BLOCKS = workgroup cap (> 0); WANT = tiles per problem the planner aims for (0 = 512 / problems)
 *   C3S_*  the streamed 3x3 kernel (csrc/conv3x3s.hip), which takes the 3x3 / stride-1 problems with at most 48 output channels
 *          per block ahead of the tile kernel: ENABLE; BLOCKS; KW = forced K split over the four waves 1/2/4 (0 = planner);
 *          WANT as above; BALANCE 0/1 = workgroups dealt to the problems of a launch in proportion to their work (1, the default) or
 *          the launch's concatenated tile list dealt over all workgroups (0, rounds 2-4); TILE_COST = the fixed cost of a tile in that
 *          model, in k-steps.  A forced register tiling (C3_MT / C3_KW) keeps a problem on conv3x3.hip.
 *   PW, PW_WGRAD, STEM, STEM_DGRAD, C3A   0/1: the pointwise forward / data-gradient kernel, the pointwise weight gradient, the
 *          7x7 stem forward and data gradient on LDS row tiles, the 64-channel row-tile 3x3 kernel (their sections below)
 *   BN_BLOCK_BYTES   bytes of the tensor one workgroup of the BatchNorm kernels handles at least (default 24576; > 0 sets) */
enum { DANET_KNOB_C3_ENABLE = 1, DANET_KNOB_C3_MT = 2, DANET_KNOB_C3_KW = 3, DANET_KNOB_C3_BLOCKS = 4, DANET_KNOB_C3_WANT = 5,
       DANET_KNOB_C3S_ENABLE = 6, DANET_KNOB_C3S_BLOCKS = 7, DANET_KNOB_C3S_KW = 8, DANET_KNOB_C3S_WANT = 9,
       DANET_KNOB_PW = 10, DANET_KNOB_PW_WGRAD = 11, DANET_KNOB_STEM = 12, DANET_KNOB_STEM_DGRAD = 13, DANET_KNOB_C3A = 14,
       DANET_KNOB_BN_BLOCK_BYTES = 15, DANET_KNOB_C3S_BALANCE = 16, DANET_KNOB_C3S_TILE_COST = 17,
       DANET_KNOB_G3 = 18 /* csrc/conv_g3.hip: narrow-group 3x3 layers (danet_conv_forward_kernel: last digit 5) */ };
long danet_knob(int id, long value);
/* danet_conv3x3_stream_plan: KW*100 + stages*10 + NT the streamed 3x3 kernel (csrc/conv3x3s.hip) would use for a problem in a launch of
 * nprob problems (0: not taken). */
int danet_conv3x3_stream_plan(int B, int H, int W, int Cin, int Cout, int nprob);
/* The streamed kernel reads a per-shape tap table (<= 112 k-step entries, danet_conv3x3_stream_table_bytes() bytes each) from
 * device memory.  The library allocates none: the caller registers a workspace for the CURRENT device once (it must outlive
 * every launch; registering again, or NULL, forgets the cached tables) and gets back the number of tables it holds.  A
 * table is computed on the host and uploaded with a synchronous copy the first time its shape is seen -- complete before any
 * stream can launch a reader; a new shape first seen while the launching stream is capturing, or with no workspace / a full
 * one, runs on conv3x3_tile_kernel instead (danet_conv_forward_multi_kernel / danet_conv_forward_kernel say which). */
size_t danet_conv3x3_stream_table_bytes(void);
int danet_conv3x3_stream_tables(void* workspace, size_t bytes);
/* The pointwise kernel (csrc/conv_pw.hip: 1x1 / stride-1 layers with <= 64 KB of packed weights and >= 8192 pixels; the layer's
 * weights in LDS, persistent workgroups, X read once and Y written once) takes such problems ahead of the gather kernel in
 * danet_conv_forward (danet_conv_forward_kernel: last digit 3).  Switch: DANET_KNOB_PW. */
/* The regressor ResNets' 7x7 / stride-2 / pad-3 stems (/root/reference/models/module/res_module.py:404, :118) on LDS tiles (csrc/conv_stem.hip):
 * 64 output channels, input channels a multiple of 16, 2 OH x 64 -> OH x 32 maps with OH % 8 == 0, at least 256 tiles (8 output rows of an
 * image each).  danet_conv_stem_ok: 1 when the kernel takes the problem.  danet_conv_stem_forward: x [B,H,W,Cin] bf16 NHWC, wp = the weight
 * packed by danet_conv_pack_weights(mode 0, chunk 16) -- K order (16-channel slab, tap, channel) --, y [B,OH,OW,64] bf16; bn_sums: optional
 * fused BatchNorm statistics [BN_NCOPY][2][64], pre-zeroed.  The data gradient of these layers stays on danet_conv_forward (transposed).
 * Switch: DANET_KNOB_STEM. */
int danet_conv_stem_ok(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups);
int danet_conv_stem_forward(const void* x, const void* wp, void* y, int B, int H, int W, int Cin, int OH, int OW, int Cout,
                            float* bn_sums, void* stream);
/* Data gradient of a stem convolution with 64 input and 64 output channels (csrc/conv_stem_dgrad.hip): (B, H, W, Cin) describe dx (the
 * convolution's input), (OH, OW, Cout) dy; weights = danet_conv_pack_weights(mode 1, chunk 16).  bn_x / bn_y / bn_saved / bn_red: the fused
 * BatchNorm-backward sums of danet_conv_forward's arguments of the same names (all NULL: none).  Replaces danet_conv_forward(transposed = 1)
 * for /root/reference/models/module/res_module.py:404 (SmplResNet.conv1) in the backward pass of models/danet/smpl_regressor.py. */
int danet_conv_stem_dgrad_ok(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups);
int danet_conv_stem_dgrad(const void* dy, const void* wp, void* dx, int B, int H, int W, int Cin, int OH, int OW, int Cout,
                          const void* bn_x, const void* bn_y, const float* bn_saved, float* bn_red, void* stream);
/* 3x3 / stride 1 / pad 1, 64 -> 64 channels on 16- or 64-wide maps (csrc/conv3x3a.hip: the BasicBlocks of the regressor ResNets' layer1,
 * /root/reference/models/module/res_module.py:27-56 under SmplResNet :404): forward (transposed = 0, weights mode 0 / chunk 16, optional output
 * statistics bn_sums) and data gradient (transposed = 1, weights mode 1 / chunk 16, optional fused BatchNorm-backward sums -- bn_gate 0: bn_y is
 * the BatchNorm's bf16 output, 2: its byte mask -- and residual addend), the contracts of danet_conv_forward's arguments of the same names. */
int danet_conv3x3a_ok(int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil, int groups);
int danet_conv3x3a(const void* x, const void* wp, void* y, int B, int H, int W, int transposed, float* bn_sums,
                   const void* bn_x, const void* bn_y, const float* bn_saved, float* bn_red, int bn_gate, const void* addend, void* stream);
/* Profiling hook: device buffer of blocks*8 ints receiving each workgroup's phase timestamps (s_memtime; NULL = off). */
void danet_conv3x3_debug(int* dev_buf);
int danet_conv_forward_multi(const void* jobs, int n, void* stream);
int danet_conv_forward_kernel(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S,
                              int stride, int pad, int dil, int groups, int transposed, int out_fp32);
                              /* MT*1000 + NT*100 + vec8*10 + fast: conv_fast_kernel<MT,NT> or conv_igemm_kernel<MT,NT,vec8>;
                                 MT*1000 + NT*100 + KW*10 + 2: conv3x3_tile_kernel with that register tiling */
int danet_conv_forward(const void* x, const void* wp, const float* bias, void* y,
                       int B, int H, int W, int Cin, int OH, int OW, int Cout,
                       int R, int S, int stride, int pad, int dil, int groups, int transposed,
                       int relu, int out_fp32, float* bn_sums,
                       const void* bn_x, const void* bn_y, const float* bn_saved, float* bn_red, const void* addend,
                       int bn_gate, void* stream);
/* 3x3 / stride 1 or 2 / pad 1 weight gradient through the LDS transpose read (conv_wgrad3x3.hip); use when
 * danet_conv_wgrad3x3_ok(...) != 0, with danet_conv_wgrad3x3_ws_floats(...) floats of scratch.  (H, W) is the INPUT
 * size; dy is [B, H/stride, W/stride, Cout]. */
int danet_conv_wgrad3x3_ok(int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil, int groups);
/* 4 x 4 maps (the regressor tails, reference models/module/res_module.py:393-464 at 768 part crops): two images share one 4 x 8 chunk;
 * such problems are accepted by danet_conv_wgrad3x3_multi only (B even). */
int danet_conv_wgrad3x3_pair_ok(int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil, int groups);
size_t danet_conv_wgrad3x3_ws_floats(int B, int H, int W, int Cin, int Cout, int groups, int stride);
/* Batched 3x3 weight gradients: weight gradients are only needed by the optimizer, so a trainer may queue them during the
 * backward pass and compute them with a few multi-problem launches (up to 20 problems per launch, grouped by kernel instance).
 * jobs: array of n { const void* x; const void* dy; float* dw; int B, H, W, Cin, Cout, groups, stride; } (host memory). */
size_t danet_conv_wgrad3x3_multi_ws_floats(const void* jobs, int n);
/* the same for the general weight-gradient kernel; jobs: { const void* x; const void* dy; float* dw;
 * int B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups; }; ws must be ZEROED by the caller from float
 * danet_conv_wgrad_multi_ws_zero_from(jobs, n) on (packed accumulators -- doubles, see danet_bn_acc_bytes; the leading part holds
 * the pointwise kernel's partial sums, which need no zeroing). */
size_t danet_conv_wgrad_multi_ws_floats(const void* jobs, int n);
size_t danet_conv_wgrad_multi_ws_zero_from(const void* jobs, int n);
int danet_conv_wgrad_multi(const void* jobs, int n, float* ws, size_t ws_floats, float beta, void* stream);
int danet_conv_wgrad3x3_multi(const void* jobs, int n, float* ws, size_t ws_floats, float beta, void* stream);
int danet_conv_wgrad3x3_kernel_id(int B, int H, int W, int Cin, int Cout, int groups, int stride);   /* CT*10 + NI */
int danet_conv_wgrad3x3(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats,
                        int B, int H, int W, int Cin, int Cout, int groups, int stride, float beta, int phase /* 0 both kernels, 1 MFMA kernel only, 2 reduction only */, void* stream);
/* 7x7 / stride 2 / pad 3 weight gradient of large batches (the regressor ResNets' stems, res_module.py:407), one
 * filter row per workgroup through the LDS transpose read (conv_wgrad_rows.hip); use when danet_conv_wgrad_rows_ok != 0. */
int danet_conv_wgrad_rows_ok(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups);
size_t danet_conv_wgrad_rows_ws_floats(int B, int OH, int OW, int Cin, int Cout, int R, int S, int groups);
int danet_conv_wgrad_rows(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats,
                          int B, int H, int W, int Cin, int OH, int OW, int Cout,
                          int R, int S, int stride, int pad, int groups, float beta, void* stream);
size_t danet_conv_wgrad_ws_floats(int Cout, int Cin_g, int R, int S);
/* The workspace danet_conv_wgrad needs for THIS problem: as above, or more for 1x1 / stride-1 layers, which run on the pointwise
 * weight-gradient kernel (csrc/conv_pw_wgrad.hip: 32-pixel chunks of dY and X staged as they lie in memory, LDS transpose reads,
 * per-workgroup partial sums reduced in a fixed order -- deterministic, no atomics) when the workspace has room for its partial
 * sums; with the smaller workspace the generic kernel runs.  danet_conv_wgrad_multi sizes its own (danet_conv_wgrad_multi_ws_floats).
 * Switch: DANET_KNOB_PW_WGRAD. */
size_t danet_conv_wgrad_ws_floats_for(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups);
int danet_conv_wgrad(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats,
                     int B, int H, int W, int Cin, int OH, int OW, int Cout,
                     int R, int S, int stride, int pad, int dil, int groups, float beta, int ws_is_zero, void* stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm2d (+ReLU, +residual add) and the HRNet fuse, NHWC bf16 viewed as [M = B*H*W, C].
 * Replaces nn.BatchNorm2d / ReLU / `out += residual` / nn.Upsample(nearest) + sum
 * (/root/reference/models/module/res_module.py:39-56,77-97; hr_module.py:111-177).
 *
 *  danet_bn_forward   training: batch statistics (biased variance for normalisation, unbiased for
 *                     the running estimate, torch semantics); y = [relu](bn(x) [+ res]);
 *                     saved [2][C] (mean, invstd) is an fp32 output, sums_ws is scratch of
 *                     danet_bn_ws_floats(C) floats.  eval: running statistics.  gamma/beta may be NULL.
 *                     relu_mask (optional, M*C/4 bytes): receives the ReLU gate of y, bit j of byte i = channel
 *                     4i+j positive, so that the backward need not read y again.
 *  danet_bn_backward  dx, optional dres (= masked dy), dparam [2][C] <- (d beta, d gamma);
 *                     red_ws: danet_bn_ws_floats(C) floats of scratch.  The ReLU gate comes from
 *                     mask_mode 0: the saved output y;  1: relu_mask written by the forward;  2: recomputed from x,
 *                     gamma, beta and saved (only without a residual; y and relu_mask may then be NULL).
 *  danet_sum_relu_*   y = [relu](sum_t nearest_upsample_{2^shift_t}(term_t)), up to 4 terms; `terms`
 *                     and `shifts` are HOST arrays of nterms entries.  The backward of one term is
 *                     the window sum of gy * (y > 0).
 */
int danet_bn_forward(const void* x, const void* res, void* y, int64_t M, int C,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float* saved, float* sums_ws, int ws_is_zero, float momentum, float eps, int training, int relu,
                     void* relu_mask, void* stream);
size_t danet_bn_ws_floats(int C);
/* The statistics workspaces (sums_ws, red_ws, bn_sums, bn_red: danet_bn_ws_floats(C) floats each, 8-byte aligned) hold
 * [32][2][C] accumulators of danet_bn_acc_bytes() bytes: 8 = double (the default), 4 = float (library built with
 * -DDANET_BN_ACC32, the round-1..4 form, for A-B timing).  A workgroup adds its fp32 partial sums into replica
 * (workgroup id % replicas) with one device-scope atomic per channel; in double precision every such addition is exact
 * (partials of a channel within 2^25 of each other), so the statistics -- and the whole training step -- do not depend on
 * the order in which workgroups arrive: two executions of the same step agree bit for bit (tests/test_gpu_zz_paths.py). */
int danet_bn_acc_bytes(void);
int danet_bn_backward(const void* dy, const void* x, const void* y, int64_t M, int C,
                      const float* gamma, const float* saved, int relu,
                      void* dx, void* dres, float* dparam, float* red_ws, int ws_is_zero,
                      int mask_mode, const void* relu_mask, const float* beta, void* stream);
/* Up to 12 independent training-mode BatchNorms per launch (HRNet branches / exchange paths in lockstep); C <= 1024 each.
 *  forward job  { const void* x, *res; void* y; const float* gamma, *beta; float* running_mean, *running_var, *saved, *sums;
 *                 void* mask; int64_t M; int C, sums_state, relu; }   sums_state 1: zeroed scratch, 2: accumulated by the conv epilogue
 *  backward job { const void* dy, *x, *y; const float* gamma, *saved; void* dx, *dres; float* dparam, *red; const float* beta;
 *                 const void* mask; int64_t M; int C, red_state, relu, mask_mode; }
 *                 red_state 1: zeroed scratch, 2: accumulated by the dgrad epilogue; mask / mask_mode as above */
/* (DANET_KNOB_BN_BLOCK_BYTES: bytes of the tensor one workgroup of these kernels handles at least -- default 24576, or
 * DANET_BN_BLOCK_BYTES; above every tensor's size each launch is a single workgroup per tensor.) */
int danet_bn_forward_multi(const void* jobs, int n, float momentum, float eps, void* stream);
int danet_bn_backward_multi(const void* jobs, int n, void* stream);
/* One-pass form of danet_bn_backward_multi: every lane keeps its share of dy / x in registers across a grid-wide barrier,
 * so dy and x are read once instead of twice.  Qualifying sets (danet_bn_backward_onepass_ok): every job with red_state 1,
 * a ReLU gate that does not need y (mask_mode 1 or 2), C <= 1024 and at most 512 x 256 x 16 channel vectors.  bar:
 * danet_bn_backward_onepass_bar_words() uints of device memory, zeroed once, shared by all such launches, which must not overlap
 * (issue them on one stream).  The barrier (two levels: eight arrival groups, then one) spin is bounded: after a timeout
 * bar[2] != 0, the results of that launch are garbage and the state must be zeroed again before another launch (pass bar + 2
 * to danet_adam_step as `poison`: such a step is then never applied).
 * max_blocks (<= 0: two workgroups per compute unit of the device, at most 512): the caller's co-residency budget.  The
 * barrier needs every workgroup of a launch resident at once; a kernel of ANOTHER stream that occupies compute units while
 * these launches run (the communication library's all-reduce kernels of a data-parallel step: one workgroup per channel)
 * can keep the last workgroups out until it finishes -- and if that kernel is itself only partly resident, for ever.  With
 * max_blocks = 2 * (compute units - channels) every launch fits beside the other kernel's worst case, so neither can wait
 * for the other; job sets are split into more launches, a single job that needs more takes the two-kernel path. */
int danet_bn_backward_onepass_bar_words(void);
int danet_bn_backward_onepass_ok(const void* jobs, int n, int max_blocks);
int danet_bn_backward_onepass(const void* jobs, int n, void* bar, int max_blocks, void* stream);
/* out[C] (DOUBLES: order-independent accumulation) = sum over the M rows of x [M, C] (bf16; _f32: fp32): the bias gradient of a convolution,
 * gy.sum(dim = (0, 2, 3)) in /root/reference's autograd.  out is zeroed here unless out_is_zero != 0 (a slice of the caller's
 * zeroed arena); C % 4 == 0. */
/* (ncopy: out is [ncopy][C] replicas, workgroup b adds into replica b % ncopy, the caller sums them; 1 = a plain [C] result) */
int danet_channel_sum(const void* x, int64_t M, int C, double* out, int out_is_zero, int ncopy, void* stream);
int danet_channel_sum_f32(const void* x, int64_t M, int C, double* out, int out_is_zero, int ncopy, void* stream);
int danet_sum_relu_forward(const void* const* terms, const int* shifts, int nterms,
                           int B, int H, int W, int C, int relu, void* y, void* stream);
int danet_sum_relu_backward(const void* gy, const void* y, int B, int H, int W, int C, int shift, int relu,
                            void* d_term, void* stream);
/* every requested shift in one launch: d0..d3 (NULL = not needed) <- [B, H >> s, W >> s, C]; gy and y are read once */
/* Multi-problem forms (host job arrays, n <= 4): the fuse sums of ONE HighResolutionModule, or their gradients, in one launch.
 *  forward job  { const void* terms[4]; int shifts[4]; int nterms, B, H, W, C, relu; void* y; }
 *  backward job { const void* gy; const void* y; int B, H, W, C, relu; void* d[4]; }   d[s]: gradient of the terms with shift s, NULL = none */
int danet_sum_relu_forward_multi(const void* jobs, int n, void* stream);
int danet_sum_relu_backward_all_multi(const void* jobs, int n, void* stream);
int danet_sum_relu_backward_all(const void* gy, const void* y, int B, int H, int W, int C, int relu,
                                void* d0, void* d1, void* d2, void* d3, void* stream);
/* fp32 instantiation of the entry points above (csrc/norm_act_f32.hip: the same kernels with 4-byte elements; BASELINE
 * config C4's arithmetic type -- the reference's nn.BatchNorm2d / ReLU / add / nn.Upsample in fp32,
 * /root/reference/models/module/hr_module.py:111-177): activations fp32 NHWC, everything else as in the bf16 forms
 * (masks: one byte per 4 channels; job structs identical).  The one-pass backward has no fp32 form. */
int danet_bn_forward_f32(const void* x, const void* res, void* y, int64_t M, int C,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float* saved, float* sums_ws, int ws_is_zero, float momentum, float eps, int training, int relu,
                         void* relu_mask, void* stream);
int danet_bn_backward_f32(const void* dy, const void* x, const void* y, int64_t M, int C,
                          const float* gamma, const float* saved, int relu,
                          void* dx, void* dres, float* dparam, float* red_ws, int ws_is_zero,
                          int mask_mode, const void* relu_mask, const float* beta, void* stream);
int danet_bn_forward_multi_f32(const void* jobs, int n, float momentum, float eps, void* stream);
int danet_bn_backward_multi_f32(const void* jobs, int n, void* stream);
int danet_sum_relu_forward_f32(const void* const* terms, const int* shifts, int nterms,
                               int B, int H, int W, int C, int relu, void* y, void* stream);
int danet_sum_relu_backward_f32(const void* gy, const void* y, int B, int H, int W, int C, int shift, int relu,
                                void* d_term, void* stream);
int danet_sum_relu_forward_multi_f32(const void* jobs, int n, void* stream);
int danet_sum_relu_backward_all_multi_f32(const void* jobs, int n, void* stream);
int danet_sum_relu_backward_all_f32(const void* gy, const void* y, int B, int H, int W, int C, int relu,
                                    void* d0, void* d1, void* d2, void* d3, void* stream);

/* ---------------------------------------------------------------------------------------
 * Joint-centric part decomposition (STN).  Replaces the 24 x (F.affine_grid + F.grid_sample) +
 * torch.cat of /root/reference/models/danet/iuv_estimator.py:193-204.
 *  x [B,H,W,C] bf16 NHWC (C % 8 == 0), theta [B,P,2,3] f32 -> y [B,OH,OW,P*C] bf16 (part-major
 *  channels, i.e. torch.cat(dim=1) order).  Bilinear, zero padding, align_corners as given.
 *  The backward (w.r.t. x only: the reference detaches theta) requires axis-aligned thetas
 *  ([[sx,0,cx],[0,sy,cy]], which is all affine_para ever builds).
 */
int danet_stn_gather_forward(const void* x, const float* theta, int B, int H, int W, int C, int P,
                             int OH, int OW, int align_corners, void* y, void* stream);
int danet_stn_gather_backward(const void* dy, const float* theta, int B, int H, int W, int C, int P,
                              int OH, int OW, int align_corners, void* dx, void* stream);
/* the same with fp32 NHWC tensors (conv.precision('fp32')) */
int danet_stn_gather_forward_f32(const void* x, const float* theta, int B, int H, int W, int C, int P,
                                 int OH, int OW, int align_corners, void* y, void* stream);
int danet_stn_gather_backward_f32(const void* dy, const float* theta, int B, int H, int W, int C, int P,
                                  int OH, int OW, int align_corners, void* dx, void* stream);

/* ---------------------------------------------------------------------------------------
 * 3x3 / stride-2 / pad-1 max pooling of the regressor stems (/root/reference/models/module/res_module.py:303) on NHWC tensors.
 *  forward: x [B,H,W,C] -> y [B,OH,OW,C], OH = (H - 1) / 2 + 1, and idx (B*OH*OW*C bytes): the position 0..8 of each maximum
 *  in its window (first maximum in row-major scan order, NaN wins: torch's rule); backward: dx as a gather over the <= 4 windows
 *  of an input pixel (deterministic).  bf16 (C % 8 == 0) and fp32 (_f32, C % 4 == 0). */
int danet_maxpool3x3s2_forward(const void* x, void* y, void* idx, int B, int H, int W, int C, void* stream);
int danet_maxpool3x3s2_backward(const void* gy, const void* idx, void* dx, int B, int H, int W, int C, void* stream);
int danet_maxpool3x3s2_forward_f32(const void* x, void* y, void* idx, int B, int H, int W, int C, void* stream);
int danet_maxpool3x3s2_backward_f32(const void* gy, const void* idx, void* dx, int B, int H, int W, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DANET_HIP_H */
