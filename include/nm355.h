/* nm355 — C ABI of the MI355X-native Neural Marionette hot path (libnm355.so).
 *
 * The reference (jinseokbae/neural_marionette) is pure PyTorch: it has no FFI / plugin
 * boundary of its own.  The boundary it exposes for this path is the nn.Module surface
 * of NeuralMarionette / KyptDetector / HSVRNNBVH; this header is the C interface that
 * the drop-in Python shells (neural_marionette_amd/modules.py) bind with ctypes, one
 * entry point per reference method.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous fp32 (or int32 where stated) memory
 *    owned by the caller, in exactly the layout of the reference tensor named in the
 *    comment; the library never keeps a caller pointer past the call's stream-ordered
 *    completion (weights are copied/re-packed into ctx-owned memory by
 *    nm_ctx_set_weights).
 *  - calls are asynchronous on the ctx stream (nm_ctx_set_stream); no hidden device
 *    synchronisation except in create / destroy / set_weights / workspace growth.
 *  - return value: 0 = OK, <0 = error (NM_ERR_*); message via nm_last_error()
 *    (thread-local).  No exceptions or abort() cross this boundary.
 *  - a ctx is bound to one device and one stream and is not thread-safe.
 */
#ifndef NM355_H
#define NM355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NM_ABI_VERSION 1

#define NM_OK 0
#define NM_ERR_ARG (-1)
#define NM_ERR_HIP (-2)
#define NM_ERR_STATE (-3)
#define NM_ERR_UNSUPPORTED (-4)
#define NM_ERR_RANGE (-5)       /* non-finite conv results (operand beyond the split-fp16 range, or non-finite input) */
#define NM_ERR_INTERNAL (-6)    /* a C++ exception (std::bad_alloc, ...) was caught at the ABI: every entry point is a function-try-block, nothing unwinds into the caller */

typedef struct nm_ctx nm_ctx;

/* Hot-path hyper-parameters (pretrained/aist/opt.pickle of the reference; fields read at
 * model/kypt_detector.py:18-68, model/hsvrnn_bvh.py:14-20). */
typedef struct nm_config {
    int32_t device;          /* HIP device ordinal */
    int32_t grid_size;       /* G: occupancy grid edge (64; multiples of 8 >= 32) */
    int32_t nkeypoints;      /* K (24) */
    int32_t nlatent;         /* Z (128) */
    int32_t nhidden;         /* H (512) */
    int32_t nneighbor;       /* N (2) affinity neighbours */
    float gaussian_sigma;    /* 1.5 */
    float sep_sigma;         /* 0.02 */
    int32_t vol_fit_chamfer; /* vol_fit_type: 1 'chamfer', 0 'none', 2 'gaussian' (kypt_detector_utils.py:154-169, as the reference computes it) */
    int32_t use_graph_traj;  /* graph_traj_weight > 0 */
} nm_config;

typedef struct nm_named_tensor {
    const char* name;        /* state_dict key of the reference module */
    const float* data;       /* device pointer, contiguous fp32, torch layout */
    int64_t numel;
} nm_named_tensor;

int nm_abi_version(void);
const char* nm_last_error(void);
/* Errors: every entry point below is a function-try-block - a C++ exception raised inside the library (std::bad_alloc
 * from a host container, ...) is caught at the boundary and reported as NM_ERR_INTERNAL with its message in nm_last_error(); nothing
 * unwinds into the caller (ctypes / cgo / JNI would abort).  nm_abi_selftest_throw raises such an exception on purpose (kind 0:
 * std::bad_alloc of a real allocation request, 1: std::length_error, 2: a non-std exception) and must return NM_ERR_INTERNAL; it needs
 * no device (tests/test_abi_cpu.py). */
int nm_abi_selftest_throw(int32_t kind);

int nm_ctx_create(nm_ctx** out, const nm_config* cfg);
int nm_ctx_destroy(nm_ctx* ctx);
/* Binds the stream every later call is enqueued on (torch.cuda.current_stream() in the shells).  A ctx works on ONE stream
 * at a time; when the handle changes, the new stream is made to wait (event) for everything the ctx queued on the old one
 * and on its own side stream, so ctx-owned weights / workspaces are never overwritten under a running kernel. */
int nm_ctx_set_stream(nm_ctx* ctx, void* hip_stream);
/* Range / finiteness status.  The default conv arithmetic splits every fp32 operand into two fp16 halves (fp32-equivalent
 * products); an activation of magnitude >= 65520 has no fp16 representation and turns the product into inf / NaN where the
 * reference's fp32 arithmetic (torch CPU ops under kypt_detector.py:81-169) stays finite.  The reference's networks never come
 * near that range (every conv input is a GroupNorm output, an occupancy or a Gaussian map), so the guard is a status word, not a
 * per-launch test: the GroupNorm finalisation of every conv ORs 1 into a ctx-owned device word when the conv's statistics are not
 * finite.  The word is READ AUTOMATICALLY (round 4): every forward-type entry point (nm_detector_forward[_train], nm_forward_fused,
 * nm_vrnn_encode / generate / rollout ...) ends with a 4-byte copy of it into a pinned host slot behind an event, and every entry
 * point begins by looking at the slots whose events have completed - no device synchronisation; a set bit makes THAT call fail
 * with a message naming the call that produced the value and the remedy (nm_set_conv_mode(ctx, 0): exact fp32 MFMA, no range
 * limit): NM_ERR_RANGE for bit 1 (non-finite conv statistics), NM_ERR_STATE for bit 2 (a persistent rollout kernel gave up a bounded
 * wait - never a hang).  nm_ctx_check_nonfinite drains the pending slots synchronously with the same two codes, word cleared; 0
 * otherwise.  NM355_RANGE_CHECK=0 removes the copies.  The op-level entry point nm_op_conv3d is synchronous about it: it
 * scans its result and re-runs the launch on the fp32 path by itself.  (The Python shells add set_conv_mode('auto').) */
int nm_ctx_check_nonfinite(nm_ctx* ctx);
/* Replaces NeuralMarionette.load_state_dict / .cuda() for the HIP path: copies every
 * tensor and re-packs conv weights into the MFMA layout.  Must be called again after an
 * optimizer step.  All 337 keys of the reference state_dict are required.  Asynchronous on the ctx
 * stream: the source tensors are read in stream order (keep them alive and unmodified until work enqueued
 * before the call's return has drained - PyTorch's stream-ordered allocator guarantees that for tensors of
 * the same stream); a repeated call reuses the ctx-owned buffers and does not synchronise the host. */
int nm_ctx_set_weights(nm_ctx* ctx, const nm_named_tensor* tensors, int32_t count);
/* Bytes of ctx-owned workspace a (B, T) call needs (allocated lazily, grown on demand). */
size_t nm_workspace_bytes(nm_ctx* ctx, int32_t B, int32_t T);
/* Device memory the context owns right now, in bytes: out[0] inference workspace, out[1] training arena (every layer output of the
 * last nm_detector_forward_train + the backward's transients), out[2] the block behind the weight-gradient stream (dY ring, upsample /
 * slot scratch, scale pool), out[3] weights and their packs. */
int nm_ctx_memory(nm_ctx* ctx, size_t out[4]);

/* KyptDetector.forward — model/kypt_detector.py:81-169.
 *  vox        (B,T,1,G,G,G)            in
 *  keypoints  (B,T,K,4)                out  [x1,x2,x3,intensity]
 *  heatmaps   (B,T,K,g,g,g) g=G/4      out
 *  first_feature (B,128,g,g,g)         out
 *  recon      (B,T,1,G,G,G)            out
 *  affinity   (N,K,K,1) or NULL        out  (get_affinity, :171-211, ver 3)
 *  losses11   11 floats                out  order: recon_loss, vol_fit_reg, kypt_const_loss,
 *             separation_loss, sparsity_loss, local_const_loss, time_const_loss,
 *             sparsity_const_loss, intensity_const_loss, graph_traj_loss, graph_vol_loss
 *  affinity_on mirrors KyptDetector.affinity_start (anneal(), :71-78). */
int nm_detector_forward(nm_ctx* ctx, const float* vox, int32_t B, int32_t T, int32_t affinity_on,
                        float* keypoints, float* heatmaps, float* first_feature, float* recon,
                        float* affinity, float* losses11);

/* The part of KyptDetector.forward the learner regime's loss reads (round 5): VoxToKyptNet (model/kypt_detector.py:299-364) and the
 * affinity (:171-211) - keypoints, heat-maps, first-frame feature - WITHOUT KyptToVoxNet (:388-460) and the eleven losses.  The
 * reference's learner-mode step (train.py:376-412 with pretrained_mode 1; model/neural_marionette.py:45-47) runs the whole detector
 * under torch.no_grad() and reads only log['keypoints'] / the affinity from it; the outputs written here are bit-identical to
 * nm_detector_forward's.  Same argument meaning as nm_detector_forward. */
int nm_detector_keypoints(nm_ctx* ctx, const float* vox, int32_t B, int32_t T, int32_t affinity_on,
                          float* keypoints, float* heatmaps, float* first_feature, float* affinity);

/* NeuralMarionette.forward with detector + learner active (model/neural_marionette.py:34-56) in one call:
 * nm_detector_forward followed by nm_vrnn_encode on the detected keypoints, with the VRNN issued on a
 * ctx-owned side stream as soon as the keypoints exist so that it runs beside the decoder (it does not
 * depend on it).  Needs nm_vrnn_set_tree.  Arguments as in the two separate calls. */
int nm_forward_fused(nm_ctx* ctx, const float* vox, int32_t B, int32_t T, int32_t affinity_on, const float* eps,
                     int32_t S, float* keypoints, float* heatmaps, float* first_feature, float* recon,
                     float* affinity, float* losses11, float* kypt_recon, float* R, float* z, float* h,
                     float* scalars2, int32_t* best_idx);

/* KyptDetector.decode_from_dyna — model/kypt_detector.py:213-241.
 *  keypoints (B,Tg,K,4), first_feature (B,128,g,g,g), first_frame (B,1,G,G,G) -> gen (B,Tg,1,G,G,G) */
int nm_decode_from_keypoints(nm_ctx* ctx, const float* keypoints, const float* first_feature,
                             const float* first_frame, int32_t B, int32_t Tg, float* gen);

/* KyptDetector.get_affinity — model/kypt_detector.py:171-210 -> (N,K,K,1); version 3 unless nm_ctx_set_affinity_ver chose another */
int nm_get_affinity(nm_ctx* ctx, float* affinity);
/* options.affinity_ver (model/kypt_detector.py:29,57-68,173-189): 3 (default, every shipped configuration: affinity_params (N,K,K-1)) or
 * 0 / 1 / 2 (affinity_params (N,K,K): row softmax / softplus Gram matrix, zero diagonal, row-normalised / softplus, zero diagonal, row
 * softmax), forward and backward.  Call before nm_ctx_set_weights (a change invalidates loaded weights: the parameter's shape differs).
 * Version 4 (Gumbel noise) is NM_ERR_UNSUPPORTED. */
int nm_ctx_set_affinity_ver(nm_ctx* ctx, int32_t ver);
/* options.gaussian_cat_type (model/kypt_detector.py:396-401): 0 'none' (default, every shipped configuration), 1 'max', 2 'sum' - the K
 * Gaussian channels of the voxel decoder's combined representation all carry the maximum / the sum clipped to [0, 1] over the K maps;
 * forward (nm_detector_forward*, nm_forward_fused, nm_decode_from_keypoints) and backward.  Takes effect at the next call. */
int nm_ctx_set_gaussian_cat(nm_ctx* ctx, int32_t cat);
/* options.fixed_sigma == 0 (model/kypt_detector.py:258-260, 303-306): the state_dict carries `kypt_detector.vox_to_kypt.sigmas` (K), the
 * detector's Gaussian maps take sigma_k = sigmoid(sigmas[k]) * 2 gaussian_sigma (nm_decode_from_keypoints keeps gaussian_sigma, as
 * decode_from_dyna does), and the backward writes that parameter's gradient.  Call before nm_ctx_set_weights.  Not implemented together
 * with vol_fit_type 'gaussian' (NM_ERR_UNSUPPORTED). */
int nm_ctx_set_learnable_sigma(nm_ctx* ctx, int32_t on);

/* Input path on the device: episodic_normalization (zero translation) + voxelize of
 * utils/dataset_utils.py:9-31, evaluated operation by operation in fp64 so that the voxel indices
 * are bit-exact.  points (T,N,3) float64 -> vox (T,1,G,G,G) fp32 {0,1};
 * idx_out (T,N,3) int32 receives the indices (may be NULL).  Does not need weights. */
int nm_voxelize_clip(nm_ctx* ctx, const double* points, int32_t T, int64_t N, double scale, float* vox,
                     int32_t* idx_out);

/* Evaluation metrics (utils/eval_utils.py).
 * nm_eval_voxel_chamfer — voxel_chamfer_distance :29-55 for every frame of a batch: gt_vox, recon (B,T,1,G,G,G) fp32 on the
 *   device (gt occupied = non-zero, recon occupied = value >= 0.5; neither is modified), per_frame (B*T) fp64 out =
 *   mean_gt min_rec d^2 + mean_rec min_gt d^2 in the reference's [-1,1] coordinates (NaN when a set is empty).  Exact:
 *   integer Euclidean distance transforms instead of the (N,M) distance matrix.
 * nm_eval_semantic — the nearest-keypoint votes of semantic_scores :59-90: keypoints (BT,K,4), gt_keypoints (BT,Kg,3);
 *   closest (BT,Kg) int32 out, counts (Kg,K) int64 ACCUMULATED (caller zeroes it for a new epoch). */
int nm_eval_voxel_chamfer(nm_ctx* ctx, const float* gt_vox, const float* recon, int32_t B, int32_t T, int32_t G, double* per_frame);
int nm_eval_semantic(nm_ctx* ctx, const float* keypoints, const float* gt_keypoints, int32_t BT, int32_t K, int32_t Kg,
                     int32_t* closest, int64_t* counts);

/* Skeleton handed to the VRNN entry points (result of process_affinity_glob,
 * utils/dyna_utils.py:6-171, computed on the host by neural_marionette_amd.skeleton):
 *  parents (K) int32, parents[root] == root;  order (K) int32 = priority.indices */
int nm_vrnn_set_tree(nm_ctx* ctx, const int32_t* parents_host, const int32_t* order_host);

/* HSVRNNBVH.get_offset — model/hsvrnn_bvh.py:236-253.  keypoints (B,T,K,4) -> offset (B,K,3) */
int nm_vrnn_offsets(nm_ctx* ctx, const float* keypoints, int32_t B, int32_t T, float* offset);

/* HSVRNNBVH.encode — model/hsvrnn_bvh.py:67-156.
 *  keypoints (B,T,K,4) in; eps (T,S,B,Z) standard-normal draws in t order (required);
 *  kypt_recon (B,T,K,4), R (B,T,K,3,3), z (B,T,Z), h (B,T+1,H) out;
 *  scalars2: kl_kypt (mean), kypt_recon_loss (mean);  best_idx (B,T) int32 out or NULL. */
int nm_vrnn_encode(nm_ctx* ctx, const float* keypoints, const float* eps, int32_t B, int32_t T,
                   int32_t S, float* kypt_recon, float* R, float* z, float* h, float* scalars2,
                   int32_t* best_idx);

/* ---- training, learner mode (pretrained_mode = 1: detector frozen, train.py:146) -------------------------
 * nm_vrnn_encode_train = nm_vrnn_encode + a ctx-owned tape of the activations the backward pass needs.
 * nm_vrnn_encode_backward back-propagates L = c_rec * kypt_recon_loss + c_kl * kl_kypt through time
 * (dscal2 = device pointer to [dL/dkl_kypt, dL/dkypt_recon_loss], i.e. the loss weights as autograd hands
 * them over) and overwrites the gradient buffers named like the reference's dyna_module parameters
 * (all 21 trainable tensors; offset_param has requires_grad = False, hsvrnn_bvh.py:64-65).
 * nm_adam_step is torch.optim.Adam's update (no weight decay / amsgrad) for one flat tensor. */
typedef struct nm_named_grad {
    const char* name;        /* state_dict key, e.g. "dyna_module.kypt_rnn_cell.weight_ih" */
    float* data;             /* device pointer, same shape as the parameter, overwritten */
    int64_t numel;
} nm_named_grad;
int nm_vrnn_encode_train(nm_ctx* ctx, const float* keypoints, const float* eps, int32_t B, int32_t T,
                         int32_t S, float* kypt_recon, float* R, float* z, float* h, float* scalars2,
                         int32_t* best_idx);
int nm_vrnn_encode_backward(nm_ctx* ctx, const float* dscal2, const nm_named_grad* grads, int32_t count);
int nm_adam_step(nm_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                 int32_t step, float lr, float beta1, float beta2, float eps);
/* the same update for `count` tensors in ONE launch (host arrays of device pointers; torch.optim.Adam semantics per tensor) */
int nm_adam_step_multi(nm_ctx* ctx, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numels, int32_t count, int32_t step, float lr,
                       float beta1, float beta2, float eps);
/* the same with a device-side guard: `ok` points to ONE float on the device (written earlier on the ctx stream, e.g. "every gradient
 * is finite"); 0.0f makes the launch a no-op - parameters, exp_avg and exp_avg_sq untouched, no host synchronisation - anything
 * else (or a null pointer) applies the update.  The trainers of train.py pass the finiteness of their gradient bucket. */
int nm_adam_step_multi_ok(nm_ctx* ctx, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numels, int32_t count, int32_t step, float lr,
                       float beta1, float beta2, float eps,
                          const float* ok);

/* ---- training, detector mode (pretrained_mode = 0: train.py:270-276, the detector trains on its 11 losses) ----
 * nm_ctx_set_training(ctx, 1) makes nm_ctx_set_weights also pack the weights of the data-gradient convolutions
 * (flipped / transposed copies); it invalidates the current weights, so call nm_ctx_set_weights afterwards.
 * nm_detector_forward_train = nm_detector_forward, but every activation stays in a ctx-owned arena (sized once for
 * forward + backward) together with a tape of the layer sequence.  The caller keeps vox, keypoints and recon alive
 * and unchanged until the backward call.
 * nm_detector_backward back-propagates L = sum_i dlosses11[i] * loss_i (dlosses11: DEVICE vector, the loss weights as
 * autograd hands them over; order as losses11) through the decoder, the losses, the heads and both feature nets
 * (= autograd of kypt_detector.py:81-169 under train.py:388-404) and overwrites the gradient buffers named like the
 * reference's kypt_detector.* parameters (all 315 of them must be present; layouts as in the state_dict). */
int nm_ctx_set_training(nm_ctx* ctx, int32_t on);
int nm_detector_forward_train(nm_ctx* ctx, const float* vox, int32_t B, int32_t T, int32_t affinity_on,
                              float* keypoints, float* heatmaps, float* first_feature, float* recon,
                              float* affinity, float* losses11);
int nm_detector_backward(nm_ctx* ctx, const float* dlosses11, const nm_named_grad* grads, int32_t count);
/* Optional hook for overlapping the gradient all-reduce (train.py:404 `loss.backward()` followed by the optimizer step; here one
 * collective per bucket chunk): a caller-owned hipEvent_t that the next nm_detector_backward calls record once every
 * kypt_detector.kypt_to_vox.* gradient has been written - the decoder's parameters come first in the backward order, the
 * heads and both feature nets follow.  The event is recorded on a ctx-owned stream (the one the weight gradients run on, behind the
 * ctx stream's walk through the decoder): wait for it with hipStreamWaitEvent / hipEventSynchronize, do not assume it orders anything
 * else on the ctx stream.  When nm_detector_backward returns, all of its work is ordered before anything enqueued on the ctx stream
 * afterwards.  NULL removes the hook. */
int nm_ctx_set_backward_event(nm_ctx* ctx, void* hip_event);

/* HSVRNNBVH.generate — model/hsvrnn_bvh.py:158-234.
 *  keypoints_cond (B,Tcond,K,4); eps_post (Tcond,S,B,Z); eps_prior (Ttot-Tcond,B,Z);
 *  out_cond (B,Tcond,K,4), out_gen (B,Ttot-Tcond,K,4); h_last (B,H) or NULL. */
int nm_vrnn_generate(nm_ctx* ctx, const float* keypoints_cond, const float* eps_post,
                     const float* eps_prior, int32_t B, int32_t Tcond, int32_t Ttot, int32_t S,
                     float* out_cond, float* out_gen, float* h_last);

/* Prior rollout from a given state — the generation loop of vis_generation.py:117-127 (per step: extract_prior_dist, rsample,
 * extract_kypt_from_latent_and_state, kypt_rnn_cell) for T steps in one call:
 *  h_in (B,H), offset (B,K,3) [get_offset], eps (T,B,Z); kp_out (B,T,K,4); h_out (B,H) or NULL.
 * For batches <= 64 rows this and nm_vrnn_generate replay a HIP graph of their launch sequence, captured on first use per
 * (B, Tcond, T) and kept in the context (three dependent launches per prior step; NM355_VRNN_GRAPH=0 enqueues them one by one). */
int nm_vrnn_rollout(nm_ctx* ctx, const float* h_in, const float* offset, const float* eps, int32_t B, int32_t T,
                    float* kp_out, float* h_out);

/* One VRNN step for hand-rolled rollouts (vis_generation.py:97-127 of the reference):
 *  posterior != 0: best-of-S posterior step against kp_obs (B,K*4), eps (S,B,Z);
 *  posterior == 0: prior step, eps (B,Z), S ignored.
 *  h_in (B,H), offset (B,K,3) -> kp_out (B,K*4), z_out (B,Z), h_out (B,H) (NULL: skip the GRU update). */
int nm_vrnn_step(nm_ctx* ctx, int32_t posterior, const float* h_in, const float* kp_obs,
                 const float* offset, const float* eps, int32_t B, int32_t S, float* kp_out,
                 float* z_out, float* h_out);

/* idx = argmin over rows r of || rows[r] - target[r*target_row_stride] ||^2 (first minimum; stride 0 =
 * one target for all rows): the sample selection of vis_generation.py:109-110 / vis_interpolation.py:113-119.
 *  rows (B,D), target (D) or (B,D), idx_out (1) int32 on the device, dist_out (1) or NULL. */
int nm_rows_argmin_dist(nm_ctx* ctx, const float* rows, const float* target, int32_t target_row_stride,
                        int32_t B, int32_t D, int32_t* idx_out, float* dist_out);

/* Sub-module callables the reference's demo scripts reach into (hsvrnn_bvh.py:29-57):
 *  which: 0 extract_post_dist (H+K*4 -> 2Z), 1 extract_prior_dist (H -> 2Z),
 *         2 root_intensity_decoder (H+Z -> 3+K, tanh), 3 joint_matrix_decoder (H+Z -> 6K) */
int nm_vrnn_mlp(nm_ctx* ctx, int32_t which, const float* x, int32_t B, float* y);
/* kypt_rnn_cell: x (B,K*4+Z), h (B,H) -> h_out (B,H) */
int nm_vrnn_gru(nm_ctx* ctx, const float* x, const float* h, int32_t B, float* h_out);
/* extract_kypt_from_latent_and_state — hsvrnn_bvh.py:255-286: dec_in (B,H+Z), offset (B,K,3)
 *  -> kp (B,K*4), R (B,K,3,3) */
int nm_vrnn_fk(nm_ctx* ctx, const float* dec_in, const float* offset, int32_t B, float* kp, float* R);

/* ---- op-level entry points (unit parity tests; activations are channels-last) ------------ */
/* Conv3d.  in [N][D][H][W][Cin8] (Cin8 = Cin rounded up to 8, extra channels zero), weight in
 * torch OIDHW layout, out [N][OD][OH][OW][Cout].  in_scale/in_shift [N][Cin8] or NULL apply
 * y = lrelu_slope(x*scale+shift) to the input first.  If gn_groups > 0, also returns the
 * following GroupNorm's per-(n,c) scale/shift (gn_gamma/gn_beta [Cout]) in gn_scale/gn_shift.
 * up2 != 0: `in` is at half resolution and its trilinear x2 upsampling (align_corners=False) is
 * what gets convolved (nn.Upsample fused into the conv, kypt_detector.py:427-429,441-444). */
int nm_op_conv3d(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                 const float* in_scale, const float* in_shift, float in_slope,
                 const float* weight, const float* bias, int32_t Cout, int32_t ks, int32_t stride,
                 int32_t pad, float* out, int32_t gn_groups, const float* gn_gamma,
                 const float* gn_beta, float* gn_scale, float* gn_shift, int32_t up2);
/* First layer of a feature net (kypt_detector.py:265): Conv3d(1+3 -> Cout, k5, p2) on
 * cat[occ, coord ramps] evaluated as conv(occ) + weight-only constant field.  occ [N][G][G][G],
 * weight (Cout,4,5,5,5) OIDHW, out [N][G][G][G][Cout] (+ following GroupNorm as in nm_op_conv3d). */
int nm_op_conv5_occ(nm_ctx* ctx, const float* occ, int32_t N, int32_t G, const float* weight,
                    const float* bias, int32_t Cout, float* out, int32_t gn_groups,
                    const float* gn_gamma, const float* gn_beta, float* gn_scale, float* gn_shift);
int nm_op_convT2(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                 const float* weight_iodhw, const float* bias, int32_t Cout, int32_t outpad, float* out,
                 int32_t gn_groups, const float* gn_gamma, const float* gn_beta, float* gn_scale,
                 float* gn_shift);
/* out = T_a(a) + T_b(b) with T(x) = lrelu_slope(x*scale+shift); b may be NULL */
int nm_op_apply2(nm_ctx* ctx, const float* a, const float* a_scale, const float* a_shift, float a_slope,
                 const float* b, const float* b_scale, const float* b_shift, float b_slope,
                 int32_t N, int32_t voxels, int32_t C, float* out);
int nm_op_upsample2(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, float* out);
int nm_op_pack_input(nm_ctx* ctx, const float* vox, int32_t B, int32_t T, int32_t G, int32_t mean_over_t, float* out);
int nm_op_cl_to_ncdhw(nm_ctx* ctx, const float* in, int32_t N, int32_t voxels, int32_t C, float* out);
/* Backward ops (detector-mode training, train.py:388-404 = autograd of the modules above; unit parity in
 * tests/test_grad_ops_gpu.py).  Tensors are channels-last like the forward ops.
 * nm_op_conv3d_backward: y = conv3d(up2 ? upsample2(a) : a, W) + b with a = lrelu(in*scale+shift) (vox_modules.py:12,27,31,53;
 *   kypt_detector.py:427-445).  d_weight is OIDHW, d_bias = sum dy, d_in = dL/da for the first dgrad_channels input
 *   channels [N][D][H][W][dgrad_channels] (NULL: skipped).  The data gradient is the forward conv kernel on flipped /
 *   transposed weights (stride 1), the transposed-conv kernel (k2 s2), plus the adjoint of the trilinear upsampling (up2).
 * nm_op_conv5_occ_backward: weight / bias gradients of the first layer conv5(cat[occ, coords]) (kypt_detector.py:265).
 * nm_op_convT2_backward: ConvTranspose3d(k2, s2, output_padding) of vox_modules.py:68; weight IODHW.
 * nm_op_gn_backward: GroupNorm(groups, eps 1e-5) + LeakyReLU(slope) on the raw tensor y: dy, dgamma, dbeta and sum_v dy. */
int nm_op_conv3d_backward(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                          const float* in_scale, const float* in_shift, float in_slope, const float* weight, int32_t Cout,
                          int32_t ks, int32_t stride, int32_t pad, int32_t up2, const float* dy, float* d_in,
                          int32_t dgrad_channels, float* d_weight, float* d_bias);
int nm_op_conv5_occ_backward(nm_ctx* ctx, const float* occ, int32_t N, int32_t G, int32_t Cout, const float* dy,
                             float* d_weight, float* d_bias, int32_t sparse_occ /* occupancy channel: 1 = matrix cores over the non-empty 4x8x8 bricks (the gather when G % 8 != 0), 2 = gather over the occupied voxels, 0 = generic dense kernel */);
int nm_op_convT2_backward(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                          const float* in_scale, const float* in_shift, float in_slope, const float* weight, int32_t Cout,
                          int32_t outpad, const float* dy, float* d_in, float* d_weight, float* d_bias);
int nm_op_gn_backward(nm_ctx* ctx, const float* y, int32_t N, int32_t voxels, int32_t C, int32_t groups, const float* gamma,
                      const float* beta, float slope, const float* dA, float* dy, float* dgamma, float* dbeta, float* dbias);
/* Conv arithmetic (per context: two contexts in one process may run in different modes; like every other launch-time
 * switch it lives in the nm_ctx).  mode 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32) for every conv.
 * mode 1 (default): layers with Cin % 16 == 0 run on the fp16 matrix cores with every fp32 operand split
 * into fp16 hi + lo*2^-11 and three products x_hi*w_hi + 2^-11 (x_hi*w_lo + x_lo*w_hi) accumulated in
 * fp32 — error within one fp32 rounding of the exact product, same tolerance class as the fp32 fma
 * chain, 16x/3 the MFMA rate.  mode 2 = mode 1 with the producer/consumer kernel (conv_f16p) on every layer it
 * supports instead of the Cout == 32 layers only (same arithmetic; a test / measurement switch).  Parity tests run in
 * all modes.  mode 3 = reduced precision for training at autocast-class accuracy (BASELINE.json config 3 names bf16): the
 * kernels of mode 1 with only the x_hi*w_hi product, i.e. operands rounded to fp16 (11 significant bits - three more than
 * bf16, same MFMA rate), fp32 accumulation, fp32 tensors, statistics, losses, master weights and optimiser; the layers mode 1
 * leaves on the fp32 cores (k = 1 weight gradients, volumes under 16^3, heads, VRNN) stay fp32.  Forward, data and weight
 * gradients all follow the mode.  Outputs differ from the reference's fp32 path by ~1e-3 relative (tests state the bound);
 * the 1e-4 parity contract holds in modes 0-2 only.
 * mode 4 = BASELINE config 3 as named ("bf16"): mode 3's arithmetic with 16-BIT STORAGE of the training path - every activation the
 * training forward keeps for the backward pass and every activation gradient with at least 32^3 voxels per frame is stored as
 * bfloat16 (8 significant bits, fp32's range: no loss scaling), converted to fp32 on read and rounded to nearest even on write;
 * master weights, GroupNorm statistics / scale / shift, every partial sum and accumulator, the losses and the Adam state stay fp32,
 * as do the tensors below 32^3 (the hourglass, heads, keypoints) and the inference forward (which keeps mode 3's fp32 workspace).
 * Halves the training arena and the bytes of the GroupNorm-backward passes; gradients agree with the fp64 oracle to a few 1e-2 in
 * whole-gradient L2 (tests state the bound). */
int nm_set_conv_mode(nm_ctx* ctx, int32_t mode);
int nm_get_conv_mode(nm_ctx* ctx);
/* Element type of the tensors the op-level entry points below (nm_op_*) read and write, for unit parity of the 16-bit storage kernels
 * of conv mode 4: in_h != 0 - the tensors on the input side (in / a / y and every gradient of their shape: d_in, dA, dy of
 * nm_op_gn_backward) are bfloat16; out_h != 0 - the tensors on the output side (out, the incoming dy of the backward ops) are.
 * Defaults 0 / 0 (fp32, the element type of every network-level entry point's arguments).  Kernels without an instantiation for the
 * requested combination fail with NM_ERR_UNSUPPORTED. */
int nm_op_set_storage16(nm_ctx* ctx, int32_t in_h, int32_t out_h);

/* ---- live kernel timing for bench.py's roofline leg -------------------------------------------
 * While enabled, every conv launch of this context is bracketed by a HIP event pair.  on = 1: launches on the ctx stream only
 * (launches the library puts on its own side stream overlap the main stream, so their event-to-event time is not their own);
 * on = 2: side-stream launches are recorded too (for listing every kernel family's share; durations include contention).
 * on = 3: as 1, but only launches of >= 20 GFLOP algorithmic work (what bench.py's timed region uses: the event pairs around the
 * many small dependent launches of the coarse hourglass levels cost the step ~1 % and no roofline is read from them).
 * Records belong to the context.
 * nm_prof_read sums duration and ALGORITHMIC flops (2*voxels*Cout*Cin*k^3, un-padded) of one
 * kernel variant (0..3 = conv_mfma_kernel<MT,NT> with (MT,NT) = (1,1),(1,2),(2,1),(2,2); 5,6 =
 * conv_f16s_kernel<2,1>/<2,2> (k1 / k3 without upsampling), 10,11 = the same kernel with the fused trilinear upsampling (NT = 1 / 2),
 * 7 = conv_f16p_kernel, 8 = conv_pool_f16s_kernel, 9 = conv_f16p2_kernel (algorithmic fp32-equivalent flops, i.e. 1/3 of the issued MFMA flops); 4 = the
 * first-layer occupancy kernel conv_k5occ_kernel, credited with the reference's dense 4-channel k5 work; 12 = conv_up2c_kernel (main + shell
 * launches); 13 = wgrad16_kernel, the split-fp16 k3 weight-gradient kernels with their fixed-order reduce (2*voxels*Cout*Cin*27);
 * 14 = conv_f16q2_kernel, the one-product modes' 64-channel conv; 15 = conv_f16r_kernel, their 32-output-channel conv with resident weights). */
int nm_prof_enable(nm_ctx* ctx, int32_t on);
int nm_prof_read(nm_ctx* ctx, int32_t variant, double* ms_total, double* flops_total, int64_t* launches);
const char* nm_prof_kernel_name(int32_t variant);
/* host helper: torch.linspace(-1, 1, n) as the kernels evaluate it */
int nm_host_linspace(int32_t n, float* out_host);

#ifdef __cplusplus
}
#endif
#endif /* NM355_H */
