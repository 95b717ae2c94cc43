// launchers of nm_heads_bwd.hip (adjoints of nm_heads.hip); `dloss` = device vector d(total)/d(loss_i), i < 11
#pragma once
#include "nm_common.h"

int nm_tail_bwd_blocks(int G);
// dA [F][G^3][C] = gradient w.r.t. the activated decoder output; part [F][blocks][C+1] -> (d w14, d b14) via nm_launch_sum_rows
// dvout (optional, C == 32): only the per-voxel factor is written, [F][G^3] - dA = dvout (x) w14[0..C) is never materialised
int nm_launch_decoder_tail_bwd(const TensorRef& x, const float* w14, const float* target, const float* recon, const float* dloss, int G,
                               float* dA, float* part, hipStream_t s, float* dvout = nullptr);
int nm_launch_sum_rows(const float* part, int rows, int cols, float* out, hipStream_t s);
int nm_chamfer_bwd_blocks(int G);
// ws: F * blocks * K * 3 floats; dkp [F][K][4] is accumulated into
int nm_launch_chamfer_bwd(const float* target, const float* keypoints, const float* tail_part, int tail_blocks, const float* dloss, int F,
                          int K, int G, float* ws, float* dkp, hipStream_t s);
// dcomb [F][g^3][Cd] (channels [0,K) gauss_t | [K,K+Fd) first feature | [K+Fd,2K+Fd) gauss_0); ws: F*K*8 floats
int nm_launch_combined_bwd(const float* dcomb, int Cd, const float* table, const float* keypoints, int B, int T, int K, int Fd, int g,
                           float width, float* ws /* F K 8 floats; F K 10 with widthk */, float* dfeat, float* dkp, hipStream_t s,
                           int cat = 0 /* gaussian_cat_type none / max / sum */, const float* widthk = nullptr /* [K] per-keypoint widths (fixed_sigma = 0) */,
                           const float* sigma_param = nullptr, float max_sigma = 0.f, float* dsigma_param = nullptr /* [K] out */);
size_t nm_heat_bwd_ws_floats(int F, int K, int g);
int nm_launch_heat_bwd(const float* head, const float* clip_head, const float* prop, const float* heat_part, const float* heat_mean,
                       const float* keypoints, const float* dkp, const float* dloss, int B, int T, int K, int Kc /* channels per voxel of the head tensors (>= K) */, int g, float* ws, float* dhead,
                       float* dchead_t, float* dclip_head, float* dprop, hipStream_t s);
// dinfl [B][K][K] (written when affinity != nullptr)
int nm_launch_clip_loss_bwd(const float* keypoints, const float* affinity, const float* dloss, int B, int T, int K, int N, float sep_sigma,
                            int use_traj, float* dkp, float* dinfl, hipStream_t s);
// vol_fit_type 'gaussian' (kypt_detector_utils.py:154-169): dkp += d (dloss[1] * volume_fitting_loss) / d keypoints
size_t nm_volfit_gauss_bwd_ws_floats(int B, int T, int G);
int nm_launch_volfit_gauss_bwd(const float* vox, const float* keypoints, const float* dloss, int B, int T, int K, int G, float sigma,
                               float* ws, float* dkp, hipStream_t s);
int nm_launch_affinity_bwd(const float* params, const float* affinity, const float* dinfl, const float* dloss, int B, int N, int K,
                           float* dparams, hipStream_t s, int ver = 3);
