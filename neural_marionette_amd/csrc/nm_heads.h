// launchers of nm_heads.hip
#pragma once
#include "nm_common.h"

int nm_launch_heatmap(const float* head, const float* clip_head, const float* prop, int F, int T, int K, int Kc /* channels per voxel of head / clip_head (>= K, % 4 == 0) */, int g,
                      float* heatmaps, float* part, hipStream_t s);
int nm_launch_keypoints(const float* part, int F, int K, int g, float* keypoints, float* heat_mean, hipStream_t s);
int nm_launch_gauss_table(const float* keypoints, int FK, int g, float width, float* table, hipStream_t s,
                          const float* widthk = nullptr /* [K]: per-keypoint widths (fixed_sigma = 0) */, int K = 0);
int nm_launch_gauss_width(const float* param, int K, float max_sigma, int g, float* widthk, hipStream_t s);
int nm_launch_combined(const float* table, const float* keypoints, const float* first_feature, int ff_stride, int F,
                       int T, int K, int Fd, int g, int Cc, float* out, hipStream_t s, int cat = 0);
// the 1x1 conv of the combined representation split by linearity (inference): per-clip part / per-frame part, see nm_heads.hip
int nm_launch_combined_rest(const float* table, const float* keypoints, const float* first_feature, int ff_stride, int nb, int T,
                            int K, int Fd, int g, int Cr, float* out, hipStream_t s);
int nm_launch_adjust_gauss(const float* table, const float* keypoints, const float* base, const float* wg, int F, int T, int K, int g,
                           int Cout, float* out, hipStream_t s);
int nm_launch_adjust_wg(const float* w, int Cout, int Cin_total, int K, float* wg, hipStream_t s);
int nm_tail_blocks(int G);
int nm_launch_decoder_tail(const TensorRef& x, const float* w14, const float* first_frames, int ff_stride_frames, int T,
                           const float* target, const float* keypoints, int K, int G, float* recon, float* part,
                           hipStream_t s);
int nm_launch_clip_loss(const float* keypoints, const float* affinity, int B, int T, int K, int N, float sep_sigma,
                        float* out, hipStream_t s);
int nm_launch_loss_finalize(const float* tail_part, int tail_blocks, int B, int T, int K, int N, int G,
                            const float* heat_mean, const float* clip_part, const float* affinity, int chamfer,
                            int use_traj, float* frame_sums /* scratch [B*T][3] */, float* losses, hipStream_t s,
                            const float* vol_override = nullptr /* [B*T][2]: vol_fit_type 'gaussian' */);
// vol_fit_type 'gaussian' (kypt_detector_utils.py:154-169): per frame (numerator, denominator) into vol [B*T][2]
size_t nm_volfit_gauss_ws_floats(int F, int G);
int nm_launch_volfit_gauss(const float* vox, const float* keypoints, int B, int T, int K, int G, float sigma, float* ws, float* vol, hipStream_t s);
int nm_launch_affinity(const float* params, int N, int K, float* out, hipStream_t s, int ver = 3);     // ver: get_affinity version (0-3)
// episodic_normalization + voxelize (utils/dataset_utils.py:9-31) on the device, fp64, bit-exact indices
int nm_launch_voxelize(const double* pts, int T, size_t N, int G, double scale, double* part_ws, float* vox, int32_t* idx_out,
                       hipStream_t s);
