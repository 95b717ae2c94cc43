// launchers of nm_grad.hip: backward kernels of the conv stack (detector-mode training)
#pragma once
#include "nm_common.h"

// ---- weight gradients ------------------------------------------------------------------------------------------------
// dW[m][c][tap] = sum_{n,v} dy[n, v, m] * act(in)[n, stride*v + tap - pad, c]      (torch OIDHW layout, fp32 MFMA, exact)
// Both operands are lazy tensors (ConvTranspose3d weight gradients swap the roles: the activated input plays dy).
// `ws` holds the per-workgroup partial tiles (nm_wgrad_ws_floats), reduced in a fixed order: deterministic, no atomics.
size_t nm_wgrad_ws_floats(int N, int OD, int OH, int OW, int M, int Nc, int ks, int stride);
// mul (optional): device scalar multiplied into the result (dy was read pre-scaled by its inverse, see nm_launch_make_scale);
// allow_f16: 3x3x3 stride-1 layers whose extents fit the 2x8x8 brick run on the split-fp16 MFMA kernel (fp32-equivalent products)
int nm_launch_wgrad(const TensorRef& in, const TensorRef& dy, int ks, int stride, int pad, int cin_real, float* ws,
                    float* dW, hipStream_t s, const float* mul = nullptr, int allow_f16 = 0);
// first layer (Basic3DBlock k5 on cat[occ, x1, x2, x3], kypt_detector.py:265): the input is rebuilt from the occupancy
// grid while staging; dW is [Cout][4][5][5][5]
size_t nm_wgrad_k5occ_ws_floats(int N, int G, int M);
// sparse_occ: the occupancy is a few per cent dense (per-frame grids): its channel is gathered over the occupied voxels and the
// coordinate channels take the dense kernel on one frame of frame-summed dy; 0: dense kernel over all frames (the clip-mean grid
// of the spatio-temporal net is the union of T frames, 15-30 % dense: the gather took 11.7 ms there against 0.5 ms)
// s_coord + the two events (optional): the coordinate channels' part runs on that stream beside the occupancy channel's kernel
int nm_launch_wgrad_k5occ(const float* occ, int N, int G, const TensorRef& dy, float* ws, float* dW, hipStream_t s, int sparse_occ = 1,
                          hipStream_t s_coord = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr);

// ---- GroupNorm + LeakyReLU backward ------------------------------------------------------------------------------------
// y: the raw conv output with its forward scale/shift/slope (lazy tensor);  dA: gradient w.r.t. the activated values.
//   dz = dA * lrelu'(scale*y + shift);   partials: per (frame, block, channel) (sum dz, sum dz*y)
int nm_gnb_blocks_per_frame(int voxels);
// dA_mul (optional, here and in gnb_apply / absmax): device scalar dA is multiplied by on read (the producer left it scaled by 2^k)
// dv, wv (optional, here and in gnb_apply): dA is the outer product dv[n][voxel] * wv[channel] and the dA pointer is not read
int nm_launch_gnb_partials(const float* dA, const TensorRef& y, float* part, hipStream_t s, const float* dA_mul = nullptr,
                           const float* dv = nullptr, const float* wv = nullptr);
// coef[n][c] = (c1, c2, c3, 0) with dy = c1*dz + c2*y + c3;  dgn[n][c] = (dgamma_n, dbeta_n, dbias_n, 0)
// fpart: the forward partial sums (sum y, sum y^2) the conv epilogue left, [N][nblk_f][C][2]; chsum (optional): the per-channel totals
// of them that the forward finalisation left ([N][C][2] doubles, nm_launch_gn_finalize) - then fpart is not read
int nm_launch_gnb_finalize(const float* bpart, int nblk_b, const float* fpart, int nblk_f, int N, int C, int groups, int voxels,
                           const float* gamma, float eps, float* coef, float* dgn, hipStream_t s, const double* chsum = nullptr);
// out[c] = sum_n src[(n*C + c)*stride + off]
int nm_launch_sum_frames(const float* src, int N, int C, int stride, int off, float* out, hipStream_t s);
int nm_launch_sum_frames3(const float* dgn, int N, int C, float* dgamma, float* dbeta, float* dbias, hipStream_t s);
// the same for many layers at once (host array of jobs, NM_SUM3_JOBS per launch); results are bit-identical to the single launches
#define NM_SUM3_JOBS 24
struct NmSum3Job { const float* dgn; float* o0; float* o1; float* o2; int N, C; };
struct NmSum3Jobs { NmSum3Job j[NM_SUM3_JOBS]; };
int nm_launch_sum_frames3_multi(const NmSum3Job* jobs, int njobs, hipStream_t s);
// out[c] = sum_{n,blk} part[((n*nblk + blk)*C + c)*2]       (bias gradient of a conv without GroupNorm)
int nm_launch_sum_partials(const float* part, int rows, int C, float* out, hipStream_t s);
// dy = c1*dz + c2*y + c3 (coef) or dy = dz (coef == nullptr)
// amax (optional): device word that receives max |dy| as float bits (integer atomicMax; zero it first)
int nm_launch_gnb_apply(const float* dA, const TensorRef& y, const float* coef, float* dy, hipStream_t s, unsigned* amax = nullptr,
                        const float* dA_mul = nullptr, const float* dv = nullptr, const float* wv = nullptr);
// Power-of-two operand scaling of the data-gradient convolutions in the split-fp16 conv mode: gradients are often
// below the fp16 normal range (6e-5), where the hi/lo split loses its low bits; dy is read as dy * 2^k through the lazy
// affine of the conv kernels and the result is multiplied by 2^-k (exact).
// h (here and below): the raw tensors are stored as bfloat16 (16-bit storage mode); for the GroupNorm backward y.h covers y, dA and dy
int nm_launch_absmax(const float* x, size_t n, unsigned* amax, hipStream_t s, const float* mul = nullptr, int h = 0);
int nm_launch_make_scale(const unsigned* amax, int count, float* scale, float* sc2 /*[2^k, 2^-k]*/, hipStream_t s, float* zero_shift = nullptr);
int nm_launch_scale_by(float* x, size_t n, const float* mul, hipStream_t s, int h = 0);

// ---- misc --------------------------------------------------------------------------------------------------------------
// adjoint of nn.Upsample(x2, trilinear, align_corners=False): dfine [N][2D][2H][2W][C] -> dcoarse [N][D][H][W][C]
int nm_launch_upsample2_adjoint(const float* dfine, int N, int D, int H, int W, int C, float* dcoarse, hipStream_t s,
                                const float* mul = nullptr /* device scalar multiplied into the result */, int h_fine = 0, int h_coarse = 0);
// OIDHW weights of the data-gradient convolution: out[ci][co][K-1-tap] = w[co][ci][tap], ci < csel
int nm_launch_flip_weight(const float* w, int Cout, int Cin, int csel, int ks, float* out, hipStream_t s);
int nm_launch_axpy(float* dst, const float* src, size_t n, hipStream_t s, int h = 0);      // dst += src
// dst = dst * *dst_mul + src * *src_mul (device scalars, null = 1); n % 4 == 0
int nm_launch_axpby(float* dst, const float* dst_mul, const float* src, const float* src_mul, size_t n, hipStream_t s, int h = 0);
