// Fused nn.Upsample(x2, trilinear, align_corners=False) + Conv3d(k3, s1, p1) on the COARSE grid (nm_up2c.hip):
// the two resolution-doubling layers of the voxel decoder (model/kypt_detector.py:425-447 of the reference).
#pragma once
#include "nm_common.h"

// true when the layer shape is taken by the composite-weight kernel (otherwise conv_f16s<.., UP2> of nm_conv.hip runs it)
bool nm_up2c_eligible(int ID, int IH, int IW, int Cin, int Cout, int ks, int stride, int pad);
// floats (4-byte units) of the packed composite weight sets of one layer: 8 parity sets of 27 coarse taps + 56 shell-correction sets
size_t nm_up2c_weight_floats(int Cin, int Co_pad);
// OIDHW fp32 (Cout, Cin, 3, 3, 3) -> every composite set, split fp16, MFMA operand layout
int nm_launch_up2c_compose(const float* w_oidhw, int Cout, int Cin, int Co_pad, void* packed, hipStream_t s);
// GroupNorm partial blocks per frame the two launches write (bricks + shell items)
int nm_up2c_blocks_per_frame(int ID, int IH, int IW);
// in: the coarse lazy tensor (N, ID, IH, IW, Cin); out: (N, 2ID, 2IH, 2IW, Cout) raw conv result; part: [N][blocks][Cout][2] or null
int nm_launch_conv_up2c(const TensorRef& in, const void* packed, const float* bias, float* out, int Cout, int Co_pad,
                        float* part, hipStream_t s, int out_h = 0 /* 1: bfloat16 output (then the input must be bfloat16 too: in.h) */);
