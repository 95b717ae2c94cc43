// The two lowest hourglass levels in one launch (nm_hgcore.hip).
#pragma once
#include "nm_common.h"

struct NmHgConv { const void* w16; const float* bias; int Cin, Cout, Co_pad, ks; };
struct NmHgNorm { const float* gamma; const float* beta; int groups; };
struct NmHgRes { NmHgConv c1, c2, cs; NmHgNorm n1, n2, ns; int has_skip; };
struct NmHgCoreParams {
    const float* in; const float* in_scale; const float* in_shift; float in_slope;   // [N][D2^3][Cin0]: the pool conv's lazy output
    float* out;                                                                          // [N][D2^3][d2 Cout] plain
    int N, D2, D3, Cin0, pitch2, pitch3;
    int scratch_items;                                                                   // nm_hg_core_scratch_items(p): partial-tile slots of conv_lds (0: every conv stores directly)
    NmHgRes e2, s3, e3, d3, d2;
    NmHgConv p3; NmHgNorm np3;
    const float* u3_w; const float* u3_bias; NmHgNorm nu3; int u3_Cin, u3_Cout;           // transposed conv: weights [tap][Cin][Cout] fp32
};
size_t nm_hg_core_lds_bytes(const NmHgCoreParams& p);
int nm_hg_core_scratch_items(const NmHgCoreParams& p);      // < 0: the frame's tensors do not fit in LDS at all
int nm_launch_hg_core(const NmHgCoreParams& p, hipStream_t s);
