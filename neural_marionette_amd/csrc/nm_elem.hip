// HBM-bound helper kernels around the conv stack (all channels-last, 16-B vector access):
//   gn_finalize   per-block (sum, sumsq) partials -> per-(frame, channel) scale / shift
//                 (nn.GroupNorm(C//16, C), eps 1e-5, biased variance: vox_modules.py:14,28,32,41,55,70)
//   gn_partials   partials of a materialised tensor (producers without a stats epilogue)
//   apply2        out = T_a(a) + T_b(b): residual sums of vox_modules.py:44-47,109-118
//   convT2        ConvTranspose3d(k2, s2, output_padding): vox_modules.py:68
//   upsample2     nn.Upsample(x2, trilinear, align_corners=False): kypt_detector.py:427,441
//   pack_input    occupancy clip -> [frame][voxel][occ, x1, x2, x3, 0,0,0,0]
//                 (add_coord_channels, kypt_detector_utils.py:4-26; clip mean kypt_detector.py:312)
//   cl_to_ncdhw   channels-last -> NCDHW (first_feature output of kypt_detector.py:166)
#include "nm_common.h"

namespace {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// HSEL: -1 run-time storage type (t.h, a uniform branch per load), 0 / 1 compile-time fp32 / bfloat16 (the batched loaders)
template <int HSEL = -1>
__device__ __forceinline__ f32x4 load_t(const TensorRef& t, size_t n, size_t vox_in_frame_times_C_plus_c, int c) {
    const size_t e = n * ((size_t)t.D * t.H * t.W * t.C) + vox_in_frame_times_C_plus_c;
    f32x4 v = HSEL < 0 ? nm_ld4(t.p, e, t.h) : (HSEL ? nm_ld4<true>(t.p, e) : nm_ld4<false>(t.p, e));     // (fp32 or bf16 storage)
    if (t.scale) {
        f32x4 sc = *reinterpret_cast<const f32x4*>(t.scale + n * t.C + c);
        f32x4 sh = *reinterpret_cast<const f32x4*>(t.shift + n * t.C + c);
        v = v * sc + sh;
    }
    if (t.slope != 1.0f) {
        v[0] = lrelu(v[0], t.slope); v[1] = lrelu(v[1], t.slope);
        v[2] = lrelu(v[2], t.slope); v[3] = lrelu(v[3], t.slope);
    }
    return v;
}

// one block per (frame, group): NT threads walk the (block, channel) partials (float2 loads, four independent chains per
// thread), fp64 sums combined in a fixed order (run-to-run identical).  NT = 256 for the usual few thousand partials, 1024 for the
// layers whose kernels leave several thousand blocks per frame (a 256-thread block took 96 us there).
template <int NT>
__global__ __launch_bounds__(NT) void gn_finalize_kernel(const float* __restrict__ part, int nblk, int C, int groups,
                                                           double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           float* __restrict__ scale, float* __restrict__ shift,
                                                           unsigned* __restrict__ nonfinite, double* __restrict__ chsum) {
    __shared__ double sh_s[NT], sh_ss[NT];
    const int n = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    const int total = nblk * cpg;
    const float* base = part + (size_t)n * nblk * C * 2;
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * NT) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT;
            if (i < total) {
                const int blk = i / cpg, c = g * cpg + i % cpg;
                const float2 q = *reinterpret_cast<const float2*>(base + ((size_t)blk * C + c) * 2);
                s[u] += (double)q.x; ss[u] += (double)q.y;
            }
        }
    }
    sh_s[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]); sh_ss[threadIdx.x] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
    __syncthreads();
    // training: the per-channel (sum y, sum y^2) for the GroupNorm backward, which otherwise walks these partials a second time
    // (with NT a multiple of cpg every item of a thread belongs to channel tid % cpg)
    if (chsum) {
        if ((int)threadIdx.x < cpg) {
            double a = 0.0, b = 0.0;
            for (int l = 0; l < NT / cpg; ++l) { a += sh_s[l * cpg + threadIdx.x]; b += sh_ss[l * cpg + threadIdx.x]; }
            double* d = chsum + ((size_t)n * C + g * cpg + threadIdx.x) * 2;
            d[0] = a; d[1] = b;
        }
        __syncthreads();
    }
    for (int st = NT / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { sh_s[threadIdx.x] += sh_s[threadIdx.x + st]; sh_ss[threadIdx.x] += sh_ss[threadIdx.x + st]; }
        __syncthreads();
    }
    const double mean = sh_s[0] / count;
    double var = sh_ss[0] / count - mean * mean;
    // a non-finite sum means the conv produced inf / NaN: in the split-fp16 mode that is what an operand beyond the fp16 range
    // (|x| >= 65520) turns into, where the reference's fp32 arithmetic stays finite - reported through the ctx's sticky flag
    if (nonfinite && threadIdx.x == 0 && !(isfinite(sh_s[0]) && isfinite(sh_ss[0]))) atomicOr(nonfinite, 1u);
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    if ((int)threadIdx.x < cpg) {
        int c = g * cpg + threadIdx.x;
        float sc = (float)rstd * gamma[c];
        scale[(size_t)n * C + c] = sc;
        shift[(size_t)n * C + c] = -sc * (float)mean + beta[c];
    }
}

// DIAGNOSTIC (NM355_GN_DIAG=1): GroupNorm statistics straight from the stored raw tensor, two passes in fp64 - the reference point for
// the accuracy of the single-pass (sum x, sum x^2) block partials the conv epilogues produce.  One block per (frame, group).
__global__ __launch_bounds__(1024) void gn_direct_kernel(const float* __restrict__ x, int voxels, int C, int groups, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                        double* __restrict__ chsum, int flags) {
    __shared__ double sh[1024];
    __shared__ double mean_s;
    const int n = blockIdx.x / groups, g = blockIdx.x % groups, cpg = C / groups;
    const int c = g * cpg + threadIdx.x % cpg, lanes = 1024 / cpg;
    const int vl = (int)threadIdx.x / cpg < lanes ? (int)threadIdx.x / cpg : voxels;     // (threads beyond lanes * cpg idle)
    const float* base = x + (size_t)n * voxels * C;
    double s = 0.0;
    for (int v = vl; v < voxels; v += lanes) s += (double)base[(size_t)v * C + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    double chs = 0.0;
    if ((int)threadIdx.x < cpg) for (int l = 0; l < lanes; ++l) chs += sh[l * cpg + threadIdx.x];
    __syncthreads();
    if ((int)threadIdx.x < cpg) sh[threadIdx.x] = chs;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < cpg; ++i) t += sh[i]; mean_s = t / ((double)voxels * cpg); }
    __syncthreads();
    const double mean = mean_s;
    double q = 0.0, q2 = 0.0;
    for (int v = vl; v < voxels; v += lanes) { const double d = (double)base[(size_t)v * C + c] - mean; q += d * d; q2 += (double)base[(size_t)v * C + c] * (double)base[(size_t)v * C + c]; }
    __syncthreads();
    sh[threadIdx.x] = q;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st]; __syncthreads(); }
    const double var = sh[0] / ((double)voxels * cpg);
    const double rstd = 1.0 / sqrt(var + (double)eps);
    __syncthreads();
    if (chsum && (flags & 2)) {
        sh[threadIdx.x] = q2;
        __syncthreads();
        if ((int)threadIdx.x < cpg) {
            double b = 0.0;
            for (int l = 0; l < lanes; ++l) b += sh[l * cpg + threadIdx.x];
            chsum[((size_t)n * C + c) * 2] = chs; chsum[((size_t)n * C + c) * 2 + 1] = b;
        }
    }
    if ((int)threadIdx.x < cpg && (flags & 1)) {
        const float sc = (float)rstd * gamma[c];
        scale[(size_t)n * C + c] = sc;
        shift[(size_t)n * C + c] = -sc * (float)mean + beta[c];
    }
}

#define NM_STATS_VB 512
__global__ __launch_bounds__(256) void gn_partials_kernel(const float* __restrict__ x, int voxels, int C, int nblk,
                                                          float* __restrict__ part, int h) {
    __shared__ float sh[256 * 2];
    const int n = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int lanes = 256 / C;
    const int c = threadIdx.x % C, vl = threadIdx.x / C;
    float s = 0.f, ss = 0.f;
    if (vl < lanes) {
        const int v0 = blk * NM_STATS_VB, v1 = min(voxels, v0 + NM_STATS_VB);
        for (int v = v0 + vl; v < v1; v += lanes) {
            float t = nm_ld1(x, ((size_t)n * voxels + v) * C + c, h);
            s += t; ss += t * t;
        }
    }
    sh[threadIdx.x * 2] = s; sh[threadIdx.x * 2 + 1] = ss;
    __syncthreads();
    if (threadIdx.x < C) {
        float a = 0.f, b = 0.f;
        for (int l = 0; l < lanes; ++l) { a += sh[(l * C + c) * 2]; b += sh[(l * C + c) * 2 + 1]; }
        float* dst = part + (((size_t)n * nblk + blk) * C + c) * 2;
        dst[0] = a; dst[1] = b;
    }
}

__global__ __launch_bounds__(256) void apply2_kernel(TensorRef a, TensorRef b, int has_b, float* __restrict__ out, int oh) {
    // grid (blocks, frames): 32-bit index arithmetic inside a frame (the flat 64-bit index cost a 64-bit division and two
    // remainders per 16-byte item)
    const unsigned per_frame = (unsigned)a.D * a.H * a.W * a.C;     // floats
    const size_t n = blockIdx.y;
    for (unsigned r = (blockIdx.x * blockDim.x + threadIdx.x) * 4u; r < per_frame; r += gridDim.x * blockDim.x * 4u) {
        const int c = (int)(r % (unsigned)a.C);
        f32x4 v = load_t(a, n, r, c);
        if (has_b) v = v + load_t(b, n, r, c);
        nm_st4(out, n * per_frame + r, v, oh);
    }
}

// ConvTranspose3d k2 s2: out[2i+a] += x[i] * W[ci][co][a]; one thread per (out voxel, 4 channels);
// the weights arrive transposed to [tap][ci][co] (nm_launch_transpose_convT_weight)
__global__ __launch_bounds__(256) void convT2_kernel(TensorRef in, const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ out, int Cout, int OD, int OH, int OW, int oh) {
    const int cq = Cout / 4;
    const size_t total = (size_t)in.N * OD * OH * OW * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int q = (int)(i % cq); size_t r = i / cq;
        int ox = (int)(r % OW); r /= OW;
        int oy = (int)(r % OH); r /= OH;
        int oz = (int)(r % OD); size_t n = r / OD;
        int co = q * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + co);
        int iz = oz >> 1, iy = oy >> 1, ix = ox >> 1;
        if (iz < in.D && iy < in.H && ix < in.W) {
            int tap = ((oz & 1) * 2 + (oy & 1)) * 2 + (ox & 1);
            size_t vo = (((size_t)iz * in.H + iy) * in.W + ix) * in.C;
            for (int c = 0; c < in.C; c += 4) {
                f32x4 x = load_t(in, n, vo + c, c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // w is pre-transposed to [tap][ci][co]: one 16-B load, coalesced across the threads of a voxel
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ((size_t)tap * in.C + c + j) * Cout + co);
                    acc[0] += x[j] * wv[0]; acc[1] += x[j] * wv[1]; acc[2] += x[j] * wv[2]; acc[3] += x[j] * wv[3];
                }
            }
        }
        nm_st4(out, ((((size_t)n * OD + oz) * OH + oy) * OW + ox) * Cout + co, acc, oh);
    }
}

// Same op for the large launches (the data gradient of the k2 s2 pool convs: 32 ch -> 64^3 is 2.1 GB of output): grid (chunks, 8
// taps); the tap's [ci][co] weight slice sits in LDS (the kernel above re-fetches 4 weight vectors per 16 FMAs through the
// texture path: 2.7 ms for that launch, load-issue bound), a thread owns (coarse voxel, 4 output channels) and walks the chunk.
__global__ __launch_bounds__(256) void convT2_lds_kernel(TensorRef in, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int Cout, int OD, int OH, int OW, int vox_per_block, int oh) {
    extern __shared__ float wl[];            // [Cin][Cout]
    const int tap = blockIdx.y, cq = Cout / 4, q = threadIdx.x % cq, vl = threadIdx.x / cq, vpi = 256 / cq;
    for (int i = threadIdx.x; i < in.C * Cout / 4; i += 256)
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(w + (size_t)tap * in.C * Cout)[i];
    __syncthreads();
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + q * 4);
    const size_t cvox = (size_t)in.D * in.H * in.W, total = (size_t)in.N * cvox;
    const size_t v0 = (size_t)blockIdx.x * vox_per_block, v1 = min(total, v0 + vox_per_block);
    const int az = tap >> 2, ay = (tap >> 1) & 1, ax = tap & 1;
    // four coarse voxels per pass: one LDS weight read feeds 4 x 4 FMAs
    for (size_t vb = v0 + (size_t)vl * 4; vb < v1; vb += (size_t)vpi * 4) {
        size_t vo[4], oo[4]; size_t nn[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t v = vb + u;
            ok[u] = v < v1;
            const size_t vv = ok[u] ? v : v0;
            const size_t n = vv / cvox; size_t r = vv % cvox;
            const int ix = (int)(r % in.W); r /= in.W;
            const int iy = (int)(r % in.H), iz = (int)(r / in.H);
            nn[u] = n;
            vo[u] = (((size_t)iz * in.H + iy) * in.W + ix) * in.C;
            oo[u] = ((((size_t)n * OD + 2 * iz + az) * OH + 2 * iy + ay) * OW + 2 * ix + ax) * Cout + q * 4;
        }
        f32x4 acc[4] = {bv, bv, bv, bv};
        for (int c = 0; c < in.C; c += 4) {
            f32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = load_t(in, nn[u], vo[u] + c, c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (size_t)(c + j) * Cout + q * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc[u][0] += x[u][j] * wv[0]; acc[u][1] += x[u][j] * wv[1]; acc[u][2] += x[u][j] * wv[2]; acc[u][3] += x[u][j] * wv[3]; }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (ok[u]) nm_st4(out, oo[u], acc[u], oh);
    }
}

// The same op on the fp32 matrix cores (exact fp32 products).  Per tap a it is a plain GEMM out[2i + a][co] = sum_ci X[i][ci] W[a][ci][co]:
// a wave owns 32 consecutive coarse voxels (M), stages their activated input once in its own LDS slice ([32][Cin + 4] floats), and
// for each tap and each 32-channel output tile runs Cin / 2 v_mfma_f32_32x32x2_f32 with the weight operand straight from L1/L2 (the
// 8 x Cin x Cout weights are at most 131 KB and hot); a lane ends up with 16 voxels of ONE output channel, so a store instruction
// writes two complete 128-byte voxel lines when Cout = 32.  convT2_lds_kernel above is VALU-FMA bound (1.06 ms for the 32 -> 32
// data gradient of the first pool, whose 2.1 GB output takes 0.5 ms to write) and re-reads the input once per tap.
// OH16: bfloat16 output - two lanes that hold neighbouring channels exchange values (one DPP move per register pair) so that every lane
// stores ONE packed dword per two accumulator registers: even lanes the channel pair of voxel row r, odd lanes that of row r + 1
template <bool OH16, bool IH16 = false>
__global__ __launch_bounds__(256) void convT2_mfma_kernel(TensorRef in, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int Cout, int tiles_per_frame, int total_tiles) {
    extern __shared__ float xs_all[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, wv = tid >> 6;
    const int Cin = in.C, P = Cin + 4;
    float* xs = xs_all + (size_t)wv * 32 * P;
    const int OH = 2 * in.H, OW = 2 * in.W;
    const size_t out_frame = (size_t)8 * in.D * in.H * in.W * Cout, in_frame = (size_t)in.D * in.H * in.W * Cin;
    for (int t = blockIdx.x * 4 + wv; t < total_tiles; t += gridDim.x * 4) {
        const int n = t / tiles_per_frame, v0 = (t % tiles_per_frame) * 32;
        // stage: 32 voxels x Cin, activated (lanes walk the 16-byte items of the tile, contiguous in memory)
        const size_t src = (size_t)n * in_frame + (size_t)v0 * Cin;   // (element offset)
        const int items = 8 * Cin;                                     // 32 * Cin / 4
        for (int i0 = lane; i0 < items; i0 += 256) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u; v[u] = i < items ? nm_ld4<IH16>(in.p, src + (size_t)i * 4) : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u;
                if (i >= items) continue;
                const int e = i * 4, vox = e / Cin, c = e % Cin;
                f32x4 x = v[u];
                if (in.scale) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(in.scale + (size_t)n * Cin + c), sh = *reinterpret_cast<const f32x4*>(in.shift + (size_t)n * Cin + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = fmaf(x[j], sc[j], sh[j]);
                }
                if (in.slope != 1.0f) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = x[j] > 0.f ? x[j] : x[j] * in.slope;
                }
                *reinterpret_cast<f32x4*>(xs + vox * P + c) = x;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // output rows of this lane's accumulator registers: voxel m = (r & 3) + 8 (r >> 2) + 4 lh of the tile -> fine base offset
        unsigned obase[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = v0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ix = v % in.W, iy = (v / in.W) % in.H, iz = v / (in.W * in.H);
            obase[r] = (unsigned)((((size_t)2 * iz * OH + 2 * iy) * OW + 2 * ix) * Cout);
        }
        float* outn = nm_eptr(out, (size_t)n * out_frame, OH16);
        for (int a = 0; a < 8; ++a) {
            const unsigned toff = (unsigned)(((size_t)(a >> 2) * OH + ((a >> 1) & 1)) * OW + (a & 1)) * Cout;
            const float* wa = w + (size_t)a * Cin * Cout;
            for (int co0 = 0; co0 < Cout; co0 += 32) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* wp = wa + co0 + l31;
                for (int k = 0; k < Cin; k += 8) {                 // (Cin % 8 == 0) four weight loads in flight per dependent MFMA chain link
                    float bb[4], aa[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { bb[u] = wp[(size_t)(k + 2 * u + lh) * Cout]; aa[u] = xs[l31 * P + k + 2 * u + lh]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u], bb[u], acc, 0, 0, 0);
                }
                const float bv = bias[co0 + l31];
                if constexpr (OH16) {
                    const bool odd = (l31 & 1) != 0;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float mine0 = acc[r] + bv, mine1 = acc[r + 1] + bv;
                        const float send = odd ? mine0 : mine1;
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                        const unsigned pk = odd ? nm_pk_bf16(recv, mine1) : nm_pk_bf16(mine0, recv);
                        const size_t e = (size_t)(odd ? obase[r + 1] : obase[r]) + toff + co0 + (l31 & ~1);
                        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(outn) + e) = pk;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) outn[(size_t)obase[r] + toff + co0 + l31] = acc[r] + bv;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The same op on the f16 matrix cores (conv modes 1 / 3 / 4): v_mfma_f32_32x32x16_f16 with the operands split into fp16 hi + lo * 2^-11
// and three products (SINGLE: hi only), the arithmetic of the convolutions (nm_conv.hip).  The fp32 form above runs Cin / 2 MFMAs of 64
// cycles per (tap, 32-channel tile) - 0.73 ms for the first pool's data gradient (64 -> 32 channels onto 64^3, 69 GFLOP), whose
// 1.07 GB of bfloat16 output take 0.2 ms to write; here it is Cin / 16 MFMAs of 32 cycles (x3 split).  The A operand needs 8
// consecutive input channels of one voxel per lane - channels-last memory as it is: every lane loads its own operand words straight
// from global memory (no LDS tile), applies the pending GroupNorm affine + LeakyReLU from a per-wave LDS table and keeps the Cin / 16
// packed operands in registers for all 8 taps.  The weights ([tap][ci][co] fp32) are converted once per workgroup into LDS planes
// [tap][co][ci] fp16 (row pitch + 16 B: conflict-free 16-B operand reads).  out_mul: device scalar the result is multiplied by (the
// inverse of the power-of-two scale a gradient input carries, DyScale).
typedef unsigned ct_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ct_half2 __attribute__((ext_vector_type(2)));
typedef float ct_f32x2 __attribute__((ext_vector_type(2)));
#define CT_SPLIT 2048.0f
__device__ __forceinline__ unsigned ct_pack_split(float v0, float v1, unsigned& lo_out) {
    ct_half2 hh = __builtin_convertvector(ct_f32x2{v0, v1}, ct_half2);
    asm volatile("" : "+v"(hh));
    ct_half2 ll;
    ll[0] = (_Float16)__builtin_fmaf((float)hh[0], -CT_SPLIT, v0 * CT_SPLIT);
    ll[1] = (_Float16)__builtin_fmaf((float)hh[1], -CT_SPLIT, v1 * CT_SPLIT);
    lo_out = __builtin_bit_cast(unsigned, ll);
    return __builtin_bit_cast(unsigned, hh);
}
#define CT_MAX_KS 8            // Cin <= 128
template <bool SINGLE, bool OH16, bool IH16>
__global__ __launch_bounds__(256) void convT2_f16_kernel(TensorRef in, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int Cout, int tiles_per_frame, int total_tiles,
                                                         const float* __restrict__ out_mul) {
    extern __shared__ char ct_lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, wv = tid >> 6;
    const int Cin = in.C, ks = Cin >> 4, WP = Cin * 2 + 16;
    char* W_hi = ct_lds; char* W_lo = W_hi + (size_t)8 * Cout * WP;
    float* tab = reinterpret_cast<float*>(ct_lds + (size_t)(SINGLE ? 1 : 2) * 8 * Cout * WP) + (size_t)wv * 2 * Cin;    // this wave's scale[Cin], shift[Cin]
    for (int i = tid; i < 8 * Cin * Cout; i += 256) {
        const int co = i % Cout, ci = (i / Cout) % Cin, a = i / (Cout * Cin);
        const float v = w[i];
        const _Float16 h = (_Float16)v;
        *reinterpret_cast<_Float16*>(W_hi + ((size_t)a * Cout + co) * WP + ci * 2) = h;
        if constexpr (!SINGLE) *reinterpret_cast<_Float16*>(W_lo + ((size_t)a * Cout + co) * WP + ci * 2) = (_Float16)__builtin_fmaf((float)h, -CT_SPLIT, v * CT_SPLIT);
    }
    __syncthreads();
    const float mul = out_mul ? *out_mul : 1.0f;
    const int OH = 2 * in.H, OW = 2 * in.W;
    const size_t out_frame = (size_t)8 * in.D * in.H * in.W * Cout, in_frame = (size_t)in.D * in.H * in.W * Cin;
    int n_tab = -1;
    for (int t = blockIdx.x * 4 + wv; t < total_tiles; t += gridDim.x * 4) {
        const int n = t / tiles_per_frame, v0 = (t % tiles_per_frame) * 32;
        if (in.scale && n != n_tab) {
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < Cin; c += 64) { tab[c] = in.scale[(size_t)n * Cin + c]; tab[Cin + c] = in.shift[(size_t)n * Cin + c]; }
            __builtin_amdgcn_wave_barrier();
            n_tab = n;
        }
        // this lane's operand words: voxel v0 + l31, channels 16 s + 8 lh .. + 8 for every k-step s
        ct_u32x4 ah[CT_MAX_KS], al[CT_MAX_KS];
        {
            const size_t e0 = (size_t)n * in_frame + (size_t)(v0 + l31) * Cin + 8 * lh;
            f32x4 r0[CT_MAX_KS], r1[CT_MAX_KS];
#pragma unroll
            for (int s = 0; s < CT_MAX_KS; ++s) {
                if (s < ks) {
                    if constexpr (IH16) {
                        const ct_u32x4 q = *reinterpret_cast<const ct_u32x4*>(reinterpret_cast<const unsigned short*>(in.p) + e0 + 16 * s);
                        r0[s] = f32x4{nm_bf_lo(q[0]), nm_bf_hi(q[0]), nm_bf_lo(q[1]), nm_bf_hi(q[1])};
                        r1[s] = f32x4{nm_bf_lo(q[2]), nm_bf_hi(q[2]), nm_bf_lo(q[3]), nm_bf_hi(q[3])};
                    } else {
                        r0[s] = *reinterpret_cast<const f32x4*>(in.p + e0 + 16 * s);
                        r1[s] = *reinterpret_cast<const f32x4*>(in.p + e0 + 16 * s + 4);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < CT_MAX_KS; ++s) {
                if (s < ks) {
                    f32x4 x0 = r0[s], x1 = r1[s];
                    if (in.scale) {
                        const int c = 16 * s + 8 * lh;
                        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(tab + c), sc1 = *reinterpret_cast<const f32x4*>(tab + c + 4);
                        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(tab + Cin + c), sh1 = *reinterpret_cast<const f32x4*>(tab + Cin + c + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { x0[j] = fmaf(x0[j], sc0[j], sh0[j]); x1[j] = fmaf(x1[j], sc1[j], sh1[j]); }
                    }
                    if (in.slope != 1.0f) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { x0[j] = x0[j] > 0.f ? x0[j] : x0[j] * in.slope; x1[j] = x1[j] > 0.f ? x1[j] : x1[j] * in.slope; }
                    }
                    unsigned l0, l1, l2, l3;
                    const unsigned h0 = ct_pack_split(x0[0], x0[1], l0), h1 = ct_pack_split(x0[2], x0[3], l1);
                    const unsigned h2 = ct_pack_split(x1[0], x1[1], l2), h3 = ct_pack_split(x1[2], x1[3], l3);
                    ah[s] = ct_u32x4{h0, h1, h2, h3}; al[s] = ct_u32x4{l0, l1, l2, l3};
                }
            }
        }
        unsigned obase[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = v0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ix = v % in.W, iy = (v / in.W) % in.H, iz = v / (in.W * in.H);
            obase[r] = (unsigned)((((size_t)2 * iz * OH + 2 * iy) * OW + 2 * ix) * Cout);
        }
        float* outn = nm_eptr(out, (size_t)n * out_frame, OH16);
        for (int a = 0; a < 8; ++a) {
            const unsigned toff = (unsigned)(((size_t)(a >> 2) * OH + ((a >> 1) & 1)) * OW + (a & 1)) * Cout;
            for (int co0 = 0; co0 < Cout; co0 += 32) {
                f32x16 acc, accl;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }
                const char* bh_p = W_hi + ((size_t)a * Cout + co0 + l31) * WP + 16 * lh;
                const char* bl_p = W_lo + ((size_t)a * Cout + co0 + l31) * WP + 16 * lh;
#pragma unroll
                for (int s = 0; s < CT_MAX_KS; ++s) {
                    if (s < ks) {
                        const nm_half8 bh = *reinterpret_cast<const nm_half8*>(bh_p + 32 * s);
                        const nm_half8 xa = __builtin_bit_cast(nm_half8, ah[s]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, bh, acc, 0, 0, 0);
                        if constexpr (!SINGLE) {
                            const nm_half8 bl = *reinterpret_cast<const nm_half8*>(bl_p + 32 * s);
                            accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, bl, accl, 0, 0, 0);
                            accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(nm_half8, al[s]), bh, accl, 0, 0, 0);
                        }
                    }
                }
                const float bv = bias[co0 + l31];
                auto fin = [&](int r) __attribute__((always_inline)) {
                    float v = acc[r];
                    if constexpr (!SINGLE) v += accl[r] * (1.0f / CT_SPLIT);
                    return v * mul + bv;
                };
                if constexpr (OH16) {
                    const bool odd = (l31 & 1) != 0;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float mine0 = fin(r), mine1 = fin(r + 1);
                        const float send = odd ? mine0 : mine1;
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                        const unsigned pk = odd ? nm_pk_bf16(recv, mine1) : nm_pk_bf16(mine0, recv);
                        const size_t e = (size_t)(odd ? obase[r + 1] : obase[r]) + toff + co0 + (l31 & ~1);
                        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(outn) + e) = pk;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) outn[(size_t)obase[r] + toff + co0 + l31] = fin(r);
                }
            }
        }
    }
}

// trilinear x2, align_corners=False: src = (dst + 0.5) / 2 - 0.5 clamped at 0
__device__ __forceinline__ void up_idx(int o, int I, int& i0, int& i1, float& l1) {
    float src = 0.5f * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    i1 = i0 + (i0 < I - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// One thread per (coarse cell, 4 channels): the cell's 2x2x2 fine children need the 3x3x3 clamped coarse neighbourhood (27 activated
// loads instead of 8 per fine voxel = 64 per cell), interpolated separably x -> y -> z.  Fine index 2i+a along an axis:
// a = 0: 0.25 c[i-1] + 0.75 c[i], a = 1: 0.75 c[i] + 0.25 c[i+1], neighbour indices clamped to the volume - exactly the
// align_corners=False weights (src = (dst + 0.5)/2 - 0.5 clamped at 0; the upper clamp folds c[I] onto c[I-1]).
__global__ __launch_bounds__(256) void upsample2_kernel(TensorRef in, float* __restrict__ out, int oh) {
    const int D = in.D, H = in.H, W = in.W, cq = in.C / 4;
    const size_t total = (size_t)in.N * D * H * W * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int q = (int)(i % cq); size_t r = i / cq;
        int x = (int)(r % W); r /= W;
        int y = (int)(r % H); r /= H;
        int z = (int)(r % D); size_t n = r / D;
        const int c = q * 4;
        const int zs[3] = {max(z - 1, 0), z, min(z + 1, D - 1)}, ys[3] = {max(y - 1, 0), y, min(y + 1, H - 1)};
        const int xs[3] = {max(x - 1, 0), x, min(x + 1, W - 1)};
        f32x4 fx[3][3][2];                       // [z][y][child x]
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const size_t rowo = ((size_t)zs[a] * H + ys[b]) * W;
                const f32x4 v0 = load_t(in, n, (rowo + xs[0]) * in.C + c, c), v1 = load_t(in, n, (rowo + xs[1]) * in.C + c, c),
                            v2 = load_t(in, n, (rowo + xs[2]) * in.C + c, c);
                fx[a][b][0] = 0.25f * v0 + 0.75f * v1; fx[a][b][1] = 0.75f * v1 + 0.25f * v2;
            }
        const int OH = 2 * H, OW = 2 * W;
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
            f32x4 fy[3][2];                      // [z][child y]
#pragma unroll
            for (int a = 0; a < 3; ++a) { fy[a][0] = 0.25f * fx[a][0][cx] + 0.75f * fx[a][1][cx]; fy[a][1] = 0.75f * fx[a][1][cx] + 0.25f * fx[a][2][cx]; }
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
                const f32x4 o0 = 0.25f * fy[0][cy] + 0.75f * fy[1][cy], o1 = 0.75f * fy[1][cy] + 0.25f * fy[2][cy];
                const size_t base = ((((size_t)n * 2 * D + 2 * z) * OH + 2 * y + cy) * OW + 2 * x + cx) * in.C + c;
                nm_st4(out, base, o0, oh);
                nm_st4(out, base + (size_t)OH * OW * in.C, o1, oh);
            }
        }
    }
}

// The same with the 3x3x3 neighbourhoods shared through LDS: a block owns 2 x 4 x 4 coarse cells and 32 channels; the clamped
// 4 x 6 x 6 coarse halo is loaded and activated once (4.5 loads per cell instead of 27), every thread then interpolates its cell's
// eight children from LDS and writes 128-byte voxel rows.  Used when C % 32 == 0 and the extents are multiples of the tile.
template <bool IH, bool OH>
__global__ __launch_bounds__(256) void upsample2_tile_kernel(TensorRef in, float* __restrict__ out, int tz, int ty, int tx) {
    __shared__ f32x4 tile[4 * 6 * 6 * 8];          // [hz][hy][hx][quad]
    const int D = in.D, H = in.H, W = in.W, chunks = in.C / 32;
    int b = blockIdx.x;
    const int ch = b % chunks; b /= chunks;
    const int bx = b % tx; b /= tx;
    const int by = b % ty; b /= ty;
    const int bz = b % tz; const size_t n = b / tz;
    const int z0 = bz * 2, y0 = by * 4, x0 = bx * 4, c0 = ch * 32;
    for (int i = threadIdx.x; i < 4 * 6 * 6 * 8; i += 256) {
        const int q = i & 7; int v = i >> 3;
        const int hx = v % 6, hy = (v / 6) % 6, hz = v / 36;
        const int gz = min(max(z0 - 1 + hz, 0), D - 1), gy = min(max(y0 - 1 + hy, 0), H - 1), gx = min(max(x0 - 1 + hx, 0), W - 1);
        tile[i] = load_t<IH ? 1 : 0>(in, n, (((size_t)gz * H + gy) * W + gx) * in.C + c0 + 4 * q, c0 + 4 * q);
    }
    __syncthreads();
    const int q = threadIdx.x & 7, cell = threadIdx.x >> 3;
    const int cx = cell & 3, cy = (cell >> 2) & 3, cz = cell >> 4;
    f32x4 fx[3][3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            const f32x4* row = tile + (((cz + a) * 6 + (cy + bb)) * 6 + cx) * 8 + q;
            const f32x4 v0 = row[0], v1 = row[8], v2 = row[16];
            fx[a][bb][0] = 0.25f * v0 + 0.75f * v1; fx[a][bb][1] = 0.75f * v1 + 0.25f * v2;
        }
    const int OHt = 2 * H, OW = 2 * W;
    const int z = z0 + cz, y = y0 + cy, x = x0 + cx;
#pragma unroll
    for (int kx = 0; kx < 2; ++kx) {
        f32x4 fy[3][2];
#pragma unroll
        for (int a = 0; a < 3; ++a) { fy[a][0] = 0.25f * fx[a][0][kx] + 0.75f * fx[a][1][kx]; fy[a][1] = 0.75f * fx[a][1][kx] + 0.25f * fx[a][2][kx]; }
#pragma unroll
        for (int ky = 0; ky < 2; ++ky) {
            const f32x4 o0 = 0.25f * fy[0][ky] + 0.75f * fy[1][ky], o1 = 0.75f * fy[1][ky] + 0.25f * fy[2][ky];
            const size_t base = ((((size_t)n * 2 * D + 2 * z) * OHt + 2 * y + ky) * OW + 2 * x + kx) * in.C + c0 + 4 * q;
            nm_st4<OH>(out, base, o0);
            nm_st4<OH>(out, base + (size_t)OHt * OW * in.C, o1);
        }
    }
}

__device__ __forceinline__ float lin_coord(int i, int G) {
    // torch.linspace(-1, 1, G) in fp32: symmetric evaluation from both ends, one rounding (fma)
    const float step = 2.0f / (float)(G - 1);
    return i < G / 2 ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(G - 1 - i), 1.0f);
}

__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ vox, int B, int T, int G, int mean_t,
                                                         float* __restrict__ out) {
    const size_t G3 = (size_t)G * G * G;
    const int frames = mean_t ? B : B * T;
    const size_t total = (size_t)frames * G3;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t f = i / G3, v = i % G3;
        float occ;
        if (mean_t) {
            float s = 0.f;
            for (int t = 0; t < T; ++t) s += vox[(f * T + t) * G3 + v];
            occ = s / (float)T;
        } else {
            occ = vox[f * G3 + v];
        }
        int x = (int)(v % G), y = (int)((v / G) % G), z = (int)(v / ((size_t)G * G));
        f32x4 a = {occ, lin_coord(z, G), lin_coord(y, G), lin_coord(x, G)};
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        f32x4* o = reinterpret_cast<f32x4*>(out + i * 8);
        o[0] = a; o[1] = b;
    }
}

// [n][vox][C] -> [n][C][vox], 32x32 tiles through LDS
__global__ __launch_bounds__(256) void cl_to_ncdhw_kernel(TensorRef in, int frame_stride, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int voxels = in.D * in.H * in.W;
    const int n = blockIdx.z, v0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const size_t nin = (size_t)n * frame_stride;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int v = v0 + j, c = c0 + tx;
        float val = 0.f;
        if (v < voxels && c < in.C) {
            val = in.p[(nin * voxels + v) * in.C + c];
            if (in.scale) val = val * in.scale[nin * in.C + c] + in.shift[nin * in.C + c];
            if (in.slope != 1.0f) val = lrelu(val, in.slope);
        }
        tile[j][tx] = val;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, v = v0 + tx;
        if (v < voxels && c < in.C) out[((size_t)n * in.C + c) * voxels + v] = tile[tx][j];
    }
}

// clip mean over T (kypt_detector.py:312): sequential sum then one division, like the reference's mean(dim=1)
__global__ __launch_bounds__(256) void mean_t_kernel(const float* __restrict__ vox, int T, size_t G3, size_t total, float* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / G3, v = i % G3;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += vox[(b * T + t) * G3 + v];
        out[i] = s / (float)T;
    }
}

// (Cin, Cout, 2,2,2) IODHW -> [tap][Cin][Cout]
__global__ void transpose_convT_weight_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * Cout * 8) return;
    int co = i % Cout, ci = (i / Cout) % Cin, tap = i / (Cout * Cin);
    out[i] = w[((size_t)ci * Cout + co) * 8 + tap];
}

// [n][C][vox] -> [n][vox][C]
__global__ __launch_bounds__(256) void ncdhw_to_cl_kernel(const float* __restrict__ in, int voxels, int C, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, v0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, v = v0 + tx;
        tile[j][tx] = (v < voxels && c < C) ? in[((size_t)n * C + c) * voxels + v] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int v = v0 + j, c = c0 + tx;
        if (v < voxels && c < C) out[((size_t)n * voxels + v) * C + c] = tile[tx][j];
    }
}

int grid_for(size_t work_items) { return (int)min((work_items + 255) / 256, (size_t)(256 * 16)); }

}  // namespace

void nm_elem_set_nonfinite_flag(unsigned* flag) { nm_ls().nf_flag = flag; }

__global__ __launch_bounds__(256) void nonfinite_scan_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ flag) {
    bool bad = false;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad |= !isfinite(x[i]);
    if (bad) atomicOr(flag, 1u);
}
int nm_launch_nonfinite_scan(const float* x, size_t n, unsigned* flag, hipStream_t s) {
    hipLaunchKernelGGL(nonfinite_scan_kernel, dim3((unsigned)min((n + 255) / 256, (size_t)4096)), dim3(256), 0, s, x, n, flag);
    return nm_check_hip(hipGetLastError(), "nonfinite_scan launch");
}

bool nm_gn_finalize_has_chsum(int C, int groups) { return groups > 0 && C % groups == 0 && 256 % (C / groups) == 0; }

int nm_launch_gn_finalize(const float* part, int N, int nblk, int C, int groups, double count, const float* gamma,
                          const float* beta, float eps, float* scale, float* shift, hipStream_t s, double* chsum) {
    if (groups <= 0 || C % groups != 0 || C / groups > 256) { nm_set_error("gn_finalize: bad groups %d for C=%d", groups, C); return NM_ERR_ARG; }
    if (chsum && !nm_gn_finalize_has_chsum(C, groups)) { nm_set_error("gn_finalize: no per-channel sums for %d channels per group", C / groups); return NM_ERR_ARG; }
    if ((long long)nblk * (C / groups) > 8192)
        hipLaunchKernelGGL(gn_finalize_kernel<1024>, dim3(N * groups), dim3(1024), 0, s, part, nblk, C, groups, count, gamma, beta, eps, scale, shift, nm_ls().nf_flag, chsum);
    else
        hipLaunchKernelGGL(gn_finalize_kernel<256>, dim3(N * groups), dim3(256), 0, s, part, nblk, C, groups, count, gamma, beta, eps, scale, shift, nm_ls().nf_flag, chsum);
    return nm_check_hip(hipGetLastError(), "gn_finalize launch");
}

int nm_launch_gn_direct(const float* x, int N, int voxels, int C, int groups, const float* gamma, const float* beta, float eps,
                        float* scale, float* shift, hipStream_t s, double* chsum) {
    if (groups <= 0 || C % groups || C / groups > 1024) { nm_set_error("gn_direct: unsupported channels"); return NM_ERR_ARG; }
    hipLaunchKernelGGL(gn_direct_kernel, dim3(N * groups), dim3(1024), 0, s, x, voxels, C, groups, gamma, beta, eps, scale, shift, chsum, nm_ls().gn_diag);
    return nm_check_hip(hipGetLastError(), "gn_direct launch");
}

int nm_stats_blocks_per_frame(int voxels) { return (voxels + NM_STATS_VB - 1) / NM_STATS_VB; }

int nm_launch_gn_partials(const float* x, int N, int voxels, int C, float* part, hipStream_t s, int h) {
    if (C > 256 || C <= 0) { nm_set_error("gn_partials: C=%d unsupported", C); return NM_ERR_ARG; }
    int nblk = nm_stats_blocks_per_frame(voxels);
    hipLaunchKernelGGL(gn_partials_kernel, dim3(N * nblk), dim3(256), 0, s, x, voxels, C, nblk, part, h);
    return nm_check_hip(hipGetLastError(), "gn_partials launch");
}

int nm_launch_apply2(const TensorRef& a, const TensorRef* b, float* out, hipStream_t s, int out_h) {
    if (a.C % 4) { nm_set_error("apply2: C %% 4 != 0"); return NM_ERR_ARG; }
    if (b && (b->N != a.N || b->D != a.D || b->H != a.H || b->W != a.W || b->C != a.C)) { nm_set_error("apply2: shape mismatch"); return NM_ERR_ARG; }
    const size_t frame4 = (size_t)a.D * a.H * a.W * a.C / 4;
    if (frame4 * 4 >= ((size_t)1 << 31)) { nm_set_error("apply2: frame too large"); return NM_ERR_ARG; }
    TensorRef bb = b ? *b : a;
    const unsigned bx = (unsigned)min((frame4 + 255) / 256, (size_t)max(1, 4096 / max(a.N, 1)));
    hipLaunchKernelGGL(apply2_kernel, dim3(bx, a.N), dim3(256), 0, s, a, bb, b ? 1 : 0, out, out_h);
    return nm_check_hip(hipGetLastError(), "apply2 launch");
}

static size_t convT2_f16_lds(int Cin, int Cout, bool single) { return (size_t)(single ? 1 : 2) * 8 * Cout * (Cin * 2 + 16) + (size_t)4 * 2 * Cin * sizeof(float); }
bool nm_convT2_f16_eligible(const TensorRef& in, int Cout, int OD, int OH, int OW, int out_h) {
    // (a per-FRAME size rule: which arithmetic a layer gets must not depend on the batch size - batch additivity of the gradients)
    const size_t fvox = (size_t)in.D * in.H * in.W;
    return nm_ls().convt_f16 && nm_conv_get_mode() != 0 && OD == 2 * in.D && OH == 2 * in.H && OW == 2 * in.W && Cout % 32 == 0 && in.C <= 128 && in.C % 16 == 0 &&
           fvox % 32 == 0 && fvox >= 512 && fvox * 8 * Cout < ((size_t)1 << 31) && (!(in.h || out_h) || nm_conv_single()) &&
           convT2_f16_lds(in.C, Cout, nm_conv_single() != 0) <= 150 * 1024;
}
template <bool SINGLE, bool OH16, bool IH16>
static int launch_convT2_f16(const TensorRef& in, const float* w, const float* bias, float* out, int Cout, int tpf, int tiles, size_t ldsb,
                             hipStream_t s, const float* out_mul) {
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convT2_f16_kernel<SINGLE, OH16, IH16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(convT2_f16)");
        attr_set.mark();
    }
    // persistent workgroups (each converts the weights into its LDS once): as many as fit the chip, at most one per four tiles
    const int per_cu = ldsb <= 40 * 1024 ? 3 : (ldsb <= 76 * 1024 ? 2 : 1);
    const dim3 g((unsigned)min((tiles + 3) / 4, 256 * per_cu));
    hipLaunchKernelGGL((convT2_f16_kernel<SINGLE, OH16, IH16>), g, dim3(256), ldsb, s, in, w, bias, out, Cout, tpf, tiles, out_mul);
    return nm_check_hip(hipGetLastError(), "convT2_f16 launch");
}

int nm_launch_convT2(const TensorRef& in, const float* w, const float* bias, float* out, int Cout, int OD, int OH,
                     int OW, hipStream_t s, int out_h, const float* out_mul) {
    if (Cout % 4 || in.C % 4) { nm_set_error("convT2: channels must be multiples of 4"); return NM_ERR_ARG; }
    if (OD < 2 * in.D || OD > 2 * in.D + 1 || OH < 2 * in.H || OH > 2 * in.H + 1 || OW < 2 * in.W || OW > 2 * in.W + 1) {
        nm_set_error("convT2: bad output size"); return NM_ERR_ARG;
    }
    size_t total = (size_t)in.N * OD * OH * OW * (Cout / 4);
    const size_t cvox = (size_t)in.N * in.D * in.H * in.W;
    const size_t fvox = (size_t)in.D * in.H * in.W;
    if (nm_convT2_f16_eligible(in, Cout, OD, OH, OW, out_h)) {
        const bool single = nm_conv_single();
        const size_t ldsb = convT2_f16_lds(in.C, Cout, single);
        {
            const int tpf = (int)(fvox / 32), tiles = (int)(cvox / 32);
            if (!single) return launch_convT2_f16<false, false, false>(in, w, bias, out, Cout, tpf, tiles, ldsb, s, out_mul);
            if (out_h && in.h) return launch_convT2_f16<true, true, true>(in, w, bias, out, Cout, tpf, tiles, ldsb, s, out_mul);
            if (out_h) return launch_convT2_f16<true, true, false>(in, w, bias, out, Cout, tpf, tiles, ldsb, s, out_mul);
            if (in.h) return launch_convT2_f16<true, false, true>(in, w, bias, out, Cout, tpf, tiles, ldsb, s, out_mul);
            return launch_convT2_f16<true, false, false>(in, w, bias, out, Cout, tpf, tiles, ldsb, s, out_mul);
        }
    }
    if (out_mul) { nm_set_error("convT2: an output multiplier exists on the f16 matrix-core kernel only (conv mode != 0, Cin %% 16 == 0, Cin <= 128, Cout %% 32 == 0)"); return NM_ERR_UNSUPPORTED; }
    if (OD == 2 * in.D && OH == 2 * in.H && OW == 2 * in.W && Cout % 32 == 0 && in.C <= 128 && in.C % 8 == 0 && fvox % 32 == 0 && cvox >= 4096 &&
        fvox * 8 * Cout < ((size_t)1 << 31)) {
        static NmDeviceOnce attr_set;
        if (!attr_set.done()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convT2_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convT2_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convT2_mfma_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convT2_mfma_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(convT2_mfma)");
            attr_set.mark();
        }
        const int tpf = (int)(fvox / 32), tiles = (int)(cvox / 32);
        const size_t ldsb = (size_t)4 * 32 * (in.C + 4) * sizeof(float);
        const dim3 gm((unsigned)min((tiles + 3) / 4, 2048));
        if (out_h && in.h) hipLaunchKernelGGL((convT2_mfma_kernel<true, true>), gm, dim3(256), ldsb, s, in, w, bias, out, Cout, tpf, tiles);
        else if (out_h) hipLaunchKernelGGL((convT2_mfma_kernel<true, false>), gm, dim3(256), ldsb, s, in, w, bias, out, Cout, tpf, tiles);
        else if (in.h) hipLaunchKernelGGL((convT2_mfma_kernel<false, true>), gm, dim3(256), ldsb, s, in, w, bias, out, Cout, tpf, tiles);
        else hipLaunchKernelGGL((convT2_mfma_kernel<false, false>), gm, dim3(256), ldsb, s, in, w, bias, out, Cout, tpf, tiles);
        return nm_check_hip(hipGetLastError(), "convT2_mfma launch");
    }
    if (OD == 2 * in.D && OH == 2 * in.H && OW == 2 * in.W && 256 % (Cout / 4) == 0 && (size_t)in.C * Cout * 4 <= 48 * 1024 && cvox >= 65536) {
        const int vpb = 2048;
        hipLaunchKernelGGL(convT2_lds_kernel, dim3((unsigned)((cvox + vpb - 1) / vpb), 8), dim3(256), (size_t)in.C * Cout * sizeof(float), s, in, w, bias,
                           out, Cout, OD, OH, OW, vpb, out_h);
        return nm_check_hip(hipGetLastError(), "convT2_lds launch");
    }
    hipLaunchKernelGGL(convT2_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, w, bias, out, Cout, OD, OH, OW, out_h);
    return nm_check_hip(hipGetLastError(), "convT2 launch");
}

int nm_launch_upsample2(const TensorRef& in, float* out, hipStream_t s, int out_h) {
    if (in.C % 4) { nm_set_error("upsample2: C %% 4 != 0"); return NM_ERR_ARG; }
    if (in.C % 32 == 0 && in.D % 2 == 0 && in.H % 4 == 0 && in.W % 4 == 0) {
        const int tz = in.D / 2, ty = in.H / 4, tx = in.W / 4;
        const dim3 g((unsigned)((size_t)in.N * tz * ty * tx * (in.C / 32)));
        if (in.h && out_h) hipLaunchKernelGGL((upsample2_tile_kernel<true, true>), g, dim3(256), 0, s, in, out, tz, ty, tx);
        else if (in.h) hipLaunchKernelGGL((upsample2_tile_kernel<true, false>), g, dim3(256), 0, s, in, out, tz, ty, tx);
        else if (out_h) hipLaunchKernelGGL((upsample2_tile_kernel<false, true>), g, dim3(256), 0, s, in, out, tz, ty, tx);
        else hipLaunchKernelGGL((upsample2_tile_kernel<false, false>), g, dim3(256), 0, s, in, out, tz, ty, tx);
        return nm_check_hip(hipGetLastError(), "upsample2 launch");
    }
    size_t total = (size_t)in.N * in.D * in.H * in.W * (in.C / 4);
    hipLaunchKernelGGL(upsample2_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, out_h);
    return nm_check_hip(hipGetLastError(), "upsample2 launch");
}

int nm_launch_pack_input(const float* vox, int B, int T, int G, int mean_over_t, float* out, hipStream_t s) {
    size_t total = (size_t)(mean_over_t ? B : B * T) * G * G * G;
    hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(total)), dim3(256), 0, s, vox, B, T, G, mean_over_t, out);
    return nm_check_hip(hipGetLastError(), "pack_input launch");
}

int nm_launch_cl_to_ncdhw_strided(const TensorRef& in, int frame_stride, float* out, hipStream_t s) {
    if (in.h) { nm_set_error("cl_to_ncdhw: bfloat16 input is not supported"); return NM_ERR_UNSUPPORTED; }
    int voxels = in.D * in.H * in.W;
    dim3 grid((voxels + 31) / 32, (in.C + 31) / 32, in.N);
    hipLaunchKernelGGL(cl_to_ncdhw_kernel, grid, dim3(256), 0, s, in, frame_stride, out);
    return nm_check_hip(hipGetLastError(), "cl_to_ncdhw launch");
}

int nm_launch_cl_to_ncdhw(const TensorRef& in, float* out, hipStream_t s) { return nm_launch_cl_to_ncdhw_strided(in, 1, out, s); }

int nm_launch_ncdhw_to_cl(const float* in, int N, int voxels, int C, float* out, hipStream_t s) {
    dim3 grid((voxels + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(ncdhw_to_cl_kernel, grid, dim3(256), 0, s, in, voxels, C, out);
    return nm_check_hip(hipGetLastError(), "ncdhw_to_cl launch");
}

int nm_launch_mean_t(const float* vox, int B, int T, size_t G3, float* out, hipStream_t s) {
    size_t total = (size_t)B * G3;
    hipLaunchKernelGGL(mean_t_kernel, dim3(grid_for(total)), dim3(256), 0, s, vox, T, G3, total, out);
    return nm_check_hip(hipGetLastError(), "mean_t launch");
}

int nm_launch_transpose_convT_weight(const float* w_iodhw, int Cin, int Cout, float* out, hipStream_t s) {
    int total = Cin * Cout * 8;
    hipLaunchKernelGGL(transpose_convT_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_iodhw, Cin, Cout, out);
    return nm_check_hip(hipGetLastError(), "transpose_convT_weight launch");
}
