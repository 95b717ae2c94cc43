// The two lowest levels of the hourglass (modules/vox_modules.py:78-120) in ONE launch: one workgroup per frame, activations in LDS.
//
//   a2 = act(GN(pool2(e1)))            (computed outside, handed over lazily)         level 2: D2^3 voxels, 32 channels
//   e2 = encoder_res2(a2)              32 -> 48
//   s3 = skip_res3(e2)                 48 -> 48
//   p3 = act(GN(pool3(e2)))            k2 s2, level 3: D3^3 voxels
//   e3 = encoder_res3(p3)              48 -> 72
//   d3 = decoder_res3(e3)              72 -> 72
//   x  = act(GN(convT(d3))) + s3       ConvTranspose3d k2 s2 (+ output_padding), back on level 2
//   out = decoder_res2(x)              48 -> 48          -> global, plain fp32
//
// As separate launches this is 13 convs + 14 GroupNorm finalisations + 6 adds per feature net on 64 (4^3) or 8 (2^3) voxels per
// frame: every launch a few workgroups, ~20-35 us each, one after the other (~0.6 ms of a 19 ms forward step on the frame net's
// stream).  A frame's tensors at these levels are 12 KB or less (41 KB at the 96^3 grid), GroupNorm is per frame, so a workgroup
// owns its frame end to end: workgroup barriers instead of kernel boundaries, statistics over data that never leaves LDS.
//
// Arithmetic = the library's fp32-equivalent convolution: every product of the k1 / k2 / k3 convs as three f16 MFMAs on operands
// split v = hi + lo * 2^-11 (v_mfma_f32_32x32x16_f16, fp32 accumulate, same weight packs as conv_mfma_kernel's small-volume core);
// the transposed conv and the GroupNorm arithmetic in fp32 VALU (statistics two-pass, accumulated in fp64).  Inference only (the
// training forward keeps every layer's output for the backward pass).
#include "nm_ctx.h"
#include "nm_hgcore.h"
#include <cstdlib>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define HG_SPLIT 2048.0f
#define HG_THREADS 512
#define HG_MAX_ITEMS 16            // (pair, K part) items of a conv: 4 KB of LDS scratch each
__device__ int g_hg_diag = 0;          // NM355_HG_DIAG: phase ablations (wrong results, timing only); set by nm_launch_hg_core
// Address spaces, explicitly: behind a (non-inlined) function boundary a pointer is generic, every access a FLAT instruction that
// counts on both the LDS and the vector-memory counter - the LDS operand reads of a k-step then wait for the weight prefetches of
// the following ones (measured: one L2 round trip per k-step, ~1 ms per launch).
#define HG_LDS __attribute__((address_space(3)))
#define HG_GLB __attribute__((address_space(1)))
typedef HG_LDS float lds_float;
typedef HG_LDS f32x4 lds_f32x4;
typedef const HG_GLB half8 glb_half8;
typedef const HG_GLB float glb_float;

__device__ __forceinline__ void hg_split8(const f32x4& a, const f32x4& b, half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v0 = (j < 2) ? a[2 * j] : b[2 * j - 4], v1 = (j < 2) ? a[2 * j + 1] : b[2 * j - 3];
        const half2v hh = __builtin_convertvector(f32x2{v0, v1}, half2v);
        hi[2 * j] = hh[0]; hi[2 * j + 1] = hh[1];
        lo[2 * j] = (_Float16)__builtin_fmaf((float)hh[0], -HG_SPLIT, v0 * HG_SPLIT);
        lo[2 * j + 1] = (_Float16)__builtin_fmaf((float)hh[1], -HG_SPLIT, v1 * HG_SPLIT);
    }
}

__device__ __forceinline__ float hg_lrelu(float v, float slope) { return fmaxf(v, v * slope); }

// conv ks^3 (stride, pad) from an LDS tensor src [Vin][sp] (fp32, activated, channels padded with zeros to a multiple of 16) into
// dst [Vout][dp] (raw output + bias; channels Cout .. dp-1 zeroed).  Implicit GEMM: rows = output voxels (32 per MFMA tile),
// columns = output channels (32 per tile), K = (tap, 16-channel chunk); the (row tile, column tile) pairs are dealt to the 4 waves.
// Weights: [tap][chunk][hi h0 | hi h1 | lo h0 | lo h1][Co_pad] half8 (the pack of nm_launch_pack_conv_weight16), fetched PF k-steps
// ahead (an L2 round trip is ~10 k-steps of MFMA work for one wave).
#define HG_PF 8
// Round 5: 512 threads per workgroup (8 waves, two per SIMD - one workgroup per CU either way, and a launch is 64 frames on 256 CUs;
// 1024 threads leave 128 registers per lane, which this k-loop spills).
// A conv has only mtiles x ntiles = 2 .. 4 (row tile, column tile) pairs - one or none per wave with four waves - so its K range
// (tap, 16-channel chunk) is cut into KS = waves / pairs parts: item (pair, part) accumulates its k-steps and parks the 32 x 32 partial
// tile in LDS (`scratch`, 4 KB per item), the workgroup then sums the KS partials of every output in part order and adds the bias.
// Two waves per SIMD hide part of each other's operand / weight latency: the weight prefetch is 8 k-steps deep instead of 12 (registers).
// `scratch` holds `sitems_` items (nm_hg_core_scratch_items: as many of HG_MAX_ITEMS as the frame's tensors leave room for).  A conv with
// more (row tile, column tile) pairs than that (the 96^3 grid's 6^3 level: 14 pairs beside 153 KB of tensors) takes no K split and
// stores its finished tiles straight from the accumulators (`direct`): no scratch at all.
__device__ void conv_lds(const float* src, int sp_, int Din_, float* dst, int dp_, int Dout_, const NmHgConv& L, int stride_, int pad_, const float* zeros,
                         float* scratch, int sitems_) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
    const int nwaves = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
    // everything that drives the loop structure in SGPRs (the layer record arrives through memory: without this the tap decode
    // and the loop bounds are per-lane VALU arithmetic - measured 4x the MFMA time of a k-step)
    const int sp = __builtin_amdgcn_readfirstlane(sp_), Din = __builtin_amdgcn_readfirstlane(Din_), dp = __builtin_amdgcn_readfirstlane(dp_);
    const int Dout = __builtin_amdgcn_readfirstlane(Dout_), stride = __builtin_amdgcn_readfirstlane(stride_), pad = __builtin_amdgcn_readfirstlane(pad_);
    const int ks = __builtin_amdgcn_readfirstlane(L.ks), Cin = __builtin_amdgcn_readfirstlane(L.Cin), Cout = __builtin_amdgcn_readfirstlane(L.Cout);
    const int Co_pad = __builtin_amdgcn_readfirstlane(L.Co_pad);
    const int Vout = Dout * Dout * Dout, mtiles = (Vout + 31) >> 5, ntiles = Co_pad >> 5;
    const int nchunk = (Cin + 15) >> 4, nk = (g_hg_diag & 2) ? 1 : ks * ks * ks * nchunk;          // (diagnostic bit 2: one k-step per conv - timing only)
    const int pairs = mtiles * ntiles;
    const int sitems = __builtin_amdgcn_readfirstlane(sitems_);
    const bool direct = sitems < pairs;                                                            // (uniform)
    const int KS = direct ? 1 : max(1, min(min(nwaves / pairs, sitems / pairs), nk));              // K parts per pair; pairs * KS <= sitems
    glb_half8* w8 = (glb_half8*)L.w16;
    const lds_float* srcl = (const lds_float*)src;
    lds_float* dstl = (lds_float*)dst;
    lds_float* scr = (lds_float*)scratch;
    const lds_float* zl = (const lds_float*)zeros;
    glb_float* biasg = (glb_float*)L.bias;
    const size_t plane = (size_t)Co_pad;
    for (int it = wave; it < pairs * KS; it += nwaves) {
        const int pr = it / KS, kp = it % KS;
        const int mt = pr / ntiles, nt = pr % ntiles;
        const int kbeg = (int)((long long)nk * kp / KS), kend = (int)((long long)nk * (kp + 1) / KS), nkl = kend - kbeg;
        const int m = mt * 32 + l31;
        const bool rowok = m < Vout;
        const int ox = m % Dout, oy = (m / Dout) % Dout, oz = m / (Dout * Dout);
        const int iz0 = oz * stride - pad, iy0 = oy * stride - pad, ix0 = ox * stride - pad;
        // per-lane: base offset of the row's tap (0,0,0) and, per axis, which of the <= 3 tap positions fall inside the volume
        const int off0 = ((iz0 * Din + iy0) * Din + ix0) * sp + 8 * h;
        int mz = 0, my = 0, mx = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            mz |= ((unsigned)(iz0 + t) < (unsigned)Din) << t; my |= ((unsigned)(iy0 + t) < (unsigned)Din) << t; mx |= ((unsigned)(ix0 + t) < (unsigned)Din) << t;
        }
        if (!rowok) mz = 0;
        f32x16 acc, accl, accm;            // hi x hi, hi x lo, lo x hi: three independent accumulation chains
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; accm[r] = 0.f; }
        glb_half8* wl = w8 + (size_t)h * plane + nt * 32 + l31 + (size_t)kbeg * 4 * plane;   // + k * 4 * plane per k-step (tap-major, chunk-minor)
        half8 bh[HG_PF], bl[HG_PF];
#pragma unroll
        for (int u = 0; u < HG_PF; ++u) {
            const int k = u < nkl ? u : nkl - 1;
            bh[u] = wl[(size_t)k * 4 * plane]; bl[u] = wl[(size_t)k * 4 * plane + 2 * plane];
        }
        // branch-free k-loop (a load or an MFMA under a condition makes hipcc drain every outstanding weight load at the join): the
        // k-steps are padded to a multiple of the prefetch depth; a padding step multiplies zeros by the last step's weights.
        // (tz, ty, tx, cb) advance as scalar counters - one division per item, none per k-step.
        const int nkp = (nkl + HG_PF - 1) / HG_PF * HG_PF;
        int cb = kbeg % nchunk, tap0 = kbeg / nchunk;
        int tx = tap0 % ks, ty = (tap0 / ks) % ks, tz = tap0 / (ks * ks);
        // the LDS operand of k-step k + 1 is requested before k-step k is split and multiplied
        auto a_addr = [&](int k) -> const lds_float* {
            const int toff = ((tz * Din + ty) * Din + tx) * sp + cb * 16;                     // scalar
            const bool ok = k < nkl && (((mz >> tz) & (my >> ty) & (mx >> tx)) & 1);
            return ok ? srcl + (off0 + toff) : zl;                                            // (taps outside the volume / padding steps read zeros)
        };
        auto advance = [&]() {
            if (++cb == nchunk) { cb = 0; if (++tx == ks) { tx = 0; if (++ty == ks) { ty = 0; ++tz; } } }
            if (tz >= ks) { tz = ks - 1; ty = ks - 1; tx = ks - 1; cb = nchunk - 1; }          // (padding steps stay on the last tap)
        };
        const lds_float* q0 = a_addr(0);
        f32x4 a = *(const lds_f32x4*)q0, b = *(const lds_f32x4*)(q0 + 4);
        advance();
        for (int k0 = 0; k0 < nkp; k0 += HG_PF) {
#pragma unroll
            for (int u = 0; u < HG_PF; ++u) {
                const int k = k0 + u;
                const lds_float* qn = a_addr(k + 1);
                const f32x4 an = *(const lds_f32x4*)qn, bn = *(const lds_f32x4*)(qn + 4);
                advance();
                half8 ah, al;
                hg_split8(a, b, ah, al);
                const half8 wh = bh[u], wlo = bl[u];
                const int kn = k + HG_PF < nkl ? k + HG_PF : nkl - 1;          // (the tail re-reads the last k-step)
                bh[u] = wl[(size_t)kn * 4 * plane]; bl[u] = wl[(size_t)kn * 4 * plane + 2 * plane];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wlo, accl, 0, 0, 0);
                accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, accm, 0, 0, 0);
                a = an; b = bn;
            }
        }
        // accumulator layout (activations first): lane holds column l31 of the tile, rows (r & 3) + 8 (r >> 2) + 4 h: partial tile
        // [row][32 columns] of item `it`
        if (direct) {
            const int n = nt * 32 + l31;
            const float bv = n < Cout ? biasg[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < Vout && n < dp) dstl[row * dp + n] = n < Cout ? (acc[r] + (accl[r] + accm[r]) * (1.0f / HG_SPLIT)) + bv : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[it * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[r] + (accl[r] + accm[r]) * (1.0f / HG_SPLIT);
        }
    }
    if (direct) {
        // columns beyond the last column tile (dp > Co_pad: a 32-channel conv into a 48-pitch tensor) are padding: zero
        const int ncov = ntiles * 32;
        if (dp > ncov)
            for (int i = threadIdx.x; i < Vout * (dp - ncov); i += blockDim.x) dstl[(i / (dp - ncov)) * dp + ncov + i % (dp - ncov)] = 0.f;
        __syncthreads();
        return;
    }
    __syncthreads();
    // dst[row][n] = sum over the pair's K parts (in part order) + bias; channels Cout .. dp - 1 zeroed
    for (int i = threadIdx.x; i < Vout * dp; i += blockDim.x) {
        const int row = i / dp, n = i % dp;
        float v = 0.f;
        if (n < Cout) {
            const int pr = (row >> 5) * ntiles + (n >> 5);
            v = scr[(pr * KS) * 1024 + (row & 31) * 32 + (n & 31)];
            for (int q = 1; q < KS; ++q) v += scr[(pr * KS + q) * 1024 + (row & 31) * 32 + (n & 31)];
            v += biasg[n];
        }
        dstl[i] = v;
    }
    __syncthreads();
}

// GroupNorm (eps 1e-5, biased variance) of buf [V][pitch] in place, then optional LeakyReLU and optional residual add:
//   buf = lrelu(buf * scale + shift) (+ add)
// A thread owns ONE channel (tid % pitch) and every (256 / pitch)-th voxel: no division per element; per-thread sums -> per-channel
// sums -> per-group sums through LDS (`red`: 256 + 96 + 8 doubles), two passes (mean, then squared deviations), fp64 accumulation.
__device__ void gn_lds(float* buf, int pitch_, int V_, int C_, const NmHgNorm& g, float slope, const float* add, int addp_, double* red_generic) {
    const int pitch = __builtin_amdgcn_readfirstlane(pitch_), V = __builtin_amdgcn_readfirstlane(V_), C = __builtin_amdgcn_readfirstlane(C_);
    const int addp = __builtin_amdgcn_readfirstlane(addp_), groups = __builtin_amdgcn_readfirstlane(g.groups);
    if (g_hg_diag & 1) return;                                      // (diagnostic bit 1: no GroupNorm - timing only)
    const int cpg = C / groups, tid = threadIdx.x, vlanes = 256 / pitch;
    const int c = tid % pitch, vl = tid / pitch;
    const bool active = tid < 256 && vl < vlanes && c < C;
    const int grp = active ? c / cpg : 0;
    lds_float* bl = (lds_float*)buf;
    const lds_float* al = (const lds_float*)add;
    glb_float* gam = (glb_float*)g.gamma; glb_float* bet = (glb_float*)g.beta;
    HG_LDS double* part = (HG_LDS double*)red_generic;       // [256] per-thread, then [96] per-channel at +256, [8] per-group at +352
    HG_LDS double* chs = part + 256; HG_LDS double* grs = part + 352;
    const float gm = active ? gam[c] : 0.f, bt = active ? bet[c] : 0.f;
    double mean = 0.0;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        double s = 0.0;
        if (active) for (int v = vl; v < V; v += vlanes) { const double d = (double)bl[v * pitch + c] - mean; s += pass ? d * d : d; }
        if (tid < 256) part[tid] = s;
        __syncthreads();
        if (tid < C) { double t = 0.0; for (int l = 0; l < vlanes; ++l) t += part[l * pitch + tid]; chs[tid] = t; }
        __syncthreads();
        if (tid < groups) { double t = 0.0; for (int j = 0; j < cpg; ++j) t += chs[tid * cpg + j]; grs[tid] = t / ((double)V * cpg); }
        __syncthreads();
        if (pass == 0) mean = grs[grp];
    }
    const double var = grs[grp];
    if (active) {
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        const float sc = rstd * gm;
        const float sh = -sc * (float)mean + bt;
        for (int v = vl; v < V; v += vlanes) {
            float y = fmaf(bl[v * pitch + c], sc, sh);
            if (slope != 1.0f) y = hg_lrelu(y, slope);
            if (add) y += al[v * addp + c];
            bl[v * pitch + c] = y;
        }
    }
    __syncthreads();
}

// Res3DBlock: out = GN(conv3(lrelu(GN(conv3 x)))) + skip(x), skip = identity or GN(conv1 x); x in `xin`, result left in `t2`
// (t1 is scratch; xin is preserved)
__device__ void res_lds(const float* xin, int xp, float* t1, float* t2, int pitch, int D, const NmHgRes& r, double* red, const float* zeros, float* scratch, int sitems) {
    const int V = D * D * D;
    conv_lds(xin, xp, D, t1, pitch, D, r.c1, 1, 1, zeros, scratch, sitems);
    gn_lds(t1, pitch, V, r.c1.Cout, r.n1, 0.01f, nullptr, 0, red);
    conv_lds(t1, pitch, D, t2, pitch, D, r.c2, 1, 1, zeros, scratch, sitems);
    if (r.has_skip) {
        gn_lds(t2, pitch, V, r.c2.Cout, r.n2, 1.0f, nullptr, 0, red);
        conv_lds(xin, xp, D, t1, pitch, D, r.cs, 1, 0, zeros, scratch, sitems);
        gn_lds(t1, pitch, V, r.cs.Cout, r.ns, 1.0f, t2, pitch, red);        // t1 = GN(cs x) + t2
        for (int i = threadIdx.x; i < V * pitch; i += blockDim.x) t2[i] = t1[i];
        __syncthreads();
    } else gn_lds(t2, pitch, V, r.c2.Cout, r.n2, 1.0f, xin, xp, red);      // t2 = GN(c2 ..) + x
}

__global__ __launch_bounds__(HG_THREADS) void hg_core_kernel(NmHgCoreParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int D2 = p.D2, D3 = p.D3, V2 = D2 * D2 * D2, V3 = D3 * D3 * D3;
    const int P2 = p.pitch2, P3 = p.pitch3;
    float* B0 = lds; float* B1 = B0 + (size_t)V2 * P2; float* B2 = B1 + (size_t)V2 * P2;
    float* C0 = B2 + (size_t)V2 * P2; float* C1 = C0 + (size_t)V3 * P3; float* C2 = C1 + (size_t)V3 * P3;
    double* red = reinterpret_cast<double*>(C2 + (size_t)V3 * P3);      // 360 doubles of reduction scratch behind the tensors (all sizes are multiples of 16 floats)
    const float* zeros = reinterpret_cast<const float*>(red + 360);     // 16 zero floats: what an out-of-volume tap reads
    float* scratch = const_cast<float*>(zeros) + 16;                     // p.scratch_items partial tiles of conv_lds
    const int sitems = p.scratch_items;
    const int NT = blockDim.x;
    for (int i = tid; i < 3 * V2 * P2 + 3 * V3 * P3 + 720 + 16; i += NT) lds[i] = 0.f;   // (padding channels and the zero block must read as zeros)
    __syncthreads();
    // a2: the pool conv's raw output with its pending GroupNorm + LeakyReLU, [V2][Cin0]
    {
        const int C = p.Cin0;
        const float* src = p.in + (size_t)n * V2 * C;
        for (int i = tid; i < V2 * C; i += NT) {
            const int v = i / C, c = i % C;
            float y = src[i];
            if (p.in_scale) y = fmaf(y, p.in_scale[(size_t)n * C + c], p.in_shift[(size_t)n * C + c]);
            if (p.in_slope != 1.0f) y = hg_lrelu(y, p.in_slope);
            B0[(size_t)v * P2 + c] = y;
        }
        __syncthreads();
    }
    res_lds(B0, P2, B1, B2, P2, D2, p.e2, red, zeros, scratch, sitems);                  // e2 -> B2
    res_lds(B2, P2, B0, B1, P2, D2, p.s3, red, zeros, scratch, sitems);                  // s3 -> B1   (B0 scratch; a2 is dead)
    conv_lds(B2, P2, D2, C0, P3, D3, p.p3, 2, 0, zeros, scratch, sitems);               // pool3(e2) -> C0
    gn_lds(C0, P3, V3, p.p3.Cout, p.np3, 0.01f, nullptr, 0, red);
    res_lds(C0, P3, C1, C2, P3, D3, p.e3, red, zeros, scratch, sitems);                  // e3 -> C2
    res_lds(C2, P3, C0, C1, P3, D3, p.d3, red, zeros, scratch, sitems);                  // d3 -> C1
    // ConvTranspose3d k2 s2 (+ output_padding): out[2i + a] = sum_ci d3[i][ci] W[a][ci][co] + b; the padding planes hold the bias only
    {
        const int Co = p.u3_Cout, Ci = p.u3_Cin;
        for (int i = tid; i < V2 * Co; i += NT) {
            const int v = i / Co, co = i % Co;
            const int ox = v % D2, oy = (v / D2) % D2, oz = v / (D2 * D2);
            float acc = p.u3_bias[co];
            if (oz < 2 * D3 && oy < 2 * D3 && ox < 2 * D3) {
                const int tap = ((oz & 1) * 2 + (oy & 1)) * 2 + (ox & 1);
                const float* x = C1 + (size_t)(((oz >> 1) * D3 + (oy >> 1)) * D3 + (ox >> 1)) * P3;
                const float* w = p.u3_w + (size_t)tap * Ci * Co + co;
                float a0 = 0.f, a1 = 0.f;
                int ci = 0;
                for (; ci + 7 < Ci; ci += 8) {                    // 8 weight loads in flight (each was a dependent L2 round trip)
                    float wv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) wv[j] = w[(size_t)(ci + j) * Co];
#pragma unroll
                    for (int j = 0; j < 8; j += 2) { a0 = fmaf(x[ci + j], wv[j], a0); a1 = fmaf(x[ci + j + 1], wv[j + 1], a1); }
                }
                for (; ci < Ci; ++ci) a0 = fmaf(x[ci], w[(size_t)ci * Co], a0);
                acc += a0 + a1;
            }
            B0[(size_t)v * P2 + co] = acc;
        }
        for (int i = tid; i < V2 * (P2 - Co); i += NT) B0[(size_t)(i / (P2 - Co)) * P2 + Co + i % (P2 - Co)] = 0.f;
        __syncthreads();
        gn_lds(B0, P2, V2, Co, p.nu3, 0.01f, B1, P2, red);       // x = lrelu(GN(.)) + s3 -> B0
    }
    res_lds(B0, P2, B1, B2, P2, D2, p.d2, red, zeros, scratch, sitems);                  // out -> B2
    {
        const int C = p.d2.c2.Cout;
        float* dst = p.out + (size_t)n * V2 * C;
        for (int i = tid; i < V2 * C; i += NT) dst[i] = B2[(size_t)(i / C) * P2 + i % C];
    }
}

}  // namespace

static size_t hg_core_tensor_bytes(const NmHgCoreParams& p) {
    return ((size_t)3 * p.D2 * p.D2 * p.D2 * p.pitch2 + (size_t)3 * p.D3 * p.D3 * p.D3 * p.pitch3) * sizeof(float) + 360 * sizeof(double) + 16 * sizeof(float);
}
// partial-tile items (4 KB each) the launch gets beside its tensors: HG_MAX_ITEMS where they fit under the 150 KB gate (the 64^3 / 88^3
// grids), else what is left - possibly none (96^3: 153 280 B of tensors), and conv_lds stores directly
int nm_hg_core_scratch_items(const NmHgCoreParams& p) {
    const size_t t = hg_core_tensor_bytes(p), gate = (size_t)150 * 1024, hard = (size_t)158 * 1024;
    if (t + (size_t)HG_MAX_ITEMS * 4096 <= gate) return HG_MAX_ITEMS;
    if (t > hard) return -1;                       // does not fit at all
    return t >= gate ? 0 : (int)((gate - t) / 4096);
}
size_t nm_hg_core_lds_bytes(const NmHgCoreParams& p) {
    const int items = nm_hg_core_scratch_items(p);
    return hg_core_tensor_bytes(p) + (size_t)(items > 0 ? items : 0) * 4096;
}

int nm_launch_hg_core(const NmHgCoreParams& p, hipStream_t s) {
    const size_t lds = nm_hg_core_lds_bytes(p);
    if (nm_hg_core_scratch_items(p) < 0 || p.scratch_items != nm_hg_core_scratch_items(p)) {
        nm_set_error("hg_core: %zu bytes of LDS per frame / scratch items %d (expected %d)", lds, p.scratch_items, nm_hg_core_scratch_items(p)); return NM_ERR_UNSUPPORTED;
    }
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hg_core_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(hg_core)");
        attr_set.mark();
    }
    static int diag_set[64];                // per device (the symbol is per-device state), stored + 1 so that 0 = never written
    const char* e = getenv("NM355_HG_DIAG");
    const int diag = e ? atoi(e) : 0;
    int dev = 0; (void)hipGetDevice(&dev); dev &= 63;
    if (diag + 1 != diag_set[dev]) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hg_diag), &diag, sizeof(int)); diag_set[dev] = diag + 1; }
    hipLaunchKernelGGL(hg_core_kernel, dim3(p.N), dim3(HG_THREADS), lds, s, p);
    return nm_check_hip(hipGetLastError(), "hg_core launch");
}
