// Detector heads, decoder tail and losses (HBM-/latency-bound kernels, wavefront = 64).
//
//   heatmap_kernel      1x1 head outputs -> LeakyReLU -> propagate(2->1) -> Softplus, heat-maps in
//                       NCDHW + per-plane marginal partial sums       (kypt_detector.py:336-343)
//   keypoints_kernel    marginals -> (x1,x2,x3,intensity)             (kypt_detector_utils.py:28-55)
//   gauss_table_kernel  separable 1-D gaussians per (frame, keypoint) (kypt_detector_utils.py:57-90)
//   combined_kernel     [gauss_t | first_feature | gauss_0 | coords]  (kypt_detector.py:406-407)
//   decoder_tail_kernel conv1x1(32->1) + sigmoid(10*(tanh(v)+first_frame-0.5)), BCE partials,
//                       occupancy-masked chamfer partials     (kypt_detector.py:410,91-92;
//                                                              kypt_detector_utils.py:140-153)
//   clip_loss_kernel / loss_finalize_kernel  separation, graph consistency, trajectory losses
//                       (kypt_detector_utils.py:92-133,172-265) and the 11 scalar means
//   affinity_kernel     get_affinity ver 3                            (kypt_detector.py:191-199)
#include "nm_common.h"
#include "nm_heads.h"

namespace {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float lin_coord(int i, int G) {
    const float step = 2.0f / (float)(G - 1);
    return i < G / 2 ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(G - 1 - i), 1.0f);
}

// deterministic block sum (256 threads); result valid in every thread
__device__ __forceinline__ float block_sum256(float v, float* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    float r = sh[0];
    __syncthreads();
    return r;
}

// grid (F, g): one z-plane of one frame.  LDS: K * g * g floats.
// head / clip_head rows hold Kc = K rounded up to 8 channels per voxel (the heads run zero-padded for keypoint counts that are not
// multiples of 8): channels >= K are read with the quad they share and dropped.
__global__ __launch_bounds__(256) void heatmap_kernel(const float* __restrict__ head, const float* __restrict__ clip_head,
                                                      const float* __restrict__ prop, int T, int K, int Kc, int g,
                                                      float* __restrict__ heatmaps, float* __restrict__ part) {
    extern __shared__ float tile[];                       // [K][g*g]
    const int f = blockIdx.x, z = blockIdx.y, b = f / T;
    const int g2 = g * g, g3 = g2 * g;
    const float w0 = prop[0], w1 = prop[1], pb = prop[2];
    for (int v = threadIdx.x; v < g2; v += 256) {
        const size_t vox = (size_t)z * g2 + v;
        const float* hp = head + ((size_t)f * g3 + vox) * Kc;
        const float* cp = clip_head + ((size_t)b * g3 + vox) * Kc;
        for (int k = 0; k < K; k += 4) {
            f32x4 a = *reinterpret_cast<const f32x4*>(hp + k);
            f32x4 c = *reinterpret_cast<const f32x4*>(cp + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k + j >= K) break;
                float hm = softplus(w0 * lrelu(a[j], 0.01f) + w1 * lrelu(c[j], 0.01f) + pb);
                heatmaps[((size_t)f * K + k + j) * g3 + vox] = hm;
                tile[(k + j) * g2 + v] = hm;
            }
        }
    }
    __syncthreads();
    // Marginal sums of the plane by WAVEFRONT SHUFFLES (BASELINE north_star: "wavefront shuffles for the ... keypoint heatmap
    // reduction"): wave w takes the keypoints k = w, w + 4, ...; a plane row lives in a group of GP lanes (GP = 8 / 16 / 32 >= g), 64 / GP
    // rows per pass.  Row y: sum over x of (hm + 1e-6) - an xor butterfly inside the group; column x: every lane adds its rows in
    // registers, the groups are combined by a butterfly across them; the plane totals (with and without the 1e-6) by butterflies over
    // the whole wave.  (Rounds 1-4 walked rows, columns and totals with serial LDS loops - 3 g^2 dependent adds per keypoint on one lane
    // each; the association order differs, which moves keypoints by ~1e-7 - see tests/test_network_gpu.py::_check_losses for what that
    // does to the trajectory loss of a clip whose keypoints barely move.)
    const int stride = 2 * g + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int GP = g <= 8 ? 8 : (g <= 16 ? 16 : 32), rpp = 64 / GP;
    const int grp = lane / GP, x = lane % GP;
    for (int k = wave; k < K; k += 4) {
        const float* tk = tile + k * g2;
        float* dst = part + (((size_t)f * K + k) * g + z) * stride;
        float col = 0.f, tot = 0.f;
        for (int y0 = 0; y0 < g; y0 += rpp) {
            const int y = y0 + grp;
            const bool ok = x < g && y < g;
            const float h = ok ? tk[y * g + x] : 0.f;
            const float h6 = ok ? h + 1e-6f : 0.f;
            col += h6; tot += h;
            float r = h6;
            for (int off = GP >> 1; off > 0; off >>= 1) r += __shfl_xor(r, off);
            if (x == 0 && y < g) dst[y] = r;
        }
        for (int off = GP; off < 64; off <<= 1) col += __shfl_xor(col, off);          // the column's rows of the other groups
        if (grp == 0 && x < g) dst[g + x] = col;
        float tot6 = grp == 0 ? col : 0.f;                                            // (columns x >= g hold zeros)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { tot += nm_sx(tot, off); tot6 += nm_sx(tot6, off); }
        if (lane == 0) { dst[2 * g] = tot; dst[2 * g + 1] = tot6; }
    }
}

// one block per frame.  Phase 1: the K*3*g marginal weights w[k][d][j] in parallel (axis 0: the plane sums, axes 1/2: sums of the
// row / column partials over z, in z order); phase 2: thread (k, d) normalises and takes the expectation in j order - the same
// arithmetic, in the same order, as one thread per keypoint walking everything (which took 137 us of dependent loads per launch)
__global__ __launch_bounds__(256) void keypoints_kernel(const float* __restrict__ part, int K, int g,
                                                        float* __restrict__ keypoints, float* __restrict__ heat_mean) {
    extern __shared__ float wbuf[];          // [K][3][g] weights, then [K] means, [K][3] coordinates
    float* means = wbuf + K * 3 * g;
    float* coord = means + K;
    const int f = blockIdx.x;
    const int stride = 2 * g + 2;
    for (int item = threadIdx.x; item < K * 3 * g; item += 256) {
        const int j = item % g, d = (item / g) % 3, k = item / (3 * g);
        const float* p = part + ((size_t)f * K + k) * g * stride;
        float w;
        if (d == 0) w = p[j * stride + 2 * g + 1];
        else { w = 0.f; for (int z = 0; z < g; ++z) w += p[z * stride + (d - 1) * g + j]; }
        wbuf[item] = w;
    }
    for (int k = threadIdx.x; k < K; k += 256) {
        const float* p = part + ((size_t)f * K + k) * g * stride;
        float tot = 0.f;
        for (int z = 0; z < g; ++z) tot += p[z * stride + 2 * g];
        const float mean = tot / (float)(g * g * g);
        means[k] = mean;
        heat_mean[(size_t)f * K + k] = mean;
    }
    __syncthreads();
    for (int kd = threadIdx.x; kd < K * 3; kd += 256) {
        const float* w = wbuf + kd * g;
        float S = 0.f;
        for (int j = 0; j < g; ++j) S += w[j];
        float c = 0.f;
        for (int j = 0; j < g; ++j) c += (w[j] / S) * lin_coord(j, g);
        coord[kd] = c;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        float mx = -INFINITY;
        for (int j = 0; j < K; ++j) mx = fmaxf(mx, means[j]);
        float* o = keypoints + ((size_t)f * K + k) * 4;
        o[0] = coord[k * 3]; o[1] = coord[k * 3 + 1]; o[2] = coord[k * 3 + 2]; o[3] = means[k] / (mx + 1e-6f);
    }
}

// E[f][k][d][j] = exp(-(lin_j - c_d)^2 / width); widthk (fixed_sigma = 0): one width per keypoint
__global__ __launch_bounds__(256) void gauss_table_kernel(const float* __restrict__ keypoints, int FK, int g, float width,
                                                          const float* __restrict__ widthk, int K, float* __restrict__ table) {
    const int total = FK * 3 * g;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int j = i % g, d = (i / g) % 3, fk = i / (3 * g);
        float diff = lin_coord(j, g) - keypoints[(size_t)fk * 4 + d];
        table[i] = expf(-(diff * diff) / (widthk ? widthk[fk % K] : width));
    }
}
// fixed_sigma = 0 (kypt_detector.py:258-260, 303-306; kypt_detector_utils.py:68): width_k = 2 (sigmoid(p_k) max_sigma / g)^2, in the fp32
// operation order of the reference's tensor arithmetic
__global__ void gauss_width_kernel(const float* __restrict__ param, int K, float max_sigma, int g, float* __restrict__ widthk) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float s = 1.0f / (1.0f + expf(-param[k]));
    const float q = (s * max_sigma) / (float)g;
    widthk[k] = 2.0f * (q * q);
}

// combined[f][v][Cc]: [0,K) gauss_t | [K,K+Fd) feature of the clip's first frame | gauss_0 | 3 coords | zero pad
// (f - f % T is frame 0 of the same clip; first_feature is channels-last with a frame stride)
__global__ __launch_bounds__(256) void combined_kernel(const float* __restrict__ table, const float* __restrict__ keypoints,
                                                       const float* __restrict__ first_feature, int ff_stride, int F, int T,
                                                       int K, int Fd, int g, int Cc, int cat, float* __restrict__ out) {
    const int g2 = g * g, g3 = g2 * g, cq = Cc / 4;
    const size_t total = (size_t)F * g3 * cq;
    // cat (options.gaussian_cat_type, kypt_detector.py:396-401): 1 'max' / 2 'sum' - each of the K Gaussian channels of a block carries
    // the maximum / the sum clipped to [0, 1] of the block's K maps at that voxel
    auto reduced = [&](int fr, int z, int y, int x) {
        float m = cat == 1 ? -INFINITY : 0.f;
        for (int k = 0; k < K; ++k) {
            const float* e = table + ((size_t)fr * K + k) * 3 * g;
            const float gk = ((e[z] * e[g + y]) * e[2 * g + x]) * keypoints[((size_t)fr * K + k) * 4 + 3];
            m = cat == 1 ? fmaxf(m, gk) : m + gk;
        }
        return cat == 1 ? m : fminf(fmaxf(m, 0.f), 1.f);
    };
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int q = (int)(i % cq); size_t r = i / cq;
        int v = (int)(r % g3); int f = (int)(r / g3);
        int b = f / T, f0 = b * T;
        int x = v % g, y = (v / g) % g, z = v / g2;
        f32x4 o;
        float red_t = 0.f, red_0 = 0.f;
        if (cat) {
            if (q * 4 < K) red_t = reduced(f, z, y, x);
            if (q * 4 + 3 >= K + Fd && q * 4 < 2 * K + Fd) red_0 = reduced(f0, z, y, x);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = q * 4 + j;
            float val = 0.f;
            if (c < K) {
                const float* e = table + ((size_t)f * K + c) * 3 * g;
                val = cat ? red_t : ((e[z] * e[g + y]) * e[2 * g + x]) * keypoints[((size_t)f * K + c) * 4 + 3];
            } else if (c < K + Fd) {
                val = first_feature[(((size_t)b * ff_stride) * g3 + v) * Fd + (c - K)];
            } else if (c < 2 * K + Fd) {
                int k = c - K - Fd;
                const float* e = table + ((size_t)f0 * K + k) * 3 * g;
                val = cat ? red_0 : ((e[z] * e[g + y]) * e[2 * g + x]) * keypoints[((size_t)f0 * K + k) * 4 + 3];
            } else if (c < 2 * K + Fd + 3) {
                int d = c - 2 * K - Fd;
                val = lin_coord(d == 0 ? z : (d == 1 ? y : x), g);
            }
            o[j] = val;
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}

// three block sums at once: xor-shuffle tree inside each wave, the four wave results through LDS (fixed order); valid in every thread
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float* sh /* >= 12 floats */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += nm_sx(a, o); b += nm_sx(b, o); c += nm_sx(c, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { float* d = sh + (threadIdx.x >> 6) * 3; d[0] = a; d[1] = b; d[2] = c; }
    __syncthreads();
    a = (sh[0] + sh[3]) + (sh[6] + sh[9]); b = (sh[1] + sh[4]) + (sh[7] + sh[10]); c = (sh[2] + sh[5]) + (sh[8] + sh[11]);
}

// grid (ceil(G3 / 1024), F).  x: raw 32-channel tensor with pending GN affine + lrelu.  A block walks FOUR 256-voxel tiles (rounds
// 1-4: one, with three block reductions behind it), two tiles' loads in flight, one three-way block reduction at the end: 482 -> 450 us
// for the 2.15 GB of a 64-frame pass at 64^3 (4.8 TB/s: the kernel sits on the HBM roof of its fp32 input).
#define NM_TAIL_TILES 4
__global__ __launch_bounds__(256) void decoder_tail_kernel(TensorRef x, const float* __restrict__ w14, const float* __restrict__ first_frames,
                                                           int ff_stride_frames, int T, const float* __restrict__ target,
                                                           const float* __restrict__ keypoints, int K, int G,
                                                           float* __restrict__ recon, float* __restrict__ part) {
    __shared__ float sh[12];
    __shared__ float kp[32 * 3];
    const int f = blockIdx.y, b = f / T;
    const size_t G3 = (size_t)G * G * G;
    const int C = x.C;
    // C == 32 (the network's decoder): eight lanes fetch one voxel's 128 contiguous bytes, so a wave load covers 1 KB of
    // consecutive voxels (a thread reading its own voxel's 32 channels touches 64 different lines per load instruction, eight
    // times over); the eight 4-channel partial dot products are summed with DPP adds and lane l keeps the voxel
    // 8 (l & 7) + (l >> 3) of its wave's 64.
    const bool coop = (C == 32) && (G3 % 256 == 0);
    const int lane = threadIdx.x & 63;
    const size_t tile0 = blockIdx.x * (size_t)(256 * NM_TAIL_TILES);
    if (keypoints && threadIdx.x < K * 3) kp[threadIdx.x] = keypoints[((size_t)f * K + threadIdx.x / 3) * 4 + threadIdx.x % 3];
    __syncthreads();
    float bce = 0.f, cham = 0.f, cnt = 0.f;
    const float bias = w14[C];
    // the voxel's tail: tanh, first-frame residual, sigmoid, losses
    auto finish = [&](size_t v, float acc) __attribute__((always_inline)) {
        acc += bias;
        const float ff = first_frames[((size_t)b * ff_stride_frames) * G3 + v];
        const float pre = 10.0f * ((tanhf(acc) + ff) - 0.5f);
        const float p = 1.0f / (1.0f + expf(-pre));
        recon[(size_t)f * G3 + v] = p;
        if (target) {
            const float y = target[(size_t)f * G3 + v];
            bce += (y - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - y * fmaxf(logf(p), -100.0f);
            if (keypoints && y != 0.f) {
                int xx = (int)(v % G), yy = (int)((v / G) % G), zz = (int)(v / ((size_t)G * G));
                float cz = lin_coord(zz, G), cy = lin_coord(yy, G), cx = lin_coord(xx, G);
                float best = INFINITY;
                for (int k = 0; k < K; ++k) {
                    float d0 = cz - kp[k * 3], d1 = cy - kp[k * 3 + 1], d2 = cx - kp[k * 3 + 2];
                    best = fminf(best, (d0 * d0 + d1 * d1) + d2 * d2);
                }
                cham += best * y; cnt += y;
            }
        }
    };
    if (coop) {
        const int cq = (lane & 7) * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(x.scale + (size_t)f * C + cq);
        const f32x4 sh4 = *reinterpret_cast<const f32x4*>(x.shift + (size_t)f * C + cq);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w14 + cq);
        auto fetch = [&](size_t t0, f32x4 (&a)[8]) __attribute__((always_inline)) {
            const size_t pw = ((size_t)f * G3 + t0 + (threadIdx.x & ~63) + (lane >> 3)) * C + cq;      // element offset (x may be bf16)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = nm_ld4(x.p, pw + (size_t)i * 8 * C, x.h);
        };
        auto dot = [&](const f32x4 (&a)[8]) __attribute__((always_inline)) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float part4 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) part4 += lrelu(a[i][j] * sc[j] + sh4[j], x.slope) * wv[j];
                part4 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part4), 0xB1, 0xf, 0xf, true));
                part4 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part4), 0x4E, 0xf, 0xf, true));
                part4 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part4), 0x141, 0xf, 0xf, true));
                acc = ((lane & 7) == i) ? part4 : acc;
            }
            return acc;
        };
        const size_t vin = (threadIdx.x & ~63) + (lane & 7) * 8 + (lane >> 3);                          // the lane's voxel inside a tile
        for (int t = 0; t < NM_TAIL_TILES; t += 2) {
            const size_t ta = tile0 + (size_t)t * 256, tb = ta + 256;
            if (ta >= G3) break;
            f32x4 a[8], c[8];
            const bool two = tb < G3;
            fetch(ta, a);
            if (two) fetch(tb, c);
            finish(ta + vin, dot(a));
            if (two) finish(tb + vin, dot(c));
        }
    } else {
        for (int t = 0; t < NM_TAIL_TILES; ++t) {
            const size_t v = tile0 + (size_t)t * 256 + threadIdx.x;
            if (v >= G3) break;
            float acc = 0.f;
            const size_t px = ((size_t)f * G3 + v) * C;
            for (int c = 0; c < C; c += 4) {
                f32x4 a = nm_ld4(x.p, px + c, x.h);
                f32x4 sc = *reinterpret_cast<const f32x4*>(x.scale + (size_t)f * C + c);
                f32x4 sh4 = *reinterpret_cast<const f32x4*>(x.shift + (size_t)f * C + c);
                f32x4 wv = *reinterpret_cast<const f32x4*>(w14 + c);
                a = a * sc + sh4;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += lrelu(a[j], x.slope) * wv[j];
            }
            finish(v, acc);
        }
    }
    if (part) {
        block_sum3(bce, cham, cnt, sh);
        if (threadIdx.x == 0) {
            float* dst = part + ((size_t)f * gridDim.x + blockIdx.x) * 3;
            dst[0] = bce; dst[1] = cham; dst[2] = cnt;
        }
    }
}

// one block per clip b: partial sums of the keypoint-only losses
//   out[b][0] separation_b   [1] sum_t local   [2] sum_t time   [3] sum vel term   [4] sum acc term
__global__ __launch_bounds__(256) void clip_loss_kernel(const float* __restrict__ keypoints, const float* __restrict__ affinity,
                                                        int T, int K, int N, float sep_sigma, float* __restrict__ out) {
    __shared__ float sh[256];
    extern __shared__ float dyn[];
    float* pos = dyn;                      // [T][K][3]
    float* mean = dyn + T * K * 3;         // [K][3]
    float* infl = mean + K * 3;            // [K][K]
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T * K * 3; i += 256) {
        int d = i % 3, tk = i / 3;
        pos[i] = keypoints[((size_t)b * T * K + tk) * 4 + d];
    }
    if (affinity) for (int i = threadIdx.x; i < K * K; i += 256) {
        float m = -INFINITY;
        for (int n = 0; n < N; ++n) m = fmaxf(m, affinity[(size_t)n * K * K + i]);
        infl[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * 3; i += 256) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += pos[t * K * 3 + i];
        mean[i] = s / (float)T;
    }
    __syncthreads();
    float sep = 0.f, loc = 0.f, tim = 0.f, vel = 0.f, acc = 0.f;
    const float sep_den = 2.0f * sep_sigma * sep_sigma;
    for (int pr = threadIdx.x; pr < K * K; pr += 256) {
        const int k = pr / K, l = pr % K;
        // separation: temporal mean of squared distances of mean-centred trajectories
        float d2s = 0.f, dsum = 0.f;
        for (int t = 0; t < T; ++t) {
            float s = 0.f, sd = 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float a = pos[(t * K + k) * 3 + d], c = pos[(t * K + l) * 3 + d];
                float u = (a - mean[k * 3 + d]) - (c - mean[l * 3 + d]);
                s += u * u;
                float w = a - c;
                sd += w * w;
            }
            d2s += s; dsum += sd;
        }
        sep += expf(-(d2s / (float)T) / sep_den);
        if (affinity) {
            const float in = infl[pr];
            const float dmean = dsum / (float)T;
            float tl = 0.f, tt = 0.f;
            for (int t = 0; t < T; ++t) {
                float sd = 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) { float w = pos[(t * K + k) * 3 + d] - pos[(t * K + l) * 3 + d]; sd += w * w; }
                tl += sd * in;
                tt += fabsf(sd - dmean) * in;
            }
            loc += tl; tim += tt;
            // trajectory: cosine of velocities / accelerations of k and l (eps 1e-6 on each norm)
            for (int t = 0; t + 1 < T; ++t) {
                float vk[3], vl[3], nk = 0.f, nl = 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    vk[d] = pos[((t + 1) * K + k) * 3 + d] - pos[(t * K + k) * 3 + d];
                    vl[d] = pos[((t + 1) * K + l) * 3 + d] - pos[(t * K + l) * 3 + d];
                    nk += vk[d] * vk[d]; nl += vl[d] * vl[d];
                }
                nk = fmaxf(sqrtf(nk), 1e-6f); nl = fmaxf(sqrtf(nl), 1e-6f);
                float cs = 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) cs += (vk[d] / nk) * (vl[d] / nl);
                vel += ((-cs + 1.0f) / 2.0f) * in;
                if (t + 2 < T) {
                    float ak[3], al[3], mk = 0.f, ml = 0.f;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float vk2 = pos[((t + 2) * K + k) * 3 + d] - pos[((t + 1) * K + k) * 3 + d];
                        float vl2 = pos[((t + 2) * K + l) * 3 + d] - pos[((t + 1) * K + l) * 3 + d];
                        ak[d] = vk2 - vk[d]; al[d] = vl2 - vl[d];
                        mk += ak[d] * ak[d]; ml += al[d] * al[d];
                    }
                    mk = fmaxf(sqrtf(mk), 1e-6f); ml = fmaxf(sqrtf(ml), 1e-6f);
                    float ca = 0.f;
#pragma unroll
                    for (int d = 0; d < 3; ++d) ca += (ak[d] / mk) * (al[d] / ml);
                    acc += ((-ca + 1.0f) / 2.0f) * in;
                }
            }
        }
    }
    float s0 = block_sum256(sep, sh), s1 = block_sum256(loc, sh), s2 = block_sum256(tim, sh);
    float s3 = block_sum256(vel, sh), s4 = block_sum256(acc, sh);
    if (threadIdx.x == 0) {
        float* o = out + (size_t)b * 5;
        o[0] = (s0 - (float)K) / (float)(K * (K - 1));
        o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4;
    }
}


// ---- vol_fit_type 'gaussian' (kypt_detector_utils.py:154-169), as the reference computes it (oracle.nm_oracle.loss_volume_gaussian has
// the derivation): the Gaussian maps are TWO-dimensional - m_k[i][j] = ((1 e0_k[i]) e1_k[j]) c2_k with e_d = exp(-(lin - c_d)^2 / w),
// w = 2 (4 sigma / G)^2, c = the keypoint's three coordinates - the mask max_k m_k multiplies the frame along its first spatial axis and
// ACROSS the batch: reg[b'][t] = sum_b sum_ij (1 - mask[b][t][i][j]) col[b'][t][i][j] / S[b'][t], col = the frame summed over its first
// axis, S its total.  Three small kernels; the projection is the only pass over the voxels.
// grid (G, F), G threads: col[f][i][j] = sum_a vox[f][a][i][j]; rowsum[f][i] = sum_j col
__global__ void volfit_proj_kernel(const float* __restrict__ vox, int G, float* __restrict__ col, float* __restrict__ rowsum) {
    extern __shared__ float vsh[];
    const int i = blockIdx.x, f = blockIdx.y, j = threadIdx.x;
    const float* p = vox + ((size_t)f * G * G + i) * G + j;
    float s = 0.f;
    for (int a = 0; a < G; ++a) s += p[(size_t)a * G * G];
    col[((size_t)f * G + i) * G + j] = s;
    vsh[j] = s;
    __syncthreads();
    if (j == 0) { float r = 0.f; for (int q = 0; q < G; ++q) r += vsh[q]; rowsum[(size_t)f * G + i] = r; }
}
// grid (G, F), G threads: mneg[f][i][j] = 1 - max_k m_k[i][j]; arg (optional): the first maximal k
__global__ void volfit_mask_kernel(const float* __restrict__ keypoints, int K, int G, float width, float* __restrict__ mneg,
                                   unsigned char* __restrict__ arg) {
    const int i = blockIdx.x, f = blockIdx.y, j = threadIdx.x;
    const float li = lin_coord(i, G), lj = lin_coord(j, G);
    float mx = -INFINITY; int am = 0;
    for (int k = 0; k < K; ++k) {
        const float* kp = keypoints + ((size_t)f * K + k) * 4;
        const float d0 = li - kp[0], d1 = lj - kp[1];
        const float m = (expf(-(d0 * d0) / width) * expf(-(d1 * d1) / width)) * kp[2];
        if (m > mx) { mx = m; am = k; }
    }
    mneg[((size_t)f * G + i) * G + j] = 1.0f - mx;
    if (arg) arg[((size_t)f * G + i) * G + j] = (unsigned char)am;
}
// grid F, 256 threads: vol[f] = (sum_b sum_ij mneg[b T + t][ij] col[f][ij], S[f])
__global__ __launch_bounds__(256) void volfit_frame_kernel(const float* __restrict__ mneg, const float* __restrict__ col,
                                                           const float* __restrict__ rowsum, int B, int T, int G, float* __restrict__ vol) {
    __shared__ float sh[256];
    const int f = blockIdx.x, t = f % T, G2 = G * G;
    float num = 0.f, den = 0.f;
    for (int p = threadIdx.x; p < G2; p += 256) {
        float ms = 0.f;
        for (int b = 0; b < B; ++b) ms += mneg[((size_t)(b * T + t)) * G2 + p];
        num += ms * col[(size_t)f * G2 + p];
    }
    for (int i = threadIdx.x; i < G; i += 256) den += rowsum[(size_t)f * G + i];
    num = block_sum256(num, sh); den = block_sum256(den, sh);
    if (threadIdx.x == 0) { vol[2 * f] = num; vol[2 * f + 1] = den; }
}

// grid F: frame_sums[f] = (sum BCE, sum chamfer, occupied count) over the frame's tail partials
__global__ __launch_bounds__(256) void tail_sums_kernel(const float* __restrict__ tail_part, int tail_blocks, float* __restrict__ frame_sums) {
    __shared__ float sh[12];
    const int f = blockIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int j = threadIdx.x; j < tail_blocks; j += 256) {
        const float* p = tail_part + ((size_t)f * tail_blocks + j) * 3;
        s0 += p[0]; s1 += p[1]; s2 += p[2];
    }
    block_sum3(s0, s1, s2, sh);
    if (threadIdx.x == 0) { float* o = frame_sums + (size_t)f * 3; o[0] = s0; o[1] = s1; o[2] = s2; }
}

// single block: the 11 scalar means of KyptDetector.forward (kypt_detector.py:155-165)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ frame_sums, int B, int T,
                                                            int K, int N, int G, const float* __restrict__ heat_mean,
                                                            const float* __restrict__ clip_part, const float* __restrict__ affinity,
                                                            int chamfer, int use_traj, const float* __restrict__ vol_override,
                                                            float* __restrict__ losses) {
    __shared__ float sh[256];
    const int F = B * T;
    const float G3 = (float)G * (float)G * (float)G;
    float rec = 0.f, vol = 0.f, sp = 0.f;
    for (int f = threadIdx.x; f < F; f += 256) {
        const float* fs = frame_sums + (size_t)f * 3;
        rec += fs[0] / G3;
        vol += vol_override ? vol_override[2 * f] / vol_override[2 * f + 1] : fs[1] / fs[2];      // (vol_fit_type 'gaussian': its own numerator / denominator)
        float a = 0.f;
        for (int k = 0; k < K; ++k) a += fabsf(heat_mean[(size_t)f * K + k]);
        sp += a / (float)K;
    }
    rec = block_sum256(rec, sh); vol = block_sum256(vol, sh); sp = block_sum256(sp, sh);
    float sep = 0.f, loc = 0.f, tim = 0.f, vel = 0.f, acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* c = clip_part + (size_t)b * 5;
        sep += c[0]; loc += c[1]; tim += c[2]; vel += c[3]; acc += c[4];
    }
    sep = block_sum256(sep, sh); loc = block_sum256(loc, sh); tim = block_sum256(tim, sh);
    vel = block_sum256(vel, sh); acc = block_sum256(acc, sh);
    float spc = 0.f;
    if (affinity) {
        for (int i = threadIdx.x; i < K * K; i += 256) {
            float s = 0.f;
            for (int n = 0; n < N; ++n)
                for (int m = 0; m < N; ++m)
                    if (m != n) { float p = affinity[(size_t)n * K * K + i] * affinity[(size_t)m * K * K + i]; s += p * p; }
            spc += s;
        }
    }
    spc = block_sum256(spc, sh);
    if (threadIdx.x == 0) {
        const float KK = (float)(K * K);
        losses[0] = rec / (float)F;
        losses[1] = chamfer ? vol / (float)F : 0.f;
        losses[2] = 0.f;
        losses[3] = sep / (float)B;
        losses[4] = sp / (float)F;
        losses[5] = affinity ? loc / KK / (float)F : 0.f;
        losses[6] = affinity ? tim / KK / (float)F : 0.f;
        losses[7] = affinity ? spc / KK : 0.f;
        losses[8] = 0.f;
        losses[9] = (affinity && use_traj) ? (vel / (float)(B * (T - 1)) + acc / (float)(B * (T - 2))) / KK : 0.f;
        losses[10] = 0.f;
    }
}

// get_affinity (kypt_detector.py:171-210), one thread per row (n, k).  ver 3 (the shipped configurations; params (N, K, K-1)): softmax
// over K-1 logits, zero re-inserted on the diagonal.  ver 0 / 1 / 2 (round 6; params (N, K, K)): 0 = row softmax; 2 = softplus, zero
// diagonal, row softmax; 1 = M = S S^T of S = softplus(params), zero diagonal, rows divided by (row sum + 1e-6).
__device__ __forceinline__ float nm_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }     // torch.nn.Softplus (beta 1, threshold 20)
__global__ void affinity_kernel(const float* __restrict__ params, int N, int K, int ver, float* __restrict__ out) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N * K) return;
    const int k = row % K;
    if (ver != 3) {
        const float* p = params + (size_t)row * K;
        float* o = out + (size_t)row * K;
        if (ver == 1) {
            const float* pn = params + (size_t)(row - k) * K;     // neighbour n's (K, K) block
            float r = 0.f;
            for (int j = 0; j < K; ++j) {
                float m = 0.f;
                if (j != k) for (int c = 0; c < K; ++c) m += nm_softplus(p[c]) * nm_softplus(pn[(size_t)j * K + c]);
                o[j] = m; r += m;
            }
            for (int j = 0; j < K; ++j) o[j] = o[j] / (r + 1e-6f);
            return;
        }
        float mx = -INFINITY;
        for (int j = 0; j < K; ++j) { const float v = ver == 0 ? p[j] : (j == k ? 0.f : nm_softplus(p[j])); mx = fmaxf(mx, v); }
        float s = 0.f;
        for (int j = 0; j < K; ++j) { const float v = ver == 0 ? p[j] : (j == k ? 0.f : nm_softplus(p[j])); s += expf(v - mx); }
        for (int j = 0; j < K; ++j) { const float v = ver == 0 ? p[j] : (j == k ? 0.f : nm_softplus(p[j])); o[j] = expf(v - mx) / s; }
        return;
    }
    const float* p = params + (size_t)row * (K - 1);
    float mx = -INFINITY;
    for (int j = 0; j < K - 1; ++j) mx = fmaxf(mx, p[j]);
    float s = 0.f;
    for (int j = 0; j < K - 1; ++j) s += expf(p[j] - mx);
    float* o = out + (size_t)row * K;
    for (int j = 0; j < K; ++j) {
        if (j == k) o[j] = 0.f;
        else { int src = j < k ? j : j - 1; o[j] = expf(p[src] - mx) / s; }
    }
}

}  // namespace

int nm_launch_heatmap(const float* head, const float* clip_head, const float* prop, int F, int T, int K, int Kc, int g,
                      float* heatmaps, float* part, hipStream_t s) {
    if (K < 1 || K > 32 || Kc % 4 || Kc < K || g > 32) { nm_set_error("heatmap: K=%d (row pitch %d), g=%d unsupported", K, Kc, g); return NM_ERR_ARG; }
    size_t lds = (size_t)K * g * g * sizeof(float);
    hipLaunchKernelGGL(heatmap_kernel, dim3(F, g), dim3(256), lds, s, head, clip_head, prop, T, K, Kc, g, heatmaps, part);
    return nm_check_hip(hipGetLastError(), "heatmap launch");
}

int nm_launch_keypoints(const float* part, int F, int K, int g, float* keypoints, float* heat_mean, hipStream_t s) {
    hipLaunchKernelGGL(keypoints_kernel, dim3(F), dim3(256), (size_t)(K * 3 * g + 4 * K) * sizeof(float), s, part, K, g, keypoints, heat_mean);
    return nm_check_hip(hipGetLastError(), "keypoints launch");
}

int nm_launch_gauss_table(const float* keypoints, int FK, int g, float width, float* table, hipStream_t s, const float* widthk, int K) {
    int total = FK * 3 * g;
    hipLaunchKernelGGL(gauss_table_kernel, dim3((total + 255) / 256), dim3(256), 0, s, keypoints, FK, g, width, widthk, K > 0 ? K : 1, table);
    return nm_check_hip(hipGetLastError(), "gauss_table launch");
}
int nm_launch_gauss_width(const float* param, int K, float max_sigma, int g, float* widthk, hipStream_t s) {
    hipLaunchKernelGGL(gauss_width_kernel, dim3((K + 63) / 64), dim3(64), 0, s, param, K, max_sigma, g, widthk);
    return nm_check_hip(hipGetLastError(), "gauss_width launch");
}

// ---- the combined representation's 1x1 conv split by linearity (inference; kypt_detector.py:380-383,406) -----------------------------
// cat[gauss_t (K), first_feature (Fd), gauss_0 (K), coords (3)] -> conv1x1 -> 128: only the first K channels change from frame to
// frame of a clip.  out[f] = W[:, :K] gauss_t[f]  +  ( W[:, K:] cat[first_feature, gauss_0, coords] + bias )[clip of f]: the bracket
// is one small conv per CLIP (combined_rest -> the generic conv), the per-frame part a 24-term dot per output that never
// materialises the 184-channel tensor (193 MB per 64 frames written and read back by a fp32-MFMA conv before).
__global__ __launch_bounds__(256) void combined_rest_kernel(const float* __restrict__ table, const float* __restrict__ keypoints,
                                                            const float* __restrict__ first_feature, int ff_stride, int nb, int T,
                                                            int K, int Fd, int g, int Cr, float* __restrict__ out) {
    const int g2 = g * g, g3 = g2 * g, cq = Cr / 4;
    const size_t total = (size_t)nb * g3 * cq;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int q = (int)(i % cq); size_t r = i / cq;
        int v = (int)(r % g3); int b = (int)(r / g3);
        int f0 = b * T;
        int x = v % g, y = (v / g) % g, z = v / g2;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = q * 4 + j;
            float val = 0.f;
            if (c < Fd) val = first_feature[(((size_t)b * ff_stride) * g3 + v) * Fd + c];
            else if (c < K + Fd) {
                int k = c - Fd;
                const float* e = table + ((size_t)f0 * K + k) * 3 * g;
                val = ((e[z] * e[g + y]) * e[2 * g + x]) * keypoints[((size_t)f0 * K + k) * 4 + 3];
            } else if (c < K + Fd + 3) {
                int d = c - K - Fd;
                val = lin_coord(d == 0 ? z : (d == 1 ? y : x), g);
            }
            o[j] = val;
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}
int nm_launch_combined_rest(const float* table, const float* keypoints, const float* first_feature, int ff_stride, int nb, int T,
                            int K, int Fd, int g, int Cr, float* out, hipStream_t s) {
    size_t total = (size_t)nb * g * g * g * (Cr / 4);
    int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(combined_rest_kernel, dim3(blocks), dim3(256), 0, s, table, keypoints, first_feature, ff_stride, nb, T, K, Fd, g, Cr, out);
    return nm_check_hip(hipGetLastError(), "combined_rest launch");
}

// out[f][v][co] = base[f / T][v][co] + sum_k wg[k][co] * gauss(f, k, v); one workgroup = 64 voxels x (Cout / 4) channel quads
// (Cout = 128: thread = (voxel lane 0..7, quad 0..31), 8 voxels per thread with the thread's K weight quads held in registers), the
// 64 x K gaussian values of the block through LDS
template <int KMAX>
__global__ __launch_bounds__(256) void adjust_gauss_kernel(const float* __restrict__ table, const float* __restrict__ keypoints,
                                                           const float* __restrict__ base, const float* __restrict__ wg, int T, int K, int g,
                                                           int Cout, float* __restrict__ out) {
    __shared__ float gs[64 * KMAX];
    const int g2 = g * g, g3 = g2 * g;
    const int f = blockIdx.y, v0 = blockIdx.x * 64, tid = threadIdx.x;
    const int cq = Cout / 4, vl = tid / cq, q = tid % cq;
    f32x4 w[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) w[k] = *reinterpret_cast<const f32x4*>(wg + (size_t)(k < K ? k : K - 1) * Cout + q * 4);
    for (int i = tid; i < 64 * KMAX; i += 256) {
        const int vv = i / KMAX, k = i % KMAX, v = v0 + vv;
        float val = 0.f;
        if (v < g3 && k < K) {
            const int x = v % g, y = (v / g) % g, z = v / g2;
            const float* e = table + ((size_t)f * K + k) * 3 * g;
            val = ((e[z] * e[g + y]) * e[2 * g + x]) * keypoints[((size_t)f * K + k) * 4 + 3];
        }
        gs[i] = val;
    }
    __syncthreads();
    for (int vv = vl; vv < 64; vv += 8) {
        const int v = v0 + vv;
        if (v >= g3) break;
        f32x4 acc = *reinterpret_cast<const f32x4*>(base + (((size_t)(f / T)) * g3 + v) * Cout + q * 4);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const float gk = gs[vv * KMAX + k];          // (k >= K: zero)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[k][j], gk, acc[j]);
        }
        *reinterpret_cast<f32x4*>(out + ((size_t)f * g3 + v) * Cout + q * 4) = acc;
    }
}
int nm_launch_adjust_gauss(const float* table, const float* keypoints, const float* base, const float* wg, int F, int T, int K, int g,
                           int Cout, float* out, hipStream_t s) {
    if (K > 32 || K < 1 || Cout % 4 || 8 * (Cout / 4) != 256) { nm_set_error("adjust_gauss: unsupported K=%d Cout=%d", K, Cout); return NM_ERR_ARG; }
    const int g3 = g * g * g;
    const dim3 grid((g3 + 63) / 64, F);
    if (K <= 8) hipLaunchKernelGGL(adjust_gauss_kernel<8>, grid, dim3(256), 0, s, table, keypoints, base, wg, T, K, g, Cout, out);
    else if (K <= 16) hipLaunchKernelGGL(adjust_gauss_kernel<16>, grid, dim3(256), 0, s, table, keypoints, base, wg, T, K, g, Cout, out);
    else if (K <= 24) hipLaunchKernelGGL(adjust_gauss_kernel<24>, grid, dim3(256), 0, s, table, keypoints, base, wg, T, K, g, Cout, out);
    else hipLaunchKernelGGL(adjust_gauss_kernel<32>, grid, dim3(256), 0, s, table, keypoints, base, wg, T, K, g, Cout, out);
    return nm_check_hip(hipGetLastError(), "adjust_gauss launch");
}
// wg[k][co] = W[co][k] for k < K (W: (Cout, Cin_total) row-major)
__global__ void adjust_wg_kernel(const float* __restrict__ w, int Cout, int Cin_total, int K, float* __restrict__ wg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K * Cout) wg[i] = w[(size_t)(i % Cout) * Cin_total + i / Cout];
}
int nm_launch_adjust_wg(const float* w, int Cout, int Cin_total, int K, float* wg, hipStream_t s) {
    hipLaunchKernelGGL(adjust_wg_kernel, dim3((K * Cout + 255) / 256), dim3(256), 0, s, w, Cout, Cin_total, K, wg);
    return nm_check_hip(hipGetLastError(), "adjust_wg launch");
}

int nm_launch_combined(const float* table, const float* keypoints, const float* first_feature, int ff_stride, int F,
                       int T, int K, int Fd, int g, int Cc, float* out, hipStream_t s, int cat) {
    size_t total = (size_t)F * g * g * g * (Cc / 4);
    int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(combined_kernel, dim3(blocks), dim3(256), 0, s, table, keypoints, first_feature, ff_stride, F, T, K, Fd,
                       g, Cc, cat, out);
    return nm_check_hip(hipGetLastError(), "combined launch");
}

int nm_tail_blocks(int G) { return (int)(((size_t)G * G * G + 256 * NM_TAIL_TILES - 1) / (256 * NM_TAIL_TILES)); }

int nm_launch_decoder_tail(const TensorRef& x, const float* w14, const float* first_frames, int ff_stride_frames, int T,
                           const float* target, const float* keypoints, int K, int G, float* recon, float* part,
                           hipStream_t s) {
    if (x.C % 4 || !x.scale) { nm_set_error("decoder_tail: needs a lazy GN input with C %% 4 == 0"); return NM_ERR_ARG; }
    hipLaunchKernelGGL(decoder_tail_kernel, dim3(nm_tail_blocks(G), x.N), dim3(256), 0, s, x, w14, first_frames,
                       ff_stride_frames, T, target, keypoints, K, G, recon, part);
    return nm_check_hip(hipGetLastError(), "decoder_tail launch");
}

int nm_launch_clip_loss(const float* keypoints, const float* affinity, int B, int T, int K, int N, float sep_sigma,
                        float* out, hipStream_t s) {
    size_t lds = ((size_t)T * K * 3 + K * 3 + K * K) * sizeof(float);
    hipLaunchKernelGGL(clip_loss_kernel, dim3(B), dim3(256), lds, s, keypoints, affinity, T, K, N, sep_sigma, out);
    return nm_check_hip(hipGetLastError(), "clip_loss launch");
}

int nm_launch_loss_finalize(const float* tail_part, int tail_blocks, int B, int T, int K, int N, int G,
                            const float* heat_mean, const float* clip_part, const float* affinity, int chamfer,
                            int use_traj, float* frame_sums, float* losses, hipStream_t s, const float* vol_override) {
    hipLaunchKernelGGL(tail_sums_kernel, dim3(B * T), dim3(256), 0, s, tail_part, tail_blocks, frame_sums);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, frame_sums, B, T, K, N, G, heat_mean,
                       clip_part, affinity, chamfer, use_traj, vol_override, losses);
    return nm_check_hip(hipGetLastError(), "loss_finalize launch");
}

size_t nm_volfit_gauss_ws_floats(int F, int G) { return (size_t)F * G * G * 2 + (size_t)F * G + (size_t)F * G * G / 4 + 64; }
// vol [F][2]: (numerator, denominator) of vol_fit_type 'gaussian' per frame; ws: nm_volfit_gauss_ws_floats(F, G) floats
int nm_launch_volfit_gauss(const float* vox, const float* keypoints, int B, int T, int K, int G, float sigma, float* ws, float* vol, hipStream_t s) {
    if (G > 1024 || K > 255) { nm_set_error("volfit_gauss: G %d / K %d unsupported", G, K); return NM_ERR_UNSUPPORTED; }
    const int F = B * T;
    float* col = ws; float* mneg = col + (size_t)F * G * G; float* rowsum = mneg + (size_t)F * G * G;
    const float width = (float)(2.0 * std::pow((double)sigma * 4.0 / (double)G, 2.0));
    hipLaunchKernelGGL(volfit_proj_kernel, dim3(G, F), dim3(G), G * sizeof(float), s, vox, G, col, rowsum);
    hipLaunchKernelGGL(volfit_mask_kernel, dim3(G, F), dim3(G), 0, s, keypoints, K, G, width, mneg, (unsigned char*)nullptr);
    hipLaunchKernelGGL(volfit_frame_kernel, dim3(F), dim3(256), 0, s, mneg, col, rowsum, B, T, G, vol);
    return nm_check_hip(hipGetLastError(), "volfit_gauss launch");
}

int nm_launch_affinity(const float* params, int N, int K, float* out, hipStream_t s, int ver) {
    hipLaunchKernelGGL(affinity_kernel, dim3((N * K + 63) / 64), dim3(64), 0, s, params, N, K, ver, out);
    return nm_check_hip(hipGetLastError(), "affinity launch");
}

// ---- input path: episodic normalisation + voxelisation on the device ---------------------------------
// Restates utils/dataset_utils.py:9-31 operation by operation in fp64 so that the voxel indices are bit-exact:
//   seq' = ((seq - bmin) * scale / (blen + 1e-5)) * 2 - 1,  idx = int32((seq' + 1) / (2/G + 1e-5)),  grid[idx] = 1.
namespace {

// per-block partial bbox of the whole episode: part[blk][6] = (min xyz, max xyz)
__global__ __launch_bounds__(256) void bbox_partial_kernel(const double* __restrict__ pts, size_t npts, double* __restrict__ part) {
    __shared__ double sh[256 * 6];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < npts; i += (size_t)gridDim.x * 256)
        for (int d = 0; d < 3; ++d) { double v = pts[i * 3 + d]; mn[d] = fmin(mn[d], v); mx[d] = fmax(mx[d], v); }
    for (int d = 0; d < 3; ++d) { sh[threadIdx.x * 6 + d] = mn[d]; sh[threadIdx.x * 6 + 3 + d] = mx[d]; }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int d = 0; d < 3; ++d) {
                sh[threadIdx.x * 6 + d] = fmin(sh[threadIdx.x * 6 + d], sh[(threadIdx.x + st) * 6 + d]);
                sh[threadIdx.x * 6 + 3 + d] = fmax(sh[threadIdx.x * 6 + 3 + d], sh[(threadIdx.x + st) * 6 + 3 + d]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[(size_t)blockIdx.x * 6 + threadIdx.x] = sh[threadIdx.x];
}

__global__ __launch_bounds__(256) void voxelize_kernel(const double* __restrict__ pts, int T, size_t N, int G, double scale,
                                                       const double* __restrict__ part, int nparts, float* __restrict__ vox,
                                                       int32_t* __restrict__ idx_out) {
    __shared__ double bb[6];
    if (threadIdx.x < 6) {
        double v = part[threadIdx.x];
        for (int j = 1; j < nparts; ++j) v = threadIdx.x < 3 ? fmin(v, part[(size_t)j * 6 + threadIdx.x]) : fmax(v, part[(size_t)j * 6 + threadIdx.x]);
        bb[threadIdx.x] = v;
    }
    __syncthreads();
    const double blen = fmax(fmax(bb[3] - bb[0], bb[4] - bb[1]), bb[5] - bb[2]);
    const double den = blen + 1e-5;
    const double step = 2.0 / (double)G + 1e-5;
    const size_t G3 = (size_t)G * G * G;
    const size_t total = (size_t)T * N;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int id[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double v = pts[i * 3 + d] - bb[d];
            v = v * scale; v = v / den; v = v * 2.0; v = v - 1.0;
            v = v + 0.0;                       // the reference adds the (zero) translation vector
            v = v - (-1.0);
            v = v / step;
            id[d] = (int)v;                    // astype(np.int32): truncation toward zero
        }
        if (idx_out) { idx_out[i * 3] = id[0]; idx_out[i * 3 + 1] = id[1]; idx_out[i * 3 + 2] = id[2]; }
        if ((unsigned)id[0] < (unsigned)G && (unsigned)id[1] < (unsigned)G && (unsigned)id[2] < (unsigned)G)
            vox[(i / N) * G3 + ((size_t)id[0] * G + id[1]) * G + id[2]] = 1.0f;     // idempotent set: no atomics needed
    }
}

}  // namespace

int nm_launch_voxelize(const double* pts, int T, size_t N, int G, double scale, double* part_ws /* >= 256*6 doubles */,
                       float* vox, int32_t* idx_out, hipStream_t s) {
    const size_t npts = (size_t)T * N;
    const int nparts = (int)((npts + 255) / 256 < 256 ? (npts + 255) / 256 : 256);
    int rc = nm_check_hip(hipMemsetAsync(vox, 0, (size_t)T * G * G * G * sizeof(float), s), "voxelize: memset");
    if (rc) return rc;
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nparts), dim3(256), 0, s, pts, npts, part_ws);
    hipLaunchKernelGGL(voxelize_kernel, dim3((unsigned)((npts + 255) / 256 < 2048 ? (npts + 255) / 256 : 2048)), dim3(256), 0, s, pts, T, N,
                       G, scale, part_ws, nparts, vox, idx_out);
    return nm_check_hip(hipGetLastError(), "voxelize launch");
}
