// Backward of the detector heads, decoder tail and losses (the adjoints of nm_heads.hip; reference: autograd of
// model/kypt_detector.py:81-169,336-353,406-410 and utils/kypt_detector_utils.py:28-265 under train.py:388-404).
// `dloss` is the device vector of d(total loss)/d(loss_i) for the 11 scalars in the order nm_detector_forward returns them:
//   0 recon  1 vol_fit  2 kypt_const  3 separation  4 sparsity  5 local  6 time  7 sparsity_const  8 intensity  9 traj  10 graph_vol
// All reductions are block-ordered sums into scratch followed by a fixed-order reduce: run-to-run identical.
#include "nm_heads_bwd.h"

namespace {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float lin_coord(int i, int G) {
    const float step = 2.0f / (float)(G - 1);
    return i < G / 2 ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(G - 1 - i), 1.0f);
}
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

// ---- decoder tail: recon = sigmoid(10*(tanh(w.a + b) + first_frame - 0.5)), BCE mean (kypt_detector.py:410,91-92) ---------------
#define NM_TAILB_VPB 2048         // voxels per block
// grid (G3/VPB, F); dA[f][v][c] = dv * w[c]; part[f][blk][C+1] = (sum dv*a_c, sum dv)
__global__ __launch_bounds__(256) void tail_bwd_kernel(TensorRef x, const float* __restrict__ w14, const float* __restrict__ target,
                                                       const float* __restrict__ recon, const float* __restrict__ dloss, float inv_count,
                                                       size_t G3, float* __restrict__ dA, float* __restrict__ part) {
    extern __shared__ float sh[];          // [256][C+1]
    const int f = blockIdx.y, C = x.C, CP = C + 1;
    const float coef = dloss[0] * inv_count;
    float* mine = sh + threadIdx.x * CP;
    for (int c = 0; c <= C; ++c) mine[c] = 0.f;
    const size_t v0 = blockIdx.x * (size_t)NM_TAILB_VPB;
    for (int it = 0; it < NM_TAILB_VPB / 256; ++it) {
        const size_t v = v0 + it * 256 + threadIdx.x;
        if (v >= G3) break;
        const size_t px = ((size_t)f * G3 + v) * C;        // element offset: x (and dA, the same shape) may be stored as bfloat16
        float acc = 0.f;
        for (int c = 0; c < C; c += 4) {
            f32x4 a = nm_ld4(x.p, px + c, x.h);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(x.scale + (size_t)f * C + c);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(x.shift + (size_t)f * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += lrelu(fmaf(a[j], sc[j], s4[j]), x.slope) * w14[c + j];
        }
        acc += w14[C];
        const float th = tanhf(acc);
        const float p = recon[(size_t)f * G3 + v], y = target[(size_t)f * G3 + v];
        // BCELoss backward (ATen): (p - y) / max(p (1 - p), 1e-12); sigmoid: * p (1 - p); * 10; tanh: * (1 - th^2)
        const float pq = (1.0f - p) * p;
        const float dv = coef * ((p - y) / fmaxf(pq, 1e-12f)) * pq * 10.0f * (1.0f - th * th);
        for (int c = 0; c < C; c += 4) {
            f32x4 a = nm_ld4(x.p, px + c, x.h);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(x.scale + (size_t)f * C + c);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(x.shift + (size_t)f * C + c);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mine[c + j] += dv * lrelu(fmaf(a[j], sc[j], s4[j]), x.slope);
                o[j] = dv * w14[c + j];
            }
            nm_st4(dA, px + c, o, x.h);
        }
        mine[C] += dv;
    }
    __syncthreads();
    for (int c = threadIdx.x; c <= C; c += 256) {
        float s = 0.f;
        for (int t = 0; t < 256; ++t) s += sh[t * CP + c];
        part[((size_t)f * gridDim.x + blockIdx.x) * CP + c] = s;
    }
}

// The network's decoder (C = 32): the voxel's activated channels stay in registers between the dot product and the outer
// product, and the per-thread partial sums of d w14 live in registers too (the generic kernel above keeps them in LDS: 33
// read-modify-writes per voxel), reduced once per block.
// (dvout: the per-voxel factor alone - dA = dvout (x) w14 is then formed by the GroupNorm backward on read, nm_grad.h)
__global__ __launch_bounds__(256) void tail_bwd32_kernel(TensorRef x, const float* __restrict__ w14, const float* __restrict__ target,
                                                         const float* __restrict__ recon, const float* __restrict__ dloss, float inv_count,
                                                         size_t G3, float* __restrict__ dA, float* __restrict__ part, float* __restrict__ dvout) {
    constexpr int C = 32;
    __shared__ float sh[256 * 33];
    const int f = blockIdx.y;
    const float coef = dloss[0] * inv_count;
    float sc[C], sf[C], w[C], acc[C + 1];
#pragma unroll
    for (int c = 0; c < C; ++c) { sc[c] = x.scale[(size_t)f * C + c]; sf[c] = x.shift[(size_t)f * C + c]; w[c] = w14[c]; acc[c] = 0.f; }
    acc[C] = 0.f;
    const float bias = w14[C];
    const size_t v0 = blockIdx.x * (size_t)NM_TAILB_VPB;
    for (int it = 0; it < NM_TAILB_VPB / 256; ++it) {
        const size_t v = v0 + it * 256 + threadIdx.x;
        if (v >= G3) break;
        const size_t px = ((size_t)f * G3 + v) * C;
        float a[C];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            const f32x4 r = nm_ld4(x.p, px + c, x.h);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[c + j] = lrelu(fmaf(r[j], sc[c + j], sf[c + j]), x.slope); dot += a[c + j] * w[c + j]; }
        }
        dot += bias;
        const float th = tanhf(dot);
        const float p = recon[(size_t)f * G3 + v], y = target[(size_t)f * G3 + v];
        const float pq = (1.0f - p) * p;
        const float dv = coef * ((p - y) / fmaxf(pq, 1e-12f)) * pq * 10.0f * (1.0f - th * th);
        if (dvout) {
            dvout[(size_t)f * G3 + v] = dv;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += dv * a[c];
        } else {
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[c + j] += dv * a[c + j]; o[j] = dv * w[c + j]; }
                nm_st4(dA, px + c, o, x.h);
            }
        }
        acc[C] += dv;
    }
#pragma unroll
    for (int c = 0; c <= C; ++c) sh[threadIdx.x * 33 + c] = acc[c];
    __syncthreads();
    for (int c = threadIdx.x; c <= C; c += 256) {
        float s = 0.f;
        for (int t = 0; t < 256; ++t) s += sh[t * 33 + c];
        part[((size_t)f * gridDim.x + blockIdx.x) * 33 + c] = s;
    }
}

// out[j] = sum_r part[r*cols + j]   (one block per column)
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ part, int rows, int cols, float* __restrict__ out) {
    __shared__ double sh[256];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)part[(size_t)r * cols + j];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) out[j] = (float)sh[0];
}

// ---- volume-fitting (chamfer) loss: mean_f [ sum_v occ * min_k |c_v - kp_k|^2 / sum_v occ ]  (kypt_detector_utils.py:140-153) ---
#define NM_CHAM_VPB 4096
// grid (blocks, F): part[f][blk][K*3] = sum over occupied voxels whose nearest keypoint is k of (c_v - kp_k)
__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float* __restrict__ target, const float* __restrict__ keypoints, int K, int G,
                                                          float* __restrict__ part) {
    __shared__ float kp[32 * 3];
    __shared__ int sk[256];
    __shared__ float sd[256 * 3];
    __shared__ unsigned long long smask[4];
    const int f = blockIdx.y;
    const size_t G3 = (size_t)G * G * G;
    if ((int)threadIdx.x < K * 3) kp[threadIdx.x] = keypoints[((size_t)f * K + threadIdx.x / 3) * 4 + threadIdx.x % 3];
    __syncthreads();
    float acc = 0.f;                                      // thread t < K*3 owns component (k, d) = (t / 3, t % 3)
    const size_t v0 = blockIdx.x * (size_t)NM_CHAM_VPB;
    for (int it = 0; it < NM_CHAM_VPB / 256; ++it) {
        const size_t v = v0 + it * 256 + threadIdx.x;
        int bk = -1; float d[3] = {0.f, 0.f, 0.f};
        if (v < G3) {
            const float y = target[(size_t)f * G3 + v];
            if (y != 0.f) {
                const int xx = (int)(v % G), yy = (int)((v / G) % G), zz = (int)(v / ((size_t)G * G));
                const float c0 = lin_coord(zz, G), c1 = lin_coord(yy, G), c2 = lin_coord(xx, G);
                float best = INFINITY;
                for (int k = 0; k < K; ++k) {
                    const float d0 = c0 - kp[k * 3], d1 = c1 - kp[k * 3 + 1], d2 = c2 - kp[k * 3 + 2];
                    const float dist = (d0 * d0 + d1 * d1) + d2 * d2;
                    if (dist < best) { best = dist; bk = k; d[0] = d0 * y; d[1] = d1 * y; d[2] = d2 * y; }
                }
            }
        }
        // the occupied voxels of the group (1-3 % of a grid, most groups have none) as one ballot per wave: the owners of the K * 3
        // components visit only those, in voxel order - the order of the 256-entry scan this replaces (0.5 ms per launch)
        const unsigned long long occm = __ballot(bk >= 0);
        if ((threadIdx.x & 63) == 0) smask[threadIdx.x >> 6] = occm;
        sk[threadIdx.x] = bk; sd[threadIdx.x * 3] = d[0]; sd[threadIdx.x * 3 + 1] = d[1]; sd[threadIdx.x * 3 + 2] = d[2];
        __syncthreads();
        if ((int)threadIdx.x < K * 3) {
            const int k = threadIdx.x / 3, dd = threadIdx.x % 3;
            for (int wv = 0; wv < 4; ++wv) {
                unsigned long long m = smask[wv];
                while (m) {
                    const int t = wv * 64 + __builtin_ctzll(m);
                    m &= m - 1;
                    if (sk[t] == k) acc += sd[t * 3 + dd];
                }
            }
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < K * 3) part[((size_t)f * gridDim.x + blockIdx.x) * (K * 3) + threadIdx.x] = acc;
}
// grid F, thread (k,d): dkp[f][k][d] += dloss[1]/F * (-2) / cnt_f * sum_blk part
__global__ void chamfer_bwd_finish_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ tail_part, int tail_blocks,
                                          const float* __restrict__ dloss, int F, int K, float* __restrict__ dkp) {
    const int f = blockIdx.x, t = threadIdx.x;
    if (t >= K * 3) return;
    // (eight loads in flight per step, added in index order: as written one by one the two sums were ~130 dependent L2 round trips)
    float cnt = 0.f, s = 0.f;
    for (int j0 = 0; j0 < tail_blocks; j0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = tail_part[((size_t)f * tail_blocks + min(j0 + u, tail_blocks - 1)) * 3 + 2];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (j0 + u < tail_blocks) cnt += v[u];
    }
    for (int j0 = 0; j0 < nblk; j0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[((size_t)f * nblk + min(j0 + u, nblk - 1)) * (K * 3) + t];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (j0 + u < nblk) s += v[u];
    }
    dkp[((size_t)f * K + t / 3) * 4 + t % 3] += (dloss[1] / (float)F) * (-2.0f * s) / cnt;
}


// ---- vol_fit_type 'gaussian' (kypt_detector_utils.py:154-169; forward: nm_heads.hip volfit_*): d loss[1] -> d keypoints ------------------
// The loss is linear in mneg = 1 - max_k m_k with the weight wsum[t][ij] = sum_b' col[b'][t][ij] / S[b'][t] - the same for every batch
// element b, the reference's cross-batch broadcast - so d m_k*[b][t][ij] = -(dloss[1] / (B T)) wsum[t][ij] at the maximal map k*.
__global__ void volfit_proj_b_kernel(const float* __restrict__ vox, int G, float* __restrict__ col, float* __restrict__ rowsum) {
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
// grid (G, T), G threads: wsum[t][i][j]
__global__ void volfit_wsum_kernel(const float* __restrict__ col, const float* __restrict__ rowsum, int B, int T, int G, float* __restrict__ wsum) {
    const int i = blockIdx.x, t = blockIdx.y, j = threadIdx.x;
    float w = 0.f;
    for (int b = 0; b < B; ++b) {
        const int f = b * T + t;
        float S = 0.f;
        for (int q = 0; q < G; ++q) S += rowsum[(size_t)f * G + q];
        w += col[((size_t)f * G + i) * G + j] / S;
    }
    wsum[((size_t)t * G + i) * G + j] = w;
}
// grid F * K, 256 threads: the pixels whose first maximal map is k
__global__ __launch_bounds__(256) void volfit_bwd_kernel(const float* __restrict__ keypoints, const float* __restrict__ wsum,
                                                         const float* __restrict__ dloss, int B, int T, int K, int G, float width,
                                                         float* __restrict__ dkp) {
    __shared__ float sh[256];
    const int fk = blockIdx.x, f = fk / K, k = fk % K, t = f % T, G2 = G * G;
    const float gl = -dloss[1] / (float)(B * T);
    const float* kq = keypoints + (size_t)f * K * 4;
    const float c0 = kq[k * 4], c1 = kq[k * 4 + 1], c2 = kq[k * 4 + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int p = threadIdx.x; p < G2; p += 256) {
        const int i = p / G, j = p % G;
        const float li = lin_coord(i, G), lj = lin_coord(j, G);
        float mx = -INFINITY; int am = 0;
        for (int q = 0; q < K; ++q) {
            const float d0 = li - kq[q * 4], d1 = lj - kq[q * 4 + 1];
            const float m = (expf(-(d0 * d0) / width) * expf(-(d1 * d1) / width)) * kq[q * 4 + 2];
            if (m > mx) { mx = m; am = q; }
        }
        if (am != k) continue;
        const float E = expf(-((li - c0) * (li - c0)) / width) * expf(-((lj - c1) * (lj - c1)) / width);
        const float gw = gl * wsum[(size_t)t * G2 + p];
        a0 += gw * (E * c2) * (2.0f * (li - c0) / width);
        a1 += gw * (E * c2) * (2.0f * (lj - c1) / width);
        a2 += gw * E;
    }
    a0 = block_sum256(a0, sh); a1 = block_sum256(a1, sh); a2 = block_sum256(a2, sh);
    if (threadIdx.x == 0) { float* o = dkp + (size_t)fk * 4; o[0] += a0; o[1] += a1; o[2] += a2; }
}

// ---- combined representation (kypt_detector.py:406) and the gaussian maps (kypt_detector_utils.py:57-90) -------------------------
// grid (F*K, 2): role 0 = channel k (gaussian of frame f), role 1 = channel K+Fd+k (gaussian of the clip's first frame)
// out[((f*K+k)*2+role)*4 + j] = (dc0, dc1, dc2, dI) contributions
__global__ __launch_bounds__(256) void gauss_bwd_kernel(const float* __restrict__ dcomb, int Cd, const float* __restrict__ table,
                                                        const float* __restrict__ keypoints, int T, int K, int Fd, int g, float width_in,
                                                        int cat, const float* __restrict__ widthk, float* __restrict__ dwidth_part,
                                                        float* __restrict__ out) {
    __shared__ float sh[256];
    const int fk = blockIdx.x, role = blockIdx.y, f = fk / K, k = fk % K;
    const float width = widthk ? widthk[k] : width_in;                 // (fixed_sigma = 0: per-keypoint widths, and their gradient below)
    float a4 = 0.f;
    const int fr = role ? (f / T) * T : f;
    const int ch = role ? K + Fd + k : k;
    const int g2 = g * g, g3 = g2 * g;
    const float* e = table + ((size_t)fr * K + k) * 3 * g;
    const float* kp = keypoints + ((size_t)fr * K + k) * 4;
    const float c0 = kp[0], c1 = kp[1], c2 = kp[2], I = kp[3];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int v = threadIdx.x; v < g3; v += 256) {
        const int x = v % g, y = (v / g) % g, z = v / g2;
        float dG = dcomb[((size_t)f * g3 + v) * Cd + ch];
        const float E = (e[z] * e[g + y]) * e[2 * g + x];
        if (cat) {
            // gaussian_cat_type 'max' / 'sum' (kypt_detector.py:396-401): the block's K channels all carry R(v) = max_k g_k(v) or
            // clip(sum_k g_k(v), 0, 1); dR(v) = the sum of the K channels' gradients; max: it belongs to the first maximal map, sum: to
            // every map where the sum lies inside [0, 1]
            const float* dc = dcomb + ((size_t)f * g3 + v) * Cd + (role ? K + Fd : 0);
            float dR = 0.f, red = cat == 1 ? -INFINITY : 0.f; int am = 0;
            for (int j = 0; j < K; ++j) {
                dR += dc[j];
                const float* ej = table + ((size_t)fr * K + j) * 3 * g;
                const float gj = ((ej[z] * ej[g + y]) * ej[2 * g + x]) * keypoints[((size_t)fr * K + j) * 4 + 3];
                if (cat == 1) { if (gj > red) { red = gj; am = j; } } else red += gj;
            }
            dG = cat == 1 ? (am == k ? dR : 0.f) : ((red >= 0.f && red <= 1.f) ? dR : 0.f);
        }
        const float GI = dG * E * I;
        a0 += GI * (2.0f * (lin_coord(z, g) - c0) / width);
        a1 += GI * (2.0f * (lin_coord(y, g) - c1) / width);
        a2 += GI * (2.0f * (lin_coord(x, g) - c2) / width);
        a3 += dG * E;
        if (dwidth_part) {                                              // d E / d width = E (sum_d diff_d^2) / width^2
            const float q0 = lin_coord(z, g) - c0, q1 = lin_coord(y, g) - c1, q2 = lin_coord(x, g) - c2;
            a4 += GI * (((q0 * q0 + q1 * q1) + q2 * q2) / (width * width));
        }
    }
    a0 = block_sum256(a0, sh); a1 = block_sum256(a1, sh); a2 = block_sum256(a2, sh); a3 = block_sum256(a3, sh);
    if (dwidth_part) { a4 = block_sum256(a4, sh); if (threadIdx.x == 0) dwidth_part[(size_t)fk * 2 + role] = a4; }
    if (threadIdx.x == 0) {
        float* o = out + (((size_t)fk) * 2 + role) * 4;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
}
// fixed_sigma = 0: d loss / d sigmas parameter [K] = (sum over frames and both roles of d loss / d width_k) * d width / d sigma * d sigma / d p,
// width = 2 (sigma / g)^2, sigma = sigmoid(p) max_sigma
__global__ void sigma_bwd_finish_kernel(const float* __restrict__ dwidth_part, const float* __restrict__ param, int F, int K, float max_sigma,
                                        int g, float* __restrict__ dparam) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += dwidth_part[((size_t)f * K + k) * 2] + dwidth_part[((size_t)f * K + k) * 2 + 1];
    const float sg = 1.0f / (1.0f + expf(-param[k])), sigma = sg * max_sigma;
    dparam[k] = s * (4.0f * sigma / ((float)g * (float)g)) * (max_sigma * sg * (1.0f - sg));
}
__global__ void gauss_bwd_finish_kernel(const float* __restrict__ part, int F, int T, int K, float* __restrict__ dkp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * K * 4) return;
    const int j = i % 4, k = (i / 4) % K, f = i / (4 * K);
    float s = part[(((size_t)f * K + k) * 2) * 4 + j];
    if (f % T == 0) for (int t = 0; t < T; ++t) s += part[(((size_t)(f + t) * K + k) * 2 + 1) * 4 + j];
    dkp[i] += s;
}
// dfeat[(b*T)][v][c] += sum_t dcomb[b*T+t][v][K+c]
__global__ __launch_bounds__(256) void first_feature_bwd_kernel(const float* __restrict__ dcomb, int Cd, int B, int T, int K, int Fd, int g3,
                                                                float* __restrict__ dfeat) {
    const int cq = Fd / 4;
    const size_t total = (size_t)B * g3 * cq;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % cq); const size_t r = i / cq;
        const int v = (int)(r % g3), b = (int)(r / g3);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if ((K & 3) == 0) for (int t = 0; t < T; ++t) s += *reinterpret_cast<const f32x4*>(dcomb + (((size_t)(b * T + t)) * g3 + v) * Cd + K + q * 4);
        else for (int t = 0; t < T; ++t) {          // (keypoint counts that are not multiples of 4: the feature block starts off a 16-byte boundary)
            const float* p = dcomb + (((size_t)(b * T + t)) * g3 + v) * Cd + K + q * 4;
            s += f32x4{p[0], p[1], p[2], p[3]};
        }
        f32x4* d = reinterpret_cast<f32x4*>(dfeat + (((size_t)b * T) * g3 + v) * Fd + q * 4);
        *d = *d + s;
    }
}

// ---- keypoints from heat-maps (kypt_detector_utils.py:28-55) + sparsity loss (:92-103) + propagate/softplus (kypt_detector.py:336-343)
// one block per frame: coef[f][k] = (A0, A1, A2, Bk) with dL/dhm[v] = A0 lin(z) + A1 lin(y) + A2 lin(x) + Bk
__global__ __launch_bounds__(64) void heat_bwd_prep_kernel(const float* __restrict__ part, const float* __restrict__ heat_mean,
                                                           const float* __restrict__ keypoints, const float* __restrict__ dkp,
                                                           const float* __restrict__ dloss, int F, int K, int g, float* __restrict__ coef) {
    __shared__ float means[64], dots[64];
    const int f = blockIdx.x, k = threadIdx.x;
    const int stride = 2 * g + 2;
    float mean = -INFINITY, dI = 0.f;
    if (k < K) { mean = heat_mean[(size_t)f * K + k]; dI = dkp[((size_t)f * K + k) * 4 + 3]; }
    means[k] = mean; dots[k] = (k < K) ? dI * mean : 0.f;
    __syncthreads();
    if (k >= K) return;
    float mx = -INFINITY; int am = 0; float dot = 0.f;
    for (int j = 0; j < K; ++j) { if (means[j] > mx) { mx = means[j]; am = j; } dot += dots[j]; }
    const float den = mx + 1e-6f;
    float dmean = dI / den - (k == am ? dot / (den * den) : 0.f);
    const float sgn = mean > 0.f ? 1.f : (mean < 0.f ? -1.f : 0.f);
    dmean += dloss[4] * sgn / ((float)K * (float)F);
    const float* p = part + ((size_t)f * K + k) * g * stride;
    float tot6 = 0.f;
    for (int z = 0; z < g; ++z) tot6 += p[z * stride + 2 * g + 1];
    const float* kp = keypoints + ((size_t)f * K + k) * 4;
    const float* dk = dkp + ((size_t)f * K + k) * 4;
    const float A0 = dk[0] / tot6, A1 = dk[1] / tot6, A2 = dk[2] / tot6;
    float* o = coef + ((size_t)f * K + k) * 4;
    o[0] = A0; o[1] = A1; o[2] = A2;
    o[3] = dmean / (float)(g * g * g) - (A0 * kp[0] + A1 * kp[1] + A2 * kp[2]);
}
// grid (F, g): dhead[f][v][k], dchead_t[f][v][k] (summed over t afterwards), pp[(f*g+z)][3] = (sum du*lrelu(head), sum du*lrelu(chead), sum du)
// (head / clip_head / dhead / dchead_t rows hold Kc = K rounded up to 8 channels; the gradients of the padded channels are zeros)
__global__ __launch_bounds__(256) void heat_bwd_kernel(const float* __restrict__ head, const float* __restrict__ clip_head,
                                                       const float* __restrict__ prop, const float* __restrict__ coef, int T, int K, int Kc, int g,
                                                       float* __restrict__ dhead, float* __restrict__ dchead_t, float* __restrict__ pp) {
    __shared__ float sh[256];
    const int f = blockIdx.x, z = blockIdx.y, b = f / T;
    const int g2 = g * g, g3 = g2 * g;
    const float w0 = prop[0], w1 = prop[1], pb = prop[2];
    const float lz = lin_coord(z, g);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < g2 * Kc; i += 256) {
        const int k = i % Kc, v = i / Kc;
        const int x = v % g, y = v / g;
        const size_t vox = (size_t)z * g2 + v;
        if (k >= K) { dhead[((size_t)f * g3 + vox) * Kc + k] = 0.f; dchead_t[((size_t)f * g3 + vox) * Kc + k] = 0.f; continue; }
        const float a = head[((size_t)f * g3 + vox) * Kc + k], c = clip_head[((size_t)b * g3 + vox) * Kc + k];
        const float la = lrelu(a, 0.01f), lc = lrelu(c, 0.01f);
        const float u = w0 * la + w1 * lc + pb;
        const float sig = u > 20.f ? 1.0f : 1.0f / (1.0f + expf(-u));
        const float* cf = coef + ((size_t)f * K + k) * 4;
        const float dhm = cf[0] * lz + cf[1] * lin_coord(y, g) + cf[2] * lin_coord(x, g) + cf[3];
        const float du = dhm * sig;
        dhead[((size_t)f * g3 + vox) * Kc + k] = du * w0 * (a > 0.f ? 1.0f : 0.01f);
        dchead_t[((size_t)f * g3 + vox) * Kc + k] = du * w1 * (c > 0.f ? 1.0f : 0.01f);
        s0 += du * la; s1 += du * lc; s2 += du;
    }
    s0 = block_sum256(s0, sh); s1 = block_sum256(s1, sh); s2 = block_sum256(s2, sh);
    if (threadIdx.x == 0) { float* o = pp + ((size_t)f * g + z) * 3; o[0] = s0; o[1] = s1; o[2] = s2; }
}
// out[b][i] = sum_t in[b*T+t][i]
__global__ __launch_bounds__(256) void sum_t_kernel(const float* __restrict__ in, int B, int T, size_t per, float* __restrict__ out) {
    const size_t total = (size_t)B * per;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / per, r = i % per;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += in[(b * T + t) * per + r];
        out[i] = s;
    }
}

// ---- keypoint-only losses of one clip (kypt_detector_utils.py:105-133,172-265) ------------------------------------------------------
// one block per clip.  dkp[b][t][k][0..2] += gradient;  dinfl[b][k][l] = d loss / d influence[k][l] contribution of this clip
__global__ __launch_bounds__(256) void clip_loss_bwd_kernel(const float* __restrict__ keypoints, const float* __restrict__ affinity,
                                                            const float* __restrict__ dloss, int B, int T, int K, int N, float sep_sigma,
                                                            int use_traj, float* __restrict__ dkp, float* __restrict__ dinfl) {
    extern __shared__ float dyn[];
    const int TK3 = T * K * 3, KK = K * K;
    float* pos = dyn;                 // [T][K][3]
    float* gp = pos + TK3;            // [T][K][3]
    float* gv = gp + TK3;             // [T][K][3]   velocity gradients (t < T-1)
    float* ga = gv + TK3;             // [T][K][3]   acceleration gradients (t < T-2)
    float* mean = ga + TK3;           // [K][3]
    float* infl = mean + K * 3;       // [K][K]
    float* ekl = infl + KK;           // [K][K]  exp(-D_kl / s)
    float* dm = ekl + KK;             // [K][K]  mean_t d_tkl
    float* sbar = dm + KK;            // [K][K]  mean_t sign(d_tkl - dm_kl)
    const int b = blockIdx.x;
    const float g_sep = dloss[3], g_loc = affinity ? dloss[5] : 0.f, g_time = affinity ? dloss[6] : 0.f;
    const float g_traj = (affinity && use_traj) ? dloss[9] : 0.f;
    for (int i = threadIdx.x; i < TK3; i += 256) {
        pos[i] = keypoints[((size_t)b * T * K + i / 3) * 4 + i % 3];
        gp[i] = 0.f; gv[i] = 0.f; ga[i] = 0.f;
    }
    for (int i = threadIdx.x; i < KK; i += 256) {
        float m = 0.f;
        if (affinity) { m = -INFINITY; for (int n = 0; n < N; ++n) m = fmaxf(m, affinity[(size_t)n * KK + i]); }
        infl[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * 3; i += 256) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += pos[t * K * 3 + i];
        mean[i] = s / (float)T;
    }
    __syncthreads();
    const float sden = 2.0f * sep_sigma * sep_sigma;
    const float c_loc = g_loc / ((float)B * (float)T * (float)KK), c_time = g_time / ((float)B * (float)T * (float)KK);
    const float c_vel = T > 1 ? g_traj / ((float)KK * (float)B * (float)(T - 1)) : 0.f;
    const float c_acc = T > 2 ? g_traj / ((float)KK * (float)B * (float)(T - 2)) : 0.f;
    // per pair: separation kernel value, temporal mean of the squared distance, mean sign; and d/d influence
    for (int pr = threadIdx.x; pr < KK; pr += 256) {
        const int k = pr / K, l = pr % K;
        float d2s = 0.f, dsum = 0.f;
        for (int t = 0; t < T; ++t) {
            float s = 0.f, sd = 0.f;
            for (int d = 0; d < 3; ++d) {
                const float a = pos[(t * K + k) * 3 + d], c = pos[(t * K + l) * 3 + d];
                const float u = (a - mean[k * 3 + d]) - (c - mean[l * 3 + d]);
                s += u * u;
                const float w = a - c;
                sd += w * w;
            }
            d2s += s; dsum += sd;
        }
        ekl[pr] = expf(-(d2s / (float)T) / sden);
        const float dmean = dsum / (float)T;
        dm[pr] = dmean;
        float ss = 0.f, tabs = 0.f, velc = 0.f, accc = 0.f;
        for (int t = 0; t < T; ++t) {
            float sd = 0.f;
            for (int d = 0; d < 3; ++d) { const float w = pos[(t * K + k) * 3 + d] - pos[(t * K + l) * 3 + d]; sd += w * w; }
            const float df = sd - dmean;
            ss += df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            tabs += fabsf(df);
        }
        sbar[pr] = ss / (float)T;
        if (g_traj != 0.f) {
            for (int t = 0; t + 1 < T; ++t) {
                float vk[3], vl[3], nk = 0.f, nl = 0.f;
                for (int d = 0; d < 3; ++d) {
                    vk[d] = pos[((t + 1) * K + k) * 3 + d] - pos[(t * K + k) * 3 + d];
                    vl[d] = pos[((t + 1) * K + l) * 3 + d] - pos[(t * K + l) * 3 + d];
                    nk += vk[d] * vk[d]; nl += vl[d] * vl[d];
                }
                nk = fmaxf(sqrtf(nk), 1e-6f); nl = fmaxf(sqrtf(nl), 1e-6f);
                float cs = 0.f;
                for (int d = 0; d < 3; ++d) cs += (vk[d] / nk) * (vl[d] / nl);
                velc += (-cs + 1.0f) / 2.0f;
                if (t + 2 < T) {
                    float ak[3], al[3], mk = 0.f, ml = 0.f;
                    for (int d = 0; d < 3; ++d) {
                        const float vk2 = pos[((t + 2) * K + k) * 3 + d] - pos[((t + 1) * K + k) * 3 + d];
                        const float vl2 = pos[((t + 2) * K + l) * 3 + d] - pos[((t + 1) * K + l) * 3 + d];
                        ak[d] = vk2 - vk[d]; al[d] = vl2 - vl[d];
                        mk += ak[d] * ak[d]; ml += al[d] * al[d];
                    }
                    mk = fmaxf(sqrtf(mk), 1e-6f); ml = fmaxf(sqrtf(ml), 1e-6f);
                    float ca = 0.f;
                    for (int d = 0; d < 3; ++d) ca += (ak[d] / mk) * (al[d] / ml);
                    accc += (-ca + 1.0f) / 2.0f;
                }
            }
        }
        if (dinfl) dinfl[(size_t)b * KK + pr] = c_loc * dsum + c_time * tabs + c_vel * velc + c_acc * accc;
    }
    __syncthreads();
    // position gradients of separation / local / time consistency, and velocity / acceleration gradients
    const float c_sep = g_sep / ((float)B * (float)K * (float)(K - 1));
    for (int i = threadIdx.x; i < T * K; i += 256) {
        const int t = i / K, k = i % K;
        float gr[3] = {0.f, 0.f, 0.f};
        for (int l = 0; l < K; ++l) {
            if (l == k) continue;
            float w[3], sd = 0.f;
            for (int d = 0; d < 3; ++d) { w[d] = pos[(t * K + k) * 3 + d] - pos[(t * K + l) * 3 + d]; sd += w[d] * w[d]; }
            // separation: 2 ordered pairs * dL/dD * (2/T) (u_k - u_l)
            const float cs = 2.0f * (-c_sep * ekl[k * K + l] / sden) * (2.0f / (float)T);
            // local + time: q_tkl + q_tlk with q = dL/dd_tkl
            const float dfk = sd - dm[k * K + l];
            const float sg = dfk > 0.f ? 1.f : (dfk < 0.f ? -1.f : 0.f);
            const float q = (c_loc + c_time * (sg - sbar[k * K + l])) * infl[k * K + l] + (c_loc + c_time * (sg - sbar[l * K + k])) * infl[l * K + k];
            for (int d = 0; d < 3; ++d) {
                const float u = w[d] - (mean[k * 3 + d] - mean[l * 3 + d]);
                gr[d] += cs * u + q * 2.0f * w[d];
            }
        }
        for (int d = 0; d < 3; ++d) gp[(t * K + k) * 3 + d] = gr[d];
        if (g_traj != 0.f && t + 1 < T) {
            float vk[3], nk2 = 0.f;
            for (int d = 0; d < 3; ++d) { vk[d] = pos[((t + 1) * K + k) * 3 + d] - pos[(t * K + k) * 3 + d]; nk2 += vk[d] * vk[d]; }
            const float nkr = sqrtf(nk2), nk = fmaxf(nkr, 1e-6f);
            const bool clamp_k = nkr < 1e-6f;
            float gvv[3] = {0.f, 0.f, 0.f};
            float ak[3], mk2 = 0.f;
            const bool has_a = t + 2 < T;
            if (has_a) for (int d = 0; d < 3; ++d) {
                ak[d] = (pos[((t + 2) * K + k) * 3 + d] - pos[((t + 1) * K + k) * 3 + d]) - vk[d];
                mk2 += ak[d] * ak[d];
            }
            const float mkr = sqrtf(mk2), mk = fmaxf(mkr, 1e-6f);
            const bool clamp_a = mkr < 1e-6f;
            float gaa[3] = {0.f, 0.f, 0.f};
            for (int l = 0; l < K; ++l) {
                const float wi = infl[k * K + l] + infl[l * K + k];
                float vl[3], nl = 0.f;
                for (int d = 0; d < 3; ++d) { vl[d] = pos[((t + 1) * K + l) * 3 + d] - pos[(t * K + l) * 3 + d]; nl += vl[d] * vl[d]; }
                nl = fmaxf(sqrtf(nl), 1e-6f);
                float cs = 0.f;
                for (int d = 0; d < 3; ++d) cs += (vk[d] / nk) * (vl[d] / nl);
                // d cos / d v_k = (vhat_l - cos * vhat_k) / |v_k|   (norm clamped at eps: the clamp has no gradient)
                const float ck = -0.5f * c_vel * wi;
                for (int d = 0; d < 3; ++d) {
                    const float dc = clamp_k ? (vl[d] / nl) / nk : ((vl[d] / nl) - cs * (vk[d] / nk)) / nk;
                    gvv[d] += l == k ? 0.f : ck * dc;      // (the influence has a zero diagonal)
                }
                if (has_a) {
                    float al[3], ml = 0.f;
                    for (int d = 0; d < 3; ++d) {
                        al[d] = (pos[((t + 2) * K + l) * 3 + d] - pos[((t + 1) * K + l) * 3 + d]) - vl[d];
                        ml += al[d] * al[d];
                    }
                    ml = fmaxf(sqrtf(ml), 1e-6f);
                    float ca = 0.f;
                    for (int d = 0; d < 3; ++d) ca += (ak[d] / mk) * (al[d] / ml);
                    const float cka = -0.5f * c_acc * wi;
                    for (int d = 0; d < 3; ++d) {
                        const float dc = clamp_a ? (al[d] / ml) / mk : ((al[d] / ml) - ca * (ak[d] / mk)) / mk;
                        gaa[d] += l == k ? 0.f : cka * dc;
                    }
                }
            }
            for (int d = 0; d < 3; ++d) { gv[(t * K + k) * 3 + d] = gvv[d]; ga[(t * K + k) * 3 + d] = has_a ? gaa[d] : 0.f; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TK3; i += 256) {
        const int t = i / (K * 3), r = i % (K * 3);
        float s = gp[i];
        // v[t] = p[t+1] - p[t];  a[t] = p[t+2] - 2 p[t+1] + p[t]
        if (t >= 1) s += gv[(t - 1) * K * 3 + r];
        if (t + 1 < T) s -= gv[t * K * 3 + r];
        if (t >= 2) s += ga[(t - 2) * K * 3 + r];
        if (t >= 1 && t + 1 < T) s -= 2.0f * ga[(t - 1) * K * 3 + r];
        if (t + 2 < T) s += ga[t * K * 3 + r];
        dkp[((size_t)b * T * K + i / 3) * 4 + i % 3] += s;
    }
}

// single block: d affinity (through the influence = max over neighbours and the neighbour-sparsity loss) -> d affinity_params
// (softmax over K-1 logits with the zero diagonal re-inserted, kypt_detector.py:191-199)
__global__ __launch_bounds__(256) void affinity_bwd_kernel(const float* __restrict__ params, const float* __restrict__ affinity,
                                                           const float* __restrict__ dinfl, const float* __restrict__ dloss, int B, int N,
                                                           int K, int ver, float* __restrict__ dparams) {
    extern __shared__ float dA[];       // [N][K][K]
    const int KK = K * K;
    const float g_spc = dloss[7];
    for (int i = threadIdx.x; i < KK; i += 256) {
        float di = 0.f;
        for (int b = 0; b < B; ++b) di += dinfl[(size_t)b * KK + i];
        int am = 0; float mx = -INFINITY;
        for (int n = 0; n < N; ++n) { const float a = affinity[(size_t)n * KK + i]; if (a > mx) { mx = a; am = n; } }
        for (int n = 0; n < N; ++n) {
            const float an = affinity[(size_t)n * KK + i];
            float others = 0.f;
            for (int m = 0; m < N; ++m) if (m != n) { const float a = affinity[(size_t)m * KK + i]; others += a * a; }
            dA[n * KK + i] = (n == am ? di : 0.f) + g_spc * 4.0f * an * others / (float)KK;
        }
    }
    __syncthreads();
    if (ver != 3) {
        // get_affinity versions 0 / 1 / 2 (kypt_detector.py:173-189; params (N, K, K)); softplus' = sigmoid (1 beyond the threshold 20)
        auto sp = [](float x) { return x > 20.f ? x : log1pf(expf(x)); };
        auto dsp = [](float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); };
        if (ver == 1) {
            // W = M' / (r + eps), M' = (S S^T) with a zero diagonal, S = softplus(params): dM' in place of dA, then dS = (dM' + dM'^T) S
            for (int row = threadIdx.x; row < N * K; row += 256) {
                const int k = row % K;
                const float* p = params + (size_t)row * K; const float* pn = params + (size_t)(row - k) * K;
                float* d = dA + (size_t)row * K;
                float r = 0.f, d1 = 0.f;
                for (int j = 0; j < K; ++j) {
                    if (j == k) continue;
                    float m = 0.f;
                    for (int c = 0; c < K; ++c) m += sp(p[c]) * sp(pn[(size_t)j * K + c]);
                    r += m; d1 += d[j] * m;
                }
                const float inv = 1.f / (r + 1e-6f);
                for (int j = 0; j < K; ++j) d[j] = (j == k) ? 0.f : d[j] * inv - d1 * inv * inv;
            }
            __syncthreads();
            for (int row = threadIdx.x; row < N * K; row += 256) {
                const int k = row % K, n = row / K;
                const float* p = params + (size_t)row * K; const float* pn = params + (size_t)(row - k) * K;
                const float* dn = dA + (size_t)n * KK;
                for (int c = 0; c < K; ++c) {
                    float ds = 0.f;
                    for (int j = 0; j < K; ++j) ds += (dn[k * K + j] + dn[j * K + k]) * sp(pn[(size_t)j * K + c]);
                    dparams[(size_t)row * K + c] = ds * dsp(p[c]);
                }
            }
            return;
        }
        for (int row = threadIdx.x; row < N * K; row += 256) {      // 0: row softmax; 2: softplus, zero diagonal, row softmax
            const int k = row % K;
            const float* p = params + (size_t)row * K; const float* P = affinity + (size_t)row * K; const float* d = dA + (size_t)row * K;
            float dot = 0.f;
            for (int j = 0; j < K; ++j) dot += P[j] * d[j];
            for (int j = 0; j < K; ++j) {
                const float dv = P[j] * (d[j] - dot);
                dparams[(size_t)row * K + j] = ver == 0 ? dv : (j == k ? 0.f : dv * dsp(p[j]));
            }
        }
        return;
    }
    for (int row = threadIdx.x; row < N * K; row += 256) {
        const int k = row % K;
        const float* p = params + (size_t)row * (K - 1);
        float mx = -INFINITY;
        for (int j = 0; j < K - 1; ++j) mx = fmaxf(mx, p[j]);
        float s = 0.f;
        for (int j = 0; j < K - 1; ++j) s += expf(p[j] - mx);
        float dot = 0.f;
        for (int j = 0; j < K - 1; ++j) { const int col = j < k ? j : j + 1; dot += (expf(p[j] - mx) / s) * dA[(size_t)row * K + col]; }
        for (int j = 0; j < K - 1; ++j) {
            const int col = j < k ? j : j + 1;
            const float P = expf(p[j] - mx) / s;
            dparams[(size_t)row * (K - 1) + j] = P * (dA[(size_t)row * K + col] - dot);
        }
    }
}

int grid_for(size_t work_items) { return (int)min((work_items + 255) / 256, (size_t)(256 * 16)); }

}  // namespace

int nm_tail_bwd_blocks(int G) { return (int)(((size_t)G * G * G + NM_TAILB_VPB - 1) / NM_TAILB_VPB); }

int nm_launch_decoder_tail_bwd(const TensorRef& x, const float* w14, const float* target, const float* recon, const float* dloss, int G,
                               float* dA, float* part, hipStream_t s, float* dvout) {
    if (dvout && x.C != 32) { nm_set_error("decoder_tail_bwd: the per-voxel form needs C == 32"); return NM_ERR_ARG; }
    if (x.C % 4 || !x.scale || x.C > 60) { nm_set_error("decoder_tail_bwd: needs a lazy GN input with C %% 4 == 0, C <= 60"); return NM_ERR_ARG; }
    const size_t G3 = (size_t)G * G * G;
    if (x.C == 32) {
        hipLaunchKernelGGL(tail_bwd32_kernel, dim3(nm_tail_bwd_blocks(G), x.N), dim3(256), 0, s, x, w14, target, recon, dloss,
                           1.0f / ((float)x.N * (float)G3), G3, dA, part, dvout);
        return nm_check_hip(hipGetLastError(), "decoder_tail_bwd launch");
    }
    const size_t lds = (size_t)256 * (x.C + 1) * sizeof(float);
    hipLaunchKernelGGL(tail_bwd_kernel, dim3(nm_tail_bwd_blocks(G), x.N), dim3(256), lds, s, x, w14, target, recon, dloss,
                       1.0f / ((float)x.N * (float)G3), G3, dA, part);
    return nm_check_hip(hipGetLastError(), "decoder_tail_bwd launch");
}

int nm_launch_sum_rows(const float* part, int rows, int cols, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_rows_kernel, dim3(cols), dim3(256), 0, s, part, rows, cols, out);
    return nm_check_hip(hipGetLastError(), "sum_rows launch");
}

int nm_chamfer_bwd_blocks(int G) { return (int)(((size_t)G * G * G + NM_CHAM_VPB - 1) / NM_CHAM_VPB); }

int nm_launch_chamfer_bwd(const float* target, const float* keypoints, const float* tail_part, int tail_blocks, const float* dloss, int F,
                          int K, int G, float* ws, float* dkp, hipStream_t s) {
    const int nblk = nm_chamfer_bwd_blocks(G);
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(nblk, F), dim3(256), 0, s, target, keypoints, K, G, ws);
    hipLaunchKernelGGL(chamfer_bwd_finish_kernel, dim3(F), dim3(128), 0, s, ws, nblk, tail_part, tail_blocks, dloss, F, K, dkp);
    return nm_check_hip(hipGetLastError(), "chamfer_bwd launch");
}

size_t nm_volfit_gauss_bwd_ws_floats(int B, int T, int G) { return (size_t)B * T * G * G + (size_t)B * T * G + (size_t)T * G * G + 64; }
int nm_launch_volfit_gauss_bwd(const float* vox, const float* keypoints, const float* dloss, int B, int T, int K, int G, float sigma,
                               float* ws, float* dkp, hipStream_t s) {
    if (G > 1024) { nm_set_error("volfit_gauss_bwd: G %d unsupported", G); return NM_ERR_UNSUPPORTED; }
    const int F = B * T;
    float* col = ws; float* rowsum = col + (size_t)F * G * G; float* wsum = rowsum + (size_t)F * G;
    const float width = (float)(2.0 * std::pow((double)sigma * 4.0 / (double)G, 2.0));
    hipLaunchKernelGGL(volfit_proj_b_kernel, dim3(G, F), dim3(G), G * sizeof(float), s, vox, G, col, rowsum);
    hipLaunchKernelGGL(volfit_wsum_kernel, dim3(G, T), dim3(G), 0, s, col, rowsum, B, T, G, wsum);
    hipLaunchKernelGGL(volfit_bwd_kernel, dim3(F * K), dim3(256), 0, s, keypoints, wsum, dloss, B, T, K, G, width, dkp);
    return nm_check_hip(hipGetLastError(), "volfit_gauss_bwd launch");
}

int nm_launch_combined_bwd(const float* dcomb, int Cd, const float* table, const float* keypoints, int B, int T, int K, int Fd, int g,
                           float width, float* ws, float* dfeat, float* dkp, hipStream_t s, int cat, const float* widthk,
                           const float* sigma_param, float max_sigma, float* dsigma_param) {
    const int F = B * T, g3 = g * g * g;
    float* dwp = widthk ? ws + (size_t)F * K * 8 : nullptr;             // [F][K][2] behind the [F][K][2][4] partials (the caller sizes ws for both)
    hipLaunchKernelGGL(gauss_bwd_kernel, dim3(F * K, 2), dim3(256), 0, s, dcomb, Cd, table, keypoints, T, K, Fd, g, width, cat, widthk, dwp, ws);
    if (widthk) hipLaunchKernelGGL(sigma_bwd_finish_kernel, dim3((K + 63) / 64), dim3(64), 0, s, dwp, sigma_param, F, K, max_sigma, g, dsigma_param);
    hipLaunchKernelGGL(gauss_bwd_finish_kernel, dim3((F * K * 4 + 255) / 256), dim3(256), 0, s, ws, F, T, K, dkp);
    hipLaunchKernelGGL(first_feature_bwd_kernel, dim3(grid_for((size_t)B * g3 * (Fd / 4))), dim3(256), 0, s, dcomb, Cd, B, T, K, Fd, g3, dfeat);
    return nm_check_hip(hipGetLastError(), "combined_bwd launch");
}

int nm_launch_heat_bwd(const float* head, const float* clip_head, const float* prop, const float* heat_part, const float* heat_mean,
                       const float* keypoints, const float* dkp, const float* dloss, int B, int T, int K, int Kc, int g, float* ws, float* dhead,
                       float* dchead_t, float* dclip_head, float* dprop, hipStream_t s) {
    const int F = B * T;
    if (K < 1 || K > 32 || Kc < K) { nm_set_error("heat_bwd: K=%d (row pitch %d) unsupported", K, Kc); return NM_ERR_ARG; }
    float* coef = ws;                         // [F][K][4]
    float* pp = ws + (size_t)F * K * 4;       // [F*g][3]
    hipLaunchKernelGGL(heat_bwd_prep_kernel, dim3(F), dim3(64), 0, s, heat_part, heat_mean, keypoints, dkp, dloss, F, K, g, coef);
    hipLaunchKernelGGL(heat_bwd_kernel, dim3(F, g), dim3(256), 0, s, head, clip_head, prop, coef, T, K, Kc, g, dhead, dchead_t, pp);
    const size_t per = (size_t)g * g * g * Kc;
    hipLaunchKernelGGL(sum_t_kernel, dim3(grid_for((size_t)B * per)), dim3(256), 0, s, dchead_t, B, T, per, dclip_head);
    hipLaunchKernelGGL(sum_rows_kernel, dim3(3), dim3(256), 0, s, pp, F * g, 3, dprop);
    return nm_check_hip(hipGetLastError(), "heat_bwd launch");
}
size_t nm_heat_bwd_ws_floats(int F, int K, int g) { return (size_t)F * K * 4 + (size_t)F * g * 3 + 64; }

int nm_launch_clip_loss_bwd(const float* keypoints, const float* affinity, const float* dloss, int B, int T, int K, int N, float sep_sigma,
                            int use_traj, float* dkp, float* dinfl, hipStream_t s) {
    const size_t lds = ((size_t)4 * T * K * 3 + K * 3 + 4 * K * K) * sizeof(float);
    if (lds > 150 * 1024) { nm_set_error("clip_loss_bwd: T*K too large (%d x %d)", T, K); return NM_ERR_UNSUPPORTED; }
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(clip_loss_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
            nm_set_error("clip_loss_bwd: cannot raise the dynamic LDS limit"); return NM_ERR_HIP;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL(clip_loss_bwd_kernel, dim3(B), dim3(256), lds, s, keypoints, affinity, dloss, B, T, K, N, sep_sigma, use_traj, dkp,
                       affinity ? dinfl : nullptr);
    return nm_check_hip(hipGetLastError(), "clip_loss_bwd launch");
}

int nm_launch_affinity_bwd(const float* params, const float* affinity, const float* dinfl, const float* dloss, int B, int N, int K,
                           float* dparams, hipStream_t s, int ver) {
    hipLaunchKernelGGL(affinity_bwd_kernel, dim3(1), dim3(256), (size_t)N * K * K * sizeof(float), s, params, affinity, dinfl, dloss, B, N, K,
                       ver, dparams);
    return nm_check_hip(hipGetLastError(), "affinity_bwd launch");
}
