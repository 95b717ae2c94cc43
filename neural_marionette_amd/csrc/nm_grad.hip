// Backward kernels of the conv stack (detector-mode training step; reference: autograd of
// modules/vox_modules.py:8-120 and model/kypt_detector.py:417-460 as driven by train.py:388-404).
//
//   wgrad_kernel         dW[m][c][tap] = sum_{n,v} dy[n,v,m] * act(in)[n, s*v + tap - pad, c]
//                        An implicit GEMM whose K dimension is the voxels: M = 32 dy channels, N = 32 input channels per
//                        workgroup, one 32x32 fp32-MFMA accumulator tile per tap kept in registers (taps dealt to the four
//                        waves) while the workgroup walks its share of the bricks; the per-workgroup tiles go to a
//                        scratch buffer and are summed in a fixed order (run-to-run identical, no float atomics).
//                        v_mfma_f32_32x32x2_f32 takes one K element per lane, so both operands are read from a plain
//                        [voxel][channel] LDS tile with conflict-free ds_read_b32 - no transposition of the
//                        channels-last activations is needed.  Bound: fp32 MFMA (157 TFLOP/s), the same FLOPs as the
//                        forward conv.
//   gnb_*                GroupNorm + LeakyReLU backward as two passes over (dA, y): per-block partial sums, per-(frame,
//                        group) coefficients, then dy = c1*dz + c2*y + c3.   HBM-bound.
//   upsample2_adjoint    adjoint of the trilinear x2 upsampling (gather form, deterministic).  HBM-bound.
#include "nm_grad.h"
#include <cstdlib>
#include <utility>

namespace {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float lin_coord(int i, int G) {
    const float step = 2.0f / (float)(G - 1);
    return i < G / 2 ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(G - 1 - i), 1.0f);
}

__device__ __forceinline__ f32x4 load_act4(const TensorRef& t, int n, int z, int y, int x, int c) {
    f32x4 v = *reinterpret_cast<const f32x4*>(t.p + ((((size_t)n * t.D + z) * t.H + y) * t.W + x) * t.C + c);
    if (t.scale) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(t.scale + (size_t)n * t.C + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(t.shift + (size_t)n * t.C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
    }
    if (t.slope != 1.0f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = lrelu(v[j], t.slope);
    }
    return v;
}

struct WgradParams {
    TensorRef in;            // MODE 0/1: activated on read; MODE 2: in.p = occupancy [N][G^3], in.D/H/W = G
    TensorRef dy;            // [N][OD][OH][OW][dy.C] (lazy affine allowed)
    float* part;
    int ks, stride, pad;
    int M, Nc;               // dy channels / input channels that take part
    int BZ, BY, BX, nbz, nby, nbx, HZ, HY, HX;
    int S, n_tiles, groups;  // split count, column tiles, tap groups per tile
    int dbg;                 // diagnostics (NM355_W16_DBG): 1 no MFMA phase, 2 no global loads, 3 no loads / LDS stores, 4 LDS stores only
};

// MODE 0: taps dealt to the waves (NTW per wave);  MODE 1: ks = 1, the waves split the voxels;
// MODE 2: first layer, columns = (dx, channel) of cat[occ, x1, x2, x3], tap groups = (dz, dy)
// HX / HD: the input / dY tensor is stored as bfloat16 (16-bit storage mode; template parameters, not run-time flags: a load under a
// uniform branch ends the batch of loads it belongs to)
template <int MODE, int NTW, bool HX = false, bool HD = false>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
    extern __shared__ float lds[];
    constexpr int CP = MODE == 2 ? 4 : 32;
    const int BV = p.BZ * p.BY * p.BX, HV = p.HZ * p.HY * p.HX;
    float* dys = lds;                    // [BV][32]
    float* as = lds + (size_t)BV * 32;   // [HV][CP]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, w = tid >> 6;
    const int tp = blockIdx.y, mt = tp / p.n_tiles, nt = tp % p.n_tiles, m0 = mt * 32, n0 = nt * 32;
    const int OD = p.dy.D, OH = p.dy.H, OW = p.dy.W;

    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int toff[NTW];
    bool tv[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int t = MODE == 1 ? 0 : w + 4 * j;
        tv[j] = t < p.groups;
        if (MODE == 2) toff[j] = ((t / 5) * p.HY + (t % 5)) * p.HX * CP;
        else { const int dz = t / (p.ks * p.ks), dyy = (t / p.ks) % p.ks, dx = t % p.ks; toff[j] = ((dz * p.HY + dyy) * p.HX + dx) * CP; }
        if (!tv[j]) toff[j] = 0;             // a tap slot beyond the layer's taps: multiplied like the others (no branch in the k-loop), never stored
    }
    const int coloff = MODE == 2 ? (l31 < 20 ? l31 : 0) : l31;

    const int per_frame = p.nbz * p.nby * p.nbx;
    const int total = p.in.N * per_frame;
    // Staging.  (1) The loads of a batch are all issued before the first of them is activated and written to LDS: one item per
    // iteration (load -> affine -> LDS store) made every item a full memory round trip.  (2) What does not depend on the brick is
    // computed once per thread: an item's position in the brick / halo (packed x | y << 8 | z << 16, bit 31 = the item exists) and its
    // element offset from the brick's first voxel - the div / mod chains of the item index were ~80 VALU instructions per item and
    // brick, more than the loads themselves (32 -> 32 pool layer: 1.45 -> 0.95 ms).  Batch sizes keep the kernel under 256 registers
    // (two workgroups per CU: larger batches, or the next brick prefetched into registers, measured 1.4-2x slower at one workgroup).
    // A thread's channel quad (i & 7) is the same for all its items.
    constexpr int DU = 8;                                 // dY items per thread: BV * 8 / 256, BV <= 256
    constexpr int AU = MODE == 2 ? 5 : 20;                // input items per thread: HV * 8 / 256 <= 18.75 (k3 halo of a 4x8x8 brick); MODE 2: HV / 256
    const int q = tid & 7;
    int d_pk[DU], d_rel[DU], a_pk[AU], a_rel[AU];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
        const int k = (tid + 256 * u) >> 3;
        const int x = k % p.BX, y = (k / p.BX) % p.BY, z = k / (p.BX * p.BY);
        d_pk[u] = k < BV ? (x | (y << 8) | (z << 16) | (1 << 31)) : 0;
        d_rel[u] = ((z * OH + y) * OW + x) * p.dy.C + m0 + 4 * q;
    }
#pragma unroll
    for (int u = 0; u < AU; ++u) {
        const int hv = MODE == 2 ? tid + 256 * u : (tid + 256 * u) >> 3;
        const int hx = hv % p.HX, hy = (hv / p.HX) % p.HY, hz = hv / (p.HX * p.HY);
        a_pk[u] = hv < HV ? (hx | (hy << 8) | (hz << 16) | (1 << 31)) : 0;
        a_rel[u] = MODE == 2 ? (hz * p.in.H + hy) * p.in.W + hx : ((hz * p.in.H + hy) * p.in.W + hx) * p.in.C + n0 + 4 * q;
    }
    for (int b = blockIdx.x; b < total; b += p.S) {
        const int n = b / per_frame; int r = b % per_frame;
        const int bx = r % p.nbx; r /= p.nbx;
        const int by = r % p.nby, bz = r / p.nby;
        const int oz0 = bz * p.BZ, oy0 = by * p.BY, ox0 = bx * p.BX;
        __syncthreads();
        {
            const int m = m0 + 4 * q;
            f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.dy.scale && m < p.M) { sc = *reinterpret_cast<const f32x4*>(p.dy.scale + (size_t)n * p.dy.C + m); sh = *reinterpret_cast<const f32x4*>(p.dy.shift + (size_t)n * p.dy.C + m); }
            const size_t base = ((((size_t)n * OD + oz0) * OH + oy0) * OW + ox0) * p.dy.C;      // element offset
            constexpr int U = 4;
#pragma unroll
            for (int u0 = 0; u0 < DU; u0 += U) {
                if (u0 * 256 >= BV * 8) break;            // (uniform)
                f32x4 v[U]; bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pk = d_pk[u0 + u];
                    ok[u] = pk < 0 && oz0 + ((pk >> 16) & 255) < OD && oy0 + ((pk >> 8) & 255) < OH && ox0 + (pk & 255) < OW && m < p.M;
                    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok[u]) v[u] = nm_ld4<HD>(p.dy.p, base + d_rel[u0 + u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (d_pk[u0 + u] >= 0) continue;
                    const int k = (tid + 256 * (u0 + u)) >> 3;
                    f32x4 t = v[u];
                    if (ok[u]) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { t[j] = fmaf(t[j], sc[j], sh[j]); if (p.dy.slope != 1.0f) t[j] = lrelu(t[j], p.dy.slope); }
                    }
                    *reinterpret_cast<f32x4*>(dys + k * 32 + 4 * q) = t;
                }
            }
        }
        const int iz0 = oz0 * p.stride - p.pad, iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
        if (MODE == 2) {
            const int G = p.in.D;
            // (signed 64-bit: the brick's first halo voxel can lie before the tensor)
            const float* base = p.in.p + (long long)n * G * G * G + ((long long)iz0 * G + iy0) * G + ix0;
            float o[AU]; bool ok[AU];
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                const int pk = a_pk[u];
                const int gz = iz0 + ((pk >> 16) & 255), gy = iy0 + ((pk >> 8) & 255), gx = ix0 + (pk & 255);
                ok[u] = pk < 0 && (unsigned)gz < (unsigned)G && (unsigned)gy < (unsigned)G && (unsigned)gx < (unsigned)G;
                o[u] = ok[u] ? base[a_rel[u]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                const int pk = a_pk[u];
                if (pk >= 0) continue;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok[u]) { v[0] = o[u]; v[1] = lin_coord(iz0 + ((pk >> 16) & 255), G); v[2] = lin_coord(iy0 + ((pk >> 8) & 255), G); v[3] = lin_coord(ix0 + (pk & 255), G); }
                *reinterpret_cast<f32x4*>(as + (tid + 256 * u) * 4) = v;
            }
        } else {
            const int c = n0 + 4 * q;
            f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.in.scale && c < p.Nc) { sc = *reinterpret_cast<const f32x4*>(p.in.scale + (size_t)n * p.in.C + c); sh = *reinterpret_cast<const f32x4*>(p.in.shift + (size_t)n * p.in.C + c); }
            const long long base = ((((long long)n * p.in.D + iz0) * p.in.H + iy0) * p.in.W + ix0) * (long long)p.in.C;   // element offset (may be negative)
            constexpr int U = 4;
#pragma unroll
            for (int u0 = 0; u0 < AU; u0 += U) {
                if (u0 * 256 >= HV * 8) break;            // (uniform)
                f32x4 v[U]; bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ui = u0 + u < AU ? u0 + u : AU - 1;
                    const int pk = u0 + u < AU ? a_pk[ui] : 0;
                    const int gz = iz0 + ((pk >> 16) & 255), gy = iy0 + ((pk >> 8) & 255), gx = ix0 + (pk & 255);
                    ok[u] = pk < 0 && (unsigned)gz < (unsigned)p.in.D && (unsigned)gy < (unsigned)p.in.H && (unsigned)gx < (unsigned)p.in.W && c < p.Nc;
                    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok[u]) v[u] = nm_ld4<HX>(p.in.p, (size_t)(base + a_rel[ui]));
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u0 + u >= AU || a_pk[u0 + u < AU ? u0 + u : 0] >= 0) continue;
                    const int hv = (tid + 256 * (u0 + u)) >> 3;
                    f32x4 t = v[u];
                    if (ok[u]) {                                                     // zero padding, not act(0)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { t[j] = fmaf(t[j], sc[j], sh[j]); if (p.in.slope != 1.0f) t[j] = lrelu(t[j], p.in.slope); }
                    }
                    *reinterpret_cast<f32x4*>(as + hv * 32 + 4 * q) = t;
                }
            }
        }
        __syncthreads();
        int pair = 0;
        for (int z = 0; z < p.BZ; ++z)
            for (int y = 0; y < p.BY; ++y)
                for (int x0 = 0; x0 < p.BX; x0 += 2, ++pair) {
                    if (MODE == 1 && (pair & 3) != w) continue;
                    const int k = (z * p.BY + y) * p.BX + x0 + lh;
                    const float a = dys[k * 32 + l31];
                    const int hb = (((z * p.stride) * p.HY + y * p.stride) * p.HX + (x0 + lh) * p.stride) * CP + coloff;
                    // (no test of tv[j] here: a branch around the MFMA made the compiler move the 16 accumulator registers in and out
                    //  of every iteration)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, as[hb + toff[j]], acc[j], 0, 0, 0);
                }
    }
    const int slot = MODE == 1 ? blockIdx.x * 4 + w : blockIdx.x;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        if (!tv[j]) continue;
        const int t = MODE == 1 ? 0 : w + 4 * j;
        float* dst = p.part + (((size_t)slot * gridDim.y + tp) * p.groups + t) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r >> 2) * 8 + lh * 4 + (r & 3)) * 32 + l31] = acc[j][r];
    }
}

// ---- 1x1x1 convolutions: every (m, n) tile of the layer in ONE workgroup -------------------------------------------------------
// dW[co][ci] = sum_v dY[v][co] X[v][ci] is a GEMM with a huge K (all voxels) and a small M x N (at most 128 x 192 here).  wgrad_kernel
// gives each 32 x 32 tile its own workgroups, every one staging its own 32-channel slices of the same voxels: the 179 -> 128 layer
// (24 tiles) read its inputs ~5 times through L2 and spent 1.07 ms on 0.32 GB.  Here a workgroup stages ALL channels of a 64-voxel
// brick once ([64][Cout] + [64][Cin] floats, activated) and its four waves share the tiles (tile t = w + 4 j, NTW accumulators per
// wave) - v_mfma_f32_32x32x2_f32, both operands single floats from LDS.  Persistent workgroups, partials in wgrad_kernel's MODE 1
// layout (the same reduce).
template <int NTW, bool H16 = false>       // H16: both tensors bfloat16 (16-bit storage mode)
__global__ __launch_bounds__(256) void wgrad_k1_kernel(WgradParams p, int m_tiles, int n_tiles) {
    extern __shared__ float lds[];
    const int Mp = m_tiles * 32 + 4, Np = n_tiles * 32 + 4;          // row pitches (+4: the two lane halves on different banks)
    float* dys = lds;                     // [64][Mp]
    float* as = lds + 64 * Mp;            // [64][Np]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, w = tid >> 6;
    const int T = m_tiles * n_tiles;
    const int V = p.dy.D * p.dy.H * p.dy.W, bpf = V / 64, total = p.in.N * bpf;
    int aoff[NTW], boff[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) { const int t = min(w + 4 * j, T - 1); aoff[j] = 32 * (t / n_tiles) + l31; boff[j] = 32 * (t % n_tiles) + l31; }
    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int mq = m_tiles * 8, nq = n_tiles * 8;                      // 16-byte items per voxel row
    for (int b = blockIdx.x; b < total; b += gridDim.x) {
        const int n = b / bpf, v0 = (b % bpf) * 64;
        __syncthreads();                  // the previous brick's operand reads
        // stage both tiles (batches of four 16-byte loads; a row's items are contiguous in memory up to the real channel count)
        auto stage = [&](const TensorRef& t, int real_c, int rowq, int pitch, float* dst) {
            const size_t src = ((size_t)n * V + v0) * t.C;             // element offset
            const int items = 64 * rowq;
            for (int i0 = tid; i0 < items; i0 += 256 * 4) {
                f32x4 v[4]; int vox[4], c[4]; bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + 256 * u;
                    vox[u] = i / rowq; c[u] = (i % rowq) * 4;
                    ok[u] = i < items && c[u] < real_c && c[u] < t.C;
                    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok[u]) v[u] = nm_ld4<H16>(t.p, src + (size_t)vox[u] * t.C + c[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (i0 + 256 * u >= items) continue;
                    f32x4 x = v[u];
                    if (ok[u]) {
                        if (t.scale) {
                            const f32x4 sc = *reinterpret_cast<const f32x4*>(t.scale + (size_t)n * t.C + c[u]), sh = *reinterpret_cast<const f32x4*>(t.shift + (size_t)n * t.C + c[u]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] = fmaf(x[e], sc[e], sh[e]);
                        }
                        if (t.slope != 1.0f) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] = lrelu(x[e], t.slope);
                        }
                    }
                    *reinterpret_cast<f32x4*>(dst + vox[u] * pitch + c[u]) = x;
                }
            }
        };
        stage(p.dy, p.M, mq, Mp, dys);
        stage(p.in, p.Nc, nq, Np, as);
        __syncthreads();
#pragma unroll 4
        for (int pr = 0; pr < 32; ++pr) {
            const int v = 2 * pr + lh;
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[v * Mp + aoff[j]], as[v * Np + boff[j]], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int t = w + 4 * j;
        if (t >= T) continue;
        float* dst = p.part + ((size_t)blockIdx.x * T + t) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r >> 2) * 8 + lh * 4 + (r & 3)) * 32 + l31] = acc[j][r];
    }
}

// ---- split-fp16 weight gradient for the 3x3x3 stride-1 convolutions (the bulk of the backward FLOPs) ---------------------------
// Same implicit GEMM (K = voxels) on v_mfma_f32_32x32x16_f16 with every fp32 operand split into fp16 hi + lo * 2^-11 and three
// products hi*hi + 2^-11 (hi*lo + lo*hi), fp32 accumulate (the forward's arithmetic, nm_conv.hip).  This MFMA wants 8
// consecutive K elements per lane, i.e. 8 voxels of one channel, so the staging transposes the channels-last tiles into
// [channel][row][x] fp16 planes (one row = the brick's 8 x positions, 10 with the halo).  The tap's x shift (dx = 0, 1, 2) is
// taken in registers: a lane reads the 10 halves of its halo row once (16 B + 4 B, aligned) and forms the three shifted
// operands with v_alignbit, so one pair of LDS reads serves three taps; the nine (dz, dy) row groups are dealt to the four waves
// (3, 2, 2, 2).  Brick 2 x 8 x 8 voxels; LDS 98 KB: A planes [32 ch][40 rows][32 B] + dY planes [32 ch][16 rows][16 B], channel
// pitch + 16 B (conflict-free 16-B operand reads).  dy arrives pre-scaled by a power of two (DyScale) - gradients sit below the
// fp16 normal range otherwise - and the reduce multiplies by its inverse.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define W16_PA (40 * 32 + 16)
#define W16_PD (16 * 16 + 16)
#define W16_LDS (2 * 32 * W16_PA + 2 * 32 * W16_PD)
#define W16_SPLIT 2048.0f

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (v0, v1) -> packed fp16 hi pair (v_cvt_pk_f16_f32) and packed lo pair, lo = fp16((v - hi) * 2^11) as one fma each (v_fma_mix)
__device__ __forceinline__ unsigned pack_split(float v0, float v1, unsigned& lo_out) {
    half2v hh = __builtin_convertvector(f32x2{v0, v1}, half2v);
    asm volatile("" : "+v"(hh));                 // keep the packed pair (no per-half re-conversion)
    half2v ll;
    ll[0] = (_Float16)__builtin_fmaf((float)hh[0], -W16_SPLIT, v0 * W16_SPLIT);
    ll[1] = (_Float16)__builtin_fmaf((float)hh[1], -W16_SPLIT, v1 * W16_SPLIT);
    lo_out = __builtin_bit_cast(unsigned, ll);
    return __builtin_bit_cast(unsigned, hh);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one staging task: the 10 (halo row) or 8 (dY row) voxels x 4 channels a thread fetches for the next brick
struct W16Task { const float* p; const float* sc; const float* sh; float slope; int stride; unsigned mask; };

__device__ __forceinline__ void w16_fetch(const W16Task& t, f32x4 (&v)[10]) {
#pragma unroll
    for (int x = 0; x < 10; ++x) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if ((t.mask >> x) & 1u) r = *reinterpret_cast<const f32x4*>(t.p + (size_t)x * t.stride);
        v[x] = r;
    }
}
// activation + fp16 hi/lo split + 16-B rows into the hi / lo planes
template <int NX>
__device__ __forceinline__ void w16_store(const W16Task& t, const f32x4 (&v)[10], char* hi_plane, char* lo_plane, int pitch, int q, int row_off) {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (t.sc) { sc = *reinterpret_cast<const f32x4*>(t.sc); sh = *reinterpret_cast<const f32x4*>(t.sh); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hi[NX / 2], lo[NX / 2];
#pragma unroll
        for (int i = 0; i < NX / 2; ++i) {
            float a0 = v[2 * i][j], a1 = v[2 * i + 1][j];
            if (t.sc) { a0 = fmaf(a0, sc[j], sh[j]); a1 = fmaf(a1, sc[j], sh[j]); }
            if (t.slope != 1.0f) { a0 = fmaxf(a0, a0 * t.slope); a1 = fmaxf(a1, a1 * t.slope); }
            if (!((t.mask >> (2 * i)) & 1u)) a0 = 0.f;             // outside the volume: zero padding, not act(0)
            if (!((t.mask >> (2 * i + 1)) & 1u)) a1 = 0.f;
            hi[i] = pack_split(a0, a1, lo[i]);
        }
        char* dh = hi_plane + (4 * q + j) * pitch + row_off; char* dl = lo_plane + (4 * q + j) * pitch + row_off;
        *reinterpret_cast<u32x4*>(dh) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(dl) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        if (NX == 10) { *reinterpret_cast<unsigned*>(dh + 16) = hi[4]; *reinterpret_cast<unsigned*>(dl + 16) = lo[4]; }
    }
}

// raw LDS words of one (dz, dy) row group for one k-step: 10 halves of the hi and lo planes
struct W16Raw { u32x4 qh, ql; unsigned eh, el; };
__device__ __forceinline__ W16Raw w16_read(const char* bh, const char* bl, int off) {
    W16Raw r;
    r.qh = *reinterpret_cast<const u32x4*>(bh + off); r.eh = *reinterpret_cast<const unsigned*>(bh + off + 16);
    r.ql = *reinterpret_cast<const u32x4*>(bl + off); r.el = *reinterpret_cast<const unsigned*>(bl + off + 16);
    return r;
}
template <int DX>
__device__ __forceinline__ half8 w16_shift(const u32x4& q, unsigned e) {
    if (DX == 0) return __builtin_bit_cast(half8, q);
    if (DX == 2) return __builtin_bit_cast(half8, u32x4{q[1], q[2], q[3], e});
    return __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                                           __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(e, q[3], 16)});
}
#define W16_MMA(ACC, ACCL, AH, AL, BH, BL)                                   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0);      \
    ACCL = nm_mfma_lo<SINGLE>(AH, BL, ACCL);                                 \
    ACCL = nm_mfma_lo<SINGLE>(AL, BH, ACCL);

// Taps per wave: two full (dz, dy) row groups {w, w + 4} (3 dx taps each, one LDS read pair per group) and, for waves 1..3, the tap
// dx = w - 1 of the ninth group: 6, 7, 7, 7 taps.  The global loads of the next brick are issued before the MFMA phase of the
// current one (the only wave on its SIMD has nothing else to hide them behind) and converted / written to LDS after it.
template <bool SINGLE>
__global__ __launch_bounds__(256) void wgrad16_kernel(WgradParams p) {
    extern __shared__ char lds8[];
    char* A_hi = lds8; char* A_lo = A_hi + 32 * W16_PA; char* D_hi = A_lo + 32 * W16_PA; char* D_lo = D_hi + 32 * W16_PD;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, w = tid >> 6;
    const int tp = blockIdx.y, mt = tp / p.n_tiles, nt = tp % p.n_tiles, m0 = mt * 32, n0 = nt * 32;
    f32x16 acc[7], accl[7];           // [0..2] group w, [3..5] group w + 4, [6] the single tap of group 8
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accl[j][r] = 0.f; }
    const int g0 = w, g1 = w + 4;
    const int off0 = ((g0 / 3) * 10 + (g0 % 3)) * 32, off1 = ((g1 / 3) * 10 + (g1 % 3)) * 32, off2 = (2 * 10 + 2) * 32;

    const int per_frame = p.nbz * p.nby * p.nbx;
    const int total = p.in.N * per_frame;
    // staging roles: every thread one halo row task (rows 0..31 x 8 channel quads); second slot: wave 0 the halo rows 32..39,
    // waves 1-2 the 16 dY rows, wave 3 none
    const int qa = tid & 7, rowa = tid >> 3;
    const int rowb = w == 0 ? 32 + (tid >> 3) : ((tid - 64) >> 3);
    W16Task ta, tb;
    f32x4 va[10], vb[10];
    auto setup = [&](int b) __attribute__((always_inline)) {
        const int n = b / per_frame; int r = b % per_frame;
        const int bx = r % p.nbx; r /= p.nbx;
        const int by = r % p.nby, bz = r / p.nby;
        const int oz0 = bz * 2, oy0 = by * 8, ox0 = bx * 8;
        auto halo_task = [&](int row, W16Task& t) __attribute__((always_inline)) {
            const int c = n0 + 4 * qa;
            const int gz = oz0 - 1 + row / 10, gy = oy0 - 1 + row % 10;
            const bool rin = (unsigned)gz < (unsigned)p.in.D && (unsigned)gy < (unsigned)p.in.H && c < p.Nc;
            unsigned mask = 0;
#pragma unroll
            for (int x = 0; x < 10; ++x) if (rin && (unsigned)(ox0 - 1 + x) < (unsigned)p.in.W) mask |= 1u << x;
            t.mask = mask; t.stride = p.in.C; t.slope = p.in.slope;
            t.p = p.in.p + ((((size_t)n * p.in.D + (rin ? gz : 0)) * p.in.H + (rin ? gy : 0)) * p.in.W + (ox0 - 1)) * p.in.C + (rin ? c : 0);
            t.sc = p.in.scale ? p.in.scale + (size_t)n * p.in.C + (rin ? c : 0) : nullptr;
            t.sh = p.in.scale ? p.in.shift + (size_t)n * p.in.C + (rin ? c : 0) : nullptr;
        };
        halo_task(rowa, ta);
        if (w == 0) halo_task(rowb, tb);
        else if (w < 3) {
            const int m = m0 + 4 * qa;
            const bool in = m < p.M;
            tb.mask = in ? 0xffu : 0u; tb.stride = p.dy.C; tb.slope = p.dy.slope;
            tb.p = p.dy.p + ((((size_t)n * p.dy.D + oz0 + (rowb >> 3)) * p.dy.H + oy0 + (rowb & 7)) * p.dy.W + ox0) * p.dy.C + (in ? m : 0);
            tb.sc = p.dy.scale ? p.dy.scale + (size_t)n * p.dy.C + (in ? m : 0) : nullptr;
            tb.sh = p.dy.scale ? p.dy.shift + (size_t)n * p.dy.C + (in ? m : 0) : nullptr;
        } else tb.mask = 0;
    };
    int b = blockIdx.x;
    if (b < total) {
        setup(b); w16_fetch(ta, va); if (w < 3) w16_fetch(tb, vb);
    }
    const char* ah_base = D_hi + l31 * W16_PD; const char* al_base = D_lo + l31 * W16_PD;
    const char* bh_base = A_hi + l31 * W16_PA; const char* bl_base = A_lo + l31 * W16_PA;
    for (; b < total; b += p.S) {
        __syncthreads();
        if (!(p.dbg == 3 && b != (int)blockIdx.x)) {
            w16_store<10>(ta, va, A_hi, A_lo, W16_PA, qa, rowa * 32);
            if (w == 0) w16_store<10>(tb, vb, A_hi, A_lo, W16_PA, qa, rowb * 32);
            else if (w < 3) w16_store<8>(tb, vb, D_hi, D_lo, W16_PD, qa, rowb * 16);
        }
        __syncthreads();
        const bool has_next = b + p.S < total;
        if (has_next && p.dbg < 2) { setup(b + p.S); w16_fetch(ta, va); if (w < 3) w16_fetch(tb, vb); }
        if (p.dbg == 1 || p.dbg == 4) continue;
        // k-steps of 16 voxels (two brick rows).  LDS reads run one row group ahead of the MFMAs that consume them: group 1 is
        // requested before the 9 MFMAs of group 0, group 2 / the next step's dY operand / the next step's group 0 before those of
        // group 1 (a wave is alone on its SIMD: nothing else covers the LDS latency).
        auto row_off = [&](int s8) __attribute__((always_inline)) { const int row = 2 * s8 + lh; return ((row >> 3) * 10 + (row & 7)) * 32; };
        half8 ah = *reinterpret_cast<const half8*>(ah_base + lh * 16), al = *reinterpret_cast<const half8*>(al_base + lh * 16);
        W16Raw r0 = w16_read(bh_base, bl_base, row_off(0) + off0), r1, r2;
        auto kstep = [&](int s8, int chunk) __attribute__((always_inline)) {
            const int brow = row_off(s8);
            r1 = w16_read(bh_base, bl_base, brow + off1);
            __builtin_amdgcn_sched_barrier(0);
            W16_MMA(acc[0], accl[0], ah, al, w16_shift<0>(r0.qh, r0.eh), w16_shift<0>(r0.ql, r0.el))
            W16_MMA(acc[1], accl[1], ah, al, w16_shift<1>(r0.qh, r0.eh), w16_shift<1>(r0.ql, r0.el))
            W16_MMA(acc[2], accl[2], ah, al, w16_shift<2>(r0.qh, r0.eh), w16_shift<2>(r0.ql, r0.el))
            __builtin_amdgcn_sched_barrier(0);
            half8 nh = ah, nl = al;
            if (w > 0) r2 = w16_read(bh_base, bl_base, brow + off2);
            if (s8 < 7) {
                const int nrow = 2 * (s8 + 1) + lh;
                nh = *reinterpret_cast<const half8*>(ah_base + nrow * 16); nl = *reinterpret_cast<const half8*>(al_base + nrow * 16);
                r0 = w16_read(bh_base, bl_base, row_off(s8 + 1) + off0);
            }
            __builtin_amdgcn_sched_barrier(0);
            W16_MMA(acc[3], accl[3], ah, al, w16_shift<0>(r1.qh, r1.eh), w16_shift<0>(r1.ql, r1.el))
            W16_MMA(acc[4], accl[4], ah, al, w16_shift<1>(r1.qh, r1.eh), w16_shift<1>(r1.ql, r1.el))
            W16_MMA(acc[5], accl[5], ah, al, w16_shift<2>(r1.qh, r1.eh), w16_shift<2>(r1.ql, r1.el))
            if (w == 1) { W16_MMA(acc[6], accl[6], ah, al, w16_shift<0>(r2.qh, r2.eh), w16_shift<0>(r2.ql, r2.el)) }
            else if (w == 2) { W16_MMA(acc[6], accl[6], ah, al, w16_shift<1>(r2.qh, r2.eh), w16_shift<1>(r2.ql, r2.el)) }
            else if (w == 3) { W16_MMA(acc[6], accl[6], ah, al, w16_shift<2>(r2.qh, r2.eh), w16_shift<2>(r2.ql, r2.el)) }
            __builtin_amdgcn_sched_barrier(0);
            ah = nh; al = nl;
        };
#pragma unroll 1
        for (int s8 = 0; s8 < 8; ++s8) kstep(s8, -1);
        // (converting the next brick's values between the MFMAs of the last k-steps was tried: it needs unrolled copies of the step to
        //  keep the register indices static, which pushes the kernel over the 256 + 256 register file - 450 spills)
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (j == 6 && w == 0) continue;
        const int t = j < 3 ? g0 * 3 + j : (j < 6 ? g1 * 3 + (j - 3) : 24 + (w - 1));
        float* dst = p.part + (((size_t)blockIdx.x * gridDim.y + tp) * 27 + t) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r >> 2) * 8 + lh * 4 + (r & 3)) * 32 + l31] = acc[j][r] + accl[j][r] * (1.0f / W16_SPLIT);
    }
}

// ---- k2 s2 weight gradient (pool convs; transposed convs with the roles swapped) on the f16 matrix cores -------------------------
// dW[co][tap][ci] = sum_v dY[v][co] * act(X[2v + tap][ci]): a GEMM with K = all coarse voxels, M = Cout, N = 8 taps x Cin, and no halo -
// every fine voxel belongs to exactly one (coarse voxel, tap).  wgrad_kernel ran it on v_mfma_f32_32x32x2_f32 with one workgroup per
// 32 x 32 channel tile pair (each staging its own slices: the 64^3 -> 32^3 layer took 1.5 ms in 16-bit storage, 1.34 GB of tensors,
// and held up the GroupNorm passes of the main queue beside it).  Here a workgroup owns a 32-channel slice of X and ALL of dY's
// channels (MT tiles of 32), so X - the large tensor - is read exactly once; brick = 1 x 8 x 8 coarse voxels = 2 x 16 x 16 fine.
// The staging transposes into [channel][fine row][x parity][8 halves] / [channel][coarse row][8 halves] planes (16-B operand
// reads: 8 consecutive coarse x = 8 K elements of v_mfma_f32_32x32x16_f16), split hi + lo * 2^-11 with three products as in
// wgrad16_kernel (SINGLE: hi only, conv modes 3 / 4).  Wave w takes the taps (dz, dy) = (w >> 1, w & 1), both dx: per k-step MT dY
// operands + 2 X operands for 2 MT MFMAs.  The next brick's loads are issued before the MFMA phase and converted after it.
// HX / HD: X / dY stored as bfloat16 (16-byte loads of 8 channels).  Partials in wgrad_kernel's layout (same reduce).
#define W2_PA (32 * 32 + 16)
#define W2_PD (8 * 16 + 16)
__host__ __device__ constexpr size_t w2_lds_bytes(int mt, bool single) { return (size_t)(single ? 1 : 2) * (32 * W2_PA + (size_t)mt * 32 * W2_PD); }

template <int MT, bool SINGLE, bool HX, bool HD>
__global__ __launch_bounds__(256) void wgrad16k2_kernel(WgradParams p) {
    extern __shared__ char lds8[];
    char* X_hi = lds8; char* D_hi = X_hi + 32 * W2_PA;
    char* X_lo = D_hi + MT * 32 * W2_PD; char* D_lo = X_lo + 32 * W2_PA;         // (SINGLE: never touched)
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, w = tid >> 6;
    const int nt = blockIdx.y, n0 = nt * 32;
    const int OD = p.dy.D, OH = p.dy.H, OW = p.dy.W, nby = OH >> 3, nbx = OW >> 3;
    const int per_frame = OD * nby * nbx, total = p.in.N * per_frame;
    f32x16 acc[2][MT], accl[2][MT];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[d][m][r] = 0.f; accl[d][m][r] = 0.f; }
    // staging tasks.  X: one fine row (2 x 16 rows of 16 voxels) x one channel group per thread - fp32: 8 quads x 16 voxels; bfloat16:
    // 4 octets x 2 half rows of 8 voxels.  dY: one coarse row of 8 voxels x one channel group (threads beyond 8 rows idle).
    constexpr int XL = HX ? 8 : 16, XG = HX ? 8 : 4, DG = HD ? 8 : 4;
    const int frow = tid >> 3, xg = HX ? (tid & 3) : (tid & 7), xh = HX ? ((tid >> 2) & 1) : 0;
    const int xc = n0 + XG * xg;
    const bool xok = xc < p.Nc;
    constexpr int DQ = MT * 32 / DG;                      // channel groups per dY voxel
    const int drow = tid / DQ, dg = tid % DQ, dc = DG * dg;
    const bool dok = drow < 8 && dc < p.M;
    u32x4 xr[XL], dr[8];
    int n_st = 0;                                         // frame of the brick held in xr / dr
    auto fetch = [&](int b) __attribute__((always_inline)) {
        const int n = b / per_frame; int r = b % per_frame;
        const int bx = r % nbx; r /= nbx;
        const int by = r % nby, oz = r / nby;
        n_st = n;
#pragma unroll
        for (int x = 0; x < XL; ++x) xr[x] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int x = 0; x < 8; ++x) dr[x] = u32x4{0u, 0u, 0u, 0u};
        if (xok) {
            const size_t e = ((((size_t)n * p.in.D + 2 * oz + (frow >> 4)) * p.in.H + 16 * by + (frow & 15)) * p.in.W + 16 * bx + 8 * xh) * p.in.C + xc;
            const char* src = reinterpret_cast<const char*>(p.in.p) + e * (HX ? 2 : 4);
            const size_t pitch = (size_t)p.in.C * (HX ? 2 : 4);
#pragma unroll
            for (int x = 0; x < XL; ++x) xr[x] = *reinterpret_cast<const u32x4*>(src + x * pitch);
        }
        if (dok) {
            const size_t e = ((((size_t)n * OD + oz) * OH + 8 * by + drow) * OW + 8 * bx) * p.dy.C + dc;
            const char* src = reinterpret_cast<const char*>(p.dy.p) + e * (HD ? 2 : 4);
            const size_t pitch = (size_t)p.dy.C * (HD ? 2 : 4);
#pragma unroll
            for (int x = 0; x < 8; ++x) dr[x] = *reinterpret_cast<const u32x4*>(src + x * pitch);
        }
    };
    // channel j of the group out of the raw words of voxel x
    auto xval = [&](int x, int j) __attribute__((always_inline)) -> float {
        if constexpr (HX) return (j & 1) ? nm_bf_hi(xr[x][j >> 1]) : nm_bf_lo(xr[x][j >> 1]);
        else return __builtin_bit_cast(float, (unsigned)xr[x][j]);
    };
    auto dval = [&](int x, int j) __attribute__((always_inline)) -> float {
        if constexpr (HD) return (j & 1) ? nm_bf_hi(dr[x][j >> 1]) : nm_bf_lo(dr[x][j >> 1]);
        else return __builtin_bit_cast(float, (unsigned)dr[x][j]);
    };
    auto store = [&]() __attribute__((always_inline)) {
        {
            float sc[XG], sh[XG];
#pragma unroll
            for (int j = 0; j < XG; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
            if (p.in.scale && xok) {
#pragma unroll
                for (int j = 0; j < XG; j += 4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(p.in.scale + (size_t)n_st * p.in.C + xc + j);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.in.shift + (size_t)n_st * p.in.C + xc + j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sc[j + i] = a[i]; sh[j + i] = c[i]; }
                }
            }
            const float slope = p.in.slope;
#pragma unroll
            for (int j = 0; j < XG; ++j) {
                unsigned hi[2][XL / 4], lo[2][XL / 4];
#pragma unroll
                for (int i = 0; i < XL / 4; ++i)
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        float a0 = fmaf(xval(4 * i + par, j), sc[j], sh[j]), a1 = fmaf(xval(4 * i + 2 + par, j), sc[j], sh[j]);
                        if (slope != 1.0f) { a0 = fmaxf(a0, a0 * slope); a1 = fmaxf(a1, a1 * slope); }
                        if (!xok) { a0 = 0.f; a1 = 0.f; }
                        hi[par][i] = pack_split(a0, a1, lo[par][i]);
                    }
                char* dh = X_hi + (XG * xg + j) * W2_PA + frow * 32 + xh * 8;
                char* dl = X_lo + (XG * xg + j) * W2_PA + frow * 32 + xh * 8;
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    if constexpr (HX) {
                        *reinterpret_cast<nm_u32x2*>(dh + par * 16) = nm_u32x2{hi[par][0], hi[par][1]};
                        if constexpr (!SINGLE) *reinterpret_cast<nm_u32x2*>(dl + par * 16) = nm_u32x2{lo[par][0], lo[par][1]};
                    } else {
                        *reinterpret_cast<u32x4*>(dh + par * 16) = u32x4{hi[par][0], hi[par][1], hi[par][2], hi[par][3]};
                        if constexpr (!SINGLE) *reinterpret_cast<u32x4*>(dl + par * 16) = u32x4{lo[par][0], lo[par][1], lo[par][2], lo[par][3]};
                    }
                }
            }
        }
        if (drow < 8) {
            float sc[DG], sh[DG];
#pragma unroll
            for (int j = 0; j < DG; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
            if (p.dy.scale && dok) {
#pragma unroll
                for (int j = 0; j < DG; j += 4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(p.dy.scale + (size_t)n_st * p.dy.C + dc + j);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.dy.shift + (size_t)n_st * p.dy.C + dc + j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sc[j + i] = a[i]; sh[j + i] = c[i]; }
                }
            }
            const float slope = p.dy.slope;
#pragma unroll
            for (int j = 0; j < DG; ++j) {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a0 = fmaf(dval(2 * i, j), sc[j], sh[j]), a1 = fmaf(dval(2 * i + 1, j), sc[j], sh[j]);
                    if (slope != 1.0f) { a0 = fmaxf(a0, a0 * slope); a1 = fmaxf(a1, a1 * slope); }
                    if (!dok) { a0 = 0.f; a1 = 0.f; }
                    hi[i] = pack_split(a0, a1, lo[i]);
                }
                *reinterpret_cast<u32x4*>(D_hi + (dc + j) * W2_PD + drow * 16) = u32x4{hi[0], hi[1], hi[2], hi[3]};
                if constexpr (!SINGLE) *reinterpret_cast<u32x4*>(D_lo + (dc + j) * W2_PD + drow * 16) = u32x4{lo[0], lo[1], lo[2], lo[3]};
            }
        }
    };
    int b = blockIdx.x;
    if (b < total) fetch(b);
    const int tap_row = (w >> 1) * 16 + (w & 1);           // fine row of coarse row 0 for this wave's (dz, dy)
    const char* bh_base = X_hi + l31 * W2_PA + tap_row * 32; const char* bl_base = X_lo + l31 * W2_PA + tap_row * 32;
    const char* ah_base = D_hi + l31 * W2_PD; const char* al_base = D_lo + l31 * W2_PD;
    for (; b < total; b += p.S) {
        __syncthreads();
        store();
        __syncthreads();
        if (b + p.S < total) fetch(b + p.S);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int row = 2 * s4 + lh;
            const half8 bh0 = *reinterpret_cast<const half8*>(bh_base + row * 64), bh1 = *reinterpret_cast<const half8*>(bh_base + row * 64 + 16);
            half8 bl0 = bh0, bl1 = bh1;
            if constexpr (!SINGLE) { bl0 = *reinterpret_cast<const half8*>(bl_base + row * 64); bl1 = *reinterpret_cast<const half8*>(bl_base + row * 64 + 16); }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const half8 ah = *reinterpret_cast<const half8*>(ah_base + m * 32 * W2_PD + row * 16);
                half8 al = ah;
                if constexpr (!SINGLE) al = *reinterpret_cast<const half8*>(al_base + m * 32 * W2_PD + row * 16);
                W16_MMA(acc[0][m], accl[0][m], ah, al, bh0, bl0)
                W16_MMA(acc[1][m], accl[1][m], ah, al, bh1, bl1)
            }
        }
    }
    // partial buffer: [slot][tile = m * n_tiles + nt][tap][32 x 32]
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m * 32 >= p.M) continue;
            float* dst = p.part + (((size_t)blockIdx.x * ((p.M + 31) / 32) * gridDim.y + (size_t)m * gridDim.y + nt) * 8 + 2 * w + d) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((r >> 2) * 8 + lh * 4 + (r & 3)) * 32 + l31] = acc[d][m][r] + accl[d][m][r] * (1.0f / W16_SPLIT);
        }
}

// ---- wgrad16t: the same weight gradient through transposing LDS reads (gfx950 ds_read_b64_tr_b16) -------------------------------
// wgrad16_kernel spends ~35-45 % of its time turning the channels-last tiles into [channel][voxel] planes with VALU before the first
// MFMA, because v_mfma_f32_32x32x16_f16 wants 8 consecutive K (= voxels) of one channel per lane.  ds_read_b64_tr_b16 delivers a
// 4 (rows = voxels) x 16 (columns = channels) block of 16-bit elements column-major: lane i of a 16-lane group receives column i of
// the 4 rows.  So LDS keeps the natural channels-last fp16 tiles - [voxel][32 channels], 64 B per voxel, filled by plain 16-byte
// stores - and an operand is two transposed reads (x = 0..3, 4..7 of a brick row).  A tap is just a row offset into the halo tile
// (no register shifting for dx), the four reads of a 32-lane half cover 256 contiguous bytes (conflict-free), the staging is the
// forward convs' activate + split, and a buffer is 66 KB: two of them, so the next brick is loaded (registers) and written (other
// buffer) while the MFMAs of the current one run.  512 threads: the 27 taps are dealt to 8 waves (4, 4, 4, 3, 3, 3, 3, 3 - two
// waves per SIMD), each holding its taps' accumulator pairs; per k-step of 16 voxels a wave reads the dY operand (4 transposed
// reads) and its taps' X operands (4 each) and issues 3 MFMAs per tap.  Partials and the reduce are wgrad16_kernel's.
#define WT_HV 400
#define WT_BV 128
#define WT_XB (WT_HV * 64)
#define WT_DB (WT_BV * 64)
#define WT_BUF (2 * WT_XB + 2 * WT_DB)
#define WT_LDS (2 * WT_BUF)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 tr_read(const char* lds_ptr) {
    // (take the 64-bit result as a whole: element-wise extraction of the builtin's vector is miscompiled by this hipcc)
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)lds_ptr));
}
__device__ __forceinline__ half8 tr_operand(const char* p) {
    const u32x2 a = tr_read(p), b = tr_read(p + 4 * 64);
    return __builtin_bit_cast(half8, u32x4{a[0], a[1], b[0], b[1]});
}

template <int DBG, bool SINGLE = false>      // DBG 0: product; 1: no MFMA loop, 2: no staging after the first brick (timing experiments, NM355_W16_DBG)
__global__ __launch_bounds__(512, 1) void wgrad16t_kernel(WgradParams p) {
    extern __shared__ char lds8[];
    // per-frame GroupNorm scale / shift of this column tile's 32 input channels: [N][2][32] floats behind the two tile buffers, loaded
    // once (a commit inside the k-loop must not wait for global memory)
    float* xtab = reinterpret_cast<float*>(lds8 + WT_LDS);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tp = blockIdx.y, mt = tp / p.n_tiles, nt = tp % p.n_tiles, m0 = mt * 32, n0 = nt * 32;
    // taps: wave w owns the tap line (dz, dy) = (w / 3, w % 3) - taps 3 w + dx, whose three X operands come out of ONE halo row read
    // (10 voxels, shifted in registers) - and waves 0..2 also tap 24 + w of the ninth line
    const int ntaps = w < 3 ? 4 : 3;
    f32x16 acc[4], accl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accl[j][r] = 0.f; }
    // transposed-read addresses: lane 4q + pp of a 16-lane group supplies row q, columns 4pp .. 4pp + 3 of the group's block;
    // group = (column block cb = (lane >> 4) & 1, lane half lh)
    const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, cb = (lane >> 4) & 1;
    const int a_lane = ((lh * 8 + q) * 32 + 16 * cb + 4 * pp) * 2;          // dY tile: brick row 2s + lh, voxel x0 + q
    const int b_lane = ((lh * 10 + q) * 32 + 16 * cb + 4 * pp) * 2;         // X tile: halo row (z + tz, 2(s & 3) + lh + ty), voxel x0 + q + tx
    const int lineoff = (((w / 3) * 10 + w % 3) * 10) * 64;                 // halo row (dz, dy) of the wave's line, x = -1
    const int extraoff = ((2 * 10 + 2) * 10 + min(w, 2)) * 64;              // tap 24 + w = (2, 2, w)

    const int per_frame = p.nbz * p.nby * p.nbx, total = p.in.N * per_frame;
    // staging roles: three (four for 64 threads) X items (halo voxel, channel octet) and one dY item per thread; octet = tid & 3
    const int oct = tid & 3;
    for (int i = tid; i < p.in.N * 64; i += 512) {        // (identity when the input carries no pending GroupNorm: the commit is branch-free)
        const int n = i >> 6, which = (i >> 5) & 1, c = n0 + (i & 31);
        xtab[i] = p.in.scale ? (c < p.Nc ? (which ? p.in.shift : p.in.scale)[(size_t)n * p.in.C + c] : 0.f) : (which ? 0.f : 1.f);
    }
    // (the table is read by other threads than those that wrote it, first in the prologue below: without this barrier a wave that ran
    //  ahead converted its first items with whatever the previous workgroup left in LDS - a run-to-run difference of ~1e-4 in the
    //  weight gradient of one layer, one run in ~80 when the kernel ran alone and one in ~5 beside the main stream's kernels; round 4)
    __syncthreads();
    char* dummy = lds8 + WT_LDS + (size_t)p.in.N * 256;   // 32 B written by the threads without a fourth X item
    f32x4 xa[4], xb[4], da, db, dsa, dsb, dha, dhb;
    dsa = dsb = f32x4{1.f, 1.f, 1.f, 1.f}; dha = dhb = f32x4{0.f, 0.f, 0.f, 0.f};
    int xn = 0, xoz = 0, xoy = 0, xox = 0;                // brick being fetched
    // the next brick arrives in two instalments (registers): dY item + X items 0, 1 at the start of a brick, X items 2, 3 mid-brick
    auto locate = [&](int b) __attribute__((always_inline)) {
        xn = b / per_frame; int r = b % per_frame;
        const int bx = r % p.nbx; r /= p.nbx;
        xoz = (r / p.nby) * 2; xoy = (r % p.nby) * 8; xox = bx * 8;
    };
    // per X item: the thread's halo voxel as packed (hx, hy, hz) and as an element offset from the brick's first halo voxel; the
    // brick part of every address is wave-uniform (scalar registers)
    int hpack[4], hoffs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int hv = min((tid >> 2) + 128 * k, WT_HV - 1), hx = hv % 10, hy = (hv / 10) % 10, hz = hv / 100;
        hpack[k] = hx | (hy << 8) | (hz << 16);
        hoffs[k] = ((hz * p.in.H + hy) * p.in.W + hx) * p.in.C + n0 + 8 * oct;
    }
    unsigned inmask = 0;                                  // bit k: X item k of the brick being fetched lies inside the tensor
    auto fetch = [&](int k0, int k1, bool with_dy) __attribute__((always_inline)) {
        const int c = n0 + 8 * oct;
        const long long base = ((((long long)xn * p.in.D + (xoz - 1)) * p.in.H + (xoy - 1)) * p.in.W + (xox - 1)) * (long long)p.in.C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < k0 || k >= k1) continue;
            xa[k] = xb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int gx = xox - 1 + (hpack[k] & 255), gy = xoy - 1 + ((hpack[k] >> 8) & 255), gz = xoz - 1 + (hpack[k] >> 16);
            const bool in = (unsigned)gz < (unsigned)p.in.D && (unsigned)gy < (unsigned)p.in.H && (unsigned)gx < (unsigned)p.in.W &&
                            (tid >> 2) + 128 * k < WT_HV;
            inmask = (inmask & ~(1u << k)) | ((in ? 1u : 0u) << k);
            if (in) {
                const float* src = p.in.p + base + hoffs[k];
                if (c < p.Nc) xa[k] = *reinterpret_cast<const f32x4*>(src);
                if (c + 4 < p.Nc) xb[k] = *reinterpret_cast<const f32x4*>(src + 4);
            }
        }
        if (with_dy) {
            const int bv = tid >> 2, z = bv >> 6, y = (bv >> 3) & 7, x = bv & 7, m = m0 + 8 * oct;
            const float* src = p.dy.p + ((((size_t)xn * p.dy.D + xoz + z) * p.dy.H + xoy + y) * p.dy.W + xox + x) * p.dy.C + m;
            da = db = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < p.M) da = *reinterpret_cast<const f32x4*>(src);
            if (m + 4 < p.M) db = *reinterpret_cast<const f32x4*>(src + 4);
            if (p.dy.scale) {                             // (else they stay 1 / 0)
                const float* ps = p.dy.scale + (size_t)xn * p.dy.C + m; const float* ph = p.dy.shift + (size_t)xn * p.dy.C + m;
                if (m < p.M) { dsa = *reinterpret_cast<const f32x4*>(ps); dha = *reinterpret_cast<const f32x4*>(ph); }
                if (m + 4 < p.M) { dsb = *reinterpret_cast<const f32x4*>(ps + 4); dhb = *reinterpret_cast<const f32x4*>(ph + 4); }
            }
        }
    };
    // (no branches from here to the store: the conversion of an item has to sit in the same basic block as the k-step's MFMAs for
    //  the scheduler to interleave the two; slope 1 makes the leaky ReLU an identity, the tables hold 1 / 0 without a GroupNorm)
    auto act_x = [&](int n, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        const float* t = xtab + n * 64 + 8 * oct;
        const f32x4 sa = *reinterpret_cast<const f32x4*>(t), sb = *reinterpret_cast<const f32x4*>(t + 4);
        const f32x4 ha = *reinterpret_cast<const f32x4*>(t + 32), hb = *reinterpret_cast<const f32x4*>(t + 36);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = fmaf(a[j], sa[j], ha[j]); b[j] = fmaf(b[j], sb[j], hb[j]);
            a[j] = fmaxf(a[j], a[j] * p.in.slope); b[j] = fmaxf(b[j], b[j] * p.in.slope);
        }
    };
    auto act_d = [&](f32x4& a, f32x4& b) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = fmaf(a[j], dsa[j], dha[j]); b[j] = fmaf(b[j], dsb[j], dhb[j]);
            a[j] = fmaxf(a[j], a[j] * p.dy.slope); b[j] = fmaxf(b[j], b[j] * p.dy.slope);
        }
    };
    auto split_store = [&](const f32x4& a, const f32x4& b, char* hi, char* lo) __attribute__((always_inline)) {
        unsigned h[4], l[4];
        h[0] = pack_split(a[0], a[1], l[0]); h[1] = pack_split(a[2], a[3], l[1]);
        h[2] = pack_split(b[0], b[1], l[2]); h[3] = pack_split(b[2], b[3], l[3]);
        *reinterpret_cast<u32x4*>(hi) = u32x4{h[0], h[1], h[2], h[3]};
        if constexpr (!SINGLE) *reinterpret_cast<u32x4*>(lo) = u32x4{l[0], l[1], l[2], l[3]};      // (conv mode 3 never reads the lo tiles)
    };
    // slot 0..3: X items, slot 4: the dY item.  Unconditional: without a next brick the stale registers go to the idle buffer.
    const float xa_on = n0 + 8 * oct < p.Nc ? 1.f : 0.f, xb_on = n0 + 8 * oct + 4 < p.Nc ? 1.f : 0.f;
    const float da_on = m0 + 8 * oct < p.M ? 1.f : 0.f, db_on = m0 + 8 * oct + 4 < p.M ? 1.f : 0.f;
    auto commit = [&](char* buf, int slot) __attribute__((always_inline)) {
        if (slot < 4) {
            const int hv = (tid >> 2) + 128 * slot;
            f32x4 a = xa[slot], b = xb[slot];
            act_x(xn, a, b);
            // zero padding, not act(0); as a multiplication (the item's registers are zero, act(0) is finite): a select would be
            // turned into a branch around the conversion
            const float in = (float)((inmask >> slot) & 1), ka = in * xa_on, kb = in * xb_on;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] *= ka; b[j] *= kb; }
            char* hi = buf + hv * 64 + oct * 16; char* lo = hi + WT_XB;
            if (slot == 3) { const bool item = hv < WT_HV; hi = item ? hi : dummy; lo = item ? lo : dummy + 16; }
            split_store(a, b, hi, lo);
        } else {
            const int bv = tid >> 2;
            f32x4 a = da, b = db;
            act_d(a, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] *= da_on; b[j] *= db_on; }
            split_store(a, b, buf + 2 * WT_XB + bv * 64 + oct * 16, buf + 2 * WT_XB + WT_DB + bv * 64 + oct * 16);
        }
    };

    int b = blockIdx.x, cur = 0;
    if (b < total) {
        locate(b); fetch(0, 4, true);
#pragma unroll
        for (int sl = 0; sl < 5; ++sl) commit(lds8, sl);
    }
    __syncthreads();
    for (; b < total; b += p.S) {
        const bool has_next = b + p.S < total && DBG != 2;
        const char* buf = lds8 + cur * WT_BUF;
        char* nbuf = lds8 + (cur ^ 1) * WT_BUF;
        // the next brick comes through registers one item at a time (dY item, then the four X items), each written to the other
        // buffer two or three k-steps (5000+ cycles) after its loads were issued: at most two items are live at once - with all
        // five in flight the kernel spills, and a scratch reload queues behind the outstanding global loads (vmcnt is in order)
        if (has_next) { locate(b + p.S); fetch(0, 1, true); }
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            if (has_next) {
                if (s8 == 2) fetch(1, 2, false);
                if (s8 == 3) fetch(2, 3, false);
                if (s8 == 4) fetch(3, 4, false);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the item conversion (branch-free) ahead of the k-step's operand reads
            constexpr int slot_of[8] = {-1, -1, 4, 0, 1, -1, 2, 3};
            if (DBG != 2 && slot_of[s8] >= 0) commit(nbuf, slot_of[s8]);
            if (DBG == 1) continue;
            const char* ap = buf + 2 * WT_XB + a_lane + s8 * 1024;
            const half8 ah = tr_operand(ap), al = tr_operand(ap + WT_DB);
            const char* xp = buf + b_lane + (((s8 >> 2) * 10 + 2 * (s8 & 3)) * 10) * 64;
            {
                // halo row of the line: voxels x = -1 .. 10 as three transposed reads (x = 9, 10 of the third are never used), hi and lo;
                // dx = 0 is registers 0..3, dx = 2 registers 1..4, dx = 1 four v_alignbit - 6 LDS reads for three taps instead of 12
                const char* lp = xp + lineoff;
                const u32x2 h0 = tr_read(lp), h1 = tr_read(lp + 4 * 64), h2 = tr_read(lp + 8 * 64);
                const u32x2 l0 = tr_read(lp + WT_XB), l1 = tr_read(lp + WT_XB + 4 * 64), l2 = tr_read(lp + WT_XB + 8 * 64);
                half8 bh[3], bl[3];
                bh[0] = __builtin_bit_cast(half8, u32x4{h0[0], h0[1], h1[0], h1[1]});
                bh[2] = __builtin_bit_cast(half8, u32x4{h0[1], h1[0], h1[1], h2[0]});
                bh[1] = __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(h0[1], h0[0], 16), __builtin_amdgcn_alignbit(h1[0], h0[1], 16),
                                                        __builtin_amdgcn_alignbit(h1[1], h1[0], 16), __builtin_amdgcn_alignbit(h2[0], h1[1], 16)});
                bl[0] = __builtin_bit_cast(half8, u32x4{l0[0], l0[1], l1[0], l1[1]});
                bl[2] = __builtin_bit_cast(half8, u32x4{l0[1], l1[0], l1[1], l2[0]});
                bl[1] = __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(l0[1], l0[0], 16), __builtin_amdgcn_alignbit(l1[0], l0[1], 16),
                                                        __builtin_amdgcn_alignbit(l1[1], l1[0], 16), __builtin_amdgcn_alignbit(l2[0], l1[1], 16)});
#pragma unroll
                for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 3; ++u) accl[u] = nm_mfma_lo<SINGLE>(ah, bl[u], accl[u]);
#pragma unroll
                for (int u = 0; u < 3; ++u) accl[u] = nm_mfma_lo<SINGLE>(al, bh[u], accl[u]);
            }
            if (ntaps == 4) {                             // waves 0..2
                const half8 bh = tr_operand(xp + extraoff), bl = tr_operand(xp + WT_XB + extraoff);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[3], 0, 0, 0);
                accl[3] = nm_mfma_lo<SINGLE>(ah, bl, accl[3]);
                accl[3] = nm_mfma_lo<SINGLE>(al, bh, accl[3]);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= ntaps) continue;
        const int t = j < 3 ? 3 * w + j : 24 + w;
        float* dst = p.part + (((size_t)blockIdx.x * gridDim.y + tp) * 27 + t) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r >> 2) * 8 + lh * 4 + (r & 3)) * 32 + l31] = acc[j][r] + accl[j][r] * (1.0f / W16_SPLIT);
    }
}

// ---- wgrad16u: wgrad16t_kernel with a statically countable memory pipeline (round 3) -------------------------------------------
// What held wgrad16t_kernel at 2.5-3x its MFMA time (6.8-8.2 us per 128-voxel brick against 2.7 us of matrix work) shows in its
// ISA: (1) every staging load sat under a condition (halo voxel inside the volume, channel tail, "is there a next brick"), so the
// compiler could not count the loads in flight and drained them - s_waitcnt vmcnt(0) - in front of every item's conversion,
// i.e. right behind the loads it had just issued for the NEXT item: five exposed memory round trips per brick; (2) a k-step's
// transposing LDS reads were issued in front of the MFMAs that consume them (their latency exposed once per k-step and wave),
// and the fourth tap of waves 0..2 sat in a branch of its own (four reads, a full wait, three MFMAs).  Here: every load is
// unconditional - the address of an item that does not exist is replaced by the tensor's first element and its value multiplied
// by zero - and "no next brick" fetches the current one again; the dY tensor's affine comes from an identity table when it has
// none; the body is instantiated for 4 and 3 taps and the wave picks its copy once, outside the brick loop; a k-step requests the
// NEXT k-step's operands before its own MFMAs.  Nothing inside the brick loop branches, so every wait is a counted one.
__device__ float w16u_ident[16] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

struct W16uHi { u32x2 a[2]; u32x2 x[3]; };       // hi halves of a k-step's operands: dY (2 transposing reads), halo row (3) - requested one k-step ahead
struct W16uLo { u32x2 a[2]; u32x2 x[3]; };       // lo halves, requested at the top of their own k-step (first used behind the hi x hi MFMAs)
struct W16uExt { u32x2 h[2], l[2]; };            // the fourth tap's X operand

#define W16U_PIN4(v) asm volatile("" : "+v"(v))
template <typename F, int... I>
__device__ __forceinline__ void w16u_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void w16u_static_for(F&& f) { w16u_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
template <int V> using ic_ = std::integral_constant<int, V>;
template <int NT, int DBG, bool SINGLE>
__device__ __forceinline__ void wgrad16u_run(const WgradParams& p, const int w) {
    extern __shared__ char lds8[];
    float* xtab = reinterpret_cast<float*>(lds8 + WT_LDS);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int tp = blockIdx.y, mt = tp / p.n_tiles, nt = tp % p.n_tiles, m0 = mt * 32, n0 = nt * 32;
    f32x16 acc[NT], accl[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accl[j][r] = 0.f; }
    const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, cb = (lane >> 4) & 1;
    // per-lane LDS byte offsets of the three operand streams (buffer base and the k-step's row are added per use: immediates)
    const int a_off = 2 * WT_XB + ((lh * 8 + q) * 32 + 16 * cb + 4 * pp) * 2;
    const int b_lane = ((lh * 10 + q) * 32 + 16 * cb + 4 * pp) * 2;
    const int x_off = b_lane + (((w / 3) * 10 + w % 3) * 10) * 64;
    const int e_off = b_lane + ((2 * 10 + 2) * 10 + min(w, 2)) * 64;
    const int per_frame = p.nbz * p.nby * p.nbx, total = p.in.N * per_frame;
    const int oct = tid & 3;
    for (int i = tid; i < p.in.N * 64; i += 512) {
        const int n = i >> 6, which = (i >> 5) & 1, c = n0 + (i & 31);
        xtab[i] = p.in.scale ? (c < p.Nc ? (which ? p.in.shift : p.in.scale)[(size_t)n * p.in.C + c] : 0.f) : (which ? 0.f : 1.f);
    }
    // (the table is read by other threads than those that wrote it, first in the prologue below: without this barrier a wave that ran
    //  ahead converted its first items with whatever the previous workgroup left in LDS - a run-to-run difference of ~1e-4 in the
    //  weight gradient of one layer, one run in ~80 when the kernel ran alone and one in ~5 beside the main stream's kernels; round 4)
    __syncthreads();
    const int dummy_off = WT_LDS + p.in.N * 256;      // 32 B written by the threads without a fourth X item
    int xn = 0, xoz = 0, xoy = 0, xox = 0;
    auto locate = [&](int b) __attribute__((always_inline)) {
        xn = b / per_frame; int r = b % per_frame;
        const int bx = r % p.nbx; r /= p.nbx;
        xoz = (r / p.nby) * 2; xoy = (r % p.nby) * 8; xox = bx * 8;
    };
    const bool ca_ok = n0 + 8 * oct < p.Nc, cb_ok = n0 + 8 * oct + 4 < p.Nc;
    const bool ma_ok = m0 + 8 * oct < p.M, mb_ok = m0 + 8 * oct + 4 < p.M;
    // X item k: halo voxel (tid >> 2) + 128 k as packed (hx, hy, hz, exists), channel octet oct
    int hpack[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int hv0 = (tid >> 2) + 128 * k, hv = min(hv0, WT_HV - 1), hx = hv % 10, hy = (hv / 10) % 10, hz = hv / 100;
        hpack[k] = hx | (hy << 8) | (hz << 16) | ((hv0 < WT_HV ? 1 : 0) << 24);
    }
    // two unconditional 16-byte loads; an item outside the volume (or the tile's channels) reads the tensor's first element and is
    // multiplied by zero at its conversion.  (The unpacking goes through an opaque copy: as a loop invariant the compiler keeps the
    // twelve coordinates in registers of their own, spills them, and a scratch reload queues behind the loads in flight.)
    auto x_inside = [&](int hp, int& gx, int& gy, int& gz) __attribute__((always_inline)) {
        gx = xox - 1 + (hp & 255); gy = xoy - 1 + ((hp >> 8) & 255); gz = xoz - 1 + ((hp >> 16) & 255);
        return (unsigned)gz < (unsigned)p.in.D && (unsigned)gy < (unsigned)p.in.H && (unsigned)gx < (unsigned)p.in.W && (hp >> 24) != 0;
    };
    auto x_issue = [&](int k, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        int hp = hpack[k], gx, gy, gz;
        asm volatile("" : "+v"(hp));
        const bool in = x_inside(hp, gx, gy, gz);
        const long long off = ((((long long)xn * p.in.D + gz) * p.in.H + gy) * p.in.W + gx) * (long long)p.in.C + n0 + 8 * oct;
        const float* src = p.in.p + off;
        const float* sa = in && ca_ok ? src : p.in.p;
        const float* sb = in && cb_ok ? src + 4 : p.in.p;
        a = *reinterpret_cast<const f32x4*>(sa); b = *reinterpret_cast<const f32x4*>(sb);
    };
    // the dY item: brick voxel tid >> 2, channel octet; values now, the tensor's pending affine (identity table without one) one
    // k-step ahead of the conversion
    auto d_issue = [&](f32x4& a, f32x4& b) __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int bv = t >> 2, z = bv >> 6, y = (bv >> 3) & 7, x = bv & 7, m = m0 + 8 * (t & 3);
        const float* src = p.dy.p + ((((size_t)xn * p.dy.D + xoz + z) * p.dy.H + xoy + y) * p.dy.W + xox + x) * p.dy.C + m;
        a = *reinterpret_cast<const f32x4*>(ma_ok ? src : p.dy.p); b = *reinterpret_cast<const f32x4*>(mb_ok ? src + 4 : p.dy.p);
    };
    auto d_affine = [&](f32x4& sa, f32x4& sb, f32x4& ha, f32x4& hb) __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const bool has = p.dy.scale != nullptr;
        const size_t so = (size_t)xn * p.dy.C + m0 + 8 * (t & 3);
        const float* ps = has ? p.dy.scale + so : w16u_ident; const float* ph = has ? p.dy.shift + so : w16u_ident + 8;
        sa = *reinterpret_cast<const f32x4*>(ma_ok ? ps : w16u_ident); ha = *reinterpret_cast<const f32x4*>(ma_ok ? ph : w16u_ident + 8);
        sb = *reinterpret_cast<const f32x4*>(mb_ok ? ps + 4 : w16u_ident); hb = *reinterpret_cast<const f32x4*>(mb_ok ? ph + 4 : w16u_ident + 8);
    };
    auto split_store = [&](const f32x4& a, const f32x4& b, char* hi, char* lo) __attribute__((always_inline)) {
        unsigned h[4], l[4];
        h[0] = pack_split(a[0], a[1], l[0]); h[1] = pack_split(a[2], a[3], l[1]);
        h[2] = pack_split(b[0], b[1], l[2]); h[3] = pack_split(b[2], b[3], l[3]);
        *reinterpret_cast<u32x4*>(hi) = u32x4{h[0], h[1], h[2], h[3]};
        if constexpr (!SINGLE) *reinterpret_cast<u32x4*>(lo) = u32x4{l[0], l[1], l[2], l[3]};
    };
    // conversions: the item's registers pass through an opaque statement first - otherwise the compiler starts the arithmetic right
    // behind the loads (it frees registers) and waits for them there
    auto x_commit = [&](char* cbase, int k, int n, f32x4 a, f32x4 b) __attribute__((always_inline)) {
        W16U_PIN4(a); W16U_PIN4(b);
        int hp = hpack[k], gx, gy, gz;
        asm volatile("" : "+v"(hp));
        const bool in = x_inside(hp, gx, gy, gz);        // (recomputed: cheaper than a register carried from the request)
        const float* t = xtab + n * 64 + 8 * oct;
        const f32x4 sa = *reinterpret_cast<const f32x4*>(t), sb = *reinterpret_cast<const f32x4*>(t + 4);
        const f32x4 ha = *reinterpret_cast<const f32x4*>(t + 32), hb = *reinterpret_cast<const f32x4*>(t + 36);
        const float ka = in && ca_ok ? 1.f : 0.f, kb = in && cb_ok ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = fmaf(a[j], sa[j], ha[j]); b[j] = fmaf(b[j], sb[j], hb[j]);
            a[j] = fmaxf(a[j], a[j] * p.in.slope) * ka; b[j] = fmaxf(b[j], b[j] * p.in.slope) * kb;
        }
        char* hi = cbase + 8192 * k; char* lo = hi + WT_XB;                 // (voxel (tid >> 2) + 128 k, octet) = byte 16 tid + 8192 k
        if (k == 3) { const bool item = tid < 64; hi = item ? hi : lds8 + dummy_off; lo = item ? lo : lds8 + dummy_off + 16; }
        split_store(a, b, hi, lo);
    };
    auto d_commit = [&](char* cbase, f32x4 a, f32x4 b, f32x4 sa, f32x4 sb, f32x4 ha, f32x4 hb) __attribute__((always_inline)) {
        W16U_PIN4(a); W16U_PIN4(b); W16U_PIN4(sa); W16U_PIN4(sb); W16U_PIN4(ha); W16U_PIN4(hb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = fmaf(a[j], sa[j], ha[j]); b[j] = fmaf(b[j], sb[j], hb[j]);
            a[j] = fmaxf(a[j], a[j] * p.dy.slope) * (ma_ok ? 1.f : 0.f); b[j] = fmaxf(b[j], b[j] * p.dy.slope) * (mb_ok ? 1.f : 0.f);
        }
        split_store(a, b, cbase + 2 * WT_XB, cbase + 2 * WT_XB + WT_DB);
    };
    auto read_hi = [&](const char* buf, int s8, W16uHi& o) __attribute__((always_inline)) {
        const char* ap = buf + a_off + s8 * 1024;
        const char* lp = buf + x_off + (((s8 >> 2) * 10 + 2 * (s8 & 3)) * 10) * 64;
        o.a[0] = tr_read(ap); o.a[1] = tr_read(ap + 4 * 64);
        o.x[0] = tr_read(lp); o.x[1] = tr_read(lp + 4 * 64); o.x[2] = tr_read(lp + 8 * 64);
    };
    auto read_lo = [&](const char* buf, int s8, W16uLo& o) __attribute__((always_inline)) {
        const char* ap = buf + a_off + s8 * 1024 + WT_DB;
        const char* lp = buf + x_off + (((s8 >> 2) * 10 + 2 * (s8 & 3)) * 10) * 64 + WT_XB;
        o.x[0] = tr_read(lp); o.x[1] = tr_read(lp + 4 * 64); o.x[2] = tr_read(lp + 8 * 64);
        o.a[0] = tr_read(ap); o.a[1] = tr_read(ap + 4 * 64);
    };
    auto read_ext = [&](const char* buf, int s8, W16uExt& e) __attribute__((always_inline)) {
        const char* xp = buf + e_off + (((s8 >> 2) * 10 + 2 * (s8 & 3)) * 10) * 64;
        e.h[0] = tr_read(xp); e.h[1] = tr_read(xp + 4 * 64);
        if constexpr (!SINGLE) { e.l[0] = tr_read(xp + WT_XB); e.l[1] = tr_read(xp + WT_XB + 4 * 64); }
    };
    auto shifted = [&](const u32x2& r0, const u32x2& r1, const u32x2& r2, half8 (&o)[3]) __attribute__((always_inline)) {
        o[0] = __builtin_bit_cast(half8, u32x4{r0[0], r0[1], r1[0], r1[1]});
        o[2] = __builtin_bit_cast(half8, u32x4{r0[1], r1[0], r1[1], r2[0]});
        o[1] = __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(r0[1], r0[0], 16), __builtin_amdgcn_alignbit(r1[0], r0[1], 16),
                                               __builtin_amdgcn_alignbit(r1[1], r1[0], 16), __builtin_amdgcn_alignbit(r2[0], r1[1], 16)});
    };
    int b = blockIdx.x, cur = 0;
    if (b < total) {                                  // the first brick: plain fetch + convert (once per workgroup)
        locate(b);
        f32x4 a, c, sa, sb, ha, hb;
        char* cbase = lds8 + tid * 16;
        d_issue(a, c); d_affine(sa, sb, ha, hb); d_commit(cbase, a, c, sa, sb, ha, hb);
#pragma unroll
        for (int k = 0; k < 4; ++k) { x_issue(k, a, c); x_commit(cbase, k, xn, a, c); }
    }
    __syncthreads();
    for (; b < total; b += p.S) {
        const bool has_next = b + p.S < total;
        const char* buf = lds8 + cur * WT_BUF;
        char* cbase = lds8 + (cur ^ 1) * WT_BUF + tid * 16;
        locate(has_next ? b + p.S : b);               // (no next brick: this one again, into the idle buffer)
        // staging of the next brick, two items in flight: requested at the end of a k-step, converted at the end of the second (dY,
        // X0, X1) or third (X2, X3) k-step after it
        f32x4 da, db, dsa, dsb, dha, dhb, xa[4], xb[4];
        constexpr bool STAGE = DBG != 2 && DBG != 3 && DBG != 4;
        if (STAGE) d_issue(da, db);
        W16uHi h0, h1;
        W16uLo lo;
        W16uExt e0;
        if (DBG != 1) read_hi(buf, 0, h0);
        // An item's conversion as six pieces (+ the request of a later item, + the dY affine's loads), placed by hand between the
        // k-step's MFMAs: left to itself the scheduler issues the MFMAs back to back and the ~130 vector instructions as one block
        // behind them, through which the matrix pipe idles (sched_group_barrier spreads some k-steps and not others).
        f32x4 ca, cb, csa, csb, cha, chb;
        float cka = 0.f, ckb = 0.f, cslope = 1.f;
        unsigned ch[4], cl[4];
        w16u_static_for<8>([&](auto S8) __attribute__((always_inline)) {
            constexpr int s8 = decltype(S8)::value;
            constexpr int CI = s8 >= 2 && s8 <= 6 ? s8 - 2 : -1;          // item converted in this k-step: 0 = dY, 1..4 = X item CI - 1
            constexpr int RI = s8 <= 3 ? s8 : -1;                          // X item requested in this k-step
            auto piece = [&](auto P) __attribute__((always_inline)) {
                constexpr int pc = decltype(P)::value;
                if constexpr (!STAGE) return;
                if constexpr (CI >= 0 && pc == 0) {
                    if constexpr (CI == 0) {
                        ca = da; cb = db; csa = dsa; csb = dsb; cha = dha; chb = dhb;
                        W16U_PIN4(ca); W16U_PIN4(cb); W16U_PIN4(csa); W16U_PIN4(csb); W16U_PIN4(cha); W16U_PIN4(chb);
                        cka = ma_ok ? 1.f : 0.f; ckb = mb_ok ? 1.f : 0.f; cslope = p.dy.slope;
                    } else {
                        ca = xa[CI - 1]; cb = xb[CI - 1];
                        W16U_PIN4(ca); W16U_PIN4(cb);
                        int hp = hpack[CI - 1], gx, gy, gz;
                        asm volatile("" : "+v"(hp));
                        const bool in = x_inside(hp, gx, gy, gz);
                        cka = in && ca_ok ? 1.f : 0.f; ckb = in && cb_ok ? 1.f : 0.f; cslope = p.in.slope;
                        const float* t = xtab + xn * 64 + 8 * oct;
                        csa = *reinterpret_cast<const f32x4*>(t); csb = *reinterpret_cast<const f32x4*>(t + 4);
                        cha = *reinterpret_cast<const f32x4*>(t + 32); chb = *reinterpret_cast<const f32x4*>(t + 36);
                    }
                }
                if constexpr (CI >= 0 && pc >= 1 && pc <= 4) {
                    constexpr int j = (pc - 1) & 1;
                    float v0, v1;
                    if constexpr (pc <= 2) {
                        v0 = fmaf(ca[2 * j], csa[2 * j], cha[2 * j]); v1 = fmaf(ca[2 * j + 1], csa[2 * j + 1], cha[2 * j + 1]);
                        v0 = fmaxf(v0, v0 * cslope) * cka; v1 = fmaxf(v1, v1 * cslope) * cka;
                    } else {
                        v0 = fmaf(cb[2 * j], csb[2 * j], chb[2 * j]); v1 = fmaf(cb[2 * j + 1], csb[2 * j + 1], chb[2 * j + 1]);
                        v0 = fmaxf(v0, v0 * cslope) * ckb; v1 = fmaxf(v1, v1 * cslope) * ckb;
                    }
                    ch[pc - 1] = pack_split(v0, v1, cl[pc - 1]);
                }
                if constexpr (CI >= 0 && pc == 5) {
                    char* hi; char* lo_;
                    if constexpr (CI == 0) { hi = cbase + 2 * WT_XB; lo_ = hi + WT_DB; }
                    else {
                        hi = cbase + 8192 * (CI - 1); lo_ = hi + WT_XB;
                        if constexpr (CI == 4) { const bool item = tid < 64; hi = item ? hi : lds8 + dummy_off; lo_ = item ? lo_ : lds8 + dummy_off + 16; }
                    }
                    *reinterpret_cast<u32x4*>(hi) = u32x4{ch[0], ch[1], ch[2], ch[3]};
                    if constexpr (!SINGLE) *reinterpret_cast<u32x4*>(lo_) = u32x4{cl[0], cl[1], cl[2], cl[3]};
                }
                if constexpr (RI >= 0 && pc == 6) x_issue(RI, xa[RI], xb[RI]);
                if constexpr (s8 == 1 && pc == 7) d_affine(dsa, dsb, dha, dhb);
            };
            if constexpr (DBG == 1) {
                w16u_static_for<8>([&](auto P) __attribute__((always_inline)) { piece(P); });
            } else {
                W16uHi& hc = (s8 & 1) ? h1 : h0;
                W16uHi& hn = (s8 & 1) ? h0 : h1;
                if (DBG == 3) { if (s8 == 0) { if (NT == 4) read_ext(buf, 0, e0); read_lo(buf, 0, lo); h1 = h0; } }
                else {
                    if (NT == 4) read_ext(buf, s8, e0);
                    if (!SINGLE) read_lo(buf, s8, lo);
                    if (s8 < 7) read_hi(buf, s8 + 1, hn);
                }
                __builtin_amdgcn_sched_barrier(0);
                const W16uHi& o = DBG == 3 ? h0 : hc;
                // the MFMAs in order, each followed by the piece of its position (DBG 5: all pieces behind the last MFMA instead)
                auto slot = [&](auto P) __attribute__((always_inline)) {
                    if constexpr (DBG != 5) { __builtin_amdgcn_sched_barrier(0); piece(P); __builtin_amdgcn_sched_barrier(0); }
                };
                const half8 ah = __builtin_bit_cast(half8, u32x4{o.a[0][0], o.a[0][1], o.a[1][0], o.a[1][1]});
                half8 bh[3], eh;
                shifted(o.x[0], o.x[1], o.x[2], bh);
                if constexpr (NT == 4) eh = __builtin_bit_cast(half8, u32x4{e0.h[0][0], e0.h[0][1], e0.h[1][0], e0.h[1][1]});
                constexpr int Q = NT == 4 ? 1 : 0;        // position shift behind each group's fourth MFMA
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[0], acc[0], 0, 0, 0); slot(ic_<0>{});
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[1], acc[1], 0, 0, 0); slot(ic_<1>{});
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[2], acc[2], 0, 0, 0); slot(ic_<2>{});
                if constexpr (NT == 4) { acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, eh, acc[3], 0, 0, 0); slot(ic_<3>{}); }
                if constexpr (!SINGLE) {
                    half8 bl[3];
                    shifted(lo.x[0], lo.x[1], lo.x[2], bl);
                    accl[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[0], accl[0], 0, 0, 0); slot(ic_<3 + Q>{});
                    accl[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[1], accl[1], 0, 0, 0); slot(ic_<4 + Q>{});
                    accl[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[2], accl[2], 0, 0, 0); slot(ic_<5 + Q>{});
                    if constexpr (NT == 4) {
                        accl[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(half8, u32x4{e0.l[0][0], e0.l[0][1], e0.l[1][0], e0.l[1][1]}), accl[3], 0, 0, 0);
                        slot(ic_<7>{});
                    }
                    const half8 al = __builtin_bit_cast(half8, u32x4{lo.a[0][0], lo.a[0][1], lo.a[1][0], lo.a[1][1]});
                    accl[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[0], accl[0], 0, 0, 0); slot(ic_<6 + 2 * Q>{});
                    accl[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[1], accl[1], 0, 0, 0); slot(ic_<7 + 2 * Q>{});
                    accl[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[2], accl[2], 0, 0, 0);
                    if constexpr (NT == 4) accl[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, eh, accl[3], 0, 0, 0);
                    if constexpr (DBG == 5) w16u_static_for<8>([&](auto P) __attribute__((always_inline)) { piece(P); });
                    // (positions 8 .. 10 of the four-tap waves carry no piece: 8 pieces in all)
                } else {
                    // one product: the pieces beyond the three / four MFMAs follow the last one
                    w16u_static_for<8 - 3 - Q>([&](auto P) __attribute__((always_inline)) { piece(ic_<decltype(P)::value + 3 + Q>{}); });
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (DBG != 4) __syncthreads();                    // (4: timing experiment without the brick barrier, no staging)
        cur ^= 1;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = j < 3 ? 3 * w + j : 24 + w;
        float* dst = p.part + (((size_t)blockIdx.x * gridDim.y + tp) * 27 + t) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r >> 2) * 8 + lh * 4 + (r & 3)) * 32 + l31] = acc[j][r] + accl[j][r] * (1.0f / W16_SPLIT);
    }
}

template <int DBG, bool SINGLE = false>
__global__ __launch_bounds__(512, 1) void wgrad16u_kernel(WgradParams p) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w < 3) wgrad16u_run<4, DBG, SINGLE>(p, w);
    else wgrad16u_run<3, DBG, SINGLE>(p, w);
}

// ---- wgrad16z: wgrad16u walking its bricks along z with the shared halo planes kept in LDS -------------------------------------
// A 2 x 8 x 8 brick needs the 4 x 10 x 10 halo of its input: 3.1 staged voxels per brick voxel, and the staging (activate, split,
// 9-10 vector instructions per element at 4 issue cycles each) costs the SIMDs about as many cycles as the MFMAs - the kernel is
// bound by vector issue, not by the matrix pipe.  Consecutive bricks of one (frame, y, x) column share two of their four halo
// planes, so here a workgroup walks whole columns: the X tile is a ring of six z-planes (four being read, the next brick's two new
// ones being written), and a brick stages 200 halo voxels instead of 400 - 1312 items instead of 2112 with the dY tile.  The first
// brick of a column stages its four planes in the open (once per 16-32 bricks).  Plane z of the column lives in slot (z + 1) mod 6.
#define WZ_PL (100 * 64)                  // one plane of one ring: 100 halo voxels x 32 fp16 channels
#define WZ_XR (6 * WZ_PL)                 // hi ring; the lo ring follows
#define WZ_D0 (2 * WZ_XR)                 // two dY buffers (hi 8 KB + lo 8 KB each)
#define WZ_DBUF (2 * WT_DB)
#define WZ_LDS (WZ_D0 + 2 * WZ_DBUF)

// H16: both operand tensors are stored as bfloat16 (16-bit storage mode): an item's eight channels are ONE 16-byte load, kept raw in
// the `a` register quad and widened (shift / mask) inside the conversion pieces - two more vector instructions per affine piece
template <int NT, int DBG, bool SINGLE, bool H16>
__device__ __forceinline__ void wgrad16z_run(const WgradParams& p, const int w) {
    extern __shared__ char lds8[];
    float* xtab = reinterpret_cast<float*>(lds8 + WZ_LDS);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int tp = blockIdx.y, mt = tp / p.n_tiles, nt = tp % p.n_tiles, m0 = mt * 32, n0 = nt * 32;
    f32x16 acc[NT], accl[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accl[j][r] = 0.f; }
    const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, cb = (lane >> 4) & 1;
    const int dz = w / 3, dy = w % 3;
    const int a_off = WZ_D0 + ((lh * 8 + q) * 32 + 16 * cb + 4 * pp) * 2;                        // dY tile: brick row 2 s + lh, voxel q
    const int x_off = (((lh + dy) * 10 + q) * 32 + 16 * cb + 4 * pp) * 2;                        // within a plane: row 2 (s & 3) + lh + dy, x = q - 1
    const int e_off = (((lh + 2) * 10 + q + min(w, 2)) * 32 + 16 * cb + 4 * pp) * 2;             // tap (2, 2, w)
    const int ncols = p.in.N * p.nby * p.nbx;
    const int oct = tid & 3;
    for (int i = tid; i < p.in.N * 64; i += 512) {
        const int n = i >> 6, which = (i >> 5) & 1, c = n0 + (i & 31);
        xtab[i] = p.in.scale ? (c < p.Nc ? (which ? p.in.shift : p.in.scale)[(size_t)n * p.in.C + c] : 0.f) : (which ? 0.f : 1.f);
    }
    // (the table is read by other threads than those that wrote it, first in the prologue below: without this barrier a wave that ran
    //  ahead converted its first items with whatever the previous workgroup left in LDS - a run-to-run difference of ~1e-4 in the
    //  weight gradient of one layer, one run in ~80 when the kernel ran alone and one in ~5 beside the main stream's kernels; round 4)
    __syncthreads();
    const int dummy_off = WZ_LDS + p.in.N * 256;
    const bool ca_ok = n0 + 8 * oct < p.Nc, cb_ok = n0 + 8 * oct + 4 < p.Nc;
    const bool ma_ok = m0 + 8 * oct < p.M, mb_ok = m0 + 8 * oct + 4 < p.M;
    int xn = 0, xoy = 0, xox = 0;
    // one 16-byte-pair item of the input: halo voxel (gz, xoy - 1 + hy, xox - 1 + hx); unconditional loads (wgrad16u)
    auto x_load = [&](int gz, int hy, int hx, bool exists, f32x4& a, f32x4& b, bool& in) __attribute__((always_inline)) {
        const int gy = xoy - 1 + hy, gx = xox - 1 + hx;
        in = (unsigned)gz < (unsigned)p.in.D && (unsigned)gy < (unsigned)p.in.H && (unsigned)gx < (unsigned)p.in.W && exists;
        const long long off = ((((long long)xn * p.in.D + gz) * p.in.H + gy) * p.in.W + gx) * (long long)p.in.C + n0 + 8 * oct;
        if constexpr (H16) a = *reinterpret_cast<const f32x4*>(in && ca_ok ? nm_eptr(p.in.p, (size_t)off, 1) : p.in.p);
        else {
            const float* src = p.in.p + off;
            a = *reinterpret_cast<const f32x4*>(in && ca_ok ? src : p.in.p); b = *reinterpret_cast<const f32x4*>(in && cb_ok ? src + 4 : p.in.p);
        }
    };
    auto d_issue = [&](int bz, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int bv = t >> 2, z = bv >> 6, y = (bv >> 3) & 7, x = bv & 7, m = m0 + 8 * (t & 3);
        const size_t eo = ((((size_t)xn * p.dy.D + 2 * bz + z) * p.dy.H + xoy + y) * p.dy.W + xox + x) * p.dy.C + m;
        if constexpr (H16) a = *reinterpret_cast<const f32x4*>(ma_ok ? nm_eptr(p.dy.p, eo, 1) : p.dy.p);
        else {
            const float* src = p.dy.p + eo;
            a = *reinterpret_cast<const f32x4*>(ma_ok ? src : p.dy.p); b = *reinterpret_cast<const f32x4*>(mb_ok ? src + 4 : p.dy.p);
        }
    };
    auto d_affine = [&](f32x4& sa, f32x4& sb, f32x4& ha, f32x4& hb) __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const bool has = p.dy.scale != nullptr;
        const size_t so = (size_t)xn * p.dy.C + m0 + 8 * (t & 3);
        const float* ps = has ? p.dy.scale + so : w16u_ident; const float* ph = has ? p.dy.shift + so : w16u_ident + 8;
        sa = *reinterpret_cast<const f32x4*>(ma_ok ? ps : w16u_ident); ha = *reinterpret_cast<const f32x4*>(ma_ok ? ph : w16u_ident + 8);
        sb = *reinterpret_cast<const f32x4*>(mb_ok ? ps + 4 : w16u_ident); hb = *reinterpret_cast<const f32x4*>(mb_ok ? ph + 4 : w16u_ident + 8);
    };
    // conversion of one item in the open (column starts): affine, leaky ReLU, zero mask, split, two 16-byte stores
    auto convert_store = [&](f32x4 a, f32x4 b, const f32x4& sa, const f32x4& sb, const f32x4& ha, const f32x4& hb, float slope, float ka, float kb,
                             char* hi, char* lo) __attribute__((always_inline)) {
        unsigned h[4], l[4];
        if constexpr (H16) {
            const unsigned u0 = nm_fbits(a[0]), u1 = nm_fbits(a[1]), u2 = nm_fbits(a[2]), u3 = nm_fbits(a[3]);
            a = f32x4{nm_bf_lo(u0), nm_bf_hi(u0), nm_bf_lo(u1), nm_bf_hi(u1)};
            b = f32x4{nm_bf_lo(u2), nm_bf_hi(u2), nm_bf_lo(u3), nm_bf_hi(u3)};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = fmaf(a[j], sa[j], ha[j]); b[j] = fmaf(b[j], sb[j], hb[j]);
            a[j] = fmaxf(a[j], a[j] * slope) * ka; b[j] = fmaxf(b[j], b[j] * slope) * kb;
        }
        h[0] = pack_split(a[0], a[1], l[0]); h[1] = pack_split(a[2], a[3], l[1]);
        h[2] = pack_split(b[0], b[1], l[2]); h[3] = pack_split(b[2], b[3], l[3]);
        *reinterpret_cast<u32x4*>(hi) = u32x4{h[0], h[1], h[2], h[3]};
        if constexpr (!SINGLE) *reinterpret_cast<u32x4*>(lo) = u32x4{l[0], l[1], l[2], l[3]};
    };
    // the in-loop X items of a thread: voxel (tid >> 2) + 128 k of the 200 of the two new planes, as packed (hx, hy, plane, exists)
    int npack[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v0 = (tid >> 2) + 128 * k, v = min(v0, 199), pl = v / 100, pv = v % 100;
        npack[k] = (pv % 10) | ((pv / 10) << 8) | (pl << 16) | ((v0 < 200 ? 1 : 0) << 24);
    }
    // (xb / db / eb: the lane's byte address of the stream in this brick's ring slot / dY buffer, one opaque register each - as an
    //  expression of bz the compiler forms one address register per read)
    auto read_hi = [&](int xb, int db, int s8, W16uHi& o) __attribute__((always_inline)) {
        const char* ap = lds8 + db + s8 * 1024;
        const char* lp = lds8 + xb + (2 * (s8 & 3) * 10) * 64;
        o.a[0] = tr_read(ap); o.a[1] = tr_read(ap + 4 * 64);
        o.x[0] = tr_read(lp); o.x[1] = tr_read(lp + 4 * 64); o.x[2] = tr_read(lp + 8 * 64);
    };
    auto shifted = [&](const u32x2& r0, const u32x2& r1, const u32x2& r2, half8 (&o)[3]) __attribute__((always_inline)) {
        o[0] = __builtin_bit_cast(half8, u32x4{r0[0], r0[1], r1[0], r1[1]});
        o[2] = __builtin_bit_cast(half8, u32x4{r0[1], r1[0], r1[1], r2[0]});
        o[1] = __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(r0[1], r0[0], 16), __builtin_amdgcn_alignbit(r1[0], r0[1], 16),
                                               __builtin_amdgcn_alignbit(r1[1], r1[0], 16), __builtin_amdgcn_alignbit(r2[0], r1[1], 16)});
    };
    constexpr bool STAGE = DBG != 2;
    float* dtab = reinterpret_cast<float*>(lds8 + WZ_LDS + p.in.N * 256 + 64);      // the dY tensor's pending affine of the column's frame: [scale 32][shift 32]
    const long long xstep = 2LL * p.in.H * p.in.W * p.in.C, dstep = 2LL * p.dy.H * p.dy.W * p.dy.C;    // one brick further along z

    int cur = 0;
    for (int col = blockIdx.x; col < ncols; col += p.S) {
        xn = col / (p.nby * p.nbx);
        { const int r = col % (p.nby * p.nbx); xoy = (r / p.nbx) * 8; xox = (r % p.nbx) * 8; }
        // per-thread constants of the column: element offsets of its in-loop items for the brick behind brick 0, (y, x) validity
        long long xoff[2], doff;
        int inyx = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int np = npack[k], gy = xoy - 1 + ((np >> 8) & 255), gx = xox - 1 + (np & 255);
            if ((unsigned)gy < (unsigned)p.in.H && (unsigned)gx < (unsigned)p.in.W && (np >> 24) != 0) inyx |= 1 << k;
            xoff[k] = ((((long long)xn * p.in.D + 3 + ((np >> 16) & 1)) * p.in.H + gy) * p.in.W + gx) * (long long)p.in.C + n0 + 8 * oct;
        }
        { const int bv = tid >> 2; doff = ((((long long)xn * p.dy.D + (bv >> 6)) * p.dy.H + xoy + ((bv >> 3) & 7)) * p.dy.W + xox + (bv & 7)) * (long long)p.dy.C + m0 + 8 * oct; }
        if (tid < 64) {
            const int which = tid >> 5, c = m0 + (tid & 31);
            dtab[tid] = p.dy.scale && c < p.M ? (which ? p.dy.shift : p.dy.scale)[(size_t)xn * p.dy.C + c] : (which ? 0.f : 1.f);
        }
        {   // column start: the four planes z = -1 .. 2 (slots 0 .. 3, contiguous) and the first dY tile, in the open, one item at a
            // time (the accumulators are live: no registers for a batch)
            {
                f32x4 da, db, dsa, dsb, dha, dhb;
                d_issue(0, da, db); d_affine(dsa, dsb, dha, dhb);
                convert_store(da, db, dsa, dsb, dha, dhb, p.dy.slope, ma_ok ? 1.f : 0.f, mb_ok ? 1.f : 0.f,
                              lds8 + WZ_D0 + cur * WZ_DBUF + tid * 16, lds8 + WZ_D0 + cur * WZ_DBUF + WT_DB + tid * 16);
            }
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const int hv0 = (tid >> 2) + 128 * k, hv = min(hv0, 399);
                f32x4 a, b; bool in;
                x_load(hv / 100 - 1, (hv / 10) % 10, hv % 10, hv0 < 400, a, b, in);
                const float* t = xtab + xn * 64 + 8 * oct;
                const f32x4 sa = *reinterpret_cast<const f32x4*>(t), sb = *reinterpret_cast<const f32x4*>(t + 4);
                const f32x4 ha = *reinterpret_cast<const f32x4*>(t + 32), hb = *reinterpret_cast<const f32x4*>(t + 36);
                const bool item = hv0 < 400;
                char* hi = item ? lds8 + tid * 16 + 8192 * k : lds8 + dummy_off; char* lo = item ? lds8 + tid * 16 + 8192 * k + WZ_XR : lds8 + dummy_off + 16;
                convert_store(a, b, sa, sb, ha, hb, p.in.slope, in && ca_ok ? 1.f : 0.f, in && cb_ok ? 1.f : 0.f, hi, lo);
            }
        }
        __syncthreads();
        for (int bz = 0; bz < p.nbz; ++bz) {
            // ring slots of this brick's operands: main line plane zb + dz, fourth tap plane zb + 2 (zb = k-step >> 2)
            const int sl = __builtin_amdgcn_readfirstlane((2 * bz) % 6);       // slot of plane 2 bz - 1 (even: 0, 2, 4)
            auto slot_off = [&](int k) __attribute__((always_inline)) { const int v = sl + k; return (v >= 6 ? v - 6 : v) * WZ_PL; };
            int xb0 = x_off + slot_off(dz), xb1 = x_off + slot_off(dz + 1), eb0 = e_off + slot_off(2), eb1 = e_off + slot_off(3);
            int dbs = a_off + cur * WZ_DBUF;
            asm volatile("" : "+v"(xb0), "+v"(xb1), "+v"(eb0), "+v"(eb1), "+v"(dbs));
            const int ndbuf = (cur ^ 1) * WZ_DBUF;
            const int ns0 = slot_off(4), ns1 = slot_off(5);                    // slots of the next brick's two new planes
            const int bzn = min(bz + 1, p.nbz - 1);        // (behind the column's last brick: its dY tile again, into the idle buffer)
            f32x4 da, db, xa[2], xb[2];
            W16uHi h0, h1;
            W16uLo lo;
            W16uExt e0;
            if (DBG != 1) read_hi(xb0, dbs, 0, h0);
            // The next brick's staging as 60 pieces of 4-6 vector instructions, one behind each MFMA (position g = 9 k-step + MFMA;
            // the one-product instantiation runs three per MFMA): a wave hides about five single-issue instructions in the 32 cycles
            // of a 32x32x16 MFMA, and blocks of 20+ (a whole element pair per gap) measured no better than the conversion as one
            // block behind the k-step's MFMAs.  g 0 dY request, 10-11 / 16-17 the two X requests, 18-31 / 32-45 / 46-59 the conversions of
            // dY / X0 / X1 (14 pieces each: table reads, 4 x (affine + slope, max + mask, split), stores).
            f32x4 ca, cb_, csa, cha;
            float cka = 0.f, ckb = 0.f, cslope = 1.f;
            f32x2 tt = f32x2{0.f, 0.f}, uu = tt;             // a channel pair through the affine / slope / mask as PACKED fp32 instructions (two elements per issue slot)
            unsigned ch[4], cl[4];
            const float* xsrc = nullptr; bool xin = false;
            auto piece = [&](auto GG) __attribute__((always_inline)) {
                constexpr int g = decltype(GG)::value;
                if constexpr (!STAGE || g < 0 || g >= 60) return;
                if constexpr (g == 0) {
                    if constexpr (H16) da = *reinterpret_cast<const f32x4*>(ma_ok ? nm_eptr(p.dy.p, (size_t)(doff + bzn * dstep), 1) : p.dy.p);
                    else {
                        const float* src = p.dy.p + (doff + bzn * dstep);
                        da = *reinterpret_cast<const f32x4*>(ma_ok ? src : p.dy.p); db = *reinterpret_cast<const f32x4*>(mb_ok ? src + 4 : p.dy.p);
                    }
                }
                if constexpr (g == 10 || g == 16) {
                    constexpr int k = g == 10 ? 0 : 1;
                    int np = npack[k];
                    asm volatile("" : "+v"(np));
                    xin = ((inyx >> k) & 1) != 0 && 2 * bz + 3 + ((np >> 16) & 1) < p.in.D;
                    xsrc = nm_eptr(p.in.p, (size_t)(xoff[k] + bz * xstep), H16);
                }
                if constexpr (g == 11 || g == 17) {
                    constexpr int k = g == 11 ? 0 : 1;
                    xa[k] = *reinterpret_cast<const f32x4*>(xin && ca_ok ? xsrc : p.in.p);
                    if constexpr (!H16) xb[k] = *reinterpret_cast<const f32x4*>(xin && cb_ok ? xsrc + 4 : p.in.p);
                }
                if constexpr (g >= 18) {
                    constexpr int item = (g - 18) / 14, c = (g - 18) % 14;          // item 0 = dY, 1 / 2 = X item 0 / 1
                    if constexpr (c == 0) {
                        const float* t;
                        if constexpr (item == 0) {
                            ca = da; if constexpr (!H16) cb_ = db;
                            cka = ma_ok ? 1.f : 0.f; ckb = mb_ok ? 1.f : 0.f; cslope = p.dy.slope;
                            t = dtab + 8 * oct;
                        } else {
                            ca = xa[item - 1]; if constexpr (!H16) cb_ = xb[item - 1];
                            int np = npack[item - 1];
                            asm volatile("" : "+v"(np));
                            const bool in = ((inyx >> (item - 1)) & 1) != 0 && 2 * bz + 3 + ((np >> 16) & 1) < p.in.D;
                            cka = in && ca_ok ? 1.f : 0.f; ckb = in && cb_ok ? 1.f : 0.f; cslope = p.in.slope;
                            t = xtab + xn * 64 + 8 * oct;
                        }
                        W16U_PIN4(ca); if constexpr (!H16) W16U_PIN4(cb_);
                        csa = *reinterpret_cast<const f32x4*>(t); cha = *reinterpret_cast<const f32x4*>(t + 32);
                    } else if constexpr (c <= 12) {
                        constexpr int j = (c - 1) / 3, part = (c - 1) % 3, e = 2 * (j & 1);
                        if constexpr (part == 0) {
                            // (one table pair of registers: the second channel quad's scale / shift replace the first's once it is through)
                            if constexpr (j == 2) {
                                const float* t = item == 0 ? dtab + 8 * oct : xtab + xn * 64 + 8 * oct;
                                csa = *reinterpret_cast<const f32x4*>(t + 4); cha = *reinterpret_cast<const f32x4*>(t + 36);
                            }
                            const f32x2 s2 = f32x2{csa[e], csa[e + 1]}, h2 = f32x2{cha[e], cha[e + 1]};
                            if constexpr (H16) {        // channel pair j of the raw item: dword j
                                const unsigned u = nm_fbits(ca[j]);
                                tt = f32x2{nm_bf_lo(u), nm_bf_hi(u)};
                            } else if constexpr (j < 2) tt = f32x2{ca[e], ca[e + 1]};
                            else tt = f32x2{cb_[e], cb_[e + 1]};
                            tt = __builtin_elementwise_fma(tt, s2, h2);
                            uu = tt * f32x2{cslope, cslope};
                        } else if constexpr (part == 1) {
                            const float k = j < 2 ? cka : ckb;
                            tt = f32x2{fmaxf(tt[0], uu[0]), fmaxf(tt[1], uu[1])} * f32x2{k, k};
                        } else ch[j] = pack_split(tt[0], tt[1], cl[j]);
                    } else {
                        char* hi; char* lo_;
                        if constexpr (item == 0) { hi = lds8 + WZ_D0 + ndbuf + tid * 16; lo_ = hi + WT_DB; }
                        else {
                            int np = npack[item - 1];
                            asm volatile("" : "+v"(np));
                            const int pv = ((np >> 8) & 255) * 10 + (np & 255);
                            hi = lds8 + ((np >> 16) & 1 ? ns1 : ns0) + pv * 64 + oct * 16; lo_ = hi + WZ_XR;
                            if constexpr (item == 2) { const bool it = (np >> 24) != 0; hi = it ? hi : lds8 + dummy_off; lo_ = it ? lo_ : lds8 + dummy_off + 16; }
                        }
                        *reinterpret_cast<u32x4*>(hi) = u32x4{ch[0], ch[1], ch[2], ch[3]};
                        if constexpr (!SINGLE) *reinterpret_cast<u32x4*>(lo_) = u32x4{cl[0], cl[1], cl[2], cl[3]};
                    }
                }
            };
            w16u_static_for<8>([&](auto S8) __attribute__((always_inline)) {
                constexpr int s8 = decltype(S8)::value;
                if constexpr (DBG == 1) {
                    w16u_static_for<9>([&](auto P) __attribute__((always_inline)) { piece(ic_<9 * s8 + decltype(P)::value>{}); });
                } else {
                    W16uHi& hc = (s8 & 1) ? h1 : h0;
                    W16uHi& hn = (s8 & 1) ? h0 : h1;
                    const int xbc = s8 < 4 ? xb0 : xb1, xbn = s8 + 1 < 4 ? xb0 : xb1, ebc = s8 < 4 ? eb0 : eb1;
                    const char* lpc = lds8 + xbc + (2 * (s8 & 3) * 10) * 64 + WZ_XR;                 // this k-step's lo halo row
                    const char* apc = lds8 + dbs + s8 * 1024 + WT_DB;                               //              lo dY operand
                    const char* lpn = lds8 + xbn + (2 * ((s8 + 1) & 3) * 10) * 64;                   // the next k-step's hi halo row
                    const char* apn = lds8 + dbs + (s8 + 1) * 1024;
                    const char* epc = lds8 + ebc + (2 * (s8 & 3) * 10) * 64;
                    // behind MFMA number `pos`: this k-step's lo operands (first used by MFMA 3 / 6), the fourth tap's operand (its
                    // MFMAs close the k-step), the next k-step's hi operands one read per gap - no block of 10-14 LDS reads in front of a
                    // k-step's first MFMA - and the staging piece of the position
                    auto slot = [&](auto P) __attribute__((always_inline)) {
                        constexpr int pos = decltype(P)::value;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!SINGLE) {
                            if constexpr (pos == 0) { lo.a[0] = tr_read(apc); lo.a[1] = tr_read(apc + 4 * 64); }
                            if constexpr (pos == 1) { lo.x[0] = tr_read(lpc); lo.x[1] = tr_read(lpc + 4 * 64); lo.x[2] = tr_read(lpc + 8 * 64); }
                            if constexpr (NT == 4 && pos == 2) { e0.h[0] = tr_read(epc); e0.h[1] = tr_read(epc + 4 * 64); }
                            if constexpr (NT == 4 && pos == 3) { e0.l[0] = tr_read(epc + WZ_XR); e0.l[1] = tr_read(epc + WZ_XR + 4 * 64); }
                            if constexpr (s8 < 7) {
                                if constexpr (pos == 4) hn.a[0] = tr_read(apn);
                                if constexpr (pos == 5) hn.a[1] = tr_read(apn + 4 * 64);
                                if constexpr (pos == 6) hn.x[0] = tr_read(lpn);
                                if constexpr (pos == 7) hn.x[1] = tr_read(lpn + 4 * 64);
                                if constexpr (pos == 8) hn.x[2] = tr_read(lpn + 8 * 64);
                            }
                            if constexpr (pos < 9) piece(ic_<9 * s8 + pos>{});
                        } else {
                            if constexpr (NT == 4 && pos == 0) { e0.h[0] = tr_read(epc); e0.h[1] = tr_read(epc + 4 * 64); }
                            if constexpr (s8 < 7 && pos == 1) { hn.a[0] = tr_read(apn); hn.a[1] = tr_read(apn + 4 * 64); }
                            if constexpr (s8 < 7 && pos == 2) { hn.x[0] = tr_read(lpn); hn.x[1] = tr_read(lpn + 4 * 64); hn.x[2] = tr_read(lpn + 8 * 64); }
                            piece(ic_<9 * s8 + 3 * pos>{}); piece(ic_<9 * s8 + 3 * pos + 1>{}); piece(ic_<9 * s8 + 3 * pos + 2>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    const half8 ah = __builtin_bit_cast(half8, u32x4{hc.a[0][0], hc.a[0][1], hc.a[1][0], hc.a[1][1]});
                    half8 bh[3];
                    shifted(hc.x[0], hc.x[1], hc.x[2], bh);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[0], acc[0], 0, 0, 0); slot(ic_<0>{});
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[1], acc[1], 0, 0, 0); slot(ic_<1>{});
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[2], acc[2], 0, 0, 0); slot(ic_<2>{});
                    if constexpr (!SINGLE) {
                        const half8 al = __builtin_bit_cast(half8, u32x4{lo.a[0][0], lo.a[0][1], lo.a[1][0], lo.a[1][1]});
                        accl[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[0], accl[0], 0, 0, 0); slot(ic_<3>{});
                        accl[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[1], accl[1], 0, 0, 0); slot(ic_<4>{});
                        accl[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[2], accl[2], 0, 0, 0); slot(ic_<5>{});
                        half8 bl[3];
                        shifted(lo.x[0], lo.x[1], lo.x[2], bl);
                        accl[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[0], accl[0], 0, 0, 0); slot(ic_<6>{});
                        accl[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[1], accl[1], 0, 0, 0); slot(ic_<7>{});
                        accl[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[2], accl[2], 0, 0, 0); slot(ic_<8>{});
                        if constexpr (NT == 4) {
                            const half8 eh = __builtin_bit_cast(half8, u32x4{e0.h[0][0], e0.h[0][1], e0.h[1][0], e0.h[1][1]});
                            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, eh, acc[3], 0, 0, 0);
                            accl[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, eh, accl[3], 0, 0, 0);
                            accl[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(half8, u32x4{e0.l[0][0], e0.l[0][1], e0.l[1][0], e0.l[1][1]}), accl[3], 0, 0, 0);
                        }
                    } else if constexpr (NT == 4) {
                        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(half8, u32x4{e0.h[0][0], e0.h[0][1], e0.h[1][0], e0.h[1][1]}), acc[3], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = j < 3 ? 3 * w + j : 24 + w;
        float* dst = p.part + (((size_t)blockIdx.x * gridDim.y + tp) * 27 + t) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r >> 2) * 8 + lh * 4 + (r & 3)) * 32 + l31] = acc[j][r] + accl[j][r] * (1.0f / W16_SPLIT);
    }
}

template <int DBG, bool SINGLE = false, bool H16 = false>
__global__ __launch_bounds__(512, 1) void wgrad16z_kernel(WgradParams p) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w < 3) wgrad16z_run<4, DBG, SINGLE, H16>(p, w);
    else wgrad16z_run<3, DBG, SINGLE, H16>(p, w);
}

// ---- first layer, sparse form ----------------------------------------------------------------------------------------------------
// dW[co][c][tap] of conv5(cat[occ, x1, x2, x3]): the three coordinate channels do not depend on the frame, so their gradient is
// the dense kernel (MODE 2) on ONE frame holding sum_n dy[n]; the occupancy channel is 1-3 % dense, so its gradient is a gather:
// every occupied voxel u adds occ[u] * dy[n, u - tap + 2, :] to the 125 tap rows.  Block = one chunk of one frame; thread =
// (output channel, tap group); the occupied voxels of a 256-voxel group are found with one ballot per wave and visited in index
// order by the whole block (uniform), per-block partials, fixed-order reduce: deterministic.
#define K5S_CHUNK 8192
template <int COUT>
__global__ __launch_bounds__(256) void wgrad_k5occ_sparse_kernel(const float* __restrict__ occ, const float* __restrict__ dy, int G, int chunks,
                                                                 float* __restrict__ part) {
    constexpr int TG = 256 / COUT, KT = (125 + TG - 1) / TG;
    __shared__ float sval[256];
    __shared__ unsigned long long smask[4];
    const int n = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int co = threadIdx.x % COUT, tg = threadIdx.x / COUT;
    const size_t G3 = (size_t)G * G * G;
    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    const size_t v0 = (size_t)ch * K5S_CHUNK, v1 = min(G3, v0 + K5S_CHUNK);
    for (size_t base = v0; base < v1; base += 256) {
        const size_t v = base + threadIdx.x;
        const float val = v < v1 ? occ[(size_t)n * G3 + v] : 0.f;
        __syncthreads();
        sval[threadIdx.x] = val;
        const unsigned long long m = __ballot(val != 0.f);
        if ((threadIdx.x & 63) == 0) smask[threadIdx.x >> 6] = m;
        __syncthreads();
        for (int wv = 0; wv < 4; ++wv) {
            unsigned long long mask = smask[wv];
            while (mask) {
                const int i = wv * 64 + __builtin_ctzll(mask);
                mask &= mask - 1;
                const float a = sval[i];
                const size_t u = base + i;
                const int ux = (int)(u % G), uy = (int)((u / G) % G), uz = (int)(u / ((size_t)G * G));
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const int tap = tg + TG * k;
                    if (tap < 125) {
                        const int oz = uz - tap / 25 + 2, oy = uy - (tap / 5) % 5 + 2, ox = ux - tap % 5 + 2;
                        if ((unsigned)oz < (unsigned)G && (unsigned)oy < (unsigned)G && (unsigned)ox < (unsigned)G)
                            acc[k] = fmaf(a, dy[((size_t)n * G3 + ((size_t)oz * G + oy) * G + ox) * COUT + co], acc[k]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int tap = tg + TG * k;
        if (tap < 125) part[((size_t)blockIdx.x * 125 + tap) * COUT + co] = acc[k];
    }
}
// ---- first layer, occupancy channel on the matrix cores with empty bricks skipped ----------------------------------------------
// dW_occ[co][tap] = sum_{n,v} dy[n,v,co] * occ[n, v + tap - 2] as an implicit GEMM with K = voxels (M = co, N = the 125 taps padded to
// 128, one 32-tap tile per wave), on v_mfma_f32_32x32x2_f32: both operands are single floats per lane straight from fp32 LDS tiles
// (dy brick [256 voxels][COUT], occupancy halo 8 x 12 x 12) - no split, no scaling, exact fp32 products.  A 4x8x8 brick whose halo
// holds no occupied voxel contributes exact zeros and is skipped before its dy is loaded: one figure in a 64^3 grid leaves ~85 % of the
// bricks empty, which is what makes the dense form cheaper than the gather above (3.3 ms per step: every occupied voxel pulls 125 x
// COUT dy values through L2 with no reuse between neighbours).  Persistent workgroups, accumulators across bricks, per-workgroup
// partials in the gather's layout, the same fixed-order reduce.
template <int COUT, bool HD = false>        // HD: dy is stored as bfloat16
__global__ __launch_bounds__(256) void wgrad_k5occ_mfma_kernel(const float* __restrict__ occ, const float* __restrict__ dy, int N, int G,
                                                               float* __restrict__ part) {
    extern __shared__ float lds[];
    constexpr int MT = COUT / 32, OT = 8 * 12 * 12;
    float* tile = lds;                    // occupancy halo [8][12][12]
    float* dys = lds + OT + 8;            // [256][COUT]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, w = tid >> 6;
    const int nb = G >> 3, nbz = G >> 2, per_frame = nbz * nb * nb, total = N * per_frame;
    const int tap = 32 * w + l31;
    const bool tv = tap < 125;
    const int toff = tv ? ((tap / 25) * 12 + (tap / 5) % 5) * 12 + tap % 5 : 0;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    for (int b = blockIdx.x; b < total; b += gridDim.x) {
        const int n = b / per_frame; int r = b % per_frame;
        const int bx = r % nb; r /= nb;
        const int oz0 = (r / nb) * 4, oy0 = (r % nb) * 8, ox0 = bx * 8;
        __syncthreads();                  // the previous brick's operand reads
        const float* src = occ + (size_t)n * G * G * G;
        // the five halo cells of a thread requested together from clamped addresses and masked afterwards (under `if (inside)` each load
        // had a full wait behind it: five dependent round trips per brick, and 85 % of the bricks are dropped right after this test)
        float ov[5]; bool in[5]; int nz = 0;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int i = min(tid + 256 * u, OT - 1);
            const int hx = i % 12, hy = (i / 12) % 12, hz = i / 144;
            const int gz = oz0 - 2 + hz, gy = oy0 - 2 + hy, gx = ox0 - 2 + hx;
            in[u] = tid + 256 * u < OT && (unsigned)gz < (unsigned)G && (unsigned)gy < (unsigned)G && (unsigned)gx < (unsigned)G;
            const int cz = min(max(gz, 0), G - 1), cy = min(max(gy, 0), G - 1), cx = min(max(gx, 0), G - 1);
            ov[u] = src[((size_t)cz * G + cy) * G + cx];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int i = tid + 256 * u;
            const float o = in[u] ? ov[u] : 0.f;
            if (i < OT) tile[i] = o;
            nz |= (o != 0.f);
        }
        if (!__syncthreads_or(nz)) continue;                       // (uniform over the workgroup; also the barrier after the tile)
        constexpr int C4 = COUT / 4, ITEMS = 256 * C4 / 256;       // 16-byte items per thread
#pragma unroll
        for (int i0 = 0; i0 < ITEMS; i0 += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + 256 * (i0 + u), vox = idx / C4, c4 = idx % C4;
                const int z = vox >> 6, y = (vox >> 3) & 7, x = vox & 7;
                v[u] = nm_ld4<HD>(dy, ((((size_t)n * G + oz0 + z) * G + oy0 + y) * G + ox0 + x) * COUT + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x4*>(dys + (size_t)(tid + 256 * (i0 + u)) * 4) = v[u];
        }
        __syncthreads();
#pragma unroll 8
        for (int s2 = 0; s2 < 128; ++s2) {
            const int v = 2 * s2 + lh;
            const float o = tile[(((v >> 6) * 12 + ((v >> 3) & 7)) * 12 + (v & 7)) + toff];
            const float bv = tv ? o : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[v * COUT + 32 * mt + l31], bv, acc[mt], 0, 0, 0);
        }
    }
    if (tv) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[((size_t)blockIdx.x * 125 + tap) * COUT + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[mt][r];
    }
}

// dW[co][0][tap] = sum_blocks part[blk][tap][co]
__global__ __launch_bounds__(256) void wgrad_k5occ_sparse_reduce_kernel(const float* __restrict__ part, int blocks, int Cout, float* __restrict__ dW) {
    const int total = 125 * Cout;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i % Cout, tap = i / Cout;
        // sixteen partials in flight, added in block order (one by one the sum was a chain of `blocks` L2 round trips: 148 us for 512)
        float s = 0.f;
        for (int b0 = 0; b0 < blocks; b0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(size_t)min(b0 + u, blocks - 1) * total + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (b0 + u < blocks) s += v[u];
        }
        dW[((size_t)co * 4) * 125 + tap] = s;
    }
}
// out[i] = sum_n x[n*per + i]
__global__ __launch_bounds__(256) void sum_frames4_kernel(const float* __restrict__ x, int N, size_t per4, float* __restrict__ out, int h) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per4; i += (size_t)gridDim.x * 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < N; ++n) s += nm_ld4(x, ((size_t)n * per4 + i) * 4, h);
        *reinterpret_cast<f32x4*>(out + i * 4) = s;
    }
}

// dW[m][c][tap] (+)= sum over slots; thread order: c fastest (coalesced partial reads)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int slots, int tiles, int n_tiles, int groups,
                                                           int M, int Cin, int taps, int k5occ, const float* __restrict__ mul,
                                                           float* __restrict__ dW) {
    const float mm = mul ? *mul : 1.0f;
    const int total = M * Cin * taps;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c = i % Cin, m = (i / Cin) % M, tap = i / (Cin * M);
        int tile, grp, col;
        if (k5occ) { tile = m >> 5; grp = tap / 5; col = (tap % 5) * 4 + c; }
        else { tile = (m >> 5) * n_tiles + (c >> 5); grp = tap; col = c & 31; }
        const float* src = part + ((size_t)tile * groups + grp) * 1024 + (m & 31) * 32 + col;
        const size_t stride = (size_t)tiles * groups * 1024;
        // eight independent chains (the slot loop is a chain of dependent L2 round trips otherwise), combined in a fixed order
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= slots; k += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += src[(size_t)(k + u) * stride];
        }
        for (int u = 0; k < slots; ++k, ++u) a[u] += src[(size_t)k * stride];
        const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        dW[((size_t)m * Cin + c) * taps + tap] = s * mm;
    }
}

// voxels per block of the partial-sum pass: 512, or 2048 for the large volumes (a 512-voxel block of a 32-channel layer lives ~4 us:
// the 64^3 layers ran 32768 such blocks per launch at 3 TB/s; derived from the voxel count alone so that every caller agrees)
__host__ __device__ inline int nm_gnb_vb(int voxels) { return voxels >= 32768 ? 2048 : 512; }
// grid (nblk, N)
__global__ __launch_bounds__(256) void gnb_partials_kernel(const float* __restrict__ dA, TensorRef y, int voxels, float* __restrict__ part,
                                                           const float* __restrict__ dmul) {
    __shared__ float sh[256 * 2];
    const float mm = dmul ? *dmul : 1.0f;          // dA arrives scaled by a power of two (nm_launch_make_scale): undone on read
    const int n = blockIdx.y, blk = blockIdx.x, C = y.C;
    const int lanes = 256 / C;
    const int c = threadIdx.x % C, vl = threadIdx.x / C;
    float s1 = 0.f, s2 = 0.f;
    if (vl < lanes) {
        const float sc = y.scale ? y.scale[(size_t)n * C + c] : 1.0f, shf = y.scale ? y.shift[(size_t)n * C + c] : 0.f;
        const int VB = nm_gnb_vb(voxels), v0 = blk * VB, v1 = min(voxels, v0 + VB);
        for (int v = v0 + vl; v < v1; v += lanes) {
            const size_t o = ((size_t)n * voxels + v) * C + c;
            const float yy = nm_ld1(y.p, o, y.h);
            const float z = fmaf(yy, sc, shf);
            const float dz = (nm_ld1(dA, o, y.h) * mm) * (z > 0.f ? 1.0f : y.slope);
            s1 += dz; s2 += dz * yy;
        }
    }
    sh[threadIdx.x * 2] = s1; sh[threadIdx.x * 2 + 1] = s2;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float a = 0.f, b = 0.f;
        for (int l = 0; l < lanes; ++l) { a += sh[(l * C + c) * 2]; b += sh[(l * C + c) * 2 + 1]; }
        float* dst = part + (((size_t)n * gridDim.x + blk) * C + c) * 2;
        dst[0] = a; dst[1] = b;
    }
}

// The same partial sums with 16-byte loads, four channels per thread and four voxels in flight (C % 4 == 0, C <= 256 with 1024 / C
// a whole number: every layer of the network).  The one-float-per-thread form above keeps two dependent 4-byte loads in flight per
// thread and reached 0.54 of the HBM rate on the 64^3 layers (2 reads of 2.1 GB in 1.6 ms).
// (dv, wv: dA given as the outer product dv[n][voxel] * wv[channel] instead of a tensor - the decoder's last conv layer, whose
//  incoming gradient is the 1x1 tail conv's weight row times one scalar per voxel: 67 MB instead of 2.1 GB, read twice)
// H: y and dA are stored as bfloat16 (16-bit storage mode): 8-byte loads of the same four channels
template <bool H>
__global__ __launch_bounds__(256) void gnb_partials4_kernel(const float* __restrict__ dA, TensorRef y, int voxels, float* __restrict__ part,
                                                            const float* __restrict__ dmul, const float* __restrict__ dv, const float* __restrict__ wv) {
    __shared__ f32x4 sh[256 * 2];
    const float mm = dmul ? *dmul : 1.0f;
    const int n = blockIdx.y, blk = blockIdx.x, C = y.C, C4 = C >> 2;
    const int lanes = 256 / C4;
    const int c = (threadIdx.x % C4) * 4, vl = threadIdx.x / C4;
    f32x4 s1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, s2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (vl < lanes) {
        f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, shf = f32x4{0.f, 0.f, 0.f, 0.f};
        if (y.scale) { sc = *reinterpret_cast<const f32x4*>(y.scale + (size_t)n * C + c); shf = *reinterpret_cast<const f32x4*>(y.shift + (size_t)n * C + c); }
        const int VB = nm_gnb_vb(voxels), v0 = blk * VB, v1 = min(voxels, v0 + VB);
        const float* yp = nm_eptr(y.p, (size_t)n * voxels * C + c, H);
        const float* dp = nm_eptr(dA, (size_t)n * voxels * C + c, H);
        const float* dvp = dv + (size_t)n * voxels;
        f32x4 w4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (dv) w4 = *reinterpret_cast<const f32x4*>(wv + c);
        auto grad = [&](int vv) __attribute__((always_inline)) {
            if (dv) { const float d = dvp[vv]; return f32x4{d * w4[0], d * w4[1], d * w4[2], d * w4[3]}; }
            return nm_ld4<H>(dp, (size_t)vv * C);
        };
        auto add = [&](const f32x4& yy, const f32x4& dd, int u) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = fmaf(yy[j], sc[j], shf[j]);
                const float dz = (dd[j] * mm) * (z > 0.f ? 1.0f : y.slope);
                s1[u][j] += dz; s2[u][j] += dz * yy[j];
            }
        };
        int v = v0 + vl;
        // (bfloat16: eight voxels in flight - the same bytes per thread as four fp32 ones; with four the 64^3 layers ran at 2.1 TB/s)
        constexpr int U = H ? 8 : 4;
        for (; v + (U - 1) * lanes < v1; v += U * lanes) {
            f32x4 yy[U], dd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                yy[u] = nm_ld4<H>(yp, (size_t)(v + u * lanes) * C);
                dd[u] = grad(v + u * lanes);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) add(yy[u], dd[u], u & 1);
        }
        // (the tail alternates the two accumulators like the main loop: the order of the sums does not depend on U)
        for (int u = 0; v < v1; v += lanes, ++u) add(nm_ld4<H>(yp, (size_t)v * C), grad(v), u & 1);
    }
    sh[threadIdx.x * 2] = s1[0] + s1[1]; sh[threadIdx.x * 2 + 1] = s2[0] + s2[1];
    __syncthreads();
    if ((int)threadIdx.x < C4) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
        for (int l = 0; l < lanes; ++l) { a += sh[(l * C4 + threadIdx.x) * 2]; b += sh[(l * C4 + threadIdx.x) * 2 + 1]; }
        float* dst = part + (((size_t)n * gridDim.x + blk) * C + c) * 2;
        *reinterpret_cast<f32x4*>(dst) = f32x4{a[0], b[0], a[1], b[1]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{a[2], b[2], a[3], b[3]};
    }
}

// one block per (frame, group); cpg <= 64 channels per group; 1024 threads: 1024 / cpg lanes walk each channel's partial blocks
// NM_GNBF_T threads: 1024 when the forward partial sums have to be walked as well, 256 when the forward finalisation left the
// per-channel totals (a 256-thread block with 8 KB of LDS finds a CU beside the other stream's persistent convolutions; the
// 1024-thread block waited for a whole CU: 49 us average in the two-stream step for microseconds of work)
template <int NM_GNBF_T>
__global__ __launch_bounds__(NM_GNBF_T) void gnb_finalize_kernel(const float* __restrict__ bpart, int nblk_b, const float* __restrict__ fpart,
                                                           int nblk_f, int C, int groups, int voxels, const float* __restrict__ gamma,
                                                           float eps, float* __restrict__ coef, float* __restrict__ dgn,
                                                           const double* __restrict__ chsum) {
    __shared__ double red[NM_GNBF_T * 4];
    __shared__ double chan[64 * 4];      // per channel: sum y, sum y^2, S1, S2
    const int n = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups;
    const int L = NM_GNBF_T / cpg;       // partial lanes per channel
    const int cl = threadIdx.x % cpg, ln = threadIdx.x / cpg, c = g * cpg + cl;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (ln < L) {
        // (the forward sums: from the forward finalisation when it left them, else a second walk over the conv's partials)
        if (!chsum) for (int b = ln; b < nblk_f; b += L) { const float* q = fpart + (((size_t)n * nblk_f + b) * C + c) * 2; a0 += q[0]; a1 += q[1]; }
        for (int b = ln; b < nblk_b; b += L) { const float* q = bpart + (((size_t)n * nblk_b + b) * C + c) * 2; a2 += q[0]; a3 += q[1]; }
    }
    red[threadIdx.x * 4] = a0; red[threadIdx.x * 4 + 1] = a1; red[threadIdx.x * 4 + 2] = a2; red[threadIdx.x * 4 + 3] = a3;
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
        double s[4] = {0, 0, 0, 0};
        for (int l = 0; l < L; ++l)
            for (int j = 0; j < 4; ++j) s[j] += red[(l * cpg + threadIdx.x) * 4 + j];
        if (chsum) { s[0] = chsum[((size_t)n * C + c) * 2]; s[1] = chsum[((size_t)n * C + c) * 2 + 1]; }
        for (int j = 0; j < 4; ++j) chan[threadIdx.x * 4 + j] = s[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
        const double Mcount = (double)voxels * cpg;
        double sy = 0, syy = 0;
        for (int j = 0; j < cpg; ++j) { sy += chan[j * 4]; syy += chan[j * 4 + 1]; }
        const double mean = sy / Mcount;
        double var = syy / Mcount - mean * mean;
        if (var < 0.0) var = 0.0;
        const double r = 1.0 / sqrt(var + (double)eps);
        double m1 = 0, m2 = 0;
        for (int j = 0; j < cpg; ++j) {
            const double gm = gamma[g * cpg + j];
            m1 += gm * chan[j * 4 + 2];
            m2 += gm * r * (chan[j * 4 + 3] - mean * chan[j * 4 + 2]);
        }
        m1 /= Mcount; m2 /= Mcount;
        const double S1 = chan[cl * 4 + 2], S2 = chan[cl * 4 + 3], sumy = chan[cl * 4];
        const double c1 = r * gamma[c], c2 = -r * r * m2, c3 = -r * m1 + r * r * m2 * mean;
        float* co = coef + ((size_t)n * C + c) * 4;
        co[0] = (float)c1; co[1] = (float)c2; co[2] = (float)c3; co[3] = 0.f;
        float* d = dgn + ((size_t)n * C + c) * 4;
        d[0] = (float)(r * (S2 - mean * S1));
        d[1] = (float)S1;
        d[2] = (float)(c1 * S1 + c2 * sumy + c3 * (double)voxels);    // sum_v dy: gradient of the conv bias
        d[3] = 0.f;
    }
}

__global__ void sum_frames_kernel(const float* __restrict__ src, int N, int C, int stride, int off, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += src[((size_t)n * C + c) * stride + off];
    out[c] = s;
}

// dgn[n][c] = (dgamma_n, dbeta_n, dbias_n, .) -> the three per-channel gradients in one launch
__global__ void sum_frames3_kernel(const float* __restrict__ dgn, int N, int C, float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    // four independent chains over the frames (16-byte loads), combined in a fixed order
    f32x4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int n = 0;
    for (; n + 4 <= N; n += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += *reinterpret_cast<const f32x4*>(dgn + ((size_t)(n + u) * C + c) * 4);
    }
    for (int u = 0; n < N; ++n, ++u) a[u] += *reinterpret_cast<const f32x4*>(dgn + ((size_t)n * C + c) * 4);
    const f32x4 t = (a[0] + a[1]) + (a[2] + a[3]);
    o0[c] = t[0]; o1[c] = t[1]; o2[c] = t[2];
}

// the same sums for up to NM_SUM3_JOBS layers in one launch (blockIdx.y = layer): the per-layer results are parameter gradients, which
// nothing in the backward walk reads - launched one by one they sat in the dependent chain of every GroupNorm layer
__global__ void sum_frames3_multi_kernel(NmSum3Jobs jobs) {
    const NmSum3Job& J = jobs.j[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x, N = J.N, C = J.C;
    if (c >= C) return;
    const float* __restrict__ dgn = J.dgn;
    f32x4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int n = 0;
    for (; n + 4 <= N; n += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += *reinterpret_cast<const f32x4*>(dgn + ((size_t)(n + u) * C + c) * 4);
    }
    for (int u = 0; n < N; ++n, ++u) a[u] += *reinterpret_cast<const f32x4*>(dgn + ((size_t)n * C + c) * 4);
    const f32x4 t = (a[0] + a[1]) + (a[2] + a[3]);
    J.o0[c] = t[0]; J.o1[c] = t[1]; J.o2[c] = t[2];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int rows, int C, float* __restrict__ out) {
    __shared__ double sh[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)part[((size_t)r * C + c) * 2];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) out[c] = (float)sh[0];
}

// block max of |v| -> one integer atomicMax per block (non-negative floats order like their bit patterns: deterministic)
__device__ __forceinline__ void block_absmax(float m, unsigned* amax) {
    __shared__ float shm[256];
    shm[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) shm[threadIdx.x] = fmaxf(shm[threadIdx.x], shm[threadIdx.x + st]); __syncthreads(); }
    if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(shm[0]));
}

__global__ __launch_bounds__(256) void gnb_apply_kernel(const float* __restrict__ dA, TensorRef y, const float* __restrict__ coef,
                                                        float* __restrict__ dy, unsigned* __restrict__ amax, const float* __restrict__ dmul,
                                                        const float* __restrict__ dv, const float* __restrict__ wv) {
    float mx = 0.f;
    const float mm = dmul ? *dmul : 1.0f;
    // grid (blocks, frames): 32-bit index arithmetic inside a frame (no 64-bit division per 16-byte item)
    const unsigned per_frame = (unsigned)y.D * y.H * y.W * y.C;
    const size_t n = blockIdx.y;
    for (unsigned r = (blockIdx.x * 256u + threadIdx.x) * 4u; r < per_frame; r += gridDim.x * 1024u) {
        const size_t e = n * per_frame + r;
        const int c = (int)(r % (unsigned)y.C);
        const f32x4 yy = nm_ld4(y.p, e, y.h);
        f32x4 d;
        if (dv) {
            const float dd = dv[n * (per_frame / (unsigned)y.C) + r / (unsigned)y.C];
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wv + c);
            d = f32x4{dd * w4[0], dd * w4[1], dd * w4[2], dd * w4[3]};
        } else d = nm_ld4(dA, e, y.h);
        d[0] *= mm; d[1] *= mm; d[2] *= mm; d[3] *= mm;
        if (y.slope != 1.0f) {
            f32x4 z = yy;
            if (y.scale) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(y.scale + n * y.C + c);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(y.shift + n * y.C + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = fmaf(yy[j], sc[j], sh[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = z[j] > 0.f ? d[j] : d[j] * y.slope;
        }
        if (coef) {
            const float* cf = coef + (n * y.C + c) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = fmaf(cf[j * 4], d[j], fmaf(cf[j * 4 + 1], yy[j], cf[j * 4 + 2]));
        }
        nm_st4(dy, e, d, y.h);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
    }
    if (amax) block_absmax(mx, amax);
}

// gnb_apply_kernel for the layers whose channel count divides 1024 (every layer of the network): a thread keeps ONE channel quad for the
// whole launch, so the per-channel operands (forward scale / shift, the three coefficients, the outer-product weights) are loaded
// once instead of per 16-byte item, and four items' tensor loads are in flight before the first is used.  The generic kernel's loop
// is one item per iteration behind three dependent waits (tensor loads, scale / shift, coefficients) and runs at 4.9 TB/s on
// occupancy alone.  Identity operands (scale 1 / shift 0, slope 1) leave the values as they are; same arithmetic per element.
template <bool RANK1, bool COEF, bool H, unsigned U = 4u>
__global__ __launch_bounds__(256) void gnb_apply4_kernel(const float* __restrict__ dA, TensorRef y, const float* __restrict__ coef,
                                                         float* __restrict__ dy, unsigned* __restrict__ amax, const float* __restrict__ dmul,
                                                         const float* __restrict__ dv, const float* __restrict__ wv) {
    float mx = 0.f;
    const float mm = dmul ? *dmul : 1.0f;
    const unsigned C = (unsigned)y.C, per_frame = (unsigned)y.D * y.H * y.W * C;
    const size_t n = blockIdx.y;
    const unsigned r0 = (blockIdx.x * 256u + threadIdx.x) * 4u, step = gridDim.x * 1024u;      // (step % C == 0: the quad is fixed)
    const unsigned c = r0 % C;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, w4 = sh, c1 = sc, c2 = sh, c3 = sh;
    if (y.scale) { sc = *reinterpret_cast<const f32x4*>(y.scale + n * C + c); sh = *reinterpret_cast<const f32x4*>(y.shift + n * C + c); }
    if (RANK1) w4 = *reinterpret_cast<const f32x4*>(wv + c);
    if (COEF) {
        const float* cf = coef + (n * C + c) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c1[j] = cf[j * 4]; c2[j] = cf[j * 4 + 1]; c3[j] = cf[j * 4 + 2]; }
    }
    const float slope = y.slope;
    const float* yp = nm_eptr(y.p, n * per_frame, H);
    const float* dp = nm_eptr(dA, n * per_frame, H);
    const float* dvp = dv + n * (per_frame / C);
    float* op = nm_eptr(dy, n * per_frame, H);
    auto one = [&](const f32x4& yy, f32x4 d, unsigned r) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[j] *= mm;
            const float z = fmaf(yy[j], sc[j], sh[j]);
            d[j] = z > 0.f ? d[j] : d[j] * slope;
            if (COEF) d[j] = fmaf(c1[j], d[j], fmaf(c2[j], yy[j], c3[j]));
        }
        nm_st4<H>(op, r, d);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
    };
    unsigned r = r0;
    // U items in flight (bfloat16: eight 8-byte loads = the bytes of four fp32 ones; NM355_GNB_U8=0: four, A/B)
    for (; r + (U - 1u) * step < per_frame && r + (U - 1u) * step >= r; r += U * step) {
        f32x4 yy[U], dd[U];
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            const unsigned ru = r + u * step;
            yy[u] = nm_ld4<H>(yp, ru);
            if (RANK1) { const float q = dvp[ru / C]; dd[u] = f32x4{q * w4[0], q * w4[1], q * w4[2], q * w4[3]}; }
            else dd[u] = nm_ld4<H>(dp, ru);
        }
#pragma unroll
        for (unsigned u = 0; u < U; ++u) one(yy[u], dd[u], r + u * step);
    }
    for (; r < per_frame; r += step) {
        const f32x4 yy = nm_ld4<H>(yp, r);
        f32x4 dd;
        if (RANK1) { const float q = dvp[r / C]; dd = f32x4{q * w4[0], q * w4[1], q * w4[2], q * w4[3]}; }
        else dd = nm_ld4<H>(dp, r);
        one(yy, dd, r);
    }
    if (amax) block_absmax(mx, amax);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n4, unsigned* __restrict__ amax, const float* __restrict__ dmul, int h) {
    const float mm = dmul ? fabsf(*dmul) : 1.0f;
    float mx = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 d = nm_ld4(x, i * 4, h);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
    }
    block_absmax(mx * mm, amax);
}

// scale[i] = 2^k with max * 2^k in (128, 256] (so that the fp16 hi/lo split of the data-gradient conv sees O(100) operands,
// far from the fp16 denormal range); sc2[0] = 2^k, sc2[1] = 2^-k
__global__ void make_scale_kernel(const unsigned* __restrict__ amax, int count, float* __restrict__ scale, float* __restrict__ sc2,
                                  float* __restrict__ shift) {
    const float mx = __uint_as_float(*amax);
    float sc = 1.0f;
    // exponent clamped to +-100: for a vanishing (denormal-sized) or huge gradient the scale and its reciprocal must both stay finite
    if (mx > 0.f && mx < INFINITY) { int e; frexpf(mx, &e); sc = ldexpf(1.0f, min(max(8 - e, -100), 100)); }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) { scale[i] = sc; if (shift) shift[i] = 0.f; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc2[0] = sc; sc2[1] = 1.0f / sc; }
}

__global__ __launch_bounds__(256) void scale_by_kernel(float* __restrict__ x, size_t n4, const float* __restrict__ mul, int h) {
    const float m = *mul;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 d = nm_ld4(x, i * 4, h);
        d[0] *= m; d[1] *= m; d[2] *= m; d[3] *= m;
        nm_st4(x, i * 4, d, h);
    }
}

// per axis the fine indices 2j-1 .. 2j+2 touch coarse j with weights 0.25, 0.75, 0.75, 0.25; at the borders the clamped
// interpolation folds the missing neighbour's share onto the edge voxel: j = 0 -> (none, 1, 0.75, 0.25), j = I-1 -> (0.25, 0.75, 1, none)
__device__ __forceinline__ void up_adj_w(int j, int I, float (&w)[4]) {
    w[0] = j > 0 ? 0.25f : 0.f; w[1] = j > 0 ? 0.75f : 1.0f;
    w[2] = j < I - 1 ? 0.75f : 1.0f; w[3] = j < I - 1 ? 0.25f : 0.f;
}

__global__ __launch_bounds__(256) void upsample2_adjoint_kernel(const float* __restrict__ dfine, int N, int D, int H, int W, int C,
                                                                float* __restrict__ dcoarse, const float* __restrict__ mul, int hf, int hc) {
    const float mm = mul ? *mul : 1.0f;
    const int cq = C / 4;
    const size_t total = (size_t)N * D * H * W * cq;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % cq); size_t r = i / cq;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); r /= H;
        const int z = (int)(r % D); const size_t n = r / D;
        float wz[4], wy[4], wx[4];
        up_adj_w(z, D, wz); up_adj_w(y, H, wy); up_adj_w(x, W, wx);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (wz[a] == 0.f) continue;
            const size_t fz = (size_t)(2 * z + a - 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (wy[b] == 0.f) continue;
                const size_t fy = (size_t)(2 * y + b - 1);
                const float wzy = wz[a] * wy[b];
                const size_t row = (((n * 2 * D + fz) * 2 * H + fy) * 2 * W) * C + q * 4;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    if (wx[cc] == 0.f) continue;
                    const f32x4 v = nm_ld4(dfine, row + (size_t)(2 * x + cc - 1) * C, hf);
                    const float wt = wzy * wx[cc];
                    acc[0] += wt * v[0]; acc[1] += wt * v[1]; acc[2] += wt * v[2]; acc[3] += wt * v[3];
                }
            }
        }
        acc[0] *= mm; acc[1] *= mm; acc[2] *= mm; acc[3] *= mm;
        nm_st4(dcoarse, i * 4, acc, hc);
    }
}

// The same adjoint with the fine gradient staged through LDS: a coarse brick of 2 x 4 x 8 voxels needs the fine region 6 x 10 x 18,
// 16 channels at a time (69 KB); every fine value is then read once from memory instead of by the 8 coarse voxels it feeds (the
// gather above makes 64 16-byte requests per output quad and is bound by L2 -> L1 traffic: 2.3 ms for the 4.3 GB input).
#define UA_BZ 2
#define UA_BY 4
#define UA_BX 8
#define UA_FZ 6
#define UA_FY 10
#define UA_FX 18
// HF / HC: storage type of the fine / coarse tensor as template parameters (a load under a run-time flag is a branch, and the 17 loads
// of a thread's batch then complete one by one: 1.8 ms per launch instead of 0.5).
// bfloat16 fine tensor (HF): a pass covers 32 channels instead of 16 - the same 4320 16-byte items (8 channels each) fill the same 69 KB
// tile, kept RAW in LDS and widened (shift / mask) when read; a voxel's 64 contiguous bytes are fetched by four lanes (with 16 channels
// per pass the bfloat16 tensor gave 32-byte pieces at a 128-byte stride: half the sectors of every request unused) and the tensor is
// walked in half as many passes.
template <bool HF, bool HC>
__global__ __launch_bounds__(256) void upsample2_adjoint_tile_kernel(const float* __restrict__ dfine, int N, int D, int H, int W, int C,
                                                                     float* __restrict__ dcoarse, const float* __restrict__ mul) {
    __shared__ f32x4 tile[UA_FZ * UA_FY * UA_FX * 4];
    const float mm = mul ? *mul : 1.0f;
    const int tid = threadIdx.x;
    const int nbz = D / UA_BZ, nby = H / UA_BY, nbx = W / UA_BX;
    int r = blockIdx.x;
    const int bx = r % nbx; r /= nbx;
    const int by = r % nby; r /= nby;
    const int bz = r % nbz; const size_t n = r / nbz;
    const int z0 = bz * UA_BZ, y0 = by * UA_BY, x0 = bx * UA_BX;
    const int fz0 = 2 * z0 - 1, fy0 = 2 * y0 - 1, fx0 = 2 * x0 - 1;
    const int FD = 2 * D, FH = 2 * H, FW = 2 * W;
    // this thread's output: coarse voxel (tid >> 2) of the brick, item tid & 3 of the chunk (4 channels, 8 with a bfloat16 fine tensor)
    const int q = tid & 3, v = tid >> 2, lx = v & 7, ly = (v >> 3) & 3, lz = v >> 5;
    float wz[4], wy[4], wx[4];
    up_adj_w(z0 + lz, D, wz); up_adj_w(y0 + ly, H, wy); up_adj_w(x0 + lx, W, wx);
    constexpr int ITEMS = UA_FZ * UA_FY * UA_FX * 4, PER = (ITEMS + 255) / 256;       // 4320 16-byte items, 17 per thread
    constexpr int CH = HF ? 32 : 16, IC = HF ? 8 : 4;                                  // channels per pass / per item
    for (int c0 = 0; c0 < C; c0 += CH) {
        __syncthreads();
        f32x4 reg[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + 256 * k;
            const int iq = i & 3, fv = i >> 2;
            const int fx = fv % UA_FX, fy = (fv / UA_FX) % UA_FY, fz = fv / (UA_FX * UA_FY);
            const int gz = fz0 + fz, gy = fy0 + fy, gx = fx0 + fx;
            reg[k] = f32x4{0.f, 0.f, 0.f, 0.f};                  // (all-zero bits are zeros in both element types)
            if (i < ITEMS && (unsigned)gz < (unsigned)FD && (unsigned)gy < (unsigned)FH && (unsigned)gx < (unsigned)FW) {
                const size_t e = (((n * FD + gz) * FH + gy) * FW + gx) * C + c0 + IC * iq;
                if constexpr (HF) reg[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(dfine) + e);
                else reg[k] = *reinterpret_cast<const f32x4*>(dfine + e);
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int i = tid + 256 * k; if (i < ITEMS) tile[i] = reg[k]; }
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float wzy = wz[a] * wy[b];
                const f32x4* row = tile + (((2 * lz + a) * UA_FY + 2 * ly + b) * UA_FX + 2 * lx) * 4 + q;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const f32x4 t = row[cc * 4];
                    const float wt = wzy * wx[cc];
                    if constexpr (HF) {
                        const unsigned u0 = nm_fbits(t[0]), u1 = nm_fbits(t[1]), u2 = nm_fbits(t[2]), u3 = nm_fbits(t[3]);
                        acc[0] += wt * nm_bf_lo(u0); acc[1] += wt * nm_bf_hi(u0); acc[2] += wt * nm_bf_lo(u1); acc[3] += wt * nm_bf_hi(u1);
                        acc2[0] += wt * nm_bf_lo(u2); acc2[1] += wt * nm_bf_hi(u2); acc2[2] += wt * nm_bf_lo(u3); acc2[3] += wt * nm_bf_hi(u3);
                    } else { acc[0] += wt * t[0]; acc[1] += wt * t[1]; acc[2] += wt * t[2]; acc[3] += wt * t[3]; }
                }
            }
        acc[0] *= mm; acc[1] *= mm; acc[2] *= mm; acc[3] *= mm;
        const size_t eo = (((n * D + z0 + lz) * H + y0 + ly) * W + x0 + lx) * C + c0 + IC * q;
        nm_st4<HC>(dcoarse, eo, acc);
        if constexpr (HF) { acc2[0] *= mm; acc2[1] *= mm; acc2[2] *= mm; acc2[3] *= mm; nm_st4<HC>(dcoarse, eo + 4, acc2); }
    }
}

// The same adjoint walking z: upsample2_adjoint_tile_kernel reads a fine region of 6 x 10 x 18 voxels for 4 x 8 x 16 it owns - 2.1x the
// tensor, and the launch ran at HBM rate / 2.1 (2.7 TB/s of useful bytes on the 64^3 layer: 1.58 ms for 4.3 GB, 1.05 ms in bfloat16).  The
// adjoint is separable: a workgroup takes a coarse (y, x) tile of 4 x 8 and ONE 128-byte channel chunk (32 fp32 / 64 bfloat16 channels:
// whole cache lines - with 64-byte chunks on the grid two workgroups fetched every line at different times, 2.05 ms), walks the fine
// planes z = 0 .. 2D - 1, reduces each plane in (y, x) from a 10 x 18 LDS tile (1.41x the plane's own voxels, every plane loaded once
// per column) and combines the reduced planes along z in registers: plane 2m feeds coarse m (tap 1) and completes coarse m - 1 (tap 3),
// plane 2m + 1 feeds coarse m (tap 2) and starts coarse m + 1 (tap 0).  Planes are double-buffered: the next plane's loads are in flight
// under the current one's sums.
#define UZ_BY 4
#define UZ_BX 8
#define UZ_FY 10
#define UZ_FX 18
template <bool HF, bool HC>
__global__ __launch_bounds__(256) void upsample2_adjoint_zwalk_kernel(const float* __restrict__ dfine, int N, int D, int H, int W, int C,
                                                                      float* __restrict__ dcoarse, const float* __restrict__ mul) {
    __shared__ f32x4 tile[2][UZ_FY * UZ_FX * 8];
    const float mm = mul ? *mul : 1.0f;
    const int tid = threadIdx.x;
    const int nby = H / UZ_BY, nbx = W / UZ_BX;
    int r = blockIdx.x;
    const int bx = r % nbx; r /= nbx;
    const int by = r % nby; const size_t n = r / nby;
    constexpr int CH = HF ? 64 : 32, IC = HF ? 8 : 4;                         // channels per workgroup (128 bytes per voxel) / per 16-byte item
    const int c0 = blockIdx.y * CH;
    const int y0 = by * UZ_BY, x0 = bx * UZ_BX, fy0 = 2 * y0 - 1, fx0 = 2 * x0 - 1;
    const int FD = 2 * D, FH = 2 * H, FW = 2 * W;
    const int q = tid & 7, v = tid >> 3, lx = v & 7, ly = v >> 3;
    float wy[4], wx[4];
    up_adj_w(y0 + ly, H, wy); up_adj_w(x0 + lx, W, wx);
    constexpr int ITEMS = UZ_FY * UZ_FX * 8, PER = (ITEMS + 255) / 256;       // 1440 16-byte items per plane, 6 per thread (the last partly)
    int ioff[PER]; unsigned ivalid = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + 256 * k, iq = i & 7, fv = i >> 3;
        const int fx = fv % UZ_FX, fy = fv / UZ_FX, gy = fy0 + fy, gx = fx0 + fx;
        const bool ok = i < ITEMS && (unsigned)gy < (unsigned)FH && (unsigned)gx < (unsigned)FW;
        ioff[k] = ok ? (gy * FW + gx) * C + c0 + IC * iq : 0;
        if (ok) ivalid |= 1u << k;
    }
    const size_t plane = (size_t)FH * FW * C, frame = n * FD * plane;
    f32x4 reg[PER];
    auto load_plane = [&](int fz) __attribute__((always_inline)) {
        const size_t base = frame + (size_t)fz * plane;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            reg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((ivalid >> k) & 1u) {
                if constexpr (HF) reg[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(dfine) + base + ioff[k]);
                else reg[k] = *reinterpret_cast<const f32x4*>(dfine + base + ioff[k]);
            }
        }
    };
    auto store_plane = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int i = tid + 256 * k; if (i < ITEMS) tile[buf][i] = reg[k]; }
    };
    // sum of the plane over this thread's 4 x 4 fine (y, x) taps (two channel quads with a bfloat16 fine tensor)
    auto inplane = [&](int buf, f32x4& a0, f32x4& a1) __attribute__((always_inline)) {
        a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4* row = tile[buf] + ((2 * ly + b) * UZ_FX + 2 * lx) * 8 + q;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const f32x4 t = row[cc * 8];
                const float wt = wy[b] * wx[cc];
                if constexpr (HF) {
                    const unsigned u0 = nm_fbits(t[0]), u1 = nm_fbits(t[1]), u2 = nm_fbits(t[2]), u3 = nm_fbits(t[3]);
                    a0[0] += wt * nm_bf_lo(u0); a0[1] += wt * nm_bf_hi(u0); a0[2] += wt * nm_bf_lo(u1); a0[3] += wt * nm_bf_hi(u1);
                    a1[0] += wt * nm_bf_lo(u2); a1[1] += wt * nm_bf_hi(u2); a1[2] += wt * nm_bf_lo(u3); a1[3] += wt * nm_bf_hi(u3);
                } else { a0[0] += wt * t[0]; a0[1] += wt * t[1]; a0[2] += wt * t[2]; a0[3] += wt * t[3]; }
            }
        }
    };
    auto emit = [&](int m, f32x4 a0, f32x4 a1) __attribute__((always_inline)) {
        const size_t eo = (((n * D + m) * H + y0 + ly) * W + x0 + lx) * C + c0 + IC * q;
        a0[0] *= mm; a0[1] *= mm; a0[2] *= mm; a0[3] *= mm;
        nm_st4<HC>(dcoarse, eo, a0);
        if constexpr (HF) { a1[0] *= mm; a1[1] *= mm; a1[2] *= mm; a1[3] *= mm; nm_st4<HC>(dcoarse, eo + 4, a1); }
    };
    // this workgroup's coarse planes [m0, m1) (gridDim.z splits of the column: the short columns of the small layers would leave the chip
    // with a handful of long serial chains) and the fine planes that feed them
    const int mper = D / (int)gridDim.z, m0 = (int)blockIdx.z * mper, m1 = m0 + mper;
    const int p0 = m0 > 0 ? 2 * m0 - 1 : 0, p1 = m1 < D ? 2 * m1 : FD - 1;
    load_plane(p0); store_plane(0);
    __syncthreads();
    f32x4 prev0 = {0.f, 0.f, 0.f, 0.f}, prev1 = prev0, cur0 = prev0, cur1 = prev0, nxt0 = prev0, nxt1 = prev0;
    int b = 0;
    for (int fz = p0; fz <= p1; ++fz) {
        const bool more = fz < p1;
        if (more) load_plane(fz + 1);
        f32x4 r0, r1;
        inplane(b, r0, r1);
        const int m = fz >> 1;
        float wa[4] = {0.f, 0.f, 0.f, 0.f}, wb[4] = {0.f, 0.f, 0.f, 0.f};
        if (fz & 1) {          // odd plane 2m + 1: tap 2 of coarse m, tap 0 of coarse m + 1
            if (m >= m0 && m < m1) up_adj_w(m, D, wa);
            if (m + 1 >= m0 && m + 1 < m1) up_adj_w(m + 1, D, wb);
#pragma unroll
            for (int j = 0; j < 4; ++j) { cur0[j] += wa[2] * r0[j]; cur1[j] += wa[2] * r1[j]; nxt0[j] = wb[0] * r0[j]; nxt1[j] = wb[0] * r1[j]; }
        } else {               // even plane 2m: tap 1 of coarse m, tap 3 of coarse m - 1 (which is then complete)
            if (m >= m0 && m < m1) up_adj_w(m, D, wa);
            if (m - 1 >= m0 && m - 1 < m1) up_adj_w(m - 1, D, wb);
#pragma unroll
            for (int j = 0; j < 4; ++j) { cur0[j] += wa[1] * r0[j]; cur1[j] += wa[1] * r1[j]; prev0[j] += wb[3] * r0[j]; prev1[j] += wb[3] * r1[j]; }
            if (m - 1 >= m0 && m - 1 < m1) emit(m - 1, prev0, prev1);
        }
        if (more) store_plane(b ^ 1);
        __syncthreads();
        b ^= 1;
        if (fz & 1) { prev0 = cur0; prev1 = cur1; cur0 = nxt0; cur1 = nxt1; }
    }
    if (m1 == D) emit(D - 1, prev0, prev1);          // (the last coarse plane has no tap-3 plane: weight 0 at the border)
}

__global__ void flip_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int csel, int taps, float* __restrict__ out) {
    const int total = csel * Cout * taps;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int t = i % taps, co = (i / taps) % Cout, ci = i / (taps * Cout);
        out[i] = w[((size_t)co * Cin + ci) * taps + (taps - 1 - t)];
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n, int h) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) nm_st1(dst, i, nm_ld1(dst, i, h) + nm_ld1(src, i, h), h);
}
// dst = dst * (*dmul) + src * (*smul)   (null multiplier = 1): the un-scaling of a data-gradient conv's result folded into the add
// that joins it with the other branch (one pass instead of scale_by + axpy)
__global__ __launch_bounds__(256) void axpby4_kernel(float* __restrict__ dst, const float* __restrict__ dmul, const float* __restrict__ src,
                                                     const float* __restrict__ smul, size_t n4, int h) {
    const float a = dmul ? *dmul : 1.0f, b = smul ? *smul : 1.0f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 d = nm_ld4(dst, i * 4, h);
        const f32x4 x = nm_ld4(src, i * 4, h);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = fmaf(d[j], a, x[j] * b);
        nm_st4(dst, i * 4, d, h);
    }
}

int grid_for(size_t work_items) { return (int)min((work_items + 255) / 256, (size_t)(256 * 16)); }

struct WgradPlan { WgradParams p; int mode, m_tiles, slots; size_t lds, ws_floats; };

// split-fp16 kernel: 3x3x3, stride 1, "same" padding, output extents multiples of the 2 x 8 x 8 brick
bool wgrad16_eligible(int OD, int OH, int OW, int ks, int stride) {
    return ks == 3 && stride == 1 && OD % 2 == 0 && OH % 8 == 0 && OW % 8 == 0;
}

WgradPlan plan_wgrad(int N, int OD, int OH, int OW, int M, int Nc, int ks, int stride, bool k5occ, bool f16 = false) {
    WgradPlan q;
    WgradParams& p = q.p;
    p.ks = ks; p.stride = stride; p.M = M; p.Nc = Nc;
    q.mode = k5occ ? 2 : (ks == 1 ? 1 : 0);
    int bz = 4, by = 8, bx = 8;
    if (f16) {
        q.mode = 3;
        p.BZ = 2; p.BY = 8; p.BX = 8; p.nbz = OD / 2; p.nby = OH / 8; p.nbx = OW / 8; p.HZ = 4; p.HY = 10; p.HX = 10;
        q.m_tiles = (M + 31) / 32; p.n_tiles = (Nc + 31) / 32; p.groups = 27;
        const int tiles = q.m_tiles * p.n_tiles, total = N * p.nbz * p.nby * p.nbx;
        // workgroups of a launch: one per CU, or - when the weight gradients run on their own stream beside the main stream's walk
        // (nm_net.hip conv_bwd) - 224, which leaves 32 CUs to the GroupNorm passes and small kernels of the walk: a whole-CU kernel on
        // every CU admits nothing beside it until it ends (73.0 -> 71.3 ms per training step; 208: 71.7, 240: 72.9, 192: 72.7; in the
        // reduced-precision mode 256 stays better: 53.4 vs 53.9).  NM355_WGRAD_WGS overrides.
        const int wgs = nm_ls().wgrad_wgs > 0 ? nm_ls().wgrad_wgs : (nm_ls().wgrad_async == 1 && !nm_ls().single ? 224 : 256);
        p.S = max(1, min(total, max(tiles, wgs) / tiles));
        // wgrad16z_kernel hands out whole (frame, y, x) columns: no more workgroups per tile pair than columns
        if (nm_ls().wgrad_tr && nm_ls().wgrad_z && p.nbz >= 2 && N <= 96) p.S = max(1, min(p.S, N * p.nby * p.nbx));
        q.slots = p.S; q.lds = W16_LDS; q.ws_floats = (size_t)q.slots * tiles * 27 * 1024;
        return q;
    }
    if (stride == 2) { bz = 2; by = 4; bx = 8; }
    p.BX = min(bx, (OW + 1) & ~1); p.BY = min(by, OH); p.BZ = min(bz, OD);
    p.nbx = (OW + p.BX - 1) / p.BX; p.nby = (OH + p.BY - 1) / p.BY; p.nbz = (OD + p.BZ - 1) / p.BZ;
    p.HZ = (p.BZ - 1) * stride + ks; p.HY = (p.BY - 1) * stride + ks; p.HX = (p.BX - 1) * stride + ks;
    q.m_tiles = (M + 31) / 32;
    p.n_tiles = k5occ ? 1 : (Nc + 31) / 32;
    p.groups = k5occ ? 25 : ks * ks * ks;
    const int tiles = q.m_tiles * p.n_tiles;
    const int total = N * p.nbz * p.nby * p.nbx;
    p.S = max(1, min(total, 512 / tiles));
    q.slots = q.mode == 1 ? p.S * 4 : p.S;
    q.lds = ((size_t)p.BZ * p.BY * p.BX * 32 + (size_t)p.HZ * p.HY * p.HX * (k5occ ? 4 : 32)) * sizeof(float);
    q.ws_floats = (size_t)q.slots * tiles * p.groups * 1024;
    return q;
}

template <int MODE, int NTW, bool HX = false, bool HD = false>
int launch_wgrad_t(const WgradPlan& q, hipStream_t s) {
    {   // the kernel's per-thread item tables: 8 dY items, 20 input items (5 in MODE 2)
        const int BV = q.p.BZ * q.p.BY * q.p.BX, HV = q.p.HZ * q.p.HY * q.p.HX;
        if (BV > 256 || (MODE == 2 ? HV > 5 * 256 : HV * 8 > 20 * 256) || q.p.HX > 255 || q.p.HY > 255 || q.p.HZ > 255) {
            nm_set_error("wgrad: brick %dx%dx%d / halo %dx%dx%d beyond the kernel's item tables", q.p.BZ, q.p.BY, q.p.BX, q.p.HZ, q.p.HY, q.p.HX);
            return NM_ERR_ARG;
        }
    }
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<MODE, NTW, HX, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            nm_set_error("wgrad: cannot raise the dynamic LDS limit"); return NM_ERR_HIP;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL((wgrad_kernel<MODE, NTW, HX, HD>), dim3(q.p.S, q.m_tiles * q.p.n_tiles), dim3(256), q.lds, s, q.p);
    return nm_check_hip(hipGetLastError(), "wgrad launch");
}


int launch_wgrad16(const WgradPlan& q, hipStream_t s) {
    // 16-bit storage: both operands bfloat16, on the z-walking kernel only (every layer of >= 32^3 voxels has nbz >= 16)
    const bool h16 = q.p.in.h || q.p.dy.h;
    if (h16 && !(q.p.in.h && q.p.dy.h && nm_conv_single() && nm_ls().wgrad_tr && nm_ls().wgrad_z && q.p.nbz >= 2 && q.p.in.N <= 96)) {
        nm_set_error("wgrad16: bfloat16 operands (in %d, dy %d) need conv mode 4, both tensors bfloat16 and the wgrad16z kernel", q.p.in.h, q.p.dy.h);
        return NM_ERR_UNSUPPORTED;
    }
    if (nm_ls().wgrad_tr && q.p.in.N <= 96) {                       // (the per-frame scale / shift table of wgrad16t_kernel lives in LDS: 256 B per frame)
        static NmDeviceOnce attr_t;
        if (!attr_t.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16t_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16t_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16t_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16t_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16u_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16u_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16u_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16u_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16u_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16u_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16u_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                nm_set_error("wgrad16t: cannot raise the dynamic LDS limit"); return NM_ERR_HIP;
            }
            attr_t.mark();
        }
        const size_t ldsb = WT_LDS + (size_t)q.p.in.N * 64 * sizeof(float) + 64;
        const dim3 grid(q.p.S, q.m_tiles * q.p.n_tiles);
        if (nm_ls().wgrad_z && q.p.nbz >= 2) {
            static NmDeviceOnce attr_z;
            if (!attr_z.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16z_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16z_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16z_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16z_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16z_kernel<0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                    nm_set_error("wgrad16z: cannot raise the dynamic LDS limit"); return NM_ERR_HIP;
                }
                attr_z.mark();
            }
            const WgradParams& pz = q.p;                  // (plan_wgrad bounded S by the column count)
            const size_t ldsz = WZ_LDS + (size_t)q.p.in.N * 64 * sizeof(float) + 64 + 256;   // tiles, X affine table, dummy item, dY affine table
            const dim3 gz(pz.S, q.m_tiles * q.p.n_tiles);
            if (h16) hipLaunchKernelGGL((wgrad16z_kernel<0, true, true>), gz, dim3(512), ldsz, s, pz);
            else if (q.p.dbg == 1) hipLaunchKernelGGL(wgrad16z_kernel<1>, gz, dim3(512), ldsz, s, pz);
            else if (q.p.dbg == 2) hipLaunchKernelGGL(wgrad16z_kernel<2>, gz, dim3(512), ldsz, s, pz);
            else if (nm_conv_single()) hipLaunchKernelGGL((wgrad16z_kernel<0, true>), gz, dim3(512), ldsz, s, pz);
            else hipLaunchKernelGGL(wgrad16z_kernel<0>, gz, dim3(512), ldsz, s, pz);
            return nm_check_hip(hipGetLastError(), "wgrad16z launch");
        }
        if (nm_ls().wgrad_u) {
            if (q.p.dbg == 1) hipLaunchKernelGGL(wgrad16u_kernel<1>, grid, dim3(512), ldsb, s, q.p);
            else if (q.p.dbg == 2) hipLaunchKernelGGL(wgrad16u_kernel<2>, grid, dim3(512), ldsb, s, q.p);
            else if (q.p.dbg == 3) hipLaunchKernelGGL(wgrad16u_kernel<3>, grid, dim3(512), ldsb, s, q.p);
            else if (q.p.dbg == 4) hipLaunchKernelGGL(wgrad16u_kernel<4>, grid, dim3(512), ldsb, s, q.p);
            else if (q.p.dbg == 5) hipLaunchKernelGGL(wgrad16u_kernel<5>, grid, dim3(512), ldsb, s, q.p);
            else if (nm_conv_single()) hipLaunchKernelGGL((wgrad16u_kernel<0, true>), grid, dim3(512), ldsb, s, q.p);
            else hipLaunchKernelGGL(wgrad16u_kernel<0>, grid, dim3(512), ldsb, s, q.p);
            return nm_check_hip(hipGetLastError(), "wgrad16u launch");
        }
        if (q.p.dbg == 1) hipLaunchKernelGGL(wgrad16t_kernel<1>, grid, dim3(512), ldsb, s, q.p);
        else if (q.p.dbg == 2) hipLaunchKernelGGL(wgrad16t_kernel<2>, grid, dim3(512), ldsb, s, q.p);
        else if (nm_conv_single()) hipLaunchKernelGGL((wgrad16t_kernel<0, true>), grid, dim3(512), ldsb, s, q.p);
        else hipLaunchKernelGGL(wgrad16t_kernel<0>, grid, dim3(512), ldsb, s, q.p);
        return nm_check_hip(hipGetLastError(), "wgrad16t launch");
    }
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            nm_set_error("wgrad16: cannot raise the dynamic LDS limit"); return NM_ERR_HIP;
        }
        attr_set.mark();
    }
    if (nm_conv_single()) hipLaunchKernelGGL(wgrad16_kernel<true>, dim3(q.p.S, q.m_tiles * q.p.n_tiles), dim3(256), q.lds, s, q.p);
    else hipLaunchKernelGGL(wgrad16_kernel<false>, dim3(q.p.S, q.m_tiles * q.p.n_tiles), dim3(256), q.lds, s, q.p);
    return nm_check_hip(hipGetLastError(), "wgrad16 launch");
}

int run_wgrad(WgradPlan& q, float* ws, float* dW, int cin_real, int taps, hipStream_t s, const float* mul = nullptr) {
    if (q.lds > 160 * 1024) { nm_set_error("wgrad: LDS tile of %zu bytes", q.lds); return NM_ERR_UNSUPPORTED; }
    q.p.part = ws;
    int rc;
    const bool h16 = q.p.in.h || q.p.dy.h;
    if (h16 && !(q.mode == 3 || (q.mode == 0 && q.p.ks == 2 && q.p.in.h) || (q.mode == 2 && !q.p.in.h))) {
        nm_set_error("wgrad: no kernel for bfloat16 operands (in %d, dy %d) with ks=%d mode=%d", q.p.in.h, q.p.dy.h, q.p.ks, q.mode); return NM_ERR_UNSUPPORTED;
    }
    // (the split-fp16 k3 weight gradients are a profiler family of their own, nm_prof_kernel_name 13: kernel + its fixed-order reduce)
    NmProfScope prof(s, q.mode == 3 ? 2.0 * q.p.in.N * (double)q.p.dy.D * q.p.dy.H * q.p.dy.W * q.p.M * (double)cin_real * taps : 0.0, 13);
    if (q.mode == 3) rc = launch_wgrad16(q, s);
    else if (q.mode == 2) rc = q.p.dy.h ? launch_wgrad_t<2, 7, false, true>(q, s) : launch_wgrad_t<2, 7>(q, s);
    else if (q.mode == 1) rc = launch_wgrad_t<1, 1>(q, s);
    else if (q.p.ks == 2 && h16) rc = q.p.dy.h ? launch_wgrad_t<0, 2, true, true>(q, s) : launch_wgrad_t<0, 2, true, false>(q, s);
    else if (q.p.ks == 2) rc = launch_wgrad_t<0, 2>(q, s);
    else if (q.p.ks == 3) rc = launch_wgrad_t<0, 7>(q, s);
    else { nm_set_error("wgrad: ks=%d unsupported", q.p.ks); return NM_ERR_UNSUPPORTED; }
    if (rc) return rc;
    const int tiles = q.m_tiles * q.p.n_tiles;
    const int total = q.p.M * cin_real * taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(min((total + 255) / 256, 4096)), dim3(256), 0, s, ws, q.slots, tiles, q.p.n_tiles,
                       q.p.groups, q.p.M, cin_real, taps, q.mode == 2 ? 1 : 0, mul, dW);
    return nm_check_hip(hipGetLastError(), "wgrad reduce launch");
}

}  // namespace

#define K1_WGS 512            // (256 / 512 / 1024 workgroups: 2.36 / 1.92 / 1.84 ms per step over the three instantiations - one serial brick chain per workgroup; the reduce grows with it)
static bool k1_wide_eligible(int OD, int OH, int OW, int M, int Nc, int ks, int stride) {
    const int m_tiles = (M + 31) / 32, n_tiles = (Nc + 31) / 32;
    return ks == 1 && stride == 1 && (OD * OH * OW) % 64 == 0 && m_tiles * n_tiles <= 24 && (size_t)64 * (m_tiles * 32 + n_tiles * 32 + 8) * 4 <= 150 * 1024;
}

// wgrad16k2_kernel: k2 s2 p0 layers whose fine grid is exactly twice the coarse one, whole 8 x 8 coarse (y, x) bricks, <= 128 dY channels
static bool k2f16_shape_eligible(int OD, int OH, int OW, int M, int ks, int stride) {
    return nm_ls().wgrad_k2f16 && ks == 2 && stride == 2 && OH % 8 == 0 && OW % 8 == 0 && OD > 0 && M <= 128;
}
static int k2f16_slots(int N, int OD, int OH, int OW, int Nc) {
    const int n_tiles = (Nc + 31) / 32, total = N * OD * (OH / 8) * (OW / 8);
    return max(1, min(total, 512 / n_tiles));
}
template <int MT, bool SINGLE, bool HX, bool HD>
static int launch_k2f16_t(const WgradParams& p, dim3 grid, hipStream_t s) {
    static NmDeviceOnce attr_set;
    const size_t ldsb = w2_lds_bytes(MT, SINGLE);
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16k2_kernel<MT, SINGLE, HX, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            nm_set_error("wgrad16k2: cannot raise the dynamic LDS limit"); return NM_ERR_HIP;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL((wgrad16k2_kernel<MT, SINGLE, HX, HD>), grid, dim3(256), ldsb, s, p);
    return nm_check_hip(hipGetLastError(), "wgrad16k2 launch");
}

size_t nm_wgrad_ws_floats(int N, int OD, int OH, int OW, int M, int Nc, int ks, int stride) {
    size_t a = plan_wgrad(N, OD, OH, OW, M, Nc, ks, stride, false).ws_floats;
    if (k1_wide_eligible(OD, OH, OW, M, Nc, ks, stride)) a = max(a, (size_t)K1_WGS * ((M + 31) / 32) * ((Nc + 31) / 32) * 1024);
    if (wgrad16_eligible(OD, OH, OW, ks, stride)) a = max(a, plan_wgrad(N, OD, OH, OW, M, Nc, ks, stride, false, true).ws_floats);
    if (k2f16_shape_eligible(OD, OH, OW, M, ks, stride)) a = max(a, (size_t)k2f16_slots(N, OD, OH, OW, Nc) * ((M + 31) / 32) * ((Nc + 31) / 32) * 8 * 1024);
    return a;
}
static int k5s_chunks(int G) { return (int)(((size_t)G * G * G + K5S_CHUNK - 1) / K5S_CHUNK); }
size_t nm_wgrad_k5occ_ws_floats(int N, int G, int M) {
    const size_t G3 = (size_t)G * G * G;
    const size_t dense = plan_wgrad(N, G, G, G, M, 4, 5, 1, true).ws_floats;                      // exact-shape fallback (other Cout)
    const size_t sparse = G3 * M + G3 + 64 + plan_wgrad(1, G, G, G, M, 4, 5, 1, true).ws_floats + (size_t)N * k5s_chunks(G) * 125 * M + 256;
    return max(dense, sparse);
}

int nm_launch_wgrad(const TensorRef& in, const TensorRef& dy, int ks, int stride, int pad, int cin_real, float* ws, float* dW,
                    hipStream_t s, const float* mul, int allow_f16) {
    if (in.C % 4 || dy.C % 4 || in.N != dy.N) { nm_set_error("wgrad: channel counts must be multiples of 4 and frame counts equal"); return NM_ERR_ARG; }
    const bool f16 = allow_f16 && pad == 1 && wgrad16_eligible(dy.D, dy.H, dy.W, ks, stride) && in.D == dy.D && in.H == dy.H && in.W == dy.W;
    if (k1_wide_eligible(dy.D, dy.H, dy.W, dy.C, in.C, ks, stride) && pad == 0 && in.D == dy.D && in.H == dy.H && in.W == dy.W) {
        WgradParams p; p.in = in; p.dy = dy; p.part = ws; p.ks = 1; p.stride = 1; p.pad = 0; p.M = dy.C; p.Nc = in.C;
        const int m_tiles = (dy.C + 31) / 32, n_tiles = (in.C + 31) / 32, T = m_tiles * n_tiles;
        const int bricks = in.N * (dy.D * dy.H * dy.W / 64), S = min(bricks, K1_WGS);
        const size_t ldsb = (size_t)64 * (m_tiles * 32 + 4 + n_tiles * 32 + 4) * sizeof(float);
        static NmDeviceOnce attr_set;
        if (!attr_set.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k1_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k1_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k1_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k1_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k1_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k1_kernel<6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                nm_set_error("wgrad_k1: cannot raise the dynamic LDS limit"); return NM_ERR_HIP;
            }
            attr_set.mark();
        }
        if (in.h != dy.h) { nm_set_error("wgrad_k1: the two tensors must have the same element type (in %d, dy %d)", in.h, dy.h); return NM_ERR_UNSUPPORTED; }
        if (in.h) {
            if (T <= 4) hipLaunchKernelGGL((wgrad_k1_kernel<1, true>), dim3(S), dim3(256), ldsb, s, p, m_tiles, n_tiles);
            else if (T <= 8) hipLaunchKernelGGL((wgrad_k1_kernel<2, true>), dim3(S), dim3(256), ldsb, s, p, m_tiles, n_tiles);
            else hipLaunchKernelGGL((wgrad_k1_kernel<6, true>), dim3(S), dim3(256), ldsb, s, p, m_tiles, n_tiles);
        } else
        if (T <= 4) hipLaunchKernelGGL(wgrad_k1_kernel<1>, dim3(S), dim3(256), ldsb, s, p, m_tiles, n_tiles);
        else if (T <= 8) hipLaunchKernelGGL(wgrad_k1_kernel<2>, dim3(S), dim3(256), ldsb, s, p, m_tiles, n_tiles);
        else hipLaunchKernelGGL(wgrad_k1_kernel<6>, dim3(S), dim3(256), ldsb, s, p, m_tiles, n_tiles);
        int rc = nm_check_hip(hipGetLastError(), "wgrad_k1 launch");
        if (rc) return rc;
        const int totalw = dy.C * cin_real;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(min((totalw + 255) / 256, 4096)), dim3(256), 0, s, ws, S, T, n_tiles, 1, dy.C, cin_real, 1, 0, mul, dW);
        return nm_check_hip(hipGetLastError(), "wgrad reduce launch");
    }
    if (allow_f16 && pad == 0 && k2f16_shape_eligible(dy.D, dy.H, dy.W, dy.C, ks, stride) && in.D == 2 * dy.D && in.H == 2 * dy.H && in.W == 2 * dy.W &&
        (!in.h || in.C % 8 == 0) && (!dy.h || (dy.C % 8 == 0 && in.h)) && (!(in.h || dy.h) || nm_conv_single())) {
        WgradParams p; p.in = in; p.dy = dy; p.part = ws; p.ks = 2; p.stride = 2; p.pad = 0; p.M = dy.C; p.Nc = in.C; p.dbg = 0;
        const int m_tiles = (dy.C + 31) / 32, n_tiles = (in.C + 31) / 32;
        p.n_tiles = n_tiles; p.groups = 8;
        p.S = k2f16_slots(in.N, dy.D, dy.H, dy.W, in.C);
        const dim3 grid(p.S, n_tiles);
        const bool single = nm_conv_single();
        int rc;
        if (m_tiles <= 2) {
            if (!single) rc = launch_k2f16_t<2, false, false, false>(p, grid, s);
            else if (dy.h) rc = launch_k2f16_t<2, true, true, true>(p, grid, s);
            else if (in.h) rc = launch_k2f16_t<2, true, true, false>(p, grid, s);
            else rc = launch_k2f16_t<2, true, false, false>(p, grid, s);
        } else {
            if (!single) rc = launch_k2f16_t<4, false, false, false>(p, grid, s);
            else if (dy.h) rc = launch_k2f16_t<4, true, true, true>(p, grid, s);
            else if (in.h) rc = launch_k2f16_t<4, true, true, false>(p, grid, s);
            else rc = launch_k2f16_t<4, true, false, false>(p, grid, s);
        }
        if (rc) return rc;
        const int totalw = dy.C * cin_real * 8;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(min((totalw + 255) / 256, 4096)), dim3(256), 0, s, ws, p.S, m_tiles * n_tiles, n_tiles, 8, dy.C, cin_real, 8, 0, mul, dW);
        return nm_check_hip(hipGetLastError(), "wgrad reduce launch");
    }
    WgradPlan q = plan_wgrad(in.N, dy.D, dy.H, dy.W, dy.C, in.C, ks, stride, false, f16);
    q.p.in = in; q.p.dy = dy; q.p.pad = pad;
    static const int dbg = getenv("NM355_W16_DBG") ? atoi(getenv("NM355_W16_DBG")) : 0;
    q.p.dbg = dbg;
    return run_wgrad(q, ws, dW, cin_real, ks * ks * ks, s, mul);
}

int nm_launch_wgrad_k5occ(const float* occ, int N, int G, const TensorRef& dy, float* ws, float* dW, hipStream_t s, int sparse_occ,
                          hipStream_t s_coord, hipEvent_t ev_fork, hipEvent_t ev_join) {
    if (dy.C % 4 || dy.D != G) { nm_set_error("wgrad_k5occ: bad dy"); return NM_ERR_ARG; }
    TensorRef in; in.p = occ; in.scale = in.shift = nullptr; in.slope = 1.0f; in.N = N; in.D = in.H = in.W = G; in.C = 1;
    const bool sparse = sparse_occ && (dy.C == 32 || dy.C == 64) && !dy.scale && dy.slope == 1.0f;
    if (!sparse) {
        WgradPlan q = plan_wgrad(N, G, G, G, dy.C, 4, 5, 1, true);
        q.p.in = in; q.p.dy = dy; q.p.pad = 2;
        return run_wgrad(q, ws, dW, 4, 125, s);
    }
    const size_t G3 = (size_t)G * G * G;
    const int C = dy.C, chunks = k5s_chunks(G);
    float* dysum = ws; float* zocc = dysum + G3 * C; float* dense_ws = zocc + ((G3 + 63) & ~(size_t)63);
    WgradPlan q = plan_wgrad(1, G, G, G, C, 4, 5, 1, true);
    float* part = dense_ws + ((q.ws_floats + 63) & ~(size_t)63);
    // coordinate channels: one frame holding the sum of dy over the frames, empty occupancy.  s_coord (with its two events): that part
    // on a second stream beside the occupancy channel's kernel below - the two are the serial tail of the backward pass (0.41 + 0.68 ms
    // at 64^3 x 64 frames with nothing else left to run); the occupancy channel's final reduce waits for both.
    const bool two = s_coord && ev_fork && ev_join && s_coord != s && sparse_occ == 1 && G % 8 == 0;
    hipStream_t sc = two ? s_coord : s;
    int rc;
    if (two) {
        if ((rc = nm_check_hip(hipEventRecord(ev_fork, s), "wgrad_k5occ: fork event"))) return rc;
        if ((rc = nm_check_hip(hipStreamWaitEvent(sc, ev_fork, 0), "wgrad_k5occ: fork wait"))) return rc;
    }
    hipLaunchKernelGGL(sum_frames4_kernel, dim3(grid_for(G3 * C / 4)), dim3(256), 0, sc, dy.p, N, G3 * C / 4, dysum, dy.h);
    rc = nm_check_hip(hipMemsetAsync(zocc, 0, G3 * sizeof(float), sc), "wgrad_k5occ: memset");
    if (rc) return rc;
    TensorRef in1 = in; in1.p = zocc; in1.N = 1;
    TensorRef dy1 = dy; dy1.p = dysum; dy1.N = 1; dy1.h = 0;         // (the frame sum is fp32)
    q.p.in = in1; q.p.dy = dy1; q.p.pad = 2;
    if ((rc = run_wgrad(q, dense_ws, dW, 4, 125, sc))) return rc;
    if (two && (rc = nm_check_hip(hipEventRecord(ev_join, sc), "wgrad_k5occ: join event"))) return rc;
    // occupancy channel: matrix cores over the non-empty bricks (sparse_occ 1, grids that are whole 4x8x8 bricks), else the gather
    if (sparse_occ == 1 && G % 8 == 0) {
        static NmDeviceOnce attr_set;
        if (!attr_set.done()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k5occ_mfma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k5occ_mfma_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(wgrad_k5occ_mfma)");
            attr_set.mark();
        }
        // eight workgroups per CU when the clip allows: a workgroup's bricks are a serial chain (halo test, dY brick through LDS, 128
        // MFMA steps, no double buffering) - 512 / 1024 / 2048 workgroups: 915 / 794 / 637 us (+ 24 / 46 / 92 us of reduce)
        const int blocks = min(2048, N * chunks);                  // (the partial buffer is sized for N * chunks blocks)
        const size_t ldsb = (size_t)(8 * 12 * 12 + 8 + 256 * C) * sizeof(float);
        if (dy.h) {
            if (C == 32) hipLaunchKernelGGL((wgrad_k5occ_mfma_kernel<32, true>), dim3(blocks), dim3(256), ldsb, s, occ, dy.p, N, G, part);
            else hipLaunchKernelGGL((wgrad_k5occ_mfma_kernel<64, true>), dim3(blocks), dim3(256), ldsb, s, occ, dy.p, N, G, part);
        } else
        if (C == 32) hipLaunchKernelGGL((wgrad_k5occ_mfma_kernel<32>), dim3(blocks), dim3(256), ldsb, s, occ, dy.p, N, G, part);
        else hipLaunchKernelGGL((wgrad_k5occ_mfma_kernel<64>), dim3(blocks), dim3(256), ldsb, s, occ, dy.p, N, G, part);
        if (two && (rc = nm_check_hip(hipStreamWaitEvent(s, ev_join, 0), "wgrad_k5occ: join wait"))) return rc;
        hipLaunchKernelGGL(wgrad_k5occ_sparse_reduce_kernel, dim3((125 * C + 255) / 256), dim3(256), 0, s, part, blocks, C, dW);
        return nm_check_hip(hipGetLastError(), "wgrad_k5occ mfma launch");
    }
    if (dy.h) { nm_set_error("wgrad_k5occ: the gather form has no bfloat16 instantiation (grids that are whole 4x8x8 bricks take the MFMA form)"); return NM_ERR_UNSUPPORTED; }
    if (C == 32) hipLaunchKernelGGL((wgrad_k5occ_sparse_kernel<32>), dim3(N * chunks), dim3(256), 0, s, occ, dy.p, G, chunks, part);
    else hipLaunchKernelGGL((wgrad_k5occ_sparse_kernel<64>), dim3(N * chunks), dim3(256), 0, s, occ, dy.p, G, chunks, part);
    hipLaunchKernelGGL(wgrad_k5occ_sparse_reduce_kernel, dim3((125 * C + 255) / 256), dim3(256), 0, s, part, N * chunks, C, dW);
    return nm_check_hip(hipGetLastError(), "wgrad_k5occ sparse launch");
}

int nm_gnb_blocks_per_frame(int voxels) { const int vb = nm_gnb_vb(voxels); return (voxels + vb - 1) / vb; }

int nm_launch_gnb_partials(const float* dA, const TensorRef& y, float* part, hipStream_t s, const float* dA_mul, const float* dv, const float* wv) {
    if (y.C > 256 || y.C <= 0) { nm_set_error("gnb_partials: C=%d unsupported", y.C); return NM_ERR_ARG; }
    const int voxels = y.D * y.H * y.W;
    if (dv && !(y.C % 4 == 0 && 1024 % y.C == 0)) { nm_set_error("gnb_partials: outer-product gradient needs C %% 4 == 0 and 1024 %% C == 0"); return NM_ERR_ARG; }
    if (y.C % 4 == 0 && 1024 % y.C == 0) {
        if (y.h) hipLaunchKernelGGL(gnb_partials4_kernel<true>, dim3(nm_gnb_blocks_per_frame(voxels), y.N), dim3(256), 0, s, dA, y, voxels, part, dA_mul, dv, wv);
        else hipLaunchKernelGGL(gnb_partials4_kernel<false>, dim3(nm_gnb_blocks_per_frame(voxels), y.N), dim3(256), 0, s, dA, y, voxels, part, dA_mul, dv, wv);
    } else
        hipLaunchKernelGGL(gnb_partials_kernel, dim3(nm_gnb_blocks_per_frame(voxels), y.N), dim3(256), 0, s, dA, y, voxels, part, dA_mul);
    return nm_check_hip(hipGetLastError(), "gnb_partials launch");
}

int nm_launch_gnb_finalize(const float* bpart, int nblk_b, const float* fpart, int nblk_f, int N, int C, int groups, int voxels,
                           const float* gamma, float eps, float* coef, float* dgn, hipStream_t s, const double* chsum) {
    if (groups <= 0 || C % groups || C / groups > 64) { nm_set_error("gnb_finalize: bad groups %d for C=%d", groups, C); return NM_ERR_ARG; }
    if (chsum) hipLaunchKernelGGL(gnb_finalize_kernel<256>, dim3(N * groups), dim3(256), 0, s, bpart, nblk_b, fpart, nblk_f, C, groups, voxels, gamma,
                                  eps, coef, dgn, chsum);
    else hipLaunchKernelGGL(gnb_finalize_kernel<1024>, dim3(N * groups), dim3(1024), 0, s, bpart, nblk_b, fpart, nblk_f, C, groups, voxels, gamma,
                            eps, coef, dgn, chsum);
    return nm_check_hip(hipGetLastError(), "gnb_finalize launch");
}

int nm_launch_sum_frames(const float* src, int N, int C, int stride, int off, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_frames_kernel, dim3((C + 63) / 64), dim3(64), 0, s, src, N, C, stride, off, out);
    return nm_check_hip(hipGetLastError(), "sum_frames launch");
}

int nm_launch_sum_frames3(const float* dgn, int N, int C, float* dgamma, float* dbeta, float* dbias, hipStream_t s) {
    hipLaunchKernelGGL(sum_frames3_kernel, dim3((C + 63) / 64), dim3(64), 0, s, dgn, N, C, dgamma, dbeta, dbias);
    return nm_check_hip(hipGetLastError(), "sum_frames3 launch");
}

int nm_launch_sum_frames3_multi(const NmSum3Job* jobs, int njobs, hipStream_t s) {
    for (int j0 = 0; j0 < njobs; j0 += NM_SUM3_JOBS) {
        NmSum3Jobs a;
        const int n = njobs - j0 < NM_SUM3_JOBS ? njobs - j0 : NM_SUM3_JOBS;
        int cmax = 0;
        for (int i = 0; i < n; ++i) { a.j[i] = jobs[j0 + i]; cmax = a.j[i].C > cmax ? a.j[i].C : cmax; }
        hipLaunchKernelGGL(sum_frames3_multi_kernel, dim3((cmax + 63) / 64, n), dim3(64), 0, s, a);
    }
    return nm_check_hip(hipGetLastError(), "sum_frames3_multi launch");
}

int nm_launch_sum_partials(const float* part, int rows, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(C), dim3(256), 0, s, part, rows, C, out);
    return nm_check_hip(hipGetLastError(), "sum_partials launch");
}

int nm_launch_gnb_apply(const float* dA, const TensorRef& y, const float* coef, float* dy, hipStream_t s, unsigned* amax, const float* dA_mul,
                        const float* dv, const float* wv) {
    if (y.C % 4) { nm_set_error("gnb_apply: C %% 4 != 0"); return NM_ERR_ARG; }
    const size_t frame4 = (size_t)y.D * y.H * y.W * y.C / 4;
    if (frame4 * 4 >= ((size_t)1 << 31)) { nm_set_error("gnb_apply: frame too large"); return NM_ERR_ARG; }
    const unsigned bx = (unsigned)min((frame4 + 255) / 256, (size_t)max(1, 4096 / max(y.N, 1)));
    if (nm_ls().gnb_apply4 && 1024 % y.C == 0) {
        const dim3 g(bx, y.N);
#define NM_GNB_APPLY4(R, Cf) do { if (y.h && nm_ls().gnb_u8) hipLaunchKernelGGL((gnb_apply4_kernel<R, Cf, true, 8u>), g, dim3(256), 0, s, dA, y, coef, dy, amax, dA_mul, dv, wv); \
                                 else if (y.h) hipLaunchKernelGGL((gnb_apply4_kernel<R, Cf, true>), g, dim3(256), 0, s, dA, y, coef, dy, amax, dA_mul, dv, wv); \
                                 else hipLaunchKernelGGL((gnb_apply4_kernel<R, Cf, false>), g, dim3(256), 0, s, dA, y, coef, dy, amax, dA_mul, dv, wv); } while (0)
        if (dv) { if (coef) NM_GNB_APPLY4(true, true); else NM_GNB_APPLY4(true, false); }
        else if (coef) NM_GNB_APPLY4(false, true);
        else NM_GNB_APPLY4(false, false);
#undef NM_GNB_APPLY4
        return nm_check_hip(hipGetLastError(), "gnb_apply4 launch");
    }
    hipLaunchKernelGGL(gnb_apply_kernel, dim3(bx, y.N), dim3(256), 0, s, dA, y, coef, dy, amax, dA_mul, dv, wv);
    return nm_check_hip(hipGetLastError(), "gnb_apply launch");
}

int nm_launch_absmax(const float* x, size_t n, unsigned* amax, hipStream_t s, const float* mul, int h) {
    if (n % 4) { nm_set_error("absmax: n %% 4 != 0"); return NM_ERR_ARG; }
    hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, x, n / 4, amax, mul, h);
    return nm_check_hip(hipGetLastError(), "absmax launch");
}

int nm_launch_make_scale(const unsigned* amax, int count, float* scale, float* sc2, hipStream_t s, float* zero_shift) {
    hipLaunchKernelGGL(make_scale_kernel, dim3((count + 255) / 256), dim3(256), 0, s, amax, count, scale, sc2, zero_shift);
    return nm_check_hip(hipGetLastError(), "make_scale launch");
}

int nm_launch_scale_by(float* x, size_t n, const float* mul, hipStream_t s, int h) {
    if (n % 4) { nm_set_error("scale_by: n %% 4 != 0"); return NM_ERR_ARG; }
    hipLaunchKernelGGL(scale_by_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, x, n / 4, mul, h);
    return nm_check_hip(hipGetLastError(), "scale_by launch");
}

int nm_launch_upsample2_adjoint(const float* dfine, int N, int D, int H, int W, int C, float* dcoarse, hipStream_t s, const float* mul, int hf, int hc) {
    if (C % 4) { nm_set_error("upsample2_adjoint: C %% 4 != 0"); return NM_ERR_ARG; }
    const size_t total = (size_t)N * D * H * W * (C / 4);
    if (nm_ls().adj_zwalk && C % (hf ? 64 : 32) == 0 && H % UZ_BY == 0 && W % UZ_BX == 0 && D >= 2 && (size_t)4 * H * W * C < ((size_t)1 << 31) && total >= 16384) {
        const unsigned cols = (unsigned)(N * (H / UZ_BY) * (W / UZ_BX)), chunks = (unsigned)(C / (hf ? 64 : 32));
        unsigned zs = 1;
        while (zs < 4 && cols * chunks * zs < 2048 && D % (2 * zs) == 0 && D / (2 * zs) >= 4) zs *= 2;
        const dim3 g(cols, chunks, zs);
        if (hf && hc) hipLaunchKernelGGL((upsample2_adjoint_zwalk_kernel<true, true>), g, dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul);
        else if (hf) hipLaunchKernelGGL((upsample2_adjoint_zwalk_kernel<true, false>), g, dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul);
        else if (hc) hipLaunchKernelGGL((upsample2_adjoint_zwalk_kernel<false, true>), g, dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul);
        else hipLaunchKernelGGL((upsample2_adjoint_zwalk_kernel<false, false>), g, dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul);
        return nm_check_hip(hipGetLastError(), "upsample2_adjoint (z walk) launch");
    }
    if (C % (hf ? 32 : 16) == 0 && D % UA_BZ == 0 && H % UA_BY == 0 && W % UA_BX == 0 && total >= 16384) {
        const size_t blocks = (size_t)N * (D / UA_BZ) * (H / UA_BY) * (W / UA_BX);
        if (hf && hc) hipLaunchKernelGGL((upsample2_adjoint_tile_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul);
        else if (hf) hipLaunchKernelGGL((upsample2_adjoint_tile_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul);
        else if (hc) hipLaunchKernelGGL((upsample2_adjoint_tile_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul);
        else hipLaunchKernelGGL((upsample2_adjoint_tile_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul);
        return nm_check_hip(hipGetLastError(), "upsample2_adjoint launch");
    }
    hipLaunchKernelGGL(upsample2_adjoint_kernel, dim3(grid_for(total)), dim3(256), 0, s, dfine, N, D, H, W, C, dcoarse, mul, hf, hc);
    return nm_check_hip(hipGetLastError(), "upsample2_adjoint launch");
}

int nm_launch_flip_weight(const float* w, int Cout, int Cin, int csel, int ks, float* out, hipStream_t s) {
    const int taps = ks * ks * ks, total = csel * Cout * taps;
    hipLaunchKernelGGL(flip_weight_kernel, dim3(min((total + 255) / 256, 4096)), dim3(256), 0, s, w, Cout, Cin, csel, taps, out);
    return nm_check_hip(hipGetLastError(), "flip_weight launch");
}

int nm_launch_axpby(float* dst, const float* dst_mul, const float* src, const float* src_mul, size_t n, hipStream_t s, int h) {
    if (n % 4) { nm_set_error("axpby: n %% 4 != 0"); return NM_ERR_ARG; }
    hipLaunchKernelGGL(axpby4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, dst, dst_mul, src, src_mul, n / 4, h);
    return nm_check_hip(hipGetLastError(), "axpby launch");
}

int nm_launch_axpy(float* dst, const float* src, size_t n, hipStream_t s, int h) {
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, s, dst, src, n, h);
    return nm_check_hip(hipGetLastError(), "axpy launch");
}
