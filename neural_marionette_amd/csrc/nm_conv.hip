// 3-D convolution as an implicit GEMM on the fp32 matrix cores of gfx950.
//
// Replaces, for the hot path, every nn.Conv3d the reference reaches through torch
// (modules/vox_modules.py:12,26,30,39,53; model/kypt_detector.py:279,295,381,429-457):
//   k5 s1 p2, k3 s1 p1, k2 s2 p0 ("pool"), k1.
//
//   Out[m, co] = sum_{tap, ci} In[vox(m) * stride + tap - pad, ci] * W[tap, ci, co]
//
// * M = output voxels of one frame, tiled as a power-of-two 3-D brick (BM = 128 * MT rows);
//   N = output channels (32 * NT per block); K = taps x Cin, walked in LDS chunks of KC
//   channels.
// * MFMA: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD).  Lane l feeds
//   A[row l&31][k = l>>5] and B[k = l>>5][col l&31]; the K order is permuted so that the
//   lane half h owns four consecutive channels 4*(2*kq+h)+{0..3}: one ds_read_b128 (A) and
//   one global 16-B load (B) feed four back-to-back MFMAs.
// * A: the input brick + halo is staged once per channel chunk into LDS as
//   [channel quad][halo voxel] float4, with the producer's GroupNorm affine + LeakyReLU
//   applied on the way in (zero padding is applied after the transform).
// * B: packed weights [tap][Cin/4][Co_pad][4] are read straight from L2 (every block reads
//   the same few hundred KB), prefetched one tap ahead in registers.
// * Epilogue: bias, coalesced channels-last store (128-B runs), and per-(block, channel)
//   sum / sum-of-squares partials for the following GroupNorm (deterministic: no float
//   atomics).
#include "nm_common.h"
#include "nm_up2c.h"
#include <type_traits>
#include <utility>
#include <vector>

namespace {

#define NM_SPLIT_SCALE 2048.0f   // lo parts of the split-fp16 operands are stored times 2^11

struct ConvParams {
    const float* in; const float* in_scale; const float* in_shift; float in_slope;
    int N, ID, IH, IW, Cin;
    const float* w; const float* bias;
    float* out; float* part;
    int OD, OH, OW, Cout, Co_pad;
    int ks, stride, pad;
    int bz_l2, by_l2, bx_l2;      // brick dims (log2)
    int nbz, nby, nbx;            // bricks per frame
    int KC;                       // channels per LDS chunk (8 or 16)
    int cin_real;                 // un-padded input channels (profiling only)
    int up2;                      // input is stored at half resolution: stage its trilinear x2 upsampling
    int HZ, HY, HX, HV, HVp;      // halo dims, voxels, padded plane stride (HVp % 8 == 2)
    int CVp;                      // up2: plane stride of the coarse LDS tile
    int ZP;                       // f16s: pitch between halo z-planes in LDS (>= HY*HX, = 4 mod 16)
    const void* w16;              // conv_mfma_kernel<1,NT>: split-fp16 weights (null: fp32 MFMA core)
    int ksplit;                   // conv_mfma_kernel<1,NT> + w16: the taps are dealt to 2 / 4 waves that share a row tile (tiny volumes)
    int st_z, st_y, st_x;         // f16s / f16p: XCD super-tile in bricks (0: bricks in linear order), see super_tile_item()
    unsigned st_sx, st_sy, st_pf; // super-tiles per row / column / frame, and ceil(2^32 / d) of each (0: d = 1): the three divisions of a brick
    unsigned st_m_sx, st_m_sy, st_m_pf;   // decode as multiplications (round 6); st_pf = 0: the host could not prove them exact, plain divisions
    const float* in_alt; const unsigned char* in_map;   // brick-sparse input (TensorRef::alt / brickmap), conv_pool_f16s only
    const float* in2; const float* in2_scale; const float* in2_shift; float in2_slope;   // un-materialised residual sum (TensorRef::p2 ...), conv_pool_f16s only
    int wdma;                     // conv_f16p2: the producers copy the weights global -> LDS by LDS-DMA instead of through registers (A/B switch)
    int in_h, out_h;              // 16-bit storage (conv mode 4): the input / output tensor is bfloat16 (kernels take it as the IO template bits 1 / 2)
#ifdef NM_DIAG
    unsigned long long* stamps;   // diagnostic build only: per-block phase timestamps
#endif
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for every global store of
// the previous brick's epilogue to be acknowledged (3-4k cycles per brick in the persistent conv_f16s loop).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }


template <bool IH = false>
__device__ __forceinline__ f32x4 load_raw(const ConvParams& p, int n, int z, int y, int x, int c) {
    return nm_ld4<IH>(p.in, ((((size_t)n * p.ID + z) * p.IH + y) * p.IW + x) * p.Cin + c);
}

// pending GroupNorm of the producer: x * scale + shift, LeakyReLU.  Written per component with scalar fma / mul / max:
// packed fp32 VALU (v_pk_*_f32) beside another wave's MFMAs costs several times its issue slot on gfx950, and the
// staging phases of these kernels always run beside the co-resident workgroup's MFMA phase.
__device__ __forceinline__ f32x4 apply_act(const ConvParams& p, f32x4 v, const f32x4& sc, const f32x4& sh) {
    if (p.in_scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(v[j], sc[j], sh[j]);
    }
    if (p.in_slope != 1.0f) {            // LeakyReLU with slope in [0,1): max(v, slope * v)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], v[j] * p.in_slope);
    }
    return v;
}

// value of an un-materialised residual sum, bit for bit what apply2_kernel (nm_elem.hip load_t) would have stored: each addend
// x * scale + shift as a multiply and an add (not fused), LeakyReLU as v > 0 ? v : v * slope, then the sum
__device__ __forceinline__ f32x4 sum_act2(const ConvParams& p, f32x4 a, const f32x4& sca, const f32x4& sha, f32x4 b, const f32x4& scb, const f32x4& shb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float u = a[j], v = b[j];
        if (p.in_scale) { u = u * sca[j]; u = u + sha[j]; }
        if (p.in_slope != 1.0f) u = u > 0.f ? u : u * p.in_slope;
        if (p.in2_scale) { v = v * scb[j]; v = v + shb[j]; }
        if (p.in2_slope != 1.0f) v = v > 0.f ? v : v * p.in2_slope;
        a[j] = u + v;
    }
    return a;
}

// activated input sample: x * scale + shift, LeakyReLU (the producer's pending GroupNorm)
template <bool IH = false>
__device__ __forceinline__ f32x4 load_act(const ConvParams& p, int n, int z, int y, int x, int c) {
    f32x4 v = nm_ld4<IH>(p.in, ((((size_t)n * p.ID + z) * p.IH + y) * p.IW + x) * p.Cin + c);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
    if (p.in_scale) {
        sc = *reinterpret_cast<const f32x4*>(p.in_scale + (size_t)n * p.Cin + c);
        sh = *reinterpret_cast<const f32x4*>(p.in_shift + (size_t)n * p.Cin + c);
    }
    return apply_act(p, v, sc, sh);
}

__device__ __forceinline__ int up_lo(int o) {          // first source index of output index o (o >= 0)
    float src = 0.5f * ((float)o + 0.5f) - 0.5f;
    return src < 0.f ? 0 : (int)src;
}

// source index / weight of output index o for scale-2 linear interpolation, align_corners=False
__device__ __forceinline__ void up_idx(int o, int I, int& i0, int& i1, float& l1) {
    float src = 0.5f * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    i1 = i0 + (i0 < I - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// v = hi + lo * 2^-11 with hi = rne_f16(v), lo = rne_f16((v - hi) * 2^11).  v - hi is exact in fp32, so the lo part is
// one v_fma_mix{lo,hi}_f16 of (hi, -2^11, v * 2^11): 2.5 VALU per value, none of them packed fp32 math.
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v0 = (j < 2) ? a[2 * j] : b[2 * j - 4], v1 = (j < 2) ? a[2 * j + 1] : b[2 * j - 3];
        half2v hh = __builtin_convertvector(f32x2{v0, v1}, half2v);        // v_cvt_pk_f16_f32
        asm volatile("" : "+v"(hh));                                         // keep the packed pair (no per-half re-conversion)
        const float t0 = v0 * NM_SPLIT_SCALE, t1 = v1 * NM_SPLIT_SCALE;
        hi[2 * j] = hh[0]; hi[2 * j + 1] = hh[1];
        lo[2 * j] = (_Float16)__builtin_fmaf((float)hh[0], -NM_SPLIT_SCALE, t0);
        lo[2 * j + 1] = (_Float16)__builtin_fmaf((float)hh[1], -NM_SPLIT_SCALE, t1);
    }
}

// The same walk on the fp16 matrix cores for the small-volume layers (conv_mfma_kernel<1,NT> with Cin % 16 == 0): with 64 or
// fewer voxels per frame a launch is one short wave of workgroups, each a serial chain of K/2 fp32 MFMAs of 64 cycles; the
// split product (3 MFMAs of 32 cycles per 16 channels) cuts the chain ~4x.  The fp32 tile in LDS is split per tap (2 LDS
// reads + 20 VALU per 3 NT MFMAs: the chain, not the throughput, is what these launches wait for).
template <int NT, bool SINGLE = false>
__device__ __forceinline__ void mfma_chunk16(const ConvParams& p, const f32x4* lds, const half8* __restrict__ wq, size_t tap_stride,
                                             int taps, int h, int arow, f32x16 (&acc)[1][NT], f32x16 (&accl)[NT], int tap0 = 0, int tstep = 1) {
    const size_t plane = (size_t)p.Co_pad;
    const int k2 = p.ks * p.ks;
    // the weights of a tap are requested while the previous tap is split and multiplied (loaded at their tap they were one exposed L2
    // round trip per tap: 27 per 16-channel chunk, most of the time of these small-volume launches)
    half8 bh[NT], bl[NT];
    if (tap0 < taps) {
        const half8* wt = wq + (size_t)tap0 * tap_stride;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { bh[nt] = wt[nt * 32]; bl[nt] = wt[2 * plane + nt * 32]; }
    }
    for (int tap = tap0; tap < taps; tap += tstep) {
        const int tz = tap / k2, ty = (tap / p.ks) % p.ks, tx = tap % p.ks;
        const half8* wn = wq + (size_t)min(tap + tstep, taps - 1) * tap_stride;        // (behind the last tap: any valid address)
        half8 nh[NT], nl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { nh[nt] = wn[nt * 32]; nl[nt] = wn[2 * plane + nt * 32]; }
        const int tapoff = (tz * p.HY + ty) * p.HX + tx;
        const f32x4 a = lds[(2 * h) * p.HVp + arow + tapoff], b = lds[(2 * h + 1) * p.HVp + arow + tapoff];
        half8 hi, lo;
        split8(a, b, hi, lo);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bh[nt], acc[0][nt], 0, 0, 0);
            accl[nt] = nm_mfma_lo<SINGLE>(hi, bl[nt], accl[nt]);
            accl[nt] = nm_mfma_lo<SINGLE>(lo, bh[nt], accl[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { bh[nt] = nh[nt]; bl[nt] = nl[nt]; }
    }
}

// One LDS-resident channel chunk (NKQ octets of channels): walk the taps, B operand
// prefetched one tap ahead, 4*NKQ*MT*NT MFMAs per tap.
template <int MT, int NT, int NKQ>
__device__ __forceinline__ void mfma_chunk(const ConvParams& p, const f32x4* lds, const f32x4* __restrict__ wq,
                                           size_t tap_stride, int taps, int h, const int (&arow)[MT],
                                           f32x16 (&acc)[MT][NT]) {
    f32x4 bcur[NKQ][NT];
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[kq][nt] = wq[(size_t)(2 * kq) * p.Co_pad + nt * 32];
    int tx = 0, ty = 0, tz = 0;
    for (int tap = 0; tap < taps; ++tap) {
        const f32x4* wn = wq + (size_t)min(tap + 1, taps - 1) * tap_stride;
        f32x4 bnxt[NKQ][NT];
#pragma unroll
        for (int kq = 0; kq < NKQ; ++kq)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bnxt[kq][nt] = wn[(size_t)(2 * kq) * p.Co_pad + nt * 32];
        const int tapoff = (tz * p.HY + ty) * p.HX + tx;
        f32x4 a[NKQ][MT];
#pragma unroll
        for (int kq = 0; kq < NKQ; ++kq)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[kq][mt] = lds[(2 * kq + h) * p.HVp + arow[mt] + tapoff];
#pragma unroll
        for (int kq = 0; kq < NKQ; ++kq)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kq][mt][s], bcur[kq][nt][s], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int kq = 0; kq < NKQ; ++kq)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bcur[kq][nt] = bnxt[kq][nt];
        if (++tx == p.ks) { tx = 0; if (++ty == p.ks) { ty = 0; ++tz; } }
    }
}

// bias (or a per-voxel constant field), coalesced channels-last store, GroupNorm partials
struct EpiArgs {
    float* out; float* part; const float* bias; const float* field;
    int OD, OH, OW, Cout, bz_l2, by_l2, bx_l2;
    int xz_tiles;      // 1: each 32-row MFMA tile is an 8(x) x 4(z) slab at one y (conflict-free ds_read_b128, see conv_f16s)
};

template <int MT, int NT>
__device__ __forceinline__ void epilogue(const EpiArgs& p, float* red /*[4][NT*32][2] LDS*/, f32x16 (&acc)[MT][NT], int n, int br,
                                         int nblk, int oz0, int oy0, int ox0, int co_base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int BXm = (1 << p.bx_l2) - 1, BYm = (1 << p.by_l2) - 1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co_base + nt * 32 + l31;
        const bool cv = co < p.Cout;
        const float bv = (cv && p.bias) ? p.bias[co] : 0.f;
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                int m = (wave * MT + mt) * 32 + row;
                int x = m & BXm, y = (m >> p.bx_l2) & BYm, z = m >> (p.bx_l2 + p.by_l2);
                if (p.xz_tiles) { const int c = row >> 2; x = (((0x96 >> c) & 1) << 2) + (row & 3); z = c >> 1; y = wave * MT + mt; }
                int oz = oz0 + z, oy = oy0 + y, ox = ox0 + x;
                if (cv && oz < p.OD && oy < p.OH && ox < p.OW) {
                    const size_t vo = (((size_t)oz * p.OH + oy) * p.OW + ox) * p.Cout + co;
                    float v = acc[mt][nt][r] + (p.field ? p.field[vo] : bv);
                    p.out[(size_t)n * p.OD * p.OH * p.OW * p.Cout + vo] = v;
                    s += v; ss += v * v;
                }
            }
        }
        if (p.part) {
            s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
            if (h == 0) { red[(wave * NT * 32 + nt * 32 + l31) * 2] = s; red[(wave * NT * 32 + nt * 32 + l31) * 2 + 1] = ss; }
        }
    }
    if (p.part) {
        __syncthreads();
        if (tid < NT * 32) {
            int co = co_base + tid;
            if (co < p.Cout) {
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) { s += red[(wv * NT * 32 + tid) * 2]; ss += red[(wv * NT * 32 + tid) * 2 + 1]; }
                float* dst = p.part + (((size_t)n * nblk + br) * p.Cout + co) * 2;
                dst[0] = s; dst[1] = ss;
            }
        }
    }
}

// Epilogue of the 8(x) x 4(z) tile mapping.  Register r of a 32x32 accumulator is row (r&3) + 8(r>>2) + 4h, i.e.
// z = r>>2, x = 4 * (h ^ G[r>>2]) + (r&3) with G = {0,1,1,0}; y is the tile index.  Interior bricks with all 32
// channels of every N tile valid take the branch-free path: one base pointer per tile, constant row offsets.
// OH16: bfloat16 output.  A lane holds ONE channel of 16 voxels; the two lanes of neighbouring channels exchange values (one DPP move
// per register pair) so that each stores one packed dword per two registers: even lanes the channel pair of row r, odd lanes of row r + 1.
template <int MT, int NT, bool OH16 = false>
__device__ __forceinline__ void epilogue_xz(const EpiArgs& p, float* red, f32x16 (&acc)[MT][NT], f32x16 (&accl)[MT][NT], int n,
                                            int br, int nblk, int oz0, int oy0, int ox0, int co_base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const bool interior = (oz0 + 4 <= p.OD) && (oy0 + 8 <= p.OH) && (ox0 + 8 <= p.OW) && (co_base + NT * 32 <= p.Cout);
    const size_t sX = (size_t)p.Cout, sZ = (size_t)p.OH * p.OW * p.Cout;
    const size_t xo0 = (size_t)(4 * h) * sX, xo1 = (size_t)(4 * (h ^ 1)) * sX;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co_base + nt * 32 + l31;
        const bool cv = co < p.Cout;
        const float bv = (cv && p.bias) ? p.bias[co] : 0.f;
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int y = wave * MT + mt;
            float* base = nm_eptr(p.out, ((((size_t)n * p.OD + oz0) * p.OH + oy0 + y) * p.OW + ox0) * sX + co, OH16);
            if constexpr (OH16) {
                const bool odd = (l31 & 1) != 0;
                unsigned short* bq = reinterpret_cast<unsigned short*>(base) - (odd ? 1 : 0);        // channel pair (co & ~1, co | 1)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int g = r >> 2;
                    const int x0 = 4 * (h ^ ((g == 1 || g == 2) ? 1 : 0)) + (r & 3);              // voxel x of register r (r + 1: x0 + 1)
                    const bool ok0 = interior || (cv && oz0 + g < p.OD && oy0 + y < p.OH && ox0 + x0 < p.OW);
                    const bool ok1 = interior || (cv && oz0 + g < p.OD && oy0 + y < p.OH && ox0 + x0 + 1 < p.OW);
                    const float v0 = (acc[mt][nt][r] + accl[mt][nt][r] * (1.0f / NM_SPLIT_SCALE)) + bv;
                    const float v1 = (acc[mt][nt][r + 1] + accl[mt][nt][r + 1] * (1.0f / NM_SPLIT_SCALE)) + bv;
                    if (ok0) { s += v0; ss += v0 * v0; }
                    if (ok1) { s += v1; ss += v1 * v1; }
                    const float send = odd ? v0 : v1;
                    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                    const unsigned pk = odd ? nm_pk_bf16(recv, v1) : nm_pk_bf16(v0, recv);
                    if (odd ? ok1 : ok0) *reinterpret_cast<unsigned*>(bq + (size_t)g * sZ + (size_t)(x0 + (odd ? 1 : 0)) * sX) = pk;
                }
            } else if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = r >> 2;
                    const size_t off = (size_t)g * sZ + ((g == 1 || g == 2) ? xo1 : xo0) + (size_t)(r & 3) * sX;
                    const float v = (acc[mt][nt][r] + accl[mt][nt][r] * (1.0f / NM_SPLIT_SCALE)) + bv;
                    base[off] = v;
                    s += v; ss += v * v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = r >> 2;
                    const int x = 4 * (h ^ ((g == 1 || g == 2) ? 1 : 0)) + (r & 3);
                    if (cv && oz0 + g < p.OD && oy0 + y < p.OH && ox0 + x < p.OW) {
                        const float v = (acc[mt][nt][r] + accl[mt][nt][r] * (1.0f / NM_SPLIT_SCALE)) + bv;
                        base[(size_t)g * sZ + (size_t)x * sX] = v;
                        s += v; ss += v * v;
                    }
                }
            }
        }
        if (p.part) {
            s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
            if (h == 0) { red[(wave * NT * 32 + nt * 32 + l31) * 2] = s; red[(wave * NT * 32 + nt * 32 + l31) * 2 + 1] = ss; }
        }
    }
    if (p.part) {
        lds_barrier();
        if (tid < NT * 32) {
            int co = co_base + tid;
            if (co < p.Cout) {
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) { s += red[(wv * NT * 32 + tid) * 2]; ss += red[(wv * NT * 32 + tid) * 2 + 1]; }
                float* dst = p.part + (((size_t)n * nblk + br) * p.Cout + co) * 2;
                dst[0] = s; dst[1] = ss;
            }
        }
    }
}

// ---- first layer, occupancy channel only ------------------------------------------------------------------------
// conv5(cat[occ, x1, x2, x3]) = conv5_ch0(occ) + field, field = conv5(cat[0, x1, x2, x3]) + bias depends on the
// weights only (the coordinate ramps and their zero padding are the same for every frame:
// kypt_detector_utils.py:19-26), so per frame only the occupancy channel is convolved: an implicit GEMM
// whose K dimension is the 125 taps (padded to 128), M a 4x8x8 brick, operands from a 8x12x12 fp32 halo tile.
struct OccParams {
    const float* occ;      // [N][G][G][G]
    const float* w;        // packed [32 tap-quads][Co_pad][4]
    const float* field;    // [G][G][G][Cout]
    float* out; float* part;
    int N, G, Cout, Co_pad;
    // sparse first layer (inference): a brick whose 8x12x12 occupancy halo is empty equals the constant field - it is not written,
    // brickmap[n][brick] = 0 tells the consumer (the pool conv) to read the field there, and its GroupNorm partial sums are the
    // field's own (field_part [brick][Cout][2], computed once per weight update by this very kernel on an empty frame).  Null: dense.
    unsigned char* brickmap; const float* field_part;
    const unsigned char* flags;    // [N][bricks]: the brick holds an occupied voxel (occ_brick_flags_kernel) - a one-load pre-filter of the halo test
    int row_walk;                  // 1: a workgroup walks one x-row of bricks (grid / (G / 8))
};

__host__ __device__ constexpr int occ_tap_off(int t) { return t < 125 ? ((t / 25) * 12 + (t / 5) % 5) * 12 + t % 5 : 0; }

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_k5occ_kernel(OccParams p) {
    __shared__ float tile[8 * 12 * 12 + 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int nb = p.G >> 3, nbz = p.G >> 2;
    const int nblk = nbz * nb * nb;
    // frame index fastest: the blocks in flight share one brick of the constant field (L2 reuse across frames)
    const int n = blockIdx.x % p.N, br = blockIdx.x / p.N;
    const int oz0 = (br / (nb * nb)) << 2, oy0 = ((br / nb) % nb) << 3, ox0 = (br % nb) << 3;
    const int co_base = blockIdx.y * (NT * 32);
    const float* src = p.occ + (size_t)n * p.G * p.G * p.G;
    for (int i = tid; i < 8 * 12 * 12; i += 256) {
        int hx = i % 12, hy = (i / 12) % 12, hz = i / 144;
        int gz = oz0 - 2 + hz, gy = oy0 - 2 + hy, gx = ox0 - 2 + hx;
        float v = 0.f;
        if ((unsigned)gz < (unsigned)p.G && (unsigned)gy < (unsigned)p.G && (unsigned)gx < (unsigned)p.G)
            v = src[((size_t)gz * p.G + gy) * p.G + gx];
        tile[i] = v;
    }
    int arow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int m = (wave * 2 + mt) * 32 + l31;
        arow[mt] = ((m >> 6) * 12 + ((m >> 3) & 7)) * 12 + (m & 7);
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    __syncthreads();
    const f32x4* __restrict__ wq = reinterpret_cast<const f32x4*>(p.w) + (size_t)h * p.Co_pad + co_base + l31;
    f32x4 bcur[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bcur[nt] = wq[nt * 32];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        f32x4 bnxt[NT];
        const int on = o < 15 ? o + 1 : 15;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bnxt[nt] = wq[(size_t)(2 * on) * p.Co_pad + nt * 32];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int off = h ? occ_tap_off(8 * o + 4 + s) : occ_tap_off(8 * o + s);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float a = tile[arow[mt] + off];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcur[nt][s], acc[mt][nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnxt[nt];
    }
    EpiArgs e;
    e.out = p.out; e.part = p.part; e.bias = nullptr; e.field = p.field;
    e.OD = p.G; e.OH = p.G; e.OW = p.G; e.Cout = p.Cout; e.bz_l2 = 2; e.by_l2 = 3; e.bx_l2 = 3; e.xz_tiles = 0;
    __syncthreads();
    epilogue<2, NT>(e, tile + 8 * 12 * 12, acc, n, br, nblk, oz0, oy0, ox0, co_base);
}

// (Cout, 4, 5,5,5) OIDHW -> the occupancy channel's taps as a (Cout, 125) matrix
__global__ void extract_occ_weight_kernel(const float* __restrict__ w, int Cout, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Cout * 125) out[i] = w[(size_t)(i / 125) * 4 * 125 + i % 125];
}

// x + the values of the other lanes of its 32-lane half: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror, then
// row_bcast15 into rows 1 and 3, whose lanes end up holding the totals of lanes 0-31 / 32-63
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWS, 0xf, true));
}
__device__ __forceinline__ float dpp_sum32(float x) {
    x = dpp_add<0xB1, 0xf>(x); x = dpp_add<0x4E, 0xf>(x); x = dpp_add<0x141, 0xf>(x); x = dpp_add<0x140, 0xf>(x);
    return dpp_add<0x142, 0xa>(x);
}
// dpp_sum32 of N values, stage by stage: a DPP add reads the register the previous VALU instruction wrote two wait states late, and a
// chain per value (5 dependent adds) was issued value after value - in-kernel stamps of conv_f16p2's brick epilogue: 2 764 cycles for
// 32 sums, 17 per add.  Same instructions, same order per value (bit-identical), N independent adds between dependent ones.
template <int N> __device__ __forceinline__ void dpp_sum32_many(float (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = dpp_add<0xB1, 0xf>(x[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = dpp_add<0x4E, 0xf>(x[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = dpp_add<0x141, 0xf>(x[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = dpp_add<0x140, 0xf>(x[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = dpp_add<0x142, 0xa>(x[i]);
    __builtin_amdgcn_sched_barrier(0);
}

// flags[n][brick] = 1 when the 4 x 8 x 8 brick holds an occupied voxel.  One workgroup per (frame, slab of four z-planes): the slab is
// read once, coalesced (the first-layer kernel tests a brick's 8 x 12 x 12 halo with five scattered loads per thread - and 85 % of its
// 65 536 workgroups at the bench shape exist only to find that halo empty; with the flags they find out from 27 bytes).
__global__ __launch_bounds__(256) void occ_brick_flags_kernel(const float* __restrict__ occ, int G, unsigned char* __restrict__ flags) {
    __shared__ int f[256];
    const int nb = G >> 3, n = blockIdx.x / (G >> 2), bz = blockIdx.x % (G >> 2);
    if ((int)threadIdx.x < nb * nb) f[threadIdx.x] = 0;
    __syncthreads();
    const f32x4* src = reinterpret_cast<const f32x4*>(occ + ((size_t)n * G + 4 * bz) * G * G);
    const int q = G >> 2;                                   // float4 per row
    for (int i = threadIdx.x; i < G * G; i += 256) {        // [4][G][G / 4]
        const f32x4 v = src[i];
        if (v[0] != 0.f || v[1] != 0.f || v[2] != 0.f || v[3] != 0.f) f[(((i / q) % G) >> 3) * nb + ((i % q) >> 1)] = 1;
    }
    __syncthreads();
    if ((int)threadIdx.x < nb * nb) flags[((size_t)n * (G >> 2) + bz) * nb * nb + threadIdx.x] = (unsigned char)f[threadIdx.x];
}

// The same layer on the fp16 matrix cores.  Voxel occupancy is 0/1 in the reference's data, exact in fp16, so the product
// needs only occ * w_hi + occ * w_lo / 2^11: two f16 MFMAs at 16x the fp32-MFMA rate (the fp32 kernel above is bound by its
// 128 fp32 MFMAs per wave).  A volume with other values takes a third MFMA with the lo part of the input (decided per
// workgroup while staging), so the result is fp32-equivalent for any input.  K = taps: k-step ks, lane half h, element j
// is tap 16 ks + 8 h + j (taps >= 125 carry zero weights).
typedef _Float16 occ_half8 __attribute__((ext_vector_type(8)));
template <int NT, bool SINGLE = false, bool WALK = false, bool OH = false>      // WALK: one workgroup per x-row of bricks (sparse inference form); OH: bfloat16 output
__global__ __launch_bounds__(256, 2) void conv_k5occ_f16_kernel(OccParams p) {
    __shared__ _Float16 tile_h[8 * 12 * 12], tile_l[8 * 12 * 12];
    __shared__ float red[512];
    __shared__ __attribute__((aligned(16))) float stage[4 * 32 * 36];      // epilogue transposition, one 32-voxel x 32-channel tile per wave
    const int nb = p.G >> 3, nbz = p.G >> 2;
    const int nblk = nbz * nb * nb;
    const int n = blockIdx.x % p.N;                                 // frame index fastest (see conv_k5occ_kernel)
    // with the occupancy flags (sparse inference form) a workgroup walks one x-row of bricks: 8 192 workgroups instead of 65 536 at the
    // bench shape, most of whose bricks cost 27 flag bytes and one partial-sum copy
    const int per_wg = WALK ? nb : 1;                               // (a template parameter: as a run-time loop the dense form lost a wave of occupancy)
    for (int sub = 0; sub < per_wg; ++sub) {
    if (sub) __syncthreads();
    // (the thread index goes through an opaque statement per brick: visible, every lane-derived LDS offset of the body is a loop invariant
    //  that hipcc keeps in a register of its own across the loop - 256 registers and 1 KB of scratch per lane)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int br = (blockIdx.x / p.N) * per_wg + sub;
    const int oz0 = (br / (nb * nb)) << 2, oy0 = ((br / nb) % nb) << 3, ox0 = (br % nb) << 3;
    const int co_base = blockIdx.y * (NT * 32);
    const float* src = p.occ + (size_t)n * p.G * p.G * p.G;
    if (p.flags && p.brickmap) {
        // pre-filter: no occupied voxel in the 27 bricks around -> none in the halo (2 voxels wide) -> the brick is the field
        int fl = 0;
        if (tid < 27) {
            const int bz = (br / (nb * nb)) + tid / 9 - 1, by = ((br / nb) % nb) + (tid / 3) % 3 - 1, bx = (br % nb) + tid % 3 - 1;
            if ((unsigned)bz < (unsigned)nbz && (unsigned)by < (unsigned)nb && (unsigned)bx < (unsigned)nb) fl = p.flags[(size_t)n * nblk + (bz * nb + by) * nb + bx];
        }
        if (!__syncthreads_or(fl)) {
            if (tid == 0 && blockIdx.y == 0) p.brickmap[(size_t)n * nblk + br] = 0;
            if (p.part && tid < NT * 32 && co_base + tid < p.Cout) {
                const float2 v = *reinterpret_cast<const float2*>(p.field_part + ((size_t)br * p.Cout + co_base + tid) * 2);
                *reinterpret_cast<float2*>(p.part + (((size_t)n * nblk + br) * p.Cout + co_base + tid) * 2) = v;
            }
            continue;
        }
    }
    int inexact = 0, occupied = 0;
    {
        // the thread's five halo cells requested together, from clamped addresses, and masked afterwards: under `if (inside)` each
        // load was followed by a full wait - five dependent memory round trips at the head of every workgroup, and 85 % of the 65 536
        // workgroups of a 64-frame launch consist of this staging alone (empty bricks)
        float v[5]; bool in[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = min(tid + 256 * k, 8 * 12 * 12 - 1);
            const int hx = i % 12, hy = (i / 12) % 12, hz = i / 144;
            const int gz = oz0 - 2 + hz, gy = oy0 - 2 + hy, gx = ox0 - 2 + hx;
            in[k] = (unsigned)gz < (unsigned)p.G && (unsigned)gy < (unsigned)p.G && (unsigned)gx < (unsigned)p.G;
            const int cz = min(max(gz, 0), p.G - 1), cy = min(max(gy, 0), p.G - 1), cx = min(max(gx, 0), p.G - 1);
            v[k] = src[((size_t)cz * p.G + cy) * p.G + cx];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = tid + 256 * k;
            const float x = in[k] ? v[k] : 0.f;
            const _Float16 hi = (_Float16)x;
            const _Float16 lo = (_Float16)((x - (float)hi) * NM_SPLIT_SCALE);
            if (i < 8 * 12 * 12) { tile_h[i] = hi; tile_l[i] = lo; inexact |= (lo != (_Float16)0.f); occupied |= (x != 0.f); }
        }
    }
    const bool need_lo = __syncthreads_or(inexact) != 0;            // (also the barrier after staging)
    // a brick whose 8x12x12 neighbourhood holds no occupied voxel (most of a 64^3 grid around one figure) is the field alone: the
    // gather + MFMA loop below would add exact zeros
    const bool any_occ = __syncthreads_or(occupied) != 0;
    if (p.brickmap) {
        if (tid == 0 && blockIdx.y == 0) p.brickmap[(size_t)n * nblk + br] = any_occ ? 1 : 0;
        if (!any_occ) {                 // the brick is the field: nothing to compute, nothing to write but its (precomputed) partial sums
            if (p.part && tid < NT * 32 && co_base + tid < p.Cout) {
                const float2 v = *reinterpret_cast<const float2*>(p.field_part + ((size_t)br * p.Cout + co_base + tid) * 2);
                *reinterpret_cast<float2*>(p.part + (((size_t)n * nblk + br) * p.Cout + co_base + tid) * 2) = v;
            }
            continue;
        }
    }
    int arow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int m = (wave * 2 + mt) * 32 + l31;
        arow[mt] = ((m >> 6) * 12 + ((m >> 3) & 7)) * 12 + (m & 7);
    }
    f32x16 acc[2][NT], accl[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; accl[mt][nt][r] = 0.f; }
    const occ_half8* __restrict__ wq = reinterpret_cast<const occ_half8*>(p.w + (size_t)128 * p.Co_pad) + (size_t)h * p.Co_pad + co_base + l31;
    if (any_occ)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        occ_half8 bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bh[nt] = wq[(size_t)(4 * ks) * p.Co_pad + nt * 32];
            bl[nt] = wq[(size_t)(4 * ks + 2) * p.Co_pad + nt * 32];
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            occ_half8 ah, al;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int off = h ? occ_tap_off(16 * ks + 8 + j) : occ_tap_off(16 * ks + j);
                ah[j] = tile_h[arow[mt] + off];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], ah, acc[mt][nt], 0, 0, 0);
                accl[mt][nt] = nm_mfma_lo<SINGLE>(bl[nt], ah, accl[mt][nt]);
            }
            if (need_lo) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int off = h ? occ_tap_off(16 * ks + 8 + j) : occ_tap_off(16 * ks + j);
                    al[j] = tile_l[arow[mt] + off];
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) accl[mt][nt] = nm_mfma_lo<SINGLE>(bh[nt], al, accl[mt][nt]);
            }
        }
    }
    // Epilogue on transposed accumulators (the MFMAs above take the weights first): a lane holds one voxel and, per N tile,
    // the channels (r & 3) + 8 (r >> 2) + 4 h in its 16 registers - 16-byte field loads and stores.  With one channel per lane
    // this layer spends its time issuing 64 dword loads / stores per lane (the fp32 and the f16 MFMA variant took the same
    // 1.58 ms); GroupNorm partials by DPP row reductions (rows 1 and 3 of the wave end up with the totals).
    const bool chan_ok = (p.Cout & 3) == 0;
    if (chan_ok && co_base + NT * 32 <= p.Cout) {
        // Whole 128-byte lines per store: the accumulator layout gives a lane 16 bytes of one voxel, so a store instruction touches
        // 32 lines, 32 bytes each.  The tile goes through LDS (per wave, no workgroup barrier) and comes back with 8 lanes per
        // voxel: one instruction writes 8 x-consecutive voxels = 1 KB contiguous, and reads the field the same way.  This layer is
        // its 2.1 GB output write: 1.25 ms with the strided stores.
        float* stg = stage + wave * (32 * 36);
        const int chunk = lane & 7;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 t1 = f32x4{0.f, 0.f, 0.f, 0.f}, t2 = t1;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * k4 + e] + accl[mt][nt][4 * k4 + e] * (1.0f / NM_SPLIT_SCALE);
                    *reinterpret_cast<f32x4*>(stg + l31 * 36 + 8 * k4 + 4 * h) = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int vv = it * 8 + (lane >> 3);
                    const int m = (wave * 2 + mt) * 32 + vv;
                    const int oz = oz0 + (m >> 6), oy = oy0 + ((m >> 3) & 7), ox = ox0 + (m & 7);
                    const size_t vo = (((size_t)oz * p.G + oy) * p.G + ox) * p.Cout + co_base + nt * 32 + chunk * 4;
                    f32x4 v = *reinterpret_cast<const f32x4*>(stg + vv * 36 + chunk * 4);
                    v += *reinterpret_cast<const f32x4*>(p.field + vo);
                    nm_st4<OH>(p.out, (size_t)n * p.G * p.G * p.G * p.Cout + vo, v);
                    t1 += v; t2 += v * v;
                }
            }
            if (p.part) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = t1[e], b = t2[e];
                    a += __shfl_xor(a, 8); b += __shfl_xor(b, 8);
                    a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                    a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                    if (lane < 8) { const int c = nt * 32 + chunk * 4 + e; red[(wave * NT * 32 + c) * 2] = a; red[(wave * NT * 32 + c) * 2 + 1] = b; }
                }
            }
        }
    } else
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float s1[16], s2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = (wave * 2 + mt) * 32 + l31;
            const int oz = oz0 + (m >> 6), oy = oy0 + ((m >> 3) & 7), ox = ox0 + (m & 7);
            const size_t vo = (((size_t)oz * p.G + oy) * p.G + ox) * p.Cout;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int co = co_base + nt * 32 + 8 * k4 + 4 * h;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * k4 + e] + accl[mt][nt][4 * k4 + e] * (1.0f / NM_SPLIT_SCALE);
                if (chan_ok && co + 4 <= p.Cout) {
                    v += *reinterpret_cast<const f32x4*>(p.field + vo + co);
                    nm_st4<OH>(p.out, (size_t)n * p.G * p.G * p.G * p.Cout + vo + co, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e < p.Cout) { v[e] += p.field[vo + co + e]; nm_st1<OH>(p.out, (size_t)n * p.G * p.G * p.G * p.Cout + vo + co + e, v[e]); }
                        else v[e] = 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[4 * k4 + e] += v[e]; s2[4 * k4 + e] += v[e] * v[e]; }
            }
        }
        if (p.part) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = s1[r], b = s2[r];
                a = dpp_sum32(a); b = dpp_sum32(b);
                if (l31 == 16) {
                    const int c = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    red[(wave * NT * 32 + c) * 2] = a; red[(wave * NT * 32 + c) * 2 + 1] = b;
                }
            }
        }
    }
    if (p.part) {
        __syncthreads();
        if (tid < NT * 32) {
            const int co = co_base + tid;
            if (co < p.Cout) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) { a += red[(wv * NT * 32 + tid) * 2]; b += red[(wv * NT * 32 + tid) * 2 + 1]; }
                float* dst = p.part + (((size_t)n * nblk + br) * p.Cout + co) * 2;
                dst[0] = a; dst[1] = b;
            }
        }
    }
    }   // bricks of this workgroup
}

// (Cout, 125) occupancy-tap matrix -> split fp16 [k-step 8][hi h0 | hi h1 | lo h0 | lo h1][Co_pad][8]
__global__ void pack_occ_weight16_kernel(const float* __restrict__ m, int Cout, int Co_pad, _Float16* __restrict__ packed) {
    const int total = 8 * 2 * Co_pad * 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, co = (i >> 3) % Co_pad, hh = ((i >> 3) / Co_pad) & 1, ks = (i >> 3) / (2 * Co_pad);
        const int t = 16 * ks + 8 * hh + j;
        const float v = (co < Cout && t < 125) ? m[(size_t)co * 125 + t] : 0.f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * NM_SPLIT_SCALE);
        packed[(((size_t)ks * 4 + hh) * Co_pad + co) * 8 + j] = hi;
        packed[(((size_t)ks * 4 + 2 + hh) * Co_pad + co) * 8 + j] = lo;
    }
}

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvParams p) {
    extern __shared__ f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;

    const int nblk = p.nbz * p.nby * p.nbx;
    const int n = blockIdx.x / nblk, br = blockIdx.x % nblk;
    const int bxi = br % p.nbx, byi = (br / p.nbx) % p.nby, bzi = br / (p.nbx * p.nby);
    const int oz0 = bzi << p.bz_l2, oy0 = byi << p.by_l2, ox0 = bxi << p.bx_l2;
    const int co_base = blockIdx.y * (NT * 32);
    const int BXm = (1 << p.bx_l2) - 1, BYm = (1 << p.by_l2) - 1;

    // LDS voxel offset of each of this lane's A rows (row m -> brick (z,y,x))
    int arow[MT];
    // (tap-split mode, tiny volumes: the 4 / ksplit row tiles that hold voxels are computed by ksplit waves each, on disjoint taps)
    const bool tsplit = MT == 1 && p.w16 && p.ksplit > 1;
    const int cwave = tsplit ? (p.ksplit == 4 ? 0 : (wave & 1)) : wave;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (cwave * MT + mt) * 32 + l31;
        int x = m & BXm, y = (m >> p.bx_l2) & BYm, z = m >> (p.bx_l2 + p.by_l2);
        bool ok = (oz0 + z < p.OD) && (oy0 + y < p.OH) && (ox0 + x < p.OW);
        arow[mt] = ok ? ((z * p.stride) * p.HY + y * p.stride) * p.HX + x * p.stride : 0;
    }

    f32x16 acc[MT][NT];
    f32x16 accl[MT == 1 ? NT : 1];                                  // correction-term accumulators of the split-fp16 core (MT == 1 only)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < (MT == 1 ? NT : 1); ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accl[nt][r] = 0.f;

    const int Q = p.Cin >> 2;
    const int taps = p.ks * p.ks * p.ks;
    const int iz0 = oz0 * p.stride - p.pad, iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
    const size_t tap_stride = (size_t)Q * p.Co_pad;                 // float4 units per tap
    const f32x4* __restrict__ w4 = reinterpret_cast<const f32x4*>(p.w);

    // p.KC channels are staged per pass (as many as the LDS budget allows: tiny volumes take the whole Cin at once,
    // one staging phase + two barriers instead of Cin/16 of them); the MFMA walk consumes them 16 at a time.
    for (int c0 = 0; c0 < p.Cin; c0 += p.KC) {
        const int kc = min(p.KC, p.Cin - c0);
        const int nq = kc >> 2;
        __syncthreads();
        // ---- stage the halo brick of channels [c0, c0+kc) ------------------------------
        if (!p.up2) {
            // (split-fp16 core: 16-channel chunks; a trailing half chunk - Cin = 72 - is filled up with zeros)
            const int nqs = (MT == 1 && p.w16) ? ((kc + 15) >> 4) << 2 : nq;
            for (int i = tid; i < p.HV * nqs; i += 256) {
                int q = i % nqs, hv = i / nqs;
                int hx = hv % p.HX, t2 = hv / p.HX;
                int hy = t2 % p.HY, hz = t2 / p.HY;
                int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q < nq && (unsigned)gz < (unsigned)p.ID && (unsigned)gy < (unsigned)p.IH && (unsigned)gx < (unsigned)p.IW)
                    v = load_act(p, n, gz, gy, gx, c0 + 4 * q);
                lds[q * p.HVp + hv] = v;
            }
        } else {
            // nn.Upsample(x2, trilinear, align_corners=False) of the activated coarse tensor (kypt_detector.py:427,441),
            // fused: (1) the coarse voxels this brick touches go to LDS once (activated), (2) the fine halo tile is
            // interpolated LDS -> LDS.
            f32x4* ldc = lds + (p.KC >> 2) * p.HVp;
            const int cz0 = up_lo(max(iz0, 0)), cy0 = up_lo(max(iy0, 0)), cx0 = up_lo(max(ix0, 0));
            const int CZ = min(p.ID - 1, up_lo(min(iz0 + p.HZ - 1, 2 * p.ID - 1)) + 1) - cz0 + 1;
            const int CY = min(p.IH - 1, up_lo(min(iy0 + p.HY - 1, 2 * p.IH - 1)) + 1) - cy0 + 1;
            const int CX = min(p.IW - 1, up_lo(min(ix0 + p.HX - 1, 2 * p.IW - 1)) + 1) - cx0 + 1;
            for (int i = tid; i < CZ * CY * CX * nq; i += 256) {
                int q = i % nq, cv = i / nq;
                int x = cv % CX, t2 = cv / CX;
                int y = t2 % CY, z = t2 / CY;
                ldc[q * p.CVp + cv] = load_act(p, n, cz0 + z, cy0 + y, cx0 + x, c0 + 4 * q);
            }
            __syncthreads();
            for (int hv = tid; hv < p.HV; hv += 256) {
                int hx = hv % p.HX, t2 = hv / p.HX;
                int hy = t2 % p.HY, hz = t2 / p.HY;
                int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
                if ((unsigned)gz < (unsigned)(2 * p.ID) && (unsigned)gy < (unsigned)(2 * p.IH) && (unsigned)gx < (unsigned)(2 * p.IW)) {
                    int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
                    up_idx(gz, p.ID, z0, z1, lz); up_idx(gy, p.IH, y0, y1, ly); up_idx(gx, p.IW, x0, x1, lx);
                    const float wz0 = 1.f - lz, wy0 = 1.f - ly, wx0 = 1.f - lx;
                    const int r00 = ((z0 - cz0) * CY + (y0 - cy0)) * CX, r01 = ((z0 - cz0) * CY + (y1 - cy0)) * CX;
                    const int r10 = ((z1 - cz0) * CY + (y0 - cy0)) * CX, r11 = ((z1 - cz0) * CY + (y1 - cy0)) * CX;
                    const int a0 = x0 - cx0, a1 = x1 - cx0;
                    for (int q = 0; q < nq; ++q) {
                        const f32x4* cq = ldc + q * p.CVp;
                        lds[q * p.HVp + hv] =
                            wz0 * (wy0 * (wx0 * cq[r00 + a0] + lx * cq[r00 + a1]) + ly * (wx0 * cq[r01 + a0] + lx * cq[r01 + a1])) +
                            lz * (wy0 * (wx0 * cq[r10 + a0] + lx * cq[r10 + a1]) + ly * (wx0 * cq[r11 + a0] + lx * cq[r11 + a1]));
                    }
                } else {
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    for (int q = 0; q < nq; ++q) lds[q * p.HVp + hv] = zero;
                }
            }
        }
        __syncthreads();

        // ---- MFMA over taps x channel octets -------------------------------------------
        if constexpr (MT == 1) {
            if (p.w16) {                                            // split-fp16 core (Cin % 16 == 0: every chunk is 16 wide)
                const half8* w8 = reinterpret_cast<const half8*>(p.w16);
                const size_t ts16 = (size_t)((p.Cin + 15) >> 4) * 4 * p.Co_pad;
                const int tap0 = tsplit ? (p.ksplit == 4 ? wave : (wave >> 1)) : 0;
                for (int s0 = 0; s0 < kc; s0 += 16)
                    mfma_chunk16<NT>(p, lds + (s0 >> 2) * p.HVp, w8 + ((size_t)((c0 + s0) >> 4) * 4 + h) * p.Co_pad + co_base + l31, ts16, taps, h,
                                     arow[0], acc, accl, tap0, tsplit ? p.ksplit : 1);
                continue;
            }
        }
        for (int s0 = 0; s0 < kc; s0 += 16) {
            const f32x4* wq = w4 + ((size_t)((c0 + s0) >> 2) + h) * p.Co_pad + co_base + l31;
            const f32x4* sub = lds + (s0 >> 2) * p.HVp;
            if (kc - s0 >= 16) mfma_chunk<MT, NT, 2>(p, sub, wq, tap_stride, taps, h, arow, acc);
            else               mfma_chunk<MT, NT, 1>(p, sub, wq, tap_stride, taps, h, arow, acc);
        }
    }

    if constexpr (MT == 1) {
        if (p.w16) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][nt][r] += accl[nt][r] * (1.0f / NM_SPLIT_SCALE);
            if (tsplit) {           // sum the tap groups: the waves whose own row tile holds voxels collect, the others end with zeros
                __syncthreads();
                float* r2 = reinterpret_cast<float*>(lds);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) r2[((wave * NT + nt) * 16 + r) * 64 + lane] = acc[0][nt][r];
                __syncthreads();
                const int rw = 4 / p.ksplit;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float sum = 0.f;
                        if (wave < rw) for (int g = 0; g < p.ksplit; ++g) sum += r2[(((wave + g * rw) * NT + nt) * 16 + r) * 64 + lane];
                        acc[0][nt][r] = sum;
                    }
            }
        }
    }
    EpiArgs e;
    e.out = p.out; e.part = p.part; e.bias = p.bias; e.field = nullptr;
    e.OD = p.OD; e.OH = p.OH; e.OW = p.OW; e.Cout = p.Cout; e.bz_l2 = p.bz_l2; e.by_l2 = p.by_l2; e.bx_l2 = p.bx_l2; e.xz_tiles = 0;
    __syncthreads();
    epilogue<MT, NT>(e, reinterpret_cast<float*>(lds), acc, n, br, nblk, oz0, oy0, ox0, co_base);
}

// OIDHW (Cout, Cin, k, k, k) -> [tap][Cin_pad/4][Co_pad][4], zero padded
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int ks,
                                        float* __restrict__ packed, int Cin_pad, int Co_pad) {
    const int taps = ks * ks * ks;
    const size_t total = (size_t)taps * Cin_pad * Co_pad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int j = i & 3;
        size_t r = i >> 2;
        int co = r % Co_pad; r /= Co_pad;
        int q = r % (Cin_pad / 4);
        int tap = r / (Cin_pad / 4);
        int ci = q * 4 + j;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap];
        packed[i] = v;
    }
}

int ceil_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }


// ---- split-fp16 variant ------------------------------------------------------------------------------------------
// Same implicit GEMM on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate) with fp32-equivalent
// accuracy: every fp32 operand v is split as v = hi + lo * 2^-11 with hi = fp16(v), lo = fp16((v - hi) * 2^11), and
//   x * w  ~=  x_hi * w_hi  +  2^-11 (x_hi * w_lo + x_lo * w_hi)
// (the dropped lo*lo term is <= 2^-22 relative, the size of one fp32 rounding).  fp16 x fp16 products are exact in the
// fp32 accumulators; the correction terms use their own accumulator so that nothing depends on fp16 subnormals.
// 3 MFMAs per k-step instead of 1, at 16x the rate.  Activations are split while they are staged into LDS (planes
// [hi|lo][lane half][halo voxel] of 8 halves = 16 B), weights are split once per weight update
// ([tap][Cin/16][hi|lo][lane half][Co_pad][8 halves]).  Requires Cin % 16 == 0; other layers use the fp32 kernel.
// Which brick a persistent workgroup processes in its iteration tau.  The eight XCDs have separate L2s and a brick's
// 6x10x10 halo is 2.3x its 4x8x8 outputs, so in linear order (each workgroup a contiguous run of bricks, the 32 CUs of an
// XCD far apart in the volume) the input is fetched from HBM about twice; conv 64^3 x 32 channels moved 8-10 GB per
// launch for 4.3 GB of tensors.  Here the workgroups of one XCD (blockIdx % 8 is the XCD label) take, at the same time,
// the st_z x st_y x st_x bricks of one super-tile (one brick per workgroup, 64 or 32 of them), and consecutive iterations
// walk neighbouring super-tiles of the same frame, so halos shared between neighbouring bricks are served by that L2.
struct BrickPos { int n, bz, by, bx; };
__device__ __forceinline__ BrickPos super_tile_item(const ConvParams& p, int b, int tau, int per, int nbz, int nby, int nbx) {
    BrickPos r;
    if (p.st_x == 0) {
        const int nbr = nbz * nby * nbx, item = b * per + tau;
        r.n = item / nbr; const int br = item % nbr;
        r.bx = br % nbx; r.by = (br / nbx) % nby; r.bz = br / (nbx * nby);
        return r;
    }
    const int l = b >> 3, S = (b & 7) * per + tau;
    if (p.st_pf) {
        // (round 6) a persistent workgroup decodes every brick it walks - nine integer divisions by launch constants, ~250 vector
        // instructions and a dozen hoisted reciprocal registers per wave, in the MFMA waves' path between two steps (in-kernel stamps of
        // conv_f16p: 720 of a 9 800-tick step).  The super-tile is always 4 x 4 bricks in (y, x); the three real divisions are by host
        // constants: q = umulhi(x, ceil(2^32 / d)), exact while x d < 2^32 (checked by choose_super_tile).
        auto fdiv = [](unsigned x, unsigned m) { return m ? __umulhi(x, m) : x; };
        const unsigned n = fdiv((unsigned)S, p.st_m_pf), si = (unsigned)S - n * p.st_pf;
        const unsigned q = fdiv(si, p.st_m_sx), sx = si - q * p.st_sx;
        const unsigned sz = fdiv(q, p.st_m_sy), sy = q - sz * p.st_sy;
        r.n = (int)n; r.bx = (int)(sx * 4u) + (l & 3); r.by = (int)(sy * 4u) + ((l >> 2) & 3); r.bz = (int)sz * p.st_z + (l >> 4);
        return r;
    }
    const int SX = nbx / p.st_x, SY = nby / p.st_y, stpf = SX * SY * (nbz / p.st_z);
    r.n = S / stpf; const int si = S % stpf;
    r.bx = (si % SX) * p.st_x + l % p.st_x;
    r.by = ((si / SX) % SY) * p.st_y + (l / p.st_x) % p.st_y;
    r.bz = (si / (SX * SY)) * p.st_z + l / (p.st_x * p.st_y);
    return r;
}

// Brick = 4(z) x 8(y) x 8(x) output voxels, 256 GEMM rows.  Each 32-row MFMA tile is the 8(x) x 4(z) slab of one
// brick row y: with the halo tile's plane pitch ZP = 4 (mod 16) slots, the four 16-lane groups of a ds_read_b128
// then hit 16 distinct 16-B slots (conflict-free; the natural (y,x) tile order is a 3-way conflict and makes the
// Cout = 32 layers LDS-bound).  Staging: thread -> one (halo row position, lane half), loops over the halo planes, so
// all index arithmetic and the y/x interpolation weights are computed once per kernel.
// IO: bit 0 = the input tensor is bfloat16, bit 1 = the output tensor is (16-bit storage, conv mode 4)
template <int MT, int NT, int KS, bool UP2, bool SINGLE = false, int IO = 0>
__global__ __launch_bounds__(256, 2) void conv_f16s_kernel(ConvParams p) {
    constexpr bool IH = (IO & 1) != 0, OH = (IO & 2) != 0;
    constexpr int HZ = KS + 3;                                     // halo planes of a 4-deep brick, stride 1
    constexpr int HB = HZ / 2;                                     // planes per load batch
    // Cout = 32 layers: the weights of 9 taps at a time are shared by the block's four waves through LDS (double
    // buffered, filled by direct-to-LDS loads one tap-group ahead).  Four waves each fetching their own copy of
    // B from L2 keep the CU's vector-memory path ~70 % busy and stretch the MFMA phase by ~40 %.
    constexpr bool BLDS = (NT == 1 && KS == 3);
    constexpr int GB = 9 * 4 * 32 * NT;                            // half8 slots of one 9-tap weight group
    extern __shared__ f32x4 lds[];
    half8* ldh = reinterpret_cast<half8*>(lds);                    // [hl*2 + h][HVp] x 16 B, voxel = hz*ZP + hy*HX + hx
    half8* ldb = ldh + 4 * p.HVp;                                  // BLDS: two weight-group buffers of GB slots
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int nblk = p.nbz * p.nby * p.nbx;
    const int co_base = blockIdx.y * (NT * 32);
    const int C16 = p.Cin >> 4;
    const half8* __restrict__ w8 = reinterpret_cast<const half8*>(p.w);
    const size_t plane = (size_t)p.Co_pad;                          // half8 units between (hl,h) planes of one (tap,c16)
    const size_t tap_stride = (size_t)C16 * 4 * plane;
    const int lane_off = h * (int)plane + l31;                      // per-lane part of every weight address
    // Persistent workgroups: each walks a contiguous run of bricks (neighbouring bricks share halo voxels -> L2 reuse
    // in time, and the ~10 us a short-lived workgroup spends being dispatched and retired is paid once).
    const int total_items = p.N * nblk;
    const int per = (total_items + (int)gridDim.x - 1) / (int)gridDim.x;
    const int item_end = min(total_items, ((int)blockIdx.x + 1) * per);
    for (int item = (int)blockIdx.x * per; item < item_end; ++item) {
    const BrickPos bp = super_tile_item(p, (int)blockIdx.x, item - (int)blockIdx.x * per, per, p.nbz, p.nby, p.nbx);
    const int n = bp.n, bxi = bp.bx, byi = bp.by, bzi = bp.bz;
    const int br = (bzi * p.nby + byi) * p.nbx + bxi;
    const int oz0 = bzi << 2, oy0 = byi << 3, ox0 = bxi << 3;
    int arow[MT];
    {
        const int c = l31 >> 2;
        const int x = (((0x96 >> c) & 1) << 2) + (l31 & 3), z = c >> 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int y = wave * MT + mt;
            bool ok = (oz0 + z < p.OD) && (oy0 + y < p.OH) && (ox0 + x < p.OW);
            arow[mt] = ok ? z * p.stride * p.ZP + y * p.stride * p.HX + x * p.stride : 0;
        }
    }
    f32x16 acc[MT][NT], accl[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; accl[mt][nt][r] = 0.f; }

    const int iz0 = oz0 * p.stride - p.pad, iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;

    // this thread's staging column: halo position (hy,hx) and lane half
    const int s_hh = tid & 1, s_r = tid >> 1;
    const bool s_on = s_r < p.HY * p.HX;
    const int s_hy = s_r / p.HX, s_hx = s_r % p.HX;
    const int s_gy = iy0 + s_hy, s_gx = ix0 + s_hx;
    const int s_lds = s_hy * p.HX + s_hx;
    const int us = UP2 ? 2 : 1;
    const bool s_in = s_on && (unsigned)s_gy < (unsigned)(us * p.IH) && (unsigned)s_gx < (unsigned)(us * p.IW);
    // up2: coarse tile geometry and this column's y/x interpolation
    int cz0 = 0, cy0 = 0, cx0 = 0, CZ = 0, CY = 0, CX = 0, u_r0 = 0, u_r1 = 0, u_a0 = 0, u_a1 = 0;
    float u_ly = 0.f, u_lx = 0.f;
    if (UP2) {
        cz0 = up_lo(max(iz0, 0)); cy0 = up_lo(max(iy0, 0)); cx0 = up_lo(max(ix0, 0));
        CZ = min(p.ID - 1, up_lo(min(iz0 + p.HZ - 1, 2 * p.ID - 1)) + 1) - cz0 + 1;
        CY = min(p.IH - 1, up_lo(min(iy0 + p.HY - 1, 2 * p.IH - 1)) + 1) - cy0 + 1;
        CX = min(p.IW - 1, up_lo(min(ix0 + p.HX - 1, 2 * p.IW - 1)) + 1) - cx0 + 1;
        if (s_in) {
            int y0, y1, x0, x1;
            up_idx(s_gy, p.IH, y0, y1, u_ly); up_idx(s_gx, p.IW, x0, x1, u_lx);
            u_r0 = (y0 - cy0) * CX; u_r1 = (y1 - cy0) * CX; u_a0 = x0 - cx0; u_a1 = x1 - cx0;
        }
    }

#ifdef NM_DIAG
#define NM_STAMP(i) do { if (p.stamps && lane == 0) p.stamps[((size_t)item * 4 + wave) * 16 + (i)] = clock64(); } while (0)
#else
#define NM_STAMP(i) do {} while (0)
#endif
    // direct-to-LDS copy of the weights of taps [9g, 9g+9) of chunk cb into weight buffer `buf`
    auto issue_b_group = [&](int cb, int g, int buf) {
        if (!BLDS) return;
        constexpr int PER_TAP = 2 * NT;                            // 1-KiB wave-instructions per tap
#pragma unroll
        for (int k = 0; k < (9 * PER_TAP + 3) / 4; ++k) {
            const int j = wave + 4 * k;                            // wave-uniform instruction index within the group
            if (j < 9 * PER_TAP) {
                // NT == 1: slot (j % 2) * 64 + lane of the tap's 128 = plane (j % 2) * 2 + h, channel l31: a wave-uniform
                // base (scalar registers) plus one per-lane offset shared by every load
                const int t = j / PER_TAP;
                const half8* src = w8 + ((size_t)(9 * g + t) * C16 * 4 + (size_t)cb * 4 + (j % PER_TAP) * 2) * plane + co_base + lane_off;
                half8* dst = ldb + buf * GB + t * (128 * NT) + (j % PER_TAP) * 64;         // + lane * 16 B by hardware
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };
    // raw input of one channel chunk, fetched one chunk ahead (BLDS variants: there are registers to spare at 2 waves/SIMD)
    constexpr bool PFA = BLDS;
    f32x4 pa[PFA && !UP2 ? HZ : 1], pb[PFA && !UP2 ? HZ : 1], prc[4];
    int pci[4] = {-1, -1, -1, -1};
    auto prefetch_chunk = [&](int c0) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        if (!UP2) {
            if (s_on) {
#pragma unroll
                for (int hz = 0; hz < (PFA && !UP2 ? HZ : 1); ++hz) {
                    const int gz = iz0 + hz;
                    const bool in = s_in && (unsigned)gz < (unsigned)p.ID;
                    pa[hz] = in ? load_raw<IH>(p, n, gz, s_gy, s_gx, c0 + 8 * s_hh) : z4;
                    pb[hz] = in ? load_raw<IH>(p, n, gz, s_gy, s_gx, c0 + 8 * s_hh + 4) : z4;
                }
            }
        } else {
            const int ncv = CZ * CY * CX * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + 256 * k;
                pci[k] = -1; prc[k] = z4;
                if (i < ncv) {
                    int q = i & 3, cv = i >> 2;
                    int x = cv % CX, t2 = cv / CX;
                    int y = t2 % CY, z = t2 / CY;
                    pci[k] = q * p.CVp + cv;
                    prc[k] = load_raw<IH>(p, n, cz0 + z, cy0 + y, cx0 + x, c0 + 4 * q);
                }
            }
        }
    };
    if (PFA) prefetch_chunk(0);
    NM_STAMP(0);
#ifdef NM_DIAG
    if (p.stamps && lane == 0) {
        p.stamps[((size_t)item * 4 + wave) * 16 + 15] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
        p.stamps[((size_t)item * 4 + wave) * 16 + 14] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
        p.stamps[((size_t)item * 4 + wave) * 16 + 13] = blockIdx.x;
    }
#endif
    for (int cb = 0; cb < C16; ++cb) {
        const int c0 = cb << 4;
        lds_barrier();
        if (cb < 2) NM_STAMP(1 + cb * 4);
        issue_b_group(cb, 0, 0);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        if (!UP2) {
            if (s_on) {
                // columns outside the volume load zeros and get scale = shift = 0, so they stage exact zeros without a
                // per-value select; whether a halo plane lies inside the volume is uniform over the workgroup
                f32x4 sca = zero4, sha = zero4, scb = zero4, shb = zero4;
                if (p.in_scale && s_in) {
                    const float* ps = p.in_scale + (size_t)n * p.Cin + c0 + 8 * s_hh; const float* ph = p.in_shift + (size_t)n * p.Cin + c0 + 8 * s_hh;
                    sca = *reinterpret_cast<const f32x4*>(ps); scb = *reinterpret_cast<const f32x4*>(ps + 4);
                    sha = *reinterpret_cast<const f32x4*>(ph); shb = *reinterpret_cast<const f32x4*>(ph + 4);
                }
                const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
                if (PFA) {
                    // the raw column was prefetched (before the loop / during the previous chunk's last tap group)
#pragma unroll
                    for (int hz = 0; hz < HZ; ++hz) {
                        half8 hi = zero8, lo = zero8;
                        if ((unsigned)(iz0 + hz) < (unsigned)p.ID) {
                            split8(apply_act(p, pa[hz], sca, sha), apply_act(p, pb[hz], scb, shb), hi, lo);
                        }
                        ldh[s_hh * p.HVp + hz * p.ZP + s_lds] = hi;
                        if constexpr (!SINGLE) ldh[(2 + s_hh) * p.HVp + hz * p.ZP + s_lds] = lo;      // (the one-product modes never read the lo planes)
                    }
                } else {
                    // global loads of half the column first (HB planes x 8 channels), then activate / split / write
#pragma unroll
                    for (int hb = 0; hb < HZ; hb += HB) {
                        f32x4 ra[HB], rb[HB];
#pragma unroll
                        for (int j = 0; j < HB; ++j) {
                            const int gz = iz0 + hb + j;
                            const bool in = s_in && (unsigned)gz < (unsigned)p.ID;
                            ra[j] = in ? load_raw<IH>(p, n, gz, s_gy, s_gx, c0 + 8 * s_hh) : zero4;
                            rb[j] = in ? load_raw<IH>(p, n, gz, s_gy, s_gx, c0 + 8 * s_hh + 4) : zero4;
                        }
#pragma unroll
                        for (int j = 0; j < HB; ++j) {
                            const int hz = hb + j;
                            half8 hi = zero8, lo = zero8;
                            if ((unsigned)(iz0 + hz) < (unsigned)p.ID)
                                split8(apply_act(p, ra[j], sca, sha), apply_act(p, rb[j], scb, shb), hi, lo);
                            ldh[s_hh * p.HVp + hz * p.ZP + s_lds] = hi;
                            if constexpr (!SINGLE) ldh[(2 + s_hh) * p.HVp + hz * p.ZP + s_lds] = lo;      // (the one-product modes never read the lo planes)
                        }
                    }
                }
            }
        } else {
            // fused Upsample(x2, trilinear): activated coarse voxels -> LDS (fp32), then interpolate + split
            f32x4* ldc = lds + 4 * p.HVp + (BLDS ? GB : 0);   // BLDS: aliases weight buffer 1 (idle during staging)
            {
                const int ncv = CZ * CY * CX * 4;
                if (!PFA) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = tid + 256 * k;
                        pci[k] = -1; prc[k] = zero4;
                        if (i < ncv) {
                            int q = i & 3, cv = i >> 2;
                            int x = cv % CX, t2 = cv / CX;
                            int y = t2 % CY, z = t2 / CY;
                            pci[k] = q * p.CVp + cv;
                            prc[k] = load_raw<IH>(p, n, cz0 + z, cy0 + y, cx0 + x, c0 + 4 * q);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (pci[k] >= 0) {
                        const int c = c0 + 4 * ((tid + 256 * k) & 3);
                        f32x4 sc = zero4, sh = zero4;
                        if (p.in_scale) { sc = *reinterpret_cast<const f32x4*>(p.in_scale + (size_t)n * p.Cin + c); sh = *reinterpret_cast<const f32x4*>(p.in_shift + (size_t)n * p.Cin + c); }
                        ldc[pci[k]] = apply_act(p, prc[k], sc, sh);
                    }
                }
                for (int i = tid + 1024; i < ncv; i += 256) {          // (never taken for 4x8x8 bricks; kept for safety)
                    int q = i & 3, cv = i >> 2;
                    int x = cv % CX, t2 = cv / CX;
                    int y = t2 % CY, z = t2 / CY;
                    ldc[q * p.CVp + cv] = load_act<IH>(p, n, cz0 + z, cy0 + y, cx0 + x, c0 + 4 * q);
                }
            }
            lds_barrier();
            if (s_on) {
                // separable: bilinear (y,x) interpolation of a coarse plane is computed once and reused by the two or
                // three fine planes that blend it (same association as the direct trilinear formula)
                const float wy0 = 1.f - u_ly, wx0 = 1.f - u_lx;
                const f32x4* cq0 = ldc + (2 * s_hh) * p.CVp;
                const f32x4* cq1 = ldc + (2 * s_hh + 1) * p.CVp;
                auto bilin = [&](const f32x4* cq, int pz) {
                    const f32x4 a = cq[pz + u_r0 + u_a0], b = cq[pz + u_r0 + u_a1], c = cq[pz + u_r1 + u_a0], d = cq[pz + u_r1 + u_a1];
                    f32x4 r;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        r[j] = __builtin_fmaf(u_ly, __builtin_fmaf(u_lx, d[j], wx0 * c[j]), wy0 * __builtin_fmaf(u_lx, b[j], wx0 * a[j]));
                    return r;
                };
                int zc0 = -1, zc1 = -1;
                f32x4 b0a = zero4, b0b = zero4, b1a = zero4, b1b = zero4;      // bilinear planes z0 / z1, channel quads a / b
#pragma unroll 1
                for (int hz = 0; hz < HZ; ++hz) {
                    const int gz = iz0 + hz;
                    f32x4 va = zero4, vb = zero4;
                    if (s_in && (unsigned)gz < (unsigned)(2 * p.ID)) {
                        int z0, z1; float lz;
                        up_idx(gz, p.ID, z0, z1, lz);
                        if (z0 != zc0) {                                       // wave-uniform: gz is the same in every lane
                            if (z0 == zc1) { b0a = b1a; b0b = b1b; }
                            else { b0a = bilin(cq0, (z0 - cz0) * CY * CX); b0b = bilin(cq1, (z0 - cz0) * CY * CX); }
                            zc0 = z0;
                        }
                        if (z1 != zc1) {
                            if (z1 == zc0) { b1a = b0a; b1b = b0b; }
                            else { b1a = bilin(cq0, (z1 - cz0) * CY * CX); b1b = bilin(cq1, (z1 - cz0) * CY * CX); }
                            zc1 = z1;
                        }
                        const float wz0 = 1.f - lz;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            va[j] = __builtin_fmaf(lz, b1a[j], wz0 * b0a[j]);
                            vb[j] = __builtin_fmaf(lz, b1b[j], wz0 * b0b[j]);
                        }
                    }
                    half8 hi, lo;
                    split8(va, vb, hi, lo);
                    ldh[s_hh * p.HVp + hz * p.ZP + s_lds] = hi;
                    if constexpr (!SINGLE) ldh[(2 + s_hh) * p.HVp + hz * p.ZP + s_lds] = lo;      // (the one-product modes never read the lo planes)
                }
            }
        }
        if (cb < 2) NM_STAMP(2 + cb * 4);
        if (BLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // weight group 0 has landed in LDS
        lds_barrier();
        if (cb < 2) NM_STAMP(3 + cb * 4);
        __builtin_amdgcn_s_setprio(2);       // MFMA phase: ahead of the co-resident workgroup's staging VALU

        if (BLDS) {
            // One software-pipelined pass over the 27 taps.  Left to itself the scheduler sinks every ds_read next to its
            // use and waits on it, and a wave that has the matrix pipe to itself then keeps it under 50 % busy.  Here
            // each operand register is refilled for the next tap right after the last MFMA that reads it (the B pair is
            // double buffered), so a read has 3-6 MFMAs (100-200 cycles) to land; sched_barrier pins that order.
            // Accumulation order per accumulator is unchanged: acc += ah*bh; accl += ah*bl, then al*bh.
            static_assert(!BLDS || MT == 2, "pipelined tap loop is written for two M tiles");
            const half8* a_h = ldh + h * p.HVp;
            const half8* a_l = ldh + (2 + h) * p.HVp;
            auto aoff = [&](int t) { return (t / 3) * p.HX + (t % 3); };                    // within one 9-tap group
            half8 ah0 = a_h[arow[0]], al0 = a_l[arow[0]], ah1 = a_h[arow[MT - 1]], al1 = a_l[arow[MT - 1]];
            half8 bhv[2], blv[2];
#pragma unroll 1
            for (int g = 0; g < 3; ++g) {
                if (g < 2) issue_b_group(cb, g + 1, (g + 1) & 1);
                else if (PFA && cb + 1 < C16) prefetch_chunk(c0 + 16);
                const half8* bb = ldb + (g & 1) * GB + h * 32 + l31;
                const int zo = g * p.ZP, zn = min(g + 1, 2) * p.ZP;             // plane offsets of this / the next group
                bhv[0] = bb[0]; blv[0] = bb[64];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int c = t & 1;
                    const int nxt = (t < 8) ? zo + aoff(t + 1) : zn;             // A offset of the next tap (next group: tap 0)
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bhv[c], acc[0][0], 0, 0, 0);
                    if (t < 8) { bhv[c ^ 1] = bb[(t + 1) * 128]; blv[c ^ 1] = bb[(t + 1) * 128 + 64]; }
                    __builtin_amdgcn_sched_barrier(0);
                    accl[0][0] = nm_mfma_lo<SINGLE>(ah0, blv[c], accl[0][0]);
                    ah0 = a_h[arow[0] + nxt];
                    __builtin_amdgcn_sched_barrier(0);
                    acc[MT - 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bhv[c], acc[MT - 1][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    accl[MT - 1][0] = nm_mfma_lo<SINGLE>(ah1, blv[c], accl[MT - 1][0]);
                    ah1 = a_h[arow[MT - 1] + nxt];
                    __builtin_amdgcn_sched_barrier(0);
                    accl[0][0] = nm_mfma_lo<SINGLE>(al0, bhv[c], accl[0][0]);
                    al0 = a_l[arow[0] + nxt];
                    __builtin_amdgcn_sched_barrier(0);
                    accl[MT - 1][0] = nm_mfma_lo<SINGLE>(al1, bhv[c], accl[MT - 1][0]);
                    al1 = a_l[arow[MT - 1] + nxt];
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (g < 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lds_barrier(); }
            }
        } else {
            // weights of (tap, cb): planes hi/lo for this lane half come straight from L2, prefetched PF taps ahead
            // (a tap is only 6-12 MFMAs = 200-400 cycles, about an L2 round trip); the A operands of tap t+1 are
            // read from LDS while the MFMAs of tap t issue
            constexpr int PF = (NT == 1) ? 2 : 1;
            constexpr int TAPS = KS * KS * KS;
            const half8* wq = w8 + (size_t)cb * 4 * plane + co_base;               // wave-uniform; lanes add lane_off
            half8 bh[PF + 1][NT], bl[PF + 1][NT];
#pragma unroll
            for (int f = 0; f < PF; ++f) {
                const half8* wf = wq + (size_t)min(f, TAPS - 1) * tap_stride;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { bh[f][nt] = wf[lane_off + nt * 32]; bl[f][nt] = (wf + 2 * plane)[lane_off + nt * 32]; }
            }
            // same rotation as above: A operands of tap t+1 are read from LDS right after the last MFMA of tap t that
            // uses the register (6-10 MFMAs ahead of their first use)
            const half8* a_h = ldh + h * p.HVp;
            const half8* a_l = ldh + (2 + h) * p.HVp;
            auto aoff = [&](int t) { return (t / KS) * p.HX + (t % KS); };                  // within one z-plane of taps
            constexpr int TP = KS * KS;
            half8 ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { ah[mt] = a_h[arow[mt]]; al[mt] = a_l[arow[mt]]; }
#pragma unroll 1
            for (int tz = 0; tz < KS; ++tz) {
                const int zo = tz * p.ZP, zn = min(tz + 1, KS - 1) * p.ZP;
#pragma unroll
                for (int t = 0; t < TP; ++t) {
                    const int tap = tz * TP + t;
                    const int nxt = (t + 1 < TP) ? zo + aoff(t + 1) : zn;
                    const half8* wn = wq + (size_t)min(tap + PF, TAPS - 1) * tap_stride;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { bh[PF][nt] = wn[lane_off + nt * 32]; bl[PF][nt] = (wn + 2 * plane)[lane_off + nt * 32]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[0][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) accl[mt][nt] = nm_mfma_lo<SINGLE>(ah[mt], bl[0][nt], accl[mt][nt]);
                        ah[mt] = a_h[arow[mt] + nxt];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) accl[mt][nt] = nm_mfma_lo<SINGLE>(al[mt], bh[0][nt], accl[mt][nt]);
                        al[mt] = a_l[arow[mt] + nxt];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int f = 0; f < PF; ++f)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) { bh[f][nt] = bh[f + 1][nt]; bl[f][nt] = bl[f + 1][nt]; }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (cb < 2) NM_STAMP(4 + cb * 4);
    }
    NM_STAMP(9);
    EpiArgs e;
    e.out = p.out; e.part = p.part; e.bias = p.bias; e.field = nullptr;
    e.OD = p.OD; e.OH = p.OH; e.OW = p.OW; e.Cout = p.Cout; e.bz_l2 = 2; e.by_l2 = 3; e.bx_l2 = 3; e.xz_tiles = 1;
    lds_barrier();
    epilogue_xz<MT, NT, OH>(e, reinterpret_cast<float*>(lds), acc, accl, n, br, nblk, oz0, oy0, ox0, co_base);
    NM_STAMP(10);
    }   // persistent item loop
}

// ---- conv_pool_f16s: the k2 s2 p0 "pool" convs in split-fp16 ------------------------------------------------------------
// Every input voxel feeds exactly one (output voxel, tap) pair, so nothing is shared between rows and nothing goes through
// LDS: lane (row l31 = output voxel, half h) loads its own 8 channels of the tap's input voxel (32 contiguous bytes),
// applies the pending GroupNorm affine + LeakyReLU, splits to hi / lo fp16 and feeds the MFMAs; the two lane halves and the
// channel chunks of one wave read each 128-byte line completely, back to back.  The generic fp32 kernel staged these layers
// 8-16 channels per pass with 512 workgroups in flight and fetched the 64^3 x 32 input 2.7x from HBM (PMC FETCH_SIZE).
// The kernel is bound by that one read of the input; weights (8 taps, a few KB per chunk) come from L2.
template <int NT, bool SINGLE = false>
__global__ __launch_bounds__(256, 2) void conv_pool_f16s_kernel(ConvParams p) {
    __shared__ float red[4 * NT * 32 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int nblk = p.nbz * p.nby * p.nbx;
    const int n = blockIdx.x / nblk, br = blockIdx.x % nblk;
    const int bxi = br % p.nbx, byi = (br / p.nbx) % p.nby, bzi = br / (p.nbx * p.nby);
    const int oz0 = bzi << 2, oy0 = byi << 3, ox0 = bxi << 3;
    const int co_base = blockIdx.y * (NT * 32);
    const int C16 = p.Cin >> 4;
    const half8* __restrict__ w8 = reinterpret_cast<const half8*>(p.w);
    const size_t plane = (size_t)p.Co_pad;
    const int lane_off = h * (int)plane + l31;
    // rows: tile mt = the 8(x) x 4(z) slab of brick row y = 2 wave + mt (the layout epilogue_xz stores)
    const int c = l31 >> 2;
    const int x = (((0x96 >> c) & 1) << 2) + (l31 & 3), z = c >> 1;
    const float* src[2];
    const float* src2[2];
    bool ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int y = 2 * wave + mt;
        ok[mt] = (oz0 + z < p.OD) && (oy0 + y < p.OH) && (ox0 + x < p.OW);
        const size_t voff = (((size_t)(2 * (oz0 + z)) * p.IH + 2 * (oy0 + y)) * p.IW + 2 * (ox0 + x)) * p.Cin + 8 * h;
        src[mt] = p.in + (size_t)n * p.ID * p.IH * p.IW * p.Cin + voff;
        src2[mt] = p.in2 ? p.in2 + (size_t)n * p.ID * p.IH * p.IW * p.Cin + voff : src[mt];
        if (p.in_map && ok[mt]) {
            // brick-sparse input: the 2x2x2 fine voxels of an output voxel lie in one 4x8x8 input brick; an unwritten brick is the
            // frame-independent tensor in_alt there
            const int fb = (((oz0 + z) >> 1) * (p.IH >> 3) + ((oy0 + y) >> 2)) * (p.IW >> 3) + ((ox0 + x) >> 2);
            if (!p.in_map[(size_t)n * (p.ID >> 2) * (p.IH >> 3) * (p.IW >> 3) + fb]) src[mt] = p.in_alt + voff;
        }
    }
    f32x16 acc[2][NT], accl[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; accl[mt][nt][r] = 0.f; }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int cb = 0; cb < C16; ++cb) {
        f32x4 sca = zero4, scb = zero4, sha = zero4, shb = zero4;
        if (p.in_scale) {
            const float* ps = p.in_scale + (size_t)n * p.Cin + cb * 16 + 8 * h; const float* ph = p.in_shift + (size_t)n * p.Cin + cb * 16 + 8 * h;
            sca = *reinterpret_cast<const f32x4*>(ps); scb = *reinterpret_cast<const f32x4*>(ps + 4);
            sha = *reinterpret_cast<const f32x4*>(ph); shb = *reinterpret_cast<const f32x4*>(ph + 4);
        }
        // second addend of an un-materialised residual sum: its own affine (and slope), apply2's arithmetic (mul, add - not fused)
        f32x4 s2a = zero4, s2b = zero4, h2a = zero4, h2b = zero4;
        if (p.in2 && p.in2_scale) {
            const float* ps = p.in2_scale + (size_t)n * p.Cin + cb * 16 + 8 * h; const float* ph = p.in2_shift + (size_t)n * p.Cin + cb * 16 + 8 * h;
            s2a = *reinterpret_cast<const f32x4*>(ps); s2b = *reinterpret_cast<const f32x4*>(ps + 4);
            h2a = *reinterpret_cast<const f32x4*>(ph); h2b = *reinterpret_cast<const f32x4*>(ph + 4);
        }
        const half8* wq = w8 + (size_t)cb * 4 * plane + co_base;
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) {
            const size_t toff = ((size_t)((tap >> 2) * p.IH + ((tap >> 1) & 1)) * p.IW + (tap & 1)) * p.Cin + cb * 16;
            f32x4 ra[2], rb[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ra[mt] = ok[mt] ? *reinterpret_cast<const f32x4*>(src[mt] + toff) : zero4;
                rb[mt] = ok[mt] ? *reinterpret_cast<const f32x4*>(src[mt] + toff + 4) : zero4;
            }
            if (p.in2) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f32x4 qa = ok[mt] ? *reinterpret_cast<const f32x4*>(src2[mt] + toff) : zero4;
                    const f32x4 qb = ok[mt] ? *reinterpret_cast<const f32x4*>(src2[mt] + toff + 4) : zero4;
                    ra[mt] = ok[mt] ? sum_act2(p, ra[mt], sca, sha, qa, s2a, h2a) : zero4;
                    rb[mt] = ok[mt] ? sum_act2(p, rb[mt], scb, shb, qb, s2b, h2b) : zero4;
                }
            }
            const half8* wt = wq + (size_t)tap * C16 * 4 * plane;
            half8 bh[NT], bl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bh[nt] = wt[lane_off + nt * 32]; bl[nt] = (wt + 2 * plane)[lane_off + nt * 32]; }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                half8 ah, al;
                if (p.in2) split8(ra[mt], rb[mt], ah, al);
                else split8(apply_act(p, ra[mt], sca, sha), apply_act(p, rb[mt], scb, shb), ah, al);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
                    accl[mt][nt] = nm_mfma_lo<SINGLE>(ah, bl[nt], accl[mt][nt]);
                    accl[mt][nt] = nm_mfma_lo<SINGLE>(al, bh[nt], accl[mt][nt]);
                }
            }
        }
    }
    EpiArgs e;
    e.out = p.out; e.part = p.part; e.bias = p.bias; e.field = nullptr;
    e.OD = p.OD; e.OH = p.OH; e.OW = p.OW; e.Cout = p.Cout; e.bz_l2 = 2; e.by_l2 = 3; e.bx_l2 = 3; e.xz_tiles = 1;
    epilogue_xz<2, NT>(e, red, acc, accl, n, br, nblk, oz0, oy0, ox0, co_base);
}

// 16-byte global load the compiler's wait-count tracking does not see.  hipcc waits vmcnt(0) at the first use of an
// ordinary load result whenever an LDS-DMA is in flight, which would drain the weight pipeline of conv_f16p at the first
// staging piece; the kernel instead waits explicitly (its group-end vmcnt(0)) before the results are used.
__device__ __forceinline__ f32x4 load16_untracked(const float* q) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(q) : "memory");
    return r;
}

// 16-byte LDS read the compiler's wait-count tracking does not see (byte address = base + OFF).  The pipelined tap stream of
// conv_f16p keeps 12 reads in flight and waits with an exact s_waitcnt lgkmcnt(10) before each MFMA; with tracked reads hipcc
// falls back to lgkmcnt(0) every second tap, which drains the read issued one instruction earlier.
template <int OFF>
__device__ __forceinline__ half8 lds_read16_untracked(unsigned base) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    half8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(base), "n"(OFF) : "memory");
    return r;
}

// 8-byte form (four bfloat16 channels of a 16-bit-storage tensor), same tracking rules
__device__ __forceinline__ nm_u32x2 load8_untracked(const float* base, unsigned byte_off) {
    nm_u32x2 r;
    const char* q = reinterpret_cast<const char*>(base) + byte_off;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(q) : "memory");
    return r;
}

// the same from a wave-uniform base and a 32-bit per-lane byte offset (one 64-bit VALU add; the scalar-base addressing form
// needs the base in SGPRs, which the compiler does not guarantee for an inline-asm operand computed through VALU divisions)
__device__ __forceinline__ f32x4 load16_untracked(const float* base, unsigned byte_off) {
    return load16_untracked(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off));
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }
template <int V> using ic = std::integral_constant<int, V>;

// ---- conv_pool_f16q: conv_pool_f16s with a counted, one-tap-ahead memory pipeline (round 3) --------------------------------------
// conv_pool_f16s_kernel's loop compiled to 247 branches and 110 s_waitcnt vmcnt(0): every load sat under a condition (row inside the
// volume, tensor has a pending affine, second addend present), so each tap was load -> full wait -> convert -> MFMAs, a chain of
// C16 x 8 memory round trips per workgroup (64 -> 128 @32^3 -> 16^3 x 64 frames: 450 us for a 0.54 GB read).  Here: rows outside
// the volume read the tensor's first voxel (their results are never stored), the frame's pending affines sit in an LDS table (identity
// where a tensor has none), the second addend is a template parameter, and a tap's input rows and weights are requested while the previous tap is
// converted and multiplied - no branch inside the loop, every wait a counted one.  Same arithmetic per element.
template <int NT, bool IN2> struct PoolBuf { f32x4 ra[2], rb[2], qa[IN2 ? 2 : 1], qb[IN2 ? 2 : 1]; half8 bh[NT], bl[NT]; };

// IO (bit 0: bfloat16 input, bit 1: bfloat16 output; IN2 = false only): a row's 16 channels of a tap are ONE 16-byte load per lane half
template <int NT, bool SINGLE, bool IN2, int IO = 0>
__global__ __launch_bounds__(256, 2) void conv_pool_f16q_kernel(ConvParams p) {
    constexpr bool IH = (IO & 1) != 0, OH = (IO & 2) != 0;
    static_assert(!(IH && IN2), "the un-materialised residual sum exists in the inference forward only (fp32 workspace)");
    __shared__ float red[4 * NT * 32 * 2];
    __shared__ __attribute__((aligned(16))) float s_aff[4][128];       // this frame's pending affines: scale, shift, second addend's scale, shift (Cin <= 128)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int nblk = p.nbz * p.nby * p.nbx;
    int n = blockIdx.x / nblk, br = blockIdx.x % nblk;
    if (p.in_map && (nblk & 7) == 0) {
        // Brick-sparse input (the sparse first layer's output): most bricks of most frames read the SAME weight-only field brick
        // (262 KB of a 33.5 MB field that no L2 holds).  In frame-major order the 64 frames' reads of one brick position are 1 024
        // workgroups apart and every frame pulls the field through the fabric again (round 4: 2.15 GB fetched for a 0.33 GB tensor).
        // Here workgroup ids walk the FRAMES of one brick position first, and a brick position's workgroups all share blockIdx % 8 -
        // the XCD - so the field brick is fetched into that XCD's L2 once and the other frames hit it there.  Same work per
        // workgroup, same results.
        const int g = (int)blockIdx.x, q = g >> 3;
        n = q % p.N; br = (q / p.N) * 8 + (g & 7);
    }
    const int bxi = br % p.nbx, byi = (br / p.nbx) % p.nby, bzi = br / (p.nbx * p.nby);
    const int oz0 = bzi << 2, oy0 = byi << 3, ox0 = bxi << 3;
    const int co_base = blockIdx.y * (NT * 32);
    const int C16 = p.Cin >> 4;
    for (int i = tid; i < 4 * p.Cin; i += 256) {
        const int which = i / p.Cin, cc = i % p.Cin;
        const float* t = which == 0 ? p.in_scale : which == 1 ? p.in_shift : which == 2 ? p.in2_scale : p.in2_shift;
        s_aff[which][cc] = t && (which < 2 || IN2) ? t[(size_t)n * p.Cin + cc] : ((which & 1) ? 0.f : 1.f);
    }
    const half8* __restrict__ w8 = reinterpret_cast<const half8*>(p.w);
    const size_t plane = (size_t)p.Co_pad;
    const int lane_off = h * (int)plane + l31;
    const int c = l31 >> 2;
    const int x = (((0x96 >> c) & 1) << 2) + (l31 & 3), z = c >> 1;
    const float* src[2];
    const float* src2[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int y = 2 * wave + mt;
        const bool ok = (oz0 + z < p.OD) && (oy0 + y < p.OH) && (ox0 + x < p.OW);
        const size_t voff = ok ? (((size_t)(2 * (oz0 + z)) * p.IH + 2 * (oy0 + y)) * p.IW + 2 * (ox0 + x)) * p.Cin + 8 * h : (size_t)(8 * h);
        const size_t foff = (size_t)n * p.ID * p.IH * p.IW * p.Cin;
        src[mt] = nm_eptr(p.in, foff + voff, IH);
        src2[mt] = IN2 ? p.in2 + foff + voff : src[mt];
        if (p.in_map && ok) {
            const int fb = (((oz0 + z) >> 1) * (p.IH >> 3) + ((oy0 + y) >> 2)) * (p.IW >> 3) + ((ox0 + x) >> 2);
            if (!p.in_map[(size_t)n * (p.ID >> 2) * (p.IH >> 3) * (p.IW >> 3) + fb]) src[mt] = p.in_alt + voff;
        }
    }
    f32x16 acc[2][NT], accl[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; accl[mt][nt][r] = 0.f; }
    // the pending affines of channel chunk cb, from the LDS table (identity where a tensor has none)
    struct Aff { f32x4 sca, scb, sha, shb, s2a, s2b, h2a, h2b; };
    auto load_aff = [&](int cb, Aff& a) __attribute__((always_inline)) {
        const int o = cb * 16 + 8 * h;
        a.sca = *reinterpret_cast<const f32x4*>(&s_aff[0][o]); a.scb = *reinterpret_cast<const f32x4*>(&s_aff[0][o + 4]);
        a.sha = *reinterpret_cast<const f32x4*>(&s_aff[1][o]); a.shb = *reinterpret_cast<const f32x4*>(&s_aff[1][o + 4]);
        if constexpr (IN2) {
            a.s2a = *reinterpret_cast<const f32x4*>(&s_aff[2][o]); a.s2b = *reinterpret_cast<const f32x4*>(&s_aff[2][o + 4]);
            a.h2a = *reinterpret_cast<const f32x4*>(&s_aff[3][o]); a.h2b = *reinterpret_cast<const f32x4*>(&s_aff[3][o + 4]);
        }
    };
    // (the chunk index goes through an opaque statement: with it visible, src + tap offset is loop-invariant for every (tap, row, tensor)
    //  and the compiler keeps all of those addresses in registers - 64 of them - and spills)
    auto issue = [&](int cb, int tap, PoolBuf<NT, IN2>& b) __attribute__((always_inline)) {
        asm volatile("" : "+s"(cb));
        const size_t toff = ((size_t)((tap >> 2) * p.IH + ((tap >> 1) & 1)) * p.IW + (tap & 1)) * p.Cin + cb * 16;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if constexpr (IH) {          // 8 bf16 channels = 16 bytes: one load, widened at the point of use (compute)
                const f32x4 q = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(src[mt]) + toff);
                b.ra[mt] = q;
            } else { b.ra[mt] = *reinterpret_cast<const f32x4*>(src[mt] + toff); b.rb[mt] = *reinterpret_cast<const f32x4*>(src[mt] + toff + 4); }
            if constexpr (IN2) { b.qa[mt] = *reinterpret_cast<const f32x4*>(src2[mt] + toff); b.qb[mt] = *reinterpret_cast<const f32x4*>(src2[mt] + toff + 4); }
        }
        if constexpr (!IN2) {      // (with a second addend in flight the weights are requested at the tap itself: registers)
            const half8* wt = w8 + (size_t)cb * 4 * plane + co_base + (size_t)tap * C16 * 4 * plane;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { b.bh[nt] = wt[lane_off + nt * 32]; if constexpr (!SINGLE) b.bl[nt] = (wt + 2 * plane)[lane_off + nt * 32]; }
        }
    };
    // branch-free forms of apply_act / sum_act2 (a uniform branch inside the tap loop ends the basic block, and the waits at its join
    // are not counted ones): the identity affine (x * 1 + 0) and slope 1 (max(v, v) / v > 0 ? v : v) leave the values as they are
    const float slope1 = p.in_slope, slope2 = p.in2_slope;
    auto act1 = [&](f32x4 v, const f32x4& sc, const f32x4& sh) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = __builtin_fmaf(v[j], sc[j], sh[j]); v[j] = fmaxf(v[j], v[j] * slope1); }
        return v;
    };
    auto act2 = [&](f32x4 a, const f32x4& sca, const f32x4& sha, const f32x4& b, const f32x4& scb, const f32x4& shb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u = a[j] * sca[j]; u = u + sha[j]; u = u > 0.f ? u : u * slope1;
            float v = b[j] * scb[j]; v = v + shb[j]; v = v > 0.f ? v : v * slope2;
            a[j] = u + v;
        }
        return a;
    };
    auto compute = [&](PoolBuf<NT, IN2>& b, const Aff& a, int cb, int tap) __attribute__((always_inline)) {
        if constexpr (IN2) {
            const half8* wt = w8 + (size_t)cb * 4 * plane + co_base + (size_t)tap * C16 * 4 * plane;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { b.bh[nt] = wt[lane_off + nt * 32]; if constexpr (!SINGLE) b.bl[nt] = (wt + 2 * plane)[lane_off + nt * 32]; }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            half8 ah, al;
            if constexpr (IN2) split8(act2(b.ra[mt], a.sca, a.sha, b.qa[mt], a.s2a, a.h2a), act2(b.rb[mt], a.scb, a.shb, b.qb[mt], a.s2b, a.h2b), ah, al);
            else if constexpr (IH) {
                const f32x4 q = b.ra[mt];
                const unsigned u0 = nm_fbits(q[0]), u1 = nm_fbits(q[1]), u2 = nm_fbits(q[2]), u3 = nm_fbits(q[3]);
                split8(act1(f32x4{nm_bf_lo(u0), nm_bf_hi(u0), nm_bf_lo(u1), nm_bf_hi(u1)}, a.sca, a.sha),
                       act1(f32x4{nm_bf_lo(u2), nm_bf_hi(u2), nm_bf_lo(u3), nm_bf_hi(u3)}, a.scb, a.shb), ah, al);
            } else split8(act1(b.ra[mt], a.sca, a.sha), act1(b.rb[mt], a.scb, a.shb), ah, al);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.bh[nt], acc[mt][nt], 0, 0, 0);
                accl[mt][nt] = nm_mfma_lo<SINGLE>(ah, b.bl[nt], accl[mt][nt]);
                accl[mt][nt] = nm_mfma_lo<SINGLE>(al, b.bh[nt], accl[mt][nt]);
            }
        }
    };
    if constexpr (IN2) {
        // two tensors per row: the pipeline's unit is one ROW TILE of a tap (16 registers in flight per buffer instead of 32); the
        // tap's weights are requested with its first row tile
        f32x4 xa[2], xb[2], ya[2], yb[2];                     // [buffer]
        auto issue_row = [&](int cb, int tap, int mt, int bi) __attribute__((always_inline)) {
            asm volatile("" : "+s"(cb));
            const size_t toff = ((size_t)((tap >> 2) * p.IH + ((tap >> 1) & 1)) * p.IW + (tap & 1)) * p.Cin + cb * 16;
            xa[bi] = *reinterpret_cast<const f32x4*>(src[mt] + toff); xb[bi] = *reinterpret_cast<const f32x4*>(src[mt] + toff + 4);
            ya[bi] = *reinterpret_cast<const f32x4*>(src2[mt] + toff); yb[bi] = *reinterpret_cast<const f32x4*>(src2[mt] + toff + 4);
        };
        Aff af;
        half8 bh[NT], bl[NT];
        issue_row(0, 0, 0, 0);
        __syncthreads();                                      // the affine table
        for (int cb = 0; cb < C16; ++cb) {
            const int cbn = cb + 1 < C16 ? cb + 1 : cb;
            load_aff(cb, af);
            static_for<16>([&](auto U) __attribute__((always_inline)) {
                constexpr int u = decltype(U)::value, tap = u >> 1, mt = u & 1;
                if constexpr (u < 15) issue_row(cb, (u + 1) >> 1, (u + 1) & 1, (u + 1) & 1); else issue_row(cbn, 0, 0, 0);
                if constexpr (mt == 0) {
                    const half8* wt = w8 + (size_t)cb * 4 * plane + co_base + (size_t)tap * C16 * 4 * plane;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { bh[nt] = wt[lane_off + nt * 32]; if constexpr (!SINGLE) bl[nt] = (wt + 2 * plane)[lane_off + nt * 32]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                half8 ah, al;
                split8(act2(xa[mt], af.sca, af.sha, ya[mt], af.s2a, af.h2a), act2(xb[mt], af.scb, af.shb, yb[mt], af.s2b, af.h2b), ah, al);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
                    accl[mt][nt] = nm_mfma_lo<SINGLE>(ah, bl[nt], accl[mt][nt]);
                    accl[mt][nt] = nm_mfma_lo<SINGLE>(al, bh[nt], accl[mt][nt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    } else {
    PoolBuf<NT, IN2> b0, b1;
    Aff af;
    issue(0, 0, b0);
    __syncthreads();                                          // the affine table
    for (int cb = 0; cb < C16; ++cb) {
        const int cbn = cb + 1 < C16 ? cb + 1 : cb;           // (behind the last chunk: its first tap again, never used)
        load_aff(cb, af);
        static_for<8>([&](auto T) __attribute__((always_inline)) {
            constexpr int tap = decltype(T)::value;
            PoolBuf<NT, IN2>& cur = (tap & 1) ? b1 : b0;
            PoolBuf<NT, IN2>& nxt = (tap & 1) ? b0 : b1;
            if constexpr (tap < 7) issue(cb, tap + 1, nxt); else issue(cbn, 0, nxt);
            __builtin_amdgcn_sched_barrier(0);
            compute(cur, af, cb, tap);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    }
    EpiArgs e;
    e.out = p.out; e.part = p.part; e.bias = p.bias; e.field = nullptr;
    e.OD = p.OD; e.OH = p.OH; e.OW = p.OW; e.Cout = p.Cout; e.bz_l2 = 2; e.by_l2 = 3; e.bx_l2 = 3; e.xz_tiles = 1;
    epilogue_xz<2, NT, OH>(e, red, acc, accl, n, br, nblk, oz0, oy0, ox0, co_base);
}

// ---- conv_f16p: k3 s1 p1 split-fp16 conv, one persistent workgroup per CU, MFMA waves + producer waves ---------------
// conv_f16s runs two independent 4-wave workgroups per CU and leaves the matrix pipe ~50 % idle: staging, the epilogue
// and the per-brick set-up of one workgroup overlap the other's MFMAs only by chance.  Here ONE workgroup owns the CU and
// walks a flat stream of steps (4x8x8 brick, 32-channel cout group, 16-channel chunk), with two kinds of waves:
//   * waves 0-3, one per SIMD, issue nothing but the step's 27 taps x 6 MFMAs, one LDS operand read per MFMA gap (three
//     taps deep, one exact lgkmcnt wait per tap) and the previous brick's epilogue, cut into <= 3-VALU pieces pinned into
//     the gaps (eight 16-byte stores from transposed accumulators, then DPP partial sums for the GroupNorm); two copies of
//     the stream, with / without epilogue pieces, instead of a branch per gap; the accumulators wait for nothing else;
//   * waves 4-7, the second wave of each SIMD, are producers: they fetch the raw input two steps ahead (16-byte pieces,
//     four lanes per voxel), apply the pending GroupNorm affine + LeakyReLU, split to hi/lo fp16 and write the other halo
//     buffer - all before the second barrier of step s, so the MFMA waves read the first operands of step s+1 during the
//     last two taps of step s - and stream the weights: 9-tap groups through LDS, three buffers (buffer g always holds tap
//     group g), loaded through registers one group ahead of the barrier that publishes them.
// Three barriers per step (tap-group ends).  Cout is processed 32 channels per step, the cout groups of a brick back to
// back (its input stays in L2).
//   LDS: halo [2 buffers][hi h0 | hi h1 | lo h0 | lo h1][600] x 16 B, weights [3][9 taps][4 planes][32] x 16 B, GroupNorm scratch.
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// IO: bit 0 = bfloat16 input (a producer piece is then an 8-byte load of the same four channels), bit 1 = bfloat16 output
// LATE (round 6; NM355_F16P_LATE=1 - opt-in, NOT the default: the schedule below is 5 % faster on this kernel, and on some boxes one
// evaluation in ~1500 of the detector's training forward came back non-finite under it, never under the round-2 schedule - a load still in
// flight somewhere in the two-step body; not root-caused in the round, profiles/r06_not_shipped_ab.txt): in-kernel stamps of the 32 -> 32 @64^3 layer
// (tools/diag_f16p_steps.py) showed the MFMA waves waiting 760 + 1188 of a 10 400-tick step at the first two barriers - the producers
// arrive last there: each weight group was loaded and WAITED for inside the phase that stores it (an L2 round trip per phase, the whole
// third phase nothing else), and the tile of step s + 1 had to be complete at the SECOND barrier (pieces dealt 4 / 6 / 0).  As in
// conv_f16p2 since round 3: the tile is complete at the step's LAST barrier (pieces 3 / 4 / 3; the MFMA waves request a step's first two
// taps' A operands behind that barrier instead of during taps 25 / 26), and every weight group is requested one phase before it is
// stored (two register sets, their roles alternate with the step's parity).
template <bool UP2, bool SINGLE = false, int IO = 0, int LATE_ = 1>
__global__ __launch_bounds__(512, 1) void conv_f16p_kernel(ConvParams p) {
    constexpr bool LATE = LATE_ == 1, LATE_M = LATE_ != 0;      // (LATE_ == 2: diagnostic - old producer schedule, MFMA waves read the next tile late)
    constexpr bool IH = (IO & 1) != 0, OH = (IO & 2) != 0;
    constexpr unsigned EB = IH ? 2u : 4u;                           // bytes per input element
    constexpr int HV = 600, ZP = 100, HX = 10;
    constexpr int GB = 9 * 4 * 32;                                  // half8 slots of one weight group
    constexpr int NP = 10;                                          // 16-byte input pieces per producer thread and step (2400 / 256)
    extern __shared__ f32x4 lds[];
    half8* ldh = reinterpret_cast<half8*>(lds);                     // [2][4][HV]
    half8* ldb = ldh + 8 * HV;                                      // [3][GB]
    float* red = reinterpret_cast<float*>(ldb + 3 * GB);            // [4 waves][32 channels][2]
    float* lbias = red + 256;                                       // [Cout] bias (zeros without one), written once by the producers
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int C16 = p.Cin >> 4;
    const int ncg = p.Cout >> 5;
    const int nbx = p.OW >> 3, nby = p.OH >> 3, nbz = p.OD >> 2, nbr = nbx * nby * nbz;
    const int total = p.N * nbr * ncg;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int id_first = (int)blockIdx.x * per, id_last = min(total, id_first + per);
    if (id_first >= id_last) return;

    struct Work { int n, oz0, oy0, ox0, cg; };
    const bool tiled = p.st_x != 0;                                 // XCD super-tile order (one cout group only), see super_tile_item
    auto decode = [&](int id) {
        Work w;
        if (tiled) {
            const BrickPos bp = super_tile_item(p, (int)blockIdx.x, id - id_first, per, nbz, nby, nbx);
            w.cg = 0; w.n = bp.n; w.ox0 = bp.bx << 3; w.oy0 = bp.by << 3; w.oz0 = bp.bz << 2;
            return w;
        }
        w.cg = id % ncg; const int bid = id / ncg;
        w.n = bid / nbr; const int br = bid % nbr;
        w.ox0 = (br % nbx) << 3; w.oy0 = ((br / nbx) % nby) << 3; w.oz0 = (br / (nbx * nby)) << 2;
        return w;
    };
    auto next_work = [&](Work w, int id) {                          // work item id from item id - 1 (linear order: no divisions)
        if (tiled) return decode(id);
        if (++w.cg == ncg) {
            w.cg = 0;
            if ((w.ox0 += 8) == p.OW) { w.ox0 = 0; if ((w.oy0 += 8) == p.OH) { w.oy0 = 0; if ((w.oz0 += 4) == p.OD) { w.oz0 = 0; ++w.n; } } }
        }
        return w;
    };
    struct Step { Work w; int id, cb; };
    auto advance = [&](Step& st) {                                  // false (and st unchanged) on the block's last step
        if (st.cb + 1 < C16) { ++st.cb; return true; }
        if (st.id + 1 >= id_last) return false;
        st.cb = 0; ++st.id; st.w = next_work(st.w, st.id);
        return true;
    };
#ifdef NM_DIAG
    int step_no = 0;
#define NM_PSTAMP(i) do { if (p.stamps && lane == 0 && step_no < 64) p.stamps[(((size_t)blockIdx.x * 64 + step_no) * 8 + wave) * 16 + (i)] = clock64(); } while (0)
#else
#define NM_PSTAMP(i) do {} while (0)
#endif
    if (wave >= 4) {
        // =========================== producer waves ===========================================================
        const int pt = tid - 256, pw = wave - 4;
        const half8* __restrict__ w8 = reinterpret_cast<const half8*>(p.w);
        const size_t plane = (size_t)p.Co_pad;
        const int lane_off = h * (int)plane + l31;
        // pieces of this thread: piece k = channels 4 * quad .. +3 (of the chunk's 16) of halo voxel (pt >> 2) + 64k; four
        // neighbouring lanes fetch one voxel's 64 contiguous bytes
        const int quad = pt & 3;
        int pc_slot[NP], pc_rel[NP], pc_pos[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int v = (pt >> 2) + 64 * k, vc = min(v, HV - 1);
            const int hz = vc / 100, r = vc % 100, hy = r / 10, hx = r % 10;
            pc_slot[k] = (quad >> 1) * HV + hz * ZP + hy * HX + hx;    // half8 slot of the hi part in a halo buffer (lo: + 2 HV)
            pc_rel[k] = ((hz * p.IH + hy) * p.IW + hx) * p.Cin + 4 * quad;
            pc_pos[k] = (v < HV) ? (hz | (hy << 8) | (hx << 16)) : -1;
        }
        using RawT = std::conditional_t<IH, nm_u32x2, f32x4>;
        RawT raw[NP]; f32x4 sc, sh, scB = {0.f, 0.f, 0.f, 0.f}, shB = scB, wreg[5], wreg2[5];      // (scB / shB: LATE's second affine pair)
#pragma unroll
        for (int i = 0; i < 5; ++i) { wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; wreg2[i] = wreg[i]; }       // (SINGLE uses three of them; all five are operands of the counted waits)
        const bool has_affine = p.in_scale != nullptr;
        const unsigned rel111 = (unsigned)(((p.IH + 1) * p.IW + 1) * p.Cin) * EB;   // halo voxel (1,1,1): always inside
        // bit k: piece k of a brick's halo tile lies inside the volume
        auto inside_mask = [&](const Work& w) {
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int hz = pc_pos[k] & 0xff, hy = (pc_pos[k] >> 8) & 0xff, hx = pc_pos[k] >> 16;
                const bool in = pc_pos[k] >= 0 && (unsigned)(w.oz0 - 1 + hz) < (unsigned)p.ID && (unsigned)(w.oy0 - 1 + hy) < (unsigned)p.IH &&
                                (unsigned)(w.ox0 - 1 + hx) < (unsigned)p.IW;
                m |= (in ? 1u : 0u) << k;
            }
            return m;
        };
        auto tile_base = [&](const Work& w, int cb) {               // wave-uniform: halo voxel (0,0,0), channel 16 cb (may lie
            return nm_eptr(p.in, (size_t)(((((long long)w.n * p.ID + (w.oz0 - 1)) * p.IH + (w.oy0 - 1)) * p.IW + (w.ox0 - 1)) * (long long)p.Cin + cb * 16), IH);   // outside)
        };
        auto load_piece = [&](const float* base, unsigned mask, auto K) {      // outside pieces fetch an inside voxel (zeroed later)
            constexpr int k = decltype(K)::value;
            if constexpr (IH) raw[k] = load8_untracked(base, ((mask >> k) & 1) ? (unsigned)pc_rel[k] * EB : rel111);
            else raw[k] = load16_untracked(base, ((mask >> k) & 1) ? (unsigned)pc_rel[k] * EB : rel111);
        };
        auto load_affine = [&](const Work& w, int cb, auto AFF) {             // always two loads: the waits below count instructions
            const float* ps = has_affine ? p.in_scale + (size_t)w.n * p.Cin + cb * 16 : p.in;
            const float* ph = has_affine ? p.in_shift + (size_t)w.n * p.Cin + cb * 16 : p.in;
            if constexpr (decltype(AFF)::value) { scB = load16_untracked(ps, 16u * quad); shB = load16_untracked(ph, 16u * quad); }
            else { sc = load16_untracked(ps, 16u * quad); sh = load16_untracked(ph, 16u * quad); }
        };
        // wait until all but the N youngest vector-memory operations are done; ties every load destination to the wait
#define NM_PRODUCER_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" \
            : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]), \
              "+v"(raw[8]), "+v"(raw[9]), "+v"(sc), "+v"(sh), "+v"(wreg[0]), "+v"(wreg[1]), "+v"(wreg[2]), "+v"(wreg[3]), "+v"(wreg[4]), \
              "+v"(wreg2[0]), "+v"(wreg2[1]), "+v"(wreg2[2]), "+v"(wreg2[3]), "+v"(wreg2[4]), "+v"(scB), "+v"(shB) :: "memory")
        auto convert = [&](int buf, unsigned mask, auto K, auto AFF) {        // activate, split, write piece k into halo buffer buf (AFF: affine pair)
            constexpr int k = decltype(K)::value;
            const f32x4& sc_ = decltype(AFF)::value ? scB : sc; const f32x4& sh_ = decltype(AFF)::value ? shB : sh;
            half4v hi4, lo4;
            const float keep = ((mask >> k) & 1) ? 1.f : 0.f;       // padding is zero AFTER the activation
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                float v0, v1;
                if constexpr (IH) { const unsigned u = raw[k][e >> 1]; v0 = nm_bf_lo(u); v1 = nm_bf_hi(u); }
                else { v0 = raw[k][e]; v1 = raw[k][e + 1]; }
                if (has_affine) { v0 = __builtin_fmaf(v0, sc_[e], sh_[e]); v1 = __builtin_fmaf(v1, sc_[e + 1], sh_[e + 1]); }
                // LeakyReLU (slope in (0, 1]) and the padding mask in two instructions per value: max(v, slope v) * keep
                v0 = fmaxf(v0 * keep, (v0 * keep) * p.in_slope); v1 = fmaxf(v1 * keep, (v1 * keep) * p.in_slope);
                half2v hv = __builtin_convertvector(f32x2{v0, v1}, half2v);
                asm volatile("" : "+v"(hv));
                const float t0 = v0 * NM_SPLIT_SCALE, t1 = v1 * NM_SPLIT_SCALE;
                hi4[e] = hv[0]; hi4[e + 1] = hv[1];
                if constexpr (!SINGLE) {                            // (conv mode 3 has no use for the lo halves)
                    lo4[e] = (_Float16)__builtin_fmaf((float)hv[0], -NM_SPLIT_SCALE, t0);
                    lo4[e + 1] = (_Float16)__builtin_fmaf((float)hv[1], -NM_SPLIT_SCALE, t1);
                }
            }
            if (pc_pos[k] >= 0) {
                half4v* dst = reinterpret_cast<half4v*>(ldh + buf * 4 * HV + pc_slot[k]) + (quad & 1);
                dst[0] = hi4;
                if constexpr (!SINGLE) dst[4 * HV] = lo4;           // + 2 HV half8 slots
            }
        };
        // direct-to-LDS copy of tap group g of (cout group cg, chunk cb) into weight buffer g: 18 one-KiB wave loads; every
        // producer wave issues five (two are issued twice) so that the counted waits are the same in all of them.  The
        // per-wave source / destination offsets inside a tap group are fixed: per call only a scalar base is computed.
        // SINGLE (conv modes 3 / 4): the lo halves (j odd) are never read by the MFMA waves - nine hi pieces, three per wave (see conv_f16p2)
        constexpr int NWP = SINGLE ? 3 : 5;
        unsigned dma_src[NWP], dma_dst[NWP];
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int j = SINGLE ? 2 * min(pw + 4 * i, 8) : ((i < 4) ? pw + 4 * i : min(pw + 16, 17)), t = j >> 1;
            dma_src[i] = (unsigned)(((size_t)t * C16 * 4 + (j & 1) * 2) * plane + lane_off) * 16u;       // bytes, + lane part
            dma_dst[i] = (unsigned)(t * 128 + (j & 1) * 64 + lane) * 16u;
        }
        // Through registers, not direct-to-LDS: an LDS-DMA instruction holds the issuing wave for ~200 cycles in this kernel
        // (15 per producer wave and step), a 16-byte load + ds_write_b128 pair a fraction of that.
        auto load_b_group = [&](int cg, int cb, int g, auto SET) {
            const float* base = reinterpret_cast<const float*>(w8 + ((size_t)(9 * g) * C16 * 4 + (size_t)cb * 4) * plane + cg * 32);
#pragma unroll
            for (int i = 0; i < NWP; ++i) { if constexpr (decltype(SET)::value) wreg2[i] = load16_untracked(base, dma_src[i]); else wreg[i] = load16_untracked(base, dma_src[i]); }
        };
        auto store_b_group = [&](int g, auto SET) {
            char* lbase = reinterpret_cast<char*>(ldb + g * GB);
#pragma unroll
            for (int i = 0; i < NWP; ++i) *reinterpret_cast<f32x4*>(lbase + dma_dst[i]) = decltype(SET)::value ? wreg2[i] : wreg[i];
        };

        // The input runs two steps ahead of the MFMA waves: during step s the tile of step s+1 (loaded during step s-1) is
        // converted piece by piece, and each piece's registers are refilled at once with the load for step s+2.
        Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
        int hb = 0;
        Step s1 = cur; bool has1 = advance(s1);                     // step s+1 (= cur on the last step: staged, never read)
        Step s2 = s1;  bool has2 = has1 && advance(s2);
        unsigned m_cvt = inside_mask(cur.w), m_ld = m_cvt;
        for (int c = pt; c < p.Cout; c += 256) lbias[c] = p.bias ? p.bias[c] : 0.f;
        // prologue: first tile + first two weight groups in the open, loads of the second tile in flight
        load_b_group(cur.w.cg, 0, 0, ic<0>{});
        load_affine(cur.w, 0, ic<0>{});
        { const float* b = tile_base(cur.w, 0); static_for<NP>([&](auto K) { load_piece(b, m_cvt, K); }); }
        NM_PRODUCER_WAIT(0);
        store_b_group(0, ic<0>{});
        load_b_group(cur.w.cg, 0, 1, ic<0>{});
        static_for<NP>([&](auto K) { convert(0, m_cvt, K, ic<0>{}); });
        NM_PRODUCER_WAIT(0);
        store_b_group(1, ic<0>{});
        m_cvt = inside_mask(s1.w);
        load_affine(s1.w, s1.cb, ic<0>{});
        { const float* b = tile_base(s1.w, s1.cb); static_for<NP>([&](auto K) { load_piece(b, m_cvt, K); }); }
        if constexpr (LATE) {
        // (weight group 2 of the first step: requested here, stored in its first phase)
        load_b_group(cur.w.cg, cur.cb, 2, ic<0>{});
        lds_barrier();
        // Loads per step and thread, in issue order: phase 0 [W0' x NWP][pieces 0-2], phase 1 [W1' x NWP][pieces 3-6], phase 2 [W2' x NWP]
        // [pieces 7-9][affine x 2]; Wg' = tap group g of the NEXT step (group 2: of the step after this phase, i.e. also the next).
        // Set PAR holds the group requested in the previous phase.
        auto step_body = [&](auto PARC) {
            constexpr int PAR = decltype(PARC)::value;
            if (s2.w.n != s1.w.n || s2.w.oz0 != s1.w.oz0 || s2.w.oy0 != s1.w.oy0 || s2.w.ox0 != s1.w.ox0) m_ld = inside_mask(s2.w);
            else m_ld = m_cvt;
            const float* b2 = tile_base(s2.w, s2.cb);
            NM_PSTAMP(0);
            // phase 0: request W0 of the next step; then wait for W2 of this step and everything older (this tile's pieces, its affine) - NOT
            // for the three piece loads phase 2 has just issued (pieces 7-9 of the tile after this one: an HBM round trip)
            load_b_group(s1.w.cg, s1.cb, 0, ic<PAR ^ 1>{});
            if constexpr (SINGLE) { NM_PRODUCER_WAIT(6); } else { NM_PRODUCER_WAIT(8); }
            NM_PSTAMP(1);
            static_for<3>([&](auto K) { convert(hb ^ 1, m_cvt, K, ic<PAR>{}); load_piece(b2, m_ld, K); });
            store_b_group(2, ic<PAR>{});                            // W2 of THIS step (buffer 2 was last read before the previous step's last barrier)
            NM_PSTAMP(2);
            lds_barrier();
            NM_PSTAMP(3);
            // phase 1: request W1 of the next step, store its W0 (younger than W0': 3 + NWP + 4 + 2 loads)
            load_b_group(s1.w.cg, s1.cb, 1, ic<PAR>{});
            static_for<4>([&](auto K) { convert(hb ^ 1, m_cvt, ic<decltype(K)::value + 3>{}, ic<PAR>{}); load_piece(b2, m_ld, ic<decltype(K)::value + 3>{}); });
            load_affine(s2.w, s2.cb, ic<PAR ^ 1>{});                 // the NEXT tile's affine into the other pair, a phase and a half before its first use
            NM_PSTAMP(4);
            if constexpr (SINGLE) { NM_PRODUCER_WAIT(12); } else { NM_PRODUCER_WAIT(14); }
            store_b_group(0, ic<PAR ^ 1>{});
            NM_PSTAMP(5);
            lds_barrier();
            NM_PSTAMP(6);
            // phase 2: request W2 of the next step (stored in its phase 0), store its W1 (younger than W1': 4 + 2 + NWP + 3); the tile of
            // step s + 1 is complete at this barrier
            load_b_group(s1.w.cg, s1.cb, 2, ic<PAR ^ 1>{});
            static_for<3>([&](auto K) { convert(hb ^ 1, m_cvt, ic<decltype(K)::value + 7>{}, ic<PAR>{}); load_piece(b2, m_ld, ic<decltype(K)::value + 7>{}); });
            if constexpr (SINGLE) { NM_PRODUCER_WAIT(12); } else { NM_PRODUCER_WAIT(14); }
            store_b_group(1, ic<PAR>{});
            NM_PSTAMP(7);
            lds_barrier();
            NM_PSTAMP(8);
        };
        // ONE straight-line body of two steps (the register sets swap roles with the step's parity): with the two parities as two
        // branches of a loop the compiler reconciles their register assignments at the join - by moving registers whose untracked loads
        // are still in flight (wrong weights whenever consecutive steps differ by more than their parity: found by the C16 = 4 tests)
        auto next_step = [&]() {
            cur = s1; s1 = s2; has1 = has2; has2 = has2 && advance(s2);
            m_cvt = m_ld; hb ^= 1;
#ifdef NM_DIAG
            ++step_no;
#endif
        };
        for (;;) {
            step_body(ic<0>{});
            if (!has1) break;
            next_step();
            step_body(ic<1>{});
            if (!has1) break;
            next_step();
        }
        } else {
        lds_barrier();
        for (;;) {
            // the tile being loaded (step s+2); its inside mask changes only with the brick
            if (s2.w.n != s1.w.n || s2.w.oz0 != s1.w.oz0 || s2.w.oy0 != s1.w.oy0 || s2.w.ox0 != s1.w.ox0) m_ld = inside_mask(s2.w);
            else m_ld = m_cvt;
            const float* b2 = tile_base(s2.w, s2.cb);
            NM_PSTAMP(0);
            // tap group 0: weights of this step's group 2; pieces 0-3.  Everything older than the 5 weight loads has landed
            // after the first wait (the tile of step s+1 and its affine were issued a step ago).
            load_b_group(cur.w.cg, cur.cb, 2, ic<0>{});
            if constexpr (SINGLE) { NM_PRODUCER_WAIT(3); } else { NM_PRODUCER_WAIT(5); }      // (the NWP weight loads are the youngest)
            NM_PSTAMP(1);
            static_for<4>([&](auto K) { convert(hb ^ 1, m_cvt, K, ic<0>{}); load_piece(b2, m_ld, K); });
            NM_PRODUCER_WAIT(4);                                    // the weight loads (older than the 4 new piece loads)
            store_b_group(2, ic<0>{});
            NM_PSTAMP(2);
            lds_barrier();
            NM_PSTAMP(3);
            // tap group 1: weights of the next step's group 0; pieces 4-9; the halo tile is complete at this barrier
            load_b_group(s1.w.cg, s1.cb, 0, ic<0>{});
            static_for<6>([&](auto K) { convert(hb ^ 1, m_cvt, ic<decltype(K)::value + 4>{}, ic<0>{}); load_piece(b2, m_ld, ic<decltype(K)::value + 4>{}); });
            load_affine(s2.w, s2.cb, ic<0>{});                               // (sc / sh are free: piece 9 was their last user)
            NM_PSTAMP(4);
            NM_PRODUCER_WAIT(8);                                    // the weight loads: 6 piece + 2 affine loads are younger
            store_b_group(0, ic<0>{});
            NM_PSTAMP(5);
            lds_barrier();
            NM_PSTAMP(6);
            // tap group 2: weights of the next step's group 1
            load_b_group(s1.w.cg, s1.cb, 1, ic<0>{});
            NM_PRODUCER_WAIT(0);                                    // (everything: the input loads are a group or more old)
            store_b_group(1, ic<0>{});
            NM_PSTAMP(7);
            lds_barrier();
            NM_PSTAMP(8);
            if (!has1) break;
            cur = s1; s1 = s2; has1 = has2; has2 = has2 && advance(s2);
            m_cvt = m_ld; hb ^= 1;
#ifdef NM_DIAG
            ++step_no;
#endif
        }
        }
        NM_PRODUCER_WAIT(0);                                        // loads still in flight for a step that does not exist
        lds_barrier();                                              // the two barriers of the MFMA waves' drain
        lds_barrier();
        return;
    }

    // =============================== MFMA waves ===============================================================
    __builtin_amdgcn_s_setprio(3);
    // MFMA rows of this wave: tile mt = the 8(x) x 4(z) slab of brick row y = 2 * wave + mt
    int arow0;
    {
        const int c = l31 >> 2;
        const int x = (((0x96 >> c) & 1) << 2) + (l31 & 3), z = c >> 1;
        arow0 = z * ZP + (2 * wave) * HX + x;
    }
    f32x16 acc[2], accl[2], outv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[mt][r] = 0.f; accl[mt][r] = 0.f; outv[mt][r] = 0.f; }

    // ---- epilogue of a finished brick, in pieces.  The MFMAs below take the weights as the first operand, so a lane holds
    // one voxel (l31 -> x, z of the slab; tile mt -> brick row y) and its 16 registers the output channels
    // (r & 3) + 8 (r >> 2) + 4 h: four consecutive channels per register quad -> 16-byte stores (a quarter of the store
    // instructions of the channel-per-lane layout; vector-memory instruction issue is what this kernel runs out of first).
    Work epi;                                                       // the brick whose outputs sit in outv
    epi.n = 0; epi.oz0 = 0; epi.oy0 = 0; epi.ox0 = 0; epi.cg = 0;
    const int vx = ((((0x96 >> (l31 >> 2)) & 1) << 2) + (l31 & 3)), vz = l31 >> 3;
    auto epi_store = [&](auto E) {                                  // store e = 4 mt + k4 of the 8 per lane
        constexpr int e = decltype(E)::value, mt = e >> 2, k4 = e & 3;
        const size_t dst = ((((size_t)epi.n * p.OD + epi.oz0 + vz) * p.OH + epi.oy0 + 2 * wave + mt) * p.OW + epi.ox0 + vx) * (size_t)p.Cout +
                           epi.cg * 32 + 8 * k4 + 4 * h;
        nm_st4<OH>(p.out, dst, f32x4{outv[mt][4 * k4], outv[mt][4 * k4 + 1], outv[mt][4 * k4 + 2], outv[mt][4 * k4 + 3]});
    };
    // GroupNorm partial sums, in place: outv[0][r] <- sum over the two tiles, outv[1][r] <- sum of squares; then over the 32
    // voxels (lanes of one half) with DPP adds; rows 1 and 3 of the wave end up holding the totals
    auto epi_sum_local = [&](auto R) {
        constexpr int r = decltype(R)::value;
        const float a = outv[0][r], b = outv[1][r];
        outv[0][r] = a + b;
        outv[1][r] = a * a + b * b;
    };
    auto epi_sum_lanes = [&](auto I) {                              // DPP step (i >> 5) of value (i & 31) = 16 m + r: consecutive
        constexpr int i = decltype(I)::value, st = i >> 5, v = i & 31, m = v >> 4, r = v & 15;    // pieces touch different registers
        constexpr int ctrl = st == 0 ? 0xB1 : st == 1 ? 0x4E : st == 2 ? 0x141 : st == 3 ? 0x140 : 0x142;   // quad_perm [1,0,3,2], [2,3,0,1],
        constexpr int rows = st == 4 ? 0xa : 0xf;                                                            // row_half_mirror, row_mirror, row_bcast15
        const float x = outv[m][r];
        outv[m][r] = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rows, 0xf, true));
    };
    auto epi_red = [&](auto R) {                                    // lanes 16 / 48 park (sum, sum of squares) of channel c(r, h)
        constexpr int r = decltype(R)::value;
        if (p.part && l31 == 16) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
            *reinterpret_cast<f32x2*>(red + (wave * 32 + c) * 2) = f32x2{outv[0][r], outv[1][r]};
        }
    };
    auto epi_part = [&]() {                                         // after a barrier: 32 threads sum the four wave partials
        if (p.part && tid < 32) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { a += red[(i * 32 + tid) * 2]; b += red[(i * 32 + tid) * 2 + 1]; }
            const int br = ((epi.oz0 >> 2) * nby + (epi.oy0 >> 3)) * nbx + (epi.ox0 >> 3);
            float* dst = p.part + (((size_t)epi.n * nbr + br) * p.Cout + epi.cg * 32 + tid) * 2;
            dst[0] = a; dst[1] = b;
        }
    };
    auto epi_take = [&](const Work& w) {                            // outputs of the brick just accumulated (64 VALU, exposed)
        f32x4 b4[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) b4[k4] = *reinterpret_cast<const f32x4*>(lbias + w.cg * 32 + 8 * k4 + 4 * h);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) outv[mt][r] = (acc[mt][r] + accl[mt][r] * (1.0f / NM_SPLIT_SCALE)) + b4[r >> 2][r & 3];
        epi = w;
    };

    Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
    int hb = 0;
    lds_barrier();                                                  // the producers' prologue

    // operand registers, three taps deep (index = tap % 3); LDS byte addresses of this lane's rows / weight column
    const unsigned vb = (unsigned)(size_t)(ldb + h * 32 + l31);
    constexpr int LO = 2 * HV * 16;                                 // hi -> lo plane of a halo tile, bytes
    constexpr int YO = HX * 16;                                     // brick row 2 wave -> 2 wave + 1
    half8 ah0[3], al0[3], ah1[3], al1[3], bhv[3], blv[3];
    {
        const unsigned va = (unsigned)(size_t)(ldh + h * HV + arow0);
        ah0[0] = lds_read16_untracked<0>(va); al0[0] = SINGLE ? half8{} : lds_read16_untracked<LO>(va);
        ah1[0] = lds_read16_untracked<YO>(va); al1[0] = SINGLE ? half8{} : lds_read16_untracked<LO + YO>(va);
        ah0[1] = lds_read16_untracked<16>(va); al0[1] = SINGLE ? half8{} : lds_read16_untracked<LO + 16>(va);
        ah1[1] = lds_read16_untracked<YO + 16>(va); al1[1] = SINGLE ? half8{} : lds_read16_untracked<LO + YO + 16>(va);
        bhv[0] = lds_read16_untracked<0>(vb); blv[0] = SINGLE ? half8{} : lds_read16_untracked<64 * 16>(vb);
        bhv[1] = lds_read16_untracked<128 * 16>(vb); blv[1] = SINGLE ? half8{} : lds_read16_untracked<192 * 16>(vb);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ah0[0]), "+v"(al0[0]), "+v"(ah1[0]), "+v"(al1[0]), "+v"(ah0[1]), "+v"(al0[1]), "+v"(ah1[1]), "+v"(al1[1]),
                       "+v"(bhv[0]), "+v"(blv[0]), "+v"(bhv[1]), "+v"(blv[1]) :: "memory");
    }
    bool pending = false;                                           // outv / epi hold a brick whose epilogue has not run
    bool part_due = false;                                          // red[] holds that brick's partial sums
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (;;) {
        Step nxt = cur;
        const bool has_next = advance(nxt);
        if (part_due) { epi_part(); part_due = false; }
        const bool run_epi = pending;
        // One read per MFMA gap, issued two taps (12 MFMAs, ~400 cycles) before its first use; before each MFMA "all but the
        // 10 youngest LDS operations are done" covers both of its operands (other LDS traffic only makes that stricter).
        const unsigned va = (unsigned)(size_t)(ldh + hb * 4 * HV + h * HV + arow0);          // this step's halo tile
        const unsigned van = (unsigned)(size_t)(ldh + (hb ^ 1) * 4 * HV + h * HV + arow0);   // the next step's
        NM_PSTAMP(0);
        // two copies of the stream (with / without epilogue pieces): a branch per gap would cost more than the pieces
        auto stream = [&](auto EPI) {
        constexpr bool with_epi = decltype(EPI)::value;
        static_for<27>([&](auto TT) {
            constexpr int tt = decltype(TT)::value, t = tt % 9, i = tt % 3, j = (tt + 2) % 3;
            auto gap = [&](auto SUB) {
                constexpr int sub = decltype(SUB)::value, q = tt * 6 + sub;
                if constexpr (with_epi) {                           // <= 3 VALU or one memory instruction per gap
                    if constexpr (q >= 2 && q < 18 && (q & 1) == 0) epi_store(ic<((q - 2) >> 1) & 7>{});
                    if constexpr (q >= 18 && q < 34) epi_sum_local(ic<(q - 18) & 15>{});
                    if constexpr (q >= 34 && q < 114) { epi_sum_lanes(ic<(2 * (q - 34)) % 160>{}); epi_sum_lanes(ic<(2 * (q - 34) + 1) % 160>{}); }
                    if constexpr (q >= 116 && q < 132) epi_red(ic<(q - 116) & 15>{});
                }
            };
            constexpr int u = tt + 2, ug = (u < 27) ? u / 9 : 0, ut = (u < 27) ? u % 9 : u - 27;
            constexpr int AO = (ug * ZP + (ut / 3) * HX + (ut % 3)) * 16;          // byte offsets of tap tt + 2 (taps 25, 26: of the
            constexpr int BO = (ug * GB + ut * 128) * 16;                           // next step's taps 0, 1)
            const unsigned vau = (u < 27) ? va : van;
            constexpr bool aread = !LATE_M || u < 27;               // LATE: the next tile is complete only at the step's last barrier
#define NM_WAIT_OPERANDS(x, y) asm volatile("" : "+v"(x), "+v"(y))
            // one wait per tap: this tap's six operands were issued during tap tt - 2; the six reads of tap tt - 1 are the only
            // younger LDS operations (anything else in the gaps only makes the wait stricter)
            // (LATE: tap 25 requests only the two B operands of the next step's tap 0 - its A operands wait for the step's last barrier -
            //  so at tap 26 the reads of tap 24 are covered by "all but the 2 youngest")
            if constexpr (LATE_M && tt == 26)
                asm volatile("s_waitcnt lgkmcnt(2)"
                             : "+v"(ah0[i]), "+v"(al0[i]), "+v"(ah1[i]), "+v"(al1[i]), "+v"(bhv[i]), "+v"(blv[i]) :: "memory");
            else
            asm volatile("s_waitcnt lgkmcnt(6)"
                         : "+v"(ah0[i]), "+v"(al0[i]), "+v"(ah1[i]), "+v"(al1[i]), "+v"(bhv[i]), "+v"(blv[i]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            NM_WAIT_OPERANDS(ah0[i], bhv[i]);
            if constexpr (tt == 0 && with_epi) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhv[i], ah0[i], zero16, 0, 0, 0);
            else acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhv[i], ah0[i], acc[0], 0, 0, 0);
            bhv[j] = lds_read16_untracked<BO>(vb);
            gap(ic<0>{});
            __builtin_amdgcn_sched_barrier(0);
            NM_WAIT_OPERANDS(ah0[i], blv[i]);
            if constexpr (tt == 0 && with_epi) accl[0] = nm_mfma_lo<SINGLE>(blv[i], ah0[i], zero16);
            else accl[0] = nm_mfma_lo<SINGLE>(blv[i], ah0[i], accl[0]);
            if constexpr (aread) ah0[j] = lds_read16_untracked<AO>(vau);
            gap(ic<1>{});
            __builtin_amdgcn_sched_barrier(0);
            NM_WAIT_OPERANDS(ah1[i], bhv[i]);
            if constexpr (tt == 0 && with_epi) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhv[i], ah1[i], zero16, 0, 0, 0);
            else acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhv[i], ah1[i], acc[1], 0, 0, 0);
            blv[j] = SINGLE ? half8{} : lds_read16_untracked<BO + 64 * 16>(vb);
            gap(ic<2>{});
            __builtin_amdgcn_sched_barrier(0);
            NM_WAIT_OPERANDS(ah1[i], blv[i]);
            if constexpr (tt == 0 && with_epi) accl[1] = nm_mfma_lo<SINGLE>(blv[i], ah1[i], zero16);
            else accl[1] = nm_mfma_lo<SINGLE>(blv[i], ah1[i], accl[1]);
            if constexpr (aread) ah1[j] = lds_read16_untracked<AO + YO>(vau);
            gap(ic<3>{});
            __builtin_amdgcn_sched_barrier(0);
            NM_WAIT_OPERANDS(al0[i], bhv[i]);
            accl[0] = nm_mfma_lo<SINGLE>(bhv[i], al0[i], accl[0]);
            if constexpr (aread) al0[j] = SINGLE ? half8{} : lds_read16_untracked<AO + LO>(vau);
            gap(ic<4>{});
            __builtin_amdgcn_sched_barrier(0);
            NM_WAIT_OPERANDS(al1[i], bhv[i]);
            accl[1] = nm_mfma_lo<SINGLE>(bhv[i], al1[i], accl[1]);
            if constexpr (aread) al1[j] = SINGLE ? half8{} : lds_read16_untracked<AO + LO + YO>(vau);
            gap(ic<5>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (t == 8) {
                // tap-group end: the producers publish weights / the next tile.  No lgkmcnt wait here - these waves write no
                // LDS the others read before a later barrier, and their operand reads are covered by the counted waits.
                NM_PSTAMP(1 + 2 * (tt / 9));
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                NM_PSTAMP(2 + 2 * (tt / 9));
            }
        });
        if constexpr (LATE_M) {                                     // the next step's tile is complete since the barrier of tap 26
            ah0[0] = lds_read16_untracked<0>(van); al0[0] = SINGLE ? half8{} : lds_read16_untracked<LO>(van);
            ah1[0] = lds_read16_untracked<YO>(van); al1[0] = SINGLE ? half8{} : lds_read16_untracked<LO + YO>(van);
            ah0[1] = lds_read16_untracked<16>(van); al0[1] = SINGLE ? half8{} : lds_read16_untracked<LO + 16>(van);
            ah1[1] = lds_read16_untracked<YO + 16>(van); al1[1] = SINGLE ? half8{} : lds_read16_untracked<LO + YO + 16>(van);
        }
        // the operands of the next step's taps 0, 1 were read one / two gaps ago; the two copies of the stream may keep them in
        // different registers, so they must have landed before the compiler moves them
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ah0[0]), "+v"(al0[0]), "+v"(ah1[0]), "+v"(al1[0]), "+v"(ah0[1]), "+v"(al0[1]), "+v"(ah1[1]), "+v"(al1[1]),
                       "+v"(bhv[0]), "+v"(blv[0]), "+v"(bhv[1]), "+v"(blv[1]) :: "memory");
        };
        if (run_epi) stream(std::true_type{}); else stream(std::false_type{});
        if (run_epi) { pending = false; part_due = true; }
        if (cur.cb == C16 - 1) { epi_take(cur.w); pending = true; }
        NM_PSTAMP(7);
        if (!has_next) break;
        cur = nxt; hb ^= 1;
#ifdef NM_DIAG
        ++step_no;
#endif
    }
    // ---- drain: the last brick's epilogue in the open
    if (part_due) epi_part();
    lds_barrier();
    static_for<8>([&](auto E) { epi_store(E); });
    static_for<16>([&](auto R) { epi_sum_local(R); });
    static_for<160>([&](auto I) { epi_sum_lanes(I); });
    static_for<16>([&](auto R) { epi_red(R); });
    lds_barrier();
    epi_part();
}

// ---- conv_f16p2: the producer / consumer kernel with 64 output channels per step -------------------------------------
// Same machine mapping as conv_f16p (one persistent 512-thread workgroup per CU, MFMA waves 0-3 + producer waves 4-7, halo
// tile double buffered, input two steps ahead), for the layers with Cout % 64 == 0.  A step is 27 taps x 12 MFMAs per wave:
// the producers' work per step (one tile, 36 KB of weights per tap group) is the same as for 32 channels while the MFMA
// waves have twice the time, so the producers stop being the bottleneck.  Differences forced by the budget:
//   * registers: 128 accumulator registers leave no room for a deferred epilogue - the MFMA waves store a finished brick
//     themselves (16-byte stores from transposed accumulators, DPP partial sums, one extra workgroup barrier per brick);
//   * LDS: two weight buffers (2 x 36 KB) instead of three, so the B operands of a tap group's first tap are read after
//     the barrier that publishes them; operands one tap (12 MFMAs) ahead, double buffered.
//   LDS: halo [2][hi h0 | hi h1 | lo h0 | lo h1][600] x 16 B, weights [2][9 taps][4 planes][64] x 16 B, GroupNorm scratch, bias.
// IO: bit 0 = bfloat16 input (a producer piece is then an 8-byte load of the same four channels), bit 1 = bfloat16 output
// WEMU (diagnostic build only, tools/diag_winograd_emu.py; WRONG RESULTS): the resource profile of a 1-D Winograd F(2,3) form of this
// kernel along z - the one fast-convolution form whose accumulators (4 positions x 32 tile rows x 64 channels = 128 registers in the
// single-accumulator form) and transformed tile (8 planes for 6) fit this machine mapping - without its arithmetic: bit 0 = a step is
// 4 position groups x 9 (ky, kx) taps x 6 MFMAs (216 for 324), one A tile and the four B reads per tap, FOUR weight groups and four
// barriers per step; bit 1 = the producers also convert 13 pieces for 10 (8 transformed planes for 6 raw ones; the extra loads and the
// transform's adds are left out).  An upper bound of what such a kernel could reach, DESIGN 4 "Why there is no Winograd kernel".
// DEFER (round 6; NM355_P2_DEFER=1, NOT the default - built, correct, slower): in-kernel stamps (tools/diag_f16p2_steps.py) put the
// exposed epilogue at 7 800 of a 64-channel brick's 58 000 cycles (24 % at Cin = 32) - 128 values per lane to combine, store, square,
// sum over 32 lanes (320 DPP adds) and park, all on the vector ALU with the matrix pipe idle.  With ONE accumulator per tile
// (conv_up2c_x16's identity x w 2^11 = x_hi (2^11 w_hi) + x_hi w_lo' + x_lo' w_hi, w_hi scaled in place behind its unscaled uses) the
// accumulators are 64 registers, the finished brick's outputs move to 64 others (one v_pk_fma per pair with the bias, exposed: ~700
// cycles) and conv_f16p's gap-interleaved epilogue runs inside the NEXT brick's first step (second copy of the tap stream, zero C operand
// for its first MFMAs, no extra barrier per brick).  Same results as the two-accumulator form to the op tests' 2e-5.  Measured
// (profiles/r06_p2_defer_ab.txt): 64 -> 64 @32^3 1.31-1.33 -> 1.36-1.43 ms, 128 -> 128 @16^3 0.63 -> 0.68: the interleaved pieces are only 40 %
// hidden (the first step of a brick grows by 4 650 cycles for 7 400 saved: the MFMA wave issues in order, its stores and DPP adds
// share the SIMD's issue with a producer wave), the single-accumulator stream itself is 1-2 % slower per tap group, and at 256
// registers the brick decode's hoisted division constants spill (a scratch reload per step).  What would be left after moving the
// decode to the producers (an LDS ring of brick origins): <= 5 % of this kernel.  The exposed epilogue got the cheap part instead:
// its arithmetic on value pairs (v_pk_*), 8 900 -> 8 160 stamped cycles.
template <bool UP2, bool SINGLE = false, int IO = 0, int WEMU = 0, bool DEFER = false>
__global__ __launch_bounds__(512, 1) void conv_f16p2_kernel(ConvParams p) {
    static_assert(!DEFER || (!SINGLE && WEMU == 0), "the deferred epilogue exists for the three-product form");
    constexpr bool IH = (IO & 1) != 0, OH = (IO & 2) != 0;
    constexpr unsigned EB = IH ? 2u : 4u;                           // bytes per input element
    constexpr int HV = 600, ZP = 100, HX = 10;
    constexpr int GB = 9 * 4 * 64;                                  // half8 slots of one weight group
    constexpr int NP = 10;                                          // 16-byte input pieces per producer thread and step (2400 / 256)
    extern __shared__ f32x4 lds[];
    half8* ldh = reinterpret_cast<half8*>(lds);                     // [2][4][HV]
    half8* ldb = ldh + 8 * HV;                                      // [2][GB]
    float* red = reinterpret_cast<float*>(ldb + 2 * GB);            // [4 waves][64 channels][2]
    float* lbias = red + 512;                                       // [Cout] bias (zeros without one), written once by the producers
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int C16 = p.Cin >> 4;
    const int ncg = p.Cout >> 6;
    const int nbx = p.OW >> 3, nby = p.OH >> 3, nbz = p.OD >> 2, nbr = nbx * nby * nbz;
    const int total = p.N * nbr * ncg;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int id_first = (int)blockIdx.x * per, id_last = min(total, id_first + per);
    if (id_first >= id_last) return;

    struct Work { int n, oz0, oy0, ox0, cg; };
    const bool tiled = p.st_x != 0;                                 // XCD super-tile order (one cout group only), see super_tile_item
    auto decode = [&](int id) {
        Work w;
        if (tiled) {
            const BrickPos bp = super_tile_item(p, (int)blockIdx.x, id - id_first, per, nbz, nby, nbx);
            w.cg = 0; w.n = bp.n; w.ox0 = bp.bx << 3; w.oy0 = bp.by << 3; w.oz0 = bp.bz << 2;
            return w;
        }
        w.cg = id % ncg; const int bid = id / ncg;
        w.n = bid / nbr; const int br = bid % nbr;
        w.ox0 = (br % nbx) << 3; w.oy0 = ((br / nbx) % nby) << 3; w.oz0 = (br / (nbx * nby)) << 2;
        return w;
    };
    auto next_work = [&](Work w, int id) {
        if (tiled) return decode(id);
        if (++w.cg == ncg) {
            w.cg = 0;
            if ((w.ox0 += 8) == p.OW) { w.ox0 = 0; if ((w.oy0 += 8) == p.OH) { w.oy0 = 0; if ((w.oz0 += 4) == p.OD) { w.oz0 = 0; ++w.n; } } }
        }
        return w;
    };
    struct Step { Work w; int id, cb; };
    auto advance = [&](Step& st) {                                  // false (and st unchanged) on the block's last step
        if (st.cb + 1 < C16) { ++st.cb; return true; }
        if (st.id + 1 >= id_last) return false;
        st.cb = 0; ++st.id; st.w = next_work(st.w, st.id);
        return true;
    };

#ifdef NM_DIAG
    int step_no = 0;                                                // (NM_PSTAMP: tools/diag_f16p2_steps.py)
#endif
    if (wave >= 4) {
        // =========================== producer waves ===========================================================
        const int pt = tid - 256, pw = wave - 4;
        const half8* __restrict__ w8 = reinterpret_cast<const half8*>(p.w);
        const size_t plane = (size_t)p.Co_pad;
        const int quad = pt & 3;
        int pc_slot[NP], pc_rel[NP], pc_pos[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int v = (pt >> 2) + 64 * k, vc = min(v, HV - 1);
            const int hz = vc / 100, r = vc % 100, hy = r / 10, hx = r % 10;
            pc_slot[k] = (quad >> 1) * HV + hz * ZP + hy * HX + hx;
            pc_rel[k] = ((hz * p.IH + hy) * p.IW + hx) * p.Cin + 4 * quad;
            pc_pos[k] = (v < HV) ? (hz | (hy << 8) | (hx << 16)) : -1;
        }
        using RawT = std::conditional_t<IH, nm_u32x2, f32x4>;
        RawT raw[NP]; f32x4 sc, sh, wreg[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};       // (SINGLE uses five of them; all nine are operands of the counted waits)
        const bool has_affine = p.in_scale != nullptr;
        const unsigned rel111 = (unsigned)(((p.IH + 1) * p.IW + 1) * p.Cin) * EB;   // halo voxel (1,1,1): always inside
        auto inside_mask = [&](const Work& w) {
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int hz = pc_pos[k] & 0xff, hy = (pc_pos[k] >> 8) & 0xff, hx = pc_pos[k] >> 16;
                const bool in = pc_pos[k] >= 0 && (unsigned)(w.oz0 - 1 + hz) < (unsigned)p.ID && (unsigned)(w.oy0 - 1 + hy) < (unsigned)p.IH &&
                                (unsigned)(w.ox0 - 1 + hx) < (unsigned)p.IW;
                m |= (in ? 1u : 0u) << k;
            }
            return m;
        };
        auto tile_base = [&](const Work& w, int cb) {
            return nm_eptr(p.in, (size_t)(((((long long)w.n * p.ID + (w.oz0 - 1)) * p.IH + (w.oy0 - 1)) * p.IW + (w.ox0 - 1)) * (long long)p.Cin + cb * 16), IH);
        };
        auto load_piece = [&](const float* base, unsigned mask, auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (IH) raw[k] = load8_untracked(base, ((mask >> k) & 1) ? (unsigned)pc_rel[k] * EB : rel111);
            else raw[k] = load16_untracked(base, ((mask >> k) & 1) ? (unsigned)pc_rel[k] * EB : rel111);
        };
        auto load_affine = [&](const Work& w, int cb) {             // always two loads: the waits below count instructions
            const float* ps = has_affine ? p.in_scale + (size_t)w.n * p.Cin + cb * 16 : p.in;
            const float* ph = has_affine ? p.in_shift + (size_t)w.n * p.Cin + cb * 16 : p.in;
            sc = load16_untracked(ps, 16u * quad);
            sh = load16_untracked(ph, 16u * quad);
        };
#define NM_PRODUCER2_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" \
            : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]), \
              "+v"(raw[8]), "+v"(raw[9]), "+v"(sc), "+v"(sh), "+v"(wreg[0]), "+v"(wreg[1]), "+v"(wreg[2]), "+v"(wreg[3]), "+v"(wreg[4]), \
              "+v"(wreg[5]), "+v"(wreg[6]), "+v"(wreg[7]), "+v"(wreg[8]) :: "memory")
        auto convert = [&](int buf, unsigned mask, auto K) {        // activate, split, write piece k into halo buffer buf
            constexpr int k = decltype(K)::value;
            half4v hi4, lo4;
            const float keep = ((mask >> k) & 1) ? 1.f : 0.f;       // padding is zero AFTER the activation
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                float v0, v1;
                if constexpr (IH) { const unsigned u = raw[k][e >> 1]; v0 = nm_bf_lo(u); v1 = nm_bf_hi(u); }
                else { v0 = raw[k][e]; v1 = raw[k][e + 1]; }
                if (has_affine) { v0 = __builtin_fmaf(v0, sc[e], sh[e]); v1 = __builtin_fmaf(v1, sc[e + 1], sh[e + 1]); }
                v0 = fmaxf(v0 * keep, (v0 * keep) * p.in_slope); v1 = fmaxf(v1 * keep, (v1 * keep) * p.in_slope);
                half2v hv = __builtin_convertvector(f32x2{v0, v1}, half2v);
                asm volatile("" : "+v"(hv));
                const float t0 = v0 * NM_SPLIT_SCALE, t1 = v1 * NM_SPLIT_SCALE;
                hi4[e] = hv[0]; hi4[e + 1] = hv[1];
                if constexpr (!SINGLE) {                            // (conv mode 3 has no use for the lo halves)
                    lo4[e] = (_Float16)__builtin_fmaf((float)hv[0], -NM_SPLIT_SCALE, t0);
                    lo4[e + 1] = (_Float16)__builtin_fmaf((float)hv[1], -NM_SPLIT_SCALE, t1);
                }
            }
            if (pc_pos[k] >= 0) {
                half4v* dst = reinterpret_cast<half4v*>(ldh + buf * 4 * HV + pc_slot[k]) + (quad & 1);
                dst[0] = hi4;
                if constexpr (!SINGLE) dst[4 * HV] = lo4;
            }
        };
        // tap group g of (cout group cg, chunk cb): 36 one-KiB wave pieces (tap t, plane r, 64 channels), nine per producer wave.
        // SINGLE (conv modes 3 / 4): the MFMA waves never read the lo planes r = 2, 3 - only the 18 hi pieces are copied, five per wave
        // (the last two of the 20 slots repeat piece 17: identical bytes to the same place).  The weight copies are three quarters of a
        // producer wave's issue slots, and in the one-product modes the producers, not the MFMA pipe, bound the kernel.
        constexpr int NW = SINGLE ? 5 : 9;
        unsigned w_src[NW], w_dst[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int j = SINGLE ? min(pw + 4 * i, 17) : pw + 4 * i;
            const int t = SINGLE ? j >> 1 : j >> 2, r = SINGLE ? j & 1 : j & 3;
            w_src[i] = (unsigned)(((size_t)t * C16 * 4 + r) * plane + lane) * 16u;
            w_dst[i] = (unsigned)(t * 256 + r * 64 + lane) * 16u;
        }
        // (wdma: the same nine 1-KiB pieces by LDS-DMA straight into weight buffer `buf` - no registers, no ds_write; the counted waits
        // are the same, a DMA instruction counts on vmcnt like the load it replaces.  The target buffer is free at issue time: it was
        // read during the previous tap group, which ended at the barrier just passed.)
        const bool wdma = p.wdma != 0;
        auto load_b_group = [&](int cg, int cb, int g, int buf) {
            const float* base = reinterpret_cast<const float*>(w8 + ((size_t)(9 * g) * C16 * 4 + (size_t)cb * 4) * plane + cg * 64);
            if (wdma) {
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const int j = SINGLE ? min(pw + 4 * i, 17) : pw + 4 * i;
                    const int t = SINGLE ? j >> 1 : j >> 2, r = SINGLE ? j & 1 : j & 3;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + w_src[i]),
                                                     (__attribute__((address_space(3))) void*)(ldb + buf * GB + t * 256 + r * 64), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NW; ++i) wreg[i] = load16_untracked(base, w_src[i]);
            }
        };
        auto store_b_group = [&](int buf) {
            if (wdma) return;
            char* lbase = reinterpret_cast<char*>(ldb + buf * GB);
#pragma unroll
            for (int i = 0; i < NW; ++i) *reinterpret_cast<f32x4*>(lbase + w_dst[i]) = wreg[i];
        };

        Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
        int hb = 0, gpar = 0;                                       // halo buffer of the step the MFMA waves run; weight buffer of its tap group
        Step s1 = cur; bool has1 = advance(s1);
        Step s2 = s1;  bool has2 = has1 && advance(s2);
        unsigned m_cvt = inside_mask(cur.w), m_ld = m_cvt;
        for (int c = pt; c < p.Cout; c += 256) lbias[c] = p.bias ? p.bias[c] : 0.f;
        // prologue: first tile + first weight group in the open, loads of the second tile in flight
        load_b_group(cur.w.cg, 0, 0, 0);
        load_affine(cur.w, 0);
        { const float* b = tile_base(cur.w, 0); static_for<NP>([&](auto K) { load_piece(b, m_cvt, K); }); }
        NM_PRODUCER2_WAIT(0);
        store_b_group(0);
        static_for<NP>([&](auto K) { convert(0, m_cvt, K); });
        m_cvt = inside_mask(s1.w);
        load_affine(s1.w, s1.cb);
        { const float* b = tile_base(s1.w, s1.cb); static_for<NP>([&](auto K) { load_piece(b, m_cvt, K); }); }
        lds_barrier();
        for (;;) {
            if (s2.w.n != s1.w.n || s2.w.oz0 != s1.w.oz0 || s2.w.oy0 != s1.w.oy0 || s2.w.ox0 != s1.w.ox0) m_ld = inside_mask(s2.w);
            else m_ld = m_cvt;
            const float* b2 = tile_base(s2.w, s2.cb);
            // The ten pieces are dealt 3 / 5 / 2 to the three tap groups (round 3; they were 4 / 6 / 0).  In-kernel stamps
            // (tools/diag_f16p2_steps.py) showed the producers, not the MFMA waves, arriving last at the first two barriers of a step
            // (MFMA waves waiting 734 + 1011 of a 15 272-cycle step) with 1 100 cycles to spare at the third: a step's 39 global loads
            // and 47 LDS stores cost these waves ~100 cycles of issue each beside the MFMA waves' streams.  The tile of step s + 1 is
            // therefore complete at the step's LAST barrier, and the MFMA waves fetch a step's first operands after that barrier.
            // tap group 0: weights of this step's group 1 (into the buffer group 2 of the last step was read from); pieces 0-2.
            // Everything older than the 9 weight loads has landed after the first wait.
            NM_PSTAMP(0);
            load_b_group(cur.w.cg, cur.cb, 1, gpar ^ 1);
            if constexpr (SINGLE) { NM_PRODUCER2_WAIT(5); } else { NM_PRODUCER2_WAIT(9); }      // (NW weight loads are the youngest)
            NM_PSTAMP(1);
            static_for<3>([&](auto K) { convert(hb ^ 1, m_cvt, K); if constexpr ((WEMU & 2) != 0 && decltype(K)::value == 0) convert(hb ^ 1, m_cvt, K); load_piece(b2, m_ld, K); });
            NM_PRODUCER2_WAIT(3);                                   // the weight loads (older than the 3 new piece loads)
            store_b_group(gpar ^ 1);
            NM_PSTAMP(2);
            lds_barrier(); gpar ^= 1;
            NM_PSTAMP(3);
            // tap group 1: weights of this step's group 2; pieces 3-7
            load_b_group(cur.w.cg, cur.cb, 2, gpar ^ 1);
            if constexpr (WEMU != 0) {                              // (emulation: pieces 3-5 here, 6-7 behind a fourth weight group)
                static_for<3>([&](auto K) { convert(hb ^ 1, m_cvt, ic<decltype(K)::value + 3>{}); if constexpr ((WEMU & 2) != 0 && decltype(K)::value == 0) convert(hb ^ 1, m_cvt, ic<3>{});
                                            load_piece(b2, m_ld, ic<decltype(K)::value + 3>{}); });
                NM_PRODUCER2_WAIT(3);
                store_b_group(gpar ^ 1);
                lds_barrier(); gpar ^= 1;
                load_b_group(cur.w.cg, cur.cb, 1, gpar ^ 1);
                static_for<2>([&](auto K) { convert(hb ^ 1, m_cvt, ic<decltype(K)::value + 6>{}); if constexpr ((WEMU & 2) != 0 && decltype(K)::value == 0) convert(hb ^ 1, m_cvt, ic<6>{});
                                            load_piece(b2, m_ld, ic<decltype(K)::value + 6>{}); });
                NM_PRODUCER2_WAIT(2);
                store_b_group(gpar ^ 1);
                lds_barrier(); gpar ^= 1;
            } else {
            static_for<5>([&](auto K) { convert(hb ^ 1, m_cvt, ic<decltype(K)::value + 3>{}); load_piece(b2, m_ld, ic<decltype(K)::value + 3>{}); });
            NM_PSTAMP(4);
            NM_PRODUCER2_WAIT(5);                                   // the weight loads: 5 piece loads are younger
            store_b_group(gpar ^ 1);
            NM_PSTAMP(5);
            lds_barrier(); gpar ^= 1;
            NM_PSTAMP(6);
            }
            // tap group 2: weights of the next step's group 0; pieces 8-9; the halo tile is complete at this barrier
            load_b_group(s1.w.cg, s1.cb, 0, gpar ^ 1);
            static_for<2>([&](auto K) { convert(hb ^ 1, m_cvt, ic<decltype(K)::value + 8>{}); load_piece(b2, m_ld, ic<decltype(K)::value + 8>{}); });
            load_affine(s2.w, s2.cb);                               // (sc / sh are free: piece 9 was their last user)
            NM_PRODUCER2_WAIT(0);
            store_b_group(gpar ^ 1);
            NM_PSTAMP(7);
            lds_barrier(); gpar ^= 1;
            NM_PSTAMP(8);
#ifdef NM_DIAG
            ++step_no;
#endif
            if constexpr (!DEFER) { if (cur.cb == C16 - 1) lds_barrier(); }      // the MFMA waves' epilogue barrier of a finished brick
            if (!has1) break;
            cur = s1; s1 = s2; has1 = has2; has2 = has2 && advance(s2);
            m_cvt = m_ld; hb ^= 1;
        }
        NM_PRODUCER2_WAIT(0);                                       // loads still in flight for a step that does not exist
        return;
    }

    // =============================== MFMA waves ===============================================================
    __builtin_amdgcn_s_setprio(3);
    int arow0;
    {
        const int c = l31 >> 2;
        const int x = (((0x96 >> c) & 1) << 2) + (l31 & 3), z = c >> 1;
        arow0 = z * ZP + (2 * wave) * HX + x;
    }
    const int vx = ((((0x96 >> (l31 >> 2)) & 1) << 2) + (l31 & 3)), vz = l31 >> 3;
    if constexpr (DEFER) {
        // ---- one accumulator per tile, the finished brick's epilogue inside the next brick's first step ----------------------------
        constexpr int LO = 2 * HV * 16, YO = HX * 16;
        f32x16 acc[2][2], outv[2][2];                               // outv: the pending brick's outputs, then its partial sums in place
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; outv[mt][nt][r] = 0.f; }
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        size_t esb = 0;                                             // pending brick: element index of (its first voxel, cout group) - wave-uniform
        const unsigned lane_off = (unsigned)(((vz * p.OH + 2 * wave) * p.OW + vx) * p.Cout + 4 * h);   // this lane's voxel and channel quad inside a brick
        size_t epart = 0, ppart = 0;                      // pending brick: its outputs' place / its partial sums' place; ppart: of the sums red[] holds
                                            // element index of (pending brick, this lane's voxel, mt = 0, nt = 0, k4 = 0)
        const size_t erow = (size_t)p.OW * p.Cout;                  // mt = 0 -> 1
        auto epi_store = [&](auto E) {                              // store e = (2 nt + mt) * 4 + k4 of the 16 per lane
            constexpr int e = decltype(E)::value, nt = e >> 3, mt = (e >> 2) & 1, k4 = e & 3;
            unsigned lo = lane_off;
            asm volatile("" : "+v"(lo));                            // (the 16 addresses are loop invariants: hoisted, they are spilled and reloaded in the gaps)
            nm_st4<OH>(p.out, esb + (mt ? erow : (size_t)0) + (size_t)(lo + (unsigned)(nt * 32 + 8 * k4)),
                       f32x4{outv[mt][nt][4 * k4], outv[mt][nt][4 * k4 + 1], outv[mt][nt][4 * k4 + 2], outv[mt][nt][4 * k4 + 3]});
        };
        auto epi_sum_local = [&](auto R) {                          // pair R of tile column nt: sums / sums of squares over the two row tiles, in place
            constexpr int nt = decltype(R)::value >> 3, r = 2 * (decltype(R)::value & 7);
            const f32x2 a = f32x2{outv[0][nt][r], outv[0][nt][r + 1]}, b = f32x2{outv[1][nt][r], outv[1][nt][r + 1]};
            const f32x2 s = a + b, q = a * a + b * b;
            outv[0][nt][r] = s[0]; outv[0][nt][r + 1] = s[1]; outv[1][nt][r] = q[0]; outv[1][nt][r + 1] = q[1];
        };
        auto epi_sum_lanes = [&](auto I) {                          // DPP step (i >> 6) of value (i & 63) = 32 m + 16 nt + r
            constexpr int i = decltype(I)::value, st = i >> 6, v = i & 63, m = v >> 5, nt = (v >> 4) & 1, r = v & 15;
            constexpr int ctrl = st == 0 ? 0xB1 : st == 1 ? 0x4E : st == 2 ? 0x141 : st == 3 ? 0x140 : 0x142;
            constexpr int rows = st == 4 ? 0xa : 0xf;
            const float x = outv[m][nt][r];
            outv[m][nt][r] = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rows, 0xf, true));
        };
        auto epi_red = [&](auto R) {                                // lanes 16 / 48 park (sum, sum of squares) of channel c(nt, r, h)
            constexpr int nt = decltype(R)::value >> 4, r = decltype(R)::value & 15;
            int ln = lane;
            asm volatile("" : "+v"(ln));                            // (addresses made here: as loop invariants they are hoisted, and then spilled)
            if (p.part && (ln & 31) == 16) *reinterpret_cast<f32x2*>(red + (wave * 64 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5)) * 2) = f32x2{outv[0][nt][r], outv[1][nt][r]};
        };
        auto epi_part = [&]() {                                     // behind a barrier: 64 threads sum the four wave partials
            int tq = tid;
            asm volatile("" : "+v"(tq));
            if (p.part && tq < 64) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { a += red[(q * 64 + tq) * 2]; b += red[(q * 64 + tq) * 2 + 1]; }
                float* dp = p.part + ppart + tq * 2;
                dp[0] = a; dp[1] = b;
            }
        };
        auto epi_take = [&](const Work& w) {                        // outputs of the brick just accumulated (exposed: 64 packed FMAs)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const float* lb = lbias + w.cg * 64 + 4 * (ln >> 5);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(lb + nt * 32 + 8 * k4);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const f32x2 a2 = f32x2{acc[mt][nt][4 * k4 + e], acc[mt][nt][4 * k4 + e + 1]};
                            const f32x2 v2 = __builtin_elementwise_fma(a2, f32x2{1.0f / NM_SPLIT_SCALE, 1.0f / NM_SPLIT_SCALE}, f32x2{b4[e], b4[e + 1]});
                            outv[mt][nt][4 * k4 + e] = v2[0]; outv[mt][nt][4 * k4 + e + 1] = v2[1];
                        }
                }
            epart = (((size_t)w.n * nbr + ((w.oz0 >> 2) * nby + (w.oy0 >> 3)) * nbx + (w.ox0 >> 3)) * p.Cout + w.cg * 64) * 2;
            esb = ((((size_t)w.n * p.OD + w.oz0) * p.OH + w.oy0) * p.OW + w.ox0) * (size_t)p.Cout + w.cg * 64;
        };

        Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
        int hb = 0, gpar = 0;
        lds_barrier();                                              // the producers' prologue
        const unsigned vb0 = (unsigned)(size_t)(ldb + h * 64 + l31);
        half8 ah0[2], al0[2], ah1[2], al1[2], bh0[2], bh1[2], bl0[2], bl1[2];
        {
            const unsigned va = (unsigned)(size_t)(ldh + h * HV + arow0);
            ah0[0] = lds_read16_untracked<0>(va); al0[0] = lds_read16_untracked<LO>(va);
            ah1[0] = lds_read16_untracked<YO>(va); al1[0] = lds_read16_untracked<LO + YO>(va);
        }
        bool pending = false, part_due = false;
        for (;;) {
            Step nxt = cur;
            const bool has_next = advance(nxt);
            if (part_due) { epi_part(); part_due = false; }
            const bool run_epi = pending;
            const unsigned va = (unsigned)(size_t)(ldh + hb * 4 * HV + h * HV + arow0);
            const unsigned van = (unsigned)(size_t)(ldh + (hb ^ 1) * 4 * HV + h * HV + arow0);
            const unsigned vbe = vb0 + (unsigned)(gpar * GB * 16), vbo = vb0 + (unsigned)((gpar ^ 1) * GB * 16);
            NM_PSTAMP(0);
            auto stream = [&](auto EPI) {
                constexpr bool with_epi = decltype(EPI)::value;
                static_for<27>([&](auto TT) {
                    constexpr int tt = decltype(TT)::value, g = tt / 9, t = tt % 9, i = tt & 1, j = i ^ 1;
                    const unsigned vb = (g & 1) ? vbo : vbe;
                    constexpr int BT = t * 256 * 16;
                    if constexpr (t == 0) {
                        bh0[i] = lds_read16_untracked<BT>(vb); bh1[i] = lds_read16_untracked<BT + 32 * 16>(vb);
                        bl0[i] = lds_read16_untracked<BT + 128 * 16>(vb); bl1[i] = lds_read16_untracked<BT + 160 * 16>(vb);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(ah0[i]), "+v"(al0[i]), "+v"(ah1[i]), "+v"(al1[i]), "+v"(bh0[i]), "+v"(bh1[i]), "+v"(bl0[i]), "+v"(bl1[i]) :: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    constexpr int u = tt + 1, ug = (u < 27) ? u / 9 : 0, ut = (u < 27) ? u % 9 : 0;
                    constexpr int AO = (ug * ZP + (ut / 3) * HX + (ut % 3)) * 16;
                    constexpr bool anext = u < 27, bnext = t < 8;
                    constexpr int BN = (t + 1) * 256 * 16;
                    constexpr bool z = with_epi && tt == 0;         // a new brick: C = 0 instead of cleared accumulators
                    auto gap = [&](auto SUB) {
                        constexpr int q = tt * 12 + decltype(SUB)::value;
                        if constexpr (with_epi) {
                            if constexpr (q >= 4 && q < 36 && (q & 1) == 0) epi_store(ic<((q - 4) >> 1) & 15>{});
                            if constexpr (q >= 36 && q < 52) epi_sum_local(ic<(q - 36) & 15>{});
                            if constexpr (q >= 52 && q < 212) { epi_sum_lanes(ic<(2 * (q - 52)) % 320>{}); epi_sum_lanes(ic<(2 * (q - 52) + 1) % 320>{}); }
                            if constexpr (q >= 214 && q < 246) epi_red(ic<(q - 214) & 31>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
#define NM_MFMA3(ACC, B, A, Z) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, A, (Z) ? zero16 : ACC, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
                    // the cross products first: 2^11 w_hi of this tap is made in their gaps
                    // the products with the unscaled w_hi first; it becomes 2^11 w_hi IN PLACE behind them (exact for |w| < 32; larger
                    // weights overflow into the range guard like any fp16 overflow), in the gaps of the cross products
                    NM_MFMA3(acc[0][0], bh0[i], al0[i], z); if constexpr (anext) ah0[j] = lds_read16_untracked<AO>(va); gap(ic<0>{});
                    NM_MFMA3(acc[0][1], bh1[i], al0[i], z); if constexpr (anext) al0[j] = lds_read16_untracked<AO + LO>(va); gap(ic<1>{});
                    NM_MFMA3(acc[1][0], bh0[i], al1[i], z); if constexpr (anext) ah1[j] = lds_read16_untracked<AO + YO>(va); gap(ic<2>{});
                    NM_MFMA3(acc[1][1], bh1[i], al1[i], z); if constexpr (anext) al1[j] = lds_read16_untracked<AO + LO + YO>(va); gap(ic<3>{});
                    NM_MFMA3(acc[0][0], bl0[i], ah0[i], false); if constexpr (bnext) bh0[j] = lds_read16_untracked<BN>(vb); gap(ic<4>{});
                    NM_MFMA3(acc[0][1], bl1[i], ah0[i], false); if constexpr (bnext) bh1[j] = lds_read16_untracked<BN + 32 * 16>(vb); gap(ic<5>{});
                    NM_MFMA3(acc[1][0], bl0[i], ah1[i], false); if constexpr (bnext) bl0[j] = lds_read16_untracked<BN + 128 * 16>(vb); bh0[i] = bh0[i] * (_Float16)NM_SPLIT_SCALE; gap(ic<6>{});
                    NM_MFMA3(acc[1][1], bl1[i], ah1[i], false); if constexpr (bnext) bl1[j] = lds_read16_untracked<BN + 160 * 16>(vb); bh1[i] = bh1[i] * (_Float16)NM_SPLIT_SCALE; gap(ic<7>{});
                    NM_MFMA3(acc[0][0], bh0[i], ah0[i], false); gap(ic<8>{});
                    NM_MFMA3(acc[0][1], bh1[i], ah0[i], false); gap(ic<9>{});
                    NM_MFMA3(acc[1][0], bh0[i], ah1[i], false); gap(ic<10>{});
                    NM_MFMA3(acc[1][1], bh1[i], ah1[i], false); gap(ic<11>{});
                    if constexpr (t == 8) {
                        NM_PSTAMP(1 + 2 * g);
                        asm volatile("s_barrier" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        NM_PSTAMP(2 + 2 * g);
                    }
                });
            };
            if (run_epi) stream(std::true_type{}); else stream(std::false_type{});
            gpar ^= 1;
            if (run_epi) { pending = false; part_due = true; ppart = epart; }
            if (cur.cb == C16 - 1) { NM_PSTAMP(8); epi_take(cur.w); pending = true; NM_PSTAMP(13); }
            // the next step's tile is complete since the barrier that ended tap group 2: its first tap's A operands (slot 0)
            if (has_next) {
                ah0[0] = lds_read16_untracked<0>(van); al0[0] = lds_read16_untracked<LO>(van);
                ah1[0] = lds_read16_untracked<YO>(van); al1[0] = lds_read16_untracked<LO + YO>(van);
            }
            NM_PSTAMP(7);
#ifdef NM_DIAG
            ++step_no;
#endif
            if (!has_next) break;
            cur = nxt; hb ^= 1;
        }
        // ---- drain: the last brick's epilogue in the open (the producers have left)
        if (part_due) epi_part();
        lds_barrier();
        static_for<16>([&](auto E) { epi_store(E); });
        static_for<16>([&](auto R) { epi_sum_local(R); });
        static_for<320>([&](auto I) { epi_sum_lanes(I); });
        static_for<32>([&](auto R) { epi_red(R); });
        lds_barrier();
        ppart = epart;
        epi_part();
        return;
    }
    f32x16 acc[2][2], accl[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; accl[mt][nt][r] = 0.f; }

    Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
    int hb = 0, gpar = 0;
    lds_barrier();                                                  // the producers' prologue

    constexpr int LO = 2 * HV * 16;                                 // hi -> lo plane of a halo tile, bytes
    constexpr int YO = HX * 16;                                     // brick row 2 wave -> 2 wave + 1
    const unsigned vb0 = (unsigned)(size_t)(ldb + h * 64 + l31);    // weight buffer 0: plane h (hi), channel l31 of N tile 0
    half8 ah0[2], al0[2], ah1[2], al1[2], bh0[2], bh1[2], bl0[2], bl1[2];      // operands, double buffered (index = tap & 1)
    {
        const unsigned va = (unsigned)(size_t)(ldh + h * HV + arow0);
        ah0[0] = lds_read16_untracked<0>(va); al0[0] = SINGLE ? half8{} : lds_read16_untracked<LO>(va);
        ah1[0] = lds_read16_untracked<YO>(va); al1[0] = SINGLE ? half8{} : lds_read16_untracked<LO + YO>(va);
    }
    for (;;) {
        Step nxt = cur;
        const bool has_next = advance(nxt);
        const unsigned va = (unsigned)(size_t)(ldh + hb * 4 * HV + h * HV + arow0);          // this step's halo tile
        const unsigned van = (unsigned)(size_t)(ldh + (hb ^ 1) * 4 * HV + h * HV + arow0);   // the next step's
        const unsigned vbe = vb0 + (unsigned)(gpar * GB * 16), vbo = vb0 + (unsigned)((gpar ^ 1) * GB * 16);   // buffers of groups 0/2 and 1
        NM_PSTAMP(0);
        constexpr int NTAP = WEMU != 0 ? 36 : 27;
        static_for<NTAP>([&](auto TT) {
            constexpr int tt = decltype(TT)::value, g = tt / 9, t = tt % 9, i = tt & 1, j = i ^ 1;
            const unsigned vb = (g & 1) ? vbo : vbe;
            constexpr int BT = t * 256 * 16;                        // byte offset of this tap in its weight buffer
            if constexpr (t == 0) {                                 // first tap of a group: its weights were published by the barrier
                bh0[i] = lds_read16_untracked<BT>(vb); bh1[i] = lds_read16_untracked<BT + 32 * 16>(vb);
                bl0[i] = SINGLE ? half8{} : lds_read16_untracked<BT + 128 * 16>(vb); bl1[i] = SINGLE ? half8{} : lds_read16_untracked<BT + 160 * 16>(vb);
            }
            // every operand of this tap has been requested (previous tap / just above): wait for all of them
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ah0[i]), "+v"(al0[i]), "+v"(ah1[i]), "+v"(al1[i]), "+v"(bh0[i]), "+v"(bh1[i]), "+v"(bl0[i]), "+v"(bl1[i]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            constexpr int u = tt + 1, ug = (u < NTAP) ? (u / 9) % 3 : 0, ut = (u < NTAP) ? u % 9 : 0;
            constexpr int AO = (ug * ZP + (ut / 3) * HX + (ut % 3)) * 16;           // A byte offset of tap tt + 1
            constexpr bool anext = u < NTAP;                        // (tap 26 requests nothing: the next tile is complete only at the step's last barrier)
            constexpr bool a1 = WEMU == 0;                          // (emulation: one A tile per tap)
            const unsigned vau = va;
            constexpr bool bnext = t < 8;                           // the next tap's weights are in this group's buffer
            constexpr int BN = (t + 1) * 256 * 16;
#define NM_MFMA2(ACC, B, A) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, A, ACC, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
#define NM_MFMA2L(ACC, B, A) ACC = nm_mfma_lo<SINGLE>(B, A, ACC); __builtin_amdgcn_sched_barrier(0)
            NM_MFMA2(acc[0][0], bh0[i], ah0[i]);  if constexpr (anext) ah0[j] = lds_read16_untracked<AO>(vau);
            NM_MFMA2(acc[0][1], bh1[i], ah0[i]);  if constexpr (anext) al0[j] = SINGLE ? half8{} : lds_read16_untracked<AO + LO>(vau);
            NM_MFMA2L(accl[0][0], bl0[i], ah0[i]); if constexpr (anext && a1) ah1[j] = lds_read16_untracked<AO + YO>(vau); else if constexpr (bnext && !a1) bh0[j] = lds_read16_untracked<BN>(vb);
            NM_MFMA2L(accl[0][1], bl1[i], ah0[i]); if constexpr (anext && a1) al1[j] = SINGLE ? half8{} : lds_read16_untracked<AO + LO + YO>(vau); else if constexpr (bnext && !a1) bh1[j] = lds_read16_untracked<BN + 32 * 16>(vb);
            NM_MFMA2L(accl[0][0], bh0[i], al0[i]); if constexpr (bnext && a1) bh0[j] = lds_read16_untracked<BN>(vb); else if constexpr (bnext && !a1) bl0[j] = SINGLE ? half8{} : lds_read16_untracked<BN + 128 * 16>(vb);
            NM_MFMA2L(accl[0][1], bh1[i], al0[i]); if constexpr (bnext && a1) bh1[j] = lds_read16_untracked<BN + 32 * 16>(vb); else if constexpr (bnext && !a1) bl1[j] = SINGLE ? half8{} : lds_read16_untracked<BN + 160 * 16>(vb);
            if constexpr (a1) {
            NM_MFMA2(acc[1][0], bh0[i], ah1[i]);  if constexpr (bnext) bl0[j] = SINGLE ? half8{} : lds_read16_untracked<BN + 128 * 16>(vb);
            NM_MFMA2(acc[1][1], bh1[i], ah1[i]);  if constexpr (bnext) bl1[j] = SINGLE ? half8{} : lds_read16_untracked<BN + 160 * 16>(vb);
            NM_MFMA2L(accl[1][0], bl0[i], ah1[i]);
            NM_MFMA2L(accl[1][1], bl1[i], ah1[i]);
            NM_MFMA2L(accl[1][0], bh0[i], al1[i]);
            NM_MFMA2L(accl[1][1], bh1[i], al1[i]);
            }
            if constexpr (t == 8) {
                // tap-group end: the producers publish the next group's weights (and, at the second barrier, the next tile)
                NM_PSTAMP(1 + 2 * (g % 3));
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                NM_PSTAMP(2 + 2 * (g % 3));
            }
        });
        if constexpr (WEMU == 0) gpar ^= 1;                         // three groups per step
        // the next step's tile is complete since the barrier that ended tap group 2: request its first tap's A operands now (slot 0;
        // the step's first wait covers them)
        if (has_next) {
            ah0[0] = lds_read16_untracked<0>(van); al0[0] = SINGLE ? half8{} : lds_read16_untracked<LO>(van);
            ah1[0] = lds_read16_untracked<YO>(van); al1[0] = SINGLE ? half8{} : lds_read16_untracked<LO + YO>(van);
        }
        if (cur.cb == C16 - 1) {
            // ---- epilogue of the finished brick (exposed): transposed accumulators -> 16-byte stores, DPP partial sums
            const Work& w = cur.w;
            NM_PSTAMP(8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // (round 6: the epilogue's arithmetic on value PAIRS - v_pk_mul_f32 / v_pk_add_f32 - the same operations per value in
                // the same order at half the vector instructions; the exposed epilogue is VALU-bound, in-kernel stamps: 7 800 of a
                // 64-channel brick's 58 000 cycles)
                float s1[16], s2[16];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const size_t dst = ((((size_t)w.n * p.OD + w.oz0 + vz) * p.OH + w.oy0 + 2 * wave + mt) * p.OW + w.ox0 + vx) * (size_t)p.Cout +
                                       w.cg * 64 + nt * 32 + 4 * h;
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbias + w.cg * 64 + nt * 32 + 8 * k4 + 4 * h);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const f32x2 a2 = f32x2{acc[mt][nt][4 * k4 + e], acc[mt][nt][4 * k4 + e + 1]};
                            const f32x2 l2 = f32x2{accl[mt][nt][4 * k4 + e], accl[mt][nt][4 * k4 + e + 1]};
                            const f32x2 v2 = (a2 + l2 * (1.0f / NM_SPLIT_SCALE)) + f32x2{b4[e], b4[e + 1]};
                            const f32x2 q2 = v2 * v2;
                            v[e] = v2[0]; v[e + 1] = v2[1];
                            if (mt == 0) { s1[4 * k4 + e] = v2[0]; s1[4 * k4 + e + 1] = v2[1]; s2[4 * k4 + e] = q2[0]; s2[4 * k4 + e + 1] = q2[1]; }
                            else {
                                const f32x2 t1 = f32x2{s1[4 * k4 + e], s1[4 * k4 + e + 1]} + v2, t2 = f32x2{s2[4 * k4 + e], s2[4 * k4 + e + 1]} + q2;
                                s1[4 * k4 + e] = t1[0]; s1[4 * k4 + e + 1] = t1[1]; s2[4 * k4 + e] = t2[0]; s2[4 * k4 + e + 1] = t2[1];
                            }
                            acc[mt][nt][4 * k4 + e] = 0.f; acc[mt][nt][4 * k4 + e + 1] = 0.f; accl[mt][nt][4 * k4 + e] = 0.f; accl[mt][nt][4 * k4 + e + 1] = 0.f;
                        }
                        nm_st4<OH>(p.out, dst + 8 * k4, v);
                    }
                }
                if (nt == 0) NM_PSTAMP(9); else NM_PSTAMP(11);
                if (p.part) {
                    dpp_sum32_many(s1); dpp_sum32_many(s2);
                    if (l31 == 16) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            *reinterpret_cast<f32x2*>(red + (wave * 64 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 2) = f32x2{s1[r], s2[r]};
                    }
                }
                if (nt == 0) NM_PSTAMP(10);
            }
            NM_PSTAMP(12);
            lds_barrier();                                          // (matched by the producers)
            NM_PSTAMP(13);
            if (p.part && tid < 64) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { a += red[(q * 64 + tid) * 2]; b += red[(q * 64 + tid) * 2 + 1]; }
                const int br = ((w.oz0 >> 2) * nby + (w.oy0 >> 3)) * nbx + (w.ox0 >> 3);
                float* dp = p.part + (((size_t)w.n * nbr + br) * p.Cout + w.cg * 64 + tid) * 2;
                dp[0] = a; dp[1] = b;
            }
        }
        NM_PSTAMP(7);
#ifdef NM_DIAG
        ++step_no;
#endif
        if (!has_next) break;
        cur = nxt; hb ^= 1;
    }
}

// ---- conv_f16q2: the one-product form (conv modes 3 / 4) of conv_f16p2, re-balanced (round 5) ---------------------------------------
// In the one-product modes conv_f16p2<SINGLE> keeps the three-product kernel's schedule with two thirds of the MFMAs deleted: a tap
// group is 36 MFMAs per wave (1 152 cycles) instead of 108, but the producers still fetch each group's weights one group ahead -
// an L2 round trip plus the register -> LDS copy no longer fits under a group, and three workgroup barriers per step expose it
// (stand-alone 64 -> 64 @32^3: 0.19-0.31 of the f16 peak, the producers' arrival at the barriers is the step).  Here the schedule is
// built for one MFMA per staged operand:
//   * hi-only tiles halve the LDS footprint, so a WHOLE step's weights (27 taps x 16 channels x 64 outputs = 54 KB) are double
//     buffered and fetched one full step ahead of their use; halo tile two steps ahead as before;
//   * ONE workgroup barrier per step instead of three;
//   * bfloat16 input (mode 4): a producer piece is a 16-byte load of EIGHT channels (one ds_write_b128 per piece, 5 pieces per
//     thread and step instead of 10 eight-byte ones);
//   * the MFMA waves keep operands two taps ahead (three register sets, counted lgkmcnt waits): a tap is only four MFMAs.
// Arithmetic per output element (k order: channel chunk outer, tap inner; epilogue) is conv_f16p2<SINGLE>'s: bit-identical results.
//   LDS: halo [2][h0 | h1][600] x 16 B, weights [2][27 taps][h0 | h1][64] x 16 B, GroupNorm scratch, bias: 148 KB + bias.
// DBG (diagnostic instantiations, NM355_Q2_DBG): 1 no MFMAs, 2 producers only keep the barriers, 3 no epilogue stores, 4 no weight copies, 5 no input staging,
// 6 = 2 + no operand reads inside the tap loop, 7 = 2 + no epilogue, 8 input pieces loaded but not converted / stored, 9 converted / stored but not loaded
template <int IO = 0, int DBG = 0>
__global__ __launch_bounds__(512, 1) void conv_f16q2_kernel(ConvParams p) {
    constexpr bool IH = (IO & 1) != 0, OH = (IO & 2) != 0;
    constexpr unsigned EB = IH ? 2u : 4u;                           // bytes per input element
    constexpr int HV = 600, ZP = 100, HX = 10;
    constexpr int WB = 27 * 2 * 64;                                 // half8 slots of one step's weights
    constexpr int NP = IH ? 5 : 10;                                 // 16-byte input pieces per producer thread and step
    constexpr int NCH = IH ? 8 : 4;                                 // channels per piece
    constexpr int NWL = 14;                                         // 1-KiB weight pieces per producer wave and step (54 / 4, the last two slots repeat piece 53)
    constexpr int NA = IH ? 4 : 2;                                  // affine loads per step
    extern __shared__ f32x4 lds[];
    half8* ldh = reinterpret_cast<half8*>(lds);                     // [2][2][HV]
    half8* ldb = ldh + 4 * HV;                                      // [2][WB]
    float* red = reinterpret_cast<float*>(ldb + 2 * WB);            // [4 waves][64 channels][2]
    float* lbias = red + 512;                                       // [Cout]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int C16 = p.Cin >> 4;
    const int ncg = p.Cout >> 6;
    const int nbx = p.OW >> 3, nby = p.OH >> 3, nbz = p.OD >> 2, nbr = nbx * nby * nbz;
    const int total = p.N * nbr * ncg;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int id_first = (int)blockIdx.x * per, id_last = min(total, id_first + per);
    if (id_first >= id_last) return;

    struct Work { int n, oz0, oy0, ox0, cg; };
    const bool tiled = p.st_x != 0;
    auto decode = [&](int id) {
        Work w;
        if (tiled) {
            const BrickPos bp = super_tile_item(p, (int)blockIdx.x, id - id_first, per, nbz, nby, nbx);
            w.cg = 0; w.n = bp.n; w.ox0 = bp.bx << 3; w.oy0 = bp.by << 3; w.oz0 = bp.bz << 2;
            return w;
        }
        w.cg = id % ncg; const int bid = id / ncg;
        w.n = bid / nbr; const int br = bid % nbr;
        w.ox0 = (br % nbx) << 3; w.oy0 = ((br / nbx) % nby) << 3; w.oz0 = (br / (nbx * nby)) << 2;
        return w;
    };
    auto next_work = [&](Work w, int id) {
        if (tiled) return decode(id);
        if (++w.cg == ncg) {
            w.cg = 0;
            if ((w.ox0 += 8) == p.OW) { w.ox0 = 0; if ((w.oy0 += 8) == p.OH) { w.oy0 = 0; if ((w.oz0 += 4) == p.OD) { w.oz0 = 0; ++w.n; } } }
        }
        return w;
    };
    struct Step { Work w; int id, cb; };
    auto advance = [&](Step& st) {                                  // false (and st unchanged) on the block's last step
        if (st.cb + 1 < C16) { ++st.cb; return true; }
        if (st.id + 1 >= id_last) return false;
        st.cb = 0; ++st.id; st.w = next_work(st.w, st.id);
        return true;
    };

    if (wave >= 4) {
        // =========================== producer waves ===========================================================
        const int pt = tid - 256, pw = wave - 4;
        const half8* __restrict__ w8 = reinterpret_cast<const half8*>(p.w);
        const size_t plane = (size_t)p.Co_pad;
        // piece k of this thread: halo voxel v, channel group sub (fp32 input: a quad of the chunk's 16 channels, v = pt / 4 + 64 k;
        // bfloat16 input: one of its two octets, v = pt / 2 + 128 k)
        const int sub = IH ? (pt & 1) : (pt & 3);
        int pc_slot[NP], pc_rel[NP], pc_pos[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int v = IH ? (pt >> 1) + 128 * k : (pt >> 2) + 64 * k, vc = min(v, HV - 1);
            const int hz = vc / 100, r = vc % 100, hy = r / 10, hx = r % 10;
            pc_slot[k] = (IH ? sub : (sub >> 1)) * HV + hz * ZP + hy * HX + hx;
            pc_rel[k] = ((hz * p.IH + hy) * p.IW + hx) * p.Cin + NCH * sub;
            pc_pos[k] = (v < HV) ? (hz | (hy << 8) | (hx << 16)) : -1;
        }
        // two register sets for the input pieces (and their affine): a tile's loads are issued TWO producer iterations before its
        // conversion - one full step of cover.  (With one set the loads of step s + 2 were issued between the conversions of step
        // s + 1 and consumed at the top of the next iteration: half a step of cover for an HBM round trip; loads alone or conversions
        // alone cost the kernel nothing, together +0.15 ms on 64 -> 64 @32^3 - tools/diag_f16q2.py variants 8 / 9.)
        f32x4 raw[2][NP], sc[2][NA / 2], sh[2][NA / 2], wreg[NWL];
#pragma unroll
        for (int i = 0; i < NWL; ++i) wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool has_affine = p.in_scale != nullptr;
        const unsigned rel111 = (unsigned)(((p.IH + 1) * p.IW + 1) * p.Cin) * EB;   // halo voxel (1,1,1): always inside
        auto inside_mask = [&](const Work& w) {
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int hz = pc_pos[k] & 0xff, hy = (pc_pos[k] >> 8) & 0xff, hx = pc_pos[k] >> 16;
                const bool in = pc_pos[k] >= 0 && (unsigned)(w.oz0 - 1 + hz) < (unsigned)p.ID && (unsigned)(w.oy0 - 1 + hy) < (unsigned)p.IH &&
                                (unsigned)(w.ox0 - 1 + hx) < (unsigned)p.IW;
                m |= (in ? 1u : 0u) << k;
            }
            return m;
        };
        auto tile_base = [&](const Work& w, int cb) {
            return nm_eptr(p.in, (size_t)(((((long long)w.n * p.ID + (w.oz0 - 1)) * p.IH + (w.oy0 - 1)) * p.IW + (w.ox0 - 1)) * (long long)p.Cin + cb * 16), IH);
        };
        auto load_piece = [&](const float* base, unsigned mask, auto K, auto SET) {
            constexpr int k = decltype(K)::value, set = decltype(SET)::value;
            raw[set][k] = load16_untracked(base, ((mask >> k) & 1) ? (unsigned)pc_rel[k] * EB : rel111);
        };
        auto load_affine = [&](const Work& w, int cb, auto SET) {   // always NA loads: the waits below count instructions
            constexpr int set = decltype(SET)::value;
            const float* ps = has_affine ? p.in_scale + (size_t)w.n * p.Cin + cb * 16 : p.in;
            const float* ph = has_affine ? p.in_shift + (size_t)w.n * p.Cin + cb * 16 : p.in;
#pragma unroll
            for (int q = 0; q < NA / 2; ++q) {
                sc[set][q] = load16_untracked(ps, 4u * (unsigned)(NCH * sub + 4 * q));
                sh[set][q] = load16_untracked(ph, 4u * (unsigned)(NCH * sub + 4 * q));
            }
        };
        // a tensor without a pending affine gets the identity, once per step (behind the wait that covers the loads above): the
        // conversion below is then branch- and select-free (x * 1 + 0 is x)
        auto fix_affine = [&](auto SET) {
            constexpr int set = decltype(SET)::value;
            if (!has_affine) {
#pragma unroll
                for (int q = 0; q < NA / 2; ++q) { sc[set][q] = f32x4{1.f, 1.f, 1.f, 1.f}; sh[set][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
        };
        // counted waits: one asm statement that the compiler must order every use of the named registers behind
#define NM_Q2_REGS_W "+v"(wreg[0]), "+v"(wreg[1]), "+v"(wreg[2]), "+v"(wreg[3]), "+v"(wreg[4]), "+v"(wreg[5]), "+v"(wreg[6]), \
                     "+v"(wreg[7]), "+v"(wreg[8]), "+v"(wreg[9]), "+v"(wreg[10]), "+v"(wreg[11]), "+v"(wreg[12]), "+v"(wreg[13])
#define NM_Q2_WAIT_W(N) asm volatile("s_waitcnt vmcnt(" #N ")" : NM_Q2_REGS_W :: "memory")
#define NM_Q2_WAIT_SET_H(S, N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(raw[S][0]), "+v"(raw[S][1]), "+v"(raw[S][2]), "+v"(raw[S][3]), "+v"(raw[S][4]), \
                             "+v"(sc[S][0]), "+v"(sc[S][1]), "+v"(sh[S][0]), "+v"(sh[S][1]) :: "memory")
#define NM_Q2_WAIT_SET_F(S, N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(raw[S][0]), "+v"(raw[S][1]), "+v"(raw[S][2]), "+v"(raw[S][3]), "+v"(raw[S][4]), \
                             "+v"(raw[S][5]), "+v"(raw[S][6]), "+v"(raw[S][7]), "+v"(raw[S][8]), "+v"(raw[S][9]), "+v"(sc[S][0]), "+v"(sh[S][0]) :: "memory")
#define NM_Q2_WAIT_SET(S, NH, NF) do { if constexpr (IH) { NM_Q2_WAIT_SET_H(S, NH); } else { NM_Q2_WAIT_SET_F(S, NF); } } while (0)
#define NM_Q2_WAIT_WP(NH, NF) do { if constexpr (IH) { NM_Q2_WAIT_W(NH); } else { NM_Q2_WAIT_W(NF); } } while (0)
        auto convert = [&](int buf, unsigned mask, auto K, auto SET) {   // activate, round to fp16, write piece k into halo buffer buf
            constexpr int k = decltype(K)::value, set = decltype(SET)::value;
            const bool keep = ((mask >> k) & 1) != 0;               // padding is zero AFTER the activation: selected on the packed halves
            unsigned pk[NCH / 2];
#pragma unroll
            for (int e = 0; e < NCH; e += 2) {
                float v0, v1;
                // (by value: bit_cast of a vector element through a reference reads element 0)
                if constexpr (IH) { const float rf = raw[set][k][e >> 1]; const unsigned u = nm_fbits(rf); v0 = nm_bf_lo(u); v1 = nm_bf_hi(u); }
                else { v0 = raw[set][k][e]; v1 = raw[set][k][e + 1]; }
                // the pair's affine and slope product as PACKED fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two elements per issue
                // slot - the producers are bound by vector issue); the same fma / mul / max per element
                f32x2 vv = f32x2{v0, v1};
                const f32x4 sq = sc[set][e >> 2], hq = sh[set][e >> 2];
                const f32x2 s2 = (e & 2) ? f32x2{sq[2], sq[3]} : f32x2{sq[0], sq[1]}, h2 = (e & 2) ? f32x2{hq[2], hq[3]} : f32x2{hq[0], hq[1]};
                vv = __builtin_elementwise_fma(vv, s2, h2);
                const f32x2 vs = vv * f32x2{p.in_slope, p.in_slope};
                vv = f32x2{fmaxf(vv[0], vs[0]), fmaxf(vv[1], vs[1])};
                half2v hv = __builtin_convertvector(vv, half2v);
                asm volatile("" : "+v"(hv));
                pk[e >> 1] = keep ? __builtin_bit_cast(unsigned, hv) : 0u;
            }
            if (pc_pos[k] >= 0) {
                if constexpr (IH) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<u32x4*>(ldh + buf * 2 * HV + pc_slot[k]) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                } else {
                    reinterpret_cast<nm_u32x2*>(ldh + buf * 2 * HV + pc_slot[k])[sub & 1] = nm_u32x2{pk[0], pk[1]};
                }
            }
        };
        // the step's weights: 54 one-KiB wave pieces (tap t, plane r of (cout group cg, chunk cb)), piece j = pw + 4 i of producer wave pw:
        // r = pw & 1 for every piece of the wave and t = (pw >> 1) + 2 i, so a piece's source is src0 + i * wstep and its LDS slot
        // dst0 + 256 i - one add per load, an immediate offset per store (slots 54, 55 - i = 13 of waves 2, 3 - repeat the wave's piece 12:
        // identical bytes to the same place)
        const unsigned wst_t = (unsigned)((size_t)C16 * 4 * plane * 16), wst_r = (unsigned)(plane * 16);
        const unsigned wsrc0 = (unsigned)(pw >> 1) * wst_t + (unsigned)(pw & 1) * wst_r + (unsigned)lane * 16u, wstep = 2u * wst_t;
        const int ilast = pw < 2 ? 13 : 12;
        const bool wdma = p.wdma != 0;
        auto load_w = [&](int cg, int cb, int buf) {
            const float* base = reinterpret_cast<const float*>(w8 + ((size_t)cb * 4) * plane + cg * 64);
            if (wdma) {
                static_for<NWL>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    const int ii = i < 13 ? i : ilast;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + wsrc0 + (unsigned)ii * wstep),
                                                     (__attribute__((address_space(3))) void*)(ldb + buf * WB + pw * 64 + ii * 256), 16, 0, 0);
                });
            } else {
                static_for<NWL>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    wreg[i] = load16_untracked(base, wsrc0 + (unsigned)(i < 13 ? i : ilast) * wstep);
                });
            }
        };
        auto store_w = [&](int buf) {
            if (wdma) return;
            half8* dst0 = ldb + buf * WB + pw * 64 + lane;
            static_for<13>([&](auto I) {
                constexpr int i = decltype(I)::value;
                *reinterpret_cast<f32x4*>(dst0 + i * 256) = wreg[i];
            });
            *reinterpret_cast<f32x4*>(dst0 + ilast * 256) = wreg[13];
        };

        Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
        int hb = 0;                                                 // buffers (halo and weights alike) of the step the MFMA waves run
        Step s1 = cur; bool has1 = advance(s1);
        Step s2 = s1;  bool has2 = has1 && advance(s2);
        Step s3 = s2;  bool has3 = has2 && advance(s3);
        unsigned m1 = inside_mask(cur.w), m2, m3;
        for (int c = pt; c < p.Cout; c += 256) lbias[c] = p.bias ? p.bias[c] : 0.f;
        // prologue: first tile + first step's weights in the open; the tiles of steps s1 (set 0) and s2 (set 1) in flight
        load_w(cur.w.cg, 0, 0);
        load_affine(cur.w, 0, ic<0>{});
        { const float* b = tile_base(cur.w, 0); static_for<NP>([&](auto K) { load_piece(b, m1, K, ic<0>{}); }); }
        NM_Q2_WAIT_SET(0, 0, 0);
        NM_Q2_WAIT_W(0);
        fix_affine(ic<0>{});
        store_w(0);
        static_for<NP>([&](auto K) { convert(0, m1, K, ic<0>{}); });
        m1 = inside_mask(s1.w);
        load_affine(s1.w, s1.cb, ic<0>{});
        { const float* b = tile_base(s1.w, s1.cb); static_for<NP>([&](auto K) { load_piece(b, m1, K, ic<0>{}); }); }
        m2 = inside_mask(s2.w);
        load_affine(s2.w, s2.cb, ic<1>{});
        { const float* b = tile_base(s2.w, s2.cb); static_for<NP>([&](auto K) { load_piece(b, m2, K, ic<1>{}); }); }
        lds_barrier();
        // one producer iteration = one step of the MFMA waves (`cur`): set SET holds the tile of s1 (converted now into the other
        // halo buffer) and is refilled with the tile of s3; the other set holds s2.  false after the block's last step.
        auto iter = [&](auto SET) -> bool {
            constexpr int set = decltype(SET)::value;
            m3 = (s3.w.n != s2.w.n || s3.w.oz0 != s2.w.oz0 || s3.w.oy0 != s2.w.oy0 || s3.w.ox0 != s2.w.ox0) ? inside_mask(s3.w) : m2;
            const float* b3 = tile_base(s3.w, s3.cb);
            // weights of step s1 into the other buffer (read last during the step before this one)
            if constexpr (DBG != 2 && DBG != 4 && DBG < 6) load_w(s1.w.cg, s1.cb, hb ^ 1);
            // in order: [tile s1 -> this set][tile s2 -> other set][these NWL weight loads]: this set has landed once only the
            // other set's loads and the weight loads are outstanding
            NM_Q2_WAIT_SET(set, 23, 26);
            fix_affine(SET);
            if constexpr (DBG != 2 && DBG != 5 && DBG < 6) {
                static_for<NP>([&](auto K) { if constexpr (DBG != 8) convert(hb ^ 1, m1, K, SET); if constexpr (DBG != 9) load_piece(b3, m3, K, SET); });
                if constexpr (DBG != 9) load_affine(s3.w, s3.cb, SET);
            }
            NM_Q2_WAIT_WP(9, 12);                                   // the weight loads: only this iteration's NP + NA loads are younger
            if constexpr (DBG != 2 && DBG != 4 && DBG < 6) store_w(hb ^ 1);
            lds_barrier();                                          // publishes tile + weights of s1; the MFMA waves are done with `cur`
            if (cur.cb == C16 - 1 && DBG != 7) lds_barrier();       // the MFMA waves' epilogue barrier of a finished brick
            if (!has1) return false;
            cur = s1; s1 = s2; has1 = has2; s2 = s3; has2 = has3; has3 = has3 && advance(s3);
            m1 = m2; m2 = m3; hb ^= 1;
            return true;
        };
        for (;;) { if (!iter(ic<0>{})) break; if (!iter(ic<1>{})) break; }
        NM_Q2_WAIT_W(0);                                            // loads still in flight for steps that do not exist
        return;
    }

    // =============================== MFMA waves ===============================================================
    __builtin_amdgcn_s_setprio(3);
    int arow0;
    {
        const int c = l31 >> 2;
        const int x = (((0x96 >> c) & 1) << 2) + (l31 & 3), z = c >> 1;
        arow0 = z * ZP + (2 * wave) * HX + x;
    }
    const int vx = ((((0x96 >> (l31 >> 2)) & 1) << 2) + (l31 & 3)), vz = l31 >> 3;
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
    int hb = 0;
    lds_barrier();                                                  // the producers' prologue

    constexpr int YO = HX * 16;                                     // brick row 2 wave -> 2 wave + 1
    half8 a0[3], a1[3], b0[3], b1[3];                               // operands of three taps in flight (index = tap % 3)
    for (;;) {
        Step nxt = cur;
        const bool has_next = advance(nxt);
        const unsigned va = (unsigned)(size_t)(ldh + hb * 2 * HV + h * HV + arow0);
        const unsigned vb = (unsigned)(size_t)(ldb + hb * WB + h * 64 + l31);
        // taps 0 and 1 (the tile and the weights were published by the barrier just passed)
        a0[0] = lds_read16_untracked<0>(va); a1[0] = lds_read16_untracked<YO>(va);
        b0[0] = lds_read16_untracked<0>(vb); b1[0] = lds_read16_untracked<32 * 16>(vb);
        a0[1] = lds_read16_untracked<16>(va); a1[1] = lds_read16_untracked<16 + YO>(va);
        b0[1] = lds_read16_untracked<128 * 16>(vb); b1[1] = lds_read16_untracked<128 * 16 + 32 * 16>(vb);
        static_for<27>([&](auto TT) {
            constexpr int tt = decltype(TT)::value, i = tt % 3, u = tt + 2, j = u % 3;
            // in-order LDS returns: at most the four reads of tap tt + 1 may still be outstanding
            if constexpr (tt < 26 && DBG != 6) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0[i]), "+v"(a1[i]), "+v"(b0[i]), "+v"(b1[i]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[i]), "+v"(a1[i]), "+v"(b0[i]), "+v"(b1[i]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            constexpr bool more = u < 27 && DBG != 6;
            constexpr int AO = more ? ((u / 9) * ZP + ((u % 9) / 3) * HX + (u % 3)) * 16 : 0;
            constexpr int BO = more ? u * 128 * 16 : 0;
#define NM_Q2_MFMA(ACC, B, A) if constexpr (DBG != 1) { NM_MFMA2(ACC, B, A); } else { asm volatile("" :: "v"(B), "v"(A)); __builtin_amdgcn_sched_barrier(0); }
            NM_Q2_MFMA(acc[0][0], b0[i], a0[i]); if constexpr (more) a0[j] = lds_read16_untracked<AO>(va);
            NM_Q2_MFMA(acc[0][1], b1[i], a0[i]); if constexpr (more) a1[j] = lds_read16_untracked<AO + YO>(va);
            NM_Q2_MFMA(acc[1][0], b0[i], a1[i]); if constexpr (more) b0[j] = lds_read16_untracked<BO>(vb);
            NM_Q2_MFMA(acc[1][1], b1[i], a1[i]); if constexpr (more) b1[j] = lds_read16_untracked<BO + 32 * 16>(vb);
        });
        asm volatile("s_barrier" ::: "memory");                     // the producers publish the next step's tile and weights
        __builtin_amdgcn_sched_barrier(0);
        if (cur.cb == C16 - 1 && (DBG != 7 || acc[0][0][0] == 12345.678f)) {
            // ---- epilogue of the finished brick (exposed): transposed accumulators -> 16-byte stores, DPP partial sums
            const Work& w = cur.w;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float s1[16], s2[16];                               // (value pairs on v_pk_* as in conv_f16p2)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const size_t dst = ((((size_t)w.n * p.OD + w.oz0 + vz) * p.OH + w.oy0 + 2 * wave + mt) * p.OW + w.ox0 + vx) * (size_t)p.Cout +
                                       w.cg * 64 + nt * 32 + 4 * h;
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbias + w.cg * 64 + nt * 32 + 8 * k4 + 4 * h);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const f32x2 a2 = f32x2{acc[mt][nt][4 * k4 + e], acc[mt][nt][4 * k4 + e + 1]};
                            const f32x2 v2 = (a2 + f32x2{0.f, 0.f} * (1.0f / NM_SPLIT_SCALE)) + f32x2{b4[e], b4[e + 1]};      // (conv_f16p2<SINGLE>'s expression: its correction accumulator is zero)
                            const f32x2 q2 = v2 * v2;
                            v[e] = v2[0]; v[e + 1] = v2[1];
                            if (mt == 0) { s1[4 * k4 + e] = v2[0]; s1[4 * k4 + e + 1] = v2[1]; s2[4 * k4 + e] = q2[0]; s2[4 * k4 + e + 1] = q2[1]; }
                            else {
                                const f32x2 t1 = f32x2{s1[4 * k4 + e], s1[4 * k4 + e + 1]} + v2, t2 = f32x2{s2[4 * k4 + e], s2[4 * k4 + e + 1]} + q2;
                                s1[4 * k4 + e] = t1[0]; s1[4 * k4 + e + 1] = t1[1]; s2[4 * k4 + e] = t2[0]; s2[4 * k4 + e + 1] = t2[1];
                            }
                            acc[mt][nt][4 * k4 + e] = 0.f; acc[mt][nt][4 * k4 + e + 1] = 0.f;
                        }
                        if constexpr (DBG != 3) nm_st4<OH>(p.out, dst + 8 * k4, v);
                        else if (v[0] == 12345.678f) nm_st4<OH>(p.out, dst + 8 * k4, v);
                    }
                }
                if (p.part) {
                    dpp_sum32_many(s1); dpp_sum32_many(s2);
                    if (l31 == 16) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            *reinterpret_cast<f32x2*>(red + (wave * 64 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 2) = f32x2{s1[r], s2[r]};
                    }
                }
            }
            lds_barrier();                                          // (matched by the producers)
            if (p.part && tid < 64) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { a += red[(q * 64 + tid) * 2]; b += red[(q * 64 + tid) * 2 + 1]; }
                const int br = ((w.oz0 >> 2) * nby + (w.oy0 >> 3)) * nbx + (w.ox0 >> 3);
                float* dp = p.part + (((size_t)w.n * nbr + br) * p.Cout + w.cg * 64 + tid) * 2;
                dp[0] = a; dp[1] = b;
            }
        }
        if (!has_next) break;
        cur = nxt; hb ^= 1;
    }
}

// ---- conv_f16r: the one-product modes' kernel for the 32-output-channel layers, weights RESIDENT in LDS (round 5) -------------------
// conv_f16p<SINGLE> runs the Cout = 32 layers (the decoder's 32 -> 32 @64^3 conv, the data gradients that end in 32 channels) with the
// three-product kernel's schedule: 9-tap weight groups streamed per step, three barriers per step, at 0.26 of the time on the matrix
// pipe (profiles/r05_pmc_mfma_train_bf16.json) - 2.3 ms per 64-frame launch at 64^3.  A 32-channel layer's hi-only weights are small:
// 27 taps x Cin x 32 outputs x 2 B = 55 KB (Cin = 32) or 110 KB (Cin = 64).  Here they are loaded ONCE per workgroup and stay; the
// producer waves only stage halo tiles (conv_f16q2's two-register-set pipeline: a tile's loads are issued two steps ahead), one
// workgroup barrier per 16-channel step, and the MFMA waves keep operands three taps ahead (a tap is two MFMAs here).
//   LDS: halo [2][h0 | h1][600] x 16 B = 38 KB, weights [27 taps][C16][h0 | h1][32] x 16 B, GroupNorm scratch, bias.
// k order per output element: channel chunk outer, tap inner (conv_f16q2's); epilogue identical.
template <int IO, int C16T>
__global__ __launch_bounds__(512, 1) void conv_f16r_kernel(ConvParams p) {
    constexpr bool IH = (IO & 1) != 0, OH = (IO & 2) != 0;
    constexpr unsigned EB = IH ? 2u : 4u;                           // bytes per input element
    constexpr int HV = 600, ZP = 100, HX = 10;
    constexpr int NP = IH ? 5 : 10;                                 // 16-byte input pieces per producer thread and step
    constexpr int NCH = IH ? 8 : 4;                                 // channels per piece
    constexpr int NA = IH ? 4 : 2;                                  // affine loads per step
    constexpr int TS = C16T * 64;                                   // half8 slots of one tap: [C16T][h0 | h1][32]
    extern __shared__ f32x4 lds[];
    half8* ldh = reinterpret_cast<half8*>(lds);                     // [2][2][HV]
    half8* ldb = ldh + 4 * HV;                                      // [27][C16T][2][32]
    float* red = reinterpret_cast<float*>(ldb + 27 * TS);           // [2 brick parities][4 waves][32 channels][2]
    float* lbias = red + 512;                                       // [32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int nbx = p.OW >> 3, nby = p.OH >> 3, nbz = p.OD >> 2, nbr = nbx * nby * nbz;
    const int total = p.N * nbr;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int id_first = (int)blockIdx.x * per, id_last = min(total, id_first + per);
    if (id_first >= id_last) return;

    struct Work { int n, oz0, oy0, ox0; };
    const bool tiled = p.st_x != 0;
    auto decode = [&](int id) {
        Work w;
        if (tiled) {
            const BrickPos bp = super_tile_item(p, (int)blockIdx.x, id - id_first, per, nbz, nby, nbx);
            w.n = bp.n; w.ox0 = bp.bx << 3; w.oy0 = bp.by << 3; w.oz0 = bp.bz << 2;
            return w;
        }
        w.n = id / nbr; const int br = id % nbr;
        w.ox0 = (br % nbx) << 3; w.oy0 = ((br / nbx) % nby) << 3; w.oz0 = (br / (nbx * nby)) << 2;
        return w;
    };
    auto next_work = [&](Work w, int id) {
        if (tiled) return decode(id);
        if ((w.ox0 += 8) == p.OW) { w.ox0 = 0; if ((w.oy0 += 8) == p.OH) { w.oy0 = 0; if ((w.oz0 += 4) == p.OD) { w.oz0 = 0; ++w.n; } } }
        return w;
    };
    struct Step { Work w; int id, cb; };
    auto advance = [&](Step& st) {                                  // false (and st unchanged) on the block's last step
        if (st.cb + 1 < C16T) { ++st.cb; return true; }
        if (st.id + 1 >= id_last) return false;
        st.cb = 0; ++st.id; st.w = next_work(st.w, st.id);
        return true;
    };

    if (wave >= 4) {
        // =========================== producer waves ===========================================================
        const int pt = tid - 256, pw = wave - 4;
        const half8* __restrict__ w8 = reinterpret_cast<const half8*>(p.w);
        const size_t plane = (size_t)p.Co_pad;
        // the layer's weights, once: 1-KiB pieces (tap t, chunk cb) = the hi planes of both lane halves x 32 outputs
        for (int j = pw; j < 27 * C16T; j += 4) ldb[j * 64 + lane] = w8[((size_t)j * 4 + (lane >> 5)) * plane + (lane & 31)];
        for (int c = pt; c < 32; c += 256) lbias[c] = p.bias ? p.bias[c] : 0.f;
        const int sub = IH ? (pt & 1) : (pt & 3);
        int pc_slot[NP], pc_rel[NP], pc_pos[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int v = IH ? (pt >> 1) + 128 * k : (pt >> 2) + 64 * k, vc = min(v, HV - 1);
            const int hz = vc / 100, r = vc % 100, hy = r / 10, hx = r % 10;
            pc_slot[k] = (IH ? sub : (sub >> 1)) * HV + hz * ZP + hy * HX + hx;
            pc_rel[k] = ((hz * p.IH + hy) * p.IW + hx) * p.Cin + NCH * sub;
            pc_pos[k] = (v < HV) ? (hz | (hy << 8) | (hx << 16)) : -1;
        }
        f32x4 raw[2][NP], sc[2][NA / 2], sh[2][NA / 2];
        const bool has_affine = p.in_scale != nullptr;
        const unsigned rel111 = (unsigned)(((p.IH + 1) * p.IW + 1) * p.Cin) * EB;   // halo voxel (1,1,1): always inside
        auto inside_mask = [&](const Work& w) {
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int hz = pc_pos[k] & 0xff, hy = (pc_pos[k] >> 8) & 0xff, hx = pc_pos[k] >> 16;
                const bool in = pc_pos[k] >= 0 && (unsigned)(w.oz0 - 1 + hz) < (unsigned)p.ID && (unsigned)(w.oy0 - 1 + hy) < (unsigned)p.IH &&
                                (unsigned)(w.ox0 - 1 + hx) < (unsigned)p.IW;
                m |= (in ? 1u : 0u) << k;
            }
            return m;
        };
        auto tile_base = [&](const Work& w, int cb) {
            return nm_eptr(p.in, (size_t)(((((long long)w.n * p.ID + (w.oz0 - 1)) * p.IH + (w.oy0 - 1)) * p.IW + (w.ox0 - 1)) * (long long)p.Cin + cb * 16), IH);
        };
        auto load_piece = [&](const float* base, unsigned mask, auto K, auto SET) {
            constexpr int k = decltype(K)::value, set = decltype(SET)::value;
            raw[set][k] = load16_untracked(base, ((mask >> k) & 1) ? (unsigned)pc_rel[k] * EB : rel111);
        };
        auto load_affine = [&](const Work& w, int cb, auto SET) {   // always NA loads: the waits below count instructions
            constexpr int set = decltype(SET)::value;
            const float* ps = has_affine ? p.in_scale + (size_t)w.n * p.Cin + cb * 16 : p.in;
            const float* ph = has_affine ? p.in_shift + (size_t)w.n * p.Cin + cb * 16 : p.in;
#pragma unroll
            for (int q = 0; q < NA / 2; ++q) {
                sc[set][q] = load16_untracked(ps, 4u * (unsigned)(NCH * sub + 4 * q));
                sh[set][q] = load16_untracked(ph, 4u * (unsigned)(NCH * sub + 4 * q));
            }
        };
        auto fix_affine = [&](auto SET) {
            constexpr int set = decltype(SET)::value;
            if (!has_affine) {
#pragma unroll
                for (int q = 0; q < NA / 2; ++q) { sc[set][q] = f32x4{1.f, 1.f, 1.f, 1.f}; sh[set][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
        };
        auto convert = [&](int buf, unsigned mask, auto K, auto SET) {   // activate, round to fp16, write piece k into halo buffer buf
            constexpr int k = decltype(K)::value, set = decltype(SET)::value;
            const bool keep = ((mask >> k) & 1) != 0;               // padding is zero AFTER the activation: selected on the packed halves
            unsigned pk[NCH / 2];
#pragma unroll
            for (int e = 0; e < NCH; e += 2) {
                float v0, v1;
                if constexpr (IH) { const float rf = raw[set][k][e >> 1]; const unsigned u = nm_fbits(rf); v0 = nm_bf_lo(u); v1 = nm_bf_hi(u); }
                else { v0 = raw[set][k][e]; v1 = raw[set][k][e + 1]; }
                f32x2 vv = f32x2{v0, v1};
                const f32x4 sq = sc[set][e >> 2], hq = sh[set][e >> 2];
                const f32x2 s2 = (e & 2) ? f32x2{sq[2], sq[3]} : f32x2{sq[0], sq[1]}, h2 = (e & 2) ? f32x2{hq[2], hq[3]} : f32x2{hq[0], hq[1]};
                vv = __builtin_elementwise_fma(vv, s2, h2);
                const f32x2 vs = vv * f32x2{p.in_slope, p.in_slope};
                vv = f32x2{fmaxf(vv[0], vs[0]), fmaxf(vv[1], vs[1])};
                half2v hv = __builtin_convertvector(vv, half2v);
                asm volatile("" : "+v"(hv));
                pk[e >> 1] = keep ? __builtin_bit_cast(unsigned, hv) : 0u;
            }
            if (pc_pos[k] >= 0) {
                if constexpr (IH) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<u32x4*>(ldh + buf * 2 * HV + pc_slot[k]) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                } else {
                    reinterpret_cast<nm_u32x2*>(ldh + buf * 2 * HV + pc_slot[k])[sub & 1] = nm_u32x2{pk[0], pk[1]};
                }
            }
        };

        Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
        int hb = 0;                                                 // halo buffer of the step the MFMA waves run
        Step s1 = cur; bool has1 = advance(s1);
        Step s2 = s1;  bool has2 = has1 && advance(s2);
        Step s3 = s2;  bool has3 = has2 && advance(s3);
        unsigned m1 = inside_mask(cur.w), m2, m3;
        // prologue: first tile in the open; the tiles of steps s1 (set 0) and s2 (set 1) in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the weight loads above are the compiler's)
        load_affine(cur.w, 0, ic<0>{});
        { const float* b = tile_base(cur.w, 0); static_for<NP>([&](auto K) { load_piece(b, m1, K, ic<0>{}); }); }
        NM_Q2_WAIT_SET(0, 0, 0);
        fix_affine(ic<0>{});
        static_for<NP>([&](auto K) { convert(0, m1, K, ic<0>{}); });
        m1 = inside_mask(s1.w);
        load_affine(s1.w, s1.cb, ic<0>{});
        { const float* b = tile_base(s1.w, s1.cb); static_for<NP>([&](auto K) { load_piece(b, m1, K, ic<0>{}); }); }
        m2 = inside_mask(s2.w);
        load_affine(s2.w, s2.cb, ic<1>{});
        { const float* b = tile_base(s2.w, s2.cb); static_for<NP>([&](auto K) { load_piece(b, m2, K, ic<1>{}); }); }
        lds_barrier();
        // one producer iteration = one step of the MFMA waves (`cur`): set SET holds the tile of s1 (converted now into the other
        // halo buffer) and is refilled with the tile of s3; the other set holds s2.  false after the block's last step.
        auto iter = [&](auto SET) -> bool {
            constexpr int set = decltype(SET)::value;
            m3 = (s3.w.n != s2.w.n || s3.w.oz0 != s2.w.oz0 || s3.w.oy0 != s2.w.oy0 || s3.w.ox0 != s2.w.ox0) ? inside_mask(s3.w) : m2;
            const float* b3 = tile_base(s3.w, s3.cb);
            NM_Q2_WAIT_SET(set, 9, 12);                             // only the other set's NP + NA loads are younger
            fix_affine(SET);
            static_for<NP>([&](auto K) { convert(hb ^ 1, m1, K, SET); load_piece(b3, m3, K, SET); });
            load_affine(s3.w, s3.cb, SET);
            lds_barrier();                                          // publishes the tile of s1; the MFMA waves are done with `cur`
            if (!has1) return false;
            cur = s1; s1 = s2; has1 = has2; s2 = s3; has2 = has3; has3 = has3 && advance(s3);
            m1 = m2; m2 = m3; hb ^= 1;
            return true;
        };
        for (;;) { if (!iter(ic<0>{})) break; if (!iter(ic<1>{})) break; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // loads still in flight for steps that do not exist
        lds_barrier();                                              // the MFMA waves' barrier behind the block's LAST epilogue (the others are deferred)
        return;
    }

    // =============================== MFMA waves ===============================================================
    __builtin_amdgcn_s_setprio(3);
    int arow0;
    {
        const int c = l31 >> 2;
        const int x = (((0x96 >> c) & 1) << 2) + (l31 & 3), z = c >> 1;
        arow0 = z * ZP + (2 * wave) * HX + x;
    }
    const int vx = ((((0x96 >> (l31 >> 2)) & 1) << 2) + (l31 & 3)), vz = l31 >> 3;
    // DEFERRED epilogue: a finished brick's accumulators stay where they are, the next brick's MFMAs go to a second set (32 registers -
    // this kernel has room), and the finished brick's epilogue - bias, 16-byte stores, the GroupNorm partial sums by DPP: ~350 vector
    // instructions per wave, in the open ~1 400 cycles per brick of 1 700 (Cin = 32) or 3 500 (Cin = 64) cycles of MFMAs - runs in
    // pieces behind the MFMAs of the next brick's first two steps: channel quads 0, 1 during chunk 0, quads 2, 3 during chunk 1 (per
    // quad: the two 4-value chunks mt = 0, 1, then four channel-sum reductions; half a unit per tap, a quarter behind each MFMA); the
    // partial-sum row after chunk 1's barrier.  Same arithmetic per element and order of every sum as the exposed form.
    f32x16 acc[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[st][mt][r] = 0.f;

    Step cur; cur.w = decode(id_first); cur.id = id_first; cur.cb = 0;
    int hb = 0;
    lds_barrier();                                                  // the producers' prologue (weights, bias, first tile)

    constexpr int YO = HX * 16;                                     // brick row 2 wave -> 2 wave + 1
    half8 a0[4], a1[4], b0[4];                                      // operands of four taps in flight (index = tap % 4)
    Work pw = cur.w;                                                // the finished brick whose accumulators wait in the other set
    bool pend = false;
    int aset = 0;                                                   // accumulator set of the brick in work
    float s1[4], s2[4];                                             // channel sums of the channel quad in work
    f32x4 bq = f32x4{0.f, 0.f, 0.f, 0.f}, vch = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    // where a lane stores a channel's totals (+ a compile-time offset per channel): lanes 16 / 48 into the wave's row of `red`, the others
    // into 8-byte slots behind the bias that nobody reads (no exec-mask branch inside the tap loop)
    // (`red` is double buffered by brick parity: the row of brick k is read - part_row, by wave 0 - while the other waves may already be
    //  storing the sums of brick k + 1 when a brick has only two steps)
    unsigned rbase = l31 == 16 ? (unsigned)(size_t)red + (unsigned)((wave * 32 + 4 * h) * 8) : (unsigned)(size_t)(lbias + 32) + (unsigned)tid * 8u;
    asm volatile("" : "+v"(rbase));
    unsigned rcur = rbase;
    int rpar = 0;
    const unsigned bbase = (unsigned)(size_t)lbias + (unsigned)(4 * h) * 4u;
    unsigned loff[2];                                               // the lane's element offset inside a brick's output (+ the brick's first element, + a channel offset)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { loff[mt] = (unsigned)(((vz * p.OH + 2 * wave + mt) * p.OW + vx) * p.Cout + 4 * h); asm volatile("" : "+v"(loff[mt])); }
    size_t obrick = 0;                                              // first element of the pending brick
    // quarter Q of unit U (0-23) of the pending brick, accumulator set SP: per channel quad k4 = U / 6 the chunks mt = 0, 1 (one value per
    // quarter), then the quad's four channel-sum reductions
    auto piece = [&](auto SP_, auto U_, auto Q_) __attribute__((always_inline)) {
        constexpr int SP = decltype(SP_)::value, U = decltype(U_)::value, Q = decltype(Q_)::value;
        constexpr int k4 = U / 6, w6 = U % 6;
        if constexpr (w6 < 2) {
            constexpr int mt = w6, r = 4 * k4 + Q;
            const float v = acc[SP][mt][r] + bq[Q];
            vch[Q] = v;
            if constexpr (mt == 0) { s1[Q] = 0.f + v; s2[Q] = v * v; } else { s1[Q] += v; s2[Q] += v * v; }
            acc[SP][mt][r] = 0.f;
            if constexpr (Q == 3) nm_st4<OH>(p.out, obrick + loff[mt] + 8 * k4, vch);
        } else {
            constexpr int e = w6 - 2, r = 4 * k4 + e;
            if constexpr (Q == 0) s1[e] = dpp_sum32(s1[e]);
            if constexpr (Q == 1) s2[e] = dpp_sum32(s2[e]);
            if constexpr (Q == 2) {
                const unsigned rb_ = rcur; const f32x2 pr_ = f32x2{s1[e], s2[e]};           // (locals: asm operands cannot name captures of a generic lambda)
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(rb_), "v"(pr_), "n"(((r & 3) + 8 * (r >> 2)) * 8) : "memory");
            }
        }
    };
    // bias of channel quad K4: an untracked 16-byte LDS read (one more entry in the in-order LDS queue: the counted waits of the tap loop
    // only ever wait for more), issued at least two taps ahead of its first use
    auto load_bias = [&](auto K4_) __attribute__((always_inline)) {
        constexpr int K4 = decltype(K4_)::value;
        bq = __builtin_bit_cast(f32x4, lds_read16_untracked<8 * K4 * 4>(bbase));
    };
    auto part_row = [&]() __attribute__((always_inline)) {            // after the barrier that publishes the four waves' channel sums
        if (p.part && tid < 32) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { a += red[rpar * 256 + (q * 32 + tid) * 2]; b += red[rpar * 256 + (q * 32 + tid) * 2 + 1]; }
            const int br = ((pw.oz0 >> 2) * nby + (pw.oy0 >> 3)) * nbx + (pw.ox0 >> 3);
            float* dp = p.part + (((size_t)pw.n * nbr + br) * p.Cout + tid) * 2;
            dp[0] = a; dp[1] = b;
        }
    };
    // one step: MFMAs into set SC; PH = 1 / 2: units 12 (PH - 1) .. 12 (PH - 1) + 11 of the pending brick behind them (unit 12 (PH - 1) + tt / 2 at tap tt < 24)
    auto run_step = [&](auto SC_, auto PH_) __attribute__((always_inline)) {
        constexpr int SC = decltype(SC_)::value, PH = decltype(PH_)::value;
        if constexpr (PH != 0) load_bias(ic<2 * (PH == 2 ? 1 : 0)>{});
        const unsigned va = (unsigned)(size_t)(ldh + hb * 2 * HV + h * HV + arow0);
        unsigned vbz[3];
        vbz[0] = (unsigned)(size_t)(ldb + cur.cb * 64 + h * 32 + l31);
        vbz[1] = vbz[0] + 9 * TS * 16; vbz[2] = vbz[0] + 18 * TS * 16;
        a0[0] = lds_read16_untracked<0>(va);  a1[0] = lds_read16_untracked<YO>(va);      b0[0] = lds_read16_untracked<0>(vbz[0]);
        a0[1] = lds_read16_untracked<16>(va); a1[1] = lds_read16_untracked<16 + YO>(va); b0[1] = lds_read16_untracked<TS * 16>(vbz[0]);
        a0[2] = lds_read16_untracked<32>(va); a1[2] = lds_read16_untracked<32 + YO>(va); b0[2] = lds_read16_untracked<2 * TS * 16>(vbz[0]);
        static_for<27>([&](auto TT) __attribute__((always_inline)) {
            constexpr int tt = decltype(TT)::value, i = tt % 4, u = tt + 3, j = u % 4;
            // in-order LDS returns: at most the reads of the next two taps (and a bias read / channel-sum store) may still be outstanding
            if constexpr (tt < 25) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a0[i]), "+v"(a1[i]), "+v"(b0[i]) :: "memory");
            else if constexpr (tt == 25) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a0[i]), "+v"(a1[i]), "+v"(b0[i]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[i]), "+v"(a1[i]), "+v"(b0[i]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            constexpr bool more = u < 27;
            constexpr int AO = more ? ((u / 9) * ZP + ((u % 9) / 3) * HX + (u % 3)) * 16 : 0;
            constexpr int BO = more ? (u % 9) * TS * 16 : 0;
            constexpr int BZ = more ? u / 9 : 0;
            constexpr bool pc = PH != 0 && tt < 24;
            constexpr int U = 12 * (PH == 2 ? 1 : 0) + tt / 2, Q0 = (tt & 1) * 2;
            NM_MFMA2(acc[SC][0], b0[i], a0[i]); if constexpr (more) { a0[j] = lds_read16_untracked<AO>(va); a1[j] = lds_read16_untracked<AO + YO>(va); }
            if constexpr (pc) { piece(ic<SC ^ 1>{}, ic<U>{}, ic<Q0>{}); __builtin_amdgcn_sched_barrier(0); }
            NM_MFMA2(acc[SC][1], b0[i], a1[i]); if constexpr (more) b0[j] = lds_read16_untracked<BO>(vbz[BZ]);
            if constexpr (pc) { piece(ic<SC ^ 1>{}, ic<U>{}, ic<Q0 + 1>{}); __builtin_amdgcn_sched_barrier(0); }
            // the phase's second channel quad: its bias two taps ahead of its first use (unit 6 of the phase = tap 12)
            if constexpr (PH != 0 && tt == 9) { load_bias(ic<2 * (PH == 2 ? 1 : 0) + 1>{}); __builtin_amdgcn_sched_barrier(0); }
        });
    };
    // the block's last brick: its epilogue in the open (accumulator set SP)
    auto flush = [&](auto SP_) __attribute__((always_inline)) {
        static_for<24>([&](auto U_) __attribute__((always_inline)) {
            constexpr int U = decltype(U_)::value;
            if constexpr (U % 6 == 0) { load_bias(ic<U / 6>{}); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq) :: "memory"); }
            static_for<4>([&](auto Q_) __attribute__((always_inline)) { piece(SP_, U_, Q_); });
        });
    };
    for (;;) {
        Step nxt = cur;
        const bool has_next = advance(nxt);
        const int ph = pend ? (cur.cb == 0 ? 1 : (cur.cb == 1 ? 2 : 0)) : 0;
        if (aset == 0) { if (ph == 0) run_step(ic<0>{}, ic<0>{}); else if (ph == 1) run_step(ic<0>{}, ic<1>{}); else run_step(ic<0>{}, ic<2>{}); }
        else           { if (ph == 0) run_step(ic<1>{}, ic<0>{}); else if (ph == 1) run_step(ic<1>{}, ic<1>{}); else run_step(ic<1>{}, ic<2>{}); }
        asm volatile("s_barrier" ::: "memory");                     // the producers publish the next step's tile
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 2) { part_row(); pend = false; }
        if (cur.cb == C16T - 1) {
            pend = true; pw = cur.w; aset ^= 1;
            rpar ^= 1; rcur = rbase + (unsigned)rpar * 1024u;
            obrick = ((((size_t)pw.n * p.OD + pw.oz0) * p.OH + pw.oy0) * p.OW + pw.ox0) * (size_t)p.Cout;
        }
        if (!has_next) break;
        cur = nxt; hb ^= 1;
    }
    if (aset == 1) flush(ic<0>{}); else flush(ic<1>{});           // (aset was toggled behind the last brick: its sums sit in the other set)
    lds_barrier();                                                  // (matched by the producers)
    part_row();
}

// OIDHW fp32 -> split fp16 [tap][Cin/16][hi|lo][lane half][Co_pad][8]
__global__ void pack_conv_weight16_kernel(const float* __restrict__ w, int Cout, int Cin, int ks, _Float16* __restrict__ packed,
                                          int Co_pad) {
    const int taps = ks * ks * ks, C16 = (Cin + 15) >> 4;              // channels beyond Cin carry zero weights
    const size_t total = (size_t)taps * C16 * 2 * Co_pad * 8;          // (tap, cb, h, co, j): writes hi and lo
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int j = i & 7; size_t r = i >> 3;
        int co = r % Co_pad; r /= Co_pad;
        int hh = r & 1; r >>= 1;
        int cb = r % C16; int tap = r / C16;
        int ci = cb * 16 + hh * 8 + j;
        float v = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + tap] : 0.f;
        _Float16 hi = (_Float16)v;
        _Float16 lo = (_Float16)((v - (float)hi) * NM_SPLIT_SCALE);
        size_t base = ((((size_t)tap * C16 + cb) * 4) * Co_pad) * 8;
        packed[base + ((size_t)hh * Co_pad + co) * 8 + j] = hi;
        packed[base + ((size_t)(2 + hh) * Co_pad + co) * 8 + j] = lo;
    }
}

struct Tiling { int MT, NT, bz_l2, by_l2, bx_l2, nbz, nby, nbx, KC, HZ, HY, HX, HV, HVp, CVp; size_t lds_bytes; };

Tiling choose_tiling(const ConvGeom& g, int Cin) {
    Tiling t;
    const int vox = g.OD * g.OH * g.OW;
    t.MT = vox >= 256 ? 2 : 1;
    t.NT = (g.Co_pad % 64 == 0) ? 2 : 1;
    const int bm_l2 = t.MT == 2 ? 8 : 7;
    t.bx_l2 = min(3, ceil_log2(g.OW));
    t.by_l2 = min(3, ceil_log2(g.OH));
    t.bz_l2 = bm_l2 - t.bx_l2 - t.by_l2;
    const int BX = 1 << t.bx_l2, BY = 1 << t.by_l2, BZ = min(1 << t.bz_l2, g.OD);
    t.nbx = (g.OW + BX - 1) / BX; t.nby = (g.OH + BY - 1) / BY; t.nbz = (g.OD + (1 << t.bz_l2) - 1) >> t.bz_l2;
    t.HX = (min(BX, g.OW) - 1) * g.stride + g.ks;
    t.HY = (min(BY, g.OH) - 1) * g.stride + g.ks;
    t.HZ = (BZ - 1) * g.stride + g.ks;
    t.HV = t.HX * t.HY * t.HZ;
    t.HVp = t.HV + ((2 - (t.HV & 7)) & 7);
    t.CVp = 0;
    if (g.up2) {
        int cv = ((t.HZ >> 1) + 2) * ((t.HY >> 1) + 2) * ((t.HX >> 1) + 2);
        t.CVp = cv + ((2 - (cv & 7)) & 7);
    }
    // channels staged per pass: 16 by default; more when the halo tile is small (<= 48 KB of LDS in total), 8 when
    // even 16 do not fit in 72 KB
    const size_t per16 = (size_t)4 * (t.HVp + t.CVp) * 16;
    t.KC = (Cin % 16 == 0) ? 16 : 8;
    if (per16 > 72 * 1024) t.KC = 8;
    else {
        int k = (int)((48 * 1024) / per16) * 16;
        if (k > 16) t.KC = min(k, (Cin + 15) & ~15);
        if (t.KC > Cin) t.KC = Cin;
        if (t.KC % 8) t.KC = (Cin % 16 == 0) ? 16 : 8;
    }
    t.lds_bytes = max((size_t)(t.KC / 4) * (t.HVp + t.CVp) * 16, (size_t)4 * 64 * 2 * sizeof(float));
    return t;
}

// ---- optional live timing of the conv launches (bench.py roofline leg) ---------------------------
// (records live in the context: NmLaunchState, nm_common.h).  By default only launches on the profiled context's main stream are
// timed - a side-stream launch overlaps the main stream, so its event-to-event time is not its own; prof_all records those too
// (bench.py lists them separately)
typedef NmProfRec ProfRec;
#define NM_PROF_ON(s) (nm_ls().prof_on && (nm_ls().prof_all || (s) == nm_ls().prof_stream))

hipEvent_t prof_event() {
    std::vector<hipEvent_t>& g_event_pool = nm_ls().event_pool;
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

template <int MT, int NT>
int launch_t(const ConvParams& p, const Tiling& t, dim3 grid, hipStream_t s) {
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<MT, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(conv)");
        attr_set.mark();
    }
    ProfRec rec;
    rec.flops = 2.0 * p.N * (double)p.OD * p.OH * p.OW * p.Cout * (double)p.cin_real * p.ks * p.ks * p.ks;
        // algorithmic work: real channels and taps, no padding
    const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
    if (prof_rec) {
        rec.a = prof_event(); rec.b = prof_event(); rec.variant = (MT - 1) * 2 + (NT - 1);
        (void)hipEventRecord(rec.a, s);
    }
    hipLaunchKernelGGL((conv_mfma_kernel<MT, NT>), grid, dim3(256), t.lds_bytes, s, p);
    if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
    return nm_check_hip(hipGetLastError(), "conv_mfma launch");
}

// XCD super-tile (see super_tile_item): the persistent grid must be 8 x (bricks per super-tile) workgroups, the brick grid
// a multiple of the super-tile, and the super-tiles split evenly over the eight XCDs
void choose_super_tile(ConvParams& p, int nblocks, int nbz, int nby, int nbx) {
    p.st_z = p.st_y = p.st_x = 0;
    p.st_sx = p.st_sy = p.st_pf = p.st_m_sx = p.st_m_sy = p.st_m_pf = 0;
    if (!nm_ls().supertile || nblocks % 8) return;
    const int gs = nblocks / 8;
    const int sz = gs == 64 ? 4 : gs == 32 ? 2 : 0;
    if (!sz || nbz % sz || nby % 4 || nbx % 4) return;
    const long long st = (long long)p.N * (nbz / sz) * (nby / 4) * (nbx / 4);
    if (st % 8 || st / 8 * nblocks != (long long)p.N * nbz * nby * nbx) return;
    p.st_z = sz; p.st_y = 4; p.st_x = 4;
    // the decode's divisions as multiplications (super_tile_item): exact while (largest dividend) x (divisor) < 2^32
    const unsigned SX = (unsigned)nbx / 4u, SY = (unsigned)nby / 4u, pf = SX * SY * (unsigned)(nbz / sz);
    const unsigned long long smax = 8ull * (unsigned long long)(((long long)p.N * nbz * nby * nbx + nblocks - 1) / nblocks) + 8ull;
    auto magic = [](unsigned d) { return d == 1u ? 0u : (unsigned)((0x100000000ull + d - 1ull) / d); };
    if (nm_ls().fast_decode && smax * pf < 0x100000000ull) {
        p.st_sx = SX; p.st_sy = SY; p.st_pf = pf; p.st_m_sx = magic(SX); p.st_m_sy = magic(SY); p.st_m_pf = magic(pf);
    }
}

// conv mode 3 (nm_ls().single): the split-fp16 kernels keep only the hi x hi product (SINGLE instantiations)

template <int MT, int NT, int KS, bool UP2, bool SINGLE, int IO = 0>
int launch_f16s_impl(const ConvParams& p_in, const Tiling& t, dim3 grid, hipStream_t s) {
    ConvParams p = p_in;
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_kernel<MT, NT, KS, UP2, SINGLE, IO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(conv_f16s)");
        attr_set.mark();
    }
    ProfRec rec;
    rec.flops = 2.0 * p.N * (double)p.OD * p.OH * p.OW * p.Cout * (double)p.cin_real * p.ks * p.ks * p.ks;
    const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
    if (prof_rec) {
        rec.a = prof_event(); rec.b = prof_event(); rec.variant = UP2 ? 10 + (NT - 1) : 5 + (NT - 1);      // the fused-upsample layers are families of their own
        (void)hipEventRecord(rec.a, s);
    }
    dim3 pgrid(min(grid.x, 512u), grid.y);                          // persistent: ~2 resident workgroups per CU
    choose_super_tile(p, (int)pgrid.x, p.nbz, p.nby, p.nbx);
    hipLaunchKernelGGL((conv_f16s_kernel<MT, NT, KS, UP2, SINGLE, IO>), pgrid, dim3(256), t.lds_bytes, s, p);
    if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
    return nm_check_hip(hipGetLastError(), "conv_f16s launch");
}
// 16-bit storage (p.in_h / p.out_h): the instantiations the training path of conv mode 4 needs - k1 layers at >= 32^3 (both tensors
// bfloat16) and the fused-upsample layer that crosses the storage threshold (fp32 in, bfloat16 out); anything else is rejected
int io16_unsupported(const char* k, const ConvParams& p) {
    nm_set_error("%s: no instantiation for bfloat16 input=%d / output=%d in this conv mode / layer shape (16-bit storage needs conv mode 4)", k, p.in_h, p.out_h);
    return NM_ERR_UNSUPPORTED;
}
template <int MT, int NT, int KS, bool UP2>
int launch_f16s(const ConvParams& p, const Tiling& t, dim3 grid, hipStream_t s) {
    const int io = (p.in_h ? 1 : 0) | (p.out_h ? 2 : 0);
    if (io) {
        if (!nm_ls().single) return io16_unsupported("conv_f16s", p);
        if constexpr (KS == 1 && !UP2) { if (io == 3) return launch_f16s_impl<MT, NT, KS, UP2, true, 3>(p, t, grid, s); }
        if constexpr (KS == 3 && UP2 && NT == 2) { if (io == 2) return launch_f16s_impl<MT, NT, KS, UP2, true, 2>(p, t, grid, s); }
        return io16_unsupported("conv_f16s", p);
    }
    return nm_ls().single ? launch_f16s_impl<MT, NT, KS, UP2, true>(p, t, grid, s) : launch_f16s_impl<MT, NT, KS, UP2, false>(p, t, grid, s);
}

template <int NT, bool SINGLE>
int launch_pool_f16s_impl(const ConvParams& p, dim3 grid, hipStream_t s) {
    ProfRec rec;
    rec.flops = 2.0 * p.N * (double)p.OD * p.OH * p.OW * p.Cout * (double)p.cin_real * 8.0;
    const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
    if (prof_rec) {
        rec.a = prof_event(); rec.b = prof_event(); rec.variant = 8;
        (void)hipEventRecord(rec.a, s);
    }
    const int io = (p.in_h ? 1 : 0) | (p.out_h ? 2 : 0);
    if (io) {                                                       // 16-bit storage: the pool convs of the training path (input always bfloat16)
        if constexpr (SINGLE) {
            if (p.Cin > 128 || p.in2 || p.in_map || !(io & 1)) return io16_unsupported("conv_pool_f16q", p);
            if (io == 3) hipLaunchKernelGGL((conv_pool_f16q_kernel<NT, true, false, 3>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((conv_pool_f16q_kernel<NT, true, false, 1>), grid, dim3(256), 0, s, p);
        } else return io16_unsupported("conv_pool_f16q", p);
    } else
    if (nm_ls().pool_q && p.Cin <= 128) {                           // (the affine table holds 128 channels)
        if (p.in2) hipLaunchKernelGGL((conv_pool_f16q_kernel<NT, SINGLE, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_pool_f16q_kernel<NT, SINGLE, false>), grid, dim3(256), 0, s, p);
    } else
    hipLaunchKernelGGL((conv_pool_f16s_kernel<NT, SINGLE>), grid, dim3(256), 0, s, p);
    if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
    return nm_check_hip(hipGetLastError(), "conv_pool_f16s launch");
}
template <int NT>
int launch_pool_f16s(const ConvParams& p, dim3 grid, hipStream_t s) {
    return nm_ls().single ? launch_pool_f16s_impl<NT, true>(p, grid, s) : launch_pool_f16s_impl<NT, false>(p, grid, s);
}

int g_num_cus = 0;

template <bool UP2, bool SINGLE, int IO = 0, int LATE = 1>
int launch_f16p_impl(const ConvParams& p_in, size_t lds_bytes, int work_items, hipStream_t s) {
    ConvParams p = p_in;
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16p_kernel<UP2, SINGLE, IO, LATE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(conv_f16p)");
        attr_set.mark();
    }
    if (g_num_cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return nm_check_hip(hipErrorUnknown, "device query");
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    ProfRec rec;
    rec.flops = 2.0 * p.N * (double)p.OD * p.OH * p.OW * p.Cout * (double)p.cin_real * 27.0;
    const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
    if (prof_rec) {
        rec.a = prof_event(); rec.b = prof_event(); rec.variant = 7;
        (void)hipEventRecord(rec.a, s);
    }
    // persistent: one workgroup per CU (NM355_CONV_WGS caps the count: the co-residency A/B of DESIGN 5 - CUs left to the other queues)
    dim3 grid((unsigned)min(work_items, nm_ls().conv_wgs > 0 ? min(nm_ls().conv_wgs, g_num_cus) : g_num_cus));
    if (p.Cout == 32) choose_super_tile(p, (int)grid.x, p.OD / 4, p.OH / 8, p.OW / 8);   // (one cout group per brick)
    hipLaunchKernelGGL((conv_f16p_kernel<UP2, SINGLE, IO, LATE>), grid, dim3(512), lds_bytes, s, p);
    if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
    return nm_check_hip(hipGetLastError(), "conv_f16p launch");
}
template <bool UP2>
int launch_f16p(const ConvParams& p, size_t lds_bytes, int work_items, hipStream_t s) {
    const int io = (p.in_h ? 1 : 0) | (p.out_h ? 2 : 0);
    if (io) {                                                       // 16-bit storage: both tensors bfloat16 (layers at >= 32^3 and their data gradients)
        if (io != 3 || !nm_ls().single) return io16_unsupported("conv_f16p", p);
        return launch_f16p_impl<UP2, true, 3>(p, lds_bytes, work_items, s);
    }
    if (nm_ls().f16p_late == 2) return launch_f16p_impl<UP2, false, 0, 2>(p, lds_bytes, work_items, s);
    if (!nm_ls().f16p_late)      // (A/B: the round-2 producer schedule)
        return nm_ls().single ? launch_f16p_impl<UP2, true, 0, 0>(p, lds_bytes, work_items, s) : launch_f16p_impl<UP2, false, 0, 0>(p, lds_bytes, work_items, s);
    return nm_ls().single ? launch_f16p_impl<UP2, true>(p, lds_bytes, work_items, s) : launch_f16p_impl<UP2, false>(p, lds_bytes, work_items, s);
}

template <bool UP2, bool SINGLE, int IO = 0, int WEMU = 0, bool DEFER = false>
int launch_f16p2_impl(const ConvParams& p_in, size_t lds_bytes, int work_items, hipStream_t s) {
    ConvParams p = p_in;
#ifdef NM_Q2_DIAG
    if constexpr (WEMU == 0 && !UP2 && !SINGLE && IO == 0) {       // diagnostic build: the Winograd resource-profile emulation (wrong results)
        static const int wemu = getenv("NM355_P2_WEMU") ? atoi(getenv("NM355_P2_WEMU")) : 0;
        if (wemu == 1) return launch_f16p2_impl<UP2, SINGLE, IO, 1>(p_in, lds_bytes, work_items, s);
        if (wemu == 3) return launch_f16p2_impl<UP2, SINGLE, IO, 3>(p_in, lds_bytes, work_items, s);
    }
#endif
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16p2_kernel<UP2, SINGLE, IO, WEMU, DEFER>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(conv_f16p2)");
        attr_set.mark();
    }
    if (g_num_cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return nm_check_hip(hipErrorUnknown, "device query");
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    ProfRec rec;
    rec.flops = 2.0 * p.N * (double)p.OD * p.OH * p.OW * p.Cout * (double)p.cin_real * 27.0;
    const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
    if (prof_rec) {
        rec.a = prof_event(); rec.b = prof_event(); rec.variant = 9;
        (void)hipEventRecord(rec.a, s);
    }
    // persistent: one workgroup per CU (NM355_CONV_WGS caps the count: the co-residency A/B of DESIGN 5 - CUs left to the other queues)
    dim3 grid((unsigned)min(work_items, nm_ls().conv_wgs > 0 ? min(nm_ls().conv_wgs, g_num_cus) : g_num_cus));
    if (p.Cout == 64) choose_super_tile(p, (int)grid.x, p.OD / 4, p.OH / 8, p.OW / 8);   // (one cout group per brick)
    hipLaunchKernelGGL((conv_f16p2_kernel<UP2, SINGLE, IO, WEMU, DEFER>), grid, dim3(512), lds_bytes, s, p);
    if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
    return nm_check_hip(hipGetLastError(), "conv_f16p2 launch");
}
// the one-product modes' own kernel (conv_f16q2_kernel); the caller has checked conv mode 3 / 4
template <int IO, int DBG = 0>
int launch_f16q2_impl(const ConvParams& p_in, int work_items, hipStream_t s) {
    ConvParams p = p_in;
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16q2_kernel<IO, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(conv_f16q2)");
        attr_set.mark();
    }
    if (g_num_cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return nm_check_hip(hipErrorUnknown, "device query");
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const size_t lds_bytes = (size_t)4 * 600 * 16 + (size_t)2 * 27 * 2 * 64 * 16 + (size_t)(512 + p.Cout) * sizeof(float);
    if (lds_bytes > 160 * 1024) { nm_set_error("conv_f16q2: %d output channels exceed the LDS budget", p.Cout); return NM_ERR_UNSUPPORTED; }
    ProfRec rec;
    rec.flops = 2.0 * p.N * (double)p.OD * p.OH * p.OW * p.Cout * (double)p.cin_real * 27.0;
    const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
    if (prof_rec) {
        rec.a = prof_event(); rec.b = prof_event(); rec.variant = 14;
        (void)hipEventRecord(rec.a, s);
    }
    // persistent: one workgroup per CU (NM355_CONV_WGS caps the count: the co-residency A/B of DESIGN 5 - CUs left to the other queues)
    dim3 grid((unsigned)min(work_items, nm_ls().conv_wgs > 0 ? min(nm_ls().conv_wgs, g_num_cus) : g_num_cus));
    if (p.Cout == 64) choose_super_tile(p, (int)grid.x, p.OD / 4, p.OH / 8, p.OW / 8);   // (one cout group per brick)
    hipLaunchKernelGGL((conv_f16q2_kernel<IO, DBG>), grid, dim3(512), lds_bytes, s, p);
    if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
    return nm_check_hip(hipGetLastError(), "conv_f16q2 launch");
}
// the one-product modes' kernel for the 32-output-channel layers (conv_f16r_kernel: resident weights); Cin = 16 C16T
template <int IO, int C16T>
int launch_f16r_impl(const ConvParams& p_in, int work_items, hipStream_t s) {
    ConvParams p = p_in;
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16r_kernel<IO, C16T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(conv_f16r)");
        attr_set.mark();
    }
    if (g_num_cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return nm_check_hip(hipErrorUnknown, "device query");
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const size_t lds_bytes = (size_t)4 * 600 * 16 + (size_t)27 * C16T * 64 * 16 + (size_t)(512 + 32) * sizeof(float) + 2048 + 256 + 1024;      // (+ the store slots of the lanes that hold no total)
    ProfRec rec;
    rec.flops = 2.0 * p.N * (double)p.OD * p.OH * p.OW * p.Cout * (double)p.cin_real * 27.0;
    const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
    if (prof_rec) {
        rec.a = prof_event(); rec.b = prof_event(); rec.variant = 15;
        (void)hipEventRecord(rec.a, s);
    }
    dim3 grid((unsigned)min(work_items, nm_ls().conv_wgs > 0 ? min(nm_ls().conv_wgs, g_num_cus) : g_num_cus));
    choose_super_tile(p, (int)grid.x, p.OD / 4, p.OH / 8, p.OW / 8);
    hipLaunchKernelGGL((conv_f16r_kernel<IO, C16T>), grid, dim3(512), lds_bytes, s, p);
    if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
    return nm_check_hip(hipGetLastError(), "conv_f16r launch");
}
template <bool UP2>
int launch_f16p2(const ConvParams& p, size_t lds_bytes, int work_items, hipStream_t s) {
    const int io = (p.in_h ? 1 : 0) | (p.out_h ? 2 : 0);
    if constexpr (!UP2) {
        if (nm_ls().single && nm_ls().f16q2 && (io == 0 || io == 3) && p.Cout <= 1024) {
#ifdef NM_Q2_DIAG
            static const int dbg = getenv("NM355_Q2_DBG") ? atoi(getenv("NM355_Q2_DBG")) : 0;      // ablations of tools/diag_f16q2.py (diagnostic build only)
            if (io == 3) switch (dbg) {
                case 1: return launch_f16q2_impl<3, 1>(p, work_items, s); case 2: return launch_f16q2_impl<3, 2>(p, work_items, s);
                case 3: return launch_f16q2_impl<3, 3>(p, work_items, s); case 4: return launch_f16q2_impl<3, 4>(p, work_items, s);
                case 5: return launch_f16q2_impl<3, 5>(p, work_items, s); case 6: return launch_f16q2_impl<3, 6>(p, work_items, s);
                case 7: return launch_f16q2_impl<3, 7>(p, work_items, s); case 8: return launch_f16q2_impl<3, 8>(p, work_items, s);
                case 9: return launch_f16q2_impl<3, 9>(p, work_items, s); default: break;
            }
#endif
            return io ? launch_f16q2_impl<3>(p, work_items, s) : launch_f16q2_impl<0>(p, work_items, s);
        }
    }
    if (io) {                                                       // 16-bit storage: both tensors bfloat16 (layers at >= 32^3 and their data gradients)
        if (io != 3 || !nm_ls().single) return io16_unsupported("conv_f16p2", p);
        return launch_f16p2_impl<UP2, true, 3>(p, lds_bytes, work_items, s);
    }
    if (nm_ls().single) return launch_f16p2_impl<UP2, true>(p, lds_bytes, work_items, s);
    if (nm_ls().p2_defer) return launch_f16p2_impl<UP2, false, 0, 0, true>(p, lds_bytes, work_items, s);      // (A/B: the deferred epilogue, slower)
    return launch_f16p2_impl<UP2, false>(p, lds_bytes, work_items, s);
}

#ifdef NM_DIAG
unsigned long long* g_stamps = nullptr;
#endif

}  // namespace

#ifdef NM_DIAG
extern "C" void nm_diag_set_stamps(void* p) { g_stamps = static_cast<unsigned long long*>(p); }
#endif
void nm_conv_set_mode(int mode) { NmLaunchState& l = nm_ls(); l.conv_mode = mode ? 1 : 0; l.f16p_all = mode == 2; l.single = mode == 3 || mode == 4; l.store16 = mode == 4; }
int nm_conv_get_mode() { const NmLaunchState& l = nm_ls(); return l.conv_mode && l.single ? (l.store16 ? 4 : 3) : (l.conv_mode && l.f16p_all ? 2 : l.conv_mode); }
int nm_conv_single() { return nm_ls().conv_mode && nm_ls().single; }

int nm_launch_pack_conv_weight16(const float* w, int Cout, int Cin, int ks, void* packed, int Co_pad, hipStream_t s) {
    if (Co_pad % 32 || Cout > Co_pad) { nm_set_error("pack_conv_weight16: bad padding Cout=%d/%d", Cout, Co_pad); return NM_ERR_ARG; }   // (channels beyond Cin: zero)
    size_t total = (size_t)ks * ks * ks * ((Cin + 15) / 16) * 2 * Co_pad * 8;
    int blocks = (int)min((total + 255) / 256, (size_t)2048);
    hipLaunchKernelGGL(pack_conv_weight16_kernel, dim3(blocks), dim3(256), 0, s, w, Cout, Cin, ks, reinterpret_cast<_Float16*>(packed), Co_pad);
    return nm_check_hip(hipGetLastError(), "pack_conv_weight16 launch");
}

__global__ __launch_bounds__(256) void pack_jobs_kernel(const NmPackJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;                           // the job whose block range holds blockIdx.x
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const NmPackJob j = jobs[lo];
    const int lb = blockIdx.x - j.blk0, taps = j.ks * j.ks * j.ks;
    auto W = [&](int co, int ci, int tap) -> float {
        if (co >= j.Cout || ci >= j.Cin) return 0.f;
        // (a layer whose channel count is padded inside the library: the source tensor has fewer rows / columns than the packed form)
        if (j.src_rows > 0 && ((j.flip ? ci : co) >= j.src_rows || (j.flip ? co : ci) >= j.src_cin)) return 0.f;
        return j.flip ? j.src[((size_t)ci * j.src_cin + co) * taps + (taps - 1 - tap)] : j.src[((size_t)co * j.src_cin + ci) * taps + tap];
    };
    const size_t total32 = (size_t)taps * j.Cin_pad * j.Co_pad;
    for (size_t i = (size_t)lb * 256 + threadIdx.x; i < total32; i += (size_t)j.nblk * 256) {         // pack_conv_weight_kernel's layout
        const int e = i & 3; size_t r = i >> 2;
        const int co = r % j.Co_pad; r /= j.Co_pad;
        const int q = r % (j.Cin_pad / 4), tap = r / (j.Cin_pad / 4);
        j.wp[i] = W(co, q * 4 + e, tap);
    }
    if (j.wp16) {                                                                                      // pack_conv_weight16_kernel's layout
        _Float16* packed = reinterpret_cast<_Float16*>(j.wp16);
        const int C16 = (j.Cin + 15) >> 4;
        const size_t total16 = (size_t)taps * C16 * 2 * j.Co_pad * 8;
        for (size_t i = (size_t)lb * 256 + threadIdx.x; i < total16; i += (size_t)j.nblk * 256) {
            const int e = i & 7; size_t r = i >> 3;
            const int co = r % j.Co_pad; r /= j.Co_pad;
            const int hh = r & 1; r >>= 1;
            const int cb = r % C16, tap = r / C16;
            const float v = W(co, cb * 16 + hh * 8 + e, tap);
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)((v - (float)h) * NM_SPLIT_SCALE);
            const size_t base = ((((size_t)tap * C16 + cb) * 4) * j.Co_pad) * 8;
            packed[base + ((size_t)hh * j.Co_pad + co) * 8 + e] = h;
            packed[base + ((size_t)(2 + hh) * j.Co_pad + co) * 8 + e] = l;
        }
    }
}

int nm_pack_job_blocks(const NmPackJob& j) {
    const size_t total = (size_t)j.ks * j.ks * j.ks * j.Cin_pad * j.Co_pad;
    return (int)min((total + 255) / 256, (size_t)64);
}

int nm_launch_pack_jobs(const NmPackJob* device_jobs, int njobs, int total_blocks, hipStream_t s) {
    if (njobs <= 0) return NM_OK;
    hipLaunchKernelGGL(pack_jobs_kernel, dim3(total_blocks), dim3(256), 0, s, device_jobs, njobs);
    return nm_check_hip(hipGetLastError(), "pack_jobs launch");
}

size_t nm_packed_weight_floats(int ks, int Cin_pad, int Co_pad) {
    return (size_t)ks * ks * ks * Cin_pad * Co_pad;
}

int nm_launch_pack_conv_weight(const float* w, int Cout, int Cin, int ks, float* packed, int Cin_pad,
                               int Co_pad, hipStream_t s) {
    if (Cin_pad % 8 || Co_pad % 32 || Cin > Cin_pad || Cout > Co_pad) {
        nm_set_error("pack_conv_weight: bad padding Cin=%d/%d Cout=%d/%d", Cin, Cin_pad, Cout, Co_pad);
        return NM_ERR_ARG;
    }
    size_t total = nm_packed_weight_floats(ks, Cin_pad, Co_pad);
    int blocks = (int)min((total + 255) / 256, (size_t)2048);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, s, w, Cout, Cin, ks, packed, Cin_pad, Co_pad);
    return nm_check_hip(hipGetLastError(), "pack_conv_weight launch");
}

static bool use_up2c(const ConvGeom& g, int Cin) {
    return g.up2 && g.up2c && nm_ls().conv_mode == 1 && nm_up2c_eligible(g.OD / 2, g.OH / 2, g.OW / 2, Cin, g.Cout, g.ks, g.stride, g.pad);
}

bool nm_conv_pool16_eligible(int Cin, int OD, int OH, int OW, bool have_w16) {
    if (!(nm_ls().conv_mode == 1 && nm_ls().pool16 && have_w16 && Cin % 16 == 0)) return false;
    ConvGeom g; g.ks = 2; g.stride = 2; g.pad = 0; g.OD = OD; g.OH = OH; g.OW = OW; g.Cout = Cin; g.Co_pad = (Cin + 31) & ~31;
    const Tiling t = choose_tiling(g, Cin);
    return t.MT == 2 && t.bx_l2 == 3 && t.by_l2 == 3 && t.bz_l2 == 2;
}

int nm_conv_blocks_per_frame(const ConvGeom& g, int Cin) {
    if (use_up2c(g, Cin)) return nm_up2c_blocks_per_frame(g.OD / 2, g.OH / 2, g.OW / 2);
    Tiling t = choose_tiling(g, 16);
    return t.nbz * t.nby * t.nbx;
}

void nm_conv_prof_enable(int on, hipStream_t stream) {
    NmLaunchState& l = nm_ls(); l.prof_on = on != 0; l.prof_all = on == 2; l.prof_stream = stream;
    l.prof_min_flops = on == 3 ? 2e10 : 0.0;       // 3: only launches that can matter to a roofline (>= 20 GFLOP algorithmic): an event pair costs
                                                   // the many small dependent launches of the hourglass levels more than it tells about them
}

// Sums the event-timed launches of one kernel variant recorded since the last reset.
// variant: see nm_prof_kernel_name.  Synchronises on the recorded events.
int nm_conv_prof_collect(int variant, double* ms_total, double* flops_total, long long* launches) {
    double ms = 0.0, fl = 0.0; long long n = 0;
    for (const ProfRec& r : nm_ls().prof) {
        if (r.variant != variant) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return NM_ERR_HIP;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return NM_ERR_HIP;
        ms += t; fl += r.flops; ++n;
    }
    *ms_total = ms; *flops_total = fl; *launches = n;
    return NM_OK;
}

NmProfScope::NmProfScope(hipStream_t stream, double flops, int variant) : s(stream) {
    on = flops > 0.0 && NM_PROF_ON(stream) && flops >= nm_ls().prof_min_flops;
    if (!on) return;
    rec.a = prof_event(); rec.b = prof_event(); rec.variant = variant; rec.flops = flops;
    (void)hipEventRecord(rec.a, s);
}
NmProfScope::~NmProfScope() {
    if (!on) return;
    (void)hipEventRecord(rec.b, s);
    nm_ls().prof.push_back(rec);
}

void nm_conv_prof_reset() {
    NmLaunchState& l = nm_ls();
    for (const ProfRec& r : l.prof) { l.event_pool.push_back(r.a); l.event_pool.push_back(r.b); }
    l.prof.clear();
}

int nm_launch_conv(const TensorRef& in, const float* w_packed, const float* bias, float* out,
                   const ConvGeom& g, float* part, hipStream_t s, int cin_real, const void* w_packed16, int out_h) {
    if (in.C % 8 != 0 || g.Co_pad % 32 != 0 || g.Cout > g.Co_pad || g.Cout <= 0) {
        nm_set_error("conv: unsupported channels Cin=%d Cout=%d Co_pad=%d", in.C, g.Cout, g.Co_pad);
        return NM_ERR_ARG;
    }
    const int us = g.up2 ? 2 : 1;
    if ((us * in.D + 2 * g.pad - g.ks) / g.stride + 1 != g.OD || (us * in.H + 2 * g.pad - g.ks) / g.stride + 1 != g.OH ||
        (us * in.W + 2 * g.pad - g.ks) / g.stride + 1 != g.OW) {
        nm_set_error("conv: geometry mismatch in=(%d,%d,%d) k=%d s=%d p=%d out=(%d,%d,%d)", in.D, in.H, in.W,
                     g.ks, g.stride, g.pad, g.OD, g.OH, g.OW);
        return NM_ERR_ARG;
    }
    if ((in.scale == nullptr) != (in.shift == nullptr)) { nm_set_error("conv: scale/shift must come together"); return NM_ERR_ARG; }
    if ((in.brickmap || in.p2) && g.up2) { nm_set_error("conv: a brick-sparse input tensor can only feed the k2 s2 split-fp16 pool kernel"); return NM_ERR_ARG; }
    if (use_up2c(g, in.C)) {
        // the fused-upsample layers on the coarse grid with composite weights (nm_up2c.hip): main + shell launch, timed together
        ProfRec rec;
        rec.flops = 2.0 * in.N * (double)g.OD * g.OH * g.OW * g.Cout * (double)(cin_real > 0 ? cin_real : in.C) * 27.0;
        const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
        if (prof_rec) {
            rec.a = prof_event(); rec.b = prof_event(); rec.variant = 12;
            (void)hipEventRecord(rec.a, s);
        }
        const int rc = nm_launch_conv_up2c(in, g.up2c, bias, out, g.Cout, g.Co_pad, part, s, out_h);
        if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
        return rc;
    }
    Tiling t = choose_tiling(g, in.C);
    if (t.lds_bytes > 160 * 1024) { nm_set_error("conv: LDS tile %zu B too large", t.lds_bytes); return NM_ERR_UNSUPPORTED; }
    ConvParams p;
    p.in = in.p; p.in_scale = in.scale; p.in_shift = in.shift; p.in_slope = in.slope;
    p.N = in.N; p.ID = in.D; p.IH = in.H; p.IW = in.W; p.Cin = in.C;
    p.w = w_packed; p.bias = bias; p.out = out; p.part = part;
    p.OD = g.OD; p.OH = g.OH; p.OW = g.OW; p.Cout = g.Cout; p.Co_pad = g.Co_pad;
    p.ks = g.ks; p.stride = g.stride; p.pad = g.pad;
    p.bz_l2 = t.bz_l2; p.by_l2 = t.by_l2; p.bx_l2 = t.bx_l2; p.nbz = t.nbz; p.nby = t.nby; p.nbx = t.nbx;
    p.cin_real = cin_real > 0 ? cin_real : in.C;
    p.up2 = g.up2 ? 1 : 0;
    p.st_z = p.st_y = p.st_x = 0;
    p.st_sx = p.st_sy = p.st_pf = p.st_m_sx = p.st_m_sy = p.st_m_pf = 0;
    p.in_alt = in.alt; p.in_map = in.brickmap;
    p.in2 = in.p2; p.in2_scale = in.scale2; p.in2_shift = in.shift2; p.in2_slope = in.slope2;
    p.wdma = nm_ls().f16p_dma;
    p.in_h = in.h; p.out_h = out_h;
    if (in.p2 && !(nm_conv_pool16_eligible(in.C, g.OD, g.OH, g.OW, w_packed16 != nullptr) && g.ks == 2 && g.stride == 2 && g.pad == 0 && !g.up2 && !in.brickmap)) {
        nm_set_error("conv: an un-materialised residual sum can only feed the k2 s2 split-fp16 pool kernel"); return NM_ERR_ARG;
    }
    if (in.brickmap && !(nm_conv_pool16_eligible(in.C, g.OD, g.OH, g.OW, w_packed16 != nullptr) && g.ks == 2 && g.stride == 2 && g.pad == 0 && !g.up2 &&
                         in.alt && in.D % 4 == 0 && in.H % 8 == 0 && in.W % 8 == 0)) {
        nm_set_error("conv: a brick-sparse input tensor can only feed the k2 s2 split-fp16 pool kernel"); return NM_ERR_ARG;
    }
    p.w16 = (nm_ls().conv_mode == 1 && nm_ls().small16 && w_packed16 && in.C % 8 == 0 && !g.up2 && t.MT == 1 && (t.KC % 16 == 0 || t.KC == in.C)) ? w_packed16 : nullptr;
    if (p.w16) t.lds_bytes = max(t.lds_bytes, (size_t)(((t.KC + 15) & ~15) / 4) * (t.HVp + t.CVp) * 16);
    p.ksplit = 1;
    if (p.w16 && nm_ls().ksplit) {       // tiny volumes: how many of the 4 row tiles (32 rows each) of the brick hold voxels at all
        const int top = ((min(g.OD, 1 << t.bz_l2) - 1) << (t.bx_l2 + t.by_l2)) + ((min(g.OH, 1 << t.by_l2) - 1) << t.bx_l2) + min(g.OW, 1 << t.bx_l2) - 1;
        const int rw = top / 32 + 1;
        p.ksplit = rw == 1 ? 4 : (rw == 2 ? 2 : 1);
        if (p.ksplit > 1) t.lds_bytes = max(t.lds_bytes, (size_t)4 * t.NT * 16 * 64 * sizeof(float));
    }
#ifdef NM_DIAG
    p.stamps = g_stamps;
#endif
    p.KC = t.KC; p.HZ = t.HZ; p.HY = t.HY; p.HX = t.HX; p.HV = t.HV; p.HVp = t.HVp; p.CVp = t.CVp; p.ZP = t.HY * t.HX;
    dim3 grid((unsigned)(in.N * t.nbz * t.nby * t.nbx), (unsigned)(g.Co_pad / (t.NT * 32)));
    if (nm_ls().conv_mode == 1 && nm_ls().pool16 && w_packed16 && in.C % 16 == 0 && g.ks == 2 && g.stride == 2 && g.pad == 0 && !g.up2 &&
        t.MT == 2 && t.bx_l2 == 3 && t.by_l2 == 3 && t.bz_l2 == 2) {
        p.w = static_cast<const float*>(w_packed16);
        return t.NT == 2 ? launch_pool_f16s<2>(p, grid, s) : launch_pool_f16s<1>(p, grid, s);
    }
    if (nm_ls().conv_mode == 1 && nm_ls().f16p2 && w_packed16 && in.C % 16 == 0 && g.ks == 3 && g.stride == 1 && g.pad == 1 && !g.up2 &&
        g.OD % 4 == 0 && g.OH % 8 == 0 && g.OW % 8 == 0 && g.OD >= 16 && g.Cout % 64 == 0) {
        const int work = in.N * (g.OD / 4) * (g.OH / 8) * (g.OW / 8) * (g.Cout / 64);
        p.w = static_cast<const float*>(w_packed16);
        const size_t lds_bytes = (size_t)8 * 600 * 16 + (size_t)2 * 9 * 4 * 64 * 16 + (size_t)(512 + g.Cout) * sizeof(float);
        return launch_f16p2<false>(p, lds_bytes, work, s);
    }
    if (nm_ls().conv_mode == 1 && nm_ls().single && nm_ls().f16r && nm_ls().f16p && w_packed16 && (in.C == 32 || in.C == 64) && g.ks == 3 && g.stride == 1 &&
        g.pad == 1 && !g.up2 && g.OD % 4 == 0 && g.OH % 8 == 0 && g.OW % 8 == 0 && g.OD >= 16 && g.Cout == 32 && (p.in_h != 0) == (p.out_h != 0)) {
        // one-product modes, 32 output channels: the layer's weights stay in LDS (conv_f16r_kernel)
        const int work = in.N * (g.OD / 4) * (g.OH / 8) * (g.OW / 8);
        p.w = static_cast<const float*>(w_packed16);
        if (in.C == 32) return p.in_h ? launch_f16r_impl<3, 2>(p, work, s) : launch_f16r_impl<0, 2>(p, work, s);
        return p.in_h ? launch_f16r_impl<3, 4>(p, work, s) : launch_f16r_impl<0, 4>(p, work, s);
    }
    if (nm_ls().conv_mode == 1 && nm_ls().f16p && w_packed16 && in.C % 16 == 0 && g.ks == 3 && g.stride == 1 && g.pad == 1 && !g.up2 &&
        g.OD % 4 == 0 && g.OH % 8 == 0 && g.OW % 8 == 0 && g.OD >= 16 && g.Cout % 32 == 0 && (nm_ls().f16p == 1 || nm_ls().f16p_all || g.Cout == 32)) {
        const int work = in.N * (g.OD / 4) * (g.OH / 8) * (g.OW / 8) * (g.Cout / 32);
        p.w = static_cast<const float*>(w_packed16);
        const size_t lds_bytes = (size_t)8 * 600 * 16 + (size_t)3 * 9 * 4 * 32 * 16 + (size_t)(256 + g.Cout) * sizeof(float);
        return launch_f16p<false>(p, lds_bytes, work, s);
    }
    if (nm_ls().conv_mode == 1 && w_packed16 && in.C % 16 == 0 && t.MT == 2 && t.bx_l2 == 3 && t.by_l2 == 3 && t.bz_l2 == 2 &&
        g.stride == 1 && (g.ks == 1 || g.ks == 3) && t.HZ == g.ks + 3 && t.HY * t.HX * 2 <= 256) {
        // halo planes padded to a pitch of 4 (mod 16) 16-B slots: conflict-free A reads (see conv_f16s_kernel)
        Tiling t16 = t;
        const int area = t.HY * t.HX;
        p.ZP = area + ((4 - (area & 15)) & 15);
        const int hv = t.HZ * p.ZP;
        p.HVp = hv + ((2 - (hv & 7)) & 7);
        const bool blds = (t.NT == 1 && g.ks == 3);
        const size_t bbytes = blds ? (size_t)2 * 9 * 4 * 32 * 16 : 0;          // two 9-tap weight groups (Cout = 32 layers)
        t16.lds_bytes = max(blds ? (size_t)4 * p.HVp * 16 + bbytes : (size_t)4 * (p.HVp + t.CVp) * 16, (size_t)4 * 64 * 2 * sizeof(float));
        if (t16.lds_bytes <= 80 * 1024) {
            p.w = static_cast<const float*>(w_packed16);
            if (g.ks == 3 && g.up2) return t.NT == 2 ? launch_f16s<2, 2, 3, true>(p, t16, grid, s) : launch_f16s<2, 1, 3, true>(p, t16, grid, s);
            if (g.ks == 3) return t.NT == 2 ? launch_f16s<2, 2, 3, false>(p, t16, grid, s) : launch_f16s<2, 1, 3, false>(p, t16, grid, s);
            if (!g.up2) return t.NT == 2 ? launch_f16s<2, 2, 1, false>(p, t16, grid, s) : launch_f16s<2, 1, 1, false>(p, t16, grid, s);
        }
        p.HVp = t.HVp;
    }
    if (p.in_h || p.out_h) return io16_unsupported("conv (fp32 MFMA kernel)", p);
    if (t.MT == 2 && t.NT == 2) return launch_t<2, 2>(p, t, grid, s);
    if (t.MT == 2 && t.NT == 1) return launch_t<2, 1>(p, t, grid, s);
    if (t.MT == 1 && t.NT == 2) return launch_t<1, 2>(p, t, grid, s);
    return launch_t<1, 1>(p, t, grid, s);
}

int nm_occ_blocks_per_frame(int G) { return (G / 8) * (G / 8) * (G / 4); }

// packs the occupancy-channel taps of a (Cout,4,5,5,5) weight into [32 tap-quads][Co_pad][4] fp32 followed by the split-fp16
// form (packed: 256 * Co_pad floats); tmp: Cout*125 floats
int nm_launch_pack_occ_weight(const float* w_oidhw, int Cout, float* tmp, float* packed, int Co_pad, hipStream_t s) {
    hipLaunchKernelGGL(extract_occ_weight_kernel, dim3((Cout * 125 + 255) / 256), dim3(256), 0, s, w_oidhw, Cout, tmp);
    int rc = nm_check_hip(hipGetLastError(), "extract_occ_weight launch");
    if (rc) return rc;
    rc = nm_launch_pack_conv_weight(tmp, Cout, 125, 1, packed, 128, Co_pad, s);
    if (rc) return rc;
    // second half of the buffer (another 128 * Co_pad floats): the split-fp16 form for conv_k5occ_f16_kernel
    hipLaunchKernelGGL(pack_occ_weight16_kernel, dim3(64), dim3(256), 0, s, tmp, Cout, Co_pad, reinterpret_cast<_Float16*>(packed + (size_t)128 * Co_pad));
    return nm_check_hip(hipGetLastError(), "pack_occ_weight16 launch");
}

int nm_launch_conv_k5occ(const float* occ, int N, int G, const float* w_packed, const float* field, float* out, int Cout,
                         int Co_pad, float* part, hipStream_t s, unsigned char* brickmap, const float* field_part, unsigned char* flags, int out_h) {
    if (G % 8 || Co_pad % 32 || Cout > Co_pad) { nm_set_error("conv_k5occ: unsupported G=%d Cout=%d", G, Cout); return NM_ERR_ARG; }
    OccParams p; p.occ = occ; p.w = w_packed; p.field = field; p.out = out; p.part = part; p.N = N; p.G = G; p.Cout = Cout; p.Co_pad = Co_pad;
    p.brickmap = brickmap; p.field_part = field_part; p.flags = nullptr; p.row_walk = 0;
    if (brickmap && flags && G <= 128 && G % 8 == 0 && nm_ls().occ_flags) {
        hipLaunchKernelGGL(occ_brick_flags_kernel, dim3((unsigned)(N * (G >> 2))), dim3(256), 0, s, occ, G, flags);
        p.flags = flags; p.row_walk = nm_ls().occ_flags >= 2 ? 1 : 0;
    }
    if (brickmap && (!(nm_ls().conv_mode == 1 && nm_ls().occ16) || (part && !field_part))) {
        nm_set_error("conv_k5occ: the brick-sparse output exists on the split-fp16 kernel only and needs the field's partial sums"); return NM_ERR_ARG;
    }
    const int NT = (Co_pad % 64 == 0) ? 2 : 1;
    dim3 grid((unsigned)(N * nm_occ_blocks_per_frame(G)), (unsigned)(Co_pad / (NT * 32)));
    if (p.row_walk) grid.x /= (unsigned)(G >> 3);         // one workgroup per x-row of bricks (conv_k5occ_f16_kernel)
    ProfRec rec;
    rec.flops = 2.0 * N * (double)G * G * G * Cout * 4.0 * 125.0;   // the reference's dense k5 layer over 4 input channels
    const bool prof_rec = NM_PROF_ON(s) && rec.flops >= nm_ls().prof_min_flops;
    if (prof_rec) {
        rec.a = prof_event(); rec.b = prof_event(); rec.variant = 4;
        (void)hipEventRecord(rec.a, s);
    }
    if (out_h) {                                                    // 16-bit storage: the dense training form on the f16 MFMA kernel only
        if (!(nm_ls().conv_mode == 1 && nm_ls().occ16 && nm_ls().single) || brickmap || Cout % 4) {
            nm_set_error("conv_k5occ: bfloat16 output needs conv mode 4, the dense (training) form and Cout %% 4 == 0"); return NM_ERR_UNSUPPORTED;
        }
        if (NT == 2) hipLaunchKernelGGL((conv_k5occ_f16_kernel<2, true, false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_k5occ_f16_kernel<1, true, false, true>), grid, dim3(256), 0, s, p);
    } else
    if (nm_ls().conv_mode == 1 && nm_ls().occ16 && p.row_walk) {
        if (NT == 2) { if (nm_ls().single) hipLaunchKernelGGL((conv_k5occ_f16_kernel<2, true, true>), grid, dim3(256), 0, s, p); else hipLaunchKernelGGL((conv_k5occ_f16_kernel<2, false, true>), grid, dim3(256), 0, s, p); }
        else { if (nm_ls().single) hipLaunchKernelGGL((conv_k5occ_f16_kernel<1, true, true>), grid, dim3(256), 0, s, p); else hipLaunchKernelGGL((conv_k5occ_f16_kernel<1, false, true>), grid, dim3(256), 0, s, p); }
    } else if (nm_ls().conv_mode == 1 && nm_ls().occ16) {
        if (NT == 2) { if (nm_ls().single) hipLaunchKernelGGL((conv_k5occ_f16_kernel<2, true>), grid, dim3(256), 0, s, p); else hipLaunchKernelGGL((conv_k5occ_f16_kernel<2>), grid, dim3(256), 0, s, p); }
        else { if (nm_ls().single) hipLaunchKernelGGL((conv_k5occ_f16_kernel<1, true>), grid, dim3(256), 0, s, p); else hipLaunchKernelGGL((conv_k5occ_f16_kernel<1>), grid, dim3(256), 0, s, p); }
    } else if (NT == 2) hipLaunchKernelGGL((conv_k5occ_kernel<2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_k5occ_kernel<1>), grid, dim3(256), 0, s, p);
    if (prof_rec) { (void)hipEventRecord(rec.b, s); nm_ls().prof.push_back(rec); }
    return nm_check_hip(hipGetLastError(), "conv_k5occ launch");
}
