// conv3(upsample2(x)) evaluated on the coarse grid: the decoder's two resolution-doubling layers
// (model/kypt_detector.py:425-447: nn.Upsample(x2, trilinear, align_corners=False) followed by Conv3d(k3, p1)).
//
// Trilinear x2 is linear and, away from the volume border, periodic with period 2: a fine output voxel o = 2i + p (parity
// p per axis) reads the three coarse voxels i-1, i, i+1 along each axis.  Per axis the fine taps t = -1, 0, +1 of the conv
// land on coarse offsets d = -1, 0, +1 with the weights
//      p = 0:  t=-1 -> (.75, .25, 0)    t=0 -> (.25, .75, 0)    t=+1 -> (0, .75, .25)
//      p = 1:  t=-1 -> (.25, .75, 0)    t=0 -> (0, .75, .25)    t=+1 -> (0, .25, .75)
// so  out[2i+p] = sum_{d, c} C_p[d][c] * a[i+d][c]  with the composite weights  C_p[d] = sum_t Uz[pz][tz][dz] Uy[..] Ux[..] w[t]:
// eight 3x3x3 convolutions of the ACTIVATED COARSE tensor, one per parity class, the same MAC count as the fine convolution and
// no interpolation arithmetic at all.  The coarse halo tile of a brick is 8x smaller than the fine one: all 64 input channels of
// a (2+2) x (8+2) x (8+2) coarse tile sit in LDS at once (split fp16 hi/lo, 104 KB), staged once per brick, and the 864 k-steps of
// the brick's eight parity classes run from it without a barrier.
//
// Machine mapping: one 512-thread workgroup per CU, wave w = parity class (pz,py,px) = bits of w, 4 MFMA row tiles per wave
// (2z x 8y x 8x coarse cells = 128 rows x 32 output channels), v_mfma_f32_32x32x16_f16 with the 3-product hi/lo split of
// nm_conv.hip (fp32-equivalent products, fp32 accumulate).  Each wave streams ITS parity's weights from L2 (2 KB per k-step for
// 12 MFMAs, prefetched two k-steps ahead); a row tile is 8(x) x 2(z) x 2(y) cells, dealt to the lanes so that each lane group of a
// ds_read_b128 hits 16 distinct 16-byte slots (z-plane pitch = 8 mod 16 slots).
//
// Volume border.  The coarse tile is staged with clamped indices, which reproduces torch's clamped interpolation for every fine
// position INSIDE the volume.  The fine convolution, however, zero-pads: taps that leave the fine volume must contribute nothing,
// whereas the composite form gives them the interpolated value f~ of the clamped tile.  For the outer one-voxel shell of the output
//      true = composite - sum_{taps t that leave the volume} w[t] f~[o+t]
// and by inclusion-exclusion over the border axes S of the voxel the subtracted sum is a signed sum of small convolutions of the
// coarse tensor (axes in S: the single outward tap, which reads the voxel's own coarse cell with weight 1; the other axes: the
// interior composite) - 9, 3 or 1 coarse taps.  conv_up2c_shell_kernel applies them to the shell (9 % of the voxels, 3 % of the
// MACs) after the main kernel and owns the shell's GroupNorm partial sums; the main kernel leaves the shell out of its own.
#include "nm_up2c.h"

namespace {

#ifndef UP2C_SCHED
#define UP2C_SCHED 1
#endif
#if UP2C_SCHED
#define UP2C_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define UP2C_SB() do {} while (0)
#endif
#define UP2C_SPLIT_SCALE 2048.0f          // same hi/lo split as nm_conv.hip: v = hi + lo * 2^-11
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BZ = 2, BY = 8, BX = 8;                 // coarse cells of a brick (per parity class: 128 GEMM rows)
constexpr int HZ = BZ + 2, HY = BY + 2, HX = BX + 2;  // coarse halo tile
constexpr int ZP = 104;                               // z-plane pitch in 16-byte slots: >= HY*HX and = 8 (mod 16)
constexpr int HVP = HZ * ZP + 1;                      // slots per (chunk, hi|lo, lane half) plane; odd: staging writes spread over the banks
constexpr int CG = 32;                                // input channels per LDS buffer (two buffers: one read by the MFMAs, one being staged)
constexpr int NPLANES = CG / 16 * 4;
constexpr size_t LDS_BYTES = (size_t)2 * NPLANES * HVP * 16 + 2 * CG * sizeof(float);      // + the next tile's GroupNorm scale / shift

struct Up2cParams {
    const float* in; const float* in_scale; const float* in_shift; float in_slope;
    int N, ID, IH, IW, Cin;
    const half8* wc;             // composite sets (see set_offset)
    const float* bias; float* out; float* part;
    int Cout, Co_pad;
    int nbz, nby, nbx;           // bricks per frame
    int nblk;                    // partial blocks per frame (bricks + shell items)
    int diag;                    // NM355_UP2C_DIAG (timing experiments only): 1 (unused), 2 no staging, 4 no shell launch, 8 no MFMA loop, 16 weights from one address, 32 no barriers, 64 (with NM355_UP2C_X16=0) the 32x32x16 kernel with one accumulator
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v0 = (j < 2) ? a[2 * j] : b[2 * j - 4], v1 = (j < 2) ? a[2 * j + 1] : b[2 * j - 3];
        half2v hh = __builtin_convertvector(f32x2{v0, v1}, half2v);
        asm volatile("" : "+v"(hh));
        const float t0 = v0 * UP2C_SPLIT_SCALE, t1 = v1 * UP2C_SPLIT_SCALE;
        hi[2 * j] = hh[0]; hi[2 * j + 1] = hh[1];
        lo[2 * j] = (_Float16)__builtin_fmaf((float)hh[0], -UP2C_SPLIT_SCALE, t0);
        lo[2 * j + 1] = (_Float16)__builtin_fmaf((float)hh[1], -UP2C_SPLIT_SCALE, t1);
    }
}

// pending GroupNorm affine + LeakyReLU of the producer (TensorRef semantics of nm_common.h)
__device__ __forceinline__ f32x4 act4(f32x4 v, const f32x4& sc, const f32x4& sh, bool affine, float slope) {
    if (affine) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(v[j], sc[j], sh[j]);
    }
    if (slope != 1.0f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], v[j] * slope);
    }
    return v;
}

// 16-bit storage (conv mode 4): eight bfloat16 channels arrive as ONE 16-byte load (kept raw in an f32x4) and are widened where
// they are used; IH = false: the two fp32 quads as before
template <bool IH>
__device__ __forceinline__ void ld8_raw(const float* base, size_t e, f32x4& a, f32x4& b) {
    if constexpr (IH) a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(base) + e);
    else { a = *reinterpret_cast<const f32x4*>(base + e); b = *reinterpret_cast<const f32x4*>(base + e + 4); }
}
template <bool IH>
__device__ __forceinline__ void widen8(f32x4& a, f32x4& b) {
    if constexpr (IH) {
        const unsigned u0 = nm_fbits(a[0]), u1 = nm_fbits(a[1]), u2 = nm_fbits(a[2]), u3 = nm_fbits(a[3]);
        a = f32x4{nm_bf_lo(u0), nm_bf_hi(u0), nm_bf_lo(u1), nm_bf_hi(u1)};
        b = f32x4{nm_bf_lo(u2), nm_bf_hi(u2), nm_bf_lo(u3), nm_bf_hi(u3)};
    }
}

// ---- composite weight sets -------------------------------------------------------------------------------------------------
// set e: e < 8: S = {} (main), parity p = e (pz = e>>2, py = (e>>1)&1, px = e&1); else S = 1 + (e-8)/8 as a bit mask
// (x = 1, y = 2, z = 4) of the axes whose tap leaves the volume, p = (e-8) % 8.  A set holds 3^(free axes) coarse taps (z, y, x
// order over the free axes, d = -1, 0, +1) as [Cin/16][tap][hi h0 | hi h1 | lo h0 | lo h1][Co_pad][8 halves]: the k-steps of a
// set (channel chunk outer, tap inner) are one linear walk through memory.
__host__ __device__ inline int set_axes(int e) { return e < 8 ? 0 : 1 + (e - 8) / 8; }
__host__ __device__ inline int set_parity(int e) { return e < 8 ? e : (e - 8) % 8; }
__host__ __device__ inline int axes_taps(int S) { int t = 27; if (S & 1) t /= 3; if (S & 2) t /= 3; if (S & 4) t /= 3; return t; }
// offset of set e in units of one tap (C16 * 4 * Co_pad half8)
__host__ __device__ inline int set_tap_offset(int e) {
    int off = 0;
    for (int i = 0; i < e; ++i) off += axes_taps(set_axes(i));
    return off;
}
constexpr int TOTAL_TAPS = 8 * 27 + 8 * (3 * 9 + 3 * 3 + 1);      // 512

__device__ __forceinline__ double ucoef(int p, int t, int d) {      // weight of coarse offset d-1 in fine tap t-1, parity p
    // in quarters; p = 0: rows t = (3 1 0) (1 3 0) (0 3 1);  p = 1: (1 3 0) (0 3 1) (0 1 3)
    const int T0[9] = {3, 1, 0, 1, 3, 0, 0, 3, 1}, T1[9] = {1, 3, 0, 0, 3, 1, 0, 1, 3};
    int v = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) if (i == t * 3 + d) v = p ? T1[i] : T0[i];
    return 0.25 * (double)v;
}

__global__ void up2c_compose_kernel(const float* __restrict__ w, int Cout, int Cin, int Co_pad, _Float16* __restrict__ packed) {
    const int C16 = Cin >> 4;
    const size_t per_tap = (size_t)C16 * 2 * Co_pad * 8;          // (cb, hh, co, j) elements, each writing hi and lo
    const size_t total = (size_t)TOTAL_TAPS * per_tap;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = i & 7; size_t r = i >> 3;
        const int co = r % Co_pad; r /= Co_pad;
        const int hh = r & 1; r >>= 1;
        const int cb = r % C16; int gt = (int)(r / C16);             // global tap index over all sets
        int e = 0;
        for (;; ++e) { const int nt = axes_taps(set_axes(e)); if (gt < nt) break; gt -= nt; }
        const int S = set_axes(e), p = set_parity(e);
        const int pa[3] = {p & 1, (p >> 1) & 1, (p >> 2) & 1};       // x, y, z parities
        int d[3] = {1, 1, 1};                                        // coarse offsets + 1 (x, y, z); axes in S stay at 0 offset
        {   // decode the tap over the free axes, z slowest
            int q = gt;
            for (int a = 0; a < 3; ++a) if (!((S >> a) & 1)) { d[a] = q % 3; q /= 3; }
        }
        const int ci = cb * 16 + hh * 8 + j;
        double v = 0.0;
        if (co < Cout && ci < Cin) {
            const float* wk = w + ((size_t)co * Cin + ci) * 27;
            for (int tz = 0; tz < 3; ++tz) for (int ty = 0; ty < 3; ++ty) for (int tx = 0; tx < 3; ++tx) {
                const int t[3] = {tx, ty, tz};
                double c = 1.0;
                for (int a = 0; a < 3; ++a) {
                    if ((S >> a) & 1) { if (t[a] != (pa[a] ? 2 : 0)) c = 0.0; }      // the single tap that leaves the volume
                    else c *= ucoef(pa[a], t[a], d[a]);
                }
                if (c != 0.0) v += c * (double)wk[(tz * 3 + ty) * 3 + tx];
            }
            const int bits = (S & 1) + ((S >> 1) & 1) + ((S >> 2) & 1);
            if (bits & 1) v = -v;                                    // inclusion-exclusion sign (-1)^|S|
        }
        const float vf = (float)v;
        const _Float16 hi = (_Float16)vf;
        const _Float16 lo = (_Float16)((vf - (float)hi) * UP2C_SPLIT_SCALE);
        // position inside the set: [chunk][tap], taps in the order they were decoded (gt)
        const size_t base = (((size_t)set_tap_offset(e) * C16) + (size_t)cb * axes_taps(S) + gt) * 4 * Co_pad * 8;
        packed[base + ((size_t)hh * Co_pad + co) * 8 + j] = hi;
        packed[base + ((size_t)(2 + hh) * Co_pad + co) * 8 + j] = lo;
    }
}

// ---- main kernel -----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ constexpr int tap_off(int t) { return (t / 9) * ZP + ((t / 3) % 3) * HX + (t % 3); }

// One STEP = (brick, 32-output-channel group, 32-input-channel group): 2 chunks x 27 taps = 54 k-steps of 12 MFMAs per wave on
// the LDS buffer `cur`.  The tile of the NEXT step is staged into the other buffer from inside the k-loop: each thread owns up to
// four (voxel, channel octet) items; an item's global loads are issued a dozen k-steps before its activate / split / LDS write,
// so neither the load latency nor the conversion VALU is exposed - they run in the shadow of the two waves' MFMAs (a loop that
// only issues MFMAs and LDS reads keeps the pipe ~100 % busy; staged as a separate phase the same work cost 22 % of the kernel:
// one workgroup per CU has nothing else to run meanwhile).  One workgroup barrier per step.
struct StepPos { int n, br, nh, cg, cz0, cy0, cx0; };

// IO: bit 0 = bfloat16 input, bit 1 = bfloat16 output (16-bit storage; the shell kernels then read-modify-write bfloat16 values)
// ONE (A/B arm of profiles/r06_mfma_shape_ab.txt): the three products in one accumulator as in conv_up2c_x16_kernel
template <bool SINGLE, int IO = 0, bool ONE = false>
__global__ __launch_bounds__(512, 1) void conv_up2c_kernel(Up2cParams p) {
    constexpr bool IH16 = (IO & 1) != 0, OH16 = (IO & 2) != 0;
    extern __shared__ f32x4 lds_raw[];
    half8* tile = reinterpret_cast<half8*>(lds_raw);               // [buffer][chunk*4 + hl*2 + h][HVP] x 16 B, slot = hz*ZP + hy*HX + hx

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
    const int C16 = p.Cin >> 4, NCG = p.Cin / CG, NH = p.Cout >> 5;
    const size_t plane = (size_t)p.Co_pad;
    const size_t kstride = 4 * plane;                             // half8 units between consecutive k-steps of a set
    const half8* __restrict__ wset = p.wc + (size_t)wave * 27 * C16 * kstride;      // wave-uniform
    const unsigned wlane = (unsigned)(h * (int)plane + l31) * 16u;                  // per-lane byte offset of every weight load
    const size_t kbytes = kstride * 16, lobytes = 2 * plane * 16;
    const int OD = 2 * p.ID, OH = 2 * p.IH, OW = 2 * p.IW;
    // A rows of this lane (16-byte slots, lane half's plane included): tile j covers y in {2j, 2j+1}.  A ds_read_b128 is served in
    // the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (quads {0,3,5,6} / {1,2,4,7} of a lane half): a group's four quads are
    // the four (x half, z) combinations of ONE y, i.e. slot bases 0, 4, 8, 12 (+10 for the other y) mod 16 - conflict-free.
    //   row l31:  x = 4 * bit3 + (l31 & 3),  z = bit4,  y = 2j + ((0x96 >> (l31 >> 2)) & 1)
    int arow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        arow[j] = h * HVP + ((l31 >> 4) & 1) * ZP + (2 * j + ((0x96 >> (l31 >> 2)) & 1)) * HX + 4 * ((l31 >> 3) & 1) + (l31 & 3);
    // staging role: channel octet tid & 3 of the 32-channel group, tile voxels tid/4 + 128 k
    const int s_oct = tid & 3;
    const int s_plane = (s_oct >> 1) * 4 + (s_oct & 1);
    constexpr int NV = HZ * HY * HX, SITEMS = (NV * 4 + 511) / 512;       // 4 items per thread (the last one for 64 threads only)
    const bool affine = p.in_scale != nullptr;

    const int bricks = p.nbz * p.nby * p.nbx, total = p.N * bricks;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int item0 = (int)blockIdx.x * per, item_end = min(total, item0 + per);
    if (item0 >= item_end) return;
    auto pos_of = [&](int item, int nh, int cg) {
        StepPos s; s.n = item / bricks; s.br = item % bricks; s.nh = nh; s.cg = cg;
        s.cx0 = (s.br % p.nbx) * BX; s.cy0 = ((s.br / p.nbx) % p.nby) * BY; s.cz0 = (s.br / (p.nbx * p.nby)) * BZ;
        return s;
    };
    // staging of one item: loads (issue) and activate / split / write (commit)
    f32x4 pr_a, pr_b, sca, scb, sha, shb;
    sca = scb = sha = shb = pr_a = pr_b = f32x4{0.f, 0.f, 0.f, 0.f};
    auto issue_affine = [&](const StepPos& s) {
        if (affine) {
            const int cbase = s.cg * CG + s_oct * 8;
            const float* ps = p.in_scale + (size_t)s.n * p.Cin + cbase; const float* ph = p.in_shift + (size_t)s.n * p.Cin + cbase;
            sca = *reinterpret_cast<const f32x4*>(ps); scb = *reinterpret_cast<const f32x4*>(ps + 4);
            sha = *reinterpret_cast<const f32x4*>(ph); shb = *reinterpret_cast<const f32x4*>(ph + 4);
        }
    };
    auto issue = [&](const StepPos& s, int k) {
        // the item's voxel index (tid >> 2) + 128 k, computed HERE by an opaque instruction pair: as a plain expression it is
        // loop-invariant, hipcc hoists it, keeps the four of them (k = 0..3) alive across the k-loop and - at 256 registers - spills
        // them; every reload was followed by s_waitcnt vmcnt(0), i.e. drained the weight loads in flight (8 drains per step)
        int v;
        asm volatile("v_lshrrev_b32 %0, 2, %1\n\tv_add_u32 %0, %2, %0" : "=v"(v) : "v"(tid), "s"(128 * k));
        if (v < NV) {
            const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
            const int gz = min(max(s.cz0 - 1 + hz, 0), p.ID - 1), gy = min(max(s.cy0 - 1 + hy, 0), p.IH - 1), gx = min(max(s.cx0 - 1 + hx, 0), p.IW - 1);
            ld8_raw<IH16>(p.in, ((((size_t)s.n * p.ID + gz) * p.IH + gy) * p.IW + gx) * p.Cin + s.cg * CG + s_oct * 8, pr_a, pr_b);
        }
    };
    auto commit = [&](half8* buf, int k) {
        int v;
        asm volatile("v_lshrrev_b32 %0, 2, %1\n\tv_add_u32 %0, %2, %0" : "=v"(v) : "v"(tid), "s"(128 * k));
        if (v < NV) {
            const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
            half8 hi, lo;
            widen8<IH16>(pr_a, pr_b);
            split8(act4(pr_a, sca, sha, affine, p.in_slope), act4(pr_b, scb, shb, affine, p.in_slope), hi, lo);
            const int slot = hz * ZP + hy * HX + hx;
            buf[s_plane * HVP + slot] = hi;
            if constexpr (!SINGLE) buf[(s_plane + 2) * HVP + slot] = lo;      // (the one-product modes never read the lo planes)
        }
    };
    auto ldw = [&](const char* base, size_t extra) { return *reinterpret_cast<const half8*>(base + extra + wlane); };

    // first tile: staged in the open
    StepPos cs = pos_of(item0, 0, 0);
    issue_affine(cs);
#pragma unroll
    for (int k = 0; k < SITEMS; ++k) { issue(cs, k); commit(tile, k); }
    lds_barrier();
    int cur = 0, item = item0;
    f32x16 acc[4], accl[4];
    // The k-steps of consecutive steps form ONE software-pipelined stream: the B operands (two k-steps ahead) and the A operands
    // (one ahead) of a step's first k-steps are fetched during the last k-steps of the step before it, so a step boundary costs no
    // pipeline refill.  Two workgroup barriers per step keep the two LDS buffers apart: at k-step 10 (before this step's first
    // write into the other buffer: every wave has left the previous step, which read that buffer) and at k-step 51 (after the last
    // write: the other buffer is complete before the first read of the next step is issued at k-step 53).
    constexpr int KS = CG / 16 * 27;                                  // k-steps per step (54)
    const char* wk = reinterpret_cast<const char*>(wset);              // (cs.nh = cs.cg = 0)
    half8 bh[3], bl[3], ah[4], al[4];
    bh[0] = ldw(wk, 0); bl[0] = ldw(wk, lobytes);
    bh[1] = ldw(wk, kbytes); bl[1] = ldw(wk, kbytes + lobytes);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ah[j] = tile[arow[j]]; al[j] = tile[arow[j] + 2 * HVP]; }
    for (;;) {
        // the step after this one
        StepPos ns = cs; bool have_next = true;
        if (cs.cg + 1 < NCG) ns.cg = cs.cg + 1;
        else if (cs.nh + 1 < NH) { ns.nh = cs.nh + 1; ns.cg = 0; }
        else if (item + 1 < item_end) ns = pos_of(item + 1, 0, 0);
        else have_next = false;
        if (cs.cg == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accl[j][r] = 0.f; }
        }
        const half8* tb = tile + cur * (NPLANES * HVP);
        half8* nb = tile + (cur ^ 1) * (NPLANES * HVP);
        const bool stage = have_next && !(p.diag & 2);
        // weights of the next step's first k-steps (a finished workgroup reads its own first set again: any valid address)
        const char* wnext = reinterpret_cast<const char*>(wset + (size_t)ns.nh * 32 + (size_t)(ns.cg * (CG / 16)) * 27 * kstride);
        if (stage) issue_affine(ns);
        if (!(p.diag & 8)) {
#pragma unroll
        for (int c = 0; c < CG / 16; ++c) {
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int ks = c * 27 + t, s = ks % 3, s2 = (ks + 2) % 3;
                if (ks + 2 < KS) { bh[s2] = ldw(wk, 2 * kbytes); bl[s2] = ldw(wk, 2 * kbytes + lobytes); }
                else { bh[s2] = ldw(wnext, (size_t)(ks + 2 - KS) * kbytes); bl[s2] = ldw(wnext, (size_t)(ks + 2 - KS) * kbytes + lobytes); }
                if (!(p.diag & 16)) wk += kbytes;
                // staging pieces of the next tile: item i loaded at k-step 1 + 13 i, written at 11 + 13 i
                if (stage) {
#pragma unroll
                    for (int i = 0; i < SITEMS; ++i) {
                        if (ks == 1 + 13 * i) issue(ns, i);
                        if (ks == 11 + 13 * i) commit(nb, i);
                    }
                }
                if (have_next && (ks == 10 || ks == 51) && !(p.diag & 32)) lds_barrier();
                UP2C_SB();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (ONE) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh[s] * (_Float16)UP2C_SPLIT_SCALE, acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh[s], acc[j], 0, 0, 0);
                }
                UP2C_SB();
                // A operands of the next k-step (after the last one: tap 0 of the other buffer)
                const half8* xb = (ks == KS - 1) ? nb : tb;
                const int nof = (ks == KS - 1) ? 0 : ((t < 26) ? c * 4 * HVP + tap_off((t + 1) % 27) : (c + 1) * 4 * HVP);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (ONE) acc[j] = nm_mfma_lo<SINGLE>(ah[j], bl[s], acc[j]);
                    else accl[j] = nm_mfma_lo<SINGLE>(ah[j], bl[s], accl[j]);
                    ah[j] = xb[arow[j] + nof];
                    UP2C_SB();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (ONE) acc[j] = nm_mfma_lo<SINGLE>(al[j], bh[s], acc[j]);
                    else accl[j] = nm_mfma_lo<SINGLE>(al[j], bh[s], accl[j]);
                    al[j] = xb[arow[j] + 2 * HVP + nof];
                    UP2C_SB();
                }
            }
        }
        }
        wk = wnext;
        if (cs.cg == NCG - 1) {
            // ---- epilogue of this 32-channel group: bias, store, GroupNorm partials
            // without the shell (one slot per wave)
            const int co = cs.nh * 32 + l31;
            const float bv = p.bias ? p.bias[co] : 0.f;
            float s = 0.f, ss = 0.f;
            const size_t sX = (size_t)p.Cout, sY = (size_t)OW * sX, sZ = (size_t)OH * sY;
            const bool border_brick = cs.cz0 == 0 || cs.cz0 + BZ == p.ID || cs.cy0 == 0 || cs.cy0 + BY == p.IH || cs.cx0 == 0 || cs.cx0 + BX == p.IW;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* base = nm_eptr(p.out, ((((size_t)cs.n * OD + 2 * cs.cz0 + pz) * OH + 2 * (cs.cy0 + 2 * j) + py) * OW + 2 * cs.cx0 + px) * sX + co, OH16);
                if constexpr (OH16) {
                    // bfloat16: the lanes of neighbouring channels exchange values so that each stores ONE packed dword per register pair
                    // (even lanes the channel pair of row r, odd lanes of row r + 1 - one x step further)
                    const bool odd = (l31 & 1) != 0;
                    unsigned short* bq = reinterpret_cast<unsigned short*>(base) - (odd ? 1 : 0);
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int x = 4 * ((r >> 2) & 1) + (r & 3), zb = (r >> 3) & 1, yb = (0x96 >> (h + 2 * (r >> 2))) & 1;
                        const float v0 = (ONE ? acc[j][r] * (1.0f / UP2C_SPLIT_SCALE) : __builtin_fmaf(accl[j][r], 1.0f / UP2C_SPLIT_SCALE, acc[j][r])) + bv;
                        const float v1 = (ONE ? acc[j][r + 1] * (1.0f / UP2C_SPLIT_SCALE) : __builtin_fmaf(accl[j][r + 1], 1.0f / UP2C_SPLIT_SCALE, acc[j][r + 1])) + bv;
                        const float send = odd ? v0 : v1;
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, true));
                        const unsigned pk = odd ? nm_pk_bf16(recv, v1) : nm_pk_bf16(v0, recv);
                        *reinterpret_cast<unsigned*>(bq + (size_t)(2 * zb) * sZ + (size_t)(2 * yb) * sY + (size_t)(2 * (x + (odd ? 1 : 0))) * sX) = pk;
                        float m0 = v0, m1 = v1;
                        if (border_brick) {
                            const int oz = 2 * (cs.cz0 + zb) + pz, oy = 2 * (cs.cy0 + 2 * j + yb) + py, ox0 = 2 * (cs.cx0 + x) + px, ox1 = ox0 + 2;
                            const bool zy = oz == 0 || oz == OD - 1 || oy == 0 || oy == OH - 1;
                            if (zy || ox0 == 0 || ox0 == OW - 1) m0 = 0.f;
                            if (zy || ox1 == 0 || ox1 == OW - 1) m1 = 0.f;
                        }
                        s += m0; ss = __builtin_fmaf(m0, m0, ss);
                        s += m1; ss = __builtin_fmaf(m1, m1, ss);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (!border_brick) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // register r of lane half h is row (r & 3) + 4 h + 8 (r >> 2) of the tile (see arow)
                        const int x = 4 * ((r >> 2) & 1) + (r & 3), zb = (r >> 3) & 1, yb = (0x96 >> (h + 2 * (r >> 2))) & 1;
                        const float v = (ONE ? acc[j][r] * (1.0f / UP2C_SPLIT_SCALE) : __builtin_fmaf(accl[j][r], 1.0f / UP2C_SPLIT_SCALE, acc[j][r])) + bv;
                        base[(size_t)(2 * zb) * sZ + (size_t)(2 * yb) * sY + (size_t)(2 * x) * sX] = v;
                        s += v; ss = __builtin_fmaf(v, v, ss);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int x = 4 * ((r >> 2) & 1) + (r & 3), zb = (r >> 3) & 1, yb = (0x96 >> (h + 2 * (r >> 2))) & 1;
                        const float v = (ONE ? acc[j][r] * (1.0f / UP2C_SPLIT_SCALE) : __builtin_fmaf(accl[j][r], 1.0f / UP2C_SPLIT_SCALE, acc[j][r])) + bv;
                        base[(size_t)(2 * zb) * sZ + (size_t)(2 * yb) * sY + (size_t)(2 * x) * sX] = v;
                        const int oz = 2 * (cs.cz0 + zb) + pz, oy = 2 * (cs.cy0 + 2 * j + yb) + py, ox = 2 * (cs.cx0 + x) + px;
                        const bool shell = oz == 0 || oz == OD - 1 || oy == 0 || oy == OH - 1 || ox == 0 || ox == OW - 1;
                        const float mv = shell ? 0.f : v;
                        s += mv; ss = __builtin_fmaf(mv, mv, ss);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (p.part) {
                s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
                if (h == 0) { float* dst = p.part + (((size_t)cs.n * p.nblk + cs.br * 8 + wave) * p.Cout + co) * 2; dst[0] = s; dst[1] = ss; }
            }
        }
        if (!have_next) break;
        if (ns.cg == 0 && ns.nh == 0) ++item;
        cs = ns; cur ^= 1;
    }
}

// ---- the same kernel on v_mfma_f32_16x16x32_f16 (round 6) -------------------------------------------------------------------
// Same bricks, same LDS image (the staging code is shared verbatim), same packed weights, same 3-product arithmetic; what changes is
// the MFMA shape: K = 32 takes BOTH 16-channel chunks of the 32-channel LDS buffer in one instruction, a row tile is 16 rows, a
// column block 16 output channels.  Per wave and tap: 8 row tiles x 2 column blocks x 3 products = 48 MFMAs of 16 cycles (the
// 32x32x16 form: 2 chunks x 4 tiles x 3 = 24 of 32 cycles) - the same FLOPs, the same operand bytes (16 A reads + 4 B loads per
// tap), a step is 27 k-steps instead of 54.  MI355X_MICROARCH.md 'DVFS give-back' item 7: where the chip holds its clock down under
// matrix load (this kernel: 1.49 GHz, profiles/r05_pmc_mfma.json) it holds a higher one on this shape.
//   lane = 16 q + r:  A row r of the tile, k = 8 q + j -> channel (q >> 1) * 16 + (q & 1) * 8 + j of the 32-channel group, i.e. LDS
//   plane (q >> 1) * 4 + hl * 2 + (q & 1);  B column r of the block, the same k -> weight chunk 2 cg + (q >> 1), half q & 1.
// Row tile j = (x parity j & 1, y pair j >> 1):  row r -> x = 2 (r & 3) + (j & 1),  z = r >> 3,  y = 2 (j >> 1) + (bit 2 ^ bit 3 of r).
// A ds_read_b128 is served in the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32): a group takes rows {0-3,12-15} of one
// plane and rows {4-11} of the NEXT plane (or the other way round); each of the two row sets covers the eight even slots (mod 16)
// - x stride 2, z-plane pitch = 8 (mod 16), y pitch 10 - and consecutive planes are HVP = 1 (mod 16) slots apart: 16 distinct slots.
// C/D: column = lane & 15 (output channel of the block), row = 4 q + reg:  x = 2 reg + (j & 1),  z = q >> 1,  y = 2 (j >> 1) + ((q ^ (q >> 1)) & 1).
template <bool SINGLE>
__device__ __forceinline__ f32x4 up2c_mfma16_lo(half8 a, half8 b, f32x4 c) {
    if constexpr (SINGLE) return c;
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <bool SINGLE>
__global__ __launch_bounds__(512, 1) void conv_up2c_x16_kernel(Up2cParams p) {
    extern __shared__ f32x4 lds_raw[];
    half8* tile = reinterpret_cast<half8*>(lds_raw);               // [buffer][chunk*4 + hl*2 + h][HVP] x 16 B, slot = hz*ZP + hy*HX + hx

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
    const int C16 = p.Cin >> 4, NCG = p.Cin / CG, NH = p.Cout >> 5;
    const size_t plane = (size_t)p.Co_pad;
    const size_t kstride = 4 * plane;                             // half8 units between consecutive taps of a chunk
    const half8* __restrict__ wset = p.wc + (size_t)wave * 27 * C16 * kstride;      // wave-uniform
    const unsigned wlane = (unsigned)((q >> 1) * 27 * (int)kstride + (q & 1) * (int)plane + r16) * 16u;      // per-lane byte offset of every weight load
    const size_t kbytes = kstride * 16, lobytes = 2 * plane * 16;
    const int OD = 2 * p.ID, OH = 2 * p.IH, OW = 2 * p.IW;
    // A rows of this lane: tile j adds the compile-time offset 2 (j >> 1) HX + (j & 1)
    const int abase = ((q >> 1) * 4 + (q & 1)) * HVP + (r16 >> 3) * ZP + (((r16 >> 2) ^ (r16 >> 3)) & 1) * HX + 2 * (r16 & 3);
    // staging role: channel octet tid & 3 of the 32-channel group, tile voxels tid/4 + 128 k
    const int s_oct = tid & 3;
    const int s_plane = (s_oct >> 1) * 4 + (s_oct & 1);
    constexpr int NV = HZ * HY * HX, SITEMS = (NV * 4 + 511) / 512;
    const bool affine = p.in_scale != nullptr;

    const int bricks = p.nbz * p.nby * p.nbx, total = p.N * bricks;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int item0 = (int)blockIdx.x * per, item_end = min(total, item0 + per);
    if (item0 >= item_end) return;
    auto pos_of = [&](int item, int nh, int cg) {
        StepPos s; s.n = item / bricks; s.br = item % bricks; s.nh = nh; s.cg = cg;
        s.cx0 = (s.br % p.nbx) * BX; s.cy0 = ((s.br / p.nbx) % p.nby) * BY; s.cz0 = (s.br / (p.nbx * p.nby)) * BZ;
        return s;
    };
    // the pending GroupNorm scale / shift of the tile being staged sit in LDS ([scale 32][shift 32] floats behind the two tile
    // buffers; 16 registers less than holding them): written by the first wave's lanes at k-step 3 (loaded at k-step 0), i.e. behind
    // the previous step's barrier at k-step 25 (its last commit was at 24) and before this step's barrier at k-step 4 (first commit at 5)
    float* aff = reinterpret_cast<float*>(tile + 2 * NPLANES * HVP);
    f32x4 pr_a, pr_b;
    pr_a = pr_b = f32x4{0.f, 0.f, 0.f, 0.f};
    float aff_v = 0.f;
    auto aff_load = [&](const StepPos& s) {
        if (affine && tid < 64) aff_v = (tid < 32 ? p.in_scale : p.in_shift)[(size_t)s.n * p.Cin + s.cg * CG + (tid & 31)];
    };
    auto aff_store = [&]() { if (affine && tid < 64) aff[tid] = aff_v; };
    auto issue = [&](const StepPos& s, int k) {
        int v;      // (opaque: see conv_up2c_kernel)
        asm volatile("v_lshrrev_b32 %0, 2, %1\n\tv_add_u32 %0, %2, %0" : "=v"(v) : "v"(tid), "s"(128 * k));
        if (v < NV) {
            const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
            const int gz = min(max(s.cz0 - 1 + hz, 0), p.ID - 1), gy = min(max(s.cy0 - 1 + hy, 0), p.IH - 1), gx = min(max(s.cx0 - 1 + hx, 0), p.IW - 1);
            ld8_raw<false>(p.in, ((((size_t)s.n * p.ID + gz) * p.IH + gy) * p.IW + gx) * p.Cin + s.cg * CG + s_oct * 8, pr_a, pr_b);
        }
    };
    auto commit = [&](half8* buf, int k) {
        int v;
        asm volatile("v_lshrrev_b32 %0, 2, %1\n\tv_add_u32 %0, %2, %0" : "=v"(v) : "v"(tid), "s"(128 * k));
        if (v < NV) {
            const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
            half8 hi, lo;
            f32x4 sca = {0.f, 0.f, 0.f, 0.f}, scb = sca, sha = sca, shb = sca;
            if (affine) {
                sca = *reinterpret_cast<const f32x4*>(aff + s_oct * 8); scb = *reinterpret_cast<const f32x4*>(aff + s_oct * 8 + 4);
                sha = *reinterpret_cast<const f32x4*>(aff + 32 + s_oct * 8); shb = *reinterpret_cast<const f32x4*>(aff + 32 + s_oct * 8 + 4);
            }
            split8(act4(pr_a, sca, sha, affine, p.in_slope), act4(pr_b, scb, shb, affine, p.in_slope), hi, lo);
            const int slot = hz * ZP + hy * HX + hx;
            buf[s_plane * HVP + slot] = hi;
            if constexpr (!SINGLE) buf[(s_plane + 2) * HVP + slot] = lo;
        }
    };
    auto ldw = [&](const char* base, size_t extra) { return *reinterpret_cast<const half8*>(base + extra + wlane); };

    // first tile: staged in the open
    StepPos cs = pos_of(item0, 0, 0);
    aff_load(cs); aff_store();
    lds_barrier();
#pragma unroll
    for (int k = 0; k < SITEMS; ++k) { issue(cs, k); commit(tile, k); }
    lds_barrier();
    int cur = 0, item = item0;
    f32x4 acc[8][2];
    // One software-pipelined stream over the steps as in conv_up2c_kernel: B operands one k-step ahead (three register sets by name,
    // two live: 27 % 3 == 0 keeps the set of a step's first k-step fixed), A operands refilled right after their last use.  Barriers
    // at k-step 4 (before the first write into the other buffer) and 25 (after the last one, before the first read of it at 26).
    constexpr int KS = 27;
    const char* wk = reinterpret_cast<const char*>(wset);              // (cs.nh = cs.cg = 0)
    // A operands: a ring of PAIR slots (pair u = row tiles 2u, 2u + 1: the two x parities of one y pair, one slot apart in LDS); pair
    // u of tap t lives in ring slot (4 t + u) % NRING and is refilled with the pair NRING places further on right after its last MFMA
    constexpr int NRING = 2;
    half8 bh[3][2], bl[3][2], ah[NRING][2], al[NRING][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) { bh[0][nb] = ldw(wk, nb * 256); bl[0][nb] = ldw(wk, nb * 256 + lobytes); }
#pragma unroll
    for (int u = 0; u < NRING; ++u)
#pragma unroll
        for (int x = 0; x < 2; ++x) { ah[u][x] = tile[abase + 2 * u * HX + x]; al[u][x] = tile[abase + 2 * u * HX + x + 2 * HVP]; }
    for (;;) {
        StepPos ns = cs; bool have_next = true;
        if (cs.cg + 1 < NCG) ns.cg = cs.cg + 1;
        else if (cs.nh + 1 < NH) { ns.nh = cs.nh + 1; ns.cg = 0; }
        else if (item + 1 < item_end) ns = pos_of(item + 1, 0, 0);
        else have_next = false;
        if (cs.cg == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[j][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const half8* tb = tile + cur * (NPLANES * HVP);
        half8* nb_ = tile + (cur ^ 1) * (NPLANES * HVP);
        const bool stage = have_next && !(p.diag & 2);
        const char* wnext = reinterpret_cast<const char*>(wset + (size_t)ns.nh * 32 + (size_t)(ns.cg * (CG / 16)) * 27 * kstride);
        if (!(p.diag & 8)) {
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const int s = t % 3, s1 = (t + 1) % 3;
            // staging pieces of the next tile: item i loaded at k-step 0 / 7 / 13 / 19, written at 5 / 12 / 18 / 24
            if (stage) {
                if (t == 0) aff_load(ns);
                if (t == 3) aff_store();
#pragma unroll
                for (int i = 0; i < SITEMS; ++i) {
                    if (t == (i == 0 ? 0 : 6 * i + 1)) issue(ns, i);
                    if (t == (i == 0 ? 5 : 6 * i + 6)) commit(nb_, i);
                }
            }
            if (have_next && (t == 4 || t == 25) && !(p.diag & 32)) lds_barrier();
            // B operands of the next k-step - requested BEHIND the staging code: a commit waits for its item with vmcnt(0) (the item's
            // loads sit under a condition, so hipcc does not count), which then finds only loads that are a k-step old
            {
                const char* wsrc = (t + 1 < KS) ? wk + kbytes : wnext;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) { bh[s1][nb] = ldw(wsrc, nb * 256); bl[s1][nb] = ldw(wsrc, nb * 256 + lobytes); }
                wk += kbytes;
            }
            // ONE accumulator for the three products: x w 2^11 = x_hi (2^11 w_hi) + x_hi w_lo' + x_lo' w_hi with the lo parts stored
            // times 2^11 as everywhere; 2^11 w_hi is exact in fp16 (|w| < 32) and made here from w_hi (four packed multiplies per block)
            half8 b2k[2];
            if constexpr (!SINGLE) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) b2k[nb] = bh[s][nb] * (_Float16)UP2C_SPLIT_SCALE;
            } else { b2k[0] = bh[s][0]; b2k[1] = bh[s][1]; }
            UP2C_SB();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rs = (4 * t + u) % NRING;
                // the pair this slot receives next: NRING pairs on, possibly in the next tap (after the last tap: tap 0 of the other buffer)
                const int un = (u + NRING) % 4, tn = t + (u + NRING) / 4;
                const half8* xb = (tn == KS) ? nb_ : tb;
                const int nof = (tn == KS) ? 0 : tap_off(tn);
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[2 * u + x][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rs][x], b2k[nb], acc[2 * u + x][nb], 0, 0, 0);
                UP2C_SB();
#pragma unroll
                for (int x = 0; x < 2; ++x) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[2 * u + x][nb] = up2c_mfma16_lo<SINGLE>(ah[rs][x], bl[s][nb], acc[2 * u + x][nb]);
                    ah[rs][x] = xb[abase + 2 * un * HX + x + nof];
                    UP2C_SB();
                }
#pragma unroll
                for (int x = 0; x < 2; ++x) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[2 * u + x][nb] = up2c_mfma16_lo<SINGLE>(al[rs][x], bh[s][nb], acc[2 * u + x][nb]);
                    if constexpr (!SINGLE) al[rs][x] = xb[abase + 2 * un * HX + x + 2 * HVP + nof];
                    UP2C_SB();
                }
            }
        }
        }
        wk = wnext;
        if (cs.cg == NCG - 1) {
            // ---- epilogue of this 32-channel group: bias, store, GroupNorm partials without the shell (one slot per wave)
            const size_t sX = (size_t)p.Cout;
            const bool border_brick = cs.cz0 == 0 || cs.cz0 + BZ == p.ID || cs.cy0 == 0 || cs.cy0 + BY == p.IH || cs.cx0 == 0 || cs.cx0 + BX == p.IW;
            const int zb = q >> 1, yb = (q ^ (q >> 1)) & 1;
            const int oz = 2 * (cs.cz0 + zb) + pz;
            const bool zshell = oz == 0 || oz == OD - 1;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int co = cs.nh * 32 + nb * 16 + r16;
                const float bv = p.bias ? p.bias[co] : 0.f;
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int oy = 2 * (cs.cy0 + 2 * (j >> 1) + yb) + py;
                    float* base = p.out + ((((size_t)cs.n * OD + oz) * OH + oy) * OW + 2 * (cs.cx0 + (j & 1)) + px) * sX + co;
                    const bool zy = zshell || oy == 0 || oy == OH - 1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[j][nb][r] * (SINGLE ? 1.0f : 1.0f / UP2C_SPLIT_SCALE) + bv;
                        base[(size_t)(4 * r) * sX] = v;
                        float mv = v;
                        if (border_brick) {
                            const int ox = 2 * (cs.cx0 + 2 * r + (j & 1)) + px;
                            if (zy || ox == 0 || ox == OW - 1) mv = 0.f;
                        }
                        s += mv; ss = __builtin_fmaf(mv, mv, ss);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (p.part) {
                    s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
                    s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
                    if (q == 0) { float* dst = p.part + (((size_t)cs.n * p.nblk + cs.br * 8 + wave) * p.Cout + co) * 2; dst[0] = s; dst[1] = ss; }
                }
            }
        }
        if (!have_next) break;
        if (ns.cg == 0 && ns.nh == 0) ++item;
        cs = ns; cur ^= 1;
    }
}

// ---- shell kernels -----------------------------------------------------------------------------------------------------------
// Ownership of the shell cells (per parity class): a cell on two or three faces belongs to an EDGE item, every other shell cell to
// the FACE item of its face.  A face cell needs one correction set (S = its face's axis, 9 coarse taps); the cells of an edge need
// three (two faces - their common taps), a corner seven.
//
// conv_up2c_face_kernel: one workgroup = one line of 32 cells on a face (along y on the x faces, along x on the y and z faces);
// its four waves are the four parity combinations of the two in-face axes, which read the SAME coarse voxels (3 lines x 34, all
// channels): staged once into LDS (activated, split), then 9 taps x Cin/16 k-steps per wave from LDS with the wave's own weights
// streamed three k-steps ahead.  The result is added to the main kernel's output; the wave writes the GroupNorm partials of its
// voxels' final values.
constexpr int FP = 36;                                // slots per staged line (34 used)
constexpr int FPV = 3 * FP + 1;                       // slots per plane

template <bool SINGLE, int IO = 0>
__global__ __launch_bounds__(256, 4) void conv_up2c_face_kernel(Up2cParams p, int TY, int TX) {
    constexpr bool IH16 = (IO & 1) != 0, OH16 = (IO & 2) != 0;
    extern __shared__ f32x4 lds_raw[];
    half8* tile = reinterpret_cast<half8*>(lds_raw);               // [chunk*4 + hl*2 + h][FPV], slot = across * FP + along
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int nX = 2 * p.ID * TY, nY = 2 * p.ID * TX, nZ = 2 * p.IH * TX, per_frame = nX + nY + nZ;
    const int n = (int)blockIdx.x / per_frame; int q = (int)blockIdx.x % per_frame;
    const int bricks = p.nbz * p.nby * p.nbx;
    const int slot_part = 8 * bricks + q * 4 + wave;
    // type 0: x face (line along y at z = f), 1: y face (along x at z = f), 2: z face (along x at y = f)
    int type, side, f, tl;
    if (q < nX) { type = 0; tl = q % TY; f = (q / TY) % p.ID; side = q / (TY * p.ID); }
    else if (q < nX + nY) { q -= nX; type = 1; tl = q % TX; f = (q / TX) % p.ID; side = q / (TX * p.ID); }
    else { q -= nX + nY; type = 2; tl = q % TX; f = (q / TX) % p.IH; side = q / (TX * p.IH); }
    const int LA = type == 0 ? p.IH : p.IW;                            // cells along the line's axis
    const int AC = type == 2 ? p.IH : p.ID;                            // extent of the across axis
    const int fb = side ? (type == 0 ? p.IW : type == 1 ? p.IH : p.ID) - 1 : 0;      // the face's coarse index on its own axis
    const int C16 = p.Cin >> 4, NH = p.Cout >> 5;
    const bool affine = p.in_scale != nullptr;
    // ---- stage 3 lines x 34 voxels x Cin channels ----
    const int noct = p.Cin >> 3, nitems = 3 * 34 * noct;             // (noct divides 256: a thread's channel octet is fixed)
    f32x4 sca = {0.f, 0.f, 0.f, 0.f}, sha = sca, scb = sca, shb = sca;
    if (affine) {
        const int oct = tid % noct;
        const float* ps = p.in_scale + (size_t)n * p.Cin + oct * 8; const float* ph = p.in_shift + (size_t)n * p.Cin + oct * 8;
        sca = *reinterpret_cast<const f32x4*>(ps); scb = *reinterpret_cast<const f32x4*>(ps + 4);
        sha = *reinterpret_cast<const f32x4*>(ph); shb = *reinterpret_cast<const f32x4*>(ph + 4);
    }
    for (int i = tid; i < nitems; i += 256) {
        const int oct = i % noct, v = i / noct, s = v % 34, a = v / 34;
        const int ca = min(max(f + a - 1, 0), AC - 1), cl = min(max(tl * 32 - 1 + s, 0), LA - 1);
        const int gz = type == 2 ? fb : ca, gy = type == 0 ? cl : (type == 1 ? fb : ca), gx = type == 0 ? fb : cl;
        f32x4 va, vb;
        ld8_raw<IH16>(p.in, ((((size_t)n * p.ID + gz) * p.IH + gy) * p.IW + gx) * p.Cin + oct * 8, va, vb);
        widen8<IH16>(va, vb);
        half8 hi, lo;
        split8(act4(va, sca, sha, affine, p.in_slope), act4(vb, scb, shb, affine, p.in_slope), hi, lo);
        const int pl = (oct >> 1) * 4 + (oct & 1);
        tile[pl * FPV + a * FP + s] = hi;
        tile[(pl + 2) * FPV + a * FP + s] = lo;
    }
    lds_barrier();
    // ---- this wave's parity class, rows, weights ----
    const int pa = wave >> 1, pl_ = wave & 1;                        // parities of the across / along axes
    int pz, py, px;
    if (type == 0) { px = side; pz = pa; py = pl_; } else if (type == 1) { py = side; pz = pa; px = pl_; } else { pz = side; py = pa; px = pl_; }
    const int par = pz * 4 + py * 2 + px;
    const int S = 1 << type;
    const int ab = pa ? AC - 1 : 0, lb = pl_ ? LA - 1 : 0;           // border indices of this class on the two in-face axes
    const bool wave_off = f == ab;                                    // the whole line lies on an edge for this class
    auto row_owned = [&](int rho) { const int u = tl * 32 + rho; return !wave_off && u < LA && u != lb; };
    const size_t plane = (size_t)p.Co_pad, kstride = 4 * plane;
    const int nk = 9 * C16;
    const int OD = 2 * p.ID, OH = 2 * p.IH, OW = 2 * p.IW;
    const int arow = h * FPV + l31;
    for (int nh = 0; nh < NH; ++nh) {
        f32x16 acc, accl;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }
        if (!wave_off) {
            const half8* wq = p.wc + (size_t)set_tap_offset(8 + (S - 1) * 8 + par) * C16 * kstride + (size_t)h * plane + nh * 32 + l31;
            half8 bh[3], bl[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) { bh[u] = wq[(size_t)min(u, nk - 1) * kstride]; bl[u] = wq[(size_t)min(u, nk - 1) * kstride + 2 * plane]; }
            for (int c = 0; c < C16; ++c) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int k = c * 9 + t, s = t % 3;
                    const half8 ah = tile[arow + c * 4 * FPV + (t / 3) * FP + (t % 3)];
                    const half8 al = tile[arow + (c * 4 + 2) * FPV + (t / 3) * FP + (t % 3)];
                    const half8 wh = bh[s], wl = bl[s];
                    const int kn = min(k + 3, nk - 1);
                    bh[s] = wq[(size_t)kn * kstride]; bl[s] = wq[(size_t)kn * kstride + 2 * plane];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
                    accl = nm_mfma_lo<SINGLE>(ah, wl, accl);
                    accl = nm_mfma_lo<SINGLE>(al, wh, accl);
                }
            }
        }
        float s = 0.f, ss = 0.f;
        const int co = nh * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row_owned(rho)) {
                const int u = tl * 32 + rho;
                const int rz = type == 2 ? fb : f, ry = type == 0 ? u : (type == 1 ? fb : f), rx = type == 0 ? fb : u;
                const size_t dst = ((((size_t)n * OD + 2 * rz + pz) * OH + 2 * ry + py) * OW + 2 * rx + px) * p.Cout + co;
                const float v = nm_ld1<OH16>(p.out, dst) + (acc[r] + accl[r] * (1.0f / UP2C_SPLIT_SCALE));
                nm_st1<OH16>(p.out, dst, v);
                s += v; ss += v * v;
            }
        }
        if (p.part) {
            s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
            if (h == 0) { float* dst = p.part + (((size_t)n * p.nblk + slot_part) * p.Cout + co) * 2; dst[0] = s; dst[1] = ss; }
        }
    }
}

// conv_up2c_edge_kernel: one wave per item = 32 cells of one parity class on one of the twelve edges (the eight corners belong to
// the edges along z).  For every non-empty subset S of a row's border axes the signed set (S, parity) is applied with the rows
// outside the subset's cells zeroed; operands straight from global memory (a few thousand items in all).
template <bool SINGLE, int IO = 0>
__global__ __launch_bounds__(256, 4) void conv_up2c_edge_kernel(Up2cParams p, int TZ, int TY, int TX, int slot0) {
    constexpr bool IH16 = (IO & 1) != 0, OH16 = (IO & 2) != 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int nXY = 8 * TZ, nXZ = 8 * TY, nYZ = 8 * TX, per_frame = nXY + nXZ + nYZ;
    const long long gitem = (long long)blockIdx.x * 4 + wave;
    if (gitem >= (long long)p.N * per_frame) return;
    const int n = (int)(gitem / per_frame); int q = (int)(gitem % per_frame);
    const int slot = slot0 + q;
    // type 0: x-y edge (line along z), 1: x-z edge (along y), 2: y-z edge (along x); q -> (tile, parity class)
    int type, tile, par;
    if (q < nXY) { type = 0; tile = q / 8; par = q % 8; }
    else if (q < nXY + nXZ) { q -= nXY; type = 1; tile = q / 8; par = q % 8; }
    else { q -= nXY + nXZ; type = 2; tile = q / 8; par = q % 8; }
    const int pz = par >> 2, py = (par >> 1) & 1, px = par & 1;
    const int xb = px ? p.IW - 1 : 0, yb = py ? p.IH - 1 : 0, zb = pz ? p.ID - 1 : 0;
    // cell of row rho
    auto cell = [&](int rho, int& iz, int& iy, int& ix, bool& owned, int& bset) {
        const int u = tile * 32 + rho;
        if (type == 0) { iz = u; iy = yb; ix = xb; owned = u < p.ID; }
        else if (type == 1) { iz = zb; iy = u; ix = xb; owned = u < p.IH && u != yb; }
        else { iz = zb; iy = yb; ix = u; owned = u < p.IW && u != xb; }
        iz = min(iz, p.ID - 1); iy = min(iy, p.IH - 1); ix = min(ix, p.IW - 1);
        bset = (ix == xb ? 1 : 0) | (iy == yb ? 2 : 0) | (iz == zb ? 4 : 0);
    };
    int iz, iy, ix, bset; bool owned;
    cell(l31, iz, iy, ix, owned, bset);
    const int C16 = p.Cin >> 4, NH = p.Cout >> 5;
    const size_t plane = (size_t)p.Co_pad, kstride = 4 * plane;
    const bool affine = p.in_scale != nullptr;
    const int OD = 2 * p.ID, OH = 2 * p.IH, OW = 2 * p.IW;
    for (int nh = 0; nh < NH; ++nh) {
        f32x16 acc, accl;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }
        for (int S = 1; S < 8; ++S) {
            const bool rowon = owned && (bset & S) == S;
            if (__ballot(rowon) == 0ull) continue;                     // wave-uniform
            const int e = 8 + (S - 1) * 8 + par;
            const int ntaps = axes_taps(S);
            const half8* wq = p.wc + (size_t)set_tap_offset(e) * C16 * kstride + (size_t)h * plane + nh * 32 + l31;
            // taps in groups of three: the loads of a group are issued together; a corner set has one tap
            for (int c = 0; c < C16; ++c) {
                const int cbase = c * 16 + 8 * h;
                f32x4 sca = {0.f, 0.f, 0.f, 0.f}, sha = sca, scb = sca, shb = sca;
                if (affine) {
                    const float* ps = p.in_scale + (size_t)n * p.Cin + cbase; const float* ph = p.in_shift + (size_t)n * p.Cin + cbase;
                    sca = *reinterpret_cast<const f32x4*>(ps); scb = *reinterpret_cast<const f32x4*>(ps + 4);
                    sha = *reinterpret_cast<const f32x4*>(ph); shb = *reinterpret_cast<const f32x4*>(ph + 4);
                }
                for (int t0 = 0; t0 < ntaps; t0 += 3) {
                    f32x4 va[3], vb[3]; half8 wh[3], wl[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int t = min(t0 + u, ntaps - 1);
                        int d[3] = {0, 0, 0}, qd = t;                      // x, y, z offsets
#pragma unroll
                        for (int a = 0; a < 3; ++a) if (!((S >> a) & 1)) { d[a] = qd % 3 - 1; qd /= 3; }
                        const int gz = min(max(iz + d[2], 0), p.ID - 1), gy = min(max(iy + d[1], 0), p.IH - 1), gx = min(max(ix + d[0], 0), p.IW - 1);
                        ld8_raw<IH16>(p.in, ((((size_t)n * p.ID + gz) * p.IH + gy) * p.IW + gx) * p.Cin + cbase, va[u], vb[u]);
                        const half8* wt = wq + ((size_t)c * ntaps + t) * kstride;
                        wh[u] = wt[0]; wl[u] = wt[2 * plane];
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        if (t0 + u < ntaps) {
                            widen8<IH16>(va[u], vb[u]);
                            f32x4 xa = act4(va[u], sca, sha, affine, p.in_slope), xb2 = act4(vb[u], scb, shb, affine, p.in_slope);
                            if (!rowon) { xa = f32x4{0.f, 0.f, 0.f, 0.f}; xb2 = xa; }
                            half8 hi, lo;
                            split8(xa, xb2, hi, lo);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, wh[u], acc, 0, 0, 0);
                            accl = nm_mfma_lo<SINGLE>(hi, wl[u], accl);
                            accl = nm_mfma_lo<SINGLE>(lo, wh[u], accl);
                        }
                    }
                }
            }
        }
        // add to the main kernel's values, partial sums of the final values
        float s = 0.f, ss = 0.f;
        const int co = nh * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = (r & 3) + 8 * (r >> 2) + 4 * h;
            int rz, ry, rx, rb; bool ro;
            cell(rho, rz, ry, rx, ro, rb);
            if (ro) {
                const size_t dst = ((((size_t)n * OD + 2 * rz + pz) * OH + 2 * ry + py) * OW + 2 * rx + px) * p.Cout + co;
                const float v = nm_ld1<OH16>(p.out, dst) + (acc[r] + accl[r] * (1.0f / UP2C_SPLIT_SCALE));
                nm_st1<OH16>(p.out, dst, v);
                s += v; ss += v * v;
            }
        }
        if (p.part) {
            s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
            if (h == 0) { float* dst = p.part + (((size_t)n * p.nblk + slot) * p.Cout + co) * 2; dst[0] = s; dst[1] = ss; }
        }
    }
}

int g_cus = 0;

}  // namespace

bool nm_up2c_eligible(int ID, int IH, int IW, int Cin, int Cout, int ks, int stride, int pad) {
    return nm_ls().up2c && ks == 3 && stride == 1 && pad == 1 && ID % BZ == 0 && IH % BY == 0 && IW % BX == 0 && Cin % CG == 0 && Cout % 32 == 0 &&
           ((Cin == 64 && Cout == 32) || (nm_ls().up2c_all && Cin <= 128 && Cout <= 64));
}

size_t nm_up2c_weight_floats(int Cin, int Co_pad) {
    return (size_t)TOTAL_TAPS * (Cin >> 4) * 4 * Co_pad * 8 / 2;       // halves -> 4-byte units
}

int nm_launch_up2c_compose(const float* w, int Cout, int Cin, int Co_pad, void* packed, hipStream_t s) {
    if (Cin % 16 || Co_pad % 32 || Cout > Co_pad) { nm_set_error("up2c_compose: bad channels Cin=%d Cout=%d/%d", Cin, Cout, Co_pad); return NM_ERR_ARG; }
    hipLaunchKernelGGL(up2c_compose_kernel, dim3(2048), dim3(256), 0, s, w, Cout, Cin, Co_pad, reinterpret_cast<_Float16*>(packed));
    return nm_check_hip(hipGetLastError(), "up2c_compose launch");
}

static void shell_tiles(int ID, int IH, int IW, int& TZ, int& TY, int& TX) { TZ = (ID + 31) / 32; TY = (IH + 31) / 32; TX = (IW + 31) / 32; }
static int face_groups(int ID, int IH, int IW) { int TZ, TY, TX; shell_tiles(ID, IH, IW, TZ, TY, TX); return 2 * ID * TY + 2 * ID * TX + 2 * IH * TX; }
static int edge_items(int ID, int IH, int IW) { int TZ, TY, TX; shell_tiles(ID, IH, IW, TZ, TY, TX); return 8 * (TZ + TY + TX); }

int nm_up2c_blocks_per_frame(int ID, int IH, int IW) {
    return 8 * (ID / BZ) * (IH / BY) * (IW / BX) + 4 * face_groups(ID, IH, IW) + edge_items(ID, IH, IW);
}

int nm_launch_conv_up2c(const TensorRef& in, const void* packed, const float* bias, float* out, int Cout, int Co_pad, float* part,
                        hipStream_t s, int out_h) {
    if (!nm_up2c_eligible(in.D, in.H, in.W, in.C, Cout, 3, 1, 1) || !packed) { nm_set_error("conv_up2c: shape not eligible"); return NM_ERR_ARG; }
    if ((in.scale == nullptr) != (in.shift == nullptr)) { nm_set_error("conv_up2c: scale/shift must come together"); return NM_ERR_ARG; }
    static NmDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up2c_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up2c_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up2c_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up2c_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up2c_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up2c_kernel<false, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up2c_x16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up2c_x16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return nm_check_hip(e, "hipFuncSetAttribute(conv_up2c)");
        attr_set.mark();
    }
    const bool single = nm_conv_single() != 0;
    const bool io16 = in.h || out_h;
    if (io16 && !single) {
        nm_set_error("conv_up2c: 16-bit storage (input %d / output %d) needs conv mode 4", in.h, out_h); return NM_ERR_UNSUPPORTED;
    }
    const int io = (in.h ? 1 : 0) | (out_h ? 2 : 0);
    if (g_cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return nm_check_hip(hipErrorUnknown, "device query");
        g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    Up2cParams p;
    p.in = in.p; p.in_scale = in.scale; p.in_shift = in.shift; p.in_slope = in.slope;
    p.N = in.N; p.ID = in.D; p.IH = in.H; p.IW = in.W; p.Cin = in.C;
    p.wc = static_cast<const half8*>(packed); p.bias = bias; p.out = out; p.part = part;
    p.Cout = Cout; p.Co_pad = Co_pad;
    p.nbz = in.D / BZ; p.nby = in.H / BY; p.nbx = in.W / BX;
    p.nblk = nm_up2c_blocks_per_frame(in.D, in.H, in.W);
    p.diag = nm_ls().up2c_diag;
    const int bricks = p.nbz * p.nby * p.nbx, total = p.N * bricks;
    if (io == 3) hipLaunchKernelGGL((conv_up2c_kernel<true, 3>), dim3((unsigned)min(total, g_cus)), dim3(512), LDS_BYTES, s, p);
    else if (io == 2) hipLaunchKernelGGL((conv_up2c_kernel<true, 2>), dim3((unsigned)min(total, g_cus)), dim3(512), LDS_BYTES, s, p);
    else if (io == 1) hipLaunchKernelGGL((conv_up2c_kernel<true, 1>), dim3((unsigned)min(total, g_cus)), dim3(512), LDS_BYTES, s, p);
    // (the one-product modes keep the 32x32x16 kernel in fp32 storage too: their bfloat16-storage instantiations are bit-compared with
    //  it, tests/test_storage16_gpu.py; NM355_UP2C_X16=2 forces the 16x16x32 form there as well - A/B)
    else if (nm_ls().up2c_x16 >= 2 && single) hipLaunchKernelGGL(conv_up2c_x16_kernel<true>, dim3((unsigned)min(total, g_cus)), dim3(512), LDS_BYTES, s, p);
    else if (nm_ls().up2c_x16 && !single) hipLaunchKernelGGL(conv_up2c_x16_kernel<false>, dim3((unsigned)min(total, g_cus)), dim3(512), LDS_BYTES, s, p);
    else if (single) hipLaunchKernelGGL(conv_up2c_kernel<true>, dim3((unsigned)min(total, g_cus)), dim3(512), LDS_BYTES, s, p);
    else if (nm_ls().up2c_diag & 64) hipLaunchKernelGGL((conv_up2c_kernel<false, 0, true>), dim3((unsigned)min(total, g_cus)), dim3(512), LDS_BYTES, s, p);
    else hipLaunchKernelGGL(conv_up2c_kernel<false>, dim3((unsigned)min(total, g_cus)), dim3(512), LDS_BYTES, s, p);
    int rc = nm_check_hip(hipGetLastError(), "conv_up2c launch");
    if (rc) return rc;
    if (nm_ls().up2c_diag & 4) return NM_OK;
    int TZ, TY, TX; shell_tiles(in.D, in.H, in.W, TZ, TY, TX);
    const int fg = face_groups(in.D, in.H, in.W), ei = edge_items(in.D, in.H, in.W);
    const size_t face_lds = (size_t)(in.C / 16 * 4) * FPV * 16;
    if (io == 3) hipLaunchKernelGGL((conv_up2c_face_kernel<true, 3>), dim3((unsigned)(p.N * fg)), dim3(256), face_lds, s, p, TY, TX);
    else if (io == 2) hipLaunchKernelGGL((conv_up2c_face_kernel<true, 2>), dim3((unsigned)(p.N * fg)), dim3(256), face_lds, s, p, TY, TX);
    else if (io == 1) hipLaunchKernelGGL((conv_up2c_face_kernel<true, 1>), dim3((unsigned)(p.N * fg)), dim3(256), face_lds, s, p, TY, TX);
    else if (single) hipLaunchKernelGGL(conv_up2c_face_kernel<true>, dim3((unsigned)(p.N * fg)), dim3(256), face_lds, s, p, TY, TX);
    else hipLaunchKernelGGL(conv_up2c_face_kernel<false>, dim3((unsigned)(p.N * fg)), dim3(256), face_lds, s, p, TY, TX);
    rc = nm_check_hip(hipGetLastError(), "conv_up2c_face launch");
    if (rc) return rc;
    const long long items = (long long)p.N * ei;
    if (io == 3) hipLaunchKernelGGL((conv_up2c_edge_kernel<true, 3>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p, TZ, TY, TX, 8 * bricks + 4 * fg);
    else if (io == 2) hipLaunchKernelGGL((conv_up2c_edge_kernel<true, 2>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p, TZ, TY, TX, 8 * bricks + 4 * fg);
    else if (io == 1) hipLaunchKernelGGL((conv_up2c_edge_kernel<true, 1>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p, TZ, TY, TX, 8 * bricks + 4 * fg);
    else if (single) hipLaunchKernelGGL(conv_up2c_edge_kernel<true>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p, TZ, TY, TX, 8 * bricks + 4 * fg);
    else hipLaunchKernelGGL(conv_up2c_edge_kernel<false>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p, TZ, TY, TX, 8 * bricks + 4 * fg);
    return nm_check_hip(hipGetLastError(), "conv_up2c_edge launch");
}
