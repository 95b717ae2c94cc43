// Shared declarations of the nm355 HIP library (gfx950 / CDNA4 only).
//
// Activation layout everywhere inside the library: channels-last
//   [frame n][z][y][x][C]   fp32, C a multiple of 8
// so that a voxel's channel vector is one contiguous run (coalesced 16-B loads, and the
// K dimension of the implicit GEMM is contiguous for the MFMA A operand).
//
// "Lazy" tensors: a conv writes its raw output plus per-block GroupNorm partial sums;
// nm_gn_finalize turns those into per-(frame, channel) scale/shift, and the *consumer*
// applies  y = lrelu(x * scale + shift)  while it stages its input tile (TensorRef).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#define NM_OK 0
#define NM_ERR_ARG (-1)
#define NM_ERR_HIP (-2)
#define NM_ERR_STATE (-3)
#define NM_ERR_UNSUPPORTED (-4)
#define NM_ERR_RANGE (-5)
#define NM_ERR_INTERNAL (-6)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Second and third product of the split-fp16 scheme (hi*lo, lo*hi).  SINGLE = the reduced-precision mode (nm_set_conv_mode 3): products
// of the fp16-rounded operands only, fp32 accumulation - what a bf16 / fp16 autocast training step computes; the operand loads that
// only feed these calls disappear with them.
typedef _Float16 nm_half8 __attribute__((ext_vector_type(8)));
template <bool SINGLE>
__device__ __forceinline__ f32x16 nm_mfma_lo(nm_half8 a, nm_half8 b, f32x16 c) {
    if constexpr (SINGLE) return c;
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}


// ---- 16-bit storage (conv mode 4, 'bf16': BASELINE config 3) -----------------------------------------------------------------------
// In this mode the activations and activation gradients of the training path with >= nm_ls().store16_min voxels per frame are stored
// as bfloat16 (same channels-last layout, half the bytes); weights, GroupNorm statistics / scale / shift, partial sums, accumulators,
// the Adam state and every tensor below the threshold stay fp32.  A tensor's element type travels as TensorRef::h (raw gradient
// pointers: as an explicit flag of the launcher); pointers stay `float*` and element offsets are scaled by the accessors below.
// Conversions: bf16 -> f32 is a shift / mask (exact), f32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32).
typedef unsigned nm_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 nm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float nm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned nm_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(nm_f32x2{a, b}, nm_bf16x2));
}
// bits of a float passed BY VALUE: __builtin_bit_cast applied directly to an element of an ext_vector reached through a reference
// (`a[1]` of an `f32x4&`) was compiled as a read of element 0 by hipcc 7.0 - always go through this
__device__ __forceinline__ unsigned nm_fbits(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float nm_bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float nm_bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ f32x4 nm_bf4_to_f32(nm_u32x2 u) { return f32x4{nm_bf_lo(u[0]), nm_bf_hi(u[0]), nm_bf_lo(u[1]), nm_bf_hi(u[1])}; }
__device__ __forceinline__ nm_u32x2 nm_f32_to_bf4(const f32x4& v) { return nm_u32x2{nm_pk_bf16(v[0], v[1]), nm_pk_bf16(v[2], v[3])}; }
// four consecutive elements starting at ELEMENT offset e (e % 4 == 0) of a tensor stored as fp32 (H = false) or bf16 (H = true)
template <bool H> __device__ __forceinline__ f32x4 nm_ld4(const float* base, size_t e) {
    if constexpr (H) return nm_bf4_to_f32(*reinterpret_cast<const nm_u32x2*>(reinterpret_cast<const unsigned short*>(base) + e));
    else return *reinterpret_cast<const f32x4*>(base + e);
}
template <bool H> __device__ __forceinline__ void nm_st4(float* base, size_t e, const f32x4& v) {
    if constexpr (H) *reinterpret_cast<nm_u32x2*>(reinterpret_cast<unsigned short*>(base) + e) = nm_f32_to_bf4(v);
    else *reinterpret_cast<f32x4*>(base + e) = v;
}
template <bool H> __device__ __forceinline__ float nm_ld1(const float* base, size_t e) {
    if constexpr (H) return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(base)[e] << 16);
    else return base[e];
}
template <bool H> __device__ __forceinline__ void nm_st1(float* base, size_t e, float v) {
    if constexpr (H) reinterpret_cast<unsigned short*>(base)[e] = (unsigned short)(nm_pk_bf16(v, 0.f) & 0xffffu);
    else base[e] = v;
}
// the same with a wave-uniform runtime flag (the HBM-bound elementwise kernels: one uniform branch per access costs nothing there)
__device__ __forceinline__ f32x4 nm_ld4(const float* base, size_t e, int h) { return h ? nm_ld4<true>(base, e) : nm_ld4<false>(base, e); }
__device__ __forceinline__ void nm_st4(float* base, size_t e, const f32x4& v, int h) { if (h) nm_st4<true>(base, e, v); else nm_st4<false>(base, e, v); }
__device__ __forceinline__ float nm_ld1(const float* base, size_t e, int h) { return h ? nm_ld1<true>(base, e) : nm_ld1<false>(base, e); }
__device__ __forceinline__ void nm_st1(float* base, size_t e, float v, int h) { if (h) nm_st1<true>(base, e, v); else nm_st1<false>(base, e, v); }
// pointer to ELEMENT e of a tensor of either type, as the float* the launchers pass around (host and device)
__host__ __device__ inline float* nm_eptr(float* base, size_t e, int h) { return h ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(base) + e) : base + e; }
__host__ __device__ inline const float* nm_eptr(const float* base, size_t e, int h) { return h ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(base) + e) : base + e; }


// ---- exact xor-partner exchange on the vector ALU ------------------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 on gfx950: a trip through the LDS crossbar (address VALU + ~100 cycles + s_waitcnt) per step,
// six dependent trips per 64-lane sum.  The VRNN kernels are nothing but such sums (one per output row): a rollout step spent 9.6 of its
// 15 us in the reductions of ONE workgroup (tools/diag_chain_stamps.py).  nm_sx(x, off) returns the value of lane (lane ^ off) for
// off = 1 .. 32 from DPP / lane-swap instructions only - EXACTLY the partner __shfl_xor reads (tools/calib/xor_partners.hip), so every
// butterfly keeps its association order and its bits:
//   1, 2: quad_perm;  4: row_shl:4 into banks 0, 2 + row_shr:4 into banks 1, 3;  8: row_ror:8;
//   16: v_permlane16_swap (odd rows of one copy <-> even rows of the other);  32: v_permlane32_swap (gfx950).
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ int nm_dpp_mov(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, BANK, false); }
__device__ __forceinline__ int nm_sx(int x, int off) {
    switch (off) {
        case 1: return nm_dpp_mov<0xB1>(x, x);
        case 2: return nm_dpp_mov<0x4E>(x, x);
        case 4: { const int r = nm_dpp_mov<0x104, 0x5>(x, x); return nm_dpp_mov<0x114, 0xA>(r, x); }
        case 8: return nm_dpp_mov<0x128>(x, x);
        case 16: { const auto p = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false); return (threadIdx.x & 16) ? (int)p[0] : (int)p[1]; }
        case 32: { const auto p = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false); return (threadIdx.x & 32) ? (int)p[0] : (int)p[1]; }
        default: return __shfl_xor(x, off);
    }
}
__device__ __forceinline__ float nm_sx(float x, int off) { return __builtin_bit_cast(float, nm_sx(__builtin_bit_cast(int, x), off)); }
// the same with the distance as a template argument (where `off` is not a constant the switch above is evaluated as SIX exchanges and
// five selects - measured: a reduction loop that hipcc did not unroll ran 2.5x slower than the ds_bpermute form it replaced)
template <int OFF> __device__ __forceinline__ float nm_sxc(float xf) {
    const int x = __builtin_bit_cast(int, xf);
    int r;
    if constexpr (OFF == 1) r = nm_dpp_mov<0xB1>(x, x);
    else if constexpr (OFF == 2) r = nm_dpp_mov<0x4E>(x, x);
    else if constexpr (OFF == 4) { const int q = nm_dpp_mov<0x104, 0x5>(x, x); r = nm_dpp_mov<0x114, 0xA>(q, x); }
    else if constexpr (OFF == 8) r = nm_dpp_mov<0x128>(x, x);
    else if constexpr (OFF == 16) { const auto p = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false); r = (threadIdx.x & 16) ? (int)p[0] : (int)p[1]; }
    else { static_assert(OFF == 32, "distance"); const auto p = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false); r = (threadIdx.x & 32) ? (int)p[0] : (int)p[1]; }
    return __builtin_bit_cast(float, r);
}

// A tensor reference with an optional pending per-(n,c) affine + leaky-relu.
struct TensorRef {
    const float* p;       // [N][D][H][W][C]
    const float* scale;   // [N][C] or nullptr  (nullptr => identity affine)
    const float* shift;   // [N][C] or nullptr
    float slope;          // leaky-relu slope applied after the affine; 1.0f => none
    int N, D, H, W, C;
    // Brick-sparse tensor (the first layer's output in inference, nm_conv.hip "sparse first layer"): brickmap[n][4x8x8 brick] == 0
    // means the brick was not written to p and equals the frame-independent tensor `alt` ([D][H][W][C], no frame stride) there.
    // Only the k2 s2 split-fp16 pool kernel reads such tensors; every other launcher rejects them.
    const float* alt = nullptr;
    const unsigned char* brickmap = nullptr;
    // Un-materialised residual sum (Res3DBlock output = GN(conv) + skip, vox_modules.py:44-47): the tensor's value is
    // T(p; scale, shift, slope) + T(p2; scale2, shift2, slope2), evaluated exactly as apply2 would have stored it.  Only the k2 s2
    // split-fp16 pool kernel reads such tensors (the block's only consumer in the inference encoder); others reject them.
    const float* p2 = nullptr; const float* scale2 = nullptr; const float* shift2 = nullptr; float slope2 = 1.0f;
    int h = 0;            // element type of p: 0 fp32, 1 bfloat16 (16-bit storage mode; the launchers that cannot read it reject it)
};

struct ConvGeom {
    int ks, stride, pad;
    int OD, OH, OW;
    int Cout;      // real output channels (multiple of 8, or 24 for the heat-map heads)
    int Co_pad;    // packed weight columns (multiple of 32)
    int up2 = 0;   // the input tensor is stored at half resolution; its trilinear x2 upsampling is convolved
    const void* up2c = nullptr;   // up2: composite weight sets of nm_up2c.hip (null: the layer stays on conv_f16s<.., UP2>)
};

void nm_set_error(const char* fmt, ...);
// "done once" flags of per-DEVICE state (hipFuncSetAttribute, hipMemcpyToSymbol): one bit per device id, so that a second context on
// another device of the same process sets its own attributes (a process-wide bool left it launching with the default LDS limit)
struct NmDeviceOnce {
    unsigned long long mask = 0;
    static unsigned long long bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    bool done() const { return (mask & bit()) != 0; }
    void mark() { mask |= bit(); }
};
int nm_check_hip(hipError_t e, const char* what);
// the catch-all of every extern "C" entry point (no C++ exception crosses the ABI): message for nm_last_error(), NM_ERR_INTERNAL
int nm_abi_catch(const char* fn) noexcept;

// ---- per-context launch state ---------------------------------------------------------
// Everything a kernel launcher consults besides its arguments lives in the nm_ctx ("separate ctxs are
// independent"): the conv arithmetic mode, the launch profiler's records, the word GroupNorm finalisation reports non-finite
// statistics into, and the NM355_* A/B switches (read from the environment once, when the context is created).  Every ABI entry
// point selects its context's state for the calling thread (NmScope, nm_ctx.h); launchers read it through nm_ls().
struct NmProfRec { hipEvent_t a, b; int variant; double flops; };
struct NmLaunchState {
    int conv_mode = 1;         // 0: exact fp32 MFMA everywhere, 1: split-fp16 MFMA where the layer shape allows
    bool f16p_all = false;     // mode 2: split-fp16 with conv_f16p on every eligible layer (parity tests of its multi-cout-group path)
    bool single = false;       // mode 3 / 4: the split-fp16 kernels keep only the hi x hi product (SINGLE instantiations)
    bool store16 = false;      // mode 4 ('bf16'): mode 3's arithmetic + bfloat16 storage of the training path's large tensors
    int op_in_h = 0, op_out_h = 0;   // op-level entry points (nm_op_*): element type of the input-side / output-side tensors (nm_op_set_storage16)
    int store16_min = 32768;   // ... those with at least this many voxels per frame (NM355_STORE16_MIN; 32^3: everything above the hourglass)
    // persistent rollout kernel (nm_vrnn.hip vrnn_prior_chain_kernel): polls before a spin gives up (NM355_CHAIN_SPIN), trailing
    // workgroups NOT launched (NM355_CHAIN_DROP_WG: the test hook that stands in for a workgroup that never becomes resident), and the
    // co-residency verdict of this context's device (-1: not asked yet, 0: the chain does not fit, 1: it does)
    int chain_spin = 1 << 20, chain_drop = 0, chain_fits = -1, post_chain_fits = -1, chain_stat_delay = 0, chain_wgpoll = 1, chain_xcd_nogo = 0, chain_xcd = 1, chain_cus = 0;
    bool prof_on = false;
    hipStream_t prof_stream = nullptr;     // main stream of the profiled context; prof_all: launches on any stream are recorded
    bool prof_all = false;
    double prof_min_flops = 0.0;           // launches below this algorithmic work are not bracketed (nm_prof_enable mode 3)
    std::vector<NmProfRec> prof;           // records of the current window
    std::vector<hipEvent_t> event_pool;
    unsigned* nf_flag = nullptr;           // sticky device word gn_finalize ORs a 1 into (null: no reporting)
    // A/B and diagnostic switches (NM355_SUPERTILE, _SMALL16, _KSPLIT, _OCC16, _POOL16, _F16P2, _F16P, _WGRAD_TR, _UP2C, _UP2C_DIAG,
    // _VRNN_MID, _VRNN_GEMM, _VRNN_GRAPH, _SPARSE_FIRST)
    int supertile, small16, ksplit, occ16, pool16, f16p2, f16p, pool_q, occ_flags, gnb_apply4, defer_sums, wgrad_async, wgrad_wgs, wgrad_tr, wgrad_u, tail_rank1, wgrad_z, up2c, up2c_diag, vrnn_mid, vrnn_postmid, vrnn_nb, vrnn_gemm, vrnn_graph, sparse_first, gn_diag, lazy_res, adjust_split, hg_core, f16p_dma, clip_occ_mfma, vrnn_chain, wgrad_k2f16, convt_f16, k5_two, clip_late, gnb_u8, adj_zwalk, f16q2, vrnn_post_chain, f16r, up2_mat, conv_wgs, up2c_x16, up2c_all, f16p_late, p2_defer, fast_decode;
    NmLaunchState();
};
NmLaunchState& nm_ls();        // the state of the context whose ABI call runs on this thread

// ---- nm_conv.hip -------------------------------------------------------------------
// packed weight layout: [tap][Cin/4][Co_pad][4]
size_t nm_packed_weight_floats(int ks, int Cin_pad, int Co_pad);
int nm_launch_pack_conv_weight(const float* w_oidhw, int Cout, int Cin, int ks, float* packed,
                               int Cin_pad, int Co_pad, hipStream_t s);
// number of per-frame partial blocks the conv epilogue writes (for sizing `part`)
int nm_conv_blocks_per_frame(const ConvGeom& g, int Cin = 16 /* decides the kernel for up2 layers */);
int nm_launch_conv(const TensorRef& in, const float* w_packed, const float* bias, float* out,
                   const ConvGeom& g, float* part /*[N][nblk][Cout][2] or null*/, hipStream_t s,
                   int cin_real = 0 /* un-padded Cin, for the profiler's FLOP count */,
                   const void* w_packed16 = nullptr /* split-fp16 weights; enables the fp16-split kernel */,
                   int out_h = 0 /* 1: `out` is bfloat16 (16-bit storage, conv mode 4); the input's type is in.h */);
// conv arithmetic: 0 = exact fp32 MFMA, 1 = split-fp16 MFMA (3 products, fp32 accumulate) where Cin % 16 == 0
void nm_conv_set_mode(int mode);
int nm_conv_get_mode();
int nm_conv_single();            // 1 in conv mode 3 (hi x hi products only)
int nm_launch_pack_conv_weight16(const float* w_oidhw, int Cout, int Cin, int ks, void* packed, int Co_pad, hipStream_t s);
// All the per-layer weight packs of one nm_ctx_set_weights as ONE launch (a training step re-packs every conv after Adam: ~330 tiny
// dependent launches otherwise).  A job packs the logical weight W(co, ci, tap), co < Cout, ci < Cin, into the fp32 layout of
// nm_launch_pack_conv_weight (wp) and, when wp16 is set, the split-fp16 layout of nm_launch_pack_conv_weight16.  The source is an
// OIDHW tensor with src_cin input channels per row: W = src[(co * src_cin + ci) * taps + tap], or with flip the data-gradient
// weight (flipped taps, transposed channels): W = src[(ci * src_cin + co) * taps + (taps - 1 - tap)].
struct NmPackJob {
    const float* src; float* wp; void* wp16;
    int Cout, Cin, ks, Cin_pad, Co_pad, src_cin, flip, blk0, nblk;
    int src_rows;      // rows of `src` (its leading dimension's extent): reads beyond it are zeros (padded layers); 0 = no limit
};
int nm_pack_job_blocks(const NmPackJob& j);       // blocks the job takes in the launch (fills nothing)
int nm_launch_pack_jobs(const NmPackJob* device_jobs, int njobs, int total_blocks, hipStream_t s);
// first layer: occupancy channel as a taps-as-K GEMM + weight-only constant field (see nm_conv.hip)
int nm_occ_blocks_per_frame(int G);
int nm_launch_pack_occ_weight(const float* w_oidhw, int Cout, float* tmp, float* packed, int Co_pad, hipStream_t s);
int nm_launch_conv_k5occ(const float* occ, int N, int G, const float* w_packed, const float* field, float* out, int Cout,
                         int Co_pad, float* part, hipStream_t s, unsigned char* brickmap = nullptr, const float* field_part = nullptr,
                         unsigned char* flags = nullptr /* N * bricks bytes of scratch: per-brick occupancy pre-filter of the sparse form */,
                         int out_h = 0 /* 1: `out` is bfloat16 (dense form only) */);
// true when nm_launch_conv sends a k2 s2 p0 conv of this input to conv_pool_f16s_kernel (the consumer of a brick-sparse tensor)
bool nm_conv_pool16_eligible(int Cin, int OD, int OH, int OW, bool have_w16);
void nm_conv_prof_enable(int on, hipStream_t stream);
int nm_conv_prof_collect(int variant, double* ms_total, double* flops_total, long long* launches);
void nm_conv_prof_reset();
// a launch (sequence) outside nm_conv.hip bracketed for the same profiler: begin records the first event when the profiler is on for
// stream s, end the second and files the record under `variant` (nm_prof_kernel_name)
struct NmProfScope {
    bool on = false; NmProfRec rec{}; hipStream_t s = nullptr;
    NmProfScope(hipStream_t stream, double flops, int variant);
    ~NmProfScope();
};

// ---- nm_elem.hip -------------------------------------------------------------------
int nm_launch_gn_finalize(const float* part, int N, int nblk, int C, int groups, double count,
                          const float* gamma, const float* beta, float eps, float* scale,
                          float* shift, hipStream_t s, double* chsum = nullptr);
// chsum (training): [N][C][2] doubles (sum y, sum y^2 per channel) for nm_launch_gnb_finalize; available when 256 % (C / groups) == 0
bool nm_gn_finalize_has_chsum(int C, int groups);
// diagnostic reference (NM355_GN_DIAG=1): statistics straight from the stored tensor, two passes in fp64
int nm_launch_gn_direct(const float* x, int N, int voxels, int C, int groups, const float* gamma, const float* beta, float eps,
                        float* scale, float* shift, hipStream_t s, double* chsum);
// sticky device word (ctx-owned) that gn_finalize ORs a 1 into when a conv's statistics are not finite; null: no reporting
void nm_elem_set_nonfinite_flag(unsigned* flag);
int nm_launch_nonfinite_scan(const float* x, size_t n, unsigned* flag, hipStream_t s);
// partial sums of an already materialised raw tensor (producers without a stats epilogue)
int nm_stats_blocks_per_frame(int voxels);
int nm_launch_gn_partials(const float* x, int N, int voxels, int C, float* part, hipStream_t s, int h = 0 /* x is bfloat16 */);
int nm_launch_apply2(const TensorRef& a, const TensorRef* b, float* out, hipStream_t s, int out_h = 0 /* 1: out is bfloat16 */);
// w_t: weights transposed to [tap][Cin][Cout] by nm_launch_transpose_convT_weight
int nm_launch_transpose_convT_weight(const float* w_iodhw, int Cin, int Cout, float* out, hipStream_t s);
// out_mul (device scalar the result is multiplied by before the bias is added) exists on the f16 matrix-core kernel only: callers test
// nm_convT2_f16_eligible first
int nm_launch_convT2(const TensorRef& in, const float* w_t, const float* bias, float* out,
                     int Cout, int OD, int OH, int OW, hipStream_t s, int out_h = 0, const float* out_mul = nullptr);
bool nm_convT2_f16_eligible(const TensorRef& in, int Cout, int OD, int OH, int OW, int out_h);
int nm_launch_upsample2(const TensorRef& in, float* out, hipStream_t s, int out_h = 0);
int nm_launch_pack_input(const float* vox, int B, int T, int G, int mean_over_t, float* out,
                         hipStream_t s);
int nm_launch_mean_t(const float* vox, int B, int T, size_t G3, float* out, hipStream_t s);
int nm_launch_cl_to_ncdhw(const TensorRef& in, float* out, hipStream_t s);
// in holds N frames picked with a stride (frame n of the output = frame n*frame_stride of in.p)
int nm_launch_cl_to_ncdhw_strided(const TensorRef& in, int frame_stride, float* out, hipStream_t s);
int nm_launch_ncdhw_to_cl(const float* in, int N, int voxels, int C, float* out, hipStream_t s);
