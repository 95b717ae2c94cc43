// The detector as a stream of kernel launches (host side of the library).
//
// Frames of all clips are folded into one batch F = B*T for the per-frame encoder and the
// decoder (the only cross-frame couplings are the clip-mean heat-map, the first frame's
// feature / gaussians, and the losses: SURVEY §5 "Long-context").  Activations are
// channels-last and "lazy": a conv leaves its raw output plus GroupNorm scale/shift, the
// consumer applies them while staging (nm_conv.hip).  Residual sums are materialised by
// apply2.
//
// Reference call stack being replaced: model/kypt_detector.py:81-169 (forward),
// :299-364 (VoxToKyptNet), :388-460 (KyptToVoxNet), :213-241 (decode_from_dyna),
// modules/vox_modules.py:8-120.
#include "nm_ctx.h"
#include "nm_heads.h"
#include "nm_grad.h"
#include "nm_heads_bwd.h"
#include "nm_up2c.h"
#include "nm_hgcore.h"
#include <cmath>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <functional>

// ---- what a training forward leaves behind for nm_detector_backward (all pointers into ctx->ws_t or caller buffers) --------
struct ConvRec {
    const ConvW* w = nullptr; const NormW* gn = nullptr;
    TensorRef in{}, out{};
    TensorRef upmat{};        // up2 layers whose forward materialised the upsampled (activated) input: that tensor (p == nullptr: fused staging, the backward pass rebuilds it)                       // lazy input / lazy output (raw conv result + pending GN affine + slope)
    const float* fpart = nullptr; int nblk = 0;  // forward GroupNorm partial sums of the conv epilogue
    const double* chsum = nullptr;               // their per-channel totals (left by the forward finalisation when it can)
    int stride = 1, pad = 0; bool up2 = false;
};
struct ResRec { ConvRec c1, c2, cs; bool has_skip = false; };
struct UpRec { const UpW* w = nullptr; TensorRef in{}, out{}; const float* fpart = nullptr; int nblk = 0; const double* chsum = nullptr; };
struct HgRec { ResRec s1, e1, s2, e2, s3, e3, d3, d2, d1; ConvRec p1, p2, p3; UpRec u3, u2, u1; };
struct FeatRec {
    const FeatNetW* w = nullptr; const float* occ = nullptr; int N = 0, G = 0;
    TensorRef first{}; const float* fpart0 = nullptr; int nblk0 = 0; const double* chsum0 = nullptr;
    ConvRec p1, p3; ResRec r2, r5; HgRec hg;
};
struct TrainTape {
    bool valid = false;
    int B = 0, T = 0, affinity_on = 0;
    FeatRec frame, clip;
    ConvRec head, clip_head, adjust, d1, d4, d8, d11;
    const float *vox = nullptr, *feat = nullptr, *head_out = nullptr, *clip_head_out = nullptr, *heat_part = nullptr, *heat_mean = nullptr;
    const float *tail_part = nullptr, *aff = nullptr, *table = nullptr, *clip_in = nullptr;
    const float *keypoints = nullptr, *recon = nullptr;       // caller buffers (kept alive by the Python autograd node)
    size_t fwd_top = 0;                                       // ws_t high-water mark of the forward
};

namespace {

const float LRELU = 0.01f;
const int FEAT = 128;
const size_t FRAME_CHUNK = 64;     // frames per pass through the conv stacks (bounds the workspace)

// ------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------
struct CopyItem { float* dst; const float* src; long long numel; long long chunk0; };
#define COPY_CHUNK 1024
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyItem* __restrict__ items, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].chunk0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1; }
    const CopyItem it = items[lo];
    const long long i0 = ((long long)blockIdx.x - it.chunk0) * COPY_CHUNK, i1 = min(it.numel, i0 + COPY_CHUNK);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) it.dst[i] = it.src[i];
}

struct Loader {
    nm_ctx* c;
    const std::map<std::string, std::pair<const float*, int64_t>>& sd;
    int rc = NM_OK;
    std::vector<CopyItem> copies;

    // one launch for all the conv weight packs (forward layouts and, in training, the data-gradient layouts)
    std::vector<NmPackJob> packs;
    void pack(const float* src, float* wp, void* wp16, int Cout, int Cin, int ks, int Cin_pad, int Co_pad, int src_cin, int flip, int src_rows = 0) {
        NmPackJob j; j.src = src; j.wp = wp; j.wp16 = wp16; j.Cout = Cout; j.Cin = Cin; j.ks = ks; j.Cin_pad = Cin_pad; j.Co_pad = Co_pad;
        j.src_cin = src_cin; j.flip = flip; j.blk0 = 0; j.nblk = 0; j.src_rows = src_rows;
        packs.push_back(j);
    }
    int flush_packs() {
        if (packs.empty()) return NM_OK;
        int blocks = 0;
        for (NmPackJob& j : packs) { j.blk0 = blocks; j.nblk = nm_pack_job_blocks(j); blocks += j.nblk; }
        const size_t bytes = packs.size() * sizeof(NmPackJob);
        c->host_table3.assign(reinterpret_cast<const char*>(packs.data()), reinterpret_cast<const char*>(packs.data()) + bytes);
        if (c->pack_table_cap < bytes) {
            if (c->pack_table) { (void)hipDeviceSynchronize(); (void)hipFree(c->pack_table); }
            c->pack_table = nullptr; c->pack_table_cap = 0;
            if (hipMalloc(&c->pack_table, bytes) != hipSuccess) { nm_set_error("set_weights: hipMalloc(pack table) failed"); return NM_ERR_HIP; }
            c->pack_table_cap = bytes;
        }
        int r = nm_check_hip(hipMemcpyAsync(c->pack_table, c->host_table3.data(), bytes, hipMemcpyHostToDevice, c->stream), "set_weights: pack table");
        if (r) return r;
        return nm_launch_pack_jobs(static_cast<const NmPackJob*>(c->pack_table), (int)packs.size(), blocks, c->stream);
    }

    // one launch for all the plain copies (biases, GroupNorm affine, VRNN matrices ...)
    int flush_copies() {
        if (copies.empty()) return NM_OK;
        long long chunks = 0;
        for (CopyItem& it : copies) { it.chunk0 = chunks; chunks += (it.numel + COPY_CHUNK - 1) / COPY_CHUNK; }
        const size_t bytes = copies.size() * sizeof(CopyItem);
        c->host_table2.assign(reinterpret_cast<const char*>(copies.data()), reinterpret_cast<const char*>(copies.data()) + bytes);
        if (c->copy_table_cap < bytes) {
            if (c->copy_table) { (void)hipDeviceSynchronize(); (void)hipFree(c->copy_table); }
            c->copy_table = nullptr; c->copy_table_cap = 0;
            if (hipMalloc(&c->copy_table, bytes) != hipSuccess) { nm_set_error("set_weights: hipMalloc(copy table) failed"); return NM_ERR_HIP; }
            c->copy_table_cap = bytes;
        }
        int r = nm_check_hip(hipMemcpyAsync(c->copy_table, c->host_table2.data(), bytes, hipMemcpyHostToDevice, c->stream), "set_weights: copy table");
        if (r) return r;
        hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)chunks), dim3(256), 0, c->stream, static_cast<const CopyItem*>(c->copy_table), (int)copies.size());
        return nm_check_hip(hipGetLastError(), "set_weights: multi copy");
    }

    const float* get(const std::string& name, int64_t numel) {
        auto it = sd.find(name);
        if (it == sd.end()) { if (!rc) { nm_set_error("set_weights: missing key '%s'", name.c_str()); rc = NM_ERR_ARG; } return nullptr; }
        if (it->second.second != numel) {
            if (!rc) { nm_set_error("set_weights: '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second.second, (long long)numel); rc = NM_ERR_ARG; }
            return nullptr;
        }
        return it->second.first;
    }
    float* copy(const std::string& name, int64_t numel) {
        const float* src = get(name, numel);
        if (!src) return nullptr;
        float* dst = nm_ctx_weight_alloc(c, numel);
        if (!dst) { if (!rc) { nm_set_error("set_weights: hipMalloc failed"); rc = NM_ERR_HIP; } return nullptr; }
        copies.push_back(CopyItem{dst, src, (long long)numel, 0});      // ~240 small tensors: copied by ONE kernel at the end (flush_copies)
        return dst;
    }
    // pad16: the layer's input tensor is built by the library itself with Cin rounded up to 16 (the decoder's 179-channel combined
    // representation), which puts the layer on the split-fp16 kernels
    // cout_layer (> Cout): the layer runs with its output channels zero-padded to cout_layer inside the library (the heat-map heads
    // for keypoint counts that are not multiples of 8: every tensor the conv kernels touch keeps C % 8 == 0); the state_dict tensors
    // keep Cout rows (ConvW::Cout_src), the padded rows of the packed weights and of the bias are zeros.
    // csel_min: input channels that need a data gradient (rounded up to 8; default: Cin rounded down to 8)
    ConvW conv(const std::string& p, int Cout, int Cin, int ks, bool pad16 = false, int cout_layer = 0, int csel_min = 0) {
        ConvW w; w.Cin = Cin; w.Cout = cout_layer > Cout ? cout_layer : Cout; w.Cout_src = Cout; w.ks = ks;
        w.Cin_pad = pad16 ? ((Cin + 15) & ~15) : ((Cin + 7) & ~7); w.Co_pad = (w.Cout + 31) & ~31; w.key = p;
        w.csel = csel_min > 0 ? std::min(w.Cin_pad, (csel_min + 7) & ~7) : (Cin & ~7);
        const float* src = get(p + ".weight", (int64_t)Cout * Cin * ks * ks * ks);
        if (w.Cout == Cout) w.bias = copy(p + ".bias", Cout);
        else {
            const float* bsrc = get(p + ".bias", Cout);
            w.bias = nm_ctx_weight_alloc(c, w.Cout);
            if (!w.bias) { if (!rc) { nm_set_error("set_weights: hipMalloc failed"); rc = NM_ERR_HIP; } return w; }
            (void)hipMemsetAsync(w.bias, 0, (size_t)w.Cout * sizeof(float), c->stream);      // (flush_copies runs behind it on the same stream)
            if (bsrc) copies.push_back(CopyItem{w.bias, bsrc, (long long)Cout, 0});
        }
        if (!src) return w;
        w.wp = nm_ctx_weight_alloc(c, nm_packed_weight_floats(ks, w.Cin_pad, w.Co_pad));
        if (!w.wp) { if (!rc) { nm_set_error("set_weights: hipMalloc failed"); rc = NM_ERR_HIP; } return w; }
        int r = NM_OK;
        if ((Cin % 8 == 0 || pad16) && Cin >= 16) {                // (Cin % 16 == 8: zero-padded, used by the small-volume core only)
            w.wp16 = nm_ctx_weight_alloc(c, nm_packed_weight_floats(ks, (Cin + 15) & ~15, w.Co_pad));      // same byte count as fp32
            if (!w.wp16) { if (!rc) { nm_set_error("set_weights: hipMalloc failed"); rc = NM_ERR_HIP; } return w; }
        }
        if (ks == 5) {         // the first layers: packed at once - first_layer_tables() runs a conv with these weights inside set_weights
            r = nm_launch_pack_conv_weight(src, Cout, Cin, ks, w.wp, w.Cin_pad, w.Co_pad, c->stream);
            if (!r && w.wp16) r = nm_launch_pack_conv_weight16(src, Cout, Cin, ks, w.wp16, w.Co_pad, c->stream);
        } else pack(src, w.wp, w.wp16, Cout, Cin, ks, w.Cin_pad, w.Co_pad, Cin, 0);      // (rows >= Cout of the packed form: zeros)
        if (!r && c->training && ks != 5) r = dgrad_packs(src, w);
        if (r && !rc) rc = r;
        return w;
    }
    // weights of the data-gradient convolution (training only): dX = conv(dY, flip/transpose(W)) for ks 1/3 (stride 1),
    // dX = convT2(dY, W) for the k2 s2 pool convs
    int dgrad_packs(const float* src, ConvW& w) {
        const int taps = w.ks * w.ks * w.ks;
        w.cd_pad = (w.csel + 31) & ~31;
        if (w.ks == 2) {
            w.wt = nm_ctx_weight_alloc(c, (size_t)w.Cin * w.Cout * 8);
            if (!w.wt) { nm_set_error("set_weights: hipMalloc failed"); return NM_ERR_HIP; }
            return nm_launch_transpose_convT_weight(src, w.Cout, w.Cin, w.wt, c->stream);
        }
        const size_t fl = nm_packed_weight_floats(w.ks, w.Cout, w.cd_pad);
        w.wd = nm_ctx_weight_alloc(c, fl);
        if (!w.wd) { nm_set_error("set_weights: hipMalloc failed"); return NM_ERR_HIP; }
        if (w.Cout % 16 == 0) {
            w.wd16 = nm_ctx_weight_alloc(c, fl);
            if (!w.wd16) { nm_set_error("set_weights: hipMalloc failed"); return NM_ERR_HIP; }
        }
        // the flipped / transposed weight (csel "output" channels = the input channels the gradient is wanted for) read in place
        // (src_rows: a padded layer's source has Cout_src rows, and csel may exceed Cin by the rounding - both read as zeros)
        pack(src, w.wd, w.wd16, w.csel, w.Cout, w.ks, w.Cout, w.cd_pad, w.Cin, 1, w.Cout_src);
        (void)taps;
        return NM_OK;
    }
    // composite weight sets of the fused-upsample layers (nm_up2c.hip); a null wup leaves the layer on conv_f16s<.., UP2>
    void up2_sets(const std::string& p, ConvW& w, int coarse) {
        w.wup = nullptr;
        // only for the layer shapes conv_up2c takes (the composition is a 27-tap fp64 kernel over 8 x 27 x Cin x Cout items and
        // runs after every optimizer step)
        if (w.ks != 3 || !nm_up2c_eligible(coarse, coarse, coarse, w.Cin, w.Cout, 3, 1, 1)) return;
        const float* src = get(p + ".weight", (int64_t)w.Cout * w.Cin * 27);
        if (!src) return;
        w.wup = nm_ctx_weight_alloc(c, nm_up2c_weight_floats(w.Cin, w.Co_pad));
        if (!w.wup) { if (!rc) { nm_set_error("set_weights: hipMalloc failed"); rc = NM_ERR_HIP; } return; }
        const int r = nm_launch_up2c_compose(src, w.Cout, w.Cin, w.Co_pad, w.wup, c->stream);
        if (r && !rc) rc = r;
    }
    NormW norm(const std::string& p, int C) {
        NormW n; n.C = C; n.groups = C / 16; n.key = p;
        n.gamma = copy(p + ".weight", C); n.beta = copy(p + ".bias", C);
        return n;
    }
    ResW res(const std::string& p, int ci, int co) {
        ResW r;
        r.c1 = conv(p + ".res_branch.0", co, ci, 3); r.n1 = norm(p + ".res_branch.1", co);
        r.c2 = conv(p + ".res_branch.3", co, co, 3); r.n2 = norm(p + ".res_branch.4", co);
        r.has_skip = ci != co;
        if (r.has_skip) { r.cs = conv(p + ".skip_con.0", co, ci, 1); r.ns = norm(p + ".skip_con.1", co); }
        return r;
    }
    PoolW pool(const std::string& p, int C) {
        PoolW w; w.c = conv(p + ".stride_conv.0", C, C, 2); w.n = norm(p + ".stride_conv.1", C); return w;
    }
    UpW up(const std::string& p, int ci, int co) {
        UpW u; u.Cin = ci; u.Cout = co; u.key = p + ".block.0";
        const float* src = get(p + ".block.0.weight", (int64_t)ci * co * 8);
        u.w = nm_ctx_weight_alloc(c, (size_t)ci * co * 8);
        if (src && u.w) { int r = nm_launch_transpose_convT_weight(src, ci, co, u.w, c->stream); if (r && !rc) rc = r; }
        if (src && c->training) {      // the adjoint of the transposed conv is the k2 s2 conv with (Cin, Cout, taps) read as OIDHW
            u.cd_pad = (ci + 31) & ~31;
            const size_t fl = nm_packed_weight_floats(2, co, u.cd_pad);
            u.wd = nm_ctx_weight_alloc(c, fl);
            int r = u.wd ? NM_OK : NM_ERR_HIP;
            if (!r && co % 16 == 0) {
                u.wd16 = nm_ctx_weight_alloc(c, fl);
                r = u.wd16 ? NM_OK : NM_ERR_HIP;
            }
            if (!r) pack(src, u.wd, u.wd16, ci, co, 2, co, u.cd_pad, co, 0);
            if (r && !rc) rc = r;
        }
        u.bias = copy(p + ".block.0.bias", co);
        u.n = norm(p + ".block.1", co);
        return u;
    }
    HourglassW hourglass(const std::string& p, int ci, int co) {
        HourglassW h;
        h.p1 = pool(p + ".encoder_pool1", ci); h.e1 = res(p + ".encoder_res1", ci, 32);
        h.p2 = pool(p + ".encoder_pool2", 32); h.e2 = res(p + ".encoder_res2", 32, 48);
        h.p3 = pool(p + ".encoder_pool3", 48); h.e3 = res(p + ".encoder_res3", 48, 72);
        h.d3 = res(p + ".decoder_res3", 72, 72); h.u3 = up(p + ".decoder_upsample3", 72, 48);
        h.d2 = res(p + ".decoder_res2", 48, 48); h.u2 = up(p + ".decoder_upsample2", 48, 32);
        h.d1 = res(p + ".decoder_res1", 32, 32); h.u1 = up(p + ".decoder_upsample1", 32, co);
        h.s1 = res(p + ".skip_res1", ci, co); h.s2 = res(p + ".skip_res2", 32, 32); h.s3 = res(p + ".skip_res3", 48, 48);
        return h;
    }
    FeatNetW featnet(const std::string& p, int cout) {
        FeatNetW f; const int c4 = cout / 4, c2 = cout / 2;
        f.c0 = conv(p + ".0.block.0", c4, 4, 5); f.n0 = norm(p + ".0.block.1", c4);
        first_layer_tables(p + ".0.block.0", f);
        f.p1 = pool(p + ".1", c4); f.r2 = res(p + ".2", c4, c2); f.p3 = pool(p + ".3", c2);
        f.hg = hourglass(p + ".4", c2, c2); f.r5 = res(p + ".5", c2, cout);
        return f;
    }
    // occupancy-channel taps packed for the taps-as-K kernel, and the constant field
    //   field = conv5(cat[0, x1, x2, x3]) + bias      (one frame, [G^3][Cout])
    // computed once per weight update with the generic conv on a zero-occupancy frame.
    void first_layer_tables(const std::string& p, FeatNetW& f) {
        const int G = c->cfg.grid_size, Cout = f.c0.Cout;
        const size_t G3 = (size_t)G * G * G;
        const float* src = get(p + ".weight", (int64_t)Cout * 4 * 125);
        if (!src || !f.c0.wp) return;
        f.occ_w = nm_ctx_weight_alloc(c, (size_t)256 * f.c0.Co_pad);          // fp32 pack + split-fp16 pack
        f.field = nm_ctx_weight_alloc(c, G3 * Cout);
        float* tmp = nm_ctx_weight_alloc(c, (size_t)Cout * 125 + G3 * 9);      // freed with the other weights at the next update
        if (!f.occ_w || !f.field || !tmp) { if (!rc) { nm_set_error("set_weights: hipMalloc failed"); rc = NM_ERR_HIP; } return; }
        float* zero = tmp + (size_t)Cout * 125; float* packed_in = zero + G3;
        int r = nm_launch_pack_occ_weight(src, Cout, tmp, f.occ_w, f.c0.Co_pad, c->stream);
        if (!r) r = nm_check_hip(hipMemsetAsync(zero, 0, G3 * sizeof(float), c->stream), "set_weights: memset");
        if (!r) r = nm_launch_pack_input(zero, 1, 1, G, 0, packed_in, c->stream);
        if (!r) {
            TensorRef t; t.p = packed_in; t.scale = nullptr; t.shift = nullptr; t.slope = 1.0f; t.N = 1; t.D = t.H = t.W = G; t.C = 8;
            ConvGeom g; g.ks = 5; g.stride = 1; g.pad = 2; g.OD = g.OH = g.OW = G; g.Cout = Cout; g.Co_pad = f.c0.Co_pad;
            // (the caller's bias tensor: the ctx-owned copy is filled by the batched copy at the end of set_weights)
            r = nm_launch_conv(t, f.c0.wp, get(p + ".bias", Cout), f.field, g, nullptr, c->stream, 4);
        }
        // the field's own GroupNorm partial sums per brick, by the first-layer kernel itself on the empty frame (bit-identical to what
        // it computes for an empty brick of any frame): what the sparse first layer reports for the bricks it skips
        f.field_part = nullptr;
        if (!r && !c->training && nm_conv_get_mode() == 1 && nm_ls().occ16 && nm_ls().sparse_first) {
            const int nblk = nm_occ_blocks_per_frame(G);
            f.field_part = nm_ctx_weight_alloc(c, (size_t)nblk * Cout * 2);
            float* scratch = nm_ctx_weight_alloc(c, G3 * Cout);           // (dense output of that one launch; reused at the next update)
            if (!f.field_part || !scratch) r = NM_ERR_HIP;
            else r = nm_launch_conv_k5occ(zero, 1, G, f.occ_w, f.field, scratch, Cout, f.c0.Co_pad, f.field_part, c->stream);
        }
        if (r && !rc) rc = r;
    }
    LinearW linear(const std::string& p, int out, int in) {
        LinearW l; l.in = in; l.out = out; l.w = copy(p + ".weight", (int64_t)out * in); l.b = copy(p + ".bias", out); return l;
    }
};

__global__ void pack_small_kernel(const float* a, int na, const float* b, int nb, float* out) {
    int i = threadIdx.x;
    if (i < na) out[i] = a[i];
    else if (i < na + nb) out[i] = b[i - na];
}

// ------------------------------------------------------------------------------------------
// graph helpers
// ------------------------------------------------------------------------------------------
struct Net {
    nm_ctx* c; hipStream_t s; Arena& ws; int rc = NM_OK;
    bool keep = false;                 // training forward: nothing is released, every activation stays for the backward pass
    explicit Net(nm_ctx* ctx) : c(ctx), s(ctx->stream), ws(ctx->ws) {}
    Net(nm_ctx* ctx, hipStream_t stream) : c(ctx), s(stream), ws(ctx->ws) {}
    void release(size_t m) { if (!keep) ws.release(m); }
    bool ok() const { return rc == NM_OK; }
    bool live() const { return rc == NM_OK && !ws.dry; }
    void run(int r) { if (r && !rc) rc = r; }
    float* alloc(size_t n) {
        float* p = ws.f(n);
        if (!p && !rc) { nm_set_error("workspace overflow (needed > %zu bytes)", ws.cap); rc = NM_ERR_STATE; }
        return p;
    }
    // 16-bit storage (conv mode 4): in the TRAINING forward a tensor the library allocates itself is bfloat16 when a frame of it has at
    // least store16_min voxels (32^3: everything above the hourglass); the inference forward keeps its fp32 workspace
    int h16(size_t voxels_per_frame) const { return keep && nm_ls().store16 && voxels_per_frame >= (size_t)nm_ls().store16_min ? 1 : 0; }
    float* alloc_e(size_t elems, int h) { return alloc(h ? (elems + 1) / 2 : elems); }      // `elems` elements of fp32 (h = 0) or bfloat16
};

TensorRef mk(const float* p, int N, int D, int H, int W, int C, const float* sc = nullptr, const float* sh = nullptr, float slope = 1.0f) {
    TensorRef t; t.p = p; t.scale = sc; t.shift = sh; t.slope = slope; t.N = N; t.D = D; t.H = H; t.W = W; t.C = C; return t;
}
TensorRef with_h(TensorRef t, int h) { t.h = h; return t; }
size_t vox(const TensorRef& t) { return (size_t)t.D * t.H * t.W; }

// conv (+ optional GroupNorm statistics): returns the lazy output
// (up2: `in` is stored at half resolution and its trilinear x2 upsampling is what gets convolved)
TensorRef conv_gn(Net& n, const TensorRef& in, const ConvW& w, const NormW* gn, int stride, int pad, float slope_after,
                  float* out_buf = nullptr, bool up2 = false, ConvRec* rec = nullptr) {
    ConvGeom g; g.ks = w.ks; g.stride = stride; g.pad = pad; g.up2 = up2 ? 1 : 0;
    const int us = up2 ? 2 : 1;
    g.OD = (us * in.D + 2 * pad - w.ks) / stride + 1; g.OH = (us * in.H + 2 * pad - w.ks) / stride + 1; g.OW = (us * in.W + 2 * pad - w.ks) / stride + 1;
    g.Cout = w.Cout; g.Co_pad = w.Co_pad; g.up2c = up2 ? w.wup : nullptr;
    const size_t ov = (size_t)g.OD * g.OH * g.OW;
    const int oh = out_buf ? 0 : n.h16(ov);                    // (a caller-named output buffer is always fp32)
    // One-product training modes, a fused-upsample layer with 32 outputs (the decoder's 64 -> 32 @64^3): the upsampled, activated input
    // is materialised ONCE here - the backward pass needs exactly that tensor for the weight gradient and used to rebuild it - and the
    // layer runs as a plain conv on conv_f16r (resident weights) instead of the composite-weight kernel's one-product instantiation
    // (conv_up2c<.., 3>: 2 x 2.4 ms per step at 0.44 of the time on the matrix pipe, profiles/r05_pmc_mfma_train_bf16.json).
    TensorRef src = in;
    const bool mat = up2 && rec && n.keep && !out_buf && nm_ls().single && nm_ls().up2_mat && nm_ls().f16r && w.wp16 && w.ks == 3 && stride == 1 && pad == 1 &&
                     w.Cout == 32 && (in.C == 32 || in.C == 64) && g.OD % 4 == 0 && g.OH % 8 == 0 && g.OW % 8 == 0 && g.OD >= 16;
    if (mat) {
        float* upb = n.alloc_e((size_t)in.N * ov * in.C, oh);
        if (n.live()) n.run(nm_launch_upsample2(in, upb, n.s, oh));
        src = with_h(mk(upb, in.N, g.OD, g.OH, g.OW, in.C), oh);
        g.up2 = 0; g.up2c = nullptr;
    }
    float* out = out_buf ? out_buf : n.alloc_e((size_t)in.N * ov * w.Cout, oh);
    float *part = nullptr, *scale = nullptr, *shift = nullptr;
    const int nblk = nm_conv_blocks_per_frame(g, in.C);
    double* chsum = nullptr;
    if (gn) {
        part = n.alloc((size_t)in.N * nblk * w.Cout * 2);
        scale = n.alloc((size_t)in.N * w.Cout); shift = n.alloc((size_t)in.N * w.Cout);
        if (rec && nm_gn_finalize_has_chsum(w.Cout, gn->groups)) chsum = reinterpret_cast<double*>(n.alloc((size_t)in.N * w.Cout * 4));
    }
    if (n.live()) {
        if (in.C != w.Cin_pad) { nm_set_error("conv_gn: input has %d channels, layer expects %d", in.C, w.Cin_pad); n.rc = NM_ERR_STATE; }
        else {
            n.run(nm_launch_conv(src, w.wp, w.bias, out, g, part, n.s, w.Cin, w.wp16, oh));
            if (gn) n.run(nm_launch_gn_finalize(part, in.N, nblk, w.Cout, gn->groups, (double)ov * (w.Cout / gn->groups),
                                                gn->gamma, gn->beta, 1e-5f, scale, shift, n.s, chsum));
            if (gn && nm_ls().gn_diag && !oh) n.run(nm_launch_gn_direct(out, in.N, (int)ov, w.Cout, gn->groups, gn->gamma, gn->beta, 1e-5f, scale, shift, n.s, chsum));
        }
    }
    TensorRef o = with_h(mk(out, in.N, g.OD, g.OH, g.OW, w.Cout, scale, shift, slope_after), oh);
    if (rec) { rec->w = &w; rec->gn = gn; rec->in = in; rec->out = o; rec->fpart = part; rec->nblk = nblk; rec->chsum = chsum; rec->stride = stride; rec->pad = pad; rec->up2 = up2; rec->upmat = mat ? src : TensorRef{}; }
    return o;
}

TensorRef add2(Net& n, const TensorRef& a, const TensorRef* b, float* out_buf = nullptr) {
    const int oh = out_buf ? 0 : n.h16(vox(a));
    float* out = out_buf ? out_buf : n.alloc_e((size_t)a.N * vox(a) * a.C, oh);
    if (n.live()) n.run(nm_launch_apply2(a, b, out, n.s, oh));
    return with_h(mk(out, a.N, a.D, a.H, a.W, a.C), oh);
}

// Res3DBlock (vox_modules.py:22-47): GN(conv3(lrelu(GN(conv3 x)))) + skip(x); the trailing
// F.leaky_relu(., True) is the identity.
// lazy_sum (inference encoder, the block in front of a pool conv): the sum is not materialised - the result is the un-materialised
// pair (TensorRef::p2), which the split-fp16 pool kernel evaluates while staging (one write and one read of the tensor less); the
// caller's mark / release then owns the two branches' buffers.
TensorRef res(Net& n, const TensorRef& x, const ResW& w, float* out_buf = nullptr, ResRec* rec = nullptr, bool lazy_sum = false) {
    if (lazy_sum && !rec && !out_buf) {
        TensorRef r1 = conv_gn(n, x, w.c1, &w.n1, 1, 1, LRELU);
        TensorRef r2 = conv_gn(n, r1, w.c2, &w.n2, 1, 1, 1.0f);
        TensorRef sk = w.has_skip ? conv_gn(n, x, w.cs, &w.ns, 1, 0, 1.0f) : x;
        TensorRef o = r2;
        o.p2 = sk.p; o.scale2 = sk.scale; o.shift2 = sk.shift; o.slope2 = sk.slope;
        return o;
    }
    const int oh = out_buf ? 0 : n.h16(vox(x));
    float* out = out_buf ? out_buf : n.alloc_e((size_t)x.N * vox(x) * w.c2.Cout, oh);
    const size_t m = n.ws.mark();
    TensorRef r1 = conv_gn(n, x, w.c1, &w.n1, 1, 1, LRELU, nullptr, false, rec ? &rec->c1 : nullptr);
    TensorRef r2 = conv_gn(n, r1, w.c2, &w.n2, 1, 1, 1.0f, nullptr, false, rec ? &rec->c2 : nullptr);
    TensorRef sk = w.has_skip ? conv_gn(n, x, w.cs, &w.ns, 1, 0, 1.0f, nullptr, false, rec ? &rec->cs : nullptr) : x;
    if (rec) rec->has_skip = w.has_skip;
    TensorRef o;
    {   // (add2 with the block's own output buffer and ITS element type)
        if (n.live()) n.run(nm_launch_apply2(r2, &sk, out, n.s, oh));
        o = with_h(mk(out, r2.N, r2.D, r2.H, r2.W, r2.C), oh);
    }
    n.release(m);
    return o;
}

TensorRef pool(Net& n, const TensorRef& x, const PoolW& w, ConvRec* rec = nullptr) {
    return conv_gn(n, x, w.c, &w.n, 2, 0, LRELU, nullptr, false, rec);
}

// Upsample3DBlock (vox_modules.py:63-75): ConvTranspose3d(k2,s2,output_padding) -> GN -> LeakyReLU (lazy)
TensorRef up(Net& n, const TensorRef& x, const UpW& w, int outpad, UpRec* rec = nullptr) {
    const int OD = 2 * x.D + outpad, OH = 2 * x.H + outpad, OW = 2 * x.W + outpad;
    const size_t ov = (size_t)OD * OH * OW;
    const int oh = n.h16(ov);
    if (oh && !n.rc) { nm_set_error("16-bit storage: a transposed-conv block of %zu voxels per frame (the hourglass must stay below the storage threshold)", ov); n.rc = NM_ERR_UNSUPPORTED; }
    float* out = n.alloc((size_t)x.N * ov * w.Cout);
    const int nblk = nm_stats_blocks_per_frame((int)ov);
    float* part = n.alloc((size_t)x.N * nblk * w.Cout * 2);
    float* scale = n.alloc((size_t)x.N * w.Cout); float* shift = n.alloc((size_t)x.N * w.Cout);
    double* chsum = (rec && nm_gn_finalize_has_chsum(w.Cout, w.n.groups)) ? reinterpret_cast<double*>(n.alloc((size_t)x.N * w.Cout * 4)) : nullptr;
    if (n.live()) {
        n.run(nm_launch_convT2(x, w.w, w.bias, out, w.Cout, OD, OH, OW, n.s));
        n.run(nm_launch_gn_partials(out, x.N, (int)ov, w.Cout, part, n.s));
        n.run(nm_launch_gn_finalize(part, x.N, nblk, w.Cout, w.n.groups, (double)ov * (w.Cout / w.n.groups), w.n.gamma,
                                    w.n.beta, 1e-5f, scale, shift, n.s, chsum));
        if (nm_ls().gn_diag) n.run(nm_launch_gn_direct(out, x.N, (int)ov, w.Cout, w.n.groups, w.n.gamma, w.n.beta, 1e-5f, scale, shift, n.s, chsum));
    }
    TensorRef o = mk(out, x.N, OD, OH, OW, w.Cout, scale, shift, LRELU);
    if (rec) { rec->w = &w; rec->in = x; rec->out = o; rec->fpart = part; rec->nblk = nblk; rec->chsum = chsum; }
    return o;
}

// HG (vox_modules.py:78-120)
// the two lowest levels as one launch (nm_hgcore.hip): parameters from the packed weights; false when a layer has no split-fp16 pack
static bool hg_core_params(const HourglassW& w, const TensorRef& a2, float* out, NmHgCoreParams& p) {
    auto cv = [](const ConvW& c, NmHgConv& o) { o.w16 = c.wp16; o.bias = c.bias; o.Cin = c.Cin; o.Cout = c.Cout; o.Co_pad = c.Co_pad; o.ks = c.ks; return c.wp16 != nullptr && c.Cin % 8 == 0; };
    auto nv = [](const NormW& g, NmHgNorm& o) { o.gamma = g.gamma; o.beta = g.beta; o.groups = g.groups; };
    auto rv = [&](const ResW& r, NmHgRes& o) {
        bool ok = cv(r.c1, o.c1) && cv(r.c2, o.c2);
        nv(r.n1, o.n1); nv(r.n2, o.n2);
        o.has_skip = r.has_skip ? 1 : 0;
        if (r.has_skip) { ok = ok && cv(r.cs, o.cs); nv(r.ns, o.ns); } else { o.cs = o.c1; o.ns = o.n1; }
        return ok;
    };
    bool ok = rv(w.e2, p.e2) && rv(w.s3, p.s3) && rv(w.e3, p.e3) && rv(w.d3, p.d3) && rv(w.d2, p.d2) && cv(w.p3.c, p.p3);
    nv(w.p3.n, p.np3); nv(w.u3.n, p.nu3);
    p.u3_w = w.u3.w; p.u3_bias = w.u3.bias; p.u3_Cin = w.u3.Cin; p.u3_Cout = w.u3.Cout;
    p.in = a2.p; p.in_scale = a2.scale; p.in_shift = a2.shift; p.in_slope = a2.slope; p.out = out;
    p.N = a2.N; p.D2 = a2.D; p.D3 = a2.D / 2; p.Cin0 = a2.C;
    int c2 = std::max(a2.C, std::max(w.e2.c2.Cout, std::max(w.s3.c2.Cout, std::max(w.u3.Cout, w.d2.c2.Cout))));
    int c3 = std::max(w.p3.c.Cout, std::max(w.e3.c2.Cout, w.d3.c2.Cout));
    p.pitch2 = (c2 + 15) & ~15; p.pitch3 = (c3 + 15) & ~15;
    ok = ok && a2.D == a2.H && a2.H == a2.W && a2.D >= 2 && a2.D <= 6 && p.u3_w && w.u3.Cout == w.s3.c2.Cout && a2.C == w.e2.c1.Cin &&
         w.e2.c2.Cout == w.s3.c1.Cin && w.e2.c2.Cout == w.p3.c.Cin && w.p3.c.Cout == w.e3.c1.Cin && w.d3.c2.Cout == w.u3.Cin &&
         w.u3.Cout == w.d2.c1.Cin && nm_hg_core_scratch_items(p) >= 0;
    p.scratch_items = ok ? nm_hg_core_scratch_items(p) : 0;
    return ok;
}

TensorRef hourglass(Net& n, const TensorRef& x0, const HourglassW& w, int Ng, HgRec* r = nullptr) {
    const int op3 = (Ng / 4) % 2, op2 = (Ng / 2) % 2, op1 = Ng % 2;
    TensorRef s1 = res(n, x0, w.s1, nullptr, r ? &r->s1 : nullptr);
    TensorRef x = res(n, pool(n, x0, w.p1, r ? &r->p1 : nullptr), w.e1, nullptr, r ? &r->e1 : nullptr);
    TensorRef s2 = res(n, x, w.s2, nullptr, r ? &r->s2 : nullptr);
    TensorRef a2 = pool(n, x, w.p2, r ? &r->p2 : nullptr);
    NmHgCoreParams hp;
    TensorRef u;
    if (!r && nm_conv_get_mode() == 1 && nm_ls().hg_core && hg_core_params(w, a2, nullptr, hp)) {
        // inference: encoder_res2 ... decoder_res2 (13 convs, 14 GroupNorms, the transposed conv and the adds) in one launch
        float* d2out = n.alloc((size_t)a2.N * vox(a2) * w.d2.c2.Cout);
        hp.out = d2out;
        if (n.live()) n.run(nm_launch_hg_core(hp, n.s));
        x = mk(d2out, a2.N, a2.D, a2.H, a2.W, w.d2.c2.Cout);
    } else {
        x = res(n, a2, w.e2, nullptr, r ? &r->e2 : nullptr);
        TensorRef s3 = res(n, x, w.s3, nullptr, r ? &r->s3 : nullptr);
        x = res(n, pool(n, x, w.p3, r ? &r->p3 : nullptr), w.e3, nullptr, r ? &r->e3 : nullptr);
        x = res(n, x, w.d3, nullptr, r ? &r->d3 : nullptr);
        u = up(n, x, w.u3, op3, r ? &r->u3 : nullptr); x = add2(n, u, &s3);
        x = res(n, x, w.d2, nullptr, r ? &r->d2 : nullptr);
    }
    u = up(n, x, w.u2, op2, r ? &r->u2 : nullptr); x = add2(n, u, &s2);
    x = res(n, x, w.d1, nullptr, r ? &r->d1 : nullptr);
    u = up(n, x, w.u1, op1, r ? &r->u1 : nullptr); x = add2(n, u, &s1);
    (void)op3;
    return x;
}

// Basic3DBlock(k5) on cat[occ, x1, x2, x3] (kypt_detector.py:265, kypt_detector_utils.py:4-26): only the occupancy
// channel is convolved per frame, the coordinate channels' contribution (+ bias) is the weight-only `field`.
TensorRef first_layer(Net& n, const float* occ, int N, int G, const FeatNetW& w, FeatRec* rec = nullptr) {
    const int Cout = w.c0.Cout;
    const size_t G3 = (size_t)G * G * G;
    const int oh = rec ? n.h16(G3) : 0;
    float* out = n.alloc_e((size_t)N * G3 * Cout, oh);
    const int nblk = nm_occ_blocks_per_frame(G);
    float* part = n.alloc((size_t)N * nblk * Cout * 2);
    // inference: bricks with an empty occupancy halo (most of the grid around one figure) are neither computed nor written; the pool
    // conv - this tensor's only consumer - reads the constant field there (TensorRef::alt / brickmap).  Training keeps the dense
    // tensor (the backward pass reads it).
    const bool sparse = !rec && w.field_part && nm_conv_get_mode() == 1 && nm_ls().occ16 && nm_ls().sparse_first &&
                        nm_conv_pool16_eligible(Cout, G / 2, G / 2, G / 2, w.p1.c.wp16 != nullptr);
    unsigned char* bmap = sparse ? reinterpret_cast<unsigned char*>(n.alloc(((size_t)N * nblk + 3) / 4)) : nullptr;
    unsigned char* bflags = sparse ? reinterpret_cast<unsigned char*>(n.alloc(((size_t)N * nblk + 3) / 4)) : nullptr;
    float* scale = n.alloc((size_t)N * Cout); float* shift = n.alloc((size_t)N * Cout);
    double* chsum = (rec && nm_gn_finalize_has_chsum(Cout, w.n0.groups)) ? reinterpret_cast<double*>(n.alloc((size_t)N * Cout * 4)) : nullptr;
    if (n.live()) {
        n.run(nm_launch_conv_k5occ(occ, N, G, w.occ_w, w.field, out, Cout, w.c0.Co_pad, part, n.s, bmap, sparse ? w.field_part : nullptr, bflags, oh));
        n.run(nm_launch_gn_finalize(part, N, nblk, Cout, w.n0.groups, (double)G3 * (Cout / w.n0.groups), w.n0.gamma, w.n0.beta,
                                    1e-5f, scale, shift, n.s, chsum));
        if (nm_ls().gn_diag && !sparse && !oh) n.run(nm_launch_gn_direct(out, N, (int)G3, Cout, w.n0.groups, w.n0.gamma, w.n0.beta, 1e-5f, scale, shift, n.s, chsum));
    }
    TensorRef o = with_h(mk(out, N, G, G, G, Cout, scale, shift, LRELU), oh);
    if (sparse) { o.alt = w.field; o.brickmap = bmap; }
    if (rec) { rec->w = &w; rec->occ = occ; rec->N = N; rec->G = G; rec->first = o; rec->fpart0 = part; rec->nblk0 = nblk; rec->chsum0 = chsum; }
    return o;
}

// _build_feature_net (kypt_detector.py:264-272); `occ` is the occupancy [N][G][G][G]
void feature_net(Net& n, const float* occ, int N, int G, const FeatNetW& w, int g, float* out_buf, FeatRec* r = nullptr) {
    const size_t m = n.ws.mark();
    TensorRef x = first_layer(n, occ, N, G, w, r);
    x = pool(n, x, w.p1, r ? &r->p1 : nullptr);
    // (inference: r2's residual sum is evaluated by the pool conv that consumes it)
    const bool lazy = !r && nm_ls().lazy_res && nm_conv_pool16_eligible(w.r2.c2.Cout, x.D / 2, x.H / 2, x.W / 2, w.p3.c.wp16 != nullptr) &&
                      x.D % 2 == 0 && x.H % 2 == 0 && x.W % 2 == 0;
    x = res(n, x, w.r2, nullptr, r ? &r->r2 : nullptr, lazy);
    x = pool(n, x, w.p3, r ? &r->p3 : nullptr);
    x = hourglass(n, x, w.hg, g, r ? &r->hg : nullptr);
    res(n, x, w.r5, out_buf, r ? &r->r5 : nullptr);
    n.release(m);
}

// KyptToVoxNet for a batch of frames (kypt_detector.py:388-460).
//   keypoints [F][K][4]; first-frame feature in channels-last with a frame stride;
//   first_frames: occupancy of each clip's first frame (frame stride ff_stride);
//   target/tail_part/chamfer only for the training-style forward.
void decode_frames(Net& n, const float* keypoints, const float* feat_cl, int feat_frame_stride, const float* first_frames,
                   int ff_stride, int B, int T, const float* target, bool chamfer, float* recon, float* tail_part,
                   TrainTape* tape = nullptr, bool learn_sigma = false) {
    nm_ctx* c = n.c;
    const DetectorW& d = c->det;
    const int K = c->cfg.nkeypoints, G = c->cfg.grid_size, g = G / 4, F = B * T;
    const int Cc = d.adjust.Cin_pad;
    const size_t g3 = (size_t)g * g * g, G3 = (size_t)G * G * G;
    const double width_d = 2.0 * std::pow((double)c->cfg.gaussian_sigma / (double)g, 2.0);
    const size_t m0 = n.ws.mark();
    float* table = n.alloc((size_t)F * K * 3 * g);
    // fixed_sigma = 0: the detector's maps take sigmoid(sigmas) * 2 gaussian_sigma per keypoint (kypt_detector.py:303-306); decode_from_dyna
    // keeps the fixed list (:226), so only the detector's own call passes learn_sigma
    float* widthk = learn_sigma ? n.alloc(64) : nullptr;
    if (n.live()) {
        if (widthk) n.run(nm_launch_gauss_width(d.sigma_param, K, 2.0f * c->cfg.gaussian_sigma, g, widthk, n.s));
        n.run(nm_launch_gauss_table(keypoints, F * K, g, (float)width_d, table, n.s, widthk, K));
    }
    const int tb = nm_tail_blocks(G);
    // whole clips per pass so that frame 0 of every clip in the pass is addressable
    const int clips_per_pass = tape ? B : ((int)(FRAME_CHUNK / (size_t)T) > 0 ? (int)(FRAME_CHUNK / (size_t)T) : 1);
    if (tape) tape->table = table;
    for (int b0 = 0; b0 < B; b0 += clips_per_pass) {
        const int nb = (B - b0) < clips_per_pass ? (B - b0) : clips_per_pass;
        const int f0 = b0 * T, nf = nb * T;
        const size_t m1 = n.ws.mark();
        TensorRef x;
        if (!tape && d.adjust_wg && d.adjust_rest.wp && c->gauss_cat == 0) {     // (the split needs the K maps apart: 'max' / 'sum' take the materialised form)
            // inference: per-clip conv over [first_feature, gauss_0, coords] (+ bias), then the per-frame gaussian part on top of it
            const int Cr = d.adjust_rest.Cin_pad;
            float* rest = n.alloc((size_t)nb * g3 * Cr);
            if (n.live())
                n.run(nm_launch_combined_rest(table + (size_t)f0 * K * 3 * g, keypoints + (size_t)f0 * K * 4,
                                              feat_cl + (size_t)b0 * feat_frame_stride * g3 * FEAT, feat_frame_stride, nb, T, K, FEAT, g, Cr, rest, n.s));
            ConvW wr = d.adjust_rest; wr.bias = d.adjust.bias;
            TensorRef base = conv_gn(n, mk(rest, nb, g, g, g, Cr), wr, nullptr, 1, 0, 1.0f);
            float* adj = n.alloc((size_t)nf * g3 * FEAT);
            if (n.live())
                n.run(nm_launch_adjust_gauss(table + (size_t)f0 * K * 3 * g, keypoints + (size_t)f0 * K * 4, base.p, d.adjust_wg, nf, T, K, g, FEAT, adj, n.s));
            x = mk(adj, nf, g, g, g, FEAT, nullptr, nullptr, LRELU);
        } else {
            float* comb = n.alloc((size_t)nf * g3 * Cc);
            if (n.live())
                n.run(nm_launch_combined(table + (size_t)f0 * K * 3 * g, keypoints + (size_t)f0 * K * 4,
                                         feat_cl + (size_t)b0 * feat_frame_stride * g3 * FEAT, feat_frame_stride, nf, T, K, FEAT, g,
                                         Cc, comb, n.s, c->gauss_cat));
            x = mk(comb, nf, g, g, g, Cc);
            x = conv_gn(n, x, d.adjust, nullptr, 1, 0, LRELU, nullptr, false, tape ? &tape->adjust : nullptr);
        }
        x = conv_gn(n, x, d.d1, &d.dn2, 1, 1, LRELU, nullptr, true, tape ? &tape->d1 : nullptr);    // Upsample(x2, trilinear) fused into the staging
        x = conv_gn(n, x, d.d4, &d.dn5, 1, 1, LRELU, nullptr, false, tape ? &tape->d4 : nullptr);
        x = conv_gn(n, x, d.d8, &d.dn9, 1, 1, LRELU, nullptr, true, tape ? &tape->d8 : nullptr);    // second Upsample(x2) likewise
        x = conv_gn(n, x, d.d11, &d.dn12, 1, 1, LRELU, nullptr, false, tape ? &tape->d11 : nullptr);
        if (n.live())
            n.run(nm_launch_decoder_tail(x, d.d14, first_frames + (size_t)b0 * ff_stride * G3, ff_stride, T,
                                         target ? target + (size_t)f0 * G3 : nullptr,
                                         (target && chamfer) ? keypoints + (size_t)f0 * K * 4 : nullptr, K, G,
                                         recon + (size_t)f0 * G3, tail_part ? tail_part + (size_t)f0 * tb * 3 : nullptr, n.s));
        n.release(m1);
    }
    n.release(m0);
}

// `after_keypoints` (optional) is called once the keypoints kernel has been enqueued, in the live pass only: the
// fused forward uses it to start the VRNN on the side stream while the decoder runs.
int detector_graph(nm_ctx* c, const float* vox_in, int B, int T, int affinity_on, float* keypoints, float* heatmaps,
                   float* first_feature, float* recon, float* affinity, float* losses,
                   const std::function<int()>* after_keypoints = nullptr, TrainTape* tape = nullptr) {
    Net n(c);
    n.keep = tape != nullptr;
    const DetectorW& d = c->det;
    const int K = c->cfg.nkeypoints, G = c->cfg.grid_size, g = G / 4, F = B * T, N = c->cfg.nneighbor;
    const size_t g3 = (size_t)g * g * g, G3 = (size_t)G * G * G;
    const int Kc = d.head.Cout;                 // head / clip_head channels per voxel: K rounded up to 8 (padded channels are zeros)
    n.ws.release(0);
    float* feat = n.alloc((size_t)F * g3 * FEAT);
    float* clip_head = n.alloc((size_t)B * g3 * Kc);
    float* heat_part = n.alloc((size_t)F * K * g * (2 * g + 2));
    float* heat_mean = n.alloc((size_t)F * K);
    float* clip_part = n.alloc((size_t)B * 5);
    const int tb = nm_tail_blocks(G);
    float* tail_part = n.alloc((size_t)F * tb * 3);
    float* frame_sums = n.alloc((size_t)F * 3);
    float* aff = affinity_on ? (affinity ? affinity : n.alloc((size_t)N * K * K)) : nullptr;

    // spatio-temporal heat-map from the clip mean, once per clip (kypt_detector.py:311-316).  Only B frames of small, latency-bound
    // launches: issued on the side stream so that it runs beside the per-frame encoder.  Its scratch - every byte it ever touches, its
    // released temporaries included - stays out of reach of the main stream until the call ends, because the two streams run
    // concurrently.
    // ORDER OF ENQUEUE: the block is ~90 launches, about a millisecond of host time, and a forward call starts with the queues empty
    // (the previous call's results were read): enqueued first, it left the main stream without work for the first 1.7-1.8 ms of every
    // call while the chip ran 4-frame kernels (gpurun_out timelines, rounds 3-4).  So the per-frame encoder's first chunk(s) go first
    // and the clip block is enqueued behind them; its scratch region is reserved BEFORE (sized by a nested measuring pass of the same
    // code) so that nothing the main stream allocates in between can alias it.
    auto clip_block = [&]() {
        Net n2(c, c->stream2);
        n2.keep = n.keep;
        float* in = n2.alloc((size_t)B * G3);
        float* fclip = n2.alloc((size_t)B * g3 * 2 * FEAT);
        if (n2.live()) n2.run(nm_launch_mean_t(vox_in, B, T, G3, in, n2.s));
        feature_net(n2, in, B, G, d.clip, g, fclip, tape ? &tape->clip : nullptr);
        conv_gn(n2, mk(fclip, B, g, g, g, 2 * FEAT), d.clip_head, nullptr, 1, 0, 1.0f, clip_head, false, tape ? &tape->clip_head : nullptr);
        if (tape) tape->clip_in = in;
        if (n2.live()) n2.run(nm_check_hip(hipEventRecord(c->ev_clip, c->stream2), "clip event"));
        n.run(n2.rc);
    };
    if (n.live()) {
        n.run(nm_check_hip(hipEventRecord(c->ev_fork, n.s), "fork event"));
        n.run(nm_check_hip(hipStreamWaitEvent(c->stream2, c->ev_fork, 0), "side stream wait"));
    }
    const size_t clip_r0 = n.ws.top;
    size_t clip_need = 0;
    {   // measuring pass: the block's high-water mark above clip_r0
        const size_t sv_peak = n.ws.peak; const bool sv_dry = n.ws.dry, sv_of = n.ws.overflow;
        n.ws.dry = true; n.ws.peak = n.ws.top;
        clip_block();
        clip_need = ((n.ws.peak - clip_r0) + 255) & ~(size_t)255;
        n.ws.top = clip_r0; n.ws.peak = sv_peak; n.ws.dry = sv_dry; n.ws.overflow = sv_of;
    }
    auto run_clip_block = [&]() {
        const size_t top_now = n.ws.top;
        n.ws.top = clip_r0;
        clip_block();
        if (n.ws.top > clip_r0 + clip_need && !n.rc) { nm_set_error("detector_forward: the clip net outgrew its reserved scratch"); n.rc = NM_ERR_STATE; }
        n.ws.top = top_now;
    };
    (void)n.ws.alloc_bytes(clip_need);                          // reserve [clip_r0, clip_r0 + clip_need)
    n.ws.top = clip_r0 + clip_need;
    const int clip_after = nm_ls().clip_late ? (tape ? 1 : 2) : 0;     // per-frame chunks enqueued before the clip block
    if (clip_after == 0) run_clip_block();
    const size_t chunk = tape ? (size_t)F : FRAME_CHUNK;       // training keeps every activation: one pass over all frames
    int chunks_done = 0;
    for (size_t f0 = 0; f0 < (size_t)F; f0 += chunk) {   // per-frame encoder (kypt_detector.py:330-336)
        const int nf = (int)(((size_t)F - f0) < chunk ? ((size_t)F - f0) : chunk);
        const size_t m = n.ws.mark();
        feature_net(n, vox_in + f0 * G3, nf, G, d.frame, g, feat + f0 * g3 * FEAT, tape ? &tape->frame : nullptr);
        n.release(m);
        if (++chunks_done == clip_after) run_clip_block();
    }
    if (chunks_done < clip_after) run_clip_block();
    {   // heads -> heat-maps -> keypoints (kypt_detector.py:336-347)
        const size_t m = n.ws.mark();
        float* head = n.alloc((size_t)F * g3 * Kc);
        conv_gn(n, mk(feat, F, g, g, g, FEAT), d.head, nullptr, 1, 0, 1.0f, head, false, tape ? &tape->head : nullptr);
        if (n.live()) {
            n.run(nm_check_hip(hipStreamWaitEvent(n.s, c->ev_clip, 0), "join clip net"));
            n.run(nm_launch_heatmap(head, clip_head, d.prop, F, T, K, Kc, g, heatmaps, heat_part, n.s));
            n.run(nm_launch_keypoints(heat_part, F, K, g, keypoints, heat_mean, n.s));
            if (after_keypoints && n.ok()) n.run((*after_keypoints)());
        }
        if (tape) { tape->head_out = head; }
        n.release(m);
    }
    if (n.live()) {   // first_feature output: frame 0 of every clip, NCDHW
        TensorRef ff = mk(feat, B, g, g, g, FEAT);
        n.run(nm_launch_cl_to_ncdhw_strided(ff, T, first_feature, n.s));
        if (affinity_on) n.run(nm_launch_affinity(d.affinity_params, N, K, aff, n.s, c->affinity_ver));
    }
    // (recon == nullptr: the keypoints-only pass of nm_detector_keypoints - no voxel decoder, no losses)
    if (recon) decode_frames(n, keypoints, feat, T, vox_in, T, B, T, vox_in, c->cfg.vol_fit_chamfer == 1, recon, tail_part, tape, c->learn_sigma != 0);
    if (tape) {
        tape->B = B; tape->T = T; tape->affinity_on = affinity_on; tape->vox = vox_in; tape->feat = feat; tape->clip_head_out = clip_head;
        tape->heat_part = heat_part; tape->heat_mean = heat_mean; tape->tail_part = tail_part; tape->aff = aff;
        tape->keypoints = keypoints; tape->recon = recon;
    }
    float* vol_fs = nullptr; float* vol_ws = nullptr;               // vol_fit_type 'gaussian' (cfg.vol_fit_chamfer == 2): its own per-frame sums
    if (recon && losses && c->cfg.vol_fit_chamfer == 2) { vol_fs = n.alloc((size_t)F * 2); vol_ws = n.alloc(nm_volfit_gauss_ws_floats(F, G)); }
    if (n.live() && recon && losses) {
        n.run(nm_launch_clip_loss(keypoints, aff, B, T, K, N, c->cfg.sep_sigma, clip_part, n.s));
        if (vol_fs) n.run(nm_launch_volfit_gauss(vox_in, keypoints, B, T, K, G, c->cfg.gaussian_sigma, vol_ws, vol_fs, n.s));
        n.run(nm_launch_loss_finalize(tail_part, tb, B, T, K, N, G, heat_mean, clip_part, aff, c->cfg.vol_fit_chamfer,
                                      c->cfg.use_graph_traj, frame_sums, losses, n.s, vol_fs));
    }
    return n.rc;
}

int decode_graph(nm_ctx* c, const float* keypoints, const float* first_feature, const float* first_frame, int B, int Tg,
                 float* gen) {
    Net n(c);
    const int G = c->cfg.grid_size, g = G / 4;
    const size_t g3 = (size_t)g * g * g;
    n.ws.release(0);
    float* feat_cl = n.alloc((size_t)B * g3 * FEAT);
    if (n.live()) n.run(nm_launch_ncdhw_to_cl(first_feature, B, (int)g3, FEAT, feat_cl, n.s));
    decode_frames(n, keypoints, feat_cl, 1, first_frame, 1, B, Tg, nullptr, false, gen, nullptr);
    return n.rc;
}

// ------------------------------------------------------------------------------------------
// backward graph (detector-mode training, train.py:388-404): the reverse walk over the tape
// ------------------------------------------------------------------------------------------
struct Bwd {
    nm_ctx* c; hipStream_t s; Arena& ws; int rc = NM_OK;
    const std::map<std::string, std::pair<float*, int64_t>>* grads;     // nullptr in the sizing pass
    float* zb = nullptr;                                                  // 512 zeros: bias of the data-gradient convolutions
    unsigned* amax_pool = nullptr; int amax_next = 0, amax_cap = 0;       // zeroed words behind zb: one max|dy| cell per scaled layer
    // per-layer (dgamma_n, dbeta_n, dbias_n) rows of the GroupNorm backward, kept until flush_sums() turns them into the three parameter
    // gradients of every layer in one launch (they are outputs only: nothing in the walk waits for them)
    float* dgn_pool = nullptr; size_t dgn_cap = 0, dgn_used = 0;
    std::vector<NmSum3Job> sums;
    float* dgn_rows(size_t n) { if (!dgn_pool || dgn_used + n > dgn_cap) return nullptr; float* p = dgn_pool + dgn_used; dgn_used += n; return p; }
    void flush_sums() {
        if (live() && !sums.empty()) run(nm_launch_sum_frames3_multi(sums.data(), (int)sums.size(), s));
        sums.clear();
    }
    // weight gradients on the context's third stream (async_w; main walk only).  The walk's dependent chain is GroupNorm backward -> data
    // gradient -> the previous layer's GroupNorm backward ...; a layer's weight gradient hangs off it, and it is matrix-core work while
    // the GroupNorm passes are HBM work: issued on a stream of its own it runs under the NEXT layers' memory passes instead of between
    // them.  What it reads must then outlive the main stream's arena frames: dY comes from a ring of three ctx-owned buffers (a slot is
    // handed out again only behind the event of the weight gradient that read it), the operand-scale vectors from a pool, and the
    // re-materialised upsample / slot workspace is scratch that only stream3 touches (in order).
    bool async_w = false;
    float* ring[3] = {nullptr, nullptr, nullptr}; size_t slot_floats = 0; int ring_i = 0; bool slot_busy[3] = {false, false, false};
    int last_slot = -1;                                   // ring slot of the dY the last norm_bwd returned (-1: arena)
    float* sscratch = nullptr; size_t sscratch_floats = 0;
    float* scpool = nullptr; size_t sc_floats = 0, sc_used = 0;
    size_t need_slot = 0, need_scratch = 0, need_sc = 0;  // sizing pass: what the walk asked for
    static size_t r64(size_t n) { return (n + 63) & ~(size_t)63; }
    float* fake() const { return reinterpret_cast<float*>((uintptr_t)256); }
    // (element counts -> floats for a tensor of the given storage type)
    static size_t fl(size_t elems, int h) { return h ? (elems + 1) / 2 : elems; }
    float* dy_alloc(size_t n) {
        last_slot = -1;
        if (!async_w) return alloc(n);
        if (ws.dry) { need_slot = std::max(need_slot, r64(n)); last_slot = 0; return fake(); }
        if (n > slot_floats) return alloc(n);
        const int k = ring_i; ring_i = (ring_i + 1) % 3;
        if (slot_busy[k]) { run(nm_check_hip(hipStreamWaitEvent(s, c->ev_w[k], 0), "backward: dY ring slot")); slot_busy[k] = false; }
        last_slot = k;
        return ring[k];
    }
    float* sc_alloc(size_t n) {
        n = r64(n);
        if (ws.dry) { need_sc += n; return fake(); }
        if (sc_used + n > sc_floats) return nullptr;
        float* p = scpool + sc_used; sc_used += n; return p;
    }
    Bwd(nm_ctx* ctx, const std::map<std::string, std::pair<float*, int64_t>>* g) : c(ctx), s(ctx->stream), ws(ctx->ws), grads(g) {}
    Bwd(nm_ctx* ctx, const std::map<std::string, std::pair<float*, int64_t>>* g, hipStream_t stream) : c(ctx), s(stream), ws(ctx->ws), grads(g) {}
    bool live() const { return rc == NM_OK && !ws.dry; }
    void run(int r) { if (r && !rc) rc = r; }
    float* alloc(size_t n) {
        float* p = ws.f(n);
        if (!p && !rc) { nm_set_error("backward workspace overflow (needed > %zu bytes)", ws.cap); rc = NM_ERR_STATE; }
        return p;
    }
    float* grad(const std::string& key, int64_t numel) {
        if (ws.dry || !grads) return reinterpret_cast<float*>((uintptr_t)256);
        auto it = grads->find(key);
        if (it == grads->end()) { if (!rc) { nm_set_error("detector_backward: no gradient buffer for '%s'", key.c_str()); rc = NM_ERR_ARG; } return nullptr; }
        if (it->second.second != numel) {
            if (!rc) { nm_set_error("detector_backward: '%s' has %lld elements, expected %lld", key.c_str(), (long long)it->second.second, (long long)numel); rc = NM_ERR_ARG; }
            return nullptr;
        }
        return it->second.first;
    }
};

size_t numel_of(const TensorRef& t) { return (size_t)t.N * t.D * t.H * t.W * t.C; }
// a tensor of `like`'s shape AND storage type without a pending affine (gradients are stored like the tensor they belong to)
TensorRef plain(const float* p, const TensorRef& like) { return with_h(mk(p, like.N, like.D, like.H, like.W, like.C), like.h); }

// GroupNorm(+LeakyReLU) backward of a lazy tensor: returns dy (gradient of the raw conv output) and writes the gradients of
// gamma / beta and of the bias of the producing conv
// (amax, optional: device word that ends up holding max |dy|, for the operand scaling of the data-gradient conv)
// dv / wv (optional): dA is the outer product dv[frame][voxel] * wv[channel] (the decoder's last conv layer) and is never materialised
// c_real (< out.C): the layer's output channels are padded inside the library; the bias gradient has c_real entries (layers without GroupNorm)
const float* norm_bwd(Bwd& b, const TensorRef& out, const NormW* gn, const float* fpart, int nblk_f, const double* chsum,
                      const std::string& bias_key, const float* dA, unsigned* amax = nullptr, const float* dA_mul = nullptr,
                      const float* dv = nullptr, const float* wv = nullptr, int c_real = 0) {
    const int N = out.N, C = out.C, V = out.D * out.H * out.W;
    const int nbb = nm_gnb_blocks_per_frame(V);
    float* dy = nullptr;
    b.last_slot = -1;
    if (gn) {
        dy = b.dy_alloc(Bwd::fl(numel_of(out), out.h));
        const size_t m = b.ws.mark();
        float* bpart = b.alloc((size_t)N * nbb * C * 2);
        float* coef = b.alloc((size_t)N * C * 4);
        float* dgn = b.dgn_rows((size_t)N * C * 4);
        const bool deferred = dgn != nullptr;
        if (!deferred) dgn = b.alloc((size_t)N * C * 4);
        float* gg = b.grad(gn->key + ".weight", C); float* gb = b.grad(gn->key + ".bias", C); float* gbias = b.grad(bias_key, C);
        if (b.live()) {
            b.run(nm_launch_gnb_partials(dA, out, bpart, b.s, dA_mul, dv, wv));
            b.run(nm_launch_gnb_finalize(bpart, nbb, fpart, nblk_f, N, C, gn->groups, V, gn->gamma, 1e-5f, coef, dgn, b.s, chsum));
            if (deferred) b.sums.push_back(NmSum3Job{dgn, gg, gb, gbias, N, C});
            else b.run(nm_launch_sum_frames3(dgn, N, C, gg, gb, gbias, b.s));
            b.run(nm_launch_gnb_apply(dA, out, coef, dy, b.s, amax, dA_mul, dv, wv));
        }
        b.ws.release(m);
        return dy;
    }
    if (dv) { nm_set_error("detector_backward: outer-product gradient into a layer without GroupNorm"); b.rc = NM_ERR_STATE; return nullptr; }
    const float* res = dA;
    if (out.slope != 1.0f || dA_mul) {
        // (from the dY ring when the weight gradients run on their own stream: the layer's weight gradient - the decoder's 179 -> 128
        //  1x1 conv is 1.6 ms of fp32 matrix work - then leaves the main stream's chain like those of the layers with a GroupNorm)
        dy = b.dy_alloc(Bwd::fl(numel_of(out), out.h));
        if (b.live()) b.run(nm_launch_gnb_apply(dA, out, nullptr, dy, b.s, amax, dA_mul));
        res = dy;
    } else if (amax && b.live()) b.run(nm_launch_absmax(dA, numel_of(out), amax, b.s, nullptr, out.h));
    const size_t m = b.ws.mark();
    float* bpart = b.alloc((size_t)N * nbb * C * 2);
    const bool padded = c_real > 0 && c_real < C;
    float* gbias = b.grad(bias_key, padded ? c_real : C);
    float* gb_all = padded ? b.alloc(C) : gbias;
    if (b.live()) {
        b.run(nm_launch_gnb_partials(res, plain(res, out), bpart, b.s));
        b.run(nm_launch_sum_partials(bpart, N * nbb, C, gb_all, b.s));
        if (padded) b.run(nm_check_hip(hipMemcpyAsync(gbias, gb_all, (size_t)c_real * sizeof(float), hipMemcpyDeviceToDevice, b.s), "backward: copy"));
    }
    b.ws.release(m);
    return res;
}

// conv (+GN +LeakyReLU) backward.  dA: gradient w.r.t. the activated output.  Returns the gradient w.r.t. the activated input
// ([N][D][H][W][csel], csel = Cin rounded down to 8) or nullptr when need_din is false.
// Operand scaling of a data-gradient conv that runs on the split-fp16 kernels (nm_grad.h): dy is read as dy * 2^k.
struct DyScale {
    unsigned* amax = nullptr; float* scale = nullptr; float* shift = nullptr; float* sc2 = nullptr;
    bool pool = false;             // vectors from the backward's scale pool (they outlive the caller's arena frame)
    void prepare(Bwd& b, bool on, int count, float* sc2_keep = nullptr) {
        if (!on) return;
        if (b.async_w) {
            float* blk = b.sc_alloc((size_t)2 * Bwd::r64(count) + 64);
            if (blk) {
                const bool pooled = b.amax_pool && b.amax_next < b.amax_cap;
                amax = pooled ? b.amax_pool + b.amax_next++ : reinterpret_cast<unsigned*>(b.alloc(64));
                scale = blk; shift = blk + Bwd::r64(count); sc2 = shift + Bwd::r64(count); pool = true;
                if (!pooled && b.live()) b.run(nm_check_hip(hipMemsetAsync(amax, 0, sizeof(unsigned), b.s), "backward: memset"));
                return;
            }
        }
        // (the max cell from the pool zeroed once per backward; the shift vector is zeroed by make_scale: two memsets per layer
        //  were 126 dependent stream operations per step)
        const bool pooled = b.amax_pool && b.amax_next < b.amax_cap;
        amax = pooled ? b.amax_pool + b.amax_next++ : reinterpret_cast<unsigned*>(b.alloc(64));
        scale = b.alloc(count); shift = b.alloc(count); sc2 = sc2_keep ? sc2_keep : b.alloc(64);
        if (!pooled && b.live()) b.run(nm_check_hip(hipMemsetAsync(amax, 0, sizeof(unsigned), b.s), "backward: memset"));
    }
    TensorRef apply(Bwd& b, const TensorRef& dyT) {
        if (!amax) return dyT;
        if (b.live()) b.run(nm_launch_make_scale(amax, dyT.N * dyT.C, scale, sc2, b.s, shift));
        TensorRef t = dyT; t.scale = scale; t.shift = shift;
        return t;
    }
    const float* inv() const { return amax ? sc2 + 1 : nullptr; }
};

// dA_mul: device scalar the incoming dA still has to be multiplied by (the producing conv_bwd left its power-of-two scale in).
// out_mul (optional): the caller feeds the result straight into the next conv_bwd as dA + dA_mul; then the un-scaling pass over
// the returned tensor is skipped and *out_mul is the scalar to hand on (nullptr when the result is already unscaled).
float* conv_bwd(Bwd& b, const ConvRec& r, const float* dA, bool need_din, const float* dA_mul = nullptr, const float** out_mul = nullptr,
                const float* dA_dv = nullptr, const float* dA_wv = nullptr) {
    const ConvW& w = *r.w;
    const TensorRef& in = r.in;
    // storage types: the input's gradient like the input; the fine-grid tensors of a fused-upsample layer like its output
    const int h_in = in.h, h_fine = r.out.h;
    float* din = need_din ? b.alloc(Bwd::fl((size_t)in.N * in.D * in.H * in.W * w.csel, h_in)) : nullptr;
    float* sc2_keep = out_mul ? b.alloc(64) : nullptr;           // outlives this call (allocated below the mark)
    if (out_mul) *out_mul = nullptr;
    const size_t m = b.ws.mark();
    const bool split = nm_conv_get_mode() != 0;          // split-fp16 kernels: dy is read pre-scaled by a power of two
    DyScale ds;
    // (the k2 s2 pool convs: their weight gradient runs on the f16 matrix cores as well, wgrad16k2_kernel)
    ds.prepare(b, split && ((r.stride == 1 && (w.ks == 3 || (need_din && w.wd16))) || (r.stride == 2 && w.ks == 2 && nm_ls().wgrad_k2f16)), r.out.N * r.out.C, sc2_keep);
    const bool padded = w.Cout_src > 0 && w.Cout_src < w.Cout;      // output channels padded inside the library (the heat-map heads)
    if (padded && r.gn && !b.rc) { nm_set_error("detector_backward: a padded layer with GroupNorm"); b.rc = NM_ERR_STATE; }
    const float* dy = norm_bwd(b, r.out, r.gn, r.fpart, r.nblk, r.chsum, w.key + ".bias", dA, ds.amax, dA_mul, dA_dv, dA_wv, padded ? w.Cout_src : 0);
    const int slot = b.last_slot;                        // >= 0: dY sits in the ring (Bwd::dy_alloc)
    const TensorRef dyT = plain(dy, r.out);
    const TensorRef dyS = ds.apply(b, dyT);
    // weight gradient; on the third stream when everything it reads outlives this call (Bwd::async_w)
    const bool up_rebuild = r.up2 && !r.upmat.p;         // (the forward kept the upsampled input: conv_gn, one-product modes)
    const size_t up_floats = up_rebuild ? Bwd::r64(Bwd::fl(numel_of(in) * 8, h_fine)) : 0;
    const size_t wg_floats = nm_wgrad_ws_floats(in.N, r.out.D, r.out.H, r.out.W, w.Cout, in.C, w.ks, r.stride);
    bool side = b.async_w && slot >= 0 && (!ds.amax || ds.pool) && !padded;
    if (side && b.ws.dry) b.need_scratch = std::max(b.need_scratch, up_floats + Bwd::r64(wg_floats));
    if (side && !b.ws.dry && up_floats + wg_floats > b.sscratch_floats) side = false;
    // (enqueued BEHIND this layer's data gradient: started together with it, the two matrix-core kernels only share the CUs; started
    //  behind it, the weight gradient runs beside the next layer's GroupNorm passes, which are HBM work)
    auto side_wgrad = [&]() {
        float* gw = b.grad(w.key + ".weight", (int64_t)w.Cout * w.Cin * w.ks * w.ks * w.ks);
        if (b.live()) {
            hipStream_t s3 = b.c->stream3;
            b.run(nm_check_hip(hipEventRecord(b.c->ev_dy, b.s), "backward: dY-ready event"));
            b.run(nm_check_hip(hipStreamWaitEvent(s3, b.c->ev_dy, 0), "backward: weight-gradient stream wait"));
            TensorRef a = r.up2 && !up_rebuild ? r.upmat : in;
            if (up_rebuild) {
                b.run(nm_launch_upsample2(in, b.sscratch, s3, h_fine));
                a = with_h(mk(b.sscratch, in.N, 2 * in.D, 2 * in.H, 2 * in.W, in.C), h_fine);
            }
            b.run(nm_launch_wgrad(a, dyS, w.ks, r.stride, r.pad, w.Cin, b.sscratch + up_floats, gw, s3, ds.inv(), split ? 1 : 0));
            b.run(nm_check_hip(hipEventRecord(b.c->ev_w[slot], s3), "backward: weight-gradient done event"));
            b.slot_busy[slot] = true;
        }
    };
    const bool side_first = nm_ls().wgrad_async == 2;          // 2: enqueue it in front of the data gradient (A/B)
    if (side && side_first) side_wgrad();
    if (!side) {
        const size_t m2 = b.ws.mark();
        TensorRef a = r.up2 && !up_rebuild ? r.upmat : in;
        if (up_rebuild) {
            float* upb = b.alloc(Bwd::fl(numel_of(in) * 8, h_fine));
            if (b.live()) b.run(nm_launch_upsample2(in, upb, b.s, h_fine));
            a = with_h(mk(upb, in.N, 2 * in.D, 2 * in.H, 2 * in.W, in.C), h_fine);
        }
        float* wsb = b.alloc(wg_floats);
        const size_t per_row = (size_t)w.Cin * w.ks * w.ks * w.ks;
        float* gw = b.grad(w.key + ".weight", (int64_t)((padded ? w.Cout_src : w.Cout) * per_row));
        // (a padded layer: the kernel writes all w.Cout rows - the padded ones are exact zeros, dY is zero there - into scratch and the
        //  state_dict's rows are copied out)
        float* gw_all = padded ? b.alloc((size_t)w.Cout * per_row) : gw;
        if (b.live()) {
            b.run(nm_launch_wgrad(a, dyS, w.ks, r.stride, r.pad, w.Cin, wsb, gw_all, b.s, ds.inv(), split ? 1 : 0));
            if (padded) b.run(nm_check_hip(hipMemcpyAsync(gw, gw_all, (size_t)w.Cout_src * per_row * sizeof(float), hipMemcpyDeviceToDevice, b.s), "backward: copy"));
        }
        b.ws.release(m2);
    }
    if (need_din) {
        if (r.stride == 1) {
            const int us = r.up2 ? 2 : 1;
            ConvGeom g; g.ks = w.ks; g.stride = 1; g.pad = w.ks - 1 - r.pad; g.OD = us * in.D; g.OH = us * in.H; g.OW = us * in.W;
            g.Cout = w.csel; g.Co_pad = w.cd_pad;
            float* dfine = r.up2 ? b.alloc(Bwd::fl((size_t)in.N * g.OD * g.OH * g.OW * w.csel, h_fine)) : din;
            if (b.live()) {
                if (!w.wd) { nm_set_error("detector_backward: weights were not packed for training (nm_ctx_set_training)"); b.rc = NM_ERR_STATE; }
            }
            if (b.live()) {
                b.run(nm_launch_conv(dyS, w.wd, b.zb, dfine, g, nullptr, b.s, w.Cout, w.wd16, r.up2 ? h_fine : h_in));
                if (r.up2) b.run(nm_launch_upsample2_adjoint(dfine, in.N, in.D, in.H, in.W, w.csel, din, b.s, ds.inv(), h_fine, h_in));
                else if (ds.inv() && !out_mul) b.run(nm_launch_scale_by(din, (size_t)in.N * in.D * in.H * in.W * w.csel, ds.inv(), b.s, h_in));
            }
            if (out_mul && !r.up2) *out_mul = ds.inv();
        } else if (b.live()) {      // k2 s2 pool conv: the transposed conv
            if (!w.wt) { nm_set_error("detector_backward: weights were not packed for training (nm_ctx_set_training)"); b.rc = NM_ERR_STATE; }
            else if (ds.amax && nm_convT2_f16_eligible(dyS, w.Cin, in.D, in.H, in.W, h_in))           // f16 matrix cores: dY pre-scaled, the result un-scaled in the epilogue
                b.run(nm_launch_convT2(dyS, w.wt, b.zb, din, w.Cin, in.D, in.H, in.W, b.s, h_in, ds.inv()));
            else b.run(nm_launch_convT2(dyT, w.wt, b.zb, din, w.Cin, in.D, in.H, in.W, b.s, h_in));
        }
    }
    if (side && !side_first) side_wgrad();
    b.ws.release(m);
    return din;
}

void add_into(Bwd& b, float* dst, const float* src, size_t n) { if (b.live()) b.run(nm_launch_axpy(dst, src, n, b.s)); }

// Res3DBlock backward: dOut is the gradient of the (materialised) block output; returns the gradient of the activated input
float* res_bwd(Bwd& b, const ResRec& r, const float* dOut) {
    const float* m1 = nullptr;
    float* d1 = conv_bwd(b, r.c2, dOut, true, nullptr, &m1);      // its power-of-two scale is undone by c1's GroupNorm backward
    // (the results of c1 and of the skip conv are left with their power-of-two scales: the add that joins the branches undoes them)
    const float *mx = nullptr, *ms = nullptr;
    float* dx = conv_bwd(b, r.c1, d1, true, m1, &mx);
    const size_t n = numel_of(r.c1.in);
    const int hx = r.c1.in.h;              // storage type of the block input's gradient (dOut of a block without skip conv has the same shape, hence the same type)
    const float* other = dOut;
    if (r.has_skip) other = conv_bwd(b, r.cs, dOut, true, nullptr, &ms);
    if (b.live()) {
        if (!r.has_skip && r.c2.out.h != hx) { nm_set_error("res_bwd: block input and output gradients differ in storage type"); b.rc = NM_ERR_STATE; return dx; }
        if ((mx || ms) && n % 4 == 0) b.run(nm_launch_axpby(dx, mx, other, ms, n, b.s, hx));
        else {
            if (mx) b.run(nm_launch_scale_by(dx, n, mx, b.s, hx));
            if (ms) b.run(nm_launch_scale_by(const_cast<float*>(other), n, ms, b.s, hx));
            b.run(nm_launch_axpy(dx, other, n, b.s, hx));
        }
    }
    return dx;
}

// Upsample3DBlock (ConvTranspose3d k2 s2 + GN + LeakyReLU) backward
float* up_bwd(Bwd& b, const UpRec& r, const float* dA) {
    const UpW& w = *r.w;
    const TensorRef& in = r.in;
    float* din = b.alloc(numel_of(in));
    const size_t m = b.ws.mark();
    DyScale ds;
    ds.prepare(b, w.wd16 && nm_conv_get_mode() != 0, r.out.N * r.out.C);
    const float* dy = norm_bwd(b, r.out, &w.n, r.fpart, r.nblk, r.chsum, w.key + ".bias", dA, ds.amax);
    const TensorRef dyT = plain(dy, r.out);
    float* wsb = b.alloc(nm_wgrad_ws_floats(in.N, in.D, in.H, in.W, w.Cin, w.Cout, 2, 2));
    float* gw = b.grad(w.key + ".weight", (int64_t)w.Cin * w.Cout * 8);
    ConvGeom g; g.ks = 2; g.stride = 2; g.pad = 0; g.OD = in.D; g.OH = in.H; g.OW = in.W; g.Cout = w.Cin; g.Co_pad = w.cd_pad;
    if (b.live()) {
        if (!w.wd) { nm_set_error("detector_backward: weights were not packed for training (nm_ctx_set_training)"); b.rc = NM_ERR_STATE; }
    }
    const TensorRef dyS = ds.apply(b, dyT);
    if (b.live() && w.wd) b.run(nm_launch_wgrad(dyS, in, 2, 2, 0, w.Cout, wsb, gw, b.s, ds.inv(), ds.amax ? 1 : 0));      // roles swapped: [Cin][Cout][8] = IODHW
    if (b.live()) {
        b.run(nm_launch_conv(dyS, w.wd, b.zb, din, g, nullptr, b.s, w.Cout, w.wd16));
        if (ds.inv()) b.run(nm_launch_scale_by(din, numel_of(in), ds.inv(), b.s));
    }
    b.ws.release(m);
    return din;
}

float* hourglass_bwd(Bwd& b, const HgRec& h, const float* dOut) {
    float* d_xd1 = up_bwd(b, h.u1, dOut);
    float* d_xa2 = res_bwd(b, h.d1, d_xd1);
    float* d_xd2 = up_bwd(b, h.u2, d_xa2);
    float* d_xa3 = res_bwd(b, h.d2, d_xd2);
    float* d_xd3 = up_bwd(b, h.u3, d_xa3);
    float* d_xe3 = res_bwd(b, h.d3, d_xd3);
    float* d_p3 = res_bwd(b, h.e3, d_xe3);
    float* d_xe2 = conv_bwd(b, h.p3, d_p3, true);
    add_into(b, d_xe2, res_bwd(b, h.s3, d_xa3), numel_of(h.p3.in));
    float* d_p2 = res_bwd(b, h.e2, d_xe2);
    float* d_xe1 = conv_bwd(b, h.p2, d_p2, true);
    add_into(b, d_xe1, res_bwd(b, h.s2, d_xa2), numel_of(h.p2.in));
    float* d_p1 = res_bwd(b, h.e1, d_xe1);
    float* d_x0 = conv_bwd(b, h.p1, d_p1, true);
    add_into(b, d_x0, res_bwd(b, h.s1, dOut), numel_of(h.p1.in));
    return d_x0;
}

void feature_net_bwd(Bwd& b, const FeatRec& r, const float* dFeat, bool sparse_occ) {
    const size_t m = b.ws.mark();
    float* d_hg = res_bwd(b, r.r5, dFeat);
    float* d_p3 = hourglass_bwd(b, r.hg, d_hg);
    float* d_r2 = conv_bwd(b, r.p3, d_p3, true);
    float* d_p1 = res_bwd(b, r.r2, d_r2);
    float* d_first = conv_bwd(b, r.p1, d_p1, true);
    // first layer: GroupNorm backward, then the weight gradient against cat[occ, coords] rebuilt from the occupancy
    const FeatNetW& w = *r.w;
    const float* dy = norm_bwd(b, r.first, &w.n0, r.fpart0, r.nblk0, r.chsum0, w.c0.key + ".bias", d_first);
    float* wsb = b.alloc(nm_wgrad_k5occ_ws_floats(r.N, r.G, w.c0.Cout));
    float* gw = b.grad(w.c0.key + ".weight", (int64_t)w.c0.Cout * 4 * 125);
    // (the main walk's first layer is the last thing the backward pass does: its two halves on two streams, nm_launch_wgrad_k5occ)
    const bool two = b.async_w && nm_ls().k5_two && b.s == b.c->stream && b.c->stream3 && b.c->stream3 != b.s;
    if (b.live()) b.run(nm_launch_wgrad_k5occ(r.occ, r.N, r.G, plain(dy, r.first), wsb, gw, b.s, sparse_occ ? 1 : 0,
                                               two ? b.c->stream3 : nullptr, b.c->ev_dy, b.c->ev_k5));
    b.ws.release(m);
}

int backward_graph(nm_ctx* c, const TrainTape& t, const float* dloss, const std::map<std::string, std::pair<float*, int64_t>>* grads) {
    Bwd b(c, grads);
    const DetectorW& d = c->det;
    const int K = c->cfg.nkeypoints, G = c->cfg.grid_size, g = G / 4, B = t.B, T = t.T, F = B * T, N = c->cfg.nneighbor;
    const size_t g3 = (size_t)g * g * g, G3 = (size_t)G * G * G;
    const double width_d = 2.0 * std::pow((double)c->cfg.gaussian_sigma / (double)g, 2.0);
    const std::string k2v = "kypt_detector.kypt_to_vox", v2k = "kypt_detector.vox_to_kypt";
    float* dkp = b.alloc((size_t)F * K * 4);
    float* dfeat = b.alloc((size_t)F * g3 * FEAT);
    b.zb = b.alloc(1024);
    b.amax_pool = reinterpret_cast<unsigned*>(b.zb + 512); b.amax_cap = 256;      // (the second stream's walk takes cells 256..511)
    if (nm_ls().wgrad_async && c->stream3) {
        b.async_w = true;
        if (!b.ws.dry) {
            b.slot_floats = c->wside_slot; b.sscratch_floats = c->wside_scratch; b.sc_floats = c->wside_sc;
            if (!c->wside || c->wside_floats < 3 * b.slot_floats + b.sscratch_floats + b.sc_floats) {
                nm_set_error("detector_backward: the weight-gradient side block was not sized (call nm_detector_forward_train first)"); return NM_ERR_STATE;
            }
            for (int k = 0; k < 3; ++k) b.ring[k] = c->wside + (size_t)k * b.slot_floats;
            b.sscratch = c->wside + 3 * b.slot_floats; b.scpool = b.sscratch + b.sscratch_floats;
        }
    }
    const size_t dgn_main = (size_t)2 << 20, dgn_side = (size_t)1 << 18;           // floats: 4 F C per GroupNorm layer (F C <= 8192: 60 layers)
    b.dgn_pool = nm_ls().defer_sums ? b.alloc(dgn_main + dgn_side) : nullptr; b.dgn_cap = dgn_main;
    if (b.live()) {
        b.run(nm_check_hip(hipMemsetAsync(dkp, 0, (size_t)F * K * 4 * sizeof(float), b.s), "backward: memset"));
        b.run(nm_check_hip(hipMemsetAsync(b.zb, 0, 1024 * sizeof(float), b.s), "backward: memset"));
    }

    {   // decoder: tail -> d11 -> d8 -> d4 -> d1 -> adjust -> combined representation
        const size_t m = b.ws.mark();
        const TensorRef& x = t.d11.out;
        const int tb = nm_tail_bwd_blocks(G), C = x.C;
        // the gradient of the decoder's last activated tensor is d14's weight row times one factor per voxel: with 32 channels and a
        // GroupNorm behind the layer only the factors are stored (67 MB for 2.1 GB) and its GroupNorm backward forms the products
        const bool rank1 = C == 32 && t.d11.gn != nullptr && nm_ls().tail_rank1;
        float* dA = rank1 ? nullptr : b.alloc(Bwd::fl((size_t)F * G3 * C, x.h));
        float* dvox = rank1 ? b.alloc((size_t)F * G3) : nullptr;
        float* part = b.alloc((size_t)F * tb * (C + 1));
        float* g14 = b.alloc(C + 1);
        float* gw14 = b.grad(k2v + ".decode_voxel_from_combined_representation.14.weight", C);
        float* gb14 = b.grad(k2v + ".decode_voxel_from_combined_representation.14.bias", 1);
        if (b.live()) {
            b.run(nm_launch_decoder_tail_bwd(x, d.d14, t.vox, t.recon, dloss, G, dA, part, b.s, dvox));
            b.run(nm_launch_sum_rows(part, F * tb, C + 1, g14, b.s));
            b.run(nm_check_hip(hipMemcpyAsync(gw14, g14, C * sizeof(float), hipMemcpyDeviceToDevice, b.s), "backward: copy"));
            b.run(nm_check_hip(hipMemcpyAsync(gb14, g14 + C, sizeof(float), hipMemcpyDeviceToDevice, b.s), "backward: copy"));
        }
        const float *m11 = nullptr, *m4 = nullptr;
        float* dx = conv_bwd(b, t.d11, dA, true, nullptr, &m11, dvox, rank1 ? d.d14 : nullptr);
        dx = conv_bwd(b, t.d8, dx, true, m11);
        dx = conv_bwd(b, t.d4, dx, true, nullptr, &m4);
        dx = conv_bwd(b, t.d1, dx, true, m4);
        float* dcomb = conv_bwd(b, t.adjust, dx, true);                 // [F][g^3][csel = 2K + FEAT]
        float* gws = b.alloc((size_t)F * K * 10);
        float* widthk = c->learn_sigma ? b.alloc(64) : nullptr;
        float* gsig = c->learn_sigma ? b.grad("kypt_detector.vox_to_kypt.sigmas", K) : nullptr;
        if (b.live()) {
            b.run(nm_check_hip(hipMemsetAsync(dfeat, 0, (size_t)F * g3 * FEAT * sizeof(float), b.s), "backward: memset"));
            if (widthk) b.run(nm_launch_gauss_width(d.sigma_param, K, 2.0f * c->cfg.gaussian_sigma, g, widthk, b.s));
            b.run(nm_launch_combined_bwd(dcomb, d.adjust.csel, t.table, t.keypoints, B, T, K, FEAT, g, (float)width_d, gws, dfeat, dkp, b.s, c->gauss_cat,
                                         widthk, d.sigma_param, 2.0f * c->cfg.gaussian_sigma, gsig));
            b.flush_sums();
            // every kypt_to_vox.* gradient is complete: the caller's collective for that bucket chunk may start behind this event
            if (c->ev_user_decoder) {
                if (b.async_w) {       // the decoder's weight gradients are on stream3: the caller's event goes behind them AND behind this point of the walk
                    b.run(nm_check_hip(hipEventRecord(c->ev_dy, b.s), "backward: decoder walk event"));
                    b.run(nm_check_hip(hipStreamWaitEvent(c->stream3, c->ev_dy, 0), "backward: weight-gradient stream wait"));
                    b.run(nm_check_hip(hipEventRecord(c->ev_user_decoder, c->stream3), "backward: decoder-done event"));
                } else b.run(nm_check_hip(hipEventRecord(c->ev_user_decoder, b.s), "backward: decoder-done event"));
            }
        }
        b.ws.release(m);
    }
    float* dinfl = t.affinity_on ? b.alloc((size_t)B * K * K) : nullptr;
    {   // keypoint-only losses
        const size_t m = b.ws.mark();
        float* cws = b.alloc((size_t)F * nm_chamfer_bwd_blocks(G) * K * 3);
        float* gvws = c->cfg.vol_fit_chamfer == 2 ? b.alloc(nm_volfit_gauss_bwd_ws_floats(B, T, G)) : nullptr;
        float* gaff = b.grad("kypt_detector.affinity_params", c->affinity_numel());
        if (b.live()) {
            if (c->cfg.vol_fit_chamfer == 1)
                b.run(nm_launch_chamfer_bwd(t.vox, t.keypoints, t.tail_part, nm_tail_blocks(G), dloss, F, K, G, cws, dkp, b.s));
            else if (c->cfg.vol_fit_chamfer == 2)
                b.run(nm_launch_volfit_gauss_bwd(t.vox, t.keypoints, dloss, B, T, K, G, c->cfg.gaussian_sigma, gvws, dkp, b.s));
            b.run(nm_launch_clip_loss_bwd(t.keypoints, t.affinity_on ? t.aff : nullptr, dloss, B, T, K, N, c->cfg.sep_sigma, c->cfg.use_graph_traj,
                                          dkp, dinfl, b.s));
            if (t.affinity_on) b.run(nm_launch_affinity_bwd(d.affinity_params, t.aff, dinfl, dloss, B, N, K, gaff, b.s, c->affinity_ver));
            else b.run(nm_check_hip(hipMemsetAsync(gaff, 0, (size_t)c->affinity_numel() * sizeof(float), b.s), "backward: memset"));
        }
        b.ws.release(m);
    }
    const int Kc = d.head.Cout;                 // channels per voxel of the head tensors and their gradients (K rounded up to 8)
    float* dchead = b.alloc((size_t)B * g3 * Kc);
    {   // keypoints <- heat-maps <- heads; head conv back into the frame features
        const size_t m = b.ws.mark();
        float* dhead = b.alloc((size_t)F * g3 * Kc);
        float* dchead_t = b.alloc((size_t)F * g3 * Kc);
        float* hws = b.alloc(nm_heat_bwd_ws_floats(F, K, g));
        float* gprop = b.alloc(4);
        float* gpw = b.grad(v2k + ".propagate_heatmaps.0.weight", 2);
        float* gpb = b.grad(v2k + ".propagate_heatmaps.0.bias", 1);
        if (b.live()) {
            b.run(nm_launch_heat_bwd(t.head_out, t.clip_head_out, d.prop, t.heat_part, t.heat_mean, t.keypoints, dkp, dloss, B, T, K, Kc, g, hws,
                                     dhead, dchead_t, dchead, gprop, b.s));
            b.run(nm_check_hip(hipMemcpyAsync(gpw, gprop, 2 * sizeof(float), hipMemcpyDeviceToDevice, b.s), "backward: copy"));
            b.run(nm_check_hip(hipMemcpyAsync(gpb, gprop + 2, sizeof(float), hipMemcpyDeviceToDevice, b.s), "backward: copy"));
        }
        float* dfh = conv_bwd(b, t.head, dhead, true);
        add_into(b, dfeat, dfh, (size_t)F * g3 * FEAT);
        b.ws.release(m);
    }
    {   // spatio-temporal net of the clip mean: B frames of small, latency-bound launches, issued on the side stream so that they
        // run beside the per-frame net's backward.  Their scratch stays out of reach of the main stream (the arena top is left at
        // this block's high-water mark), exactly as in the forward.
        Bwd b2(c, grads, c->stream2);
        b2.zb = b.zb; b2.amax_pool = b.amax_pool + 256; b2.amax_cap = 256;
        if (b.dgn_pool) { b2.dgn_pool = b.dgn_pool + dgn_main; b2.dgn_cap = dgn_side; }
        if (b.live()) {
            b.run(nm_check_hip(hipEventRecord(c->ev_fork, b.s), "backward: fork event"));
            b.run(nm_check_hip(hipStreamWaitEvent(c->stream2, c->ev_fork, 0), "backward: side stream wait"));
        }
        b2.rc = b.rc;
        const size_t saved_peak = b2.ws.peak;
        b2.ws.peak = b2.ws.top;
        float* dfclip = conv_bwd(b2, t.clip_head, dchead, true);
        // (the clip-mean grid is the union of T frames, 15-30 % dense: the gather form of the sparse first-layer gradient loses there, but
        //  the matrix-core form - frame-summed dY for the coordinate channels, non-empty bricks for the occupancy channel, exact fp32
        //  products - does not depend on 0 / 1 occupancies)
        feature_net_bwd(b2, t.clip, dfclip, nm_ls().clip_occ_mfma != 0);
        const size_t local_peak = b2.ws.peak;
        b2.ws.peak = saved_peak > local_peak ? saved_peak : local_peak;
        b2.ws.top = local_peak;
        b2.flush_sums();
        if (b2.live()) b2.run(nm_check_hip(hipEventRecord(c->ev_side, c->stream2), "backward: side event"));
        b.run(b2.rc);
    }
    feature_net_bwd(b, t.frame, dfeat, true);
    b.flush_sums();
    if (b.live()) b.run(nm_check_hip(hipStreamWaitEvent(b.s, c->ev_side, 0), "backward: join side stream"));
    if (b.async_w) {
        if (b.ws.dry) { c->wside_slot = b.need_slot; c->wside_scratch = b.need_scratch; c->wside_sc = b.need_sc; }
        else if (b.rc == NM_OK) {
            b.run(nm_check_hip(hipEventRecord(c->ev_wjoin, c->stream3), "backward: weight-gradient stream event"));
            b.run(nm_check_hip(hipStreamWaitEvent(b.s, c->ev_wjoin, 0), "backward: join weight-gradient stream"));
        }
    }
    if (!b.ws.dry && b.rc != NM_OK) {
        // a failed walk may have queued side-stream work that still writes the caller's gradient buffers and reads the dY ring: the
        // header promises that nothing of this call is in flight on another stream when it returns an error
        (void)hipStreamSynchronize(c->stream2);
        if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    }
    return b.rc;
}

template <class Fn>
int with_workspace(nm_ctx* c, Fn&& graph) {
    // pass 1: dry run to size the workspace; pass 2: launch
    c->ws.dry = true; c->ws.peak = 0; c->ws.overflow = false;
    int rc = graph();
    c->ws.dry = false;
    if (rc) return rc;
    rc = nm_ctx_reserve(c, c->ws.peak + 4096);
    if (rc) return rc;
    return graph();
}

int check_ready(nm_ctx* c, const char* who) {
    if (!c) { nm_set_error("%s: null ctx", who); return NM_ERR_ARG; }
    if (!c->has_weights) { nm_set_error("%s: nm_ctx_set_weights has not been called", who); return NM_ERR_STATE; }
    nm_elem_set_nonfinite_flag(c->nf_flag);        // GroupNorm finalisation reports non-finite conv statistics into this ctx's status word
    int rc = nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
    if (!rc) rc = nm_nf_poll(c);
    return rc;
}

}  // namespace

// ---- deferred range guard (include/nm355.h "Range / finiteness status") -----------------------------------------------------------
// nf_post: behind a forward-type call, on its stream: status word -> pinned host slot, event.  nf_poll: at the entry of every later
// call: slots whose event has completed are read (hipEventQuery, no wait); a slot two or more calls old is waited for - the device
// is at least one whole call behind it, so the wait returns at once and the host never runs more than ~2 calls ahead of a report.
// A set slot clears the device word, drops the younger slots (they saw the same sticky word) and fails the CURRENT call with
// NM_ERR_RANGE naming the call that overflowed: its outputs, handed out earlier, hold NaN / inf.
int nm_nf_poll(nm_ctx* c) {
    if (!c->range_check || !c->nf_host) return NM_OK;
    int bad = -1;
    for (int i = 0; i < 4; ++i) {
        if (!c->nf_busy[i]) continue;
        const bool old = c->nf_seq[i] + 2 <= c->nf_calls;
        hipError_t e = old ? hipEventSynchronize(c->ev_nf[i]) : hipEventQuery(c->ev_nf[i]);
        if (e == hipErrorNotReady) continue;
        if (e != hipSuccess) return nm_check_hip(e, "range guard: status event");
        c->nf_busy[i] = false;
        if (c->nf_host[i] && (bad < 0 || c->nf_seq[i] < c->nf_seq[bad])) { bad = i; c->nf_last = c->nf_host[i]; }
        else c->nf_host[i] = 0;
    }
    if (bad < 0) return NM_OK;
    const uint64_t seq = c->nf_seq[bad]; const char* who = c->nf_who[bad] ? c->nf_who[bad] : "?";
    for (int i = 0; i < 4; ++i) {          // error path: drain the younger copies (they read the same sticky word), then clear it
        if (c->nf_busy[i]) (void)hipEventSynchronize(c->ev_nf[i]);
        c->nf_busy[i] = false; c->nf_host[i] = 0;
    }
    (void)hipMemsetAsync(c->nf_flag, 0, sizeof(unsigned), c->stream);
    if (c->nf_last & 2u) {
        // the chain's workgroups were not all resident (a busy or partitioned device): this context stops using it - later rollouts take
        // the launch-per-phase steps, and the captured graphs that hold the persistent launch are dropped
        (void)hipDeviceSynchronize();            // (rare error path: nothing of the aborted rollout may still reference the graphs' buffers)
        c->ls.vrnn_chain = 0;
        nm_vrnn_free_graphs(c);
        nm_set_error("call #%llu (%s): the persistent rollout kernel timed out waiting for its workgroups (its outputs are invalid); "
                     "this context now takes the launch-per-phase steps (as NM355_VRNN_CHAIN=0 does from the start) - repeat the call",
                     (unsigned long long)seq, who);
        return NM_ERR_STATE;
    }
    nm_set_error("call #%llu (%s) produced non-finite values - its outputs are invalid: %s", (unsigned long long)seq, who,
                 nm_conv_get_mode() != 0
                 ? "in the split-fp16 conv mode an activation beyond the fp16 range (|x| >= 65520) or a non-finite input does that where the "
                   "reference's fp32 arithmetic stays finite - set conv mode 'fp32' (exact fp32 MFMA, no range limit) or 'auto' and run again"
                 : "the input or the weights hold inf / NaN (exact fp32 mode has no range limit of its own)");
    return NM_ERR_RANGE;
}
void nm_nf_post(nm_ctx* c, const char* who) {
    ++c->nf_calls;
    if (!c->range_check || !c->nf_host) return;
    int k = -1;
    for (int i = 0; i < 4; ++i) if (!c->nf_busy[i]) { k = i; break; }
    if (k < 0) return;                     // (cannot happen: nf_poll retires every slot two calls old)
    c->nf_host[k] = 0;
    if (hipMemcpyAsync(c->nf_host + k, c->nf_flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return;
    if (hipEventRecord(c->ev_nf[k], c->stream) != hipSuccess) { (void)hipStreamSynchronize(c->stream); return; }
    c->nf_busy[k] = true; c->nf_seq[k] = c->nf_calls; c->nf_who[k] = who;
}


void nm_net_free_tape(nm_ctx* c) { delete c->tape; c->tape = nullptr; }

int nm_net_set_weights(nm_ctx* c, const std::map<std::string, std::pair<const float*, int64_t>>& sd) { NmScope nm_scope_(c);
    if (c->tape) c->tape->valid = false;
    nm_vrnn_invalidate_tape(c);
    int rc = NM_OK;
    c->owned_cursor = 0;                   // buffers are reused in call order (nm_ctx_weight_alloc): no free, no sync
    c->has_weights = false;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4;
    Loader L{c, sd};
    DetectorW& d = c->det;
    const std::string v = "kypt_detector.vox_to_kypt", k2v = "kypt_detector.kypt_to_vox";
    const std::string dec = k2v + ".decode_voxel_from_combined_representation";
    d.affinity_params = L.copy("kypt_detector.affinity_params", c->affinity_numel());
    d.sigma_param = c->learn_sigma ? L.copy("kypt_detector.vox_to_kypt.sigmas", K) : nullptr;
    d.zeros = nm_ctx_weight_alloc(c, 512);
    if (d.zeros) (void)hipMemsetAsync(d.zeros, 0, 512 * sizeof(float), c->stream);
    d.frame = L.featnet(v + ".extract_features", FEAT);
    // (the two heat-map heads run with their K output channels zero-padded to Kc = K rounded up to 8: head / clip_head tensors and their
    //  gradients have Kc channels per voxel; the heat-map kernels read channels < K only)
    const int Kc = (K + 7) & ~7;
    d.head = L.conv(v + ".extract_heatmaps_from_features.0", K, FEAT, 1, false, Kc);
    d.clip = L.featnet(v + ".extract_spatio_temporal_features", 2 * FEAT);
    d.clip_head = L.conv(v + ".extract_spatio_temporal_heatmaps_from_features.0", K, 2 * FEAT, 1, false, Kc);
    {
        const float* pw = L.get(v + ".propagate_heatmaps.0.weight", 2);
        const float* pb = L.get(v + ".propagate_heatmaps.0.bias", 1);
        d.prop = nm_ctx_weight_alloc(c, 3);
        if (pw && pb && d.prop) hipLaunchKernelGGL(pack_small_kernel, dim3(1), dim3(64), 0, c->stream, pw, 2, pb, 1, d.prop);
    }
    {   // inference: the same layer split by linearity (nm_heads.hip): per-clip conv over the frame-independent channels + per-frame gaussians
        const std::string key = k2v + ".adjust_combined_representation.0";
        const int Cin = FEAT + 2 * K + 3, Cr = Cin - K;
        ConvW& w = d.adjust_rest;
        w = ConvW(); w.Cin = Cr; w.Cout = FEAT; w.ks = 1; w.Cin_pad = (Cr + 7) & ~7; w.Co_pad = (FEAT + 31) & ~31; w.key = key;
        d.adjust_wg = nullptr;
        const float* src = L.get(key + ".weight", (int64_t)FEAT * Cin);
        if (src && !c->training && nm_ls().adjust_split) {
            w.wp = nm_ctx_weight_alloc(c, nm_packed_weight_floats(1, w.Cin_pad, w.Co_pad));
            d.adjust_wg = nm_ctx_weight_alloc(c, (size_t)K * FEAT);
            if (!w.wp || !d.adjust_wg) { if (!L.rc) { nm_set_error("set_weights: hipMalloc failed"); L.rc = NM_ERR_HIP; } }
            else {
                L.pack(src + K, w.wp, nullptr, FEAT, Cr, 1, w.Cin_pad, w.Co_pad, Cin, 0);      // columns K.. of every row (row stride Cin)
                int r = nm_launch_adjust_wg(src, FEAT, Cin, K, d.adjust_wg, c->stream);
                if (r && !L.rc) L.rc = r;
            }
        }
    }
    d.adjust = L.conv(k2v + ".adjust_combined_representation.0", FEAT, FEAT + 2 * K + 3, 1, false, 0, FEAT + 2 * K);       // (pad16 = true puts it on conv_f16s: measured 648 us vs 319 us on the fp32 kernel, which stages all 184 channels per pass; a 1-tap layer has 6 MFMAs per staged 16-channel chunk)
    d.d1 = L.conv(dec + ".1", FEAT / 2, FEAT, 3); L.up2_sets(dec + ".1", d.d1, c->cfg.grid_size / 4); d.dn2 = L.norm(dec + ".2", FEAT / 2);
    d.d4 = L.conv(dec + ".4", FEAT / 2, FEAT / 2, 3); d.dn5 = L.norm(dec + ".5", FEAT / 2);
    d.d8 = L.conv(dec + ".8", FEAT / 4, FEAT / 2, 3); L.up2_sets(dec + ".8", d.d8, c->cfg.grid_size / 2); d.dn9 = L.norm(dec + ".9", FEAT / 4);
    d.d11 = L.conv(dec + ".11", FEAT / 4, FEAT / 4, 3); d.dn12 = L.norm(dec + ".12", FEAT / 4);
    {
        const float* w = L.get(dec + ".14.weight", FEAT / 4);
        const float* b = L.get(dec + ".14.bias", 1);
        d.d14 = nm_ctx_weight_alloc(c, FEAT / 4 + 1);
        if (w && b && d.d14) hipLaunchKernelGGL(pack_small_kernel, dim3(1), dim3(64), 0, c->stream, w, FEAT / 4, b, 1, d.d14);
    }
    VrnnW& r = c->vrnn;
    const std::string m = "dyna_module";
    r.h0 = L.copy(m + ".init_kypt_rnn_state", H);
    r.offset_param = L.copy(m + ".offset_param", (int64_t)K * 3);
    r.post0 = L.linear(m + ".extract_post_dist.0", 128, H + S4); r.post2 = L.linear(m + ".extract_post_dist.2", 2 * Z, 128);
    r.prior0 = L.linear(m + ".extract_prior_dist.0", 128, H); r.prior2 = L.linear(m + ".extract_prior_dist.2", 2 * Z, 128);
    r.root0 = L.linear(m + ".root_intensity_decoder.0", 128, H + Z); r.root2 = L.linear(m + ".root_intensity_decoder.2", 3 + K, 128);
    r.joint0 = L.linear(m + ".joint_matrix_decoder.0", 128, H + Z); r.joint2 = L.linear(m + ".joint_matrix_decoder.2", 6 * K, 128);
    r.w_ih = L.copy(m + ".kypt_rnn_cell.weight_ih", (int64_t)3 * H * (S4 + Z));
    r.w_hh = L.copy(m + ".kypt_rnn_cell.weight_hh", (int64_t)3 * H * H);
    r.b_ih = L.copy(m + ".kypt_rnn_cell.bias_ih", 3 * H);
    r.b_hh = L.copy(m + ".kypt_rnn_cell.bias_hh", 3 * H);
    if (!r.parents) {                   // one block: parents[K], order[K], lvl_joint[K], lvl_start[K + 2] (nm_vrnn_set_tree fills them)
        if (hipMalloc(reinterpret_cast<void**>(&r.parents), (4 * K + 2) * sizeof(int32_t)) != hipSuccess) {
            nm_set_error("set_weights: hipMalloc(tree) failed"); return NM_ERR_HIP;
        }
        r.order = r.parents + K; r.lvl_joint = r.parents + 2 * K; r.lvl_start = r.parents + 3 * K;
    }
    if (L.rc) return L.rc;
    if ((rc = L.flush_copies())) return rc;
    if ((rc = L.flush_packs())) return rc;
    rc = nm_check_hip(hipGetLastError(), "set_weights: pack kernels");
    if (rc) return rc;
    if (c->owned_cursor < c->owned.size()) {           // fewer buffers than last time (training packs switched off): drop the rest
        (void)hipDeviceSynchronize();
        for (size_t i = c->owned_cursor; i < c->owned.size(); ++i) (void)hipFree(c->owned[i]);
        c->owned.resize(c->owned_cursor); c->owned_bytes.resize(c->owned_cursor);
    }
    // (no final sync: the pack kernels read the caller's tensors in stream order, before anything the caller enqueues afterwards)
    c->has_weights = true;
    return NM_OK;
}

extern "C" {

size_t nm_workspace_bytes(nm_ctx* c, int32_t B, int32_t T) try { NmScope nm_scope_(c);
    if (!c || !c->has_weights || B <= 0 || T <= 0) return 0;
    c->ws.dry = true; c->ws.peak = 0;
    float dummy = 0.f; float* dp = &dummy;
    (void)detector_graph(c, dp, B, T, 1, dp, dp, dp, dp, dp, dp);
    c->ws.dry = false;
    return c->ws.peak + 4096;
} catch (...) { if (c) c->ws.dry = false; (void)nm_abi_catch("nm_workspace_bytes"); return 0; }

int nm_ctx_memory(nm_ctx* c, size_t out[4]) try { NmScope nm_scope_(c);
    if (!c || !out) { nm_set_error("ctx_memory: null argument"); return NM_ERR_ARG; }
    out[0] = c->ws.cap + c->ws2.cap; out[1] = c->ws_t.cap; out[2] = c->wside_floats * sizeof(float); out[3] = 0;
    for (size_t b : c->owned_bytes) out[3] += b;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_memory"); }

int nm_detector_forward(nm_ctx* c, const float* vox, int32_t B, int32_t T, int32_t affinity_on, float* keypoints,
                        float* heatmaps, float* first_feature, float* recon, float* affinity, float* losses11) try { NmScope nm_scope_(c);
    int rc = check_ready(c, "detector_forward");
    if (rc) return rc;
    if (!vox || !keypoints || !heatmaps || !first_feature || !recon || !losses11 || B <= 0 || T <= 0) {
        nm_set_error("detector_forward: null / non-positive argument"); return NM_ERR_ARG;
    }
    rc = with_workspace(c, [&]() { return detector_graph(c, vox, B, T, affinity_on, keypoints, heatmaps, first_feature, recon, affinity, losses11); });
    if (!rc) nm_nf_post(c, "nm_detector_forward");
    return rc;
} catch (...) { return nm_abi_catch("nm_detector_forward"); }

int nm_detector_keypoints(nm_ctx* c, const float* vox, int32_t B, int32_t T, int32_t affinity_on, float* keypoints, float* heatmaps,
                          float* first_feature, float* affinity) try { NmScope nm_scope_(c);
    int rc = check_ready(c, "detector_keypoints");
    if (rc) return rc;
    if (!vox || !keypoints || !heatmaps || !first_feature || B <= 0 || T <= 0) { nm_set_error("detector_keypoints: null / non-positive argument"); return NM_ERR_ARG; }
    rc = with_workspace(c, [&]() { return detector_graph(c, vox, B, T, affinity_on, keypoints, heatmaps, first_feature, nullptr, affinity, nullptr); });
    if (!rc) nm_nf_post(c, "nm_detector_keypoints");
    return rc;
} catch (...) { return nm_abi_catch("nm_detector_keypoints"); }

int nm_forward_fused(nm_ctx* c, const float* vox, int32_t B, int32_t T, int32_t affinity_on, const float* eps, int32_t S,
                     float* keypoints, float* heatmaps, float* first_feature, float* recon, float* affinity, float* losses11,
                     float* kypt_recon, float* R, float* z, float* h, float* scalars2, int32_t* best_idx) try { NmScope nm_scope_(c);
    int rc = check_ready(c, "forward_fused");
    if (rc) return rc;
    if (!vox || !eps || !keypoints || !heatmaps || !first_feature || !recon || !losses11 || !kypt_recon || !R || !z || !h ||
        !scalars2 || B <= 0 || T <= 0 || S <= 0) { nm_set_error("forward_fused: null / non-positive argument"); return NM_ERR_ARG; }
    if (!c->vrnn.has_tree) { nm_set_error("forward_fused: nm_vrnn_set_tree has not been called"); return NM_ERR_STATE; }
    // VRNN encode on the side stream, beside the decoder: it only needs the keypoints
    std::function<int()> hook = [&]() -> int {
        int r = nm_check_hip(hipEventRecord(c->ev_kp, c->stream), "keypoints event");
        if (!r) r = nm_check_hip(hipStreamWaitEvent(c->stream2, c->ev_kp, 0), "side stream wait");
        if (r) return r;
        std::swap(c->ws, c->ws2);
        hipStream_t main = c->stream;
        c->stream = c->stream2;
        c->in_fused = true;
        r = nm_vrnn_encode(c, keypoints, eps, B, T, S, kypt_recon, R, z, h, scalars2, best_idx);
        c->in_fused = false;
        c->stream = main;
        std::swap(c->ws, c->ws2);
        if (!r) r = nm_check_hip(hipEventRecord(c->ev_side, c->stream2), "side event");
        return r;
    };
    rc = with_workspace(c, [&]() { return detector_graph(c, vox, B, T, affinity_on, keypoints, heatmaps, first_feature, recon, affinity, losses11, &hook); });
    if (rc) return rc;
    rc = nm_check_hip(hipStreamWaitEvent(c->stream, c->ev_side, 0), "join side stream");
    if (!rc) nm_nf_post(c, "nm_forward_fused");
    return rc;
} catch (...) { return nm_abi_catch("nm_forward_fused"); }

int nm_decode_from_keypoints(nm_ctx* c, const float* keypoints, const float* first_feature, const float* first_frame,
                             int32_t B, int32_t Tg, float* gen) try { NmScope nm_scope_(c);
    int rc = check_ready(c, "decode_from_keypoints");
    if (rc) return rc;
    if (!keypoints || !first_feature || !first_frame || !gen || B <= 0 || Tg <= 0) {
        nm_set_error("decode_from_keypoints: null / non-positive argument"); return NM_ERR_ARG;
    }
    rc = with_workspace(c, [&]() { return decode_graph(c, keypoints, first_feature, first_frame, B, Tg, gen); });
    if (!rc) nm_nf_post(c, "nm_decode_from_keypoints");
    return rc;
} catch (...) { return nm_abi_catch("nm_decode_from_keypoints"); }

// the ctx-owned block behind the weight-gradient stream (ring of dY buffers + scratch + scale pool), grown to the last sizing pass
static int reserve_wside(nm_ctx* c) {
    const size_t want = 3 * c->wside_slot + c->wside_scratch + c->wside_sc;
    if (want <= c->wside_floats) return NM_OK;
    int rc = nm_check_hip(hipDeviceSynchronize(), "reserve: device sync");
    if (rc) return rc;
    if (c->wside) { (void)hipFree(c->wside); c->wside = nullptr; c->wside_floats = 0; }
    const size_t got = want + (want / 16 < ((size_t)16 << 20) ? want / 16 : ((size_t)16 << 20)) + 4096;      // (floats: at most 64 MB of growth slack)
    rc = nm_check_hip(hipMalloc(reinterpret_cast<void**>(&c->wside), got * sizeof(float)), "reserve: hipMalloc weight-gradient side block");
    if (rc) return rc;
    c->wside_floats = got;
    return NM_OK;
}

int nm_detector_forward_train(nm_ctx* c, const float* vox, int32_t B, int32_t T, int32_t affinity_on, float* keypoints,
                              float* heatmaps, float* first_feature, float* recon, float* affinity, float* losses11) try { NmScope nm_scope_(c);
    int rc = check_ready(c, "detector_forward_train");
    if (rc) return rc;
    if (!c->training) { nm_set_error("detector_forward_train: call nm_ctx_set_training(ctx, 1) and nm_ctx_set_weights first"); return NM_ERR_STATE; }
    if (!vox || !keypoints || !heatmaps || !first_feature || !recon || !losses11 || B <= 0 || T <= 0) {
        nm_set_error("detector_forward_train: null / non-positive argument"); return NM_ERR_ARG;
    }
    if (nm_ls().store16) {
        // 16-bit storage keeps the hourglass (and everything the heads, keypoints and VRNN read) in fp32: the g = G / 4 grid must stay
        // below the storage threshold (G <= 124 at the default 32^3)
        const size_t g = (size_t)c->cfg.grid_size / 4;
        if (g * g * g >= (size_t)nm_ls().store16_min) {
            nm_set_error("detector_forward_train: conv mode 4 (bfloat16 storage) needs (grid_size / 4)^3 < %d voxels", nm_ls().store16_min);
            return NM_ERR_UNSUPPORTED;
        }
    }
    if (!c->tape) c->tape = new TrainTape();
    TrainTape& t = *c->tape;
    t.valid = false;
    std::swap(c->ws, c->ws_t);
    // sizing pass over forward + backward (the arena must not move between the two calls), then the forward proper
    c->ws.dry = true; c->ws.peak = 0; c->ws.overflow = false;
    rc = detector_graph(c, vox, B, T, affinity_on, keypoints, heatmaps, first_feature, recon, affinity, losses11, nullptr, &t);
    c->wside_slot = c->wside_scratch = c->wside_sc = 0;
    if (!rc) rc = backward_graph(c, t, nullptr, nullptr);
    c->ws.dry = false;
    if (!rc) rc = nm_ctx_reserve(c, c->ws.peak + 4096);
    if (!rc) rc = reserve_wside(c);
    if (!rc) rc = detector_graph(c, vox, B, T, affinity_on, keypoints, heatmaps, first_feature, recon, affinity, losses11, nullptr, &t);
    t.fwd_top = c->ws.top;
    std::swap(c->ws, c->ws_t);
    t.valid = rc == NM_OK;
    if (!rc) nm_nf_post(c, "nm_detector_forward_train");
    return rc;
} catch (...) { return nm_abi_catch("nm_detector_forward_train"); }

int nm_detector_backward(nm_ctx* c, const float* dlosses11, const nm_named_grad* grads, int32_t count) try { NmScope nm_scope_(c);
    int rc = check_ready(c, "detector_backward");
    if (rc) return rc;
    if (!c->tape || !c->tape->valid) { nm_set_error("detector_backward: no training forward to back-propagate (or the weights changed since)"); return NM_ERR_STATE; }
    if (!dlosses11 || !grads || count <= 0) { nm_set_error("detector_backward: null argument"); return NM_ERR_ARG; }
    std::map<std::string, std::pair<float*, int64_t>> gm;
    for (int i = 0; i < count; ++i) {
        if (!grads[i].name || !grads[i].data) { nm_set_error("detector_backward: entry %d is null", i); return NM_ERR_ARG; }
        gm[grads[i].name] = std::make_pair(grads[i].data, grads[i].numel);
    }
    TrainTape& t = *c->tape;
    std::swap(c->ws, c->ws_t);
    c->ws.top = t.fwd_top;
    rc = backward_graph(c, t, dlosses11, &gm);
    std::swap(c->ws, c->ws_t);
    t.valid = false;
    return rc;
} catch (...) { return nm_abi_catch("nm_detector_backward"); }

int nm_ctx_set_backward_event(nm_ctx* c, void* hip_event) try { NmScope nm_scope_(c);
    if (!c) { nm_set_error("set_backward_event: null ctx"); return NM_ERR_ARG; }
    c->ev_user_decoder = static_cast<hipEvent_t>(hip_event);
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_set_backward_event"); }

int nm_voxelize_clip(nm_ctx* c, const double* points, int32_t T, int64_t N, double scale, float* vox, int32_t* idx_out) try { NmScope nm_scope_(c);
    if (!c || !points || !vox || T <= 0 || N <= 0) { nm_set_error("voxelize_clip: bad argument"); return NM_ERR_ARG; }
    int rc = nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
    if (rc) return rc;
    if ((rc = nm_ctx_reserve(c, 256 * 6 * sizeof(double) + 4096))) return rc;
    c->ws.release(0);
    double* part = static_cast<double*>(c->ws.alloc_bytes(256 * 6 * sizeof(double)));
    return nm_launch_voxelize(points, T, (size_t)N, c->cfg.grid_size, scale, part, vox, idx_out, c->stream);
} catch (...) { return nm_abi_catch("nm_voxelize_clip"); }

int nm_get_affinity(nm_ctx* c, float* affinity) try { NmScope nm_scope_(c);
    int rc = check_ready(c, "get_affinity");
    if (rc) return rc;
    if (!affinity) { nm_set_error("get_affinity: null output"); return NM_ERR_ARG; }
    return nm_launch_affinity(c->det.affinity_params, c->cfg.nneighbor, c->cfg.nkeypoints, affinity, c->stream, c->affinity_ver);
} catch (...) { return nm_abi_catch("nm_get_affinity"); }

int nm_ctx_set_learnable_sigma(nm_ctx* c, int32_t on) try { NmScope nm_scope_(c);
    if (!c) { nm_set_error("ctx_set_learnable_sigma: null context"); return NM_ERR_ARG; }
    if (c->cfg.vol_fit_chamfer == 2 && on) { nm_set_error("ctx_set_learnable_sigma: not implemented together with vol_fit_type 'gaussian'"); return NM_ERR_UNSUPPORTED; }
    if ((on != 0) != (c->learn_sigma != 0)) { c->learn_sigma = on ? 1 : 0; c->has_weights = false; }      // one more / one fewer tensor in the state_dict
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_set_learnable_sigma"); }

int nm_ctx_set_gaussian_cat(nm_ctx* c, int32_t cat) try { NmScope nm_scope_(c);
    if (!c) { nm_set_error("ctx_set_gaussian_cat: null context"); return NM_ERR_ARG; }
    if (cat < 0 || cat > 2) { nm_set_error("ctx_set_gaussian_cat: %d (0 'none', 1 'max', 2 'sum')", cat); return NM_ERR_UNSUPPORTED; }
    c->gauss_cat = cat;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_set_gaussian_cat"); }

int nm_ctx_set_affinity_ver(nm_ctx* c, int32_t ver) try { NmScope nm_scope_(c);
    if (!c) { nm_set_error("ctx_set_affinity_ver: null context"); return NM_ERR_ARG; }
    if (ver < 0 || ver > 3) { nm_set_error("ctx_set_affinity_ver: version %d (0 .. 3; 4 = Gumbel noise is not implemented)", ver); return NM_ERR_UNSUPPORTED; }
    if (ver != c->affinity_ver) { c->affinity_ver = ver; c->has_weights = false; }      // the parameter's shape changes: weights must be set again
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_set_affinity_ver"); }

}  // extern "C"
