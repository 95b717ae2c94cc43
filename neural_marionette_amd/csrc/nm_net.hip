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
#include <cmath>
#include <functional>

namespace {

const float LRELU = 0.01f;
const int FEAT = 128;
const size_t FRAME_CHUNK = 64;     // frames per pass through the conv stacks (bounds the workspace)

// ------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------
struct Loader {
    nm_ctx* c;
    const std::map<std::string, std::pair<const float*, int64_t>>& sd;
    int rc = NM_OK;

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
        int r = nm_check_hip(hipMemcpyAsync(dst, src, numel * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "set_weights: copy");
        if (r && !rc) rc = r;
        return dst;
    }
    ConvW conv(const std::string& p, int Cout, int Cin, int ks) {
        ConvW w; w.Cin = Cin; w.Cout = Cout; w.ks = ks; w.Cin_pad = (Cin + 7) & ~7; w.Co_pad = (Cout + 31) & ~31;
        const float* src = get(p + ".weight", (int64_t)Cout * Cin * ks * ks * ks);
        w.bias = copy(p + ".bias", Cout);
        if (!src) return w;
        w.wp = nm_ctx_weight_alloc(c, nm_packed_weight_floats(ks, w.Cin_pad, w.Co_pad));
        if (!w.wp) { if (!rc) { nm_set_error("set_weights: hipMalloc failed"); rc = NM_ERR_HIP; } return w; }
        int r = nm_launch_pack_conv_weight(src, Cout, Cin, ks, w.wp, w.Cin_pad, w.Co_pad, c->stream);
        if (!r && Cin % 8 == 0 && Cin >= 16) {                      // (Cin % 16 == 8: zero-padded, used by the small-volume core only)
            w.wp16 = nm_ctx_weight_alloc(c, nm_packed_weight_floats(ks, (Cin + 15) & ~15, w.Co_pad));      // same byte count as fp32
            if (!w.wp16) { if (!rc) { nm_set_error("set_weights: hipMalloc failed"); rc = NM_ERR_HIP; } return w; }
            r = nm_launch_pack_conv_weight16(src, Cout, Cin, ks, w.wp16, w.Co_pad, c->stream);
        }
        if (r && !rc) rc = r;
        return w;
    }
    NormW norm(const std::string& p, int C) {
        NormW n; n.C = C; n.groups = C / 16;
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
        UpW u; u.Cin = ci; u.Cout = co;
        const float* src = get(p + ".block.0.weight", (int64_t)ci * co * 8);
        u.w = nm_ctx_weight_alloc(c, (size_t)ci * co * 8);
        if (src && u.w) { int r = nm_launch_transpose_convT_weight(src, ci, co, u.w, c->stream); if (r && !rc) rc = r; }
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
            r = nm_launch_conv(t, f.c0.wp, f.c0.bias, f.field, g, nullptr, c->stream, 4);
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
    explicit Net(nm_ctx* ctx) : c(ctx), s(ctx->stream), ws(ctx->ws) {}
    Net(nm_ctx* ctx, hipStream_t stream) : c(ctx), s(stream), ws(ctx->ws) {}
    bool ok() const { return rc == NM_OK; }
    bool live() const { return rc == NM_OK && !ws.dry; }
    void run(int r) { if (r && !rc) rc = r; }
    float* alloc(size_t n) {
        float* p = ws.f(n);
        if (!p && !rc) { nm_set_error("workspace overflow (needed > %zu bytes)", ws.cap); rc = NM_ERR_STATE; }
        return p;
    }
};

TensorRef mk(const float* p, int N, int D, int H, int W, int C, const float* sc = nullptr, const float* sh = nullptr, float slope = 1.0f) {
    TensorRef t; t.p = p; t.scale = sc; t.shift = sh; t.slope = slope; t.N = N; t.D = D; t.H = H; t.W = W; t.C = C; return t;
}
size_t vox(const TensorRef& t) { return (size_t)t.D * t.H * t.W; }

// conv (+ optional GroupNorm statistics): returns the lazy output
// (up2: `in` is stored at half resolution and its trilinear x2 upsampling is what gets convolved)
TensorRef conv_gn(Net& n, const TensorRef& in, const ConvW& w, const NormW* gn, int stride, int pad, float slope_after,
                  float* out_buf = nullptr, bool up2 = false) {
    ConvGeom g; g.ks = w.ks; g.stride = stride; g.pad = pad; g.up2 = up2 ? 1 : 0;
    const int us = up2 ? 2 : 1;
    g.OD = (us * in.D + 2 * pad - w.ks) / stride + 1; g.OH = (us * in.H + 2 * pad - w.ks) / stride + 1; g.OW = (us * in.W + 2 * pad - w.ks) / stride + 1;
    g.Cout = w.Cout; g.Co_pad = w.Co_pad;
    const size_t ov = (size_t)g.OD * g.OH * g.OW;
    float* out = out_buf ? out_buf : n.alloc((size_t)in.N * ov * w.Cout);
    float *part = nullptr, *scale = nullptr, *shift = nullptr;
    const int nblk = nm_conv_blocks_per_frame(g);
    if (gn) {
        part = n.alloc((size_t)in.N * nblk * w.Cout * 2);
        scale = n.alloc((size_t)in.N * w.Cout); shift = n.alloc((size_t)in.N * w.Cout);
    }
    if (n.live()) {
        if (in.C != w.Cin_pad) { nm_set_error("conv_gn: input has %d channels, layer expects %d", in.C, w.Cin_pad); n.rc = NM_ERR_STATE; }
        else {
            n.run(nm_launch_conv(in, w.wp, w.bias, out, g, part, n.s, w.Cin, w.wp16));
            if (gn) n.run(nm_launch_gn_finalize(part, in.N, nblk, w.Cout, gn->groups, (double)ov * (w.Cout / gn->groups),
                                                gn->gamma, gn->beta, 1e-5f, scale, shift, n.s));
        }
    }
    return mk(out, in.N, g.OD, g.OH, g.OW, w.Cout, scale, shift, slope_after);
}

TensorRef add2(Net& n, const TensorRef& a, const TensorRef* b, float* out_buf = nullptr) {
    float* out = out_buf ? out_buf : n.alloc((size_t)a.N * vox(a) * a.C);
    if (n.live()) n.run(nm_launch_apply2(a, b, out, n.s));
    return mk(out, a.N, a.D, a.H, a.W, a.C);
}

// Res3DBlock (vox_modules.py:22-47): GN(conv3(lrelu(GN(conv3 x)))) + skip(x); the trailing
// F.leaky_relu(., True) is the identity.
TensorRef res(Net& n, const TensorRef& x, const ResW& w, float* out_buf = nullptr) {
    float* out = out_buf ? out_buf : n.alloc((size_t)x.N * vox(x) * w.c2.Cout);
    const size_t m = n.ws.mark();
    TensorRef r1 = conv_gn(n, x, w.c1, &w.n1, 1, 1, LRELU);
    TensorRef r2 = conv_gn(n, r1, w.c2, &w.n2, 1, 1, 1.0f);
    TensorRef sk = w.has_skip ? conv_gn(n, x, w.cs, &w.ns, 1, 0, 1.0f) : x;
    TensorRef o = add2(n, r2, &sk, out);
    n.ws.release(m);
    return o;
}

TensorRef pool(Net& n, const TensorRef& x, const PoolW& w) { return conv_gn(n, x, w.c, &w.n, 2, 0, LRELU); }

// Upsample3DBlock (vox_modules.py:63-75): ConvTranspose3d(k2,s2,output_padding) -> GN -> LeakyReLU (lazy)
TensorRef up(Net& n, const TensorRef& x, const UpW& w, int outpad) {
    const int OD = 2 * x.D + outpad, OH = 2 * x.H + outpad, OW = 2 * x.W + outpad;
    const size_t ov = (size_t)OD * OH * OW;
    float* out = n.alloc((size_t)x.N * ov * w.Cout);
    const int nblk = nm_stats_blocks_per_frame((int)ov);
    float* part = n.alloc((size_t)x.N * nblk * w.Cout * 2);
    float* scale = n.alloc((size_t)x.N * w.Cout); float* shift = n.alloc((size_t)x.N * w.Cout);
    if (n.live()) {
        n.run(nm_launch_convT2(x, w.w, w.bias, out, w.Cout, OD, OH, OW, n.s));
        n.run(nm_launch_gn_partials(out, x.N, (int)ov, w.Cout, part, n.s));
        n.run(nm_launch_gn_finalize(part, x.N, nblk, w.Cout, w.n.groups, (double)ov * (w.Cout / w.n.groups), w.n.gamma,
                                    w.n.beta, 1e-5f, scale, shift, n.s));
    }
    return mk(out, x.N, OD, OH, OW, w.Cout, scale, shift, LRELU);
}

// HG (vox_modules.py:78-120)
TensorRef hourglass(Net& n, const TensorRef& x0, const HourglassW& w, int Ng) {
    const int op3 = (Ng / 4) % 2, op2 = (Ng / 2) % 2, op1 = Ng % 2;
    TensorRef s1 = res(n, x0, w.s1);
    TensorRef x = res(n, pool(n, x0, w.p1), w.e1);
    TensorRef s2 = res(n, x, w.s2);
    x = res(n, pool(n, x, w.p2), w.e2);
    TensorRef s3 = res(n, x, w.s3);
    x = res(n, pool(n, x, w.p3), w.e3);
    x = res(n, x, w.d3);
    TensorRef u = up(n, x, w.u3, op3); x = add2(n, u, &s3);
    x = res(n, x, w.d2);
    u = up(n, x, w.u2, op2); x = add2(n, u, &s2);
    x = res(n, x, w.d1);
    u = up(n, x, w.u1, op1); x = add2(n, u, &s1);
    return x;
}

// Basic3DBlock(k5) on cat[occ, x1, x2, x3] (kypt_detector.py:265, kypt_detector_utils.py:4-26): only the occupancy
// channel is convolved per frame, the coordinate channels' contribution (+ bias) is the weight-only `field`.
TensorRef first_layer(Net& n, const float* occ, int N, int G, const FeatNetW& w) {
    const int Cout = w.c0.Cout;
    const size_t G3 = (size_t)G * G * G;
    float* out = n.alloc((size_t)N * G3 * Cout);
    const int nblk = nm_occ_blocks_per_frame(G);
    float* part = n.alloc((size_t)N * nblk * Cout * 2);
    float* scale = n.alloc((size_t)N * Cout); float* shift = n.alloc((size_t)N * Cout);
    if (n.live()) {
        n.run(nm_launch_conv_k5occ(occ, N, G, w.occ_w, w.field, out, Cout, w.c0.Co_pad, part, n.s));
        n.run(nm_launch_gn_finalize(part, N, nblk, Cout, w.n0.groups, (double)G3 * (Cout / w.n0.groups), w.n0.gamma, w.n0.beta,
                                    1e-5f, scale, shift, n.s));
    }
    return mk(out, N, G, G, G, Cout, scale, shift, LRELU);
}

// _build_feature_net (kypt_detector.py:264-272); `occ` is the occupancy [N][G][G][G]
void feature_net(Net& n, const float* occ, int N, int G, const FeatNetW& w, int g, float* out_buf) {
    const size_t m = n.ws.mark();
    TensorRef x = first_layer(n, occ, N, G, w);
    x = pool(n, x, w.p1);
    x = res(n, x, w.r2);
    x = pool(n, x, w.p3);
    x = hourglass(n, x, w.hg, g);
    res(n, x, w.r5, out_buf);
    n.ws.release(m);
}

// KyptToVoxNet for a batch of frames (kypt_detector.py:388-460).
//   keypoints [F][K][4]; first-frame feature in channels-last with a frame stride;
//   first_frames: occupancy of each clip's first frame (frame stride ff_stride);
//   target/tail_part/chamfer only for the training-style forward.
void decode_frames(Net& n, const float* keypoints, const float* feat_cl, int feat_frame_stride, const float* first_frames,
                   int ff_stride, int B, int T, const float* target, bool chamfer, float* recon, float* tail_part) {
    nm_ctx* c = n.c;
    const DetectorW& d = c->det;
    const int K = c->cfg.nkeypoints, G = c->cfg.grid_size, g = G / 4, F = B * T;
    const int Cc = d.adjust.Cin_pad;
    const size_t g3 = (size_t)g * g * g, G3 = (size_t)G * G * G;
    const double width_d = 2.0 * std::pow((double)c->cfg.gaussian_sigma / (double)g, 2.0);
    const size_t m0 = n.ws.mark();
    float* table = n.alloc((size_t)F * K * 3 * g);
    if (n.live()) n.run(nm_launch_gauss_table(keypoints, F * K, g, (float)width_d, table, n.s));
    const int tb = nm_tail_blocks(G);
    // whole clips per pass so that frame 0 of every clip in the pass is addressable
    const int clips_per_pass = (int)(FRAME_CHUNK / (size_t)T) > 0 ? (int)(FRAME_CHUNK / (size_t)T) : 1;
    for (int b0 = 0; b0 < B; b0 += clips_per_pass) {
        const int nb = (B - b0) < clips_per_pass ? (B - b0) : clips_per_pass;
        const int f0 = b0 * T, nf = nb * T;
        const size_t m1 = n.ws.mark();
        float* comb = n.alloc((size_t)nf * g3 * Cc);
        if (n.live())
            n.run(nm_launch_combined(table + (size_t)f0 * K * 3 * g, keypoints + (size_t)f0 * K * 4,
                                     feat_cl + (size_t)b0 * feat_frame_stride * g3 * FEAT, feat_frame_stride, nf, T, K, FEAT, g,
                                     Cc, comb, n.s));
        TensorRef x = mk(comb, nf, g, g, g, Cc);
        x = conv_gn(n, x, d.adjust, nullptr, 1, 0, LRELU);
        x = conv_gn(n, x, d.d1, &d.dn2, 1, 1, LRELU, nullptr, true);    // Upsample(x2, trilinear) fused into the staging
        x = conv_gn(n, x, d.d4, &d.dn5, 1, 1, LRELU);
        x = conv_gn(n, x, d.d8, &d.dn9, 1, 1, LRELU, nullptr, true);    // second Upsample(x2) likewise
        x = conv_gn(n, x, d.d11, &d.dn12, 1, 1, LRELU);
        if (n.live())
            n.run(nm_launch_decoder_tail(x, d.d14, first_frames + (size_t)b0 * ff_stride * G3, ff_stride, T,
                                         target ? target + (size_t)f0 * G3 : nullptr,
                                         (target && chamfer) ? keypoints + (size_t)f0 * K * 4 : nullptr, K, G,
                                         recon + (size_t)f0 * G3, tail_part ? tail_part + (size_t)f0 * tb * 3 : nullptr, n.s));
        n.ws.release(m1);
    }
    n.ws.release(m0);
}

// `after_keypoints` (optional) is called once the keypoints kernel has been enqueued, in the live pass only: the
// fused forward uses it to start the VRNN on the side stream while the decoder runs.
int detector_graph(nm_ctx* c, const float* vox_in, int B, int T, int affinity_on, float* keypoints, float* heatmaps,
                   float* first_feature, float* recon, float* affinity, float* losses,
                   const std::function<int()>* after_keypoints = nullptr) {
    Net n(c);
    const DetectorW& d = c->det;
    const int K = c->cfg.nkeypoints, G = c->cfg.grid_size, g = G / 4, F = B * T, N = c->cfg.nneighbor;
    const size_t g3 = (size_t)g * g * g, G3 = (size_t)G * G * G;
    n.ws.release(0);
    float* feat = n.alloc((size_t)F * g3 * FEAT);
    float* clip_head = n.alloc((size_t)B * g3 * K);
    float* heat_part = n.alloc((size_t)F * K * g * (2 * g + 2));
    float* heat_mean = n.alloc((size_t)F * K);
    float* clip_part = n.alloc((size_t)B * 5);
    const int tb = nm_tail_blocks(G);
    float* tail_part = n.alloc((size_t)F * tb * 3);
    float* aff = affinity_on ? (affinity ? affinity : n.alloc((size_t)N * K * K)) : nullptr;

    {   // spatio-temporal heat-map from the clip mean, once per clip (kypt_detector.py:311-316).  Only B frames of
        // small, latency-bound launches: issued on the side stream so that it runs beside the per-frame encoder.  Its
        // scratch stays allocated (no release) until the call ends because the two streams run concurrently.
        Net n2(c, c->stream2);
        if (n.live()) {
            n.run(nm_check_hip(hipEventRecord(c->ev_fork, n.s), "fork event"));
            n.run(nm_check_hip(hipStreamWaitEvent(c->stream2, c->ev_fork, 0), "side stream wait"));
        }
        float* in = n2.alloc((size_t)B * G3);
        float* fclip = n2.alloc((size_t)B * g3 * 2 * FEAT);
        if (n2.live()) n2.run(nm_launch_mean_t(vox_in, B, T, G3, in, n2.s));
        const size_t saved_peak = n2.ws.peak;
        n2.ws.peak = n2.ws.top;
        feature_net(n2, in, B, G, d.clip, g, fclip);
        const size_t local_peak = n2.ws.peak;                        // high-water mark of the clip net's scratch
        n2.ws.peak = saved_peak > local_peak ? saved_peak : local_peak;
        n2.ws.top = local_peak;                                      // keep that scratch out of reach of the main stream
        conv_gn(n2, mk(fclip, B, g, g, g, 2 * FEAT), d.clip_head, nullptr, 1, 0, 1.0f, clip_head);
        if (n2.live()) n2.run(nm_check_hip(hipEventRecord(c->ev_clip, c->stream2), "clip event"));
        n.run(n2.rc);
    }
    for (size_t f0 = 0; f0 < (size_t)F; f0 += FRAME_CHUNK) {   // per-frame encoder (kypt_detector.py:330-336)
        const int nf = (int)(((size_t)F - f0) < FRAME_CHUNK ? ((size_t)F - f0) : FRAME_CHUNK);
        const size_t m = n.ws.mark();
        feature_net(n, vox_in + f0 * G3, nf, G, d.frame, g, feat + f0 * g3 * FEAT);
        n.ws.release(m);
    }
    {   // heads -> heat-maps -> keypoints (kypt_detector.py:336-347)
        const size_t m = n.ws.mark();
        float* head = n.alloc((size_t)F * g3 * K);
        conv_gn(n, mk(feat, F, g, g, g, FEAT), d.head, nullptr, 1, 0, 1.0f, head);
        if (n.live()) {
            n.run(nm_check_hip(hipStreamWaitEvent(n.s, c->ev_clip, 0), "join clip net"));
            n.run(nm_launch_heatmap(head, clip_head, d.prop, F, T, K, g, heatmaps, heat_part, n.s));
            n.run(nm_launch_keypoints(heat_part, F, K, g, keypoints, heat_mean, n.s));
            if (after_keypoints && n.ok()) n.run((*after_keypoints)());
        }
        n.ws.release(m);
    }
    if (n.live()) {   // first_feature output: frame 0 of every clip, NCDHW
        TensorRef ff = mk(feat, B, g, g, g, FEAT);
        n.run(nm_launch_cl_to_ncdhw_strided(ff, T, first_feature, n.s));
        if (affinity_on) n.run(nm_launch_affinity(d.affinity_params, N, K, aff, n.s));
    }
    decode_frames(n, keypoints, feat, T, vox_in, T, B, T, vox_in, c->cfg.vol_fit_chamfer != 0, recon, tail_part);
    if (n.live()) {
        n.run(nm_launch_clip_loss(keypoints, aff, B, T, K, N, c->cfg.sep_sigma, clip_part, n.s));
        n.run(nm_launch_loss_finalize(tail_part, tb, B, T, K, N, G, heat_mean, clip_part, aff, c->cfg.vol_fit_chamfer,
                                      c->cfg.use_graph_traj, losses, n.s));
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
    return nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
}

}  // namespace

int nm_net_set_weights(nm_ctx* c, const std::map<std::string, std::pair<const float*, int64_t>>& sd) {
    int rc = nm_check_hip(hipStreamSynchronize(c->stream), "set_weights: sync");
    if (rc) return rc;
    for (void* p : c->owned) (void)hipFree(p);
    c->owned.clear();
    c->has_weights = false;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, N = c->cfg.nneighbor, S4 = K * 4;
    Loader L{c, sd};
    DetectorW& d = c->det;
    const std::string v = "kypt_detector.vox_to_kypt", k2v = "kypt_detector.kypt_to_vox";
    const std::string dec = k2v + ".decode_voxel_from_combined_representation";
    d.affinity_params = L.copy("kypt_detector.affinity_params", (int64_t)N * K * (K - 1));
    d.frame = L.featnet(v + ".extract_features", FEAT);
    d.head = L.conv(v + ".extract_heatmaps_from_features.0", K, FEAT, 1);
    d.clip = L.featnet(v + ".extract_spatio_temporal_features", 2 * FEAT);
    d.clip_head = L.conv(v + ".extract_spatio_temporal_heatmaps_from_features.0", K, 2 * FEAT, 1);
    {
        const float* pw = L.get(v + ".propagate_heatmaps.0.weight", 2);
        const float* pb = L.get(v + ".propagate_heatmaps.0.bias", 1);
        d.prop = nm_ctx_weight_alloc(c, 3);
        if (pw && pb && d.prop) hipLaunchKernelGGL(pack_small_kernel, dim3(1), dim3(64), 0, c->stream, pw, 2, pb, 1, d.prop);
    }
    d.adjust = L.conv(k2v + ".adjust_combined_representation.0", FEAT, FEAT + 2 * K + 3, 1);
    d.d1 = L.conv(dec + ".1", FEAT / 2, FEAT, 3); d.dn2 = L.norm(dec + ".2", FEAT / 2);
    d.d4 = L.conv(dec + ".4", FEAT / 2, FEAT / 2, 3); d.dn5 = L.norm(dec + ".5", FEAT / 2);
    d.d8 = L.conv(dec + ".8", FEAT / 4, FEAT / 2, 3); d.dn9 = L.norm(dec + ".9", FEAT / 4);
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
    if (!r.parents) {
        if (hipMalloc(reinterpret_cast<void**>(&r.parents), K * sizeof(int32_t)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&r.order), K * sizeof(int32_t)) != hipSuccess) {
            nm_set_error("set_weights: hipMalloc(tree) failed"); return NM_ERR_HIP;
        }
    }
    if (L.rc) return L.rc;
    rc = nm_check_hip(hipGetLastError(), "set_weights: pack kernels");
    if (rc) return rc;
    rc = nm_check_hip(hipStreamSynchronize(c->stream), "set_weights: final sync");
    if (rc) return rc;
    c->has_weights = true;
    return NM_OK;
}

extern "C" {

size_t nm_workspace_bytes(nm_ctx* c, int32_t B, int32_t T) {
    if (!c || !c->has_weights || B <= 0 || T <= 0) return 0;
    c->ws.dry = true; c->ws.peak = 0;
    float dummy = 0.f; float* dp = &dummy;
    (void)detector_graph(c, dp, B, T, 1, dp, dp, dp, dp, dp, dp);
    c->ws.dry = false;
    return c->ws.peak + 4096;
}

int nm_detector_forward(nm_ctx* c, const float* vox, int32_t B, int32_t T, int32_t affinity_on, float* keypoints,
                        float* heatmaps, float* first_feature, float* recon, float* affinity, float* losses11) {
    int rc = check_ready(c, "detector_forward");
    if (rc) return rc;
    if (!vox || !keypoints || !heatmaps || !first_feature || !recon || !losses11 || B <= 0 || T <= 0) {
        nm_set_error("detector_forward: null / non-positive argument"); return NM_ERR_ARG;
    }
    return with_workspace(c, [&]() { return detector_graph(c, vox, B, T, affinity_on, keypoints, heatmaps, first_feature, recon, affinity, losses11); });
}

int nm_forward_fused(nm_ctx* c, const float* vox, int32_t B, int32_t T, int32_t affinity_on, const float* eps, int32_t S,
                     float* keypoints, float* heatmaps, float* first_feature, float* recon, float* affinity, float* losses11,
                     float* kypt_recon, float* R, float* z, float* h, float* scalars2, int32_t* best_idx) {
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
        r = nm_vrnn_encode(c, keypoints, eps, B, T, S, kypt_recon, R, z, h, scalars2, best_idx);
        c->stream = main;
        std::swap(c->ws, c->ws2);
        if (!r) r = nm_check_hip(hipEventRecord(c->ev_side, c->stream2), "side event");
        return r;
    };
    rc = with_workspace(c, [&]() { return detector_graph(c, vox, B, T, affinity_on, keypoints, heatmaps, first_feature, recon, affinity, losses11, &hook); });
    if (rc) return rc;
    return nm_check_hip(hipStreamWaitEvent(c->stream, c->ev_side, 0), "join side stream");
}

int nm_decode_from_keypoints(nm_ctx* c, const float* keypoints, const float* first_feature, const float* first_frame,
                             int32_t B, int32_t Tg, float* gen) {
    int rc = check_ready(c, "decode_from_keypoints");
    if (rc) return rc;
    if (!keypoints || !first_feature || !first_frame || !gen || B <= 0 || Tg <= 0) {
        nm_set_error("decode_from_keypoints: null / non-positive argument"); return NM_ERR_ARG;
    }
    return with_workspace(c, [&]() { return decode_graph(c, keypoints, first_feature, first_frame, B, Tg, gen); });
}

int nm_voxelize_clip(nm_ctx* c, const double* points, int32_t T, int64_t N, double scale, float* vox, int32_t* idx_out) {
    if (!c || !points || !vox || T <= 0 || N <= 0) { nm_set_error("voxelize_clip: bad argument"); return NM_ERR_ARG; }
    int rc = nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
    if (rc) return rc;
    if ((rc = nm_ctx_reserve(c, 256 * 6 * sizeof(double) + 4096))) return rc;
    c->ws.release(0);
    double* part = static_cast<double*>(c->ws.alloc_bytes(256 * 6 * sizeof(double)));
    return nm_launch_voxelize(points, T, (size_t)N, c->cfg.grid_size, scale, part, vox, idx_out, c->stream);
}

int nm_get_affinity(nm_ctx* c, float* affinity) {
    int rc = check_ready(c, "get_affinity");
    if (rc) return rc;
    if (!affinity) { nm_set_error("get_affinity: null output"); return NM_ERR_ARG; }
    return nm_launch_affinity(c->det.affinity_params, c->cfg.nneighbor, c->cfg.nkeypoints, affinity, c->stream);
}

}  // extern "C"
