// C ABI (include/nm355.h): context management, error reporting and the op-level entry
// points used by the unit parity tests.  The network-level entry points live in
// nm_net.hip / nm_vrnn.hip.
#include "nm_ctx.h"
#include <mutex>
#include "nm_grad.h"
#include "nm_up2c.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <new>
#include <stdexcept>
#include <vector>

static thread_local char g_err[512] = "";

thread_local NmLaunchState* nm_tls_ls = nullptr;
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
NmLaunchState::NmLaunchState()
    : supertile(env_int("NM355_SUPERTILE", 1)),     // 0: linear brick order (diagnostic)
      small16(env_int("NM355_SMALL16", 1)),         // 0: small volumes on the fp32 MFMA core (diagnostic)
      ksplit(env_int("NM355_KSPLIT", 1)),           // 0: no tap split on the tiny volumes (diagnostic)
      occ16(env_int("NM355_OCC16", 1)),             // 0: first layer on the fp32 MFMA kernel (diagnostic)
      pool16(env_int("NM355_POOL16", 1)),           // 0: pool convs on the fp32 kernel (diagnostic)
      f16p2(env_int("NM355_F16P2", 1)),             // 0: Cout % 64 == 0 layers stay on conv_f16s (diagnostic)
      // conv_f16p use: 0 never (conv_f16s everywhere), 1 every eligible layer, 2 (default) only Cout == 32 layers - with more cout
      // groups per brick it re-stages the input per group and measures a little slower than conv_f16s (A/B in one gpurun call)
      f16p(env_int("NM355_F16P", 2)),
      pool_q(env_int("NM355_POOL_Q", 1)),           // 0: conv_pool_f16s_kernel (conditional loads, one memory round trip per tap) instead of conv_pool_f16q_kernel
      occ_flags(env_int("NM355_OCC_FLAGS", 2)),     // sparse first layer: 0 every brick tests its own halo; 1 per-brick occupancy flags as a pre-filter; 2 (default) also one workgroup per x-row of bricks
      gnb_apply4(env_int("NM355_GNB_APPLY4", 1)),   // 0: gnb_apply_kernel (one 16-byte item per iteration) for every channel count
      defer_sums(env_int("NM355_DEFER_SUMS", 1)),      // 0: the per-layer gamma / beta / bias gradient sums are launched inside each GroupNorm backward (A/B)
      wgrad_async(env_int("NM355_WGRAD_ASYNC", 1)),    // 0: the weight gradients of the training backward stay in the main stream's chain (A/B)
      wgrad_wgs(env_int("NM355_WGRAD_WGS", 0)),       // workgroups of a split-fp16 weight-gradient launch (0: 224 beside the main stream's walk, else 256; plan_wgrad)
      wgrad_tr(env_int("NM355_WGRAD_TR", 1)),       // 0: wgrad16_kernel (VALU transposition) instead of wgrad16t_kernel
      wgrad_u(env_int("NM355_WGRAD_U", 1)),         // 0: wgrad16t_kernel (conditional staging loads) instead of wgrad16u_kernel
      tail_rank1(env_int("NM355_TAIL_RANK1", 1)),   // 0: the decoder tail's backward materialises its [F][G^3][32] gradient (A/B)
      wgrad_z(env_int("NM355_WGRAD_Z", 1)),         // 0: bricks in any order with their full halo (wgrad16u_kernel) instead of z-columns with a plane ring
      up2c(env_int("NM355_UP2C", 1)),               // 0: fused-upsample layers stay on conv_f16s (diagnostic / A-B)
      up2c_diag(env_int("NM355_UP2C_DIAG", 0)),
      vrnn_mid(env_int("NM355_VRNN_MID", 1)),          // 0: prior steps as six launches instead of three (A/B)
      vrnn_postmid(env_int("NM355_VRNN_POSTMID", 0)),   // 1: posterior steps of encode as three launches (vrnn_post_mid_kernel): measured slower, see there
      vrnn_nb(env_int("NM355_VRNN_NB", 2)),             // batch rows per wavefront pass of the VRNN row kernels (4 / 8: the slow instantiations, A/B)
      vrnn_gemm(env_int("NM355_VRNN_GEMM", 1)),    // 0: one wavefront per output row at every batch size (A/B)
      vrnn_graph(env_int("NM355_VRNN_GRAPH", 1)),       // 0: rollouts enqueue their launches one by one instead of replaying a captured graph (A/B)
      sparse_first(env_int("NM355_SPARSE_FIRST", 1)), // 0: the first layer writes its dense output in inference too (A/B)
      gn_diag(env_int("NM355_GN_DIAG", 0)),                 // 1: GroupNorm statistics recomputed from the stored tensor in fp64 (diagnostic)
      lazy_res(env_int("NM355_LAZY_RES", 1)),         // 0: every residual sum is materialised by apply2 (A/B)
      adjust_split(env_int("NM355_ADJUST_SPLIT", 1)), // 0: the decoder's first 1x1 conv runs over the materialised 184-channel tensor in inference too (A/B)
      hg_core(env_int("NM355_HG_CORE", 1)),           // 0: the two lowest hourglass levels as separate launches in inference too (A/B)
      f16p_dma(env_int("NM355_F16P_DMA", 0)),         // 1: conv_f16p2's producers copy the weights by LDS-DMA instead of through registers (A/B)
      clip_occ_mfma(env_int("NM355_CLIP_OCC_MFMA", 1)),  // 0: the clip-mean net's first-layer weight gradient as the dense all-frames kernel (A/B)
      vrnn_chain(env_int("NM355_VRNN_CHAIN", 1)),        // 0: the prior steps of a rollout as three launches per step instead of one persistent launch (A/B)
      wgrad_k2f16(env_int("NM355_WGRAD_K2F16", 1)),     // 0: the k2 s2 weight gradients on the fp32-MFMA kernel in every conv mode (A/B)
      convt_f16(env_int("NM355_CONVT_F16", 1)),         // 0: the transposed convs (and the pool convs' data gradient) on the fp32-MFMA kernel in every conv mode (A/B)
      k5_two(env_int("NM355_K5_TWO", 0)),               // 1: the two halves of the first layer's weight gradient on two streams (A/B: 44.6 vs 44.7 / 73.3 vs 73.6 ms per step - the tail of one step already overlaps the head of the next)
      clip_late(env_int("NM355_CLIP_LATE", 1)),         // 0: the clip-mean net is enqueued before the per-frame encoder instead of behind its first chunk(s) (A/B)
      gnb_u8(env_int("NM355_GNB_U8", 1)),               // 0: four instead of eight items in flight in the bfloat16 GroupNorm-backward apply (A/B)
      adj_zwalk(env_int("NM355_ADJ_ZWALK", 1)),        // 0: the trilinear upsample's adjoint on the 2 x 4 x 8 brick kernel (6 x 10 x 18 fine tile) instead of the z-walking one (A/B)
      f16q2(env_int("NM355_F16Q2", 1)),                 // 0: the one-product modes (3 / 4) keep conv_f16p2<SINGLE> instead of their own kernel conv_f16q2 (A/B)
      vrnn_post_chain(env_int("NM355_VRNN_POST_CHAIN", 2))   // 0: the posterior steps of encode as six launches each; 1: one persistent launch (vrnn_post_chain_kernel) for a stand-alone encode only; 2 (default since round 6): inside nm_forward_fused too - 148 instead of 238 launches per forward step for +0.03 ... 0.13 ms (profiles/r06_forward_ab.txt)
      , f16r(env_int("NM355_F16R", 1))                      // 0: the one-product modes keep conv_f16p<SINGLE> on the 32-output-channel layers instead of conv_f16r (resident weights; A/B)
      , up2_mat(env_int("NM355_UP2_MAT", 1))                // 0: the one-product training forward keeps the fused-upsample staging on its 32-output layer instead of materialising the upsampled input for conv_f16r (A/B)
      , conv_wgs(env_int("NM355_CONV_WGS", 0))              // > 0: the persistent producer / consumer convs launch at most this many workgroups (co-residency A/B: CUs left free for the other queues)
      , up2c_x16(env_int("NM355_UP2C_X16", 1))              // 0: conv_up2c stays on v_mfma_f32_32x32x16_f16 (A/B; 1: conv_up2c_x16_kernel, v_mfma_f32_16x16x32_f16, fp32-storage modes)
      , up2c_all(env_int("NM355_UP2C_ALL", 0))              // 1: the decoder's FIRST fused-upsample layer (128 -> 64 @16^3 -> 32^3) on the composite-weight kernel too (A/B)
      , f16p_late(env_int("NM355_F16P_LATE", 0))            // 1: conv_f16p's LATE producer schedule (round 6: -0.05 ms per forward) - NOT the default: on some boxes one evaluation in ~1500 came back with non-finite decoder outputs under it (0 in 18 000 with the round-2 schedule, 12 in 18 000 with LATE, same box: profiles/r06_not_shipped_ab.txt)
      , p2_defer(env_int("NM355_P2_DEFER", 0))              // 1: conv_f16p2 with one accumulator per tile and the epilogue deferred into the next brick's first step (A/B: slower)
      , fast_decode(env_int("NM355_FAST_DECODE", 1))        // 0: the persistent convs decode a brick's position with integer divisions instead of host-made reciprocal multiplications (A/B)
{ store16_min = env_int("NM355_STORE16_MIN", 32768); chain_spin = env_int("NM355_CHAIN_SPIN", 1 << 20); chain_drop = env_int("NM355_CHAIN_DROP_WG", 0); chain_stat_delay = env_int("NM355_CHAIN_STAT_DELAY", 0);
  chain_wgpoll = env_int("NM355_CHAIN_WGPOLL", 1);      // 0: every wave of the cross-XCD rollout chain polls for itself (round 5; A/B)
  chain_xcd_nogo = env_int("NM355_CHAIN_XCD_NOGO", 0);  // test hook: the one-XCD chain never starts (its fallback must do the work)
  chain_xcd = env_int("NM355_CHAIN_XCD", 1); }           // 0: no one-XCD launch in front of the cross-XCD rollout chain (A/B); 1: for B = 1; 2: for B <= 8
NmLaunchState& nm_ls() {
    static thread_local NmLaunchState outside;       // launchers reached outside an ABI call (none in the product path)
    return nm_tls_ls ? *nm_tls_ls : outside;
}

void nm_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int nm_abi_catch(const char* fn) noexcept {
    try { throw; }
    catch (const std::bad_alloc&) { nm_set_error("%s: out of host memory (std::bad_alloc)", fn); }
    catch (const std::exception& e) { nm_set_error("%s: internal error: %s", fn, e.what()); }
    catch (...) { nm_set_error("%s: internal error (unknown C++ exception)", fn); }
    return NM_ERR_INTERNAL;
}

int nm_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return NM_OK;
    nm_set_error("%s: %s", what, hipGetErrorString(e));
    return NM_ERR_HIP;
}

int nm_ctx_reserve(nm_ctx* ctx, size_t bytes) {
    if (ctx->ws.cap >= bytes) return NM_OK;
    int rc = nm_check_hip(hipStreamSynchronize(ctx->stream), "reserve: stream sync");
    if (rc) return rc;
    if (ctx->ws.base) { (void)hipFree(ctx->ws.base); ctx->ws.base = nullptr; ctx->ws.cap = 0; }
    size_t want = bytes + (bytes / 16 < ((size_t)64 << 20) ? bytes / 16 : ((size_t)64 << 20)) + (1 << 20);      // growth slack, at most 64 MB (1/16 of a 10 GB training arena was 0.66 GB)
    rc = nm_check_hip(hipMalloc(reinterpret_cast<void**>(&ctx->ws.base), want), "reserve: hipMalloc workspace");
    if (rc) return rc;
    ctx->ws.cap = want;
    return NM_OK;
}

// Weight buffers are handed out in call order.  A repeated set_weights (every optimizer step) asks for the same sizes in the same
// order and gets the same buffers back - no hipFree / hipMalloc, hence no synchronisation: the pack kernels that overwrite them are
// stream-ordered behind the previous step's readers.  On the first mismatch the tail of the list is released (after a sync).
float* nm_ctx_weight_alloc(nm_ctx* ctx, size_t floats) {
    const size_t bytes = (floats ? floats : 1) * sizeof(float);
    if (ctx->owned_cursor < ctx->owned.size()) {
        if (ctx->owned_bytes[ctx->owned_cursor] == bytes) return static_cast<float*>(ctx->owned[ctx->owned_cursor++]);
        (void)hipDeviceSynchronize();
        for (size_t i = ctx->owned_cursor; i < ctx->owned.size(); ++i) (void)hipFree(ctx->owned[i]);
        ctx->owned.resize(ctx->owned_cursor); ctx->owned_bytes.resize(ctx->owned_cursor);
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    ctx->weights_epoch++;
    ctx->owned.push_back(p); ctx->owned_bytes.push_back(bytes); ctx->owned_cursor = ctx->owned.size();
    return static_cast<float*>(p);
}

extern "C" {

int nm_abi_version(void) { return NM_ABI_VERSION; }
const char* nm_last_error(void) { return g_err; }

// the weight-gradient stream; NM355_WGRAD_PRIO=1: at the lowest priority the device offers (A/B)
static hipError_t create_wgrad_stream(hipStream_t* s) {
    const char* e = getenv("NM355_WGRAD_PRIO");
    if (e && atoi(e) != 0) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            return hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// The two side streams (clip-mean net / VRNN beside the frame stack; weight gradients) are shared by all contexts of a process on a
// device.  HIP maps streams onto a few hardware queues round-robin (4 by default): a SECOND context with streams of its own got queues
// that the runtime arbitrates differently and its bf16 training step ran at 48 ms instead of 44 (same kernels, same durations when
// alone; 53 ms with GPU_MAX_HW_QUEUES=8, 44 with 3: tools/time_train_steps.py) - bench.py's train_bf16, measured in a second context,
// showed it.  Sharing adds ordering between contexts used concurrently from different threads, never a missing dependency (each
// context keeps its own events).  NM355_OWN_STREAMS=1 / NM355_WGRAD_PRIO=1: streams per context as before.
#define NM_MAX_DEVICES 16
static std::mutex g_side_mu;
static hipStream_t g_side[NM_MAX_DEVICES][2];
static int g_side_refs[NM_MAX_DEVICES];
static bool own_streams() { const char* e = getenv("NM355_OWN_STREAMS"); const char* p = getenv("NM355_WGRAD_PRIO"); return (e && atoi(e) != 0) || (p && atoi(p) != 0); }
static hipError_t acquire_side_streams(int dev, hipStream_t* s2, hipStream_t* s3, bool* shared) {
    if (own_streams() || dev < 0 || dev >= NM_MAX_DEVICES) {
        *shared = false;
        hipError_t e = hipStreamCreateWithFlags(s2, hipStreamNonBlocking);
        return e != hipSuccess ? e : create_wgrad_stream(s3);
    }
    std::lock_guard<std::mutex> lk(g_side_mu);
    if (g_side_refs[dev] == 0) {
        hipError_t e = hipStreamCreateWithFlags(&g_side[dev][0], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g_side[dev][1], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    ++g_side_refs[dev];
    *s2 = g_side[dev][0]; *s3 = g_side[dev][1]; *shared = true;
    return hipSuccess;
}
static void release_side_streams(int dev, hipStream_t s2, hipStream_t s3, bool shared) {
    if (!shared) { if (s2) (void)hipStreamDestroy(s2); if (s3) (void)hipStreamDestroy(s3); return; }
    std::lock_guard<std::mutex> lk(g_side_mu);
    if (--g_side_refs[dev] == 0) {
        (void)hipStreamDestroy(g_side[dev][0]); (void)hipStreamDestroy(g_side[dev][1]);
        g_side[dev][0] = g_side[dev][1] = nullptr;
    }
}

// Test hook of the exception barrier (tests/test_abi_cpu.py; needs no device): throws inside an entry point exactly as a failing
// std::vector / std::string / operator new of the real entry points would.  kind 0: std::bad_alloc from a real allocation request
// (SIZE_MAX / 2 bytes), 1: std::length_error from std::vector::reserve, 2: a non-std exception.  Must return NM_ERR_INTERNAL.
int nm_abi_selftest_throw(int32_t kind) try {
    if (kind == 0) { volatile size_t n = ~(size_t)0 / 2; char* p = static_cast<char*>(::operator new(n)); p[0] = 1; ::operator delete(p); }
    else if (kind == 1) { std::vector<double> v; v.reserve(v.max_size() + 1); }
    else if (kind == 2) throw 42;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_abi_selftest_throw"); }

int nm_ctx_create(nm_ctx** out, const nm_config* cfg) try {
    if (!out || !cfg) { nm_set_error("ctx_create: null argument"); return NM_ERR_ARG; }
    if (cfg->grid_size < 32 || cfg->grid_size % 8) { nm_set_error("ctx_create: grid_size %d unsupported", cfg->grid_size); return NM_ERR_UNSUPPORTED; }
    // (any keypoint count in [2, 32]: the reference's dataset configs use 12 / 22 / 24 / 28, dataset/config.py:97,124, train.py:60; the
    //  heat-map heads run zero-padded to a multiple of 8 channels inside the library, everything else takes K as it is)
    if (cfg->nkeypoints <= 1 || cfg->nkeypoints > 32) { nm_set_error("ctx_create: nkeypoints %d unsupported (2 .. 32)", cfg->nkeypoints); return NM_ERR_UNSUPPORTED; }
    int ndev = 0;
    int rc = nm_check_hip(hipGetDeviceCount(&ndev), "ctx_create: hipGetDeviceCount");
    if (rc) return rc;
    if (cfg->device < 0 || cfg->device >= ndev) { nm_set_error("ctx_create: device %d of %d", cfg->device, ndev); return NM_ERR_ARG; }
    rc = nm_check_hip(hipSetDevice(cfg->device), "ctx_create: hipSetDevice");
    if (rc) return rc;
    hipDeviceProp_t prop;
    rc = nm_check_hip(hipGetDeviceProperties(&prop, cfg->device), "ctx_create: props");
    if (rc) return rc;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        nm_set_error("ctx_create: device arch '%s' is not gfx950 (this library is MI355X-only)", prop.gcnArchName);
        return NM_ERR_UNSUPPORTED;
    }
    nm_ctx* c = new nm_ctx();
    c->cfg = *cfg;
    if (hipMalloc(reinterpret_cast<void**>(&c->nf_flag), sizeof(unsigned)) != hipSuccess || hipMemset(c->nf_flag, 0, sizeof(unsigned)) != hipSuccess) {
        nm_set_error("ctx_create: could not allocate the status word");
        delete c;
        return NM_ERR_HIP;
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&c->nf_host), 4 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
        nm_set_error("ctx_create: could not allocate the pinned status slots");
        delete c;
        return NM_ERR_HIP;
    }
    for (int i = 0; i < 4; ++i) c->nf_host[i] = 0;
    { const char* e = getenv("NM355_RANGE_CHECK"); c->range_check = e ? atoi(e) : 1; }
    if (hipMalloc(reinterpret_cast<void**>(&c->vrnn_cnt), 256 * sizeof(int32_t)) != hipSuccess || hipMemset(c->vrnn_cnt, 0, 256 * sizeof(int32_t)) != hipSuccess) {
        nm_set_error("ctx_create: could not allocate the VRNN arrival counters");
        delete c;
        return NM_ERR_HIP;
    }
    if (acquire_side_streams(cfg->device, &c->stream2, &c->stream3, &c->side_shared) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_w[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_w[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_w[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_dy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_wjoin, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_k5, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_clip, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_kp, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_nf[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_nf[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_nf[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_nf[3], hipEventDisableTiming) != hipSuccess) {
        nm_set_error("ctx_create: could not create the side stream / events");
        if (c->stream2 || c->stream3) release_side_streams(cfg->device, c->stream2, c->stream3, c->side_shared);
        delete c;
        return NM_ERR_HIP;
    }
    *out = c;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_create"); }

int nm_ctx_destroy(nm_ctx* ctx) try {
    if (!ctx) return NM_OK;
    (void)hipSetDevice(ctx->cfg.device);
    (void)hipDeviceSynchronize();
    for (void* p : ctx->owned) (void)hipFree(p);
    if (ctx->ws.base) (void)hipFree(ctx->ws.base);
    if (ctx->ws2.base) (void)hipFree(ctx->ws2.base);
    if (ctx->vrnn_cnt) (void)hipFree(ctx->vrnn_cnt);
    if (ctx->ws_t.base) (void)hipFree(ctx->ws_t.base);
    if (ctx->copy_table) (void)hipFree(ctx->copy_table);
    if (ctx->pack_table) (void)hipFree(ctx->pack_table);
    if (ctx->nf_flag) (void)hipFree(ctx->nf_flag);
    if (ctx->nf_host) (void)hipHostFree(ctx->nf_host);
    for (hipEvent_t e : ctx->ev_nf) if (e) (void)hipEventDestroy(e);
    if (ctx->vrnn.parents) (void)hipFree(ctx->vrnn.parents);
    nm_vrnn_free_graphs(ctx);
    nm_net_free_tape(ctx);
    nm_vrnn_free_tape(ctx);
    if (ctx->stream2 || ctx->stream3) release_side_streams(ctx->cfg.device, ctx->stream2, ctx->stream3, ctx->side_shared);
    if (ctx->wside) (void)hipFree(ctx->wside);
    for (hipEvent_t e : {ctx->ev_fork, ctx->ev_clip, ctx->ev_kp, ctx->ev_side, ctx->ev_w[0], ctx->ev_w[1], ctx->ev_w[2], ctx->ev_dy, ctx->ev_wjoin, ctx->ev_k5}) if (e) (void)hipEventDestroy(e);
    for (const NmProfRec& r : ctx->ls.prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : ctx->ls.event_pool) (void)hipEventDestroy(e);
    delete ctx;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_destroy"); }

int nm_ctx_set_stream(nm_ctx* ctx, void* hip_stream) try { NmScope nm_scope_(ctx);
    if (!ctx) { nm_set_error("set_stream: null ctx"); return NM_ERR_ARG; }
    hipStream_t ns = static_cast<hipStream_t>(hip_stream);
    if (ns != ctx->stream && ctx->stream_bound) {
        // ctx-owned state (packed weights, workspaces, tapes) may still be in use by work queued on the old stream (and on the
        // side stream it forked): the new stream starts behind it
        (void)hipSetDevice(ctx->cfg.device);
        int rc = nm_check_hip(hipEventRecord(ctx->ev_fork, ctx->stream), "set_stream: record on the old stream");
        if (!rc) rc = nm_check_hip(hipStreamWaitEvent(ns, ctx->ev_fork, 0), "set_stream: new stream waits for the old one");
        if (!rc) rc = nm_check_hip(hipEventRecord(ctx->ev_fork, ctx->stream2), "set_stream: record on the side stream");
        if (!rc) rc = nm_check_hip(hipStreamWaitEvent(ns, ctx->ev_fork, 0), "set_stream: new stream waits for the side stream");
        if (rc) return rc;
    }
    ctx->stream = ns;
    ctx->stream_bound = true;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_set_stream"); }

int nm_ctx_check_nonfinite(nm_ctx* ctx) try { NmScope nm_scope_(ctx);
    if (!ctx) { nm_set_error("check_nonfinite: null ctx"); return NM_ERR_ARG; }
    (void)hipSetDevice(ctx->cfg.device);
    unsigned v = 0;
    int rc = nm_check_hip(hipStreamSynchronize(ctx->stream), "check_nonfinite: sync");
    if (!rc) rc = nm_check_hip(hipMemcpy(&v, ctx->nf_flag, sizeof(v), hipMemcpyDeviceToHost), "check_nonfinite: read");
    if (rc) return rc;
    for (int i = 0; i < 4; ++i) {          // the stream is drained: every pending status copy has landed; this report consumes them
        if (ctx->nf_busy[i] && ctx->nf_host[i]) v |= ctx->nf_host[i];
        ctx->nf_busy[i] = false; ctx->nf_host[i] = 0;
    }
    if (!v) return NM_OK;
    (void)hipMemset(ctx->nf_flag, 0, sizeof(unsigned));
    if (v & 2u) {
        // (as in nm_nf_poll: this context stops using the persistent chain; its captured graphs hold that launch and are dropped)
        (void)hipDeviceSynchronize();
        ctx->ls.vrnn_chain = 0;
        nm_vrnn_free_graphs(ctx);
        nm_set_error("the persistent rollout kernel timed out waiting for its workgroups since the last check (its outputs are invalid); "
                     "this context now takes the launch-per-phase steps (as NM355_VRNN_CHAIN=0 does from the start) - repeat the call");
        return NM_ERR_STATE;
    }
    nm_set_error("a convolution produced non-finite values since the last check: %s", nm_conv_get_mode() != 0
                 ? "in the split-fp16 conv mode an activation beyond the fp16 range (|x| >= 65520) or a non-finite input does that - "
                   "set conv mode 'fp32' (exact fp32 MFMA, no range limit) and run again"
                 : "the input or the weights hold inf / NaN (exact fp32 mode has no range limit of its own)");
    return NM_ERR_RANGE;
} catch (...) { return nm_abi_catch("nm_ctx_check_nonfinite"); }

int nm_ctx_set_weights(nm_ctx* ctx, const nm_named_tensor* tensors, int32_t count) try { NmScope nm_scope_(ctx);
    if (!ctx || !tensors || count <= 0) { nm_set_error("set_weights: bad arguments"); return NM_ERR_ARG; }
    std::map<std::string, std::pair<const float*, int64_t>> sd;
    for (int i = 0; i < count; ++i) {
        if (!tensors[i].name || !tensors[i].data) { nm_set_error("set_weights: entry %d is null", i); return NM_ERR_ARG; }
        sd[tensors[i].name] = std::make_pair(tensors[i].data, tensors[i].numel);
    }
    (void)hipSetDevice(ctx->cfg.device);
    return nm_net_set_weights(ctx, sd);
} catch (...) { return nm_abi_catch("nm_ctx_set_weights"); }

int nm_ctx_set_training(nm_ctx* ctx, int32_t on) try { NmScope nm_scope_(ctx);
    if (!ctx) { nm_set_error("set_training: null ctx"); return NM_ERR_ARG; }
    if (ctx->training != (on != 0)) { ctx->training = on != 0; ctx->has_weights = false; }     // the next call needs nm_ctx_set_weights again
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_ctx_set_training"); }

int nm_set_conv_mode(nm_ctx* ctx, int32_t mode) try {
    if (!ctx || mode < 0 || mode > 4) { nm_set_error("set_conv_mode: mode must be 0 (fp32 MFMA), 1 (split-fp16 MFMA), 2 (split-fp16, conv_f16p wherever eligible), 3 (fp16 products, fp32 accumulation) or 4 (mode 3 + bfloat16 storage of the training path's tensors)"); return NM_ERR_ARG; }
    NmScope sc(ctx);
    nm_conv_set_mode(mode);
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_set_conv_mode"); }

int nm_get_conv_mode(nm_ctx* ctx) try { NmScope sc(ctx); return nm_conv_get_mode(); } catch (...) { return nm_abi_catch("nm_get_conv_mode"); }

int nm_op_set_storage16(nm_ctx* ctx, int32_t in_h, int32_t out_h) try {
    if (!ctx) { nm_set_error("op_set_storage16: null ctx"); return NM_ERR_ARG; }
    ctx->ls.op_in_h = in_h ? 1 : 0; ctx->ls.op_out_h = out_h ? 1 : 0;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_op_set_storage16"); }

int nm_prof_enable(nm_ctx* ctx, int32_t on) try { NmScope nm_scope_(ctx);
    if (!ctx) { nm_set_error("prof_enable: null ctx"); return NM_ERR_ARG; }
    nm_conv_prof_enable(on, ctx->stream);         // 1: launches on the context's main stream, 2: side-stream launches too
    if (on) nm_conv_prof_reset();
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_prof_enable"); }

int nm_prof_read(nm_ctx* ctx, int32_t variant, double* ms_total, double* flops_total, int64_t* launches) try { NmScope nm_scope_(ctx);
    if (!ctx || !ms_total || !flops_total || !launches || variant < 0 || variant > 15) { nm_set_error("prof_read: bad argument"); return NM_ERR_ARG; }
    long long n = 0;
    int rc = nm_conv_prof_collect(variant, ms_total, flops_total, &n);
    if (rc) { nm_set_error("prof_read: event query failed"); return rc; }
    *launches = n;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_prof_read"); }

const char* nm_prof_kernel_name(int32_t variant) {
    static const char* names[16] = {"conv_mfma_kernel<1,1>", "conv_mfma_kernel<1,2>", "conv_mfma_kernel<2,1>", "conv_mfma_kernel<2,2>",
                                   "conv_k5occ_kernel", "conv_f16s_kernel<2,1>", "conv_f16s_kernel<2,2>", "conv_f16p_kernel",
                                   "conv_pool_f16s_kernel", "conv_f16p2_kernel", "conv_f16s_kernel<2,1,3,up2>", "conv_f16s_kernel<2,2,3,up2>",
                                   "conv_up2c_kernel", "wgrad16_kernel", "conv_f16q2_kernel", "conv_f16r_kernel"};
    return (variant >= 0 && variant < 16) ? names[variant] : "";
}

int nm_host_linspace(int32_t n, float* out) try {
    if (n < 2 || !out) { nm_set_error("linspace: bad arguments"); return NM_ERR_ARG; }
    const float step = 2.0f / (float)(n - 1);
    for (int i = 0; i < n; ++i) out[i] = i < n / 2 ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(n - 1 - i), 1.0f);
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_host_linspace"); }

// ---- op-level entry points -------------------------------------------------------------------
static TensorRef make_ref(const float* p, const float* sc, const float* sh, float slope, int N, int D, int H, int W, int C) {
    TensorRef t; t.p = p; t.scale = sc; t.shift = sh; t.slope = slope; t.N = N; t.D = D; t.H = H; t.W = W; t.C = C;
    return t;
}

// 16-bit storage at the op level (nm_op_set_storage16): element type of the tensors on the input side (in / a / y, and the gradients of
// their shape) and on the output side (out, dy and the fine-grid temporaries)
static int op_ih() { return nm_ls().op_in_h; }
static int op_oh() { return nm_ls().op_out_h; }
static TensorRef with_h(TensorRef t, int h) { t.h = h; return t; }

static int finish_gn(nm_ctx* ctx, const float* part, int N, int nblk, int C, int groups, double count,
                     const float* gamma, const float* beta, float* scale, float* shift) {
    nm_elem_set_nonfinite_flag(nullptr);           // (op-level entry points check their results themselves)
    return nm_launch_gn_finalize(part, N, nblk, C, groups, count, gamma, beta, 1e-5f, scale, shift, ctx->stream);
}

int nm_op_conv3d(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                 const float* in_scale, const float* in_shift, float in_slope, const float* weight,
                 const float* bias, int32_t Cout, int32_t ks, int32_t stride, int32_t pad, float* out,
                 int32_t gn_groups, const float* gn_gamma, const float* gn_beta, float* gn_scale, float* gn_shift,
                 int32_t up2) try { NmScope nm_scope_(ctx);
    if (!ctx || !in || !weight || !out) { nm_set_error("op_conv3d: null argument"); return NM_ERR_ARG; }
    const int Cin_pad = (Cin + 7) & ~7, Co_pad = (Cout + 31) & ~31;
    const int us = up2 ? 2 : 1;
    ConvGeom g; g.ks = ks; g.stride = stride; g.pad = pad; g.up2 = up2 ? 1 : 0;
    g.OD = (us * D + 2 * pad - ks) / stride + 1; g.OH = (us * H + 2 * pad - ks) / stride + 1; g.OW = (us * W + 2 * pad - ks) / stride + 1;
    g.Cout = Cout; g.Co_pad = Co_pad;
    const bool want_up2c = up2 && nm_up2c_eligible(D, H, W, Cin, Cout, ks, stride, pad);
    const size_t wflu = want_up2c ? nm_up2c_weight_floats(Cin, Co_pad) : 0;
    if (want_up2c) g.up2c = reinterpret_cast<const void*>((uintptr_t)256);      // (placeholder while sizing; the real pointer follows)
    const int nblk = nm_conv_blocks_per_frame(g, Cin_pad);
    const size_t wfl = nm_packed_weight_floats(ks, Cin_pad, Co_pad);
    const bool want16 = Cin % 8 == 0 && Cin >= 16;
    const size_t wfl16 = want16 ? nm_packed_weight_floats(ks, (Cin + 15) & ~15, Co_pad) : 0;
    const size_t pfl = (size_t)N * nblk * Cout * 2;
    int rc = nm_ctx_reserve(ctx, (wfl + wfl16 + wflu + pfl) * sizeof(float) + 8192);
    if (rc) return rc;
    ctx->ws.release(0);
    float* wp = ctx->ws.f(wfl);
    float* wp16 = want16 ? ctx->ws.f(wfl16) : nullptr;
    float* wup = want_up2c ? ctx->ws.f(wflu) : nullptr;
    float* part = gn_groups > 0 ? ctx->ws.f(pfl) : nullptr;
    rc = nm_launch_pack_conv_weight(weight, Cout, Cin, ks, wp, Cin_pad, Co_pad, ctx->stream);
    if (rc) return rc;
    if (wp16 && (rc = nm_launch_pack_conv_weight16(weight, Cout, Cin, ks, wp16, Co_pad, ctx->stream))) return rc;
    if (wup && (rc = nm_launch_up2c_compose(weight, Cout, Cin, Co_pad, wup, ctx->stream))) return rc;
    g.up2c = wup;
    TensorRef t = with_h(make_ref(in, in_scale, in_shift, in_slope, N, D, H, W, Cin_pad), op_ih());
    rc = nm_launch_conv(t, wp, bias, out, g, part, ctx->stream, Cin, wp16, op_oh());
    if (rc) return rc;
    int nblk_used = nblk;
    if (nm_conv_get_mode() != 0 && !op_oh() && !op_ih()) {
        // op-level calls are synchronous about the fp16 range: operands beyond it make the split product inf / NaN where fp32 is
        // finite - scan the result and, if that happened, run the launch again on the exact fp32 MFMA path
        const int mode = nm_conv_get_mode();
        unsigned v = 0;
        if ((rc = nm_launch_nonfinite_scan(out, (size_t)N * g.OD * g.OH * g.OW * Cout, ctx->nf_flag, ctx->stream))) return rc;
        if ((rc = nm_check_hip(hipStreamSynchronize(ctx->stream), "op_conv3d: sync"))) return rc;
        if ((rc = nm_check_hip(hipMemcpy(&v, ctx->nf_flag, sizeof(v), hipMemcpyDeviceToHost), "op_conv3d: status"))) return rc;
        if (v) {
            (void)hipMemset(ctx->nf_flag, 0, sizeof(unsigned));
            nm_conv_set_mode(0);
            ConvGeom g32 = g; g32.up2c = nullptr;
            nblk_used = nm_conv_blocks_per_frame(g32, Cin_pad);      // (the fp32 kernel's own partial-block layout; never more blocks than sized for)
            rc = nm_launch_conv(t, wp, bias, out, g32, part, ctx->stream, Cin, nullptr);
            nm_conv_set_mode(mode);
            if (rc) return rc;
        }
    }
    if (gn_groups > 0)
        rc = finish_gn(ctx, part, N, nblk_used, Cout, gn_groups, (double)g.OD * g.OH * g.OW * (Cout / gn_groups), gn_gamma, gn_beta, gn_scale, gn_shift);
    return rc;
} catch (...) { return nm_abi_catch("nm_op_conv3d"); }

int nm_op_conv5_occ(nm_ctx* ctx, const float* occ, int32_t N, int32_t G, const float* weight, const float* bias, int32_t Cout,
                    float* out, int32_t gn_groups, const float* gn_gamma, const float* gn_beta, float* gn_scale, float* gn_shift) try { NmScope nm_scope_(ctx);
    if (!ctx || !occ || !weight || !bias || !out) { nm_set_error("op_conv5_occ: null argument"); return NM_ERR_ARG; }
    const int Co_pad = (Cout + 31) & ~31;
    const size_t G3 = (size_t)G * G * G;
    const int nblk = nm_occ_blocks_per_frame(G);
    const size_t fl = nm_packed_weight_floats(5, 8, Co_pad) + (size_t)256 * Co_pad + (size_t)Cout * 125 + G3 * (9 + Cout) + (size_t)N * nblk * Cout * 2;
    int rc = nm_ctx_reserve(ctx, fl * sizeof(float) + 16384);
    if (rc) return rc;
    ctx->ws.release(0);
    float* wfull = ctx->ws.f(nm_packed_weight_floats(5, 8, Co_pad)); float* wocc = ctx->ws.f((size_t)256 * Co_pad);
    float* tmp = ctx->ws.f((size_t)Cout * 125); float* zero = ctx->ws.f(G3); float* packed_in = ctx->ws.f(G3 * 8);
    float* field = ctx->ws.f(G3 * Cout); float* part = ctx->ws.f((size_t)N * nblk * Cout * 2);
    hipStream_t s = ctx->stream;
    if ((rc = nm_launch_pack_conv_weight(weight, Cout, 4, 5, wfull, 8, Co_pad, s))) return rc;
    if ((rc = nm_launch_pack_occ_weight(weight, Cout, tmp, wocc, Co_pad, s))) return rc;
    if ((rc = nm_check_hip(hipMemsetAsync(zero, 0, G3 * sizeof(float), s), "memset"))) return rc;
    if ((rc = nm_launch_pack_input(zero, 1, 1, G, 0, packed_in, s))) return rc;
    ConvGeom g; g.ks = 5; g.stride = 1; g.pad = 2; g.OD = g.OH = g.OW = G; g.Cout = Cout; g.Co_pad = Co_pad;
    if ((rc = nm_launch_conv(make_ref(packed_in, nullptr, nullptr, 1.0f, 1, G, G, G, 8), wfull, bias, field, g, nullptr, s, 4))) return rc;
    if ((rc = nm_launch_conv_k5occ(occ, N, G, wocc, field, out, Cout, Co_pad, gn_groups > 0 ? part : nullptr, s, nullptr, nullptr, nullptr, op_oh()))) return rc;
    if (gn_groups > 0) rc = finish_gn(ctx, part, N, nblk, Cout, gn_groups, (double)G3 * (Cout / gn_groups), gn_gamma, gn_beta, gn_scale, gn_shift);
    return rc;
} catch (...) { return nm_abi_catch("nm_op_conv5_occ"); }

int nm_op_convT2(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                 const float* weight, const float* bias, int32_t Cout, int32_t outpad, float* out,
                 int32_t gn_groups, const float* gn_gamma, const float* gn_beta, float* gn_scale, float* gn_shift) try { NmScope nm_scope_(ctx);
    if (!ctx || !in || !weight || !out || !bias) { nm_set_error("op_convT2: null argument"); return NM_ERR_ARG; }
    const int OD = 2 * D + outpad, OH = 2 * H + outpad, OW = 2 * W + outpad;
    const int vox = OD * OH * OW, nblk = nm_stats_blocks_per_frame(vox);
    int rc = nm_ctx_reserve(ctx, ((size_t)N * nblk * Cout * 2 + (size_t)Cin * Cout * 8) * sizeof(float) + 8192);
    if (rc) return rc;
    ctx->ws.release(0);
    float* wt = ctx->ws.f((size_t)Cin * Cout * 8);
    float* part = ctx->ws.f((size_t)N * nblk * Cout * 2);
    if ((rc = nm_launch_transpose_convT_weight(weight, Cin, Cout, wt, ctx->stream))) return rc;
    TensorRef t = with_h(make_ref(in, nullptr, nullptr, 1.0f, N, D, H, W, Cin), op_ih());
    rc = nm_launch_convT2(t, wt, bias, out, Cout, OD, OH, OW, ctx->stream, op_oh());
    if (rc || gn_groups <= 0) return rc;
    rc = nm_launch_gn_partials(out, N, vox, Cout, part, ctx->stream, op_oh());
    if (rc) return rc;
    return finish_gn(ctx, part, N, nblk, Cout, gn_groups, (double)vox * (Cout / gn_groups), gn_gamma, gn_beta, gn_scale, gn_shift);
} catch (...) { return nm_abi_catch("nm_op_convT2"); }

int nm_op_apply2(nm_ctx* ctx, const float* a, const float* a_scale, const float* a_shift, float a_slope,
                 const float* b, const float* b_scale, const float* b_shift, float b_slope, int32_t N,
                 int32_t voxels, int32_t C, float* out) try { NmScope nm_scope_(ctx);
    if (!ctx || !a || !out) { nm_set_error("op_apply2: null argument"); return NM_ERR_ARG; }
    TensorRef ta = with_h(make_ref(a, a_scale, a_shift, a_slope, N, 1, 1, voxels, C), op_ih());
    TensorRef tb = with_h(make_ref(b, b_scale, b_shift, b_slope, N, 1, 1, voxels, C), op_ih());
    return nm_launch_apply2(ta, b ? &tb : nullptr, out, ctx->stream, op_oh());
} catch (...) { return nm_abi_catch("nm_op_apply2"); }

int nm_op_upsample2(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, float* out) try { NmScope nm_scope_(ctx);
    if (!ctx || !in || !out) { nm_set_error("op_upsample2: null argument"); return NM_ERR_ARG; }
    return nm_launch_upsample2(with_h(make_ref(in, nullptr, nullptr, 1.0f, N, D, H, W, C), op_ih()), out, ctx->stream, op_oh());
} catch (...) { return nm_abi_catch("nm_op_upsample2"); }

int nm_op_pack_input(nm_ctx* ctx, const float* vox, int32_t B, int32_t T, int32_t G, int32_t mean_over_t, float* out) try { NmScope nm_scope_(ctx);
    if (!ctx || !vox || !out) { nm_set_error("op_pack_input: null argument"); return NM_ERR_ARG; }
    return nm_launch_pack_input(vox, B, T, G, mean_over_t, out, ctx->stream);
} catch (...) { return nm_abi_catch("nm_op_pack_input"); }

int nm_op_cl_to_ncdhw(nm_ctx* ctx, const float* in, int32_t N, int32_t voxels, int32_t C, float* out) try { NmScope nm_scope_(ctx);
    if (!ctx || !in || !out) { nm_set_error("op_cl_to_ncdhw: null argument"); return NM_ERR_ARG; }
    return nm_launch_cl_to_ncdhw(make_ref(in, nullptr, nullptr, 1.0f, N, 1, 1, voxels, C), out, ctx->stream);
} catch (...) { return nm_abi_catch("nm_op_cl_to_ncdhw"); }

// ---- backward ops (unit parity of the detector-mode training kernels) ----------------------------------------------------
// Gradients of  y = conv3d(up2? upsample2(a) : a, W) + b  with  a = lrelu(in*scale + shift):
//   d_weight (OIDHW), d_bias, and d_in = dL/da for the first `dgrad_channels` input channels (null: skipped).
int nm_op_conv3d_backward(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                          const float* in_scale, const float* in_shift, float in_slope, const float* weight, int32_t Cout,
                          int32_t ks, int32_t stride, int32_t pad, int32_t up2, const float* dy, float* d_in,
                          int32_t dgrad_channels, float* d_weight, float* d_bias) try { NmScope nm_scope_(ctx);
    if (!ctx || !in || !weight || !dy || !d_weight || !d_bias) { nm_set_error("op_conv3d_backward: null argument"); return NM_ERR_ARG; }
    if (Cout % 8) { nm_set_error("op_conv3d_backward: Cout %% 8 != 0"); return NM_ERR_ARG; }
    const int Cin_pad = (Cin + 7) & ~7, us = up2 ? 2 : 1, taps = ks * ks * ks;
    const int FD = us * D, FH = us * H, FW = us * W;
    const int OD = (FD + 2 * pad - ks) / stride + 1, OH = (FH + 2 * pad - ks) / stride + 1, OW = (FW + 2 * pad - ks) / stride + 1;
    const size_t fine = (size_t)N * FD * FH * FW, ov = (size_t)OD * OH * OW;
    const int csel = dgrad_channels, co_pad2 = (csel + 31) & ~31;
    const size_t wsf = nm_wgrad_ws_floats(N, OD, OH, OW, Cout, Cin_pad, ks, stride);
    const int nbb = nm_gnb_blocks_per_frame((int)ov);
    const size_t wfl2 = nm_packed_weight_floats(ks, Cout, co_pad2);
    size_t need = wsf + (size_t)N * nbb * Cout * 2 + (up2 ? fine * Cin_pad : 0) + (size_t)csel * Cout * taps + 2 * wfl2 + co_pad2 +
                  ((up2 && d_in) ? fine * csel : 0) + 16384;
    int rc = nm_ctx_reserve(ctx, need * sizeof(float));
    if (rc) return rc;
    ctx->ws.release(0);
    hipStream_t s = ctx->stream;
    TensorRef a = with_h(make_ref(in, in_scale, in_shift, in_slope, N, D, H, W, Cin_pad), op_ih());
    TensorRef dyT = with_h(make_ref(dy, nullptr, nullptr, 1.0f, N, OD, OH, OW, Cout), op_oh());
    if (up2) {
        float* up = ctx->ws.f(fine * Cin_pad);
        if ((rc = nm_launch_upsample2(a, up, s, op_oh()))) return rc;
        a = with_h(make_ref(up, nullptr, nullptr, 1.0f, N, FD, FH, FW, Cin_pad), op_oh());
    }
    float* ws = ctx->ws.f(wsf);
    if ((rc = nm_launch_wgrad(a, dyT, ks, stride, pad, Cin, ws, d_weight, s, nullptr, nm_conv_get_mode() != 0))) return rc;
    float* bp = ctx->ws.f((size_t)N * nbb * Cout * 2);
    if ((rc = nm_launch_gnb_partials(dy, dyT, bp, s))) return rc;
    if ((rc = nm_launch_sum_partials(bp, N * nbb, Cout, d_bias, s))) return rc;
    if (!d_in) return NM_OK;
    if (csel <= 0 || csel % 8 || csel > Cin) { nm_set_error("op_conv3d_backward: dgrad_channels %d", csel); return NM_ERR_ARG; }
    if (stride == 1) {
        float* wf = ctx->ws.f((size_t)csel * Cout * taps);
        float* wp = ctx->ws.f(wfl2); float* wp16 = ctx->ws.f(wfl2); float* zb = ctx->ws.f(co_pad2);
        if ((rc = nm_launch_flip_weight(weight, Cout, Cin, csel, ks, wf, s))) return rc;
        if ((rc = nm_launch_pack_conv_weight(wf, csel, Cout, ks, wp, Cout, co_pad2, s))) return rc;
        const bool h16 = Cout % 16 == 0;
        if (h16 && (rc = nm_launch_pack_conv_weight16(wf, csel, Cout, ks, wp16, co_pad2, s))) return rc;
        if ((rc = nm_check_hip(hipMemsetAsync(zb, 0, co_pad2 * sizeof(float), s), "memset"))) return rc;
        ConvGeom g; g.ks = ks; g.stride = 1; g.pad = ks - 1 - pad; g.OD = FD; g.OH = FH; g.OW = FW; g.Cout = csel; g.Co_pad = co_pad2;
        float* dfine = up2 ? ctx->ws.f(fine * csel) : d_in;
        if ((rc = nm_launch_conv(dyT, wp, zb, dfine, g, nullptr, s, Cout, h16 ? wp16 : nullptr, up2 ? op_oh() : op_ih()))) return rc;
        if (up2) rc = nm_launch_upsample2_adjoint(dfine, N, D, H, W, csel, d_in, s, nullptr, op_oh(), op_ih());
        return rc;
    }
    if (stride == 2 && ks == 2 && pad == 0 && !up2) {
        float* wt = ctx->ws.f((size_t)csel * Cout * 8); float* zb = ctx->ws.f(co_pad2);
        if (csel != Cin) { nm_set_error("op_conv3d_backward: pool dgrad needs all input channels"); return NM_ERR_ARG; }
        if ((rc = nm_launch_transpose_convT_weight(weight, Cout, Cin, wt, s))) return rc;
        if ((rc = nm_check_hip(hipMemsetAsync(zb, 0, co_pad2 * sizeof(float), s), "memset"))) return rc;
        return nm_launch_convT2(dyT, wt, zb, d_in, Cin, D, H, W, s, op_ih());
    }
    nm_set_error("op_conv3d_backward: unsupported geometry"); return NM_ERR_UNSUPPORTED;
} catch (...) { return nm_abi_catch("nm_op_conv3d_backward"); }

// first layer: dW [Cout][4][125] and d_bias of conv5(cat[occ, coords]) given dy
int nm_op_conv5_occ_backward(nm_ctx* ctx, const float* occ, int32_t N, int32_t G, int32_t Cout, const float* dy,
                             float* d_weight, float* d_bias, int32_t sparse_occ) try { NmScope nm_scope_(ctx);
    if (!ctx || !occ || !dy || !d_weight || !d_bias) { nm_set_error("op_conv5_occ_backward: null argument"); return NM_ERR_ARG; }
    const size_t wsf = nm_wgrad_k5occ_ws_floats(N, G, Cout);
    const int nbb = nm_gnb_blocks_per_frame(G * G * G);
    int rc = nm_ctx_reserve(ctx, (wsf + (size_t)N * nbb * Cout * 2) * sizeof(float) + 8192);
    if (rc) return rc;
    ctx->ws.release(0);
    TensorRef dyT = with_h(make_ref(dy, nullptr, nullptr, 1.0f, N, G, G, G, Cout), op_oh());
    float* ws = ctx->ws.f(wsf); float* bp = ctx->ws.f((size_t)N * nbb * Cout * 2);
    if ((rc = nm_launch_wgrad_k5occ(occ, N, G, dyT, ws, d_weight, ctx->stream, sparse_occ))) return rc;
    if ((rc = nm_launch_gnb_partials(dy, dyT, bp, ctx->stream))) return rc;
    return nm_launch_sum_partials(bp, N * nbb, Cout, d_bias, ctx->stream);
} catch (...) { return nm_abi_catch("nm_op_conv5_occ_backward"); }

// Gradients of  y = convT3d_k2s2(a, W) + b  (a = lrelu(in*scale+shift)); weight IODHW
int nm_op_convT2_backward(nm_ctx* ctx, const float* in, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                          const float* in_scale, const float* in_shift, float in_slope, const float* weight, int32_t Cout,
                          int32_t outpad, const float* dy, float* d_in, float* d_weight, float* d_bias) try { NmScope nm_scope_(ctx);
    if (!ctx || !in || !weight || !dy || !d_in || !d_weight || !d_bias) { nm_set_error("op_convT2_backward: null argument"); return NM_ERR_ARG; }
    if (Cin % 8 || Cout % 8) { nm_set_error("op_convT2_backward: channels must be multiples of 8"); return NM_ERR_ARG; }
    const int OD = 2 * D + outpad, OH = 2 * H + outpad, OW = 2 * W + outpad;
    const size_t ov = (size_t)OD * OH * OW;
    const int ci_pad = (Cin + 31) & ~31;
    const size_t wsf = nm_wgrad_ws_floats(N, D, H, W, Cin, Cout, 2, 2);
    const int nbb = nm_gnb_blocks_per_frame((int)ov);
    const size_t wfl = nm_packed_weight_floats(2, Cout, ci_pad);
    int rc = nm_ctx_reserve(ctx, (wsf + (size_t)N * nbb * Cout * 2 + 2 * wfl + ci_pad) * sizeof(float) + 16384);
    if (rc) return rc;
    ctx->ws.release(0);
    hipStream_t s = ctx->stream;
    TensorRef a = make_ref(in, in_scale, in_shift, in_slope, N, D, H, W, Cin);
    TensorRef dyT = make_ref(dy, nullptr, nullptr, 1.0f, N, OD, OH, OW, Cout);
    float* ws = ctx->ws.f(wsf);
    if ((rc = nm_launch_wgrad(dyT, a, 2, 2, 0, Cout, ws, d_weight, s, nullptr, nm_conv_get_mode() != 0))) return rc;       // roles swapped: [Cin][Cout][8] = IODHW
    float* bp = ctx->ws.f((size_t)N * nbb * Cout * 2);
    if ((rc = nm_launch_gnb_partials(dy, dyT, bp, s))) return rc;
    if ((rc = nm_launch_sum_partials(bp, N * nbb, Cout, d_bias, s))) return rc;
    float* wp = ctx->ws.f(wfl); float* wp16 = ctx->ws.f(wfl); float* zb = ctx->ws.f(ci_pad);
    if ((rc = nm_launch_pack_conv_weight(weight, Cin, Cout, 2, wp, Cout, ci_pad, s))) return rc;   // IODHW read as OIDHW of the adjoint conv
    const bool h16 = Cout % 16 == 0;
    if (h16 && (rc = nm_launch_pack_conv_weight16(weight, Cin, Cout, 2, wp16, ci_pad, s))) return rc;
    if ((rc = nm_check_hip(hipMemsetAsync(zb, 0, ci_pad * sizeof(float), s), "memset"))) return rc;
    ConvGeom g; g.ks = 2; g.stride = 2; g.pad = 0; g.OD = D; g.OH = H; g.OW = W; g.Cout = Cin; g.Co_pad = ci_pad;
    return nm_launch_conv(dyT, wp, zb, d_in, g, nullptr, s, Cout, h16 ? wp16 : nullptr);
} catch (...) { return nm_abi_catch("nm_op_convT2_backward"); }

// GroupNorm(groups) + LeakyReLU(slope) backward on a raw tensor y [N][voxels][C]: dy, dgamma, dbeta, and sum_v dy (the conv bias gradient)
int nm_op_gn_backward(nm_ctx* ctx, const float* y, int32_t N, int32_t voxels, int32_t C, int32_t groups, const float* gamma,
                      const float* beta, float slope, const float* dA, float* dy, float* dgamma, float* dbeta, float* dbias) try { NmScope nm_scope_(ctx);
    if (!ctx || !y || !gamma || !beta || !dA || !dy || !dgamma || !dbeta || !dbias) { nm_set_error("op_gn_backward: null argument"); return NM_ERR_ARG; }
    const int nbf = nm_stats_blocks_per_frame(voxels), nbb = nm_gnb_blocks_per_frame(voxels);
    int rc = nm_ctx_reserve(ctx, ((size_t)N * (nbf + nbb) * C * 2 + (size_t)N * C * 10) * sizeof(float) + 16384);
    if (rc) return rc;
    ctx->ws.release(0);
    hipStream_t s = ctx->stream;
    float* fpart = ctx->ws.f((size_t)N * nbf * C * 2); float* bpart = ctx->ws.f((size_t)N * nbb * C * 2);
    float* scale = ctx->ws.f((size_t)N * C); float* shift = ctx->ws.f((size_t)N * C);
    float* coef = ctx->ws.f((size_t)N * C * 4); float* dgn = ctx->ws.f((size_t)N * C * 4);
    if ((rc = nm_launch_gn_partials(y, N, voxels, C, fpart, s, op_ih()))) return rc;
    if ((rc = finish_gn(ctx, fpart, N, nbf, C, groups, (double)voxels * (C / groups), gamma, beta, scale, shift))) return rc;
    TensorRef yT = with_h(make_ref(y, scale, shift, slope, N, 1, 1, voxels, C), op_ih());
    if ((rc = nm_launch_gnb_partials(dA, yT, bpart, s))) return rc;
    if ((rc = nm_launch_gnb_finalize(bpart, nbb, fpart, nbf, N, C, groups, voxels, gamma, 1e-5f, coef, dgn, s))) return rc;
    if ((rc = nm_launch_sum_frames(dgn, N, C, 4, 0, dgamma, s))) return rc;
    if ((rc = nm_launch_sum_frames(dgn, N, C, 4, 1, dbeta, s))) return rc;
    if ((rc = nm_launch_sum_frames(dgn, N, C, 4, 2, dbias, s))) return rc;
    return nm_launch_gnb_apply(dA, yT, coef, dy, s);
} catch (...) { return nm_abi_catch("nm_op_gn_backward"); }

}  // extern "C"
