// Context object behind the C ABI (include/nm355.h): device, stream, packed weights,
// stream-ordered bump-allocated workspace.
#pragma once
#include "nm_common.h"
#include "../../include/nm355.h"
#include <map>
#include <string>
#include <vector>

struct Arena {
    char* base = nullptr;
    size_t cap = 0, top = 0, peak = 0;
    bool dry = false;            // dry run: only measure the high-water mark, launch nothing
    bool overflow = false;

    void* alloc_bytes(size_t bytes) {
        size_t a = (top + 255) & ~(size_t)255;
        top = a + bytes;
        if (top > peak) peak = top;
        if (dry) return reinterpret_cast<void*>((uintptr_t)256);
        if (top > cap) { overflow = true; return nullptr; }
        return base + a;
    }
    float* f(size_t n) { return static_cast<float*>(alloc_bytes(n * sizeof(float))); }
    size_t mark() const { return top; }
    void release(size_t m) { top = m; }
};

// ---- network weights (ctx-owned device copies) -------------------------------------------
struct ConvW {
    int Cin = 0, Cin_pad = 0, Cout = 0, Co_pad = 0, ks = 0;
    int Cout_src = 0;         // rows of the state_dict weight / bias (< Cout when the library pads the layer's output channels: the heat-map heads)
    float* wp = nullptr;      // fp32 packed [tap][Cin/4][Co_pad][4]
    void* wp16 = nullptr;     // split-fp16 packed (Cin % 16 == 0 only), see nm_conv.hip
    float* bias = nullptr;
    void* wup = nullptr;      // fused-upsample layers: composite weight sets of nm_up2c.hip
    // training (nm_ctx_set_training): state_dict key prefix and the weights of the data-gradient convolution
    std::string key;
    int csel = 0, cd_pad = 0;  // input channels that receive a gradient (Cin rounded down to 8, or what Loader::conv was asked for) and their packed width
    float* wd = nullptr;       // ks 1/3: flipped + transposed, fp32 packed [tap][Cout/4][cd_pad][4]
    void* wd16 = nullptr;      //         same, split-fp16 (Cout % 16 == 0)
    float* wt = nullptr;       // ks 2 (stride 2): [tap][Cout][Cin] for the transposed-conv kernel
};
struct NormW { int C = 0, groups = 0; float* gamma = nullptr; float* beta = nullptr; std::string key; };
struct ResW { ConvW c1, c2, cs; NormW n1, n2, ns; bool has_skip = false; };
struct PoolW { ConvW c; NormW n; };
struct UpW {
    int Cin = 0, Cout = 0; float* w = nullptr; float* bias = nullptr; NormW n;
    std::string key; int cd_pad = 0; float* wd = nullptr; void* wd16 = nullptr;   // training: the adjoint k2 s2 conv (packed)
};
struct HourglassW { PoolW p1, p2, p3; ResW e1, e2, e3, d3, d2, d1, s1, s2, s3; UpW u3, u2, u1; };
struct FeatNetW {
    ConvW c0; NormW n0; PoolW p1, p3; ResW r2, r5; HourglassW hg;
    float* occ_w = nullptr;    // occupancy-channel taps of c0: fp32 [32 tap-quads][Co_pad][4], then split fp16 [8 k-steps][4][Co_pad][8]
    float* field = nullptr;    // conv5(cat[0, coords]) + bias for one frame: [G^3][Cout]
    float* field_part = nullptr;  // GroupNorm partial sums of the field per 4x8x8 brick [bricks][Cout][2] (sparse first layer)
};
struct LinearW { int in = 0, out = 0; float* w = nullptr; float* b = nullptr; };

struct DetectorW {
    FeatNetW frame, clip;
    ConvW head, clip_head, adjust;
    ConvW adjust_rest;                     // inference: columns K.. of `adjust` (first_feature, gauss_0, coords) as a conv of its own
    float* adjust_wg = nullptr;            //            columns 0..K-1 transposed to [K][Cout] (the per-frame gaussian part)
    float* prop = nullptr;                 // [w0, w1, b] of propagate_heatmaps (device)
    float* sigma_param = nullptr;          // vox_to_kypt.sigmas [K] (fixed_sigma = 0 only)
    float* zeros = nullptr;                // 512 zeros (bias of the data-gradient convolutions)
    ConvW d1, d4, d8, d11; NormW dn2, dn5, dn9, dn12;
    float* d14 = nullptr;                  // [32 weights, bias] of the final 1x1 conv (device)
    float* affinity_params = nullptr;      // (N,K,K-1); (N,K,K) for get_affinity versions 0 / 1 / 2
};

struct VrnnW {
    LinearW post0, post2, prior0, prior2, root0, root2, joint0, joint2;
    float *w_ih = nullptr, *w_hh = nullptr, *b_ih = nullptr, *b_hh = nullptr;
    float* h0 = nullptr;                   // init_kypt_rnn_state (1,H)
    float* offset_param = nullptr;         // (K,3)
    int32_t* parents = nullptr;            // device (K)
    int32_t* order = nullptr;              // device (K)
    int32_t* lvl_joint = nullptr;          // device (K): joints grouped by tree depth, level 0 (the root) first
    int32_t* lvl_start = nullptr;          // device (nlevels + 1): first entry of each level in lvl_joint
    int nlevels = 0;
    std::vector<int32_t> parents_h, order_h;
    bool has_tree = false;
    uint64_t tree_epoch = 0;               // bumped by nm_vrnn_set_tree (captured rollout graphs bake the level count in)
};

struct nm_ctx {
    nm_config cfg;
    NmLaunchState ls;                      // conv mode, profiler records, status word, A/B switches: nothing of this is process-wide
    hipStream_t stream = nullptr;
    bool stream_bound = false;             // nm_ctx_set_stream has been called (nullptr = the legacy default stream is a valid choice)
    hipStream_t stream2 = nullptr;         // ctx-owned side stream: clip-mean net / VRNN run beside the frame stack
    hipEvent_t ev_fork = nullptr, ev_clip = nullptr, ev_kp = nullptr, ev_side = nullptr;
    // weight gradients of the training backward on a stream of their own (nm_net.hip conv_bwd): their inputs that must outlive the main
    // stream's arena frames (a ring of dY buffers, the re-materialised upsample + slot workspace, the operand-scale vectors) live in one
    // ctx-owned block sized by the sizing pass of nm_detector_forward_train
    hipStream_t stream3 = nullptr;
    bool side_shared = false;              // stream2 / stream3 are the process-wide pair of this device (nm_api.hip acquire_side_streams)
    hipEvent_t ev_w[3] = {nullptr, nullptr, nullptr}, ev_dy = nullptr, ev_wjoin = nullptr, ev_k5 = nullptr;
    float* wside = nullptr; size_t wside_floats = 0;                     // the block and its capacity
    size_t wside_slot = 0, wside_scratch = 0, wside_sc = 0;              // floats per ring slot / scratch / scale pool (last sizing pass)
    hipEvent_t ev_user_decoder = nullptr;  // caller's event, recorded by nm_detector_backward once the decoder's gradients are complete
    unsigned* nf_flag = nullptr;           // sticky: 1 = a conv produced non-finite values (nm_ctx_check_nonfinite reads and clears it)
    // Deferred range guard (nm_net.hip nf_post / nf_poll): every forward-type call ends with an asynchronous copy of the status word
    // into one of four pinned host slots + an event; the next calls' entry reads the slots whose event has completed (and waits for
    // the ones two or more calls old, which never stalls the pipeline) - no host synchronisation per call.
    unsigned* nf_host = nullptr;           // pinned [4]
    hipEvent_t ev_nf[4] = {nullptr, nullptr, nullptr, nullptr};
    bool nf_busy[4] = {false, false, false, false};
    uint64_t nf_seq[4] = {0, 0, 0, 0};     // call number of the slot's copy
    const char* nf_who[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t nf_calls = 0;                 // forward-type calls so far
    unsigned nf_last = 0;                  // status bits of the slot that tripped (1: non-finite conv statistics, 2: rollout time-out)
    int range_check = 1;                   // NM355_RANGE_CHECK=0 switches the deferred guard off (A/B)
    int learn_sigma = 0;                   // options.fixed_sigma == 0 (kypt_detector.py:258-260): nm_ctx_set_learnable_sigma
    int gauss_cat = 0;                     // options.gaussian_cat_type (kypt_detector.py:396-401): 0 'none', 1 'max', 2 'sum' (nm_ctx_set_gaussian_cat)
    int affinity_ver = 3;                  // get_affinity version (kypt_detector.py:171-210): 3 = the shipped configurations; 0 / 1 / 2 by nm_ctx_set_affinity_ver
    int64_t affinity_numel() const { return (int64_t)cfg.nneighbor * cfg.nkeypoints * (affinity_ver == 3 ? cfg.nkeypoints - 1 : cfg.nkeypoints); }
    Arena ws;                              // activations / scratch, reset per call
    Arena ws2;                             // scratch of work issued on stream2 (VRNN beside the decoder)
    std::vector<void*> owned;              // weight allocations
    std::vector<size_t> owned_bytes;       // their sizes: a repeated nm_ctx_set_weights walks the same sequence and reuses them (no sync)
    size_t owned_cursor = 0;
    uint64_t weights_epoch = 0;            // bumped whenever a weight buffer is (re)allocated: captured graphs hold the old pointers
    bool has_weights = false;
    bool training = false;                 // nm_ctx_set_training: set_weights also packs the data-gradient weights
    Arena ws_t;                            // activations retained between nm_detector_forward_train and nm_detector_backward
    std::vector<char> host_table2;         // host staging of set_weights' copy table
    void* copy_table = nullptr; size_t copy_table_cap = 0;      // its device copy
    std::vector<char> host_table3;         // the same for the weight-pack job table (nm_launch_pack_jobs)
    void* pack_table = nullptr; size_t pack_table_cap = 0;
    std::vector<char> host_table;          // host staging of nm_adam_step_multi's pointer table (kept alive across the async copy)
    struct TrainTape* tape = nullptr;      // what the backward pass needs of the last training forward (nm_net.hip)
    void* vtape = nullptr;                 // VrnnTape of the last nm_vrnn_encode_train (nm_vrnn.hip)
    void* vgraphs = nullptr;               // cache of captured rollout graphs (nm_vrnn.hip)
    bool in_fused = false;                 // nm_forward_fused is running the VRNN encode on the side stream beside the decoder
    int32_t* vrnn_cnt = nullptr;           // 256 arrival counters of vrnn_post_mid_kernel (zero between launches)
    DetectorW det;
    VrnnW vrnn;
};

// selects ctx->ls for the calling thread for the duration of one ABI call (nested calls restore the outer selection)
extern thread_local NmLaunchState* nm_tls_ls;
struct NmScope {
    NmLaunchState* prev;
    explicit NmScope(nm_ctx* c) : prev(nm_tls_ls) { if (c) nm_tls_ls = &c->ls; }
    ~NmScope() { nm_tls_ls = prev; }
    NmScope(const NmScope&) = delete;
    NmScope& operator=(const NmScope&) = delete;
};

// deferred status (nm_net.hip): poll at the entry of an ABI call, post behind a call that can set the status word
int nm_nf_poll(nm_ctx* c);
void nm_nf_post(nm_ctx* c, const char* who);
int nm_ctx_reserve(nm_ctx* ctx, size_t bytes);        // grow the workspace (synchronises)
float* nm_ctx_weight_alloc(nm_ctx* ctx, size_t floats);

// nm_vrnn.hip
void nm_vrnn_free_tape(nm_ctx* ctx);
void nm_vrnn_invalidate_tape(nm_ctx* ctx);
void nm_vrnn_free_graphs(nm_ctx* ctx);          // captured rollout graphs (nm_vrnn_generate / nm_vrnn_rollout)
// nm_net.hip
void nm_net_free_tape(nm_ctx* ctx);
int nm_net_set_weights(nm_ctx* ctx, const std::map<std::string, std::pair<const float*, int64_t>>& sd);
