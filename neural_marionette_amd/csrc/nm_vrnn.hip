// Hierarchical-skeleton VRNN (model/hsvrnn_bvh.py, utils/geo_utils.py) on gfx950.
//
// A timestep is latency-bound (3.2 MMAC per sample, 6.1 MB of weights that stay in L2 /
// Infinity Cache), so it is organised as a short chain of wide launches, every one of
// which spreads its rows over the whole chip (one wavefront per output row, lanes split
// the K dimension, xor-shuffle tree reduction -> deterministic):
//   1. h-phase     every product that only needs h_{t-1}: first layers of the prior and
//                  posterior MLPs, the h-halves of both FK decoders' first layers
//                  (W [h, z_i] = W_h h + W_z z_i, shared by the S samples) and the GRU's W_hh h
//   2. dist        second layers -> (mu, softplus(std)+1e-4), z_i = mu + eps_i * std
//   3. dec1/dec2   z-halves of the decoders (+ shared h-half), LeakyReLU; heads (tanh / 6-D)
//   4. fk          6-D -> SO(3), forward kinematics along the tree, best-of-S argmin, KL
//   5. gru         W_ih [kp*, z*] + gates -> h_t
// A kernel boundary on one stream costs ~1.5 us on MI355X, less than a grid barrier, so the
// phases are separate launches rather than one persistent kernel.
#include "nm_ctx.h"
#include <cmath>
#include <algorithm>
#include <vector>

namespace {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }


struct LinJob {
    const float* W; int ldw; int col0;
    const float* xa; int na; int lda;
    const float* xb; int nb; int ldb;
    const float* bias;
    const float* add; int ldadd; int add_mod;
    float* out; int ldo;
    int rows; int act; int batch;
    const float* gate; int ldgate;     // backward: multiply by LeakyReLU'(saved activation) = (gate > 0 ? 1 : 0.01)
};
struct LinJobs { LinJob j[5]; int n; int start[6]; };

// acc[s] += sum_k W[k] * x[b0+s][k] over this lane's k's.  NB = samples per wavefront pass.
// 16-B loads when rows are 4-float aligned (every weight / activation stride of this model is).
template <int NB>
__device__ __forceinline__ void dot_seg(const float* __restrict__ w, const float* __restrict__ x, int n, int ld, int b0,
                                        int batch, int lane, float (&acc)[NB]) {
    if (!x) return;
    const bool vec = ((n | ld) & 3) == 0 && ((((uintptr_t)w) | ((uintptr_t)x)) & 15) == 0;
    if (vec) {
        // the NB row loads of a step first, then the arithmetic: written load-use-load-use the compiler kept ONE register quad for
        // the rows and waited (s_waitcnt vmcnt(0)) between them - NB dependent memory round trips per step (B = 4: 31 us per
        // launch where B = 1 takes 4.7)
        const float* xr[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) { int b = b0 + s; b = b < batch ? b : batch - 1; xr[s] = x + (size_t)b * ld; }
        for (int k = lane * 4; k < n; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
            f32x4 xv[NB];
#pragma unroll
            for (int s = 0; s < NB; ++s) xv[s] = *reinterpret_cast<const f32x4*>(xr[s] + k);
#pragma unroll
            for (int s = 0; s < NB; ++s) asm volatile("" : "+v"(xv[s]));
#pragma unroll
            for (int s = 0; s < NB; ++s) acc[s] += ((wv[0] * xv[s][0] + wv[1] * xv[s][1]) + wv[2] * xv[s][2]) + wv[3] * xv[s][3];
        }
    } else {
        for (int k = lane; k < n; k += 64) {
            const float wv = w[k];
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                int b = b0 + s; b = b < batch ? b : batch - 1;
                acc[s] += wv * x[(size_t)b * ld + k];
            }
        }
    }
}
template <int NB>
__device__ __forceinline__ void wave_reduce(float (&acc)[NB]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int s = 0; s < NB; ++s) acc[s] += nm_sx(acc[s], off);
}
template <int NB>
__device__ __forceinline__ float pick(const float (&acc)[NB], int s) {
    float v = acc[0];
#pragma unroll
    for (int i = 1; i < NB; ++i) v = (s == i) ? acc[i] : v;
    return v;
}

// out[b][r] = act(W[r, col0:col0+na+nb] . [xa[b] | xb[b]] + bias[r] + add[b % add_mod][r])
template <int NB>
__global__ __launch_bounds__(256) void linear_rows_kernel(LinJobs jobs) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    int ji = -1;
#pragma unroll
    for (int i = 0; i < 5; ++i) if (i < jobs.n && w >= jobs.start[i] && w < jobs.start[i + 1]) ji = i;
    if (ji < 0) return;
    ji = __builtin_amdgcn_readfirstlane(ji);            // wave-uniform: keep the job record in SGPRs
    const LinJob& J = jobs.j[ji];
    const int r = w - jobs.start[ji];
    const int b0 = blockIdx.y * NB;
    if (b0 >= J.batch) return;
    float acc[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) acc[s] = 0.f;
    const float* wr = J.W + (size_t)r * J.ldw + J.col0;
    // epilogue operands requested BEFORE the dot products, unconditionally, from addresses that are valid either way (a load under
    // a condition is waited for at the join, and one issued after the reduction is a second dependent L2 round trip per launch)
    const int bl = min(b0 + (lane < NB ? lane : 0), J.batch - 1);
    const float bias_v = *(J.bias ? J.bias + r : wr);
    const float add_v = *(J.add ? J.add + (size_t)(bl % J.add_mod) * J.ldadd + r : wr);
    const float gate_v = *(J.gate ? J.gate + (size_t)bl * J.ldgate + r : wr);
    dot_seg(wr, J.xa, J.na, J.lda, b0, J.batch, lane, acc);
    dot_seg(wr + J.na, J.xb, J.nb, J.ldb, b0, J.batch, lane, acc);
    wave_reduce(acc);
    if (lane < NB && b0 + lane < J.batch) {
        const int b = b0 + lane;
        float v = pick(acc, lane);
        if (J.bias) v += bias_v;
        if (J.add) v += add_v;
        if (J.act == 1) v = lrelu(v, 0.01f);
        else if (J.act == 2) v = tanhf(v);
        if (J.gate) v *= (gate_v > 0.f ? 1.0f : 0.01f);
        J.out[(size_t)b * J.ldo + r] = v;
    }
}

// ---- the same jobs as a tiled GEMM on the fp32 matrix cores, for large batches ------------------------------------------------
// The reference's interpolation demo pushes S = 10 000 sample rows through every MLP / GRU matrix of the VRNN
// (vis_interpolation.py:91-143): 10 000 x 640 . 640 x 128, 10 000 x 512 . 512 x 1536 ... - real GEMMs, where one wavefront per
// output row re-streams the whole batch per row.  Here: workgroup tile 128 samples x 64 outputs, 4 waves x (32 samples x 64
// outputs), v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate), K in LDS chunks of 32 with the permuted-K operand
// trick of conv_mfma_kernel (lane half h owns k = 8q + 4h .. +3: one ds_read_b128 feeds four MFMAs; row pitch 36 floats keeps the
// 16-lane groups on distinct banks).  Same LinJob semantics (two input segments, column offset, bias, broadcast add, activation).
#define GM_BM 128
#define GM_BN 64
#define GM_KC 32
#define GM_PITCH 36
struct GemmJobs { LinJob j[5]; int n; int tile0[6]; int mt[5]; };       // tile0: first workgroup of job i; mt: its sample tiles

__global__ __launch_bounds__(256) void gemm_rows_mfma_kernel(GemmJobs jobs) {
    __shared__ float xs[GM_BM * GM_PITCH];
    __shared__ float wsm[GM_BN * GM_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    int ji = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i) if (i < jobs.n && (int)blockIdx.x >= jobs.tile0[i]) ji = i;
    ji = __builtin_amdgcn_readfirstlane(ji);
    const LinJob& J = jobs.j[ji];
    const int t = (int)blockIdx.x - jobs.tile0[ji];
    const int b0 = (t % jobs.mt[ji]) * GM_BM, r0 = (t / jobs.mt[ji]) * GM_BN;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int K = J.na + J.nb;
    for (int k0 = 0; k0 < K; k0 += GM_KC) {
        // stage x[b0 .. +128][k0 .. +32] and W[r0 .. +64][col0 + k0 .. +32]  (a chunk lies inside one input segment)
        const bool in_a = k0 < J.na;
        const float* xsrc = in_a ? J.xa : J.xb;
        const int ld = in_a ? J.lda : J.ldb, kk = in_a ? k0 : k0 - J.na;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < (GM_BM * GM_KC / 4) / 256; ++i) {
            const int e = tid + 256 * i, row = e >> 3, c4 = (e & 7) * 4;
            const int b = min(b0 + row, J.batch - 1);
            *reinterpret_cast<f32x4*>(xs + row * GM_PITCH + c4) = *reinterpret_cast<const f32x4*>(xsrc + (size_t)b * ld + kk + c4);
        }
#pragma unroll
        for (int i = 0; i < (GM_BN * GM_KC / 4) / 256; ++i) {
            const int e = tid + 256 * i, row = e >> 3, c4 = (e & 7) * 4;
            const int r = min(r0 + row, J.rows - 1);
            *reinterpret_cast<f32x4*>(wsm + row * GM_PITCH + c4) = *reinterpret_cast<const f32x4*>(J.W + (size_t)r * J.ldw + J.col0 + k0 + c4);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < GM_KC / 8; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xs + (wave * 32 + l31) * GM_PITCH + 8 * q + 4 * h);
            f32x4 bq[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bq[nt] = *reinterpret_cast<const f32x4*>(wsm + (nt * 32 + l31) * GM_PITCH + 8 * q + 4 * h);
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sidx], bq[nt][sidx], acc[nt], 0, 0, 0);
        }
    }
    // epilogue: lane = output row r0 + nt*32 + l31; register r = sample (r & 3) + 8 (r >> 2) + 4 h of the wave's 32
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int r = r0 + nt * 32 + l31;
        if (r >= J.rows) continue;
        const float bv = J.bias ? J.bias[r] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int b = b0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (b >= J.batch) continue;
            float v = acc[nt][i] + bv;
            if (J.add) v += J.add[(size_t)(b % J.add_mod) * J.ldadd + r];
            if (J.act == 1) v = lrelu(v, 0.01f);
            else if (J.act == 2) v = tanhf(v);
            if (J.gate) v *= (J.gate[(size_t)b * J.ldgate + r] > 0.f ? 1.0f : 0.01f);
            J.out[(size_t)b * J.ldo + r] = v;
        }
    }
}

// GRU cell from precomputed input / hidden projections (large batches: gi from the GEMM above): torch.nn.GRUCell's gates r, z, n
__global__ __launch_bounds__(256) void gru_gates_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
                                                        int ldh, float* __restrict__ hout, int ldo, int H, int B) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= (size_t)B * H) return;
    const int b = (int)(i / H), j = (int)(i % H);
    const float* a = gi + (size_t)b * 3 * H; const float* g = gh + (size_t)b * 3 * H;
    const float rg = sigmoidf(a[j] + g[j]);
    const float zg = sigmoidf(a[H + j] + g[H + j]);
    const float ng = tanhf(a[2 * H + j] + rg * g[2 * H + j]);
    const float hp = h[(size_t)b * ldh + j];
    hout[(size_t)b * ldo + j] = (hp - ng) * zg + ng;
}

// second layer of a prior / posterior MLP: rows (r, r+Z) -> mu, std = softplus(.)+1e-4 and, when eps is given,
// z[i][b][r] = mu + eps[i][b][r] * std   (hsvrnn_bvh.py:93-107)
struct DistJob { const float* W; const float* bias; const float* x; float* mu; float* sig; const float* eps; float* z; int S; float* raw_s; };
template <int NB>
__global__ __launch_bounds__(256) void dist_rows_kernel(DistJob a, DistJob b, int njobs, int Z, int hid, int B) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= njobs * Z) return;
    const DistJob& J = (w < Z) ? a : b;
    const int r = w % Z;
    const int b0 = blockIdx.y * NB;
    float am[NB], as[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) { am[s] = 0.f; as[s] = 0.f; }
    const float bias_mu = J.bias[r], bias_s = J.bias[r + Z];       // (before the dot products: one L2 round trip, not two)
    dot_seg(J.W + (size_t)r * hid, J.x, hid, hid, b0, B, lane, am);
    dot_seg(J.W + (size_t)(r + Z) * hid, J.x, hid, hid, b0, B, lane, as);
    wave_reduce(am); wave_reduce(as);
    if (lane < NB && b0 + lane < B) {
        const int bb = b0 + lane;
        const float mu = pick(am, lane) + bias_mu;
        const float sraw = pick(as, lane) + bias_s;
        const float sg = softplus(sraw) + 1e-4f;
        J.mu[(size_t)bb * Z + r] = mu; J.sig[(size_t)bb * Z + r] = sg;
        if (J.raw_s) J.raw_s[(size_t)bb * Z + r] = sraw;
        if (J.eps) for (int i = 0; i < J.S; ++i) {
            const size_t o = ((size_t)i * B + bb) * Z + r;
            J.z[o] = mu + J.eps[o] * sg;
        }
    }
}

// GRUCell (gate order r,z,n; ATen form h' = (h - n) * z + n).  gh = W_hh h + b_hh from the h-phase.
template <int NB>
__global__ __launch_bounds__(256) void gru_rows_kernel(const float* __restrict__ W_ih, const float* __restrict__ b_ih,
                                                       const float* __restrict__ xa, int na, int lda,
                                                       const float* __restrict__ xb, int nb, int ldb,
                                                       const float* __restrict__ gh, const float* __restrict__ h, int ldh,
                                                       float* __restrict__ hout, int ldo, int H, int B,
                                                       float* __restrict__ tape_gates /* [4][B][H]: r, z, n, W_hn h + b_hn; or null */) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= H) return;
    const int b0 = blockIdx.y * NB;
    const int in = na + nb;
    float ar[NB], az[NB], an[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) { ar[s] = 0.f; az[s] = 0.f; an[s] = 0.f; }
    const float* w0 = W_ih + (size_t)j * in; const float* w1 = W_ih + (size_t)(H + j) * in; const float* w2 = W_ih + (size_t)(2 * H + j) * in;
    // gate operands requested before the dot products (one L2 round trip instead of two dependent ones)
    const int bl = min(b0 + (lane < NB ? lane : 0), B - 1);
    const float* gl = gh + (size_t)bl * 3 * H;
    const float g_r = gl[j], g_z = gl[H + j], g_n = gl[2 * H + j], hp = h[(size_t)bl * ldh + j];
    const float bi_r = b_ih[j], bi_z = b_ih[H + j], bi_n = b_ih[2 * H + j];
    dot_seg(w0, xa, na, lda, b0, B, lane, ar); dot_seg(w0 + na, xb, nb, ldb, b0, B, lane, ar);
    dot_seg(w1, xa, na, lda, b0, B, lane, az); dot_seg(w1 + na, xb, nb, ldb, b0, B, lane, az);
    dot_seg(w2, xa, na, lda, b0, B, lane, an); dot_seg(w2 + na, xb, nb, ldb, b0, B, lane, an);
    wave_reduce(ar); wave_reduce(az); wave_reduce(an);
    if (lane < NB && b0 + lane < B) {
        const int b = b0 + lane;
        const float rg = sigmoidf((pick(ar, lane) + bi_r) + g_r);
        const float zg = sigmoidf((pick(az, lane) + bi_z) + g_z);
        const float ng = tanhf((pick(an, lane) + bi_n) + rg * g_n);
        hout[(size_t)b * ldo + j] = (hp - ng) * zg + ng;
        if (tape_gates) {
            const size_t o = (size_t)b * H + j, pl = (size_t)B * H;
            tape_gates[o] = rg; tape_gates[pl + o] = zg; tape_gates[2 * pl + o] = ng; tape_gates[3 * pl + o] = g_n;
        }
    }
}

// ---- forward kinematics + best-of-S selection --------------------------------------------------------------------
struct FkArgs {
    const float* root;  int ldr;       // [S*B][ldr]  tanh MLP output: root xyz (3) + K intensities
    const float* rot;                  // [S*B][6K]
    const float* offset;               // [B][K][3]
    const float* obs; int ldobs;       // [B][..] detected keypoints (K*4) or null (prior step)
    const float* z;                    // [S*B][Z]
    const int32_t* order; const int32_t* parents;
    const int32_t* lvl_joint; const int32_t* lvl_start; int nlevels;   // joints grouped by tree depth (level 0 = the root): nm_vrnn_set_tree
    const float *qmu, *qsig, *pmu, *psig;   // posterior / prior params [B][Z] (KL) or null
    float* out_kp; int ldkp;           // best keypoints (K*4)
    float* out_z; int ldz;
    float* out_R; int ldR;             // best global rotations (K*9) or null
    int32_t* best; int ldbest;         // or null
    float* kl; float* rec; int ldstat; // per-sample sums (or null)
    int K, S, B, Z;
    // training tape (all null in inference): hidden layers of the best sample's decoders, its heads, rotations, noise
    const float* hr; const float* hj; const float* eps;              // sources: [S*B][128], [S*B][128], (S,B,Z)
    float *t_hr, *t_hj, *t_raw, *t_rot6, *t_Rl, *t_Rg, *t_eps;       // [B][128], [B][128], [B][3+K], [B][6K], [B][9K], [B][9K], [B][Z]
};

// Forward kinematics over S samples of one batch element, level by level (called by all threads of the workgroup; ends with a barrier).
// Rl: local rotations [S][K][9] (in LDS, complete), Rg / pos: outputs [S][K][9] / [S][K][3].  root: tanh-MLP rows [S*B][ldr].
// The tree tables and this batch element's bone offsets come from LDS (FkTables, filled at kernel start): read from global memory
// inside the level loop every level is a chain of three dependent L2 round trips (level bounds -> joint -> parent), ~2 us per level.
// A thread works for ONE sample: sample `smp` (or -1), as lane `lt` of the `per` threads of that sample.
struct FkTables { int32_t lvl_joint[32], lvl_start[34], parents[32]; float offset[96]; };
__device__ __forceinline__ void fk_tables_load(FkTables& tb, const int32_t* __restrict__ lvl_joint, const int32_t* __restrict__ lvl_start,
                                               const int32_t* __restrict__ parents, const float* __restrict__ offset_b, int K, int nlevels, int tid) {
    // one batch of independent, UNCONDITIONAL loads from clamped addresses (a load under a branch makes hipcc drain all outstanding
    // loads at the join: three branches were three serial L2 round trips); the caller's next barrier publishes the table
    const int32_t lj = lvl_joint[min(tid, K - 1)], pa = parents[min(tid, K - 1)], ls = lvl_start[min(tid, nlevels)];
    const float of = offset_b[min(tid, K * 3 - 1)];
    if (tid < K) { tb.lvl_joint[tid] = lj; tb.parents[tid] = pa; }
    if (tid <= nlevels) tb.lvl_start[tid] = ls;
    if (tid < K * 3) tb.offset[tid] = of;
}
__device__ __forceinline__ void fk_levels(const FkTables& tb, int nlevels, const float* Rl, float* Rg, float* pos,
                                          const float* __restrict__ root, int ldr, int K, int B, int b, int smp, int lt, int per) {
    const int rootj = tb.lvl_joint[0];
    if (smp >= 0) {                        // strided like the level loops: a sample may own fewer than 12 threads (S > 21 at 256 threads)
        const int i = smp;
        for (int u = lt; u < 12; u += per) {
            if (u < 9) Rg[(i * K + rootj) * 9 + u] = Rl[(i * K + rootj) * 9 + u];
            else pos[(i * K + rootj) * 3 + (u - 9)] = root[((size_t)(i * B + b)) * ldr + (u - 9)];
        }
    }
    __syncthreads();
    for (int l = 1; l <= nlevels; ++l) {
        if (smp >= 0) {
            const int i = smp;
            if (l < nlevels) {             // global rotations of level l: G = P(parent) . L
                const int j0 = tb.lvl_start[l], nj = tb.lvl_start[l + 1] - j0;
                for (int u = lt; u < nj * 9; u += per) {
                    const int idx = tb.lvl_joint[j0 + u / 9], e = u % 9, r = e / 3, c = e % 3;
                    const float* P = Rg + (i * K + tb.parents[idx]) * 9; const float* L = Rl + (i * K + idx) * 9;
                    Rg[(i * K + idx) * 9 + e] = (P[r * 3] * L[c] + P[r * 3 + 1] * L[3 + c]) + P[r * 3 + 2] * L[6 + c];
                }
            }
            if (l >= 2) {                  // positions of level l - 1: p = G . offset + p(parent)
                const int j0 = tb.lvl_start[l - 1], nj = tb.lvl_start[l] - j0;
                for (int u = lt; u < nj * 3; u += per) {
                    const int idx = tb.lvl_joint[j0 + u / 3], r = u % 3;
                    const float* G = Rg + (i * K + idx) * 9; const float* of = tb.offset + idx * 3;
                    pos[(i * K + idx) * 3 + r] = ((G[r * 3] * of[0] + G[r * 3 + 1] * of[1]) + G[r * 3 + 2] * of[2]) + pos[(i * K + tb.parents[idx]) * 3 + r];
                }
            }
        }
        __syncthreads();
    }
}

// fk_levels for ONE sample by ONE wave (the persistent rollout's middle workgroup, round 6): the same per-element arithmetic, the level
// loop ordered by the wave's own LDS queue (a wave's LDS operations complete in order) instead of nlevels + 1 workgroup barriers.
__device__ __forceinline__ void fk_levels_wave(const FkTables& tb, int nlevels, const float* Rl, float* Rg, float* pos, const float* root, int K, int lane) {
    const int rootj = tb.lvl_joint[0];
    if (lane < 12) {
        if (lane < 9) Rg[rootj * 9 + lane] = Rl[rootj * 9 + lane];
        else pos[rootj * 3 + (lane - 9)] = root[lane - 9];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int l = 1; l <= nlevels; ++l) {
        if (l < nlevels) {
            const int j0 = tb.lvl_start[l], nj = tb.lvl_start[l + 1] - j0;
            for (int u = lane; u < nj * 9; u += 64) {
                const int idx = tb.lvl_joint[j0 + u / 9], e = u % 9, r = e / 3, c = e % 3;
                const float* P = Rg + tb.parents[idx] * 9; const float* L = Rl + idx * 9;
                Rg[idx * 9 + e] = (P[r * 3] * L[c] + P[r * 3 + 1] * L[3 + c]) + P[r * 3 + 2] * L[6 + c];
            }
        }
        if (l >= 2) {
            const int j0 = tb.lvl_start[l - 1], nj = tb.lvl_start[l] - j0;
            for (int u = lane; u < nj * 3; u += 64) {
                const int idx = tb.lvl_joint[j0 + u / 3], r = u % 3;
                const float* G = Rg + idx * 9; const float* of = tb.offset + idx * 3;
                pos[idx * 3 + r] = ((G[r * 3] * of[0] + G[r * 3 + 1] * of[1]) + G[r * 3 + 2] * of[2]) + pos[tb.parents[idx] * 3 + r];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

__global__ __launch_bounds__(256) void fk_kernel(FkArgs a) {
    extern __shared__ float sm[];
    const int K = a.K, S = a.S, b = blockIdx.x;
    float* Rl = sm;                    // [S][K][9]
    float* Rg = Rl + S * K * 9;        // [S][K][9]
    float* pos = Rg + S * K * 9;       // [S][K][3]
    float* dist = pos + S * K * 3;     // [S] + [S][K]
    __shared__ float red[256];
    __shared__ int best_s;
    __shared__ FkTables tb;
    fk_tables_load(tb, a.lvl_joint, a.lvl_start, a.parents, a.offset + (size_t)b * K * 3, K, a.nlevels, threadIdx.x);
    // 6-D -> rotation (geo_utils.py:56-78)
    for (int t = threadIdx.x; t < S * K; t += 256) {
        const int i = t / K, k = t % K;
        const float* p = a.rot + ((size_t)(i * a.B + b)) * 6 * K + k * 6;
        float x0 = p[0], x1 = p[1], x2 = p[2], y0 = p[3], y1 = p[4], y2 = p[5];
        float nx = sqrtf((x0 * x0 + x1 * x1) + x2 * x2) + 1e-10f;
        x0 /= nx; x1 /= nx; x2 /= nx;
        float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
        float nz = sqrtf((z0 * z0 + z1 * z1) + z2 * z2) + 1e-10f;
        z0 /= nz; z1 /= nz; z2 /= nz;
        float yy0 = z1 * x2 - z2 * x1, yy1 = z2 * x0 - z0 * x2, yy2 = z0 * x1 - z1 * x0;
        float* R = Rl + (i * K + k) * 9;
        R[0] = x0; R[1] = yy0; R[2] = z0; R[3] = x1; R[4] = yy1; R[5] = z1; R[6] = x2; R[7] = yy2; R[8] = z2;
    }
    __syncthreads();
    // chain along the tree (geo_utils.py:16-25, hsvrnn_bvh.py:272-277), LEVEL-PARALLEL: a joint only needs its parent, so all joints of
    // one tree depth (of all S samples) are computed together - depth ~4-6 dependent steps instead of K - 1 = 23 on one thread per
    // sample.  Step l computes the global rotations of level l and the positions of level l - 1 (which need that level's rotation,
    // finished one barrier ago).  Per element the arithmetic is the serial chain's.
    {
        const int per = S <= 256 ? 256 / S : 1, smp = (int)threadIdx.x / per;
        fk_levels(tb, a.nlevels, Rl, Rg, pos, a.root, a.ldr, K, a.B, b, smp < S ? smp : -1, (int)threadIdx.x % per, per);
    }
    // squared distance to the observed keypoints: per (sample, joint) term, then thread i adds its sample's K terms in joint order
    float* dk = dist + S;              // [S][K]
    if (a.obs) {
        const float* ob = a.obs + (size_t)b * a.ldobs;
        for (int t = threadIdx.x; t < S * K; t += 256) {
            const int i = t / K, k = t % K;
            const float* rt = a.root + ((size_t)(i * a.B + b)) * a.ldr;
            const float u0 = ob[k * 4] - pos[(i * K + k) * 3], u1 = ob[k * 4 + 1] - pos[(i * K + k) * 3 + 1], u2 = ob[k * 4 + 2] - pos[(i * K + k) * 3 + 2];
            const float u3 = ob[k * 4 + 3] - (rt[3 + k] + 1.0f) * 0.5f;
            dk[t] = ((u0 * u0 + u1 * u1) + u2 * u2) + u3 * u3;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < S) {
        float d = 0.f;
        if (a.obs) for (int k = 0; k < K; ++k) d += dk[threadIdx.x * K + k];
        dist[threadIdx.x] = d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int bi = 0; float bd = dist[0];
        for (int i = 1; i < S; ++i) if (dist[i] < bd) { bd = dist[i]; bi = i; }
        best_s = bi;
        if (a.best) a.best[(size_t)b * a.ldbest] = bi;
        if (a.rec) a.rec[(size_t)b * a.ldstat] = bd;
    }
    __syncthreads();
    const int bi = best_s;
    const float* rt = a.root + ((size_t)(bi * a.B + b)) * a.ldr;
    for (int t = threadIdx.x; t < K * 4; t += 256) {
        const int k = t >> 2, c = t & 3;
        a.out_kp[(size_t)b * a.ldkp + t] = c < 3 ? pos[(bi * K + k) * 3 + c] : (rt[3 + k] + 1.0f) * 0.5f;
    }
    if (a.out_z) for (int t = threadIdx.x; t < a.Z; t += 256) a.out_z[(size_t)b * a.ldz + t] = a.z[((size_t)(bi * a.B + b)) * a.Z + t];
    if (a.out_R) for (int t = threadIdx.x; t < K * 9; t += 256) a.out_R[(size_t)b * a.ldR + t] = Rg[bi * K * 9 + t];
    if (a.t_hr) {
        const size_t sb = (size_t)(bi * a.B + b);
        for (int t = threadIdx.x; t < 128; t += 256) { a.t_hr[(size_t)b * 128 + t] = a.hr[sb * 128 + t]; a.t_hj[(size_t)b * 128 + t] = a.hj[sb * 128 + t]; }
        for (int t = threadIdx.x; t < 3 + K; t += 256) a.t_raw[(size_t)b * (3 + K) + t] = rt[t];
        for (int t = threadIdx.x; t < 6 * K; t += 256) a.t_rot6[(size_t)b * 6 * K + t] = a.rot[sb * 6 * K + t];
        for (int t = threadIdx.x; t < 9 * K; t += 256) { a.t_Rl[(size_t)b * 9 * K + t] = Rl[bi * K * 9 + t]; a.t_Rg[(size_t)b * 9 * K + t] = Rg[bi * K * 9 + t]; }
        for (int t = threadIdx.x; t < a.Z; t += 256) a.t_eps[(size_t)b * a.Z + t] = a.eps[sb * a.Z + t];
    }
    if (a.kl) {
        float v = 0.f;
        for (int t = threadIdx.x; t < a.Z; t += 256) {
            const size_t o = (size_t)b * a.Z + t;
            const float ratio = a.qsig[o] / a.psig[o];
            const float vr = ratio * ratio;
            const float dm = (a.qmu[o] - a.pmu[o]) / a.psig[o];
            v += 0.5f * (((vr + dm * dm) - 1.0f) - logf(vr));
        }
        red[threadIdx.x] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
        if (threadIdx.x == 0) a.kl[(size_t)b * a.ldstat] = red[0];
    }
}

// ---- prior step, middle phases in ONE workgroup per sample (rollout latency, vis_generation.py:97-127 / hsvrnn_bvh.py:208-225) ---
// A prior step is a chain of dependent phases: h-phase -> distribution + sample -> decoder hidden layers -> heads -> forward
// kinematics -> GRU.  The first and the last stream megabytes of weights and want the whole chip; the four in between touch
// 128 + 128 + 87 KB of weights and a few hundred floats of state per sample: as separate launches each of them is a dispatch
// round trip (~6 us) around ~2 us of work.  Here ONE 1024-thread workgroup per batch element runs the four back to back with
// workgroup barriers, state in LDS: a step is 3 dependent launches instead of 6.
// What makes one workgroup fast enough (round 2's version, 8 waves x 85 rows fetched a few at a time, took ~45 us: every batch of
// rows was a dependent L2 round trip): the weights do not depend on the data, so EVERY weight row the workgroup will need in the
// three matrix phases is requested in the first instructions of the kernel - 22 16-byte loads per lane, 88 registers - and the
// phases then run from registers; a wavefront holds TWO rows per load (a 128-column row is 32 lanes x 4 floats: lanes 0-31 one row,
// lanes 32-63 the other) and the per-row bias / noise / h-half values are preloaded the same way.  Arithmetic per row is
// dot_seg's / wave_reduce's for n = 128 (((w0 x0 + w1 x1) + w2 x2) + w3 x3 per lane, xor-shuffle tree 16..1 - the tree's first step,
// offset 32, adds the zeros of the idle half there): the fused step is bit-identical to the six-launch step.
struct MidArgs {
    const float *hid_prior, *rh, *jh, *eps, *offset;            // [B][128] x3, [B][Z], [B][K][3]
    const float *w_p2, *b_p2, *w_root0, *w_joint0, *w_root2, *b_root2, *w_joint2, *b_joint2;
    const int32_t *order, *parents, *lvl_joint, *lvl_start; int nlevels;
    float *out_kp, *out_z; int ldkp, ldz;
    int B, K, Z, H;
};

__device__ __forceinline__ float half_reduce(float v) {          // sum over the 32 lanes of this lane's half; every lane gets it
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += nm_sx(v, off);
    return v;
}
__device__ __forceinline__ float dot4(const f32x4& w, const f32x4& x) { return ((w[0] * x[0] + w[1] * x[1]) + w[2] * x[2]) + w[3] * x[3]; }

// ---- NV sums over WIDTH lanes in one pass (round 6) -------------------------------------------------------------------------------------
// A butterfly per value costs log2(WIDTH) exchange + add steps for EACH of the NV values a lane holds.  Reduce-scatter: at distance D the
// lanes with bit D clear keep the lower half of their live values and take the partner's partial sums of those, the lanes with bit D set
// the upper half - half as many values are alive after every step, NV - 1 + log2(WIDTH / NV) exchanges in all instead of NV log2(WIDTH).
// For each value the additions are the butterfly's, in the butterfly's order (distance WIDTH / 2 first, 1 last; a + b = b + a bit for
// bit), so the result is the butterfly's bit for bit.  On return v[0] of lane l is the total of value rs_index<NV, WIDTH>(l); all
// WIDTH / NV lanes that share an index hold it.
template <int NV, int WIDTH> __device__ __forceinline__ int rs_index(int lane) {
    int idx = 0, n = NV;
#pragma unroll
    for (int d = WIDTH / 2; d > 0 && n > 1; d >>= 1) { n >>= 1; if (lane & d) idx += n; }
    return idx;
}
template <int N, int D, int NV> __device__ __forceinline__ void rs_step(float (&v)[NV], int lane) {
    if constexpr (D >= 1) {
        if constexpr (N > 1) {
            constexpr int HALF = N / 2;
            const bool up = (lane & D) != 0;
#pragma unroll
            for (int k = 0; k < HALF; ++k) {
                const float keep = up ? v[k + HALF] : v[k], send = up ? v[k] : v[k + HALF];
                v[k] = keep + nm_sxc<D>(send);
            }
            rs_step<HALF, D / 2, NV>(v, lane);
        } else {
            v[0] += nm_sxc<D>(v[0]);
            rs_step<1, D / 2, NV>(v, lane);
        }
    }
}
template <int NV, int WIDTH> __device__ __forceinline__ float reduce_scatter(float (&v)[NV], int lane) {
    rs_step<NV, WIDTH / 2, NV>(v, lane);
    return v[0];
}

#define MID_PAIRS 8        // row pairs per wavefront and phase: 16 waves x 8 pairs x 2 rows = 256 rows
__global__ __launch_bounds__(1024) void vrnn_prior_mid_kernel(MidArgs a) {
    __shared__ __attribute__((aligned(16))) float s_z[128], s_hr[128], s_hj[128], s_root[36], s_rot[192];
    __shared__ float s_Rl[32 * 9], s_Rg[32 * 9], s_pos[32 * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31, b = blockIdx.x;
    const int K = a.K, Z = a.Z, H = a.H, R0 = 3 + K, J6 = 6 * K, rows_c = R0 + J6;
    // ---- every weight row of the three matrix phases, requested up front ----------------------------------------------
    f32x4 wa[MID_PAIRS], wb[MID_PAIRS], wc[MID_PAIRS];
#pragma unroll
    for (int u = 0; u < MID_PAIRS; ++u) {
        const int p = wave * MID_PAIRS + u;                      // A: lower half row p (mu), upper half row p + Z (raw std)
        wa[u] = *reinterpret_cast<const f32x4*>(a.w_p2 + (size_t)(p + half * Z) * 128 + l32 * 4);
        // B: lower half root0 row p, upper half joint0 row p (the z-columns H .. H+Z of the first decoder layers)
        wb[u] = *reinterpret_cast<const f32x4*>((half ? a.w_joint0 : a.w_root0) + (size_t)p * (H + Z) + H + l32 * 4);
        // C: rows [0, R0) root / intensity head, [R0, R0 + J6) 6-D rotations (rows beyond: clamped, never stored).  The ADDRESS is
        // selected, the load is unconditional: a conditional load is a branch, and hipcc drains every outstanding load at its join
        const int r = min(p * 2 + half, rows_c - 1);
        const float* pc = r < R0 ? a.w_root2 + (size_t)r * 128 : a.w_joint2 + (size_t)(r - R0) * 128;
        wc[u] = *reinterpret_cast<const f32x4*>(pc + l32 * 4);
    }
    const f32x4 xh = *reinterpret_cast<const f32x4*>(a.hid_prior + (size_t)b * 128 + l32 * 4);
    // per-row scalars: lane u (< MID_PAIRS) of each half finishes pair u of its wave (unconditional loads from clamped / selected
    // addresses for the same reason; lanes >= MID_PAIRS read pair 0's values and never use them)
    const int pl = wave * MID_PAIRS + (l32 < MID_PAIRS ? l32 : 0);
    const float bias_a = a.b_p2[pl], bias_a2 = a.b_p2[pl + Z], epsv = a.eps[(size_t)b * Z + pl];
    const float add_b = (half ? a.jh : a.rh)[(size_t)b * 128 + pl];
    const int rl = min(pl * 2 + half, rows_c - 1);
    const float bias_c = *(rl < R0 ? a.b_root2 + rl : a.b_joint2 + (rl - R0));
    __shared__ FkTables tb;
    fk_tables_load(tb, a.lvl_joint, a.lvl_start, a.parents, a.offset + (size_t)b * K * 3, K, a.nlevels, tid);
    // ---- A. prior distribution parameters and the sample --------------------------------------------------------------
    {
        float mine = 0.f, other = 0.f;
#pragma unroll
        for (int u = 0; u < MID_PAIRS; ++u) {
            const float v = half_reduce(dot4(wa[u], xh));
            const float o = nm_sx(v, 32);
            if (l32 == u) { mine = v; other = o; }
        }
        if (!half && l32 < MID_PAIRS) {
            const int p = wave * MID_PAIRS + l32;
            const float mu = mine + bias_a, sraw = other + bias_a2;
            const float sg = softplus(sraw) + 1e-4f;
            const float z = mu + epsv * sg;
            s_z[p] = z;
            a.out_z[(size_t)b * a.ldz + p] = z;
        }
    }
    __syncthreads();
    // ---- B. decoder hidden layers: z-half of the first layers + the h-half (and bias) the h-phase left in rh / jh -------
    {
        const f32x4 xz = *reinterpret_cast<const f32x4*>(s_z + l32 * 4);
        float mine = 0.f;
#pragma unroll
        for (int u = 0; u < MID_PAIRS; ++u) {
            const float v = half_reduce(dot4(wb[u], xz));
            if (l32 == u) mine = v;
        }
        if (l32 < MID_PAIRS) (half ? s_hj : s_hr)[wave * MID_PAIRS + l32] = lrelu(mine + add_b, 0.01f);
    }
    __syncthreads();
    // ---- C. heads: root / intensity (tanh) and 6-D rotations ---------------------------------------------------------------
    {
        const f32x4 xr = *reinterpret_cast<const f32x4*>(s_hr + l32 * 4);
        const f32x4 xj = *reinterpret_cast<const f32x4*>(s_hj + l32 * 4);
        float mine = 0.f;
#pragma unroll
        for (int u = 0; u < MID_PAIRS; ++u) {
            const int r = (wave * MID_PAIRS + u) * 2 + half;
            const float v = half_reduce(dot4(wc[u], r < R0 ? xr : xj));
            if (l32 == u) mine = v;
        }
        if (l32 < MID_PAIRS) {
            const int r = (wave * MID_PAIRS + l32) * 2 + half;
            if (r < R0) s_root[r] = tanhf(mine + bias_c);
            else if (r < rows_c) s_rot[r - R0] = mine + bias_c;
        }
    }
    __syncthreads();
    // ---- D. forward kinematics (fk_kernel with S = 1, no observation) ----------------------------------------------------
    if (tid < K) {
        const float* p = s_rot + tid * 6;
        float x0 = p[0], x1 = p[1], x2 = p[2], y0 = p[3], y1 = p[4], y2 = p[5];
        float nx = sqrtf((x0 * x0 + x1 * x1) + x2 * x2) + 1e-10f;
        x0 /= nx; x1 /= nx; x2 /= nx;
        float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
        float nz = sqrtf((z0 * z0 + z1 * z1) + z2 * z2) + 1e-10f;
        z0 /= nz; z1 /= nz; z2 /= nz;
        float yy0 = z1 * x2 - z2 * x1, yy1 = z2 * x0 - z0 * x2, yy2 = z0 * x1 - z1 * x0;
        float* R = s_Rl + tid * 9;
        R[0] = x0; R[1] = yy0; R[2] = z0; R[3] = x1; R[4] = yy1; R[5] = z1; R[6] = x2; R[7] = yy2; R[8] = z2;
    }
    __syncthreads();
    // (root row read from LDS: B = 1, b = 0 in fk_levels' indexing of `root`)
    fk_levels(tb, a.nlevels, s_Rl, s_Rg, s_pos, s_root, 36, K, 1, 0, 0, tid, 1024);
    if (tid < K * 4) {
        const int k = tid >> 2, c = tid & 3;
        a.out_kp[(size_t)b * a.ldkp + tid] = c < 3 ? s_pos[k * 3 + c] : (s_root[3 + k] + 1.0f) * 0.5f;
    }
}

// ---- the prior steps of a rollout as ONE persistent launch (BASELINE north_star: "one kernel per timestep"; here: one per rollout) -----
// vis_generation.py:97-127 / hsvrnn_bvh.py:208-225.  A prior step is h-phase -> middle phases -> GRU, three all-to-all hand-offs per
// step.  As launches (the default until round 4: 3 per step inside a captured graph, 19.9-21 us per step at B = 1) every launch
// re-streams its weights from L2 / Infinity Cache - 3.9 MB (h-phase), 343 KB (middle) and 1.4 MB (GRU) per step - and pays a dispatch.
// The weights do not depend on the step, so here NW = 64 worker workgroups and B <= 4 middle workgroups (512 threads each, one per CU,
// all resident: 68 of 256 CUs) load their weight rows into REGISTERS (the heads' 87 KB into LDS) once and then run the whole chain:
//   workers: wave gw (of 512) owns h-phase rows gw + 512 i (prior0 | root0_h | joint0_h | W_hh: 1920 rows x 512) and GRU unit j = gw;
//   middle workgroup b: vrnn_prior_mid_kernel's four phases for batch element b.
// Hand-offs as DATA-TAGGED GRANULES (MI355X_MICROARCH.md, persistent-kernel price list: "granules for latency", handoff-1to1 0.8-1.0 us
// against 1.3-2.2 us for payload + flag): every value handed over is one naturally aligned 8-byte {value, step tag} written by ONE
// sc1 (agent-scope, write-through) store; a consumer loads the granules it needs with sc1 loads (they bypass the CU's L1, which another
// CU's stores never refresh) and repeats the loads until every tag is the step's - no counters, no fences, no barrier between
// workgroups.  A first version with three counters per step (sc1 payload, drained, one atomic add per workgroup, sc1 poll, sc1 loads)
// measured 21.2 us per step at B = 1 - no better than three launches: each hop was a store drain, an atomic, a poll and a dependent
// load, ~7 us (profiles/r04_rollout_ab.txt).  The granule buffers are zeroed before the launch (tag 0 = never written); a buffer is
// rewritten only after every reader of its previous contents has produced the values the writer itself waited for (the dependency
// chain of the step), so an exact tag match is unambiguous.  Every spin is bounded: after NM_CHAIN_SPIN polls a wave sets the abort word
// and bit 1 of the context's status word, every workgroup leaves, and the next library call fails - a rollout must never hang the device.
// Arithmetic per output row is the row kernels' (dot_seg for n = 512 / 96 + 128, wave_reduce's xor tree, the same epilogues): outputs are
// bit-identical to the launch-per-phase step (tools/diag_chain.py, tests/test_network_gpu.py::test_config5_rollout64).
#define NM_CHAIN_NW 64                 // worker workgroups of 8 waves: 512 waves; 4 h-phase rows (1920 / 512, the last waves fewer) and ONE GRU unit (H = 512) per wave
static long long* g_chain_stamps = nullptr;      // set by the undeclared diagnostic export nm_diag_set_chain_stamps (tools/diag_chain_stamps.py)
#define NM_CHAIN_NWX 24                // worker workgroups of the one-XCD form: 192 waves, 10 h-phase rows and up to 3 GRU units each; + B <= 8 middle workgroups <= the 32 CUs of an XCD
#define NM_CHAIN_T 512                 // threads per workgroup: two waves per SIMD, 256 registers each (the middle role keeps 2 x 16 weight quads per lane)
#define NM_CHAIN_PAIRS 16              // row pairs per wave and matrix phase of the middle role: 8 waves x 16 pairs x 2 rows = 256 rows
#define NM_CHAIN_SPIN (1 << 20)
typedef unsigned long long nm_gran;    // {float value (low dword), step tag (high dword)}
struct ChainArgs {
    const float *w_prior0, *b_prior0, *w_root0, *b_root0, *w_joint0, *b_joint0, *w_hh, *b_hh, *w_ih, *b_ih;
    MidArgs mid;                        // weights / tree / offset of the middle phases (its data pointers are not used)
    const float* h0; int ldh0;          // [B][ldh0] state before the first step (written before the launch: plain loads)
    const float* eps;                   // [T][B][Z]
    float* out_kp; int ldkp;            // [B][ldkp], step t at column t * 4K          (the call's output, plain stores)
    float* h_out;                       // [B][H] state after the last step             (plain stores)
    nm_gran *g_hid, *g_rh, *g_jh;       // [B][128] x 3   h-phase -> middle
    nm_gran* g_gh;                      // [B][3H]        h-phase -> GRU
    nm_gran* g_kpz;                     // [B][4K + Z]    middle -> GRU
    nm_gran* g_h[2];                    // [B][H] x 2     GRU of step t -> h-phase / GRU of step t + 1 (buffer t & 1)
    unsigned* abort;                    // zero before the launch; 1: some wave timed out
    unsigned* status;                   // the context's status word: bit 1 = a spin timed out
    int B, T, K, Z, H;
    int backoff;                        // s_sleep(2) units between polls
    int spin_limit;                     // polls before a wave gives up (NM355_CHAIN_SPIN; default NM_CHAIN_SPIN)
    long long* stamps;                  // diagnostic phase stamps (null in product calls)
    int* ctl;                           // one-XCD form: [0] chosen XCD + 1, [1] roles taken, [2] go (1) / no-go (2), [3] workgroups that completed; zero before the launches (null: no one-XCD launch in front)
    int force_nogo;                     // test hook: role 0 of the one-XCD form always decides no-go
    int wg_poll;                        // 1: one wave per worker workgroup polls, LDS + a workgroup barrier hand the values to the other seven (NM355_CHAIN_WGPOLL)
};

__device__ __forceinline__ void gran_store(nm_gran* p, float v, unsigned tag) {
    __hip_atomic_store(p, (nm_gran)__builtin_bit_cast(unsigned, v) | ((nm_gran)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One-XCD form (round 6): producer and consumer run on the SAME XCD (checked against HW_REG_XCC_ID at run time, never assumed from the
// launch order), whose L2 is their point of coherence: a PLAIN 8-byte store keeps the line in that L2 and an sc1 load (bypasses the
// reader's L1, L2-served) sees it 0.34 us later on an idle chip, against 0.76 us for the write-through sc1 store + fabric read of the
// cross-XCD form (tools/calib/hop_latency.hip, profiles/r06_hop_latency.txt; a plain store is never seen from another XCD - same file).
template <bool XCD> __device__ __forceinline__ void gran_store_t(nm_gran* p, float v, unsigned tag) {
    if constexpr (XCD) {
        const nm_gran g = (nm_gran)__builtin_bit_cast(unsigned, v) | ((nm_gran)tag << 32);
        asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(g) : "memory");
    } else gran_store(p, v, tag);
}
// (round-5 form, NM_GRAN_DWORDX2 == 0 only) HARDWARE ASSUMPTION (stated): a naturally aligned 16-byte global load observes each of its two naturally aligned 8-byte
// halves whole - a granule {value, tag} written by ONE 8-byte store is never seen as {new tag, old value}.  The ISA text promises
// single-copy atomicity for naturally aligned accesses up to 8 bytes and says nothing of 16-byte ones; on gfx950 a dwordx4 load of a
// 16-byte-aligned address is served from one 64-byte L2 sector in one request and the writer's dwordx2 store updates its 8 bytes of
// that sector in one piece, so a half cannot tear.  What the kernel relies on is exactly that, no more: the two halves MAY come from
// different moments (one granule of the step, the other still of an older step - the poll then repeats).  Checked, not proved:
// tests/test_network_gpu.py::test_persistent_rollout_long_chain_stays_bit_identical (6 000 steps at B = 4 beside a second context
// hammering the device: one torn granule would change every later step) and tools/stress_identity.py.  NM355_VRNN_CHAIN=0 is the
// path that does not depend on it.
// A lane's granule loads of one poll as ONE asm statement: all loads issued (16-byte sc1 loads = two neighbouring granules, 8-byte = one),
// then one wait - the compiler must not touch a destination between its load and the wait (an untracked load's register is stale until
// then), hence a single statement with early-clobber outputs
#ifndef NM_GRAN_DWORDX2
#define NM_GRAN_DWORDX2 1
#endif
#if NM_GRAN_DWORDX2
// Round 6 (verdict r5 item 11): every granule is read by an 8-byte load of its own - the access size the ISA's single-copy atomicity
// covers - instead of 16-byte loads of granule PAIRS.  A pair is two loads at offset 0 / 8 from the same address register; the
// returned register layout is the pair loads' (value, tag, value, tag).  Cost against the pair loads (same call, one device):
// profiles/r06_granule_loads_ab.txt.  -DNM_GRAN_DWORDX2=0 (make x4: libnm355_x4.so) builds the round-5 loads for that A/B.
#define NM_GL2(D, A, OFF) "global_load_dwordx2 " D ", " A ", off offset:" OFF " sc1\n\t"
__device__ __forceinline__ f32x4 gran_join(const nm_f32x2& lo, const nm_f32x2& hi) { return f32x4{lo[0], lo[1], hi[0], hi[1]}; }
__device__ __forceinline__ void gran_ld_2q1(const nm_gran* p0, const nm_gran* p1, const nm_gran* s0, f32x4& a0, f32x4& a1, nm_f32x2& b0) {
    nm_f32x2 t0, t1, t2, t3;
    asm volatile(NM_GL2("%0", "%5", "0") NM_GL2("%1", "%5", "8") NM_GL2("%2", "%6", "0") NM_GL2("%3", "%6", "8") NM_GL2("%4", "%7", "0") "s_waitcnt vmcnt(0)"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(b0) : "v"(p0), "v"(p1), "v"(s0) : "memory");
    a0 = gran_join(t0, t1); a1 = gran_join(t2, t3);
}
__device__ __forceinline__ void gran_ld_4q1(const nm_gran* p0, const nm_gran* p1, const nm_gran* p2, const nm_gran* p3, const nm_gran* s0,
                                            f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, nm_f32x2& b0) {
    nm_f32x2 t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(NM_GL2("%0", "%9", "0") NM_GL2("%1", "%9", "8") NM_GL2("%2", "%10", "0") NM_GL2("%3", "%10", "8") NM_GL2("%4", "%11", "0") NM_GL2("%5", "%11", "8")
                 NM_GL2("%6", "%12", "0") NM_GL2("%7", "%12", "8") NM_GL2("%8", "%13", "0") "s_waitcnt vmcnt(0)"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(b0)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(s0) : "memory");
    a0 = gran_join(t0, t1); a1 = gran_join(t2, t3); a2 = gran_join(t4, t5); a3 = gran_join(t6, t7);
}
__device__ __forceinline__ void gran_ld_4q4(const nm_gran* p0, const nm_gran* p1, const nm_gran* p2, const nm_gran* p3, const nm_gran* s0,
                                            const nm_gran* s1, const nm_gran* s2, const nm_gran* s3, f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3,
                                            nm_f32x2& b0, nm_f32x2& b1, nm_f32x2& b2, nm_f32x2& b3) {
    nm_f32x2 t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(NM_GL2("%0", "%12", "0") NM_GL2("%1", "%12", "8") NM_GL2("%2", "%13", "0") NM_GL2("%3", "%13", "8") NM_GL2("%4", "%14", "0") NM_GL2("%5", "%14", "8")
                 NM_GL2("%6", "%15", "0") NM_GL2("%7", "%15", "8") NM_GL2("%8", "%16", "0") NM_GL2("%9", "%17", "0") NM_GL2("%10", "%18", "0") NM_GL2("%11", "%19", "0")
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(s0), "v"(s1), "v"(s2), "v"(s3) : "memory");
    a0 = gran_join(t0, t1); a1 = gran_join(t2, t3); a2 = gran_join(t4, t5); a3 = gran_join(t6, t7);
}
__device__ __forceinline__ void gran_ld_5q(const nm_gran* p0, const nm_gran* p1, const nm_gran* p2, const nm_gran* p3, const nm_gran* p4,
                                           f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4) {
    nm_f32x2 t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
    asm volatile(NM_GL2("%0", "%10", "0") NM_GL2("%1", "%10", "8") NM_GL2("%2", "%11", "0") NM_GL2("%3", "%11", "8") NM_GL2("%4", "%12", "0") NM_GL2("%5", "%12", "8")
                 NM_GL2("%6", "%13", "0") NM_GL2("%7", "%13", "8") NM_GL2("%8", "%14", "0") NM_GL2("%9", "%14", "8") "s_waitcnt vmcnt(0)"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8), "=&v"(t9)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4) : "memory");
    a0 = gran_join(t0, t1); a1 = gran_join(t2, t3); a2 = gran_join(t4, t5); a3 = gran_join(t6, t7); a4 = gran_join(t8, t9);
}
#else
__device__ __forceinline__ void gran_ld_5q(const nm_gran* p0, const nm_gran* p1, const nm_gran* p2, const nm_gran* p3, const nm_gran* p4,
                                           f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4) {
    asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                 "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4) : "memory");
}
__device__ __forceinline__ void gran_ld_2q1(const nm_gran* p0, const nm_gran* p1, const nm_gran* s0, f32x4& a0, f32x4& a1, nm_f32x2& b0) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx2 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0) : "v"(p0), "v"(p1), "v"(s0) : "memory");
}
__device__ __forceinline__ void gran_ld_4q1(const nm_gran* p0, const nm_gran* p1, const nm_gran* p2, const nm_gran* p3, const nm_gran* s0,
                                            f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, nm_f32x2& b0) {
    asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                 "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx2 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(s0) : "memory");
}
__device__ __forceinline__ void gran_ld_4q4(const nm_gran* p0, const nm_gran* p1, const nm_gran* p2, const nm_gran* p3, const nm_gran* s0,
                                            const nm_gran* s1, const nm_gran* s2, const nm_gran* s3, f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3,
                                            nm_f32x2& b0, nm_f32x2& b1, nm_f32x2& b2, nm_f32x2& b3) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx2 %4, %12, off sc1\n\tglobal_load_dwordx2 %5, %13, off sc1\n\t"
                 "global_load_dwordx2 %6, %14, off sc1\n\tglobal_load_dwordx2 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(s0), "v"(s1), "v"(s2), "v"(s3) : "memory");
}
#endif
// polls until `ok` holds in every lane of the wave (the body re-issues the lane's loads and re-evaluates ok); false: aborted
#define NM_CHAIN_POLL(LOADS, OKEXPR)                                                                                  \
    {                                                                                                                \
        int spins_ = 0; bool alive_ = true;                                                                          \
        for (;;) {                                                                                                   \
            LOADS;                                                                                                   \
            if (__all(OKEXPR)) break;                                                                                \
            if ((++spins_ & 63) == 0) {                                                                              \
                if (__hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { alive_ = false; break; }   \
                if (spins_ > a.spin_limit) {                                                                         \
                    __hip_atomic_store(a.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                       \
                    __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                   \
                    alive_ = false; break;                                                                           \
                }                                                                                                    \
            }                                                                                                        \
            for (int sl_ = 0; sl_ < a.backoff; ++sl_) __builtin_amdgcn_s_sleep(2);                                   \
        }                                                                                                            \
        if (!alive_) { wave_alive = false; }                                                                         \
    }

// diagnostic stamps (tools/diag_chain_stamps.py; a.stamps is null in every product call): s_memrealtime (100 MHz) of thread 0 of worker
// workgroup 0 (role 0) and of the first middle workgroup (role 1) at the phase boundaries of step t: [role][t < 128][8]
#define NM_STAMP(ROLE, K) do { if (a.stamps && tid == 0 && t < 128 && ((ROLE) == 0 ? wg == 0 : wg == NW)) a.stamps[((ROLE) * 128 + t) * 8 + (K)] = wall_clock64(); } while (0)
template <bool XCD>
__global__ __launch_bounds__(NM_CHAIN_T) void vrnn_prior_chain_kernel(ChainArgs a) {
    __shared__ int s_dead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, T = a.T, K = a.K, Z = a.Z, H = a.H, S4 = 4 * K;
    constexpr int NW = XCD ? NM_CHAIN_NWX : NM_CHAIN_NW;
    bool wave_alive = true;
    if (tid == 0) s_dead = 0;
    int wg = (int)blockIdx.x;                  // logical workgroup: [0, NW) workers, [NW, NW + B) the middle workgroup of batch element wg - NW
    if constexpr (XCD) {
        // ---- one-XCD form: 256 workgroups are launched (one per CU), each reads the XCD it runs on from HW_REG_XCC_ID; the first
        // arriver's XCD is THE XCD, its workgroups take the NW + B roles in arrival order and every other workgroup leaves at once.
        // Role 0 decides: all roles taken within ~30 us -> go; else no-go, nobody starts, ctl[3] never reaches NW + B and the
        // cross-XCD kernel enqueued behind this launch runs the rollout (a busy XCD costs two launches, never a wrong or missing result).
        __shared__ int s_role;
        if (tid == 0) {
            const int x = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) + 1;       // HW_REG_XCC_ID[3:0] + 1
            int chosen = atomicCAS(a.ctl, 0, x);
            if (chosen == 0) chosen = x;
            int r = -1;
            if (chosen == x) {
                r = atomicAdd(a.ctl + 1, 1);
                if (r >= NW + B) r = -1;
                else if (r == 0) {
                    const long long t0 = wall_clock64();
                    int n = 0;
                    while ((n = __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < NW + B && wall_clock64() - t0 < 3000) __builtin_amdgcn_s_sleep(8);
                    const bool go = n >= NW + B && !a.force_nogo;      // (force_nogo: NM355_CHAIN_XCD_NOGO, the test hook for "this XCD cannot seat the roles")
                    __hip_atomic_store(a.ctl + 2, go ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!go) r = -1;
                } else {
                    int go = 0, spins = 0;
                    while ((go = __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(8);
                    if (go != 1) r = -1;
                }
            }
            s_role = r;
        }
        __syncthreads();
        wg = s_role;
        if (wg < 0) return;
    } else if (a.ctl && __hip_atomic_load(a.ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NM_CHAIN_NWX + B) {
        return;                                  // the one-XCD launch in front of this one completed the rollout (uniform: every workgroup reads the same word)
    }
    if (wg >= NW) {
        // =========================== middle workgroup of batch element b ========================================================
        const MidArgs& m = a.mid;
        __shared__ __attribute__((aligned(16))) float s_z[128], s_hr[128], s_hj[128], s_root[36], s_rot[192];
        __shared__ float s_Rl[32 * 9], s_pos[32 * 3];               // (the path walk keeps the global rotations in registers)
        __shared__ FkTables tb;
        const int half = lane >> 5, l32 = lane & 31, b = wg - NW;
        const int R0 = 3 + K, J6 = 6 * K, rows_c = R0 + J6;
        // weights of the distribution (A) and decoder-hidden (B) phases in registers, those of the heads (C: 3 + 7K rows x 128) in LDS - all
        // three in registers (3 x 16 quads) overflow the 256 a 512-thread workgroup has per lane
        extern __shared__ f32x4 s_wc[];                                  // [rows_c][32 quads]
        f32x4 wa[NM_CHAIN_PAIRS], wb[NM_CHAIN_PAIRS];
#pragma unroll
        for (int u = 0; u < NM_CHAIN_PAIRS; ++u) {
            const int p = wave * NM_CHAIN_PAIRS + u;
            wa[u] = *reinterpret_cast<const f32x4*>(m.w_p2 + (size_t)(p + half * Z) * 128 + l32 * 4);
            wb[u] = *reinterpret_cast<const f32x4*>((half ? m.w_joint0 : m.w_root0) + (size_t)p * (H + Z) + H + l32 * 4);
            const int r = min(p * 2 + half, rows_c - 1);
            const float* pc = r < R0 ? m.w_root2 + (size_t)r * 128 : m.w_joint2 + (size_t)(r - R0) * 128;
            if (p * 2 + half < rows_c) s_wc[(p * 2 + half) * 32 + l32] = *reinterpret_cast<const f32x4*>(pc + l32 * 4);
        }
        // (round 6: the 16 row pairs of a wave are summed by reduce_scatter - lane l32 of a half ends up with pair rs_index(l32), two lanes per pair)
        const int pu = rs_index<NM_CHAIN_PAIRS, 32>(l32);
        const bool pown = (l32 & 1) == 0;                            // the lane of its pair that writes
        const int pl = wave * NM_CHAIN_PAIRS + pu;
        const float bias_a = m.b_p2[pl], bias_a2 = m.b_p2[pl + Z];
        const int rl = min(pl * 2 + half, rows_c - 1);
        const float bias_c = *(rl < R0 ? m.b_root2 + rl : m.b_joint2 + (rl - R0));
        fk_tables_load(tb, m.lvl_joint, m.lvl_start, m.parents, m.offset + (size_t)b * K * 3, K, m.nlevels, tid);
        __syncthreads();
        // (round 6) the kinematic chain per joint: s_path[j] = the joints from the root's child down to j (s_plen[j] of them; the root itself
        // has none).  Lane j of wave 0 walks its own path every step - G <- G L_c, p <- G offset_c + p from the root down, the very
        // products and sums the level-by-level evaluation makes for joint j and its ancestors, in the same order - so a step's kinematics
        // need no exchange between lanes beyond the local rotations in LDS and no barrier per tree level.
        __shared__ unsigned char s_path[32][32];
        __shared__ int s_plen[32];
        if (tid < K) {
            const int rootj = tb.lvl_joint[0];
            int n = 0, c = tid;
            unsigned char tmp[32];
            while (c != rootj && n < 32) { tmp[n++] = (unsigned char)c; c = tb.parents[c]; }
            for (int i = 0; i < n; ++i) s_path[tid][i] = tmp[n - 1 - i];
            s_plen[tid] = n;
        }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const unsigned tag = (unsigned)t + 1u;
            NM_STAMP(1, 0);
            if (a.stamps && tid == 0 && t < 128 && wg == NW) a.stamps[(128 + t) * 8 + 7] = clock64();       // (shader clock: the diagnostic derives the clock the chip runs this kernel at)
            const float epsv = a.eps[((size_t)t * B + b) * Z + pl];                       // (an input of the call: plain load)
            // this lane's inputs of the step: hid_prior[4 l32 .. +3] (two granule pairs) and rh / jh [pl]
            f32x4 g0, g1; nm_f32x2 ga;
            const nm_gran* ph = a.g_hid + (size_t)b * 128 + l32 * 4;
            const nm_gran* pa = (half ? a.g_jh : a.g_rh) + (size_t)b * 128 + pl;
            NM_CHAIN_POLL(gran_ld_2q1(ph, ph + 2, pa, g0, g1, ga),
                          nm_fbits(g0[1]) == tag && nm_fbits(g0[3]) == tag && nm_fbits(g1[1]) == tag && nm_fbits(g1[3]) == tag && nm_fbits(ga[1]) == tag);
            if (!wave_alive) s_dead = 1;
            NM_STAMP(1, 1);
            const f32x4 xh = {g0[0], g0[2], g1[0], g1[2]};
            const float add_b = ga[0];
            // ---- A (vrnn_prior_mid_kernel) ----
            {
                float va[NM_CHAIN_PAIRS];
#pragma unroll
                for (int u = 0; u < NM_CHAIN_PAIRS; ++u) va[u] = dot4(wa[u], xh);
                const float mine = reduce_scatter<NM_CHAIN_PAIRS, 32>(va, l32);
                const float other = nm_sx(mine, 32);
                if (!half && pown) {
                    const int p = pl;
                    const float mu = mine + bias_a, sraw = other + bias_a2;
                    const float sg = softplus(sraw) + 1e-4f;
                    const float z = mu + epsv * sg;
                    s_z[p] = z;
                    gran_store_t<XCD>(a.g_kpz + (size_t)b * (S4 + Z) + S4 + p, z, tag);
                }
            }
            __syncthreads();
            if (s_dead) return;                                          // (uniform: read behind the barrier)
            NM_STAMP(1, 2);
            // ---- B ----
            {
                const f32x4 xz = *reinterpret_cast<const f32x4*>(s_z + l32 * 4);
                float vb[NM_CHAIN_PAIRS];
#pragma unroll
                for (int u = 0; u < NM_CHAIN_PAIRS; ++u) vb[u] = dot4(wb[u], xz);
                const float mine = reduce_scatter<NM_CHAIN_PAIRS, 32>(vb, l32);
                if (pown) (half ? s_hj : s_hr)[pl] = lrelu(mine + add_b, 0.01f);
            }
            __syncthreads();
            NM_STAMP(1, 3);
            // ---- C ----
            {
                const f32x4 xr = *reinterpret_cast<const f32x4*>(s_hr + l32 * 4);
                const f32x4 xj = *reinterpret_cast<const f32x4*>(s_hj + l32 * 4);
                float vc[NM_CHAIN_PAIRS];
#pragma unroll
                for (int u = 0; u < NM_CHAIN_PAIRS; ++u) {
                    const int r = (wave * NM_CHAIN_PAIRS + u) * 2 + half;
                    const f32x4 wq = s_wc[min(r, rows_c - 1) * 32 + l32];
                    vc[u] = dot4(wq, r < R0 ? xr : xj);
                }
                const float mine = reduce_scatter<NM_CHAIN_PAIRS, 32>(vc, l32);
                if (pown) {
                    const int r = pl * 2 + half;
                    if (r < R0) s_root[r] = tanhf(mine + bias_c);
                    else if (r < rows_c) s_rot[r - R0] = mine + bias_c;
                }
            }
            __syncthreads();
            NM_STAMP(1, 4);
            // ---- D: forward kinematics, by wave 0 alone (round 6: lane j walks joint j's chain from the root; no barrier per tree level) ----
            if (wave == 0) {
                if (lane < K) {
                    const float* p = s_rot + lane * 6;
                    float x0 = p[0], x1 = p[1], x2 = p[2], y0 = p[3], y1 = p[4], y2 = p[5];
                    float nx = sqrtf((x0 * x0 + x1 * x1) + x2 * x2) + 1e-10f;
                    x0 /= nx; x1 /= nx; x2 /= nx;
                    float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
                    float nz = sqrtf((z0 * z0 + z1 * z1) + z2 * z2) + 1e-10f;
                    z0 /= nz; z1 /= nz; z2 /= nz;
                    float yy0 = z1 * x2 - z2 * x1, yy1 = z2 * x0 - z0 * x2, yy2 = z0 * x1 - z1 * x0;
                    float* R = s_Rl + lane * 9;
                    R[0] = x0; R[1] = yy0; R[2] = z0; R[3] = x1; R[4] = yy1; R[5] = z1; R[6] = x2; R[7] = yy2; R[8] = z2;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < K) {
                    const int rootj = tb.lvl_joint[0];
                    float G[9], pp[3];
#pragma unroll
                    for (int e = 0; e < 9; ++e) G[e] = s_Rl[rootj * 9 + e];
#pragma unroll
                    for (int r = 0; r < 3; ++r) pp[r] = s_root[r];
                    const int n = s_plen[lane];
                    for (int i = 0; i < n; ++i) {
                        const int c = s_path[lane][i];
                        float L[9], Gn[9];
#pragma unroll
                        for (int e = 0; e < 9; ++e) L[e] = s_Rl[c * 9 + e];
#pragma unroll
                        for (int e = 0; e < 9; ++e) { const int r = e / 3, cc = e % 3; Gn[e] = (G[r * 3] * L[cc] + G[r * 3 + 1] * L[3 + cc]) + G[r * 3 + 2] * L[6 + cc]; }
                        const float* of = tb.offset + c * 3;
#pragma unroll
                        for (int r = 0; r < 3; ++r) pp[r] = ((Gn[r * 3] * of[0] + Gn[r * 3 + 1] * of[1]) + Gn[r * 3 + 2] * of[2]) + pp[r];
#pragma unroll
                        for (int e = 0; e < 9; ++e) G[e] = Gn[e];
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) s_pos[lane * 3 + r] = pp[r];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                NM_STAMP(1, 5);
                for (int i = lane; i < S4; i += 64) {
                    const int kk = i >> 2, c = i & 3;
                    const float v = c < 3 ? s_pos[kk * 3 + c] : (s_root[3 + kk] + 1.0f) * 0.5f;
                    a.out_kp[(size_t)b * a.ldkp + (size_t)t * S4 + i] = v;             // the call's output (read after the launch)
                    gran_store_t<XCD>(a.g_kpz + (size_t)b * (S4 + Z) + i, v, tag);     // the GRU's input
                }
            }
            NM_STAMP(1, 6);
            __syncthreads();                                             // (the LDS state of this step is dead; s_z .. are rewritten next step)
        }
        if (XCD && tid == 0 && !s_dead) atomicAdd(a.ctl + 3, 1);
        return;
    }
    // =============================== worker workgroup =================================================================================
    const int gw = wg * (NM_CHAIN_T / 64) + wave;                    // global wave index, 0 .. 8 NW - 1
    constexpr int NWV = NW * (NM_CHAIN_T / 64);
    constexpr int HR = (384 + 3 * 512 + NWV - 1) / NWV;              // h-phase rows per wave (H = 512): 4 of 512 waves, 10 of 192
    constexpr int UN = (512 + NWV - 1) / NWV;                        // GRU units per wave: 1 / 3
    // ---- h-phase rows r = gw + NWV * i: [0,128) prior0 (lrelu) -> hid, [128,256) root0 -> rh, [256,384) joint0 -> jh, [384,1920) W_hh -> gh
    f32x4 wh[HR][2]; float bh[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int r = min(gw + NWV * i, 383 + 3 * H);
        const float* wr; const float* br;
        if (r < 128) { wr = a.w_prior0 + (size_t)r * H; br = a.b_prior0 + r; }
        else if (r < 256) { wr = a.w_root0 + (size_t)(r - 128) * (H + Z); br = a.b_root0 + (r - 128); }
        else if (r < 384) { wr = a.w_joint0 + (size_t)(r - 256) * (H + Z); br = a.b_joint0 + (r - 256); }
        else { wr = a.w_hh + (size_t)(r - 384) * H; br = a.b_hh + (r - 384); }
        wh[i][0] = *reinterpret_cast<const f32x4*>(wr + lane * 4);
        wh[i][1] = *reinterpret_cast<const f32x4*>(wr + 256 + lane * 4);
        bh[i] = *br;
    }
    // ---- GRU units j = gw + NWV * u: rows j, H + j, 2H + j of W_ih over [keypoints (4K) | z (Z)]; lane < 4K / 4 holds the keypoint
    // quad, lane < Z / 4 the latent quad (dot_seg's lane -> column assignment for n = 4K and n = Z)
    const int in = S4 + Z;
    f32x4 wk[UN][3], wz[UN][3]; float bi[UN][3];
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int ju = min(gw + NWV * u, H - 1);
            const float* w = a.w_ih + (size_t)(g * H + ju) * in;
            wk[u][g] = *reinterpret_cast<const f32x4*>(w + (lane * 4 < S4 ? lane * 4 : 0));
            wz[u][g] = *reinterpret_cast<const f32x4*>(w + S4 + (lane * 4 < Z ? lane * 4 : 0));
            bi[u][g] = a.b_ih[g * H + ju];
        }
    const bool kq = lane * 4 < S4, zq = lane * 4 < Z;
    // (reduce-scatter of the wave's h-phase rows: lane l owns row index rs_index(l), its first lane of four stores)
    const int hidx = rs_index<16, 64>(lane), hrow = gw + NWV * min(hidx, HR - 1);
    const bool hown = (lane & 3) == 0 && hidx < HR && hrow < 384 + 3 * H;
    float hbias = 0.f;
    if (hown) hbias = hrow < 128 ? a.b_prior0[hrow] : hrow < 256 ? a.b_root0[hrow - 128] : hrow < 384 ? a.b_joint0[hrow - 256] : a.b_hh[hrow - 384];
    static_assert(HR <= 16 && 3 * UN <= 16, "reduce_scatter<16, 64> holds the wave's rows / gate sums");
    if (XCD || a.wg_poll) {
        // ---- round 6: ONE wave of the workgroup polls, the other seven take the step's inputs from LDS.  With every wave polling,
        // 512 waves re-read the same 4 KB of h granules (and 1.8 KB of keypoint | latent granules) until the tags match: that polling
        // traffic is a good part of a hop's ~5.7 us (profiles/r04_rollout_ab.txt).  Wave 0 polls h_{t-1}[b] (8 granules per lane) /
        // the middle workgroup's keypoints | latent plus the workgroup's own granules of W_hh h and h_{t-1} (the units of a workgroup
        // are UN runs of eight: j = 8 wg + wave + NWV u), stores the VALUES to LDS, one workgroup barrier publishes them.  Two LDS
        // buffers by iteration parity: wave 0 refills a buffer two barriers after the other waves read it.  Arithmetic per row /
        // unit is unchanged (same lane -> column assignment, same reduction trees): outputs bit-identical to the per-wave polls.
        __shared__ __attribute__((aligned(16))) float s_hx[2][512];
        __shared__ __attribute__((aligned(16))) float s_kz[2][256];
        __shared__ __attribute__((aligned(16))) float s_g[2][UN * 32];          // [u][r | z | n | h_{t-1}][8 units]
        int it = 0;
        __syncthreads();                                             // (s_dead = 0 visible)
        for (int t = 0; t < T; ++t) {
            const unsigned tag = (unsigned)t + 1u;
#pragma unroll 1
            for (int b = 0; b < B; ++b, ++it) {
                f32x4 x0, x1;
                if (b == 0) NM_STAMP(0, 0);
                if (t == 0) {
                    x0 = *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * a.ldh0 + lane * 4);
                    x1 = *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * a.ldh0 + 256 + lane * 4);
                } else {
                    float* sh = s_hx[it & 1];
                    if (wave == 0) {
                        const nm_gran* ph = a.g_h[(t - 1) & 1] + (size_t)b * H;
                        f32x4 q0, q1, q2, q3; nm_f32x2 qj;
                        NM_CHAIN_POLL(gran_ld_4q1(ph + lane * 4, ph + lane * 4 + 2, ph + 256 + lane * 4, ph + 256 + lane * 4 + 2, ph, q0, q1, q2, q3, qj),
                                      nm_fbits(q0[1]) == (unsigned)t && nm_fbits(q0[3]) == (unsigned)t && nm_fbits(q1[1]) == (unsigned)t && nm_fbits(q1[3]) == (unsigned)t &&
                                      nm_fbits(q2[1]) == (unsigned)t && nm_fbits(q2[3]) == (unsigned)t && nm_fbits(q3[1]) == (unsigned)t && nm_fbits(q3[3]) == (unsigned)t);
                        *reinterpret_cast<f32x4*>(sh + lane * 4) = f32x4{q0[0], q0[2], q1[0], q1[2]};
                        *reinterpret_cast<f32x4*>(sh + 256 + lane * 4) = f32x4{q2[0], q2[2], q3[0], q3[2]};
                        if (!wave_alive && lane == 0) s_dead = 1;
                    }
                    __syncthreads();
                    if (s_dead) return;
                    x0 = *reinterpret_cast<const f32x4*>(sh + lane * 4); x1 = *reinterpret_cast<const f32x4*>(sh + 256 + lane * 4);
                }
                if (b == 0) NM_STAMP(0, 1);
                {   // the wave's HR row sums in one reduce-scatter (lane l ends up with row hidx = rs_index(l)); one predicated store each
                    float vh[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (i < HR) { float acc = 0.f; acc += dot4(wh[i][0], x0); acc += dot4(wh[i][1], x1); vh[i] = acc; }
                        else vh[i] = 0.f;
                    }
                    const float tot = reduce_scatter<16, 64>(vh, lane);
                    if (hown) {
                        const float v = tot + hbias;
                        if (hrow < 128) gran_store_t<XCD>(a.g_hid + (size_t)b * 128 + hrow, lrelu(v, 0.01f), tag);
                        else if (hrow < 256) gran_store_t<XCD>(a.g_rh + (size_t)b * 128 + (hrow - 128), v, tag);
                        else if (hrow < 384) gran_store_t<XCD>(a.g_jh + (size_t)b * 128 + (hrow - 256), v, tag);
                        else gran_store_t<XCD>(a.g_gh + (size_t)b * 3 * H + (hrow - 384), v, tag);
                    }
                }
            }
#pragma unroll 1
            for (int b = 0; b < B; ++b, ++it) {
                float* sk = s_kz[it & 1]; float* sg = s_g[it & 1];
                if (b == 0) NM_STAMP(0, 2);
                if (wave == 0) {
                    const nm_gran* pk = a.g_kpz + (size_t)b * (S4 + Z);
                    const nm_gran* pg = a.g_gh + (size_t)b * 3 * H;
                    // lane l < 16 UN: run u = l >> 4, array (l >> 2) & 3 (W_hh h of gate r | z | n, or h_{t-1}), granule pair l & 3 of the run's eight units
                    const int gu = lane >> 4, garr = (lane >> 2) & 3, j0 = wg * 8 + NWV * gu + 2 * (lane & 3);
                    const bool gon = lane < 16 * UN && j0 < H;
                    const bool hprev = garr == 3 && t > 0;           // (t = 0: h_{t-1} is the call's input; the lanes re-read gate n's granules, unused)
                    const nm_gran* px = !gon ? pg : (hprev ? a.g_h[(t - 1) & 1] + (size_t)b * H + j0 : pg + (garr == 3 ? 2 : garr) * H + j0);
                    const unsigned xtag = hprev ? (unsigned)t : tag;
                    f32x4 k0, k1, z0, z1, gx;
                    NM_CHAIN_POLL(gran_ld_5q(pk + (kq ? lane * 4 : 0), pk + (kq ? lane * 4 : 0) + 2, pk + S4 + (zq ? lane * 4 : 0), pk + S4 + (zq ? lane * 4 : 0) + 2,
                                             px, k0, k1, z0, z1, gx),
                                  nm_fbits(k0[1]) == tag && nm_fbits(k0[3]) == tag && nm_fbits(k1[1]) == tag && nm_fbits(k1[3]) == tag &&
                                  nm_fbits(z0[1]) == tag && nm_fbits(z0[3]) == tag && nm_fbits(z1[1]) == tag && nm_fbits(z1[3]) == tag &&
                                  (!gon || (nm_fbits(gx[1]) == xtag && nm_fbits(gx[3]) == xtag)));
                    if (kq) *reinterpret_cast<f32x4*>(sk + lane * 4) = f32x4{k0[0], k0[2], k1[0], k1[2]};
                    if (zq) *reinterpret_cast<f32x4*>(sk + 128 + lane * 4) = f32x4{z0[0], z0[2], z1[0], z1[2]};
                    if (lane < 16 * UN) { sg[gu * 32 + garr * 8 + 2 * (lane & 3)] = gx[0]; sg[gu * 32 + garr * 8 + 2 * (lane & 3) + 1] = gx[2]; }
                    if (!wave_alive && lane == 0) s_dead = 1;
                }
                __syncthreads();
                if (s_dead) return;
                if (b == 0) NM_STAMP(0, 3);
                f32x4 xkq = {0.f, 0.f, 0.f, 0.f}, xzq = xkq;
                if (kq) xkq = *reinterpret_cast<const f32x4*>(sk + lane * 4);
                if (zq) xzq = *reinterpret_cast<const f32x4*>(sk + 128 + lane * 4);
                float vg[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) vg[i] = 0.f;
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    float ar = 0.f, az = 0.f, an = 0.f;
                    if (kq) { ar += dot4(wk[u][0], xkq); az += dot4(wk[u][1], xkq); an += dot4(wk[u][2], xkq); }
                    if (zq) { ar += dot4(wz[u][0], xzq); az += dot4(wz[u][1], xzq); an += dot4(wz[u][2], xzq); }
                    vg[3 * u] = ar; vg[3 * u + 1] = az; vg[3 * u + 2] = an;
                }
                const float gtot = reduce_scatter<16, 64>(vg, lane);      // value i of the wave in the lanes with rs_index == i: lane RSL(i) is one
#define RSL(i) ((((i) >> 3) & 1) * 32 + (((i) >> 2) & 1) * 16 + (((i) >> 1) & 1) * 8 + ((i) & 1) * 4)
                {   // the gates of the wave's UN units, unit u in lane u (the three sums of unit u sit in lanes RSL(3u), RSL(3u + 1), RSL(3u + 2))
                    float ar = 0.f, az = 0.f, an = 0.f;
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const float r_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gtot), RSL(3 * u)));
                        const float z_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gtot), RSL(3 * u + 1)));
                        const float n_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gtot), RSL(3 * u + 2)));
                        if (lane == u) { ar = r_; az = z_; an = n_; }
                    }
                    const int ul = min(lane, UN - 1), j = gw + NWV * ul;
                    if (lane < UN && j < H) {
                        float b0 = bi[0][0], b1 = bi[0][1], b2 = bi[0][2];
#pragma unroll
                        for (int u = 1; u < UN; ++u) if (lane == u) { b0 = bi[u][0]; b1 = bi[u][1]; b2 = bi[u][2]; }
                        const float hpv = t == 0 ? a.h0[(size_t)b * a.ldh0 + j] : sg[ul * 32 + 24 + wave];
                        const float rg = sigmoidf((ar + b0) + sg[ul * 32 + wave]);
                        const float zg = sigmoidf((az + b1) + sg[ul * 32 + 8 + wave]);
                        const float ng = tanhf((an + b2) + rg * sg[ul * 32 + 16 + wave]);
                        const float hn = (hpv - ng) * zg + ng;
                        gran_store_t<XCD>(a.g_h[t & 1] + (size_t)b * H + j, hn, tag);
                        if (t == T - 1) a.h_out[(size_t)b * H + j] = hn;
                    }
                }
                if (b == 0) NM_STAMP(0, 4);
            }
        }
        if (XCD && tid == 0) atomicAdd(a.ctl + 3, 1);
        return;
    }
    if constexpr (!XCD) {
    const int j = gw, jc = min(j, H - 1);
    for (int t = 0; t < T && wave_alive; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        // ---- h-phase of step t for every batch element: h_t[b] columns 4 lane .. +3 and 256 + 4 lane .. +3 (t = 0: the call's input, plain)
#pragma unroll 1
        for (int b = 0; b < B && wave_alive; ++b) {
            f32x4 x0, x1;
            if (t == 0) {
                x0 = *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * a.ldh0 + lane * 4);
                x1 = *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * a.ldh0 + 256 + lane * 4);
            } else {
                const nm_gran* ph = a.g_h[(t - 1) & 1] + (size_t)b * H;
                f32x4 q0, q1, q2, q3; nm_f32x2 qj;
                NM_CHAIN_POLL(gran_ld_4q1(ph + lane * 4, ph + lane * 4 + 2, ph + 256 + lane * 4, ph + 256 + lane * 4 + 2, ph + jc, q0, q1, q2, q3, qj),
                              nm_fbits(q0[1]) == (unsigned)t && nm_fbits(q0[3]) == (unsigned)t && nm_fbits(q1[1]) == (unsigned)t && nm_fbits(q1[3]) == (unsigned)t &&
                              nm_fbits(q2[1]) == (unsigned)t && nm_fbits(q2[3]) == (unsigned)t && nm_fbits(q3[1]) == (unsigned)t && nm_fbits(q3[3]) == (unsigned)t);
                x0 = f32x4{q0[0], q0[2], q1[0], q1[2]}; x1 = f32x4{q2[0], q2[2], q3[0], q3[2]};
            }
            if (!wave_alive) break;
#pragma unroll
            for (int i = 0; i < HR; ++i) {
                const int r = gw + NWV * i;
                float acc = 0.f;
                acc += dot4(wh[i][0], x0);
                acc += dot4(wh[i][1], x1);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += nm_sx(acc, off);
                if (lane == 0 && r < 384 + 3 * H) {
                    const float v = acc + bh[i];
                    if (r < 128) gran_store_t<XCD>(a.g_hid + (size_t)b * 128 + r, lrelu(v, 0.01f), tag);
                    else if (r < 256) gran_store_t<XCD>(a.g_rh + (size_t)b * 128 + (r - 128), v, tag);
                    else if (r < 384) gran_store_t<XCD>(a.g_jh + (size_t)b * 128 + (r - 256), v, tag);
                    else gran_store_t<XCD>(a.g_gh + (size_t)b * 3 * H + (r - 384), v, tag);
                }
            }
        }
        // ---- GRU unit j of step t: keypoints | latent of the middle workgroups, W_hh h + b_hh of three h-phase rows
#pragma unroll 1
        for (int b = 0; b < B && wave_alive && j < H; ++b) {
            const nm_gran* pk = a.g_kpz + (size_t)b * (S4 + Z);
            const nm_gran* pg = a.g_gh + (size_t)b * 3 * H;
            // (h_t[b][j] for the blend: its granule carries tag t - it was complete when this wave ran the h-phase; t = 0: the call's input)
            const nm_gran* pp = t == 0 ? pg + jc : a.g_h[(t - 1) & 1] + (size_t)b * H + jc;
            f32x4 k0, k1, z0, z1; nm_f32x2 gr, gz, gn, gp;
            NM_CHAIN_POLL(gran_ld_4q4(pk + (kq ? lane * 4 : 0), pk + (kq ? lane * 4 : 0) + 2, pk + S4 + (zq ? lane * 4 : 0), pk + S4 + (zq ? lane * 4 : 0) + 2,
                                      pg + jc, pg + H + jc, pg + 2 * H + jc, pp, k0, k1, z0, z1, gr, gz, gn, gp),
                          nm_fbits(k0[1]) == tag && nm_fbits(k0[3]) == tag && nm_fbits(k1[1]) == tag && nm_fbits(k1[3]) == tag &&
                          nm_fbits(z0[1]) == tag && nm_fbits(z0[3]) == tag && nm_fbits(z1[1]) == tag && nm_fbits(z1[3]) == tag &&
                          nm_fbits(gr[1]) == tag && nm_fbits(gz[1]) == tag && nm_fbits(gn[1]) == tag && (t == 0 || nm_fbits(gp[1]) == (unsigned)t));
            if (!wave_alive) break;
            const float hpv = t == 0 ? a.h0[(size_t)b * a.ldh0 + jc] : gp[0];
            const f32x4 xkq = {k0[0], k0[2], k1[0], k1[2]}, xzq = {z0[0], z0[2], z1[0], z1[2]};
            float ar = 0.f, az = 0.f, an = 0.f;
            if (kq) { ar += dot4(wk[0][0], xkq); az += dot4(wk[0][1], xkq); an += dot4(wk[0][2], xkq); }
            if (zq) { ar += dot4(wz[0][0], xzq); az += dot4(wz[0][1], xzq); an += dot4(wz[0][2], xzq); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { ar += nm_sx(ar, off); az += nm_sx(az, off); an += nm_sx(an, off); }
            if (lane == 0) {
                const float rg = sigmoidf((ar + bi[0][0]) + gr[0]);
                const float zg = sigmoidf((az + bi[0][1]) + gz[0]);
                const float ng = tanhf((an + bi[0][2]) + rg * gn[0]);
                const float hn = (hpv - ng) * zg + ng;
                gran_store_t<XCD>(a.g_h[t & 1] + (size_t)b * H + j, hn, tag);
                if (t == T - 1) a.h_out[(size_t)b * H + j] = hn;
            }
        }
    }
    }
}

// ---- the posterior steps of HSVRNNBVH.encode as ONE persistent launch (round 5; BASELINE north_star "one kernel per timestep") --------
// hsvrnn_bvh.py:86-135.  The step of vrnn_prior_chain_kernel with the posterior's extra stage: S samples per clip are decoded and the
// one nearest to the detected keypoints is kept.  Roles (512 threads each, one workgroup per CU, all resident):
//   * NM_CHAIN_NW worker workgroups: wave gw owns h-phase row gw (one of prior0 | post0 | root0_h | joint0_h, 128 rows each; the
//     post0 rows also take the detected keypoints' 4K columns) and W_hh rows gw + 512 i, and GRU unit j = gw;
//   * S x B sample workgroups: workgroup (s, b) = vrnn_post_mid_kernel's phases for sample s of clip b (posterior parameters from
//     hid_post[b], z = mu + eps[t][s][b] sigma, decoder hidden layers, heads, kinematics, squared distance to the observation); it
//     publishes its keypoints | latent and LAST its distance as granules, then polls the clip's S distances itself: the nearest sample
//     (lowest index on ties: fk_kernel's scan) writes the step's outputs - keypoints, latent, rotations, index, distance - from its
//     own LDS, so nothing but the distances crosses workgroups for the selection;
//   * B statistics workgroups: the prior's and the posterior's parameters of clip b (both second layers register-resident) and the
//     KL term of the step - off the recurrence's critical path, read by nobody inside the launch.
// A GRU wave polls the S distance granules of its clip, picks the same sample and reads THAT sample's keypoint | latent granules.
// Hand-offs, tags, bounded spins and the abort word are vrnn_prior_chain_kernel's; arithmetic per row, the kinematic chain, the
// distance sum (joint order), the argmin and the KL tree are the launch-per-phase step's: outputs bit-identical to it.
struct PostChainArgs {
    const float *w_prior0, *b_prior0, *w_post0, *b_post0, *w_root0, *b_root0, *w_joint0, *b_joint0, *w_hh, *b_hh, *w_ih, *b_ih;
    const float *w_q2, *b_q2, *w_p2, *b_p2;     // second layers of the posterior / prior MLPs
    MidArgs mid;                        // decoder weights / tree / offset (its data pointers are not used)
    const float* h0;                    // [H] init_kypt_rnn_state (every clip starts from it)
    const float* obs; int ldobs;        // detected keypoints [B][ldobs], step t at column t * 4K
    const float* eps;                   // [T][S][B][Z]
    float* out_kp; int ldkp;            // [B][ldkp], step t at column t * 4K
    float* out_z; int ldz;              // [B][ldz], step t at column t * Z
    float* out_R; int ldR;              // [B][ldR], step t at column t * 9K (or null)
    float* out_h; int ldh;              // [B][ldh]: state after step t at column (t + 1) * H (column 0 is written by the caller)
    int32_t* best; int ldbest;          // [B][ldbest] (or null)
    float *kl, *rec; int ldstat;        // per (clip, step) sums (kl may be null)
    nm_gran *g_hidq, *g_rh, *g_jh;      // [B][128] x 3   h-phase -> sample workgroups
    // h-phase -> statistics workgroups: one slot PER STEP ([T][B][128] each with statistics workgroups, else one slot that nobody reads).
    // Nothing inside the launch waits for a statistics workgroup, so nothing may overwrite what it has not read yet: a single slot
    // rewritten every step (round 5) let a statistics workgroup that fell one step behind - a contended device - poll for a tag that
    // was gone, time out and abort the whole encode (advisor finding).  Written once, a slot can be read arbitrarily late.
    nm_gran *g_sq, *g_hidp;
    nm_gran* g_gh;                      // [B][3H]        h-phase -> GRU
    nm_gran* g_kpz;                     // [S][B][4K + Z] sample workgroup -> GRU
    nm_gran* g_d[2];                    // [S][B] x 2     sample workgroup -> GRU and the clip's other sample workgroups (buffer t & 1: a sample workgroup may
                                        //                still be reading step t's distances when a peer publishes step t + 1's; buffer t & 1 is rewritten at
                                        //                step t + 2, which needs every sample's distance of step t + 1, i.e. every reader of step t done)
    nm_gran* g_h[2];                    // [B][H] x 2     GRU of step t -> h-phase / GRU of step t + 1 (buffer t & 1)
    unsigned* abort; unsigned* status;
    int B, S, T, K, Z, H;
    int nstat;                          // statistics workgroups: B (encode: KL) or 0 (the conditioning steps of generate)
    int backoff, spin_limit;
    int stat_delay;                     // NM355_CHAIN_STAT_DELAY (test hook): s_sleep(127) units a statistics workgroup idles before every step - a lagging one
};

__device__ __forceinline__ void gran_ld_1(const nm_gran* p, nm_f32x2& v) {
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
}

__global__ __launch_bounds__(NM_CHAIN_T) void vrnn_post_chain_kernel(PostChainArgs a) {
    __shared__ int s_dead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, S = a.S, T = a.T, K = a.K, Z = a.Z, H = a.H, S4 = 4 * K;
    bool wave_alive = true;
    if (tid == 0) s_dead = 0;
    const int role = (int)blockIdx.x - NM_CHAIN_NW;        // < 0 worker, [0, B) statistics workgroup of clip b, then S B sample workgroups s * B + b
    const int nstat = a.nstat;
    if (role >= 0 && role < nstat) {
        // =========================== statistics workgroup of clip b: prior / posterior parameters, KL ===========================
        const int half = lane >> 5, l32 = lane & 31, b = role;
        __shared__ float s_red[256];
        f32x4 wq[NM_CHAIN_PAIRS], wp[NM_CHAIN_PAIRS];
#pragma unroll
        for (int u = 0; u < NM_CHAIN_PAIRS; ++u) {
            const int p = wave * NM_CHAIN_PAIRS + u;
            wq[u] = *reinterpret_cast<const f32x4*>(a.w_q2 + (size_t)(p + half * Z) * 128 + l32 * 4);
            wp[u] = *reinterpret_cast<const f32x4*>(a.w_p2 + (size_t)(p + half * Z) * 128 + l32 * 4);
        }
        const int pl = wave * NM_CHAIN_PAIRS + (l32 < NM_CHAIN_PAIRS ? l32 : 0);
        const float bq = a.b_q2[pl], bq2 = a.b_q2[pl + Z], bp = a.b_p2[pl], bp2 = a.b_p2[pl + Z];
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const unsigned tag = (unsigned)t + 1u;
            for (int sl = 0; sl < a.stat_delay; ++sl) __builtin_amdgcn_s_sleep(127);
            f32x4 q0, q1, p0, p1; nm_f32x2 dummy;
            const nm_gran* pq = a.g_sq + ((size_t)t * B + b) * 128 + l32 * 4;
            const nm_gran* pp = a.g_hidp + ((size_t)t * B + b) * 128 + l32 * 4;
            NM_CHAIN_POLL(gran_ld_4q1(pq, pq + 2, pp, pp + 2, pq, q0, q1, p0, p1, dummy),
                          nm_fbits(q0[1]) == tag && nm_fbits(q0[3]) == tag && nm_fbits(q1[1]) == tag && nm_fbits(q1[3]) == tag &&
                          nm_fbits(p0[1]) == tag && nm_fbits(p0[3]) == tag && nm_fbits(p1[1]) == tag && nm_fbits(p1[3]) == tag);
            if (!wave_alive) s_dead = 1;
            const f32x4 xq = {q0[0], q0[2], q1[0], q1[2]}, xp = {p0[0], p0[2], p1[0], p1[2]};
            float qm = 0.f, qs = 0.f, pm = 0.f, ps = 0.f;
#pragma unroll
            for (int u = 0; u < NM_CHAIN_PAIRS; ++u) {
                const float vq = half_reduce(dot4(wq[u], xq)), oq = nm_sx(vq, 32);
                const float vp = half_reduce(dot4(wp[u], xp)), op = nm_sx(vp, 32);
                if (l32 == u) { qm = vq; qs = oq; pm = vp; ps = op; }
            }
            if (tid < 256) s_red[tid] = 0.f;
            __syncthreads();
            if (s_dead) return;
            if (!half && l32 < NM_CHAIN_PAIRS) {
                const int p = wave * NM_CHAIN_PAIRS + l32;
                const float qmu = qm + bq, qsg = softplus(qs + bq2) + 1e-4f, pmu = pm + bp, psg = softplus(ps + bp2) + 1e-4f;
                const float ratio = qsg / psg, vr = ratio * ratio, dm = (qmu - pmu) / psg;
                s_red[p] = 0.f + 0.5f * (((vr + dm * dm) - 1.0f) - logf(vr));
            }
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) { if (tid < st) s_red[tid] += s_red[tid + st]; __syncthreads(); }
            if (tid == 0 && a.kl) a.kl[(size_t)b * a.ldstat + t] = s_red[0];
            __syncthreads();
        }
        return;
    }
    if (role >= 0) {
        // =========================== sample workgroup (s, b) =====================================================================
        const MidArgs& m = a.mid;
        __shared__ __attribute__((aligned(16))) float s_z[128], s_hr[128], s_hj[128], s_root[36], s_rot[192];
        __shared__ float s_Rl[32 * 9], s_Rg[32 * 9], s_pos[32 * 3], s_dk[32];
        __shared__ int s_win;
        __shared__ FkTables tb;
        const int half = lane >> 5, l32 = lane & 31, smp = (role - nstat) / B, b = (role - nstat) % B;
        const int R0 = 3 + K, J6 = 6 * K, rows_c = R0 + J6;
        extern __shared__ f32x4 s_wc[];                                  // [rows_c][32 quads]
        f32x4 wa[NM_CHAIN_PAIRS], wb[NM_CHAIN_PAIRS];
#pragma unroll
        for (int u = 0; u < NM_CHAIN_PAIRS; ++u) {
            const int p = wave * NM_CHAIN_PAIRS + u;
            wa[u] = *reinterpret_cast<const f32x4*>(a.w_q2 + (size_t)(p + half * Z) * 128 + l32 * 4);
            wb[u] = *reinterpret_cast<const f32x4*>((half ? m.w_joint0 : m.w_root0) + (size_t)p * (H + Z) + H + l32 * 4);
            const int r = min(p * 2 + half, rows_c - 1);
            const float* pc = r < R0 ? m.w_root2 + (size_t)r * 128 : m.w_joint2 + (size_t)(r - R0) * 128;
            if (p * 2 + half < rows_c) s_wc[(p * 2 + half) * 32 + l32] = *reinterpret_cast<const f32x4*>(pc + l32 * 4);
        }
        // (round 6, as in vrnn_prior_chain_kernel: the 16 row pairs of a wave summed by reduce_scatter - lane l32 holds pair rs_index(l32))
        const int pu = rs_index<NM_CHAIN_PAIRS, 32>(l32);
        const bool pown = (l32 & 1) == 0;
        const int pl = wave * NM_CHAIN_PAIRS + pu;
        const float bias_a = a.b_q2[pl], bias_a2 = a.b_q2[pl + Z];
        const int rl = min(pl * 2 + half, rows_c - 1);
        const float bias_c = *(rl < R0 ? m.b_root2 + rl : m.b_joint2 + (rl - R0));
        fk_tables_load(tb, m.lvl_joint, m.lvl_start, m.parents, m.offset + (size_t)b * K * 3, K, m.nlevels, tid);
        __syncthreads();
        __shared__ unsigned char s_path[32][32];                         // joint j's chain from the root's child down to j (vrnn_prior_chain_kernel)
        __shared__ int s_plen[32];
        if (tid < K) {
            const int rootj = tb.lvl_joint[0];
            int n = 0, c = tid;
            unsigned char tmp[32];
            while (c != rootj && n < 32) { tmp[n++] = (unsigned char)c; c = tb.parents[c]; }
            for (int i = 0; i < n; ++i) s_path[tid][i] = tmp[n - 1 - i];
            s_plen[tid] = n;
        }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const unsigned tag = (unsigned)t + 1u;
            const float epsv = a.eps[(((size_t)t * S + smp) * B + b) * Z + pl];             // (inputs of the call: plain loads)
            const float obsv = a.obs[(size_t)b * a.ldobs + (size_t)t * S4 + min(tid, S4 - 1)];
            f32x4 g0, g1; nm_f32x2 ga;
            const nm_gran* ph = a.g_hidq + (size_t)b * 128 + l32 * 4;
            const nm_gran* pa = (half ? a.g_jh : a.g_rh) + (size_t)b * 128 + pl;
            NM_CHAIN_POLL(gran_ld_2q1(ph, ph + 2, pa, g0, g1, ga),
                          nm_fbits(g0[1]) == tag && nm_fbits(g0[3]) == tag && nm_fbits(g1[1]) == tag && nm_fbits(g1[3]) == tag && nm_fbits(ga[1]) == tag);
            if (!wave_alive) s_dead = 1;
            const f32x4 xh = {g0[0], g0[2], g1[0], g1[2]};
            const float add_b = ga[0];
            // ---- A: posterior parameters and this workgroup's sample ----
            {
                float va[NM_CHAIN_PAIRS];
#pragma unroll
                for (int u = 0; u < NM_CHAIN_PAIRS; ++u) va[u] = dot4(wa[u], xh);
                const float mine = reduce_scatter<NM_CHAIN_PAIRS, 32>(va, l32);
                const float other = nm_sxc<32>(mine);
                if (!half && pown) {
                    const int p = pl;
                    const float mu = mine + bias_a, sraw = other + bias_a2;
                    const float sg = softplus(sraw) + 1e-4f;
                    const float z = mu + epsv * sg;
                    s_z[p] = z;
                    gran_store(a.g_kpz + ((size_t)smp * B + b) * (S4 + Z) + S4 + p, z, tag);
                }
            }
            __syncthreads();
            if (s_dead) return;                                          // (uniform: read behind the barrier)
            // ---- B ----
            {
                const f32x4 xz = *reinterpret_cast<const f32x4*>(s_z + l32 * 4);
                float vb[NM_CHAIN_PAIRS];
#pragma unroll
                for (int u = 0; u < NM_CHAIN_PAIRS; ++u) vb[u] = dot4(wb[u], xz);
                const float mine = reduce_scatter<NM_CHAIN_PAIRS, 32>(vb, l32);
                if (pown) (half ? s_hj : s_hr)[pl] = lrelu(mine + add_b, 0.01f);
            }
            __syncthreads();
            // ---- C ----
            {
                const f32x4 xr = *reinterpret_cast<const f32x4*>(s_hr + l32 * 4);
                const f32x4 xj = *reinterpret_cast<const f32x4*>(s_hj + l32 * 4);
                float vc[NM_CHAIN_PAIRS];
#pragma unroll
                for (int u = 0; u < NM_CHAIN_PAIRS; ++u) {
                    const int r = (wave * NM_CHAIN_PAIRS + u) * 2 + half;
                    const f32x4 wq = s_wc[min(r, rows_c - 1) * 32 + l32];
                    vc[u] = dot4(wq, r < R0 ? xr : xj);
                }
                const float mine = reduce_scatter<NM_CHAIN_PAIRS, 32>(vc, l32);
                if (pown) {
                    const int r = pl * 2 + half;
                    if (r < R0) s_root[r] = tanhf(mine + bias_c);
                    else if (r < rows_c) s_rot[r - R0] = mine + bias_c;
                }
            }
            __syncthreads();
            // ---- D: forward kinematics by wave 0, lane j walking joint j's chain from the root (round 6; vrnn_prior_chain_kernel) ----
            if (wave == 0) {
                if (lane < K) {
                    const float* p = s_rot + lane * 6;
                    float x0 = p[0], x1 = p[1], x2 = p[2], y0 = p[3], y1 = p[4], y2 = p[5];
                    float nx = sqrtf((x0 * x0 + x1 * x1) + x2 * x2) + 1e-10f;
                    x0 /= nx; x1 /= nx; x2 /= nx;
                    float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
                    float nz = sqrtf((z0 * z0 + z1 * z1) + z2 * z2) + 1e-10f;
                    z0 /= nz; z1 /= nz; z2 /= nz;
                    float yy0 = z1 * x2 - z2 * x1, yy1 = z2 * x0 - z0 * x2, yy2 = z0 * x1 - z1 * x0;
                    float* R = s_Rl + lane * 9;
                    R[0] = x0; R[1] = yy0; R[2] = z0; R[3] = x1; R[4] = yy1; R[5] = z1; R[6] = x2; R[7] = yy2; R[8] = z2;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < K) {
                    const int rootj = tb.lvl_joint[0];
                    float G[9], pp[3];
#pragma unroll
                    for (int e = 0; e < 9; ++e) G[e] = s_Rl[rootj * 9 + e];
#pragma unroll
                    for (int r = 0; r < 3; ++r) pp[r] = s_root[r];
                    const int n = s_plen[lane];
                    for (int i = 0; i < n; ++i) {
                        const int c = s_path[lane][i];
                        float L[9], Gn[9];
#pragma unroll
                        for (int e = 0; e < 9; ++e) L[e] = s_Rl[c * 9 + e];
#pragma unroll
                        for (int e = 0; e < 9; ++e) { const int r = e / 3, cc = e % 3; Gn[e] = (G[r * 3] * L[cc] + G[r * 3 + 1] * L[3 + cc]) + G[r * 3 + 2] * L[6 + cc]; }
                        const float* of = tb.offset + c * 3;
#pragma unroll
                        for (int r = 0; r < 3; ++r) pp[r] = ((Gn[r * 3] * of[0] + Gn[r * 3 + 1] * of[1]) + Gn[r * 3 + 2] * of[2]) + pp[r];
#pragma unroll
                        for (int e = 0; e < 9; ++e) G[e] = Gn[e];
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) s_pos[lane * 3 + r] = pp[r];
#pragma unroll
                    for (int e = 0; e < 9; ++e) s_Rg[lane * 9 + e] = G[e];
                }
            }
            __syncthreads();
            // ---- E: keypoints, distance to the observation (per-joint terms, summed in joint order) ----
            float kpv = 0.f;
            if (tid < S4) {
                const int k = tid >> 2, c = tid & 3;
                kpv = c < 3 ? s_pos[k * 3 + c] : (s_root[3 + k] + 1.0f) * 0.5f;
                gran_store(a.g_kpz + ((size_t)smp * B + b) * (S4 + Z) + tid, kpv, tag);
            }
            {
                const float u = obsv - kpv, q = u * u;
                const float q1 = __shfl_down(q, 1), q2 = __shfl_down(q, 2), q3 = __shfl_down(q, 3);
                if (tid < S4 && (tid & 3) == 0) s_dk[tid >> 2] = ((q + q1) + q2) + q3;
            }
            __syncthreads();
            // ---- F: publish the distance (last), then find the clip's nearest sample; the winner writes the step's outputs ----
            if (wave == 0) {
                float d = 0.f;
                if (lane == 0) {
                    for (int k = 0; k < K; ++k) d += s_dk[k];
                    gran_store(a.g_d[t & 1] + (size_t)smp * B + b, d, tag);
                }
                nm_f32x2 gd;
                const nm_gran* pd = a.g_d[t & 1] + (size_t)min(lane, S - 1) * B + b;
                NM_CHAIN_POLL(gran_ld_1(pd, gd), nm_fbits(gd[1]) == tag);
                // first minimum in sample order (fk_kernel's scan): butterfly over (distance, index), the lower index wins a tie
                float bd = lane < S ? gd[0] : INFINITY; int bi = lane < S ? lane : 1 << 20;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = nm_sx(bd, off); const int oi = nm_sx(bi, off);
                    if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
                }
                if (lane == 0) {
                    s_win = (wave_alive && bi == smp) ? 1 : 0;
                    if (!wave_alive) s_dead = 1;
                    if (s_win) {
                        if (a.best) a.best[(size_t)b * a.ldbest + t] = bi;
                        if (a.rec) a.rec[(size_t)b * a.ldstat + t] = bd;
                    }
                }
            }
            __syncthreads();
            if (s_dead) return;
            if (s_win) {
                if (tid < S4) a.out_kp[(size_t)b * a.ldkp + (size_t)t * S4 + tid] = kpv;
                if (tid < Z) a.out_z[(size_t)b * a.ldz + (size_t)t * Z + tid] = s_z[tid];
                if (a.out_R && tid < K * 9) a.out_R[(size_t)b * a.ldR + (size_t)t * K * 9 + tid] = s_Rg[tid];
            }
            __syncthreads();                                             // (the LDS state of this step is dead)
        }
        return;
    }
    // =============================== worker workgroup (waves are independent: no workgroup barrier below) ==========================
    const int gw = (int)blockIdx.x * (NM_CHAIN_T / 64) + wave;       // global wave index, 0 .. 511
    constexpr int NWV = NM_CHAIN_NW * (NM_CHAIN_T / 64);
    constexpr int HR = 4;
    // ---- h-phase rows: i = 0: row gw of [prior0 | post0 | root0_h | joint0_h] (128 each); i = 1 .. 3: W_hh rows gw + 512 (i - 1)
    f32x4 wh[HR][2]; float bh[HR];
    f32x4 wobs = {0.f, 0.f, 0.f, 0.f};                               // post0 rows: the detected keypoints' columns H .. H + 4K
    const int sect = gw >> 7, rsec = gw & 127;                       // section of row gw and its row inside the section
    const bool oq = sect == 1 && lane * 4 < S4;
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const float* wr; const float* br;
        if (i == 0) {
            if (sect == 0) { wr = a.w_prior0 + (size_t)rsec * H; br = a.b_prior0 + rsec; }
            else if (sect == 1) { wr = a.w_post0 + (size_t)rsec * (H + S4); br = a.b_post0 + rsec; }
            else if (sect == 2) { wr = a.w_root0 + (size_t)rsec * (H + Z); br = a.b_root0 + rsec; }
            else { wr = a.w_joint0 + (size_t)rsec * (H + Z); br = a.b_joint0 + rsec; }
        } else { const int r = gw + NWV * (i - 1); wr = a.w_hh + (size_t)r * H; br = a.b_hh + r; }
        wh[i][0] = *reinterpret_cast<const f32x4*>(wr + lane * 4);
        wh[i][1] = *reinterpret_cast<const f32x4*>(wr + 256 + lane * 4);
        bh[i] = *br;
        if (i == 0 && sect == 1) wobs = *reinterpret_cast<const f32x4*>(wr + H + (oq ? lane * 4 : 0));
    }
    const int in = S4 + Z, j = gw, jc = min(j, H - 1);
    f32x4 wk[3], wz[3]; float bi3[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float* w = a.w_ih + (size_t)(g * H + jc) * in;
        wk[g] = *reinterpret_cast<const f32x4*>(w + (lane * 4 < S4 ? lane * 4 : 0));
        wz[g] = *reinterpret_cast<const f32x4*>(w + S4 + (lane * 4 < Z ? lane * 4 : 0));
        bi3[g] = a.b_ih[g * H + jc];
    }
    const bool kq = lane * 4 < S4, zq = lane * 4 < Z;
    for (int t = 0; t < T && wave_alive; ++t) {
        const unsigned tag = (unsigned)t + 1u;
#pragma unroll 1
        for (int b = 0; b < B && wave_alive; ++b) {
            f32x4 x0, x1;
            if (t == 0) {
                x0 = *reinterpret_cast<const f32x4*>(a.h0 + lane * 4);
                x1 = *reinterpret_cast<const f32x4*>(a.h0 + 256 + lane * 4);
            } else {
                const nm_gran* ph = a.g_h[(t - 1) & 1] + (size_t)b * H;
                f32x4 q0, q1, q2, q3; nm_f32x2 qj;
                NM_CHAIN_POLL(gran_ld_4q1(ph + lane * 4, ph + lane * 4 + 2, ph + 256 + lane * 4, ph + 256 + lane * 4 + 2, ph + jc, q0, q1, q2, q3, qj),
                              nm_fbits(q0[1]) == (unsigned)t && nm_fbits(q0[3]) == (unsigned)t && nm_fbits(q1[1]) == (unsigned)t && nm_fbits(q1[3]) == (unsigned)t &&
                              nm_fbits(q2[1]) == (unsigned)t && nm_fbits(q2[3]) == (unsigned)t && nm_fbits(q3[1]) == (unsigned)t && nm_fbits(q3[3]) == (unsigned)t);
                x0 = f32x4{q0[0], q0[2], q1[0], q1[2]}; x1 = f32x4{q2[0], q2[2], q3[0], q3[2]};
            }
            if (!wave_alive) break;
            f32x4 xo = {0.f, 0.f, 0.f, 0.f};
            if (oq) xo = *reinterpret_cast<const f32x4*>(a.obs + (size_t)b * a.ldobs + (size_t)t * S4 + lane * 4);
            float vh[HR];
#pragma unroll
            for (int i = 0; i < HR; ++i) {
                float acc = 0.f;
                acc += dot4(wh[i][0], x0);
                acc += dot4(wh[i][1], x1);
                if (i == 0 && oq) acc += dot4(wobs, xo);             // (dot_seg's second segment: the lanes that hold a keypoint quad)
                vh[i] = acc;
            }
            // (round 6) the four row sums in one reduce-scatter: row i ends up in the lanes with bits 5:4 = i (rs_index<4, 64>), lane 16 i stores it
            const float htot = reduce_scatter<HR, 64>(vh, lane);
#pragma unroll
            for (int i = 0; i < HR; ++i) {
                if (lane == 32 * (i >> 1) + 16 * (i & 1)) {
                    const float v = htot + bh[i];
                    if (i == 0) {
                        if (sect == 0) gran_store(a.g_hidp + ((size_t)(a.nstat ? t : 0) * B + b) * 128 + rsec, lrelu(v, 0.01f), tag);
                        else if (sect == 1) {
                            gran_store(a.g_hidq + (size_t)b * 128 + rsec, lrelu(v, 0.01f), tag);
                            if (a.nstat) gran_store(a.g_sq + ((size_t)t * B + b) * 128 + rsec, lrelu(v, 0.01f), tag);
                        }
                        else if (sect == 2) gran_store(a.g_rh + (size_t)b * 128 + rsec, v, tag);
                        else gran_store(a.g_jh + (size_t)b * 128 + rsec, v, tag);
                    } else gran_store(a.g_gh + (size_t)b * 3 * H + gw + NWV * (i - 1), v, tag);
                }
            }
        }
        // ---- GRU unit j of step t: the nearest sample's keypoints | latent, W_hh h + b_hh of three h-phase rows
#pragma unroll 1
        for (int b = 0; b < B && wave_alive && j < H; ++b) {
            nm_f32x2 gd;
            const nm_gran* pd = a.g_d[t & 1] + (size_t)min(lane, S - 1) * B + b;
            NM_CHAIN_POLL(gran_ld_1(pd, gd), nm_fbits(gd[1]) == tag);
            if (!wave_alive) break;
            float bd = lane < S ? gd[0] : INFINITY; int bi = lane < S ? lane : 1 << 20;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = nm_sx(bd, off); const int oi = nm_sx(bi, off);
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            bi = __builtin_amdgcn_readfirstlane(bi);                       // (wave-uniform also when a NaN distance broke the order)
            const nm_gran* pk = a.g_kpz + ((size_t)bi * B + b) * (S4 + Z);
            const nm_gran* pg = a.g_gh + (size_t)b * 3 * H;
            const nm_gran* pp = t == 0 ? pg + jc : a.g_h[(t - 1) & 1] + (size_t)b * H + jc;
            f32x4 k0, k1, z0, z1; nm_f32x2 gr, gz, gn, gp;
            NM_CHAIN_POLL(gran_ld_4q4(pk + (kq ? lane * 4 : 0), pk + (kq ? lane * 4 : 0) + 2, pk + S4 + (zq ? lane * 4 : 0), pk + S4 + (zq ? lane * 4 : 0) + 2,
                                      pg + jc, pg + H + jc, pg + 2 * H + jc, pp, k0, k1, z0, z1, gr, gz, gn, gp),
                          nm_fbits(k0[1]) == tag && nm_fbits(k0[3]) == tag && nm_fbits(k1[1]) == tag && nm_fbits(k1[3]) == tag &&
                          nm_fbits(z0[1]) == tag && nm_fbits(z0[3]) == tag && nm_fbits(z1[1]) == tag && nm_fbits(z1[3]) == tag &&
                          nm_fbits(gr[1]) == tag && nm_fbits(gz[1]) == tag && nm_fbits(gn[1]) == tag && (t == 0 || nm_fbits(gp[1]) == (unsigned)t));
            if (!wave_alive) break;
            const float hpv = t == 0 ? a.h0[jc] : gp[0];
            const f32x4 xkq = {k0[0], k0[2], k1[0], k1[2]}, xzq = {z0[0], z0[2], z1[0], z1[2]};
            float ar = 0.f, az = 0.f, an = 0.f;
            if (kq) { ar += dot4(wk[0], xkq); az += dot4(wk[1], xkq); an += dot4(wk[2], xkq); }
            if (zq) { ar += dot4(wz[0], xzq); az += dot4(wz[1], xzq); an += dot4(wz[2], xzq); }
            {   // (round 6) the three gate sums in one reduce-scatter (a fourth, zero, pads it): sum i in the lanes with bits 5:4 = i
                float vg[4] = {ar, az, an, 0.f};
                const float gt = reduce_scatter<4, 64>(vg, lane);
                ar = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gt), 0));
                az = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gt), 16));
                an = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gt), 32));
            }
            if (lane == 0) {
                const float rg = sigmoidf((ar + bi3[0]) + gr[0]);
                const float zg = sigmoidf((az + bi3[1]) + gz[0]);
                const float ng = tanhf((an + bi3[2]) + rg * gn[0]);
                const float hn = (hpv - ng) * zg + ng;
                gran_store(a.g_h[t & 1] + (size_t)b * H + j, hn, tag);
                a.out_h[(size_t)b * a.ldh + (size_t)(t + 1) * H + j] = hn;
            }
        }
    }
}

// ---- posterior step of encode (inference), middle phases: one workgroup per (sample, batch element) ---------------------------------
// NOT the default (NM355_VRNN_POSTMID=1 selects it; parity-tested): measured against the six-launch step once the row kernels had
// been fixed (pick_nb), it loses - encode alone 52.6 us per timestep against 40.7, and in the forward +0.4 ms: a 1024-thread,
// 128-register workgroup only starts on a completely free CU, so beside the decoder's persistent convolutions it advances at their
// kernel boundaries only (8 of 16 steps done when the decoder ends, against 11), and the two device-scope fences of the selection
// write back / invalidate the L2 (40 us per launch in the forward's tail).  Kept as the measured alternative of BASELINE's "one
// kernel per timestep".
// hsvrnn_bvh.py:86-135: S posterior samples per clip and timestep, each decoded to keypoints; the sample nearest to the detected
// keypoints is kept.  Here the step is three launches like a prior step of the rollout: h-phase, THIS kernel, GRU.  Workgroup (s, b) is vrnn_prior_mid_kernel with
// the posterior's distribution (post2 on hid_post[b]) and noise eps[s][b], plus the squared distance of its keypoints to the
// observation; the sample-0 workgroup of a clip also evaluates the prior's parameters and the KL term.  The LAST workgroup of a
// clip to finish (device-scope counter behind a release fence; acquire fence before it reads the others' results) picks the
// nearest sample - lowest index on ties, fk_kernel's scan - and copies its keypoints / latent / rotations to the step's outputs.
// Arithmetic per row, the kinematic chain, the distance sum (joint order) and the KL reduction tree are fk_kernel's and the row
// kernels': the fused step is bit-identical to the six-launch step (tests/test_network_gpu.py).
struct PostArgs {
    const float *hid_post, *hid_prior, *rh, *jh, *eps, *offset, *obs; int ldobs;     // [B][128] x4, (S,B,Z), [B][K][3], [B][ldobs]
    const float *w_q2, *b_q2, *w_p2, *b_p2, *w_root0, *w_joint0, *w_root2, *b_root2, *w_joint2, *b_joint2;
    const int32_t *parents, *lvl_joint, *lvl_start; int nlevels;
    float *zall, *kpall, *rall, *dall;          // per (s, b): [S*B][Z], [S*B][128], [S*B][9K], [S*B]
    int32_t* counter;                           // [B], zero between launches
    float* out_kp; int ldkp; float* out_z; int ldz; float* out_R; int ldR; int32_t* best; int ldbest; float* kl; float* rec; int ldstat;
    int B, S, K, Z, H;
};

__global__ __launch_bounds__(1024) void vrnn_post_mid_kernel(PostArgs a) {
    __shared__ __attribute__((aligned(16))) float s_z[128], s_hr[128], s_hj[128], s_root[36], s_rot[192], s_mu[128], s_sg[128];
    __shared__ float s_Rl[32 * 9], s_Rg[32 * 9], s_pos[32 * 3], s_dk[32], s_red[256];
    __shared__ int s_last, s_best;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
    const int sb = blockIdx.x, b = sb % a.B, smp = sb / a.B;
    const int K = a.K, Z = a.Z, H = a.H, R0 = 3 + K, J6 = 6 * K, rows_c = R0 + J6;
    f32x4 wa[MID_PAIRS], wb[MID_PAIRS], wc[MID_PAIRS];
#pragma unroll
    for (int u = 0; u < MID_PAIRS; ++u) {
        const int p = wave * MID_PAIRS + u;
        wa[u] = *reinterpret_cast<const f32x4*>(a.w_q2 + (size_t)(p + half * Z) * 128 + l32 * 4);
        wb[u] = *reinterpret_cast<const f32x4*>((half ? a.w_joint0 : a.w_root0) + (size_t)p * (H + Z) + H + l32 * 4);
        const int r = min(p * 2 + half, rows_c - 1);
        const float* pc = r < R0 ? a.w_root2 + (size_t)r * 128 : a.w_joint2 + (size_t)(r - R0) * 128;
        wc[u] = *reinterpret_cast<const f32x4*>(pc + l32 * 4);
    }
    const f32x4 xh = *reinterpret_cast<const f32x4*>(a.hid_post + (size_t)b * 128 + l32 * 4);
    const int pl = wave * MID_PAIRS + (l32 < MID_PAIRS ? l32 : 0);
    const float bias_a = a.b_q2[pl], bias_a2 = a.b_q2[pl + Z], epsv = a.eps[(size_t)sb * Z + pl];
    const float add_b = (half ? a.jh : a.rh)[(size_t)b * 128 + pl];
    const int rl = min(pl * 2 + half, rows_c - 1);
    const float bias_c = *(rl < R0 ? a.b_root2 + rl : a.b_joint2 + (rl - R0));
    const float obsv = a.obs[(size_t)b * a.ldobs + min(tid, K * 4 - 1)];
    __shared__ FkTables tb;
    fk_tables_load(tb, a.lvl_joint, a.lvl_start, a.parents, a.offset + (size_t)b * K * 3, K, a.nlevels, tid);
    // ---- A. posterior distribution parameters and this workgroup's sample
    {
        float mine = 0.f, other = 0.f;
#pragma unroll
        for (int u = 0; u < MID_PAIRS; ++u) {
            const float v = half_reduce(dot4(wa[u], xh));
            const float o = nm_sx(v, 32);
            if (l32 == u) { mine = v; other = o; }
        }
        if (!half && l32 < MID_PAIRS) {
            const int p = wave * MID_PAIRS + l32;
            const float mu = mine + bias_a, sraw = other + bias_a2;
            const float sg = softplus(sraw) + 1e-4f;
            const float z = mu + epsv * sg;
            s_z[p] = z; s_mu[p] = mu; s_sg[p] = sg;
            a.zall[(size_t)sb * Z + p] = z;
        }
    }
    __syncthreads();
    // ---- B. decoder hidden layers
    {
        const f32x4 xz = *reinterpret_cast<const f32x4*>(s_z + l32 * 4);
        float mine = 0.f;
#pragma unroll
        for (int u = 0; u < MID_PAIRS; ++u) {
            const float v = half_reduce(dot4(wb[u], xz));
            if (l32 == u) mine = v;
        }
        if (l32 < MID_PAIRS) (half ? s_hj : s_hr)[wave * MID_PAIRS + l32] = lrelu(mine + add_b, 0.01f);
    }
    __syncthreads();
    // ---- C. heads
    {
        const f32x4 xr = *reinterpret_cast<const f32x4*>(s_hr + l32 * 4);
        const f32x4 xj = *reinterpret_cast<const f32x4*>(s_hj + l32 * 4);
        float mine = 0.f;
#pragma unroll
        for (int u = 0; u < MID_PAIRS; ++u) {
            const int r = (wave * MID_PAIRS + u) * 2 + half;
            const float v = half_reduce(dot4(wc[u], r < R0 ? xr : xj));
            if (l32 == u) mine = v;
        }
        if (l32 < MID_PAIRS) {
            const int r = (wave * MID_PAIRS + l32) * 2 + half;
            if (r < R0) s_root[r] = tanhf(mine + bias_c);
            else if (r < rows_c) s_rot[r - R0] = mine + bias_c;
        }
    }
    // (the sample-0 workgroup: the prior's rows into the registers phase A is done with - in flight across D)
    const bool klwg = smp == 0 && a.kl != nullptr;
    f32x4 xp = xh;
    float pb = 0.f, pb2 = 0.f;
    if (klwg) {
#pragma unroll
        for (int u = 0; u < MID_PAIRS; ++u) {
            const int p = wave * MID_PAIRS + u;
            wa[u] = *reinterpret_cast<const f32x4*>(a.w_p2 + (size_t)(p + half * Z) * 128 + l32 * 4);
        }
        xp = *reinterpret_cast<const f32x4*>(a.hid_prior + (size_t)b * 128 + l32 * 4);
        pb = a.b_p2[pl]; pb2 = a.b_p2[pl + Z];
    }
    __syncthreads();
    // ---- D. forward kinematics
    if (tid < K) {
        const float* p = s_rot + tid * 6;
        float x0 = p[0], x1 = p[1], x2 = p[2], y0 = p[3], y1 = p[4], y2 = p[5];
        float nx = sqrtf((x0 * x0 + x1 * x1) + x2 * x2) + 1e-10f;
        x0 /= nx; x1 /= nx; x2 /= nx;
        float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
        float nz = sqrtf((z0 * z0 + z1 * z1) + z2 * z2) + 1e-10f;
        z0 /= nz; z1 /= nz; z2 /= nz;
        float yy0 = z1 * x2 - z2 * x1, yy1 = z2 * x0 - z0 * x2, yy2 = z0 * x1 - z1 * x0;
        float* R = s_Rl + tid * 9;
        R[0] = x0; R[1] = yy0; R[2] = z0; R[3] = x1; R[4] = yy1; R[5] = z1; R[6] = x2; R[7] = yy2; R[8] = z2;
    }
    __syncthreads();
    fk_levels(tb, a.nlevels, s_Rl, s_Rg, s_pos, s_root, 36, K, 1, 0, 0, tid, 1024);
    // ---- E. this sample's keypoints, rotations, distance to the observation (fk_kernel: per-joint terms, summed in joint order)
    float kpv = 0.f;
    if (tid < K * 4) {
        const int k = tid >> 2, c = tid & 3;
        kpv = c < 3 ? s_pos[k * 3 + c] : (s_root[3 + k] + 1.0f) * 0.5f;
        a.kpall[(size_t)sb * 128 + tid] = kpv;
    }
    if (tid < K * 9) a.rall[(size_t)sb * 9 * K + tid] = s_Rg[tid];
    {
        // u_c = obs - kp per component; lanes 4k .. 4k+3 hold joint k's four: ((u0^2 + u1^2) + u2^2) + u3^2
        const float u = obsv - kpv, q = u * u;
        const float q1 = __shfl_down(q, 1), q2 = __shfl_down(q, 2), q3 = __shfl_down(q, 3);
        if (tid < K * 4 && (tid & 3) == 0) s_dk[tid >> 2] = ((q + q1) + q2) + q3;
    }
    __syncthreads();
    if (tid == 0) {
        float d = 0.f;
        for (int k = 0; k < K; ++k) d += s_dk[k];
        a.dall[sb] = d;
    }
    // ---- F. sample 0: prior parameters and the KL term (fk_kernel's 256-entry tree)
    if (klwg) {
        float mine = 0.f, other = 0.f;
#pragma unroll
        for (int u = 0; u < MID_PAIRS; ++u) {
            const float v = half_reduce(dot4(wa[u], xp));
            const float o = nm_sx(v, 32);
            if (l32 == u) { mine = v; other = o; }
        }
        if (tid < 256) s_red[tid] = 0.f;
        __syncthreads();
        if (!half && l32 < MID_PAIRS) {
            const int p = wave * MID_PAIRS + l32;
            const float pmu = mine + pb, psg = softplus(other + pb2) + 1e-4f;
            const float ratio = s_sg[p] / psg, vr = ratio * ratio, dm = (s_mu[p] - pmu) / psg;
            s_red[p] = 0.f + 0.5f * (((vr + dm * dm) - 1.0f) - logf(vr));
        }
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (tid < st) s_red[tid] += s_red[tid + st]; __syncthreads(); }
        if (tid == 0) a.kl[(size_t)b * a.ldstat] = s_red[0];
    }
    // ---- G. the clip's last workgroup selects
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(a.counter + b, 1) == a.S - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (tid == 0) {
        int bi = 0; float bd = __builtin_nontemporal_load(a.dall + b);
        for (int i = 1; i < a.S; ++i) { const float d = __builtin_nontemporal_load(a.dall + (size_t)i * a.B + b); if (d < bd) { bd = d; bi = i; } }
        s_best = bi;
        if (a.best) a.best[(size_t)b * a.ldbest] = bi;
        if (a.rec) a.rec[(size_t)b * a.ldstat] = bd;
        a.counter[b] = 0;
    }
    __syncthreads();
    const size_t src = (size_t)s_best * a.B + b;
    if (tid < K * 4) a.out_kp[(size_t)b * a.ldkp + tid] = __builtin_nontemporal_load(a.kpall + src * 128 + tid);
    if (tid < Z) a.out_z[(size_t)b * a.ldz + tid] = __builtin_nontemporal_load(a.zall + src * Z + tid);
    if (a.out_R && tid < K * 9) a.out_R[(size_t)b * a.ldR + tid] = __builtin_nontemporal_load(a.rall + src * 9 * K + tid);
}

// get_offset (hsvrnn_bvh.py:236-253): lower median over T of |p_k - p_parent(k)| times unit(offset_param[k])
__global__ __launch_bounds__(64) void offsets_kernel(const float* __restrict__ kp, const float* __restrict__ offset_param,
                                                     const int32_t* __restrict__ parents, int T, int K, float* __restrict__ out) {
    extern __shared__ float d[];       // [K][T]
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= K) return;
    const int par = parents[k];
    float* dk = d + k * T;
    for (int t = 0; t < T; ++t) {
        const float* p = kp + (((size_t)b * T + t) * K + k) * 4; const float* q = kp + (((size_t)b * T + t) * K + par) * 4;
        float u0 = p[0] - q[0], u1 = p[1] - q[1], u2 = p[2] - q[2];
        dk[t] = sqrtf((u0 * u0 + u1 * u1) + u2 * u2);
    }
    for (int i = 1; i < T; ++i) {      // insertion sort
        float v = dk[i]; int j = i - 1;
        while (j >= 0 && dk[j] > v) { dk[j + 1] = dk[j]; --j; }
        dk[j + 1] = v;
    }
    const float med = dk[(T - 1) / 2];
    const float* op = offset_param + k * 3;
    const float nrm = sqrtf((op[0] * op[0] + op[1] * op[1]) + op[2] * op[2]) + 1e-10f;
    float* o = out + ((size_t)b * K + k) * 3;
    o[0] = (op[0] / nrm) * med; o[1] = (op[1] / nrm) * med; o[2] = (op[2] / nrm) * med;
}

__global__ void broadcast_rows_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int ld, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * B) dst[(size_t)(i / n) * ld + i % n] = src[i % n];
}

// kl_kypt = mean over (B,T,Z); kypt_recon_loss = mean over (B,T)  (hsvrnn_bvh.py:137-151)
__global__ __launch_bounds__(256) void vrnn_stats_kernel(const float* __restrict__ kl, const float* __restrict__ rec, int n, int Z,
                                                         float* __restrict__ out2) {
    __shared__ float sa[256], sb[256];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { a += kl[i]; b += rec[i]; }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) { sa[threadIdx.x] += sa[threadIdx.x + st]; sb[threadIdx.x] += sb[threadIdx.x + st]; } __syncthreads(); }
    if (threadIdx.x == 0) { out2[0] = sa[0] / ((float)n * (float)Z); out2[1] = sb[0] / (float)n; }
}

// ------------------------------------------------------------------------------------------------------------------
// ===================================================================================================================
// Training (learner mode): back-propagation through time of HSVRNNBVH.encode
// (hsvrnn_bvh.py:67-156) for L = c_rec * kypt_recon_loss + c_kl * kl_kypt.  The best-of-S argmin is a constant of
// the backward pass (the reference's autograd only flows through the gathered sample), offsets are detached
// (hsvrnn_bvh.py:253), the detected keypoints are detached (neural_marionette.py:53).
// ===================================================================================================================
// per-step slices of the training tape (see VrnnTape)
struct StepTape { float *hid_p, *hid_q, *pmu, *psig, *praw, *qmu, *qsig, *qraw, *hr, *hj, *raw, *rot6, *Rl, *Rg, *eps, *gates; };

struct VrnnTape {
    int B = 0, T = 0, S = 0;
    bool valid = false;          // set by a completed training forward, cleared by nm_ctx_set_weights and by the backward
    float* base = nullptr; size_t cap = 0;
    // [T][B][...] planes
    float *hid_p, *hid_q, *pmu, *psig, *praw, *qmu, *qsig, *qraw, *hr, *hj, *raw, *rot6, *Rl, *Rg, *eps, *gates;
    float *kp_obs, *kp_rec, *z, *h, *offset;       // copies of the call's inputs / outputs ((B,T,..) layouts)
    StepTape at(int t, int K, int Z, int H) const {
        StepTape s; const size_t b = (size_t)t * B;
        s.hid_p = hid_p + b * 128; s.hid_q = hid_q + b * 128; s.pmu = pmu + b * Z; s.psig = psig + b * Z; s.praw = praw + b * Z;
        s.qmu = qmu + b * Z; s.qsig = qsig + b * Z; s.qraw = qraw + b * Z; s.hr = hr + b * 128; s.hj = hj + b * 128;
        s.raw = raw + b * (3 + K); s.rot6 = rot6 + b * 6 * K; s.Rl = Rl + b * 9 * K; s.Rg = Rg + b * 9 * K; s.eps = eps + b * Z;
        s.gates = gates + b * 4 * H;
        return s;
    }
};
// the tape belongs to the context (nm_ctx::vtape): two contexts in one process never see each other's forward
VrnnTape& ctx_tape(nm_ctx* c) {
    if (!c->vtape) c->vtape = new VrnnTape();
    return *static_cast<VrnnTape*>(c->vtape);
}

int tape_reserve(VrnnTape& tp, int B, int T, int S, int K, int Z, int H, hipStream_t s) {
    const size_t per_tb = 128 * 4 + 6 * Z + (3 + K) + 6 * K + 18 * K + Z + 4 * H;
    const size_t floats = (size_t)T * B * per_tb + (size_t)B * T * (K * 4 * 2 + Z) + (size_t)B * (T + 1) * H + (size_t)B * K * 3 + 1024;
    if (tp.cap < floats) {
        if (hipStreamSynchronize(s) != hipSuccess) return NM_ERR_HIP;
        if (tp.base) (void)hipFree(tp.base);
        tp.base = nullptr; tp.cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&tp.base), floats * sizeof(float)) != hipSuccess) { nm_set_error("vrnn tape: hipMalloc failed"); return NM_ERR_HIP; }
        tp.cap = floats;
    }
    tp.B = B; tp.T = T; tp.S = S;
    float* q = tp.base; const size_t tb = (size_t)T * B;
    auto take = [&](size_t n) { float* r = q; q += n; return r; };
    tp.hid_p = take(tb * 128); tp.hid_q = take(tb * 128); tp.pmu = take(tb * Z); tp.psig = take(tb * Z); tp.praw = take(tb * Z);
    tp.qmu = take(tb * Z); tp.qsig = take(tb * Z); tp.qraw = take(tb * Z); tp.hr = take(tb * 128); tp.hj = take(tb * 128);
    tp.raw = take(tb * (3 + K)); tp.rot6 = take(tb * 6 * K); tp.Rl = take(tb * 9 * K); tp.Rg = take(tb * 9 * K); tp.eps = take(tb * Z);
    tp.gates = take(tb * 4 * H);
    tp.kp_obs = take((size_t)B * T * K * 4); tp.kp_rec = take((size_t)B * T * K * 4); tp.z = take((size_t)B * T * Z);
    tp.h = take((size_t)B * (T + 1) * H); tp.offset = take((size_t)B * K * 3);
    return NM_OK;
}

// out[c][r] = in[r][c]
__global__ void transpose_kernel(const float* __restrict__ in, int rows, int cols, int ldin, float* __restrict__ out, int ldo, int col_off) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) { int r = r0 + j, c = c0 + tx; tile[j][tx] = (r < rows && c < cols) ? in[(size_t)r * ldin + c] : 0.f; }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) { int c = c0 + j, r = r0 + tx; if (c < cols && r < rows) out[(size_t)c * ldo + col_off + r] = tile[tx][j]; }
}
// out[c][col_off + r] = in[r][c] for r < rows, c < cols (in has leading dimension ldin)
int launch_transpose(const float* in, int rows, int cols, int ldin, float* out, int ldo, int col_off, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, in, rows, cols, ldin, out, ldo, col_off);
    return nm_check_hip(hipGetLastError(), "transpose launch");
}

// GRU cell backward (elementwise part): gates from the tape, dh_t -> dgi, dgh, dh_{t-1} (direct path)
__global__ void gru_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ gates, const float* __restrict__ hprev, int ldh,
                               int B, int H, float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dhprev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i % H;
    const size_t pl = (size_t)B * H;
    const float r = gates[i], z = gates[pl + i], n = gates[2 * pl + i], ghn = gates[3 * pl + i];
    const float d = dh[i], hp = hprev[(size_t)b * ldh + j];
    const float dn = d * (1.f - z), dz = d * (hp - n);
    const float dpn = dn * (1.f - n * n), dr = dpn * ghn;
    const float dpz = dz * z * (1.f - z), dpr = dr * r * (1.f - r);
    float* gi = dgi + (size_t)b * 3 * H; float* gh = dgh + (size_t)b * 3 * H;
    gi[j] = dpr; gi[H + j] = dpz; gi[2 * H + j] = dpn;
    gh[j] = dpr; gh[H + j] = dpz; gh[2 * H + j] = dpn * r;
    dhprev[i] = d * z;
}

__device__ __forceinline__ void cross3(const float* u, const float* v, float* o) {
    o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0];
}

// forward kinematics + 6-D rotation backward for one sample per block.
//   dx [B][ldx]: gradient wrt the GRU input (first K*4 entries = d kp*); adds the reconstruction-loss term.
//   outputs: draw [B][3+K] (already through tanh'), drot6 [B][6K]
__global__ __launch_bounds__(64) void fk_bwd_kernel(const float* __restrict__ dx, int ldx, const float* __restrict__ kp_rec,
                                                    const float* __restrict__ kp_obs, int ldkp, const float* __restrict__ dscal,
                                                    float inv_bt, const float* __restrict__ Rl, const float* __restrict__ Rg,
                                                    const float* __restrict__ offset, const float* __restrict__ raw,
                                                    const float* __restrict__ rot6, const int32_t* __restrict__ order,
                                                    const int32_t* __restrict__ parents, int K, float* __restrict__ draw,
                                                    float* __restrict__ drot6) {
    extern __shared__ float sm[];
    float* dpos = sm;               // [K][3]
    float* dRg = dpos + K * 3;      // [K][9]
    float* dRl = dRg + K * 9;       // [K][9]
    const int b = blockIdx.x, t = threadIdx.x;
    const float c_rec = dscal[1] * inv_bt;
    const float* rw = raw + (size_t)b * (3 + K);
    for (int i = t; i < K * 4; i += 64) {
        const int k = i >> 2, c = i & 3;
        const float g = dx[(size_t)b * ldx + i] + c_rec * 2.f * (kp_rec[(size_t)b * ldkp + i] - kp_obs[(size_t)b * ldkp + i]);
        if (c < 3) dpos[k * 3 + c] = g;
        else draw[(size_t)b * (3 + K) + 3 + k] = g * 0.5f * (1.f - rw[3 + k] * rw[3 + k]);
    }
    for (int i = t; i < K * 9; i += 64) { dRg[i] = 0.f; dRl[i] = 0.f; }
    __syncthreads();
    if (t == 0) {
        const float* RL = Rl + (size_t)b * 9 * K; const float* RG = Rg + (size_t)b * 9 * K;
        const float* off = offset + (size_t)b * K * 3;
        for (int o = K - 1; o >= 1; --o) {
            const int idx = order[o], par = parents[idx];
            float* gG = dRg + idx * 9;
            for (int r = 0; r < 3; ++r) {
                for (int m = 0; m < 3; ++m) gG[r * 3 + m] += dpos[idx * 3 + r] * off[idx * 3 + m];
                dpos[par * 3 + r] += dpos[idx * 3 + r];
            }
            const float* P = RG + par * 9; const float* L = RL + idx * 9;
            float* gL = dRl + idx * 9; float* gP = dRg + par * 9;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    gL[r * 3 + c] = (P[0 * 3 + r] * gG[0 * 3 + c] + P[1 * 3 + r] * gG[1 * 3 + c]) + P[2 * 3 + r] * gG[2 * 3 + c];      // P^T gG
                    gP[r * 3 + c] += (gG[r * 3 + 0] * L[c * 3 + 0] + gG[r * 3 + 1] * L[c * 3 + 1]) + gG[r * 3 + 2] * L[c * 3 + 2];    // gG L^T
                }
        }
        const int root = order[0];
        for (int e = 0; e < 9; ++e) dRl[root * 9 + e] = dRg[root * 9 + e];
        for (int c = 0; c < 3; ++c) draw[(size_t)b * (3 + K) + c] = dpos[root * 3 + c] * (1.f - rw[c] * rw[c]);
    }
    __syncthreads();
    if (t < K) {     // 6-D -> rotation backward (geo_utils.py:56-78)
        const float* p6 = rot6 + (size_t)b * 6 * K + t * 6;
        const float a[3] = {p6[0], p6[1], p6[2]}, bb[3] = {p6[3], p6[4], p6[5]};
        const float na = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]), da = na + 1e-10f;
        const float x[3] = {a[0] / da, a[1] / da, a[2] / da};
        float c[3]; cross3(x, bb, c);
        const float nc = sqrtf((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]), dc = nc + 1e-10f;
        const float z[3] = {c[0] / dc, c[1] / dc, c[2] / dc};
        const float* g = dRl + t * 9;
        float gx[3] = {g[0], g[3], g[6]}, gy[3] = {g[1], g[4], g[7]}, gz[3] = {g[2], g[5], g[8]};
        float tmp[3];
        cross3(x, gy, tmp); for (int i = 0; i < 3; ++i) gz[i] += tmp[i];          // y = z x x
        cross3(gy, z, tmp); for (int i = 0; i < 3; ++i) gx[i] += tmp[i];
        const float cg = c[0] * gz[0] + c[1] * gz[1] + c[2] * gz[2];
        float gc[3];
        for (int i = 0; i < 3; ++i) gc[i] = gz[i] / dc - (nc > 0.f ? c[i] * cg / (nc * dc * dc) : 0.f);
        cross3(bb, gc, tmp); for (int i = 0; i < 3; ++i) gx[i] += tmp[i];          // c = x x b
        float gb[3]; cross3(gc, x, gb);
        const float ag = a[0] * gx[0] + a[1] * gx[1] + a[2] * gx[2];
        float* o = drot6 + (size_t)b * 6 * K + t * 6;
        for (int i = 0; i < 3; ++i) { o[i] = gx[i] / da - (na > 0.f ? a[i] * ag / (na * da * da) : 0.f); o[3 + i] = gb[i]; }
    }
}

// reparameterisation + KL backward: dz* -> (d raw_q, d raw_p) of the second MLP layers (mu | pre-softplus std)
__global__ void dist_bwd_kernel(const float* __restrict__ dx, int ldx, int xoff, const float* __restrict__ dzdec, int lddz,
                                const float* __restrict__ eps, const float* __restrict__ qmu, const float* __restrict__ qsig,
                                const float* __restrict__ qraw, const float* __restrict__ pmu, const float* __restrict__ psig,
                                const float* __restrict__ praw, const float* __restrict__ dscal, float inv_btz, int B, int Z,
                                float* __restrict__ drq, float* __restrict__ drp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Z) return;
    const int b = i / Z, j = i % Z;
    const float dz = dx[(size_t)b * ldx + xoff + j] + dzdec[(size_t)b * lddz + j];
    const float ckl = dscal[0] * inv_btz;
    const float qs = qsig[i], ps = psig[i], d = qmu[i] - pmu[i], ips2 = 1.f / (ps * ps);
    const float dqm = dz + ckl * d * ips2, dpm = -ckl * d * ips2;
    const float dqs = dz * eps[i] + ckl * (qs * ips2 - 1.f / qs);
    const float dps = ckl * (-(qs * qs + d * d) * ips2 / ps + 1.f / ps);
    const float sq = qraw[i] > 20.f ? 1.f : sigmoidf(qraw[i]), sp = praw[i] > 20.f ? 1.f : sigmoidf(praw[i]);
    drq[(size_t)b * 2 * Z + j] = dqm; drq[(size_t)b * 2 * Z + Z + j] = dqs * sq;
    drp[(size_t)b * 2 * Z + j] = dpm; drp[(size_t)b * 2 * Z + Z + j] = dps * sp;
}

// weight gradient of a linear layer over all (t,b) samples: dW[r][k] = sum_s dA[s][r] * X[s][k], db[r] = sum_s dA[s][r].
// X = [xa | xb] with per-sample strides (samples are indexed s = t*B + b).  One wavefront per output row.
struct WgSeg { const float* p; int n; size_t st, sb; };
__global__ __launch_bounds__(256) void wgrad_rows_kernel(const float* __restrict__ dA, int ldA, int rows, WgSeg xa, WgSeg xb, int T, int B,
                                                         float* __restrict__ dW, float* __restrict__ db) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int in = xa.n + xb.n, S = T * B;
    for (int k0 = 0; k0 < in; k0 += 64) {
        const int k = k0 + lane;
        float acc = 0.f;
        if (k < in) {
            const bool first = k < xa.n;
            const WgSeg& sg = first ? xa : xb;
            const int kk = first ? k : k - xa.n;
            for (int s = 0; s < S; ++s) {
                const int t = s / B, b = s % B;
                acc += dA[(size_t)s * ldA + r] * sg.p[(size_t)t * sg.st + (size_t)b * sg.sb + kk];
            }
            dW[(size_t)r * in + k] = acc;
        }
    }
    if (db && lane == 0) { float a = 0.f; for (int s = 0; s < S; ++s) a += dA[(size_t)s * ldA + r]; db[r] = a; }
}

__global__ void colsum_kernel(const float* __restrict__ x, int rows, int cols, float* __restrict__ out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float a = 0.f;
    for (int r = 0; r < rows; ++r) a += x[(size_t)r * cols + c];
    out[c] = a;
}

// fused Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float lr, float b1, float b2, float eps, float bc1, float bc2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] = b1 * m[i] + (1.f - b1) * gi;
        const float vi = v[i] = b2 * v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

// the same update for a list of tensors in one launch: block b works on chunk b of the concatenation (table on the device)
struct AdamItem { float* p; const float* g; float* m; float* v; long long numel; long long chunk0; };
#define ADAM_CHUNK 4096
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamItem* __restrict__ items, int n, float lr, float b1, float b2, float eps,
                                                         float bc1, float bc2, const float* __restrict__ ok) {
    if (ok && *ok == 0.f) return;            // a step whose gradient bucket was not finite: parameters and both moments stay as they are
    int lo = 0, hi = n - 1;                   // last item whose first chunk is <= blockIdx.x
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].chunk0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1; }
    const AdamItem it = items[lo];
    const long long i0 = ((long long)blockIdx.x - it.chunk0) * ADAM_CHUNK, i1 = min(it.numel, i0 + ADAM_CHUNK);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float gi = it.g[i];
        const float mi = it.m[i] = b1 * it.m[i] + (1.f - b1) * gi;
        const float vi = it.v[i] = b2 * it.v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        it.p[i] = it.p[i] - (lr / bc1) * (mi / denom);
    }
}

struct StepBufs {
    float *hid_prior, *hid_post, *rh, *jh, *gh, *pmu, *psig, *qmu, *qsig, *z, *hr, *hj, *rootout, *rot;
    float *rall, *dall;   // vrnn_post_mid_kernel: every sample's global rotations [S*B][9K] and distance to the observation [S*B]
    float* gi;        // [B][3H] input projection of the GRU (large batches only, else null)
};

void add_job(LinJobs& J, const LinearW& L, int col0, const float* xa, int na, int lda, const float* xb, int nb, int ldb,
             bool bias, const float* add, int ldadd, int add_mod, float* out, int ldo, int act, int batch) {
    LinJob& j = J.j[J.n];
    j.W = L.w; j.ldw = L.in; j.col0 = col0; j.xa = xa; j.na = na; j.lda = lda; j.xb = xb; j.nb = nb; j.ldb = ldb;
    j.bias = bias ? L.b : nullptr; j.add = add; j.ldadd = ldadd; j.add_mod = add_mod > 0 ? add_mod : 1;
    j.out = out; j.ldo = ldo; j.rows = L.out; j.act = act; j.batch = batch; j.gate = nullptr; j.ldgate = 0;
    J.start[J.n + 1] = J.start[J.n] + L.out;
    J.n++;
}

// rows per wavefront pass: at most two.  The 4- and 8-row instantiations of the row kernels - meant to amortise a weight row over
// more batch rows - measure 30 / 22 us per launch (linear_rows / gru_rows, idle device, tools/time_vrnn_rows.py) where the 1- and
// 2-row ones take 4.8: with the batch split into more workgroups instead, nm_vrnn_gru at B = 4 / 8 / 16 runs 9.3 / 10.0 / 12.6 us
// instead of 50 / 53 / 64, and a posterior step of encode (B = 4, S = 10) 41 us instead of 95 (NM355_VRNN_NB = 4 / 8 restores them)
int pick_nb(int batch) {
    const int cap = nm_ls().vrnn_nb;
    const int nb = batch >= 8 ? 8 : (batch >= 4 ? 4 : (batch >= 2 ? 2 : 1));
    return nb < cap ? nb : cap;
}

// prior steps of a rollout (nm_ls().vrnn_mid, NM355_VRNN_MID): 1 (default) three dependent launches (h-phase, vrnn_prior_mid_kernel,
// GRU), 0 six.  A/B: profiles/r03_rollout_ab.txt (tools/time_rollout.py, bit-identical outputs).
#define NM_GEMM_MIN_BATCH 128

// batches of >= 128 rows take the MFMA GEMM (every input segment of this model is a multiple of 32 columns wide)
bool gemm_eligible(const LinJobs& J) {
    if (!nm_ls().vrnn_gemm) return false;
    for (int i = 0; i < J.n; ++i) {
        const LinJob& j = J.j[i];
        if (j.batch < NM_GEMM_MIN_BATCH || j.na % GM_KC || j.nb % GM_KC || (j.col0 & 3) || (j.ldw & 3) || (j.lda & 3) || (j.nb && (j.ldb & 3))) return false;
    }
    return J.n > 0;
}

int launch_jobs(const LinJobs& J, hipStream_t s) {
    if (gemm_eligible(J)) {
        GemmJobs G; G.n = J.n; G.tile0[0] = 0;
        for (int i = 0; i < J.n; ++i) {
            G.j[i] = J.j[i];
            G.mt[i] = (J.j[i].batch + GM_BM - 1) / GM_BM;
            G.tile0[i + 1] = G.tile0[i] + G.mt[i] * ((J.j[i].rows + GM_BN - 1) / GM_BN);
        }
        hipLaunchKernelGGL(gemm_rows_mfma_kernel, dim3(G.tile0[J.n]), dim3(256), 0, s, G);
        return nm_check_hip(hipGetLastError(), "gemm_rows_mfma launch");
    }
    int maxb = 0;
    for (int i = 0; i < J.n; ++i) maxb = J.j[i].batch > maxb ? J.j[i].batch : maxb;
    const int nb = pick_nb(maxb);
    dim3 grid((J.start[J.n] + 3) / 4, (maxb + nb - 1) / nb);
    if (nb == 1) hipLaunchKernelGGL((linear_rows_kernel<1>), grid, dim3(256), 0, s, J);
    else if (nb == 2) hipLaunchKernelGGL((linear_rows_kernel<2>), grid, dim3(256), 0, s, J);
    else if (nb == 4) hipLaunchKernelGGL((linear_rows_kernel<4>), grid, dim3(256), 0, s, J);
    else hipLaunchKernelGGL((linear_rows_kernel<8>), grid, dim3(256), 0, s, J);
    return nm_check_hip(hipGetLastError(), "linear_rows launch");
}

int launch_gru(const float* W_ih, const float* b_ih, const float* xa, int na, int lda, const float* xb, int nb_, int ldb,
               const float* gh, const float* h, int ldh, float* hout, int ldo, int H, int B, hipStream_t s, float* tg = nullptr,
               float* gi_scratch = nullptr) {
    if (gi_scratch && !tg && nm_ls().vrnn_gemm && B >= NM_GEMM_MIN_BATCH && na % GM_KC == 0 && nb_ % GM_KC == 0) {
        // large batch: input projection as a GEMM, then the element-wise gates
        LinJobs J; J.n = 0; J.start[0] = 0;
        LinearW ih; ih.in = na + nb_; ih.out = 3 * H; ih.w = const_cast<float*>(W_ih); ih.b = const_cast<float*>(b_ih);
        add_job(J, ih, 0, xa, na, lda, xb, nb_, ldb, true, nullptr, 0, 1, gi_scratch, 3 * H, 0, B);
        int rc = launch_jobs(J, s);
        if (rc) return rc;
        const size_t total = (size_t)B * H;
        hipLaunchKernelGGL(gru_gates_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gi_scratch, gh, h, ldh, hout, ldo, H, B);
        return nm_check_hip(hipGetLastError(), "gru_gates launch");
    }
    const int nb = pick_nb(B);
    dim3 grid((H + 3) / 4, (B + nb - 1) / nb);
    if (nb == 1) hipLaunchKernelGGL((gru_rows_kernel<1>), grid, dim3(256), 0, s, W_ih, b_ih, xa, na, lda, xb, nb_, ldb, gh, h, ldh, hout, ldo, H, B, tg);
    else if (nb == 2) hipLaunchKernelGGL((gru_rows_kernel<2>), grid, dim3(256), 0, s, W_ih, b_ih, xa, na, lda, xb, nb_, ldb, gh, h, ldh, hout, ldo, H, B, tg);
    else if (nb == 4) hipLaunchKernelGGL((gru_rows_kernel<4>), grid, dim3(256), 0, s, W_ih, b_ih, xa, na, lda, xb, nb_, ldb, gh, h, ldh, hout, ldo, H, B, tg);
    else hipLaunchKernelGGL((gru_rows_kernel<8>), grid, dim3(256), 0, s, W_ih, b_ih, xa, na, lda, xb, nb_, ldb, gh, h, ldh, hout, ldo, H, B, tg);
    return nm_check_hip(hipGetLastError(), "gru launch");
}

StepBufs alloc_step(Arena& ws, int B, int S, int K, int Z, int H) {
    StepBufs b;
    b.hid_prior = ws.f((size_t)B * 128); b.hid_post = ws.f((size_t)B * 128); b.rh = ws.f((size_t)B * 128); b.jh = ws.f((size_t)B * 128);
    b.gh = ws.f((size_t)B * 3 * H);
    b.pmu = ws.f((size_t)B * Z); b.psig = ws.f((size_t)B * Z); b.qmu = ws.f((size_t)B * Z); b.qsig = ws.f((size_t)B * Z);
    b.z = ws.f((size_t)S * B * Z); b.hr = ws.f((size_t)S * B * 128); b.hj = ws.f((size_t)S * B * 128);
    b.rootout = ws.f((size_t)S * B * (3 + K)); b.rot = ws.f((size_t)S * B * 6 * K);
    b.rall = ws.f((size_t)S * B * 9 * K); b.dall = ws.f((size_t)S * B);
    b.gi = B >= NM_GEMM_MIN_BATCH ? ws.f((size_t)B * 3 * H) : nullptr;
    return b;
}

struct StepIO {
    const StepTape* tape = nullptr;
    const float* h; int ldh;           // h_{t-1} [B][ldh]
    const float* obs; int ldobs;       // detected keypoints (posterior) or null (prior)
    const float* eps;                  // (S,B,Z) posterior / (B,Z) prior
    const float* offset;               // [B][K][3]
    float* out_kp; int ldkp; float* out_z; int ldz; float* out_R; int ldR;
    int32_t* best; int ldbest; float* kl; float* rec; int ldstat;
    float* hout; int ldho;
    bool want_prior;                   // also evaluate the prior (encode: KL)
};

// one VRNN timestep (posterior best-of-S when io.obs != null, otherwise a single prior sample)
int vrnn_step(nm_ctx* c, const StepBufs& sb, const StepIO& io, int B, int S) {
    const VrnnW& w = c->vrnn;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4;
    hipStream_t s = c->stream;
    const bool post = io.obs != nullptr;
    const bool prior = !post || io.want_prior;
    if (!post) S = 1;
    int rc;
    {   // 1. h-phase
        LinJobs J; J.n = 0; J.start[0] = 0;
        if (prior) add_job(J, w.prior0, 0, io.h, H, io.ldh, nullptr, 0, 0, true, nullptr, 0, 1, sb.hid_prior, 128, 1, B);
        if (post) add_job(J, w.post0, 0, io.h, H, io.ldh, io.obs, S4, io.ldobs, true, nullptr, 0, 1, sb.hid_post, 128, 1, B);
        add_job(J, w.root0, 0, io.h, H, io.ldh, nullptr, 0, 0, true, nullptr, 0, 1, sb.rh, 128, 0, B);
        add_job(J, w.joint0, 0, io.h, H, io.ldh, nullptr, 0, 0, true, nullptr, 0, 1, sb.jh, 128, 0, B);
        LinearW hh; hh.in = H; hh.out = 3 * H; hh.w = w.w_hh; hh.b = w.b_hh;
        add_job(J, hh, 0, io.h, H, io.ldh, nullptr, 0, 0, true, nullptr, 0, 1, sb.gh, 3 * H, 0, B);
        if ((rc = launch_jobs(J, s))) return rc;
    }
    if (!post && B <= 64 && !io.tape && !io.out_R && !io.best && !io.kl && !io.rec && nm_ls().vrnn_mid && K >= 2 && K <= 32 && Z == 128) {
        // 2-4 in one workgroup per batch element (prior steps of a rollout): see vrnn_prior_mid_kernel
        MidArgs a;
        a.hid_prior = sb.hid_prior; a.rh = sb.rh; a.jh = sb.jh; a.eps = io.eps; a.offset = io.offset;
        a.w_p2 = w.prior2.w; a.b_p2 = w.prior2.b; a.w_root0 = w.root0.w; a.w_joint0 = w.joint0.w;
        a.w_root2 = w.root2.w; a.b_root2 = w.root2.b; a.w_joint2 = w.joint2.w; a.b_joint2 = w.joint2.b;
        a.order = w.order; a.parents = w.parents; a.lvl_joint = w.lvl_joint; a.lvl_start = w.lvl_start; a.nlevels = w.nlevels;
        a.out_kp = io.out_kp; a.ldkp = io.ldkp; a.out_z = io.out_z; a.ldz = io.ldz;
        a.B = B; a.K = K; a.Z = Z; a.H = H;
        hipLaunchKernelGGL(vrnn_prior_mid_kernel, dim3(B), dim3(1024), 0, s, a);
        if ((rc = nm_check_hip(hipGetLastError(), "vrnn_prior_mid launch"))) return rc;
        if (io.hout && (rc = launch_gru(w.w_ih, w.b_ih, io.out_kp, S4, io.ldkp, io.out_z, Z, io.ldz, sb.gh, io.h, io.ldh, io.hout, io.ldho, H, B, s, nullptr, sb.gi))) return rc;
        return NM_OK;
    }
    if (post && !io.tape && nm_ls().vrnn_postmid && c->vrnn_cnt && B <= 256 && (size_t)S * B <= 4096 && K >= 2 && K <= 32 && Z == 128 && io.out_kp && io.out_z) {
        // 2-4 of a posterior step in one workgroup per (sample, clip), selection by the clip's last workgroup: vrnn_post_mid_kernel
        PostArgs a;
        a.hid_post = sb.hid_post; a.hid_prior = sb.hid_prior; a.rh = sb.rh; a.jh = sb.jh; a.eps = io.eps; a.offset = io.offset;
        a.obs = io.obs; a.ldobs = io.ldobs;
        a.w_q2 = w.post2.w; a.b_q2 = w.post2.b; a.w_p2 = w.prior2.w; a.b_p2 = w.prior2.b; a.w_root0 = w.root0.w; a.w_joint0 = w.joint0.w;
        a.w_root2 = w.root2.w; a.b_root2 = w.root2.b; a.w_joint2 = w.joint2.w; a.b_joint2 = w.joint2.b;
        a.parents = w.parents; a.lvl_joint = w.lvl_joint; a.lvl_start = w.lvl_start; a.nlevels = w.nlevels;
        a.zall = sb.z; a.kpall = sb.hr; a.rall = sb.rall; a.dall = sb.dall; a.counter = c->vrnn_cnt;
        a.out_kp = io.out_kp; a.ldkp = io.ldkp; a.out_z = io.out_z; a.ldz = io.ldz; a.out_R = io.out_R; a.ldR = io.ldR;
        a.best = io.best; a.ldbest = io.ldbest; a.kl = (io.want_prior && io.kl) ? io.kl : nullptr; a.rec = io.rec; a.ldstat = io.ldstat;
        a.B = B; a.S = S; a.K = K; a.Z = Z; a.H = H;
        hipLaunchKernelGGL(vrnn_post_mid_kernel, dim3(S * B), dim3(1024), 0, s, a);
        if ((rc = nm_check_hip(hipGetLastError(), "vrnn_post_mid launch"))) return rc;
        if (io.hout && (rc = launch_gru(w.w_ih, w.b_ih, io.out_kp, S4, io.ldkp, io.out_z, Z, io.ldz, sb.gh, io.h, io.ldh, io.hout, io.ldho, H, B, s, nullptr, sb.gi))) return rc;
        return NM_OK;
    }
    {   // 2. distribution parameters + samples
        DistJob jp{w.prior2.w, w.prior2.b, sb.hid_prior, sb.pmu, sb.psig, post ? nullptr : io.eps, sb.z, 1, io.tape ? io.tape->praw : nullptr};
        DistJob jq{w.post2.w, w.post2.b, sb.hid_post, sb.qmu, sb.qsig, io.eps, sb.z, S, io.tape ? io.tape->qraw : nullptr};
        DistJob first = prior ? jp : jq, second = jq;
        const int nj = (prior && post) ? 2 : 1;
        const int nb = pick_nb(B);
        dim3 grid((nj * Z + 3) / 4, (B + nb - 1) / nb);
        if (nb == 1) hipLaunchKernelGGL((dist_rows_kernel<1>), grid, dim3(256), 0, s, first, second, nj, Z, 128, B);
        else if (nb == 2) hipLaunchKernelGGL((dist_rows_kernel<2>), grid, dim3(256), 0, s, first, second, nj, Z, 128, B);
        else if (nb == 4) hipLaunchKernelGGL((dist_rows_kernel<4>), grid, dim3(256), 0, s, first, second, nj, Z, 128, B);
        else hipLaunchKernelGGL((dist_rows_kernel<8>), grid, dim3(256), 0, s, first, second, nj, Z, 128, B);
        if ((rc = nm_check_hip(hipGetLastError(), "dist_rows launch"))) return rc;
    }
    {   // 3. decoders: z-halves of the first layers (+ shared h-half), then the heads
        LinJobs J; J.n = 0; J.start[0] = 0;
        add_job(J, w.root0, H, sb.z, Z, Z, nullptr, 0, 0, false, sb.rh, 128, B, sb.hr, 128, 1, S * B);
        add_job(J, w.joint0, H, sb.z, Z, Z, nullptr, 0, 0, false, sb.jh, 128, B, sb.hj, 128, 1, S * B);
        if ((rc = launch_jobs(J, s))) return rc;
        LinJobs J2; J2.n = 0; J2.start[0] = 0;
        add_job(J2, w.root2, 0, sb.hr, 128, 128, nullptr, 0, 0, true, nullptr, 0, 1, sb.rootout, 3 + K, 2, S * B);
        add_job(J2, w.joint2, 0, sb.hj, 128, 128, nullptr, 0, 0, true, nullptr, 0, 1, sb.rot, 6 * K, 0, S * B);
        if ((rc = launch_jobs(J2, s))) return rc;
    }
    {   // 4. forward kinematics, best-of-S, KL
        FkArgs a;
        a.root = sb.rootout; a.ldr = 3 + K; a.rot = sb.rot; a.offset = io.offset; a.obs = io.obs; a.ldobs = io.ldobs; a.z = sb.z;
        a.order = w.order; a.parents = w.parents; a.lvl_joint = w.lvl_joint; a.lvl_start = w.lvl_start; a.nlevels = w.nlevels;
        const bool kl = post && io.want_prior && io.kl;
        a.qmu = kl ? sb.qmu : nullptr; a.qsig = sb.qsig; a.pmu = sb.pmu; a.psig = sb.psig;
        a.out_kp = io.out_kp; a.ldkp = io.ldkp; a.out_z = io.out_z; a.ldz = io.ldz; a.out_R = io.out_R; a.ldR = io.ldR;
        a.best = io.best; a.ldbest = io.ldbest; a.kl = kl ? io.kl : nullptr; a.rec = io.rec; a.ldstat = io.ldstat;
        a.K = K; a.S = S; a.B = B; a.Z = Z;
        a.hr = sb.hr; a.hj = sb.hj; a.eps = io.eps;
        a.t_hr = a.t_hj = a.t_raw = a.t_rot6 = a.t_Rl = a.t_Rg = a.t_eps = nullptr;
        if (io.tape) { a.t_hr = io.tape->hr; a.t_hj = io.tape->hj; a.t_raw = io.tape->raw; a.t_rot6 = io.tape->rot6; a.t_Rl = io.tape->Rl; a.t_Rg = io.tape->Rg; a.t_eps = io.tape->eps; }
        size_t lds = ((size_t)S * K * 22 + S) * sizeof(float);
        hipLaunchKernelGGL(fk_kernel, dim3(B), dim3(256), lds, s, a);
        if ((rc = nm_check_hip(hipGetLastError(), "fk launch"))) return rc;
    }
    if (io.hout) {   // 5. GRU
        if ((rc = launch_gru(w.w_ih, w.b_ih, io.out_kp, S4, io.ldkp, io.out_z, Z, io.ldz, sb.gh, io.h, io.ldh, io.hout, io.ldho, H, B, s, io.tape ? io.tape->gates : nullptr, sb.gi))) return rc;
    }
    return NM_OK;
}

int ready(nm_ctx* c, const char* who, bool need_tree) {
    if (!c) { nm_set_error("%s: null ctx", who); return NM_ERR_ARG; }
    if (!c->has_weights) { nm_set_error("%s: nm_ctx_set_weights has not been called", who); return NM_ERR_STATE; }
    if (need_tree && !c->vrnn.has_tree) { nm_set_error("%s: nm_vrnn_set_tree has not been called (the reference builds it in encode())", who); return NM_ERR_STATE; }
    int rc = nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
    if (!rc) rc = nm_nf_poll(c);            // deferred status of earlier calls (range guard, rollout time-out)
    return rc;
}

int max_fk_lds(nm_ctx* c, int S) {
    size_t lds = ((size_t)S * c->cfg.nkeypoints * 22 + S) * sizeof(float);
    if (lds > 60 * 1024) { nm_set_error("vrnn: S=%d samples exceed the FK kernel's LDS budget", S); return NM_ERR_UNSUPPORTED; }
    return NM_OK;
}

}  // namespace

void nm_vrnn_free_tape(nm_ctx* c) {
    VrnnTape* tp = static_cast<VrnnTape*>(c->vtape);
    if (!tp) return;
    if (tp->base) (void)hipFree(tp->base);
    delete tp;
    c->vtape = nullptr;
}
void nm_vrnn_invalidate_tape(nm_ctx* c) { if (c->vtape) static_cast<VrnnTape*>(c->vtape)->valid = false; }

extern "C" {

void nm_diag_set_chain_stamps(void* p) { g_chain_stamps = static_cast<long long*>(p); }

int nm_vrnn_set_tree(nm_ctx* c, const int32_t* parents, const int32_t* order) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_set_tree", false);
    if (rc) return rc;
    const int K = c->cfg.nkeypoints;
    if (!parents || !order) { nm_set_error("vrnn_set_tree: null argument"); return NM_ERR_ARG; }
    std::vector<char> seen(K, 0);
    for (int i = 0; i < K; ++i) {
        int k = order[i];
        if (k < 0 || k >= K || seen[k]) { nm_set_error("vrnn_set_tree: order is not a permutation"); return NM_ERR_ARG; }
        int p = parents[k];
        if (p < 0 || p >= K) { nm_set_error("vrnn_set_tree: parent out of range"); return NM_ERR_ARG; }
        if (i == 0 ? (p != k) : !seen[p]) { nm_set_error("vrnn_set_tree: parent of joint %d does not precede it in the order", k); return NM_ERR_ARG; }
        seen[k] = 1;
    }
    c->vrnn.parents_h.assign(parents, parents + K); c->vrnn.order_h.assign(order, order + K);
    // joints grouped by depth (level 0 = the root) for the level-parallel kinematic chain: a joint needs only its parent
    std::vector<int32_t> depth(K, 0), table(4 * K + 2, 0);
    int nlev = 1;
    for (int i = 1; i < K; ++i) { const int k = order[i]; depth[k] = depth[parents[k]] + 1; nlev = std::max(nlev, depth[k] + 1); }
    int32_t* lj = table.data() + 2 * K; int32_t* ls = table.data() + 3 * K;
    int n = 0;
    for (int l = 0; l < nlev; ++l) { ls[l] = n; for (int i = 0; i < K; ++i) if (depth[order[i]] == l) lj[n++] = order[i]; }
    ls[nlev] = n;
    std::copy(parents, parents + K, table.begin()); std::copy(order, order + K, table.begin() + K);
    c->vrnn.nlevels = nlev;
    c->vrnn.tree_epoch++;
    // (pageable host memory: the copy has left `table` when hipMemcpyAsync returns; the sync orders it before later launches on
    // other streams)
    rc = nm_check_hip(hipMemcpyAsync(c->vrnn.parents, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream), "set_tree copy");
    if (rc) return rc;
    rc = nm_check_hip(hipStreamSynchronize(c->stream), "set_tree sync");
    if (rc) return rc;
    c->vrnn.has_tree = true;
    return NM_OK;
} catch (...) { return nm_abi_catch("nm_vrnn_set_tree"); }

int nm_vrnn_offsets(nm_ctx* c, const float* keypoints, int32_t B, int32_t T, float* offset) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_offsets", true);
    if (rc) return rc;
    if (!keypoints || !offset || B <= 0 || T <= 0) { nm_set_error("vrnn_offsets: bad argument"); return NM_ERR_ARG; }
    const int K = c->cfg.nkeypoints;
    if ((size_t)K * T * sizeof(float) > 60 * 1024) { nm_set_error("vrnn_offsets: T too large"); return NM_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(offsets_kernel, dim3(B), dim3(64), (size_t)K * T * sizeof(float), c->stream, keypoints, c->vrnn.offset_param,
                       c->vrnn.parents, T, K, offset);
    return nm_check_hip(hipGetLastError(), "offsets launch");
} catch (...) { return nm_abi_catch("nm_vrnn_offsets"); }

// ---- host side of vrnn_post_chain_kernel ----------------------------------------------------------------------------------------------
// Tstat: steps of a launch WITH statistics workgroups (encode: the KL term), 0 for the conditioning steps of generate / rollouts
static size_t post_chain_granules(int B, int S, int K, int Z, int H, int Tstat) {
    const size_t nd = ((size_t)S * B + 1) & ~(size_t)1;              // (an even count: the arrays behind it are read in 16-byte granule pairs)
    return (size_t)B * (3 * 128 + 5 * H) + (size_t)S * B * (4 * K + Z) + 2 * nd + (size_t)2 * (Tstat > 0 ? Tstat : 1) * B * 128;
}
static size_t post_chain_gran_floats(int B, int S, int K, int Z, int H, int Tstat) { return 2 * post_chain_granules(B, S, K, Z, H, Tstat) + 64; }
// Can the chain run: shape limits, and EVERY workgroup resident at once - the kernel's workgroups spin on each other and an ordinary launch
// does not promise co-residency: workgroups of this shape (512 threads at up to 256 registers, the heads' weights in dynamic LDS) per
// CU x the device's CUs must cover the NM_CHAIN_NW workers + S B sample + nstat statistics workgroups (asked once per context).
static bool post_chain_fits(nm_ctx* c, int B, int S, int nstat, int& rc) {
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden;
    rc = NM_OK;
    if (!(nm_ls().vrnn_chain && S >= 1 && S <= 16 && (size_t)S * B <= 96 && B <= 16 && K >= 2 && K <= 32 && Z == 128 && H == 512 && c->nf_flag)) return false;
    if (nm_ls().post_chain_fits < 0) {
        static NmDeviceOnce attr_set;
        if (!attr_set.done()) {
            if ((rc = nm_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&vrnn_post_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "hipFuncSetAttribute(vrnn_post_chain)"))) return false;
            attr_set.mark();
        }
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&vrnn_post_chain_kernel), NM_CHAIN_T, (size_t)(3 + 7 * 32) * 128 * sizeof(float)) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->cfg.device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        nm_ls().post_chain_fits = per_cu * cus;         // resident workgroups the device can hold
    }
    return nm_ls().post_chain_fits >= NM_CHAIN_NW + S * B + nstat;
}
// T posterior steps from init_kypt_rnn_state: obs [B][ldobs] (step t at column t * 4K), eps [T][S][B][Z]; out_h [B][ldh] receives the
// state after step t at column (t + 1) * H.  out_R / best / kl / rec may be null (kl null: no statistics workgroups).  `gran`:
// post_chain_gran_floats() floats of scratch (zeroed here, on the stream).
static int launch_post_chain(nm_ctx* c, const float* obs, int ldobs, const float* eps, const float* offset, int B, int T, int S,
                             float* out_kp, int ldkp, float* out_z, int ldz, float* out_R, int ldR, float* out_h, int ldh,
                             int32_t* best, int ldbest, float* kl, float* rec, int ldstat, float* gran) {
    const VrnnW& w = c->vrnn;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4;
    const size_t nd = ((size_t)S * B + 1) & ~(size_t)1;
    const int Tstat = kl ? T : 0;
    const size_t ngran = post_chain_granules(B, S, K, Z, H, Tstat), nst = (size_t)(Tstat > 0 ? Tstat : 1) * B * 128;
    nm_gran* gb = reinterpret_cast<nm_gran*>(gran);
    PostChainArgs a;
    a.w_prior0 = w.prior0.w; a.b_prior0 = w.prior0.b; a.w_post0 = w.post0.w; a.b_post0 = w.post0.b;
    a.w_root0 = w.root0.w; a.b_root0 = w.root0.b; a.w_joint0 = w.joint0.w; a.b_joint0 = w.joint0.b;
    a.w_hh = w.w_hh; a.b_hh = w.b_hh; a.w_ih = w.w_ih; a.b_ih = w.b_ih;
    a.w_q2 = w.post2.w; a.b_q2 = w.post2.b; a.w_p2 = w.prior2.w; a.b_p2 = w.prior2.b;
    MidArgs& m = a.mid;
    m.hid_prior = m.rh = m.jh = m.eps = nullptr; m.offset = offset;
    m.w_p2 = w.prior2.w; m.b_p2 = w.prior2.b; m.w_root0 = w.root0.w; m.w_joint0 = w.joint0.w;
    m.w_root2 = w.root2.w; m.b_root2 = w.root2.b; m.w_joint2 = w.joint2.w; m.b_joint2 = w.joint2.b;
    m.order = w.order; m.parents = w.parents; m.lvl_joint = w.lvl_joint; m.lvl_start = w.lvl_start; m.nlevels = w.nlevels;
    m.out_kp = nullptr; m.ldkp = 0; m.out_z = nullptr; m.ldz = 0; m.B = B; m.K = K; m.Z = Z; m.H = H;
    a.h0 = w.h0; a.obs = obs; a.ldobs = ldobs; a.eps = eps;
    a.out_kp = out_kp; a.ldkp = ldkp; a.out_z = out_z; a.ldz = ldz; a.out_R = out_R; a.ldR = ldR;
    a.out_h = out_h; a.ldh = ldh; a.best = best; a.ldbest = ldbest; a.kl = kl; a.rec = rec; a.ldstat = ldstat;
    a.g_hidq = gb; a.g_rh = gb + (size_t)B * 128; a.g_jh = gb + (size_t)2 * B * 128;
    a.g_gh = gb + (size_t)3 * B * 128; a.g_kpz = a.g_gh + (size_t)B * 3 * H; a.g_d[0] = a.g_kpz + (size_t)S * B * (S4 + Z); a.g_d[1] = a.g_d[0] + nd;
    a.g_h[0] = a.g_d[1] + nd; a.g_h[1] = a.g_h[0] + (size_t)B * H;
    a.g_sq = a.g_h[1] + (size_t)B * H; a.g_hidp = a.g_sq + nst;
    a.abort = reinterpret_cast<unsigned*>(a.g_hidp + nst);
    a.status = c->nf_flag;
    a.B = B; a.S = S; a.T = T; a.K = K; a.Z = Z; a.H = H; a.nstat = kl ? B : 0;
    { static const int bo = getenv("NM355_CHAIN_BACKOFF") ? atoi(getenv("NM355_CHAIN_BACKOFF")) : 0; a.backoff = bo; }
    a.spin_limit = nm_ls().chain_spin > 0 ? nm_ls().chain_spin : NM_CHAIN_SPIN;
    a.stat_delay = nm_ls().chain_stat_delay;
    int rc;
    if ((rc = nm_check_hip(hipMemsetAsync(gb, 0, ngran * sizeof(nm_gran) + 64, c->stream), "vrnn post chain: granule buffers"))) return rc;
    const int drop = std::min(std::max(nm_ls().chain_drop, 0), B);          // (NM355_CHAIN_DROP_WG, test hook: see rollout_steps)
    hipLaunchKernelGGL(vrnn_post_chain_kernel, dim3(NM_CHAIN_NW + a.nstat + S * B - drop), dim3(NM_CHAIN_T), (size_t)(3 + 7 * K) * 128 * sizeof(float), c->stream, a);
    return nm_check_hip(hipGetLastError(), "vrnn_post_chain launch");
}

static int encode_impl(nm_ctx* c, const float* keypoints, const float* eps, int32_t B, int32_t T, int32_t S, float* kypt_recon,
                       float* R, float* z, float* h, float* scalars2, int32_t* best_idx, bool train) {
    int rc = ready(c, "vrnn_encode", true);
    if (rc) return rc;
    if (!keypoints || !eps || !kypt_recon || !R || !z || !h || !scalars2 || B <= 0 || T <= 0 || S <= 0) {
        nm_set_error("vrnn_encode: null / non-positive argument (eps must be supplied explicitly)"); return NM_ERR_ARG;
    }
    if ((rc = max_fk_lds(c, S))) return rc;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4;
    size_t need = ((size_t)B * (4 * 128 + 6 * H + 4 * Z + K * 3 + 2 * T) + (size_t)S * B * (Z + 256 + 3 + K + 6 * K + 9 * K + 1)) * sizeof(float) + 64 * 256 +
                  ((size_t)B * (4 * 128 + 5 * H) + (size_t)S * B * (S4 + Z + 1)) * sizeof(nm_gran) + 4096;          // (+ the posterior chain's granule buffers)
    if ((rc = nm_ctx_reserve(c, need))) return rc;
    c->ws.release(0);
    StepBufs sb = alloc_step(c->ws, B, S, K, Z, H);
    float* offset = c->ws.f((size_t)B * K * 3);
    float* kl = c->ws.f((size_t)B * T); float* rec = c->ws.f((size_t)B * T);
    if (c->ws.overflow) { nm_set_error("vrnn_encode: workspace overflow"); return NM_ERR_STATE; }
    VrnnTape* tpp = train ? &ctx_tape(c) : nullptr;
    if (train) { tpp->valid = false; if ((rc = tape_reserve(*tpp, B, T, S, K, Z, H, c->stream))) return rc; }
    if ((rc = nm_vrnn_offsets(c, keypoints, B, T, offset))) return rc;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3((H * B + 255) / 256), dim3(256), 0, c->stream, c->vrnn.h0, H, h, (T + 1) * H, B);
    // The T posterior steps as ONE persistent launch (vrnn_post_chain_kernel) when the shape allows and every workgroup of it can be
    // resident.  Stand-alone encode only (NM355_VRNN_POST_CHAIN: 0 off, 1 default, 2 also inside nm_forward_fused): beside the decoder's
    // one-workgroup-per-CU persistent convolutions the chain's ~100 spinning workgroups take their CUs for the whole encode, where the
    // launch-per-phase steps run in the gaps for free (A/B: profiles/r05_encode_ab.txt: +0.17 ms per forward step).  The training forward
    // keeps the launches (it fills the tape).
    const int pc = nm_ls().vrnn_post_chain;
    bool chain = !train && pc && (pc >= 2 || !c->in_fused) && post_chain_fits(c, B, S, B, rc);
    if (rc) return rc;
    if (chain) {
        float* gran = c->ws.f(post_chain_gran_floats(B, S, K, Z, H, T));
        if (c->ws.overflow) { nm_set_error("vrnn_encode: workspace overflow"); return NM_ERR_STATE; }
        if ((rc = launch_post_chain(c, keypoints, T * S4, eps, offset, B, T, S, kypt_recon, T * S4, z, T * Z, R, T * K * 9, h, (T + 1) * H, best_idx, T,
                                    kl, rec, T, gran))) return rc;
    }
    for (int t = 0; t < (chain ? 0 : T); ++t) {
        StepIO io;
        io.h = h + (size_t)t * H; io.ldh = (T + 1) * H;
        io.obs = keypoints + (size_t)t * S4; io.ldobs = T * S4;
        io.eps = eps + (size_t)t * S * B * Z; io.offset = offset;
        io.out_kp = kypt_recon + (size_t)t * S4; io.ldkp = T * S4;
        io.out_z = z + (size_t)t * Z; io.ldz = T * Z;
        io.out_R = R + (size_t)t * K * 9; io.ldR = T * K * 9;
        io.best = best_idx ? best_idx + t : nullptr; io.ldbest = T;
        io.kl = kl + t; io.rec = rec + t; io.ldstat = T;
        io.hout = h + (size_t)(t + 1) * H; io.ldho = (T + 1) * H;
        io.want_prior = true;
        StepTape st; StepBufs sbt = sb;
        if (train) {   // this step's hidden layers / distribution parameters land in the tape instead of the scratch
            st = tpp->at(t, K, Z, H); io.tape = &st;
            sbt.hid_prior = st.hid_p; sbt.hid_post = st.hid_q; sbt.pmu = st.pmu; sbt.psig = st.psig; sbt.qmu = st.qmu; sbt.qsig = st.qsig;
        }
        if ((rc = vrnn_step(c, sbt, io, B, S))) return rc;
    }
    if (train) {
        hipStream_t s = c->stream;
        const struct { float* dst; const float* src; size_t n; } cp[5] = {
            {tpp->kp_obs, keypoints, (size_t)B * T * S4}, {tpp->kp_rec, kypt_recon, (size_t)B * T * S4}, {tpp->z, z, (size_t)B * T * Z},
            {tpp->h, h, (size_t)B * (T + 1) * H}, {tpp->offset, offset, (size_t)B * K * 3}};
        for (const auto& e : cp)
            if ((rc = nm_check_hip(hipMemcpyAsync(e.dst, e.src, e.n * sizeof(float), hipMemcpyDeviceToDevice, s), "vrnn_encode_train: tape copy"))) return rc;
    }
    hipLaunchKernelGGL(vrnn_stats_kernel, dim3(1), dim3(256), 0, c->stream, kl, rec, B * T, Z, scalars2);
    rc = nm_check_hip(hipGetLastError(), "vrnn_encode");
    if (train) tpp->valid = rc == NM_OK;
    return rc;
}

int nm_vrnn_encode(nm_ctx* c, const float* keypoints, const float* eps, int32_t B, int32_t T, int32_t S, float* kypt_recon,
                   float* R, float* z, float* h, float* scalars2, int32_t* best_idx) try { NmScope nm_scope_(c);
    return encode_impl(c, keypoints, eps, B, T, S, kypt_recon, R, z, h, scalars2, best_idx, false);
} catch (...) { return nm_abi_catch("nm_vrnn_encode"); }

int nm_vrnn_encode_train(nm_ctx* c, const float* keypoints, const float* eps, int32_t B, int32_t T, int32_t S, float* kypt_recon,
                         float* R, float* z, float* h, float* scalars2, int32_t* best_idx) try { NmScope nm_scope_(c);
    return encode_impl(c, keypoints, eps, B, T, S, kypt_recon, R, z, h, scalars2, best_idx, true);
} catch (...) { return nm_abi_catch("nm_vrnn_encode_train"); }

static int launch_wgrad(const float* dA, int ldA, int rows, WgSeg xa, WgSeg xb, int T, int B, float* dW, float* db, hipStream_t s) {
    hipLaunchKernelGGL(wgrad_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, dA, ldA, rows, xa, xb, T, B, dW, db);
    return nm_check_hip(hipGetLastError(), "wgrad launch");
}

int nm_vrnn_encode_backward(nm_ctx* c, const float* dscal2, const nm_named_grad* grads, int32_t count) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_encode_backward", true);
    if (rc) return rc;
    VrnnTape& tp = ctx_tape(c);
    if (!tp.base || tp.B <= 0 || !tp.valid) {
        nm_set_error("vrnn_encode_backward: no recorded forward on this context (call nm_vrnn_encode_train first; a tape is consumed "
                     "by one backward and dropped by nm_ctx_set_weights)");
        return NM_ERR_STATE;
    }
    tp.valid = false;            // one backward per forward
    if (!dscal2 || !grads || count <= 0) { nm_set_error("vrnn_encode_backward: bad argument"); return NM_ERR_ARG; }
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4, B = tp.B, T = tp.T, R0 = 3 + K, J6 = 6 * K;
    const VrnnW& w = c->vrnn;
    std::map<std::string, std::pair<float*, int64_t>> out;
    for (int i = 0; i < count; ++i) if (grads[i].name && grads[i].data) out[grads[i].name] = std::make_pair(grads[i].data, grads[i].numel);
    auto G = [&](const char* name, int64_t numel) -> float* {
        auto it = out.find(std::string("dyna_module.") + name);
        if (it == out.end() || it->second.second != numel) { if (!rc) { nm_set_error("vrnn_encode_backward: gradient buffer '%s' missing or wrong size", name); rc = NM_ERR_ARG; } return nullptr; }
        return it->second.first;
    };
    float* g_ih = G("kypt_rnn_cell.weight_ih", (int64_t)3 * H * (S4 + Z)); float* g_hh = G("kypt_rnn_cell.weight_hh", (int64_t)3 * H * H);
    float* g_bih = G("kypt_rnn_cell.bias_ih", 3 * H); float* g_bhh = G("kypt_rnn_cell.bias_hh", 3 * H);
    float* g_r0 = G("root_intensity_decoder.0.weight", (int64_t)128 * (H + Z)); float* g_r0b = G("root_intensity_decoder.0.bias", 128);
    float* g_r2 = G("root_intensity_decoder.2.weight", (int64_t)R0 * 128); float* g_r2b = G("root_intensity_decoder.2.bias", R0);
    float* g_j0 = G("joint_matrix_decoder.0.weight", (int64_t)128 * (H + Z)); float* g_j0b = G("joint_matrix_decoder.0.bias", 128);
    float* g_j2 = G("joint_matrix_decoder.2.weight", (int64_t)J6 * 128); float* g_j2b = G("joint_matrix_decoder.2.bias", J6);
    float* g_q0 = G("extract_post_dist.0.weight", (int64_t)128 * (H + S4)); float* g_q0b = G("extract_post_dist.0.bias", 128);
    float* g_q2 = G("extract_post_dist.2.weight", (int64_t)2 * Z * 128); float* g_q2b = G("extract_post_dist.2.bias", 2 * Z);
    float* g_p0 = G("extract_prior_dist.0.weight", (int64_t)128 * H); float* g_p0b = G("extract_prior_dist.0.bias", 128);
    float* g_p2 = G("extract_prior_dist.2.weight", (int64_t)2 * Z * 128); float* g_p2b = G("extract_prior_dist.2.bias", 2 * Z);
    float* g_h0 = G("init_kypt_rnn_state", H);
    if (rc) return rc;

    const size_t tb = (size_t)T * B;
    const size_t fl = (size_t)(S4 + Z) * 3 * H + (size_t)H * 3 * H + 128 * R0 + 128 * J6 + (size_t)(H + Z) * 256 + 2 * 128 * 2 * Z + (size_t)H * 256 +
                      tb * (6 * H + R0 + J6 + 4 * 128 + 4 * Z) + (size_t)B * (2 * H + S4 + Z + Z) + 4096;
    if ((rc = nm_ctx_reserve(c, fl * sizeof(float)))) return rc;
    Arena& ws = c->ws; ws.release(0);
    hipStream_t s = c->stream;
    float* WihT = ws.f((size_t)(S4 + Z) * 3 * H); float* WhhT = ws.f((size_t)H * 3 * H);
    float* Wr2T = ws.f(128 * R0); float* Wj2T = ws.f(128 * J6); float* WdecT = ws.f((size_t)(H + Z) * 256);
    float* Wq2T = ws.f(128 * 2 * Z); float* Wp2T = ws.f(128 * 2 * Z); float* WpqT = ws.f((size_t)H * 256);
    float* dgi = ws.f(tb * 3 * H); float* dgh = ws.f(tb * 3 * H); float* draw = ws.f(tb * R0); float* drot6 = ws.f(tb * J6);
    float* da_r = ws.f(tb * 128); float* da_j = ws.f(tb * 128); float* da_q = ws.f(tb * 128); float* da_p = ws.f(tb * 128);
    float* drq = ws.f(tb * 2 * Z); float* drp = ws.f(tb * 2 * Z);
    float* dh[2] = {ws.f((size_t)B * H), ws.f((size_t)B * H)};
    float* dx = ws.f((size_t)B * (S4 + Z)); float* dzdec = ws.f((size_t)B * Z);
    if (ws.overflow) { nm_set_error("vrnn_encode_backward: workspace overflow"); return NM_ERR_STATE; }

    // transposed weights (the backward products x = dy W are row dots of W^T)
    if ((rc = launch_transpose(w.w_ih, 3 * H, S4 + Z, S4 + Z, WihT, 3 * H, 0, s))) return rc;
    if ((rc = launch_transpose(w.w_hh, 3 * H, H, H, WhhT, 3 * H, 0, s))) return rc;
    if ((rc = launch_transpose(w.root2.w, R0, 128, 128, Wr2T, R0, 0, s))) return rc;
    if ((rc = launch_transpose(w.joint2.w, J6, 128, 128, Wj2T, J6, 0, s))) return rc;
    if ((rc = launch_transpose(w.root0.w, 128, H + Z, H + Z, WdecT, 256, 0, s))) return rc;
    if ((rc = launch_transpose(w.joint0.w, 128, H + Z, H + Z, WdecT, 256, 128, s))) return rc;
    if ((rc = launch_transpose(w.post2.w, 2 * Z, 128, 128, Wq2T, 2 * Z, 0, s))) return rc;
    if ((rc = launch_transpose(w.prior2.w, 2 * Z, 128, 128, Wp2T, 2 * Z, 0, s))) return rc;
    if ((rc = launch_transpose(w.post0.w, 128, H, H + S4, WpqT, 256, 0, s))) return rc;
    if ((rc = launch_transpose(w.prior0.w, 128, H, H, WpqT, 256, 128, s))) return rc;
    if ((rc = nm_check_hip(hipMemsetAsync(dh[0], 0, (size_t)B * H * sizeof(float), s), "memset dh"))) return rc;

    auto job = [&](LinJobs& J, const float* W, int ldw, int rows, const float* xa, int na, int lda, const float* xb, int nb, int ldb,
                   const float* add, int ldadd, float* o, int ldo, const float* gate, int ldgate) {
        LinearW L; L.w = const_cast<float*>(W); L.b = nullptr; L.in = ldw; L.out = rows;
        add_job(J, L, 0, xa, na, lda, xb, nb, ldb, false, add, ldadd, B, o, ldo, 0, B);
        J.j[J.n - 1].gate = gate; J.j[J.n - 1].ldgate = ldgate;
    };
    int cur = 0;
    const float inv_bt = 1.0f / (float)(B * T), inv_btz = 1.0f / ((float)B * (float)T * (float)Z);
    for (int t = T - 1; t >= 0; --t) {
        const StepTape st = tp.at(t, K, Z, H);
        const float* hprev = tp.h + (size_t)t * H;                        // (B, T+1, H) layout
        float* dgi_t = dgi + (size_t)t * B * 3 * H; float* dgh_t = dgh + (size_t)t * B * 3 * H;
        float* draw_t = draw + (size_t)t * B * R0; float* drot_t = drot6 + (size_t)t * B * J6;
        float* dar = da_r + (size_t)t * B * 128; float* daj = da_j + (size_t)t * B * 128;
        float* daq = da_q + (size_t)t * B * 128; float* dap = da_p + (size_t)t * B * 128;
        float* drq_t = drq + (size_t)t * B * 2 * Z; float* drp_t = drp + (size_t)t * B * 2 * Z;
        float* dhp = dh[cur ^ 1];
        hipLaunchKernelGGL(gru_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, s, dh[cur], st.gates, hprev, (T + 1) * H, B, H, dgi_t, dgh_t, dhp);
        { LinJobs J; J.n = 0; J.start[0] = 0;
          job(J, WihT, 3 * H, S4 + Z, dgi_t, 3 * H, 3 * H, nullptr, 0, 0, nullptr, 0, dx, S4 + Z, nullptr, 0);
          job(J, WhhT, 3 * H, H, dgh_t, 3 * H, 3 * H, nullptr, 0, 0, dhp, H, dhp, H, nullptr, 0);
          if ((rc = launch_jobs(J, s))) return rc; }
        hipLaunchKernelGGL(fk_bwd_kernel, dim3(B), dim3(64), (size_t)K * 21 * sizeof(float), s, dx, S4 + Z, tp.kp_rec + (size_t)t * S4,
                           tp.kp_obs + (size_t)t * S4, T * S4, dscal2, inv_bt, st.Rl, st.Rg, tp.offset, st.raw, st.rot6, w.order, w.parents, K,
                           draw_t, drot_t);
        { LinJobs J; J.n = 0; J.start[0] = 0;
          job(J, Wr2T, R0, 128, draw_t, R0, R0, nullptr, 0, 0, nullptr, 0, dar, 128, st.hr, 128);
          job(J, Wj2T, J6, 128, drot_t, J6, J6, nullptr, 0, 0, nullptr, 0, daj, 128, st.hj, 128);
          if ((rc = launch_jobs(J, s))) return rc; }
        { LinJobs J; J.n = 0; J.start[0] = 0;
          job(J, WdecT, 256, H, dar, 128, 128, daj, 128, 128, dhp, H, dhp, H, nullptr, 0);
          job(J, WdecT + (size_t)H * 256, 256, Z, dar, 128, 128, daj, 128, 128, nullptr, 0, dzdec, Z, nullptr, 0);
          if ((rc = launch_jobs(J, s))) return rc; }
        hipLaunchKernelGGL(dist_bwd_kernel, dim3((B * Z + 255) / 256), dim3(256), 0, s, dx, S4 + Z, S4, dzdec, Z, st.eps, st.qmu, st.qsig, st.qraw,
                           st.pmu, st.psig, st.praw, dscal2, inv_btz, B, Z, drq_t, drp_t);
        { LinJobs J; J.n = 0; J.start[0] = 0;
          job(J, Wq2T, 2 * Z, 128, drq_t, 2 * Z, 2 * Z, nullptr, 0, 0, nullptr, 0, daq, 128, st.hid_q, 128);
          job(J, Wp2T, 2 * Z, 128, drp_t, 2 * Z, 2 * Z, nullptr, 0, 0, nullptr, 0, dap, 128, st.hid_p, 128);
          if ((rc = launch_jobs(J, s))) return rc; }
        { LinJobs J; J.n = 0; J.start[0] = 0;
          job(J, WpqT, 256, H, daq, 128, 128, dap, 128, 128, dhp, H, dhp, H, nullptr, 0);
          if ((rc = launch_jobs(J, s))) return rc; }
        cur ^= 1;
    }
    // init_kypt_rnn_state is expanded over the batch: its gradient is the batch sum of dh_0
    hipLaunchKernelGGL(colsum_kernel, dim3((H + 255) / 256), dim3(256), 0, s, dh[cur], B, H, g_h0);

    // weight gradients: one GEMM-like pass per layer over all (t, b) samples
    const size_t hB = (size_t)(T + 1) * H;
    WgSeg none{nullptr, 0, 0, 0};
    WgSeg x_h{tp.h, H, (size_t)H, hB};                                       // h_{t-1}: (B, T+1, H)
    WgSeg x_kp{tp.kp_rec, S4, (size_t)S4, (size_t)T * S4}, x_obs{tp.kp_obs, S4, (size_t)S4, (size_t)T * S4};
    WgSeg x_z{tp.z, Z, (size_t)Z, (size_t)T * Z};
    WgSeg x_hr{tp.hr, 128, (size_t)B * 128, 128}, x_hj{tp.hj, 128, (size_t)B * 128, 128};
    WgSeg x_uq{tp.hid_q, 128, (size_t)B * 128, 128}, x_up{tp.hid_p, 128, (size_t)B * 128, 128};
    if ((rc = launch_wgrad(dgi, 3 * H, 3 * H, x_kp, x_z, T, B, g_ih, g_bih, s))) return rc;
    if ((rc = launch_wgrad(dgh, 3 * H, 3 * H, x_h, none, T, B, g_hh, g_bhh, s))) return rc;
    if ((rc = launch_wgrad(draw, R0, R0, x_hr, none, T, B, g_r2, g_r2b, s))) return rc;
    if ((rc = launch_wgrad(drot6, J6, J6, x_hj, none, T, B, g_j2, g_j2b, s))) return rc;
    if ((rc = launch_wgrad(da_r, 128, 128, x_h, x_z, T, B, g_r0, g_r0b, s))) return rc;
    if ((rc = launch_wgrad(da_j, 128, 128, x_h, x_z, T, B, g_j0, g_j0b, s))) return rc;
    if ((rc = launch_wgrad(drq, 2 * Z, 2 * Z, x_uq, none, T, B, g_q2, g_q2b, s))) return rc;
    if ((rc = launch_wgrad(drp, 2 * Z, 2 * Z, x_up, none, T, B, g_p2, g_p2b, s))) return rc;
    if ((rc = launch_wgrad(da_q, 128, 128, x_h, x_obs, T, B, g_q0, g_q0b, s))) return rc;
    if ((rc = launch_wgrad(da_p, 128, 128, x_h, none, T, B, g_p0, g_p0b, s))) return rc;
    return nm_check_hip(hipGetLastError(), "vrnn_encode_backward");
} catch (...) { return nm_abi_catch("nm_vrnn_encode_backward"); }

int nm_adam_step(nm_ctx* c, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, int32_t step, float lr,
                 float beta1, float beta2, float eps) try { NmScope nm_scope_(c);
    if (!c || !param || !grad || !exp_avg || !exp_avg_sq || numel <= 0 || step <= 0) { nm_set_error("adam_step: bad argument"); return NM_ERR_ARG; }
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    int blocks = (int)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, c->stream, param, grad, exp_avg, exp_avg_sq, (size_t)numel, lr, beta1, beta2, eps, bc1, bc2);
    return nm_check_hip(hipGetLastError(), "adam launch");
} catch (...) { return nm_abi_catch("nm_adam_step"); }

static int adam_multi_impl(nm_ctx* c, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                           const int64_t* numels, int32_t count, int32_t step, float lr, float beta1, float beta2, float eps, const float* ok);
int nm_adam_step_multi(nm_ctx* c, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const int64_t* numels, int32_t count, int32_t step, float lr, float beta1, float beta2, float eps) try { NmScope nm_scope_(c);
    return adam_multi_impl(c, params, grads, exp_avg, exp_avg_sq, numels, count, step, lr, beta1, beta2, eps, nullptr);
} catch (...) { return nm_abi_catch("nm_adam_step_multi"); }
int nm_adam_step_multi_ok(nm_ctx* c, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                          const int64_t* numels, int32_t count, int32_t step, float lr, float beta1, float beta2, float eps, const float* ok) try { NmScope nm_scope_(c);
    return adam_multi_impl(c, params, grads, exp_avg, exp_avg_sq, numels, count, step, lr, beta1, beta2, eps, ok);
} catch (...) { return nm_abi_catch("nm_adam_step_multi_ok"); }
static int adam_multi_impl(nm_ctx* c, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                           const int64_t* numels, int32_t count, int32_t step, float lr, float beta1, float beta2, float eps, const float* ok) {
    if (!c || !params || !grads || !exp_avg || !exp_avg_sq || !numels || count <= 0 || step <= 0) { nm_set_error("adam_step_multi: bad argument"); return NM_ERR_ARG; }
    c->host_table.resize((size_t)count * sizeof(AdamItem));
    AdamItem* items = reinterpret_cast<AdamItem*>(c->host_table.data());
    long long chunks = 0;
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numels[i] <= 0) { nm_set_error("adam_step_multi: entry %d is null / empty", i); return NM_ERR_ARG; }
        items[i] = AdamItem{params[i], grads[i], exp_avg[i], exp_avg_sq[i], (long long)numels[i], chunks};
        chunks += (numels[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
    }
    int rc = nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
    if (rc) return rc;
    const size_t bytes = (size_t)count * sizeof(AdamItem);
    if ((rc = nm_ctx_reserve(c, bytes + 4096))) return rc;
    c->ws.release(0);
    AdamItem* dev = static_cast<AdamItem*>(c->ws.alloc_bytes(bytes));
    if ((rc = nm_check_hip(hipMemcpyAsync(dev, items, bytes, hipMemcpyHostToDevice, c->stream), "adam_step_multi: table upload"))) return rc;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, c->stream, dev, count, lr, beta1, beta2, eps, bc1, bc2, ok);
    return nm_check_hip(hipGetLastError(), "adam_multi launch");
}

// ---- rollouts: nm_vrnn_generate (hsvrnn_bvh.py:158-234) and nm_vrnn_rollout (the prior loop of vis_generation.py:117-127) ------
// A rollout is hundreds of dependent ~3 us launches; enqueued one by one the host (3-5 us per launch) is the bottleneck.  For the
// small batches of the latency path the whole step sequence is therefore captured ONCE per (kind, B, Tcond, Ttot, S) into a HIP
// graph (stream capture of the very same launch code on the ctx-owned side stream) and replayed with one hipGraphLaunch.  The graph
// works on buffers it owns; a call copies its inputs in and its outputs out (device-to-device, a few KB).  A graph is dropped when
// the weight buffers were re-allocated or the tree changed (epochs below); weight VALUES may change freely (same buffers).
struct RolloutBufs {
    StepBufs sb;
    float *kp_cond, *eps_post, *eps_prior, *out_cond, *out_gen, *h_in, *offset, *hbuf[2], *zbuf, *chain_g;         // chain_g: the persistent rollout kernel's granule buffers (+ its abort word)
    float *pc_z = nullptr, *pc_h = nullptr, *pc_g = nullptr;     // posterior chain of the conditioning steps: latents [B][Tcond Z], states [B][(Tcond + 1) H], granules
};
struct RolloutGraph {
    int kind, B, Tcond, Ttot, S;
    uint64_t wepoch, tepoch;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    char* base = nullptr;
    RolloutBufs rb; int cur = 0;
};
struct GraphCache { std::vector<RolloutGraph*> items; bool broken = false; };

static void free_rollout_graph(RolloutGraph* g) {
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->base) (void)hipFree(g->base);
    delete g;
}
}  // extern "C"
void nm_vrnn_free_graphs(nm_ctx* c) {
    GraphCache* gc = static_cast<GraphCache*>(c->vgraphs);
    if (!gc) return;
    for (RolloutGraph* g : gc->items) free_rollout_graph(g);
    delete gc;
    c->vgraphs = nullptr;
}
extern "C" {

static size_t rollout_floats(int B, int Tcond, int Ttot, int S, int K, int Z, int H) {
    const size_t S4 = (size_t)K * 4, Tg = Ttot - Tcond;
    return (size_t)B * (4 * 128 + 3 * H + 4 * Z) + (size_t)S * B * (Z + 256 + 3 + K + 6 * K + 9 * K + 1) + (B >= NM_GEMM_MIN_BATCH ? (size_t)B * 3 * H : 0)
         + (size_t)B * Tcond * S4 * 2 + (size_t)Tcond * S * B * Z + Tg * B * Z + (size_t)B * Tg * S4 + (size_t)B * K * 3 + 3 * (size_t)B * H + (size_t)B * Z
         + 2 * (size_t)B * (3 * 128 + 3 * H + S4 + Z + 2 * H) + 64 + 64 * 36          // (256-byte alignment of each of the ~32 pieces)
         + (size_t)B * Tcond * Z + (size_t)B * (Tcond + 1) * H + post_chain_gran_floats(B, S, K, Z, H, 0);
}
static RolloutBufs carve_rollout(Arena& ws, int B, int Tcond, int Ttot, int S, int K, int Z, int H) {
    RolloutBufs r;
    const size_t S4 = (size_t)K * 4, Tg = Ttot - Tcond;
    r.sb = alloc_step(ws, B, S, K, Z, H);
    r.kp_cond = ws.f((size_t)B * Tcond * S4 + 1); r.eps_post = ws.f((size_t)Tcond * S * B * Z + 1); r.eps_prior = ws.f(Tg * B * Z + 1);
    r.out_cond = ws.f((size_t)B * Tcond * S4 + 1); r.out_gen = ws.f((size_t)B * Tg * S4 + 1);
    r.h_in = ws.f((size_t)B * H); r.offset = ws.f((size_t)B * K * 3);
    r.hbuf[0] = ws.f((size_t)B * H); r.hbuf[1] = ws.f((size_t)B * H); r.zbuf = ws.f((size_t)B * Z);
    r.chain_g = ws.f(2 * (size_t)B * (3 * 128 + 3 * H + S4 + Z + 2 * H) + 64);
    r.pc_z = ws.f((size_t)B * Tcond * Z + 1); r.pc_h = ws.f((size_t)B * (Tcond + 1) * H); r.pc_g = ws.f(post_chain_gran_floats(B, S, K, Z, H, 0));
    return r;
}

// the step sequence itself, on c->stream.  kind 0: generate (offsets from the conditioning keypoints, h from init_kypt_rnn_state,
// Tcond posterior steps, then prior steps); kind 1: prior steps only, from the given state r.h_in and offsets r.offset.
static int rollout_steps(nm_ctx* c, RolloutBufs& r, int kind, int B, int Tcond, int Ttot, int S, int* cur_out) {
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4, Tg = Ttot - Tcond;
    int rc;
    const float* h = r.h_in;
    int nxt = 0;                        // hbuf the next step writes
    if (kind == 0) {
        if ((rc = nm_vrnn_offsets(c, r.kp_cond, B, Tcond, r.offset))) return rc;
        hipLaunchKernelGGL(broadcast_rows_kernel, dim3((H * B + 255) / 256), dim3(256), 0, c->stream, c->vrnn.h0, H, r.hbuf[0], H, B);
        h = r.hbuf[0]; nxt = 1;
    }
    // the prior steps as ONE persistent launch (vrnn_prior_chain_kernel) when the shape allows: weights register-resident for the whole
    // chain, three counter hand-offs per step instead of three launches (NM355_VRNN_CHAIN=0: the launch-per-phase steps, A/B)
    bool chain = nm_ls().vrnn_chain && nm_ls().vrnn_mid && Tg >= 1 && B <= 4 && K >= 2 && K <= 32 && Z == 128 && H == 512 && c->vrnn_cnt && c->nf_flag;
    const size_t chain_lds = (size_t)(3 + 7 * K) * 128 * sizeof(float);
    // the conditioning (posterior) steps of generate as ONE persistent launch too (vrnn_post_chain_kernel without statistics workgroups:
    // generate evaluates no prior during conditioning, hsvrnn_bvh.py:171-202): a whole HSVRNNBVH.generate is then two kernels
    int t_first = 0;
    rc = NM_OK;
    if (kind == 0 && Tcond >= 1 && r.pc_g && nm_ls().vrnn_post_chain && nm_ls().vrnn_mid && post_chain_fits(c, B, S, 0, rc)) {
        if ((rc = launch_post_chain(c, r.kp_cond, Tcond * S4, r.eps_post, r.offset, B, Tcond, S, r.out_cond, Tcond * S4, r.pc_z, Tcond * Z, nullptr, 0,
                                    r.pc_h, (Tcond + 1) * H, nullptr, 0, nullptr, nullptr, 0, r.pc_g))) return rc;
        // the state after the last conditioning step, as the dense [B][H] the prior steps take
        if ((rc = nm_check_hip(hipMemcpy2DAsync(r.hbuf[0], (size_t)H * sizeof(float), r.pc_h + (size_t)Tcond * H, (size_t)(Tcond + 1) * H * sizeof(float),
                                                (size_t)H * sizeof(float), B, hipMemcpyDeviceToDevice, c->stream), "rollout: state copy"))) return rc;
        h = r.hbuf[0]; nxt = 1; t_first = Tcond;
    }
    if (rc) return rc;
    if (chain) {
        // Co-residency: the kernel's workgroups spin on each other, so ALL of them must be resident at once - an ordinary launch does not
        // promise that.  Asked once per context: workgroups of this shape (512 threads at up to 256 registers, the heads' weights in
        // dynamic LDS) per CU x the device's CUs must cover the NM_CHAIN_NW workers + 4 middle workgroups; a partition with fewer CUs
        // (CPX mode: 32) takes the launch-per-phase steps.  A device that is busy with OTHER work when the chain starts is what the
        // bounded spins are for: the time-out is reported by the next call and switches the chain off for this context (nm_nf_poll).
        if (nm_ls().chain_fits < 0) {
            static NmDeviceOnce attr_set;
            if (!attr_set.done()) {
                if ((rc = nm_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&vrnn_prior_chain_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024), "hipFuncSetAttribute(vrnn_prior_chain)"))) return rc;
                if ((rc = nm_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&vrnn_prior_chain_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024), "hipFuncSetAttribute(vrnn_prior_chain one-XCD)"))) return rc;
                attr_set.mark();
            }
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&vrnn_prior_chain_kernel<false>), NM_CHAIN_T, (size_t)(3 + 7 * 32) * 128 * sizeof(float)) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->cfg.device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
            nm_ls().chain_fits = ((long long)per_cu * cus >= NM_CHAIN_NW + 4) ? 1 : 0;
            nm_ls().chain_cus = cus;
        }
        chain = nm_ls().chain_fits == 1;
    }
    for (int t = t_first; t < (chain ? Tcond : Ttot); ++t) {
        StepIO io;
        const bool post = t < Tcond;
        io.h = h; io.ldh = H;
        io.obs = post ? r.kp_cond + (size_t)t * S4 : nullptr; io.ldobs = Tcond * S4;
        io.eps = post ? r.eps_post + (size_t)t * S * B * Z : r.eps_prior + (size_t)(t - Tcond) * B * Z;
        io.offset = r.offset;
        io.out_kp = post ? r.out_cond + (size_t)t * S4 : r.out_gen + (size_t)(t - Tcond) * S4; io.ldkp = (post ? Tcond : Tg) * S4;
        io.out_z = r.zbuf; io.ldz = Z; io.out_R = nullptr; io.ldR = 0;
        io.best = nullptr; io.ldbest = 0; io.kl = nullptr; io.rec = nullptr; io.ldstat = 0;
        io.hout = r.hbuf[nxt]; io.ldho = H; io.want_prior = false;
        if ((rc = vrnn_step(c, r.sb, io, B, S))) return rc;
        h = io.hout; nxt ^= 1;
    }
    if (chain) {
        const VrnnW& w = c->vrnn;
        ChainArgs a;
        a.w_prior0 = w.prior0.w; a.b_prior0 = w.prior0.b; a.w_root0 = w.root0.w; a.b_root0 = w.root0.b; a.w_joint0 = w.joint0.w; a.b_joint0 = w.joint0.b;
        a.w_hh = w.w_hh; a.b_hh = w.b_hh; a.w_ih = w.w_ih; a.b_ih = w.b_ih;
        MidArgs& m = a.mid;
        m.hid_prior = m.rh = m.jh = m.eps = nullptr; m.offset = r.offset;
        m.w_p2 = w.prior2.w; m.b_p2 = w.prior2.b; m.w_root0 = w.root0.w; m.w_joint0 = w.joint0.w;
        m.w_root2 = w.root2.w; m.b_root2 = w.root2.b; m.w_joint2 = w.joint2.w; m.b_joint2 = w.joint2.b;
        m.order = w.order; m.parents = w.parents; m.lvl_joint = w.lvl_joint; m.lvl_start = w.lvl_start; m.nlevels = w.nlevels;
        m.out_kp = nullptr; m.ldkp = 0; m.out_z = nullptr; m.ldz = 0; m.B = B; m.K = K; m.Z = Z; m.H = H;
        a.h0 = h; a.ldh0 = H; a.eps = r.eps_prior; a.out_kp = r.out_gen; a.ldkp = Tg * S4;
        a.h_out = r.hbuf[nxt];
        // granule buffers: [hid | rh | jh : B x 128 each][gh : B x 3H][kpz : B x (4K + Z)][h0 | h1 : B x H each][abort word]
        nm_gran* gb = reinterpret_cast<nm_gran*>(r.chain_g);
        a.g_hid = gb; a.g_rh = gb + (size_t)B * 128; a.g_jh = gb + (size_t)2 * B * 128; a.g_gh = gb + (size_t)3 * B * 128;
        a.g_kpz = a.g_gh + (size_t)B * 3 * H; a.g_h[0] = a.g_kpz + (size_t)B * (S4 + Z); a.g_h[1] = a.g_h[0] + (size_t)B * H;
        a.abort = reinterpret_cast<unsigned*>(a.g_h[1] + (size_t)B * H);
        a.ctl = nullptr; a.stamps = g_chain_stamps;
        a.status = c->nf_flag;
        a.B = B; a.T = Tg; a.K = K; a.Z = Z; a.H = H;
        // poll back-off in s_sleep(2) units (0 / 1 / 4 / 16 / 64: 17.0 / 18.0 / 17.3 / 19.5 / 25.9 us per step at B = 1)
        { static const int bo = getenv("NM355_CHAIN_BACKOFF") ? atoi(getenv("NM355_CHAIN_BACKOFF")) : 0; a.backoff = bo; }
        a.spin_limit = nm_ls().chain_spin > 0 ? nm_ls().chain_spin : NM_CHAIN_SPIN;
        a.wg_poll = nm_ls().chain_wgpoll; a.force_nogo = nm_ls().chain_xcd_nogo;
        const size_t gbytes = (size_t)B * (3 * 128 + 3 * H + S4 + Z + 2 * H) * sizeof(nm_gran) + 64;
        if ((rc = nm_check_hip(hipMemsetAsync(r.chain_g, 0, gbytes, c->stream), "rollout: granule buffers"))) return rc;
        // (NM355_CHAIN_DROP_WG, test hook: the last workgroups are not launched - what a workgroup that never becomes resident looks like
        //  to the others; their spins run into the limit, the abort word ends the kernel, bit 1 of the status word reports it)
        const int drop = std::min(std::max(nm_ls().chain_drop, 0), B);
        // One-XCD form first (NM355_CHAIN_XCD, default on; B <= 8, H = 512): one workgroup per CU of the device, of which the NM_CHAIN_NWX + B
        // that share the first arriver's XCD run the rollout with plain-store / L2-served hand-offs; if that XCD cannot seat them all
        // (other work on its CUs) nobody starts and the cross-XCD launch behind it - which otherwise returns at its first instruction -
        // does the work.  The two launches share inputs, outputs and the (zeroed) granule buffers.
        // (B = 1 only by default: at B = 3 the one-XCD kernel itself is faster too - 767 against 818 us per 64 steps - but the call is not,
        //  1.08 against 1.02 ms: profiles/r06_rollout_ab.txt; NM355_CHAIN_XCD=2 forces it for B <= 8)
        if (nm_ls().chain_xcd && (B == 1 || (nm_ls().chain_xcd >= 2 && B <= 8)) && H == 512 && nm_ls().chain_cus >= 64 && drop == 0) {
            a.ctl = reinterpret_cast<int*>(a.abort) + 4;
            hipLaunchKernelGGL(vrnn_prior_chain_kernel<true>, dim3(nm_ls().chain_cus), dim3(NM_CHAIN_T), chain_lds, c->stream, a);
            if ((rc = nm_check_hip(hipGetLastError(), "vrnn_prior_chain (one XCD) launch"))) return rc;
        }
        hipLaunchKernelGGL(vrnn_prior_chain_kernel<false>, dim3(NM_CHAIN_NW + B - drop), dim3(NM_CHAIN_T), chain_lds, c->stream, a);
        if ((rc = nm_check_hip(hipGetLastError(), "vrnn_prior_chain launch"))) return rc;
        *cur_out = nxt;                 // the last step's state, as plain floats
        return NM_OK;
    }
    *cur_out = nxt ^ 1;                 // the hbuf holding the last state
    return nm_check_hip(hipGetLastError(), "rollout launches");
}

static int copy_dd(float* dst, const float* src, size_t n, hipStream_t s, const char* what) {
    if (!n || !dst || !src) return NM_OK;
    return nm_check_hip(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s), what);
}

// kind 0 / 1 as above; user pointers may be null where the kind does not use them
static int rollout_impl(nm_ctx* c, int kind, const float* kp_cond, const float* eps_post, const float* eps_prior, const float* h_in,
                        const float* offset_in, int B, int Tcond, int Ttot, int S, float* out_cond, float* out_gen, float* h_last) {
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4, Tg = Ttot - Tcond;
    int rc;
    const size_t n_kc = (size_t)B * Tcond * S4, n_ep = (size_t)Tcond * S * B * Z, n_er = (size_t)Tg * B * Z, n_og = (size_t)B * Tg * S4;
    GraphCache* gc = static_cast<GraphCache*>(c->vgraphs);
    const bool want_graph = nm_ls().vrnn_graph && B <= 64 && !(gc && gc->broken);
    if (want_graph) {
        if (!gc) { gc = new GraphCache(); c->vgraphs = gc; }
        RolloutGraph* g = nullptr;
        for (size_t i = 0; i < gc->items.size(); ++i) {
            RolloutGraph* it = gc->items[i];
            if (it->wepoch != c->weights_epoch || it->tepoch != c->vrnn.tree_epoch) {       // stale: pointers / level count baked in
                (void)hipStreamSynchronize(c->stream);
                free_rollout_graph(it); gc->items.erase(gc->items.begin() + i); --i; continue;
            }
            if (it->kind == kind && it->B == B && it->Tcond == Tcond && it->Ttot == Ttot && it->S == S) g = it;
        }
        if (!g) {
            if (gc->items.size() >= 16) {                      // bounded cache: drop the oldest entry
                (void)hipStreamSynchronize(c->stream);
                free_rollout_graph(gc->items.front()); gc->items.erase(gc->items.begin());
            }
            g = new RolloutGraph();
            g->kind = kind; g->B = B; g->Tcond = Tcond; g->Ttot = Ttot; g->S = S; g->wepoch = c->weights_epoch; g->tepoch = c->vrnn.tree_epoch;
            const size_t bytes = rollout_floats(B, Tcond, Ttot, S, K, Z, H) * sizeof(float);
            bool ok = hipMalloc(reinterpret_cast<void**>(&g->base), bytes) == hipSuccess;
            if (ok) {
                Arena ar; ar.base = g->base; ar.cap = bytes;
                g->rb = carve_rollout(ar, B, Tcond, Ttot, S, K, Z, H);
                ok = !ar.overflow;
            }
            if (ok) {
                // capture the launch sequence on the side stream (a capture cannot run on the legacy default stream, which the caller's
                // stream may be); relaxed mode: other threads of the process (torch's allocator) stay free to call the runtime
                // (a stream of its own for the capture: the side streams are shared by the contexts of a process, nm_api.hip)
                hipStream_t main = c->stream, cap = nullptr;
                ok = hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) == hipSuccess && hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed) == hipSuccess;
                if (ok) {
                    c->stream = cap;
                    const int r2 = rollout_steps(c, g->rb, kind, B, Tcond, Ttot, S, &g->cur);
                    c->stream = main;
                    ok = hipStreamEndCapture(cap, &g->graph) == hipSuccess && r2 == NM_OK && g->graph;
                }
                if (cap) (void)hipStreamDestroy(cap);
                if (ok) ok = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) == hipSuccess;
            }
            if (!ok) {                                          // no graph on this runtime: the eager launch sequence below is the same work
                (void)hipGetLastError();
                free_rollout_graph(g); g = nullptr; gc->broken = true;
            } else gc->items.push_back(g);
        }
        if (g) {
            hipStream_t s = c->stream;
            RolloutBufs& r = g->rb;
            if ((rc = copy_dd(r.kp_cond, kp_cond, n_kc, s, "rollout: in"))) return rc;
            if ((rc = copy_dd(r.eps_post, eps_post, n_ep, s, "rollout: in"))) return rc;
            if ((rc = copy_dd(r.eps_prior, eps_prior, n_er, s, "rollout: in"))) return rc;
            if (kind == 1) {
                if ((rc = copy_dd(r.h_in, h_in, (size_t)B * H, s, "rollout: in"))) return rc;
                if ((rc = copy_dd(r.offset, offset_in, (size_t)B * K * 3, s, "rollout: in"))) return rc;
            }
            if ((rc = nm_check_hip(hipGraphLaunch(g->exec, s), "rollout: hipGraphLaunch"))) return rc;
            if ((rc = copy_dd(out_cond, r.out_cond, n_kc, s, "rollout: out"))) return rc;
            if ((rc = copy_dd(out_gen, r.out_gen, n_og, s, "rollout: out"))) return rc;
            return copy_dd(h_last, Ttot > 0 ? r.hbuf[g->cur] : r.h_in, (size_t)B * H, s, "rollout: h_last");
        }
    }
    // eager: the same launch sequence straight onto the ctx stream, working on the caller's buffers
    if ((rc = nm_ctx_reserve(c, (rollout_floats(B, 0, 0, S, K, Z, H) + (size_t)B * Tcond * Z + (size_t)B * (Tcond + 1) * H + 1024) * sizeof(float) + 4096))) return rc;
    c->ws.release(0);
    RolloutBufs r;
    r.sb = alloc_step(c->ws, B, S, K, Z, H);
    r.hbuf[0] = c->ws.f((size_t)B * H); r.hbuf[1] = c->ws.f((size_t)B * H); r.zbuf = c->ws.f((size_t)B * Z);
    r.chain_g = c->ws.f(2 * (size_t)B * (3 * 128 + 3 * H + K * 4 + Z + 2 * H) + 64);
    if (kind == 0) { r.pc_z = c->ws.f((size_t)B * Tcond * Z + 1); r.pc_h = c->ws.f((size_t)B * (Tcond + 1) * H); r.pc_g = c->ws.f(post_chain_gran_floats(B, S, K, Z, H, 0)); }
    r.offset = kind == 0 ? c->ws.f((size_t)B * K * 3) : const_cast<float*>(offset_in);
    if (c->ws.overflow) { nm_set_error("vrnn rollout: workspace overflow"); return NM_ERR_STATE; }
    r.kp_cond = const_cast<float*>(kp_cond); r.eps_post = const_cast<float*>(eps_post); r.eps_prior = const_cast<float*>(eps_prior);
    r.out_cond = out_cond; r.out_gen = out_gen; r.h_in = const_cast<float*>(h_in);
    int cur = 0;
    if ((rc = rollout_steps(c, r, kind, B, Tcond, Ttot, S, &cur))) return rc;
    return copy_dd(h_last, Ttot > 0 ? r.hbuf[cur] : h_in, (size_t)B * H, c->stream, "rollout: h_last");
}

int nm_vrnn_generate(nm_ctx* c, const float* keypoints_cond, const float* eps_post, const float* eps_prior, int32_t B,
                     int32_t Tcond, int32_t Ttot, int32_t S, float* out_cond, float* out_gen, float* h_last) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_generate", true);
    if (rc) return rc;
    if (!keypoints_cond || !eps_post || !out_cond || B <= 0 || Tcond <= 0 || Ttot < Tcond || S <= 0 || (Ttot > Tcond && (!eps_prior || !out_gen))) {
        nm_set_error("vrnn_generate: bad argument"); return NM_ERR_ARG;
    }
    if ((rc = max_fk_lds(c, S))) return rc;
    rc = rollout_impl(c, 0, keypoints_cond, eps_post, eps_prior, nullptr, nullptr, B, Tcond, Ttot, S, out_cond, out_gen, h_last);
    if (!rc) nm_nf_post(c, "nm_vrnn_generate");
    return rc;
} catch (...) { return nm_abi_catch("nm_vrnn_generate"); }

int nm_vrnn_rollout(nm_ctx* c, const float* h_in, const float* offset, const float* eps, int32_t B, int32_t T, float* kp_out, float* h_out) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_rollout", true);
    if (rc) return rc;
    if (!h_in || !offset || !eps || !kp_out || B <= 0 || T <= 0) { nm_set_error("vrnn_rollout: bad argument"); return NM_ERR_ARG; }
    if ((rc = max_fk_lds(c, 1))) return rc;
    rc = rollout_impl(c, 1, nullptr, nullptr, eps, h_in, offset, B, 0, T, 1, nullptr, kp_out, h_out);
    if (!rc) nm_nf_post(c, "nm_vrnn_rollout");
    return rc;
} catch (...) { return nm_abi_catch("nm_vrnn_rollout"); }

int nm_vrnn_step(nm_ctx* c, int32_t posterior, const float* h_in, const float* kp_obs, const float* offset, const float* eps,
                 int32_t B, int32_t S, float* kp_out, float* z_out, float* h_out) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_step", true);
    if (rc) return rc;
    if (!h_in || !offset || !eps || !kp_out || !z_out || B <= 0 || (posterior && (!kp_obs || S <= 0))) {
        nm_set_error("vrnn_step: bad argument"); return NM_ERR_ARG;
    }
    if (!posterior) S = 1;
    if ((rc = max_fk_lds(c, S))) return rc;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4;
    size_t need = ((size_t)B * (4 * 128 + 6 * H + 4 * Z) + (size_t)S * B * (Z + 256 + 3 + K + 6 * K + 9 * K + 1)) * sizeof(float) + 64 * 256;
    if ((rc = nm_ctx_reserve(c, need))) return rc;
    c->ws.release(0);
    StepBufs sb = alloc_step(c->ws, B, S, K, Z, H);
    if (c->ws.overflow) { nm_set_error("vrnn_step: workspace overflow"); return NM_ERR_STATE; }
    StepIO io;
    io.h = h_in; io.ldh = H; io.obs = posterior ? kp_obs : nullptr; io.ldobs = S4; io.eps = eps; io.offset = offset;
    io.out_kp = kp_out; io.ldkp = S4; io.out_z = z_out; io.ldz = Z; io.out_R = nullptr; io.ldR = 0;
    io.best = nullptr; io.ldbest = 0; io.kl = nullptr; io.rec = nullptr; io.ldstat = 0;
    io.hout = h_out; io.ldho = H; io.want_prior = false;
    return vrnn_step(c, sb, io, B, S);
} catch (...) { return nm_abi_catch("nm_vrnn_step"); }

// idx = argmin_r sum_d (rows[r][d] - target[r * tstride + d])^2  (first minimum), one block
__global__ __launch_bounds__(256) void rows_argmin_kernel(const float* __restrict__ rows, const float* __restrict__ target, int tstride,
                                                          int B, int D, int32_t* __restrict__ idx, float* __restrict__ dist) {
    __shared__ float bd[256]; __shared__ int bi[256];
    float best = INFINITY; int besti = 0x7fffffff;
    for (int r = threadIdx.x; r < B; r += 256) {
        float d = 0.f;
        for (int k = 0; k < D; ++k) { float u = rows[(size_t)r * D + k] - target[(size_t)r * tstride + k]; d += u * u; }
        if (d < best) { best = d; besti = r; }
    }
    bd[threadIdx.x] = best; bi[threadIdx.x] = besti;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            float o = bd[threadIdx.x + st]; int oi = bi[threadIdx.x + st];
            if (o < bd[threadIdx.x] || (o == bd[threadIdx.x] && oi < bi[threadIdx.x])) { bd[threadIdx.x] = o; bi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { idx[0] = bi[0]; if (dist) dist[0] = bd[0]; }
}

int nm_rows_argmin_dist(nm_ctx* c, const float* rows, const float* target, int32_t target_row_stride, int32_t B, int32_t D,
                        int32_t* idx_out, float* dist_out) try { NmScope nm_scope_(c);
    if (!c || !rows || !target || !idx_out || B <= 0 || D <= 0) { nm_set_error("rows_argmin_dist: bad argument"); return NM_ERR_ARG; }
    hipLaunchKernelGGL(rows_argmin_kernel, dim3(1), dim3(256), 0, c->stream, rows, target, target_row_stride, B, D, idx_out, dist_out);
    return nm_check_hip(hipGetLastError(), "rows_argmin launch");
} catch (...) { return nm_abi_catch("nm_rows_argmin_dist"); }

int nm_vrnn_mlp(nm_ctx* c, int32_t which, const float* x, int32_t B, float* y) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_mlp", false);
    if (rc) return rc;
    if (!x || !y || B <= 0 || which < 0 || which > 3) { nm_set_error("vrnn_mlp: bad argument"); return NM_ERR_ARG; }
    const VrnnW& w = c->vrnn;
    const LinearW* l0[4] = {&w.post0, &w.prior0, &w.root0, &w.joint0};
    const LinearW* l2[4] = {&w.post2, &w.prior2, &w.root2, &w.joint2};
    if ((rc = nm_ctx_reserve(c, (size_t)B * 128 * sizeof(float) + 4096))) return rc;
    c->ws.release(0);
    float* hid = c->ws.f((size_t)B * 128);
    LinJobs J; J.n = 0; J.start[0] = 0;
    add_job(J, *l0[which], 0, x, l0[which]->in, l0[which]->in, nullptr, 0, 0, true, nullptr, 0, 1, hid, 128, 1, B);
    if ((rc = launch_jobs(J, c->stream))) return rc;
    LinJobs J2; J2.n = 0; J2.start[0] = 0;
    add_job(J2, *l2[which], 0, hid, 128, 128, nullptr, 0, 0, true, nullptr, 0, 1, y, l2[which]->out, which == 2 ? 2 : 0, B);
    return launch_jobs(J2, c->stream);
} catch (...) { return nm_abi_catch("nm_vrnn_mlp"); }

int nm_vrnn_gru(nm_ctx* c, const float* x, const float* h, int32_t B, float* h_out) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_gru", false);
    if (rc) return rc;
    if (!x || !h || !h_out || B <= 0) { nm_set_error("vrnn_gru: bad argument"); return NM_ERR_ARG; }
    const VrnnW& w = c->vrnn;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, in = K * 4 + Z;
    if ((rc = nm_ctx_reserve(c, (size_t)B * 3 * H * sizeof(float) + 4096))) return rc;
    c->ws.release(0);
    float* gh = c->ws.f((size_t)B * 3 * H);
    LinJobs J; J.n = 0; J.start[0] = 0;
    LinearW hh; hh.in = H; hh.out = 3 * H; hh.w = w.w_hh; hh.b = w.b_hh;
    add_job(J, hh, 0, h, H, H, nullptr, 0, 0, true, nullptr, 0, 1, gh, 3 * H, 0, B);
    if ((rc = launch_jobs(J, c->stream))) return rc;
    return launch_gru(w.w_ih, w.b_ih, x, in, in, nullptr, 0, 0, gh, h, H, h_out, H, H, B, c->stream);
} catch (...) { return nm_abi_catch("nm_vrnn_gru"); }

int nm_vrnn_fk(nm_ctx* c, const float* dec_in, const float* offset, int32_t B, float* kp, float* R) try { NmScope nm_scope_(c);
    int rc = ready(c, "vrnn_fk", true);
    if (rc) return rc;
    if (!dec_in || !offset || !kp || !R || B <= 0) { nm_set_error("vrnn_fk: bad argument"); return NM_ERR_ARG; }
    const VrnnW& w = c->vrnn;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, in = H + Z;
    if ((rc = nm_ctx_reserve(c, (size_t)B * (256 + 3 + K + 6 * K + Z) * sizeof(float) + 8192))) return rc;
    c->ws.release(0);
    float* hr = c->ws.f((size_t)B * 128); float* hj = c->ws.f((size_t)B * 128);
    float* rootout = c->ws.f((size_t)B * (3 + K)); float* rot = c->ws.f((size_t)B * 6 * K);
    LinJobs J; J.n = 0; J.start[0] = 0;
    add_job(J, w.root0, 0, dec_in, in, in, nullptr, 0, 0, true, nullptr, 0, 1, hr, 128, 1, B);
    add_job(J, w.joint0, 0, dec_in, in, in, nullptr, 0, 0, true, nullptr, 0, 1, hj, 128, 1, B);
    if ((rc = launch_jobs(J, c->stream))) return rc;
    LinJobs J2; J2.n = 0; J2.start[0] = 0;
    add_job(J2, w.root2, 0, hr, 128, 128, nullptr, 0, 0, true, nullptr, 0, 1, rootout, 3 + K, 2, B);
    add_job(J2, w.joint2, 0, hj, 128, 128, nullptr, 0, 0, true, nullptr, 0, 1, rot, 6 * K, 0, B);
    if ((rc = launch_jobs(J2, c->stream))) return rc;
    FkArgs a;
    a.root = rootout; a.ldr = 3 + K; a.rot = rot; a.offset = offset; a.obs = nullptr; a.ldobs = 0; a.z = nullptr;
    a.order = w.order; a.parents = w.parents; a.lvl_joint = w.lvl_joint; a.lvl_start = w.lvl_start; a.nlevels = w.nlevels; a.qmu = a.qsig = a.pmu = a.psig = nullptr;
    a.out_kp = kp; a.ldkp = K * 4; a.out_z = nullptr; a.ldz = 0; a.out_R = R; a.ldR = K * 9;
    a.best = nullptr; a.ldbest = 0; a.kl = nullptr; a.rec = nullptr; a.ldstat = 0; a.K = K; a.S = 1; a.B = B; a.Z = Z;
    a.hr = a.hj = a.eps = nullptr; a.t_hr = a.t_hj = a.t_raw = a.t_rot6 = a.t_Rl = a.t_Rg = a.t_eps = nullptr;
    size_t lds = ((size_t)K * 22 + 1) * sizeof(float);
    hipLaunchKernelGGL(fk_kernel, dim3(B), dim3(256), lds, c->stream, a);
    return nm_check_hip(hipGetLastError(), "fk launch");
} catch (...) { return nm_abi_catch("nm_vrnn_fk"); }

}  // extern "C"
