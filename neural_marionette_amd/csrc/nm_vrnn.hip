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
#pragma unroll 2
        for (int k = lane * 4; k < n; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                int b = b0 + s; b = b < batch ? b : batch - 1;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * ld + k);
                acc[s] += ((wv[0] * xv[0] + wv[1] * xv[1]) + wv[2] * xv[2]) + wv[3] * xv[3];
            }
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
        for (int s = 0; s < NB; ++s) acc[s] += __shfl_xor(acc[s], off);
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
    dot_seg(wr, J.xa, J.na, J.lda, b0, J.batch, lane, acc);
    dot_seg(wr + J.na, J.xb, J.nb, J.ldb, b0, J.batch, lane, acc);
    wave_reduce(acc);
    if (lane < NB && b0 + lane < J.batch) {
        const int b = b0 + lane;
        float v = pick(acc, lane);
        if (J.bias) v += J.bias[r];
        if (J.add) v += J.add[(size_t)(b % J.add_mod) * J.ldadd + r];
        if (J.act == 1) v = lrelu(v, 0.01f);
        else if (J.act == 2) v = tanhf(v);
        J.out[(size_t)b * J.ldo + r] = v;
    }
}

// second layer of a prior / posterior MLP: rows (r, r+Z) -> mu, std = softplus(.)+1e-4 and, when eps is given,
// z[i][b][r] = mu + eps[i][b][r] * std   (hsvrnn_bvh.py:93-107)
struct DistJob { const float* W; const float* bias; const float* x; float* mu; float* sig; const float* eps; float* z; int S; };
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
    dot_seg(J.W + (size_t)r * hid, J.x, hid, hid, b0, B, lane, am);
    dot_seg(J.W + (size_t)(r + Z) * hid, J.x, hid, hid, b0, B, lane, as);
    wave_reduce(am); wave_reduce(as);
    if (lane < NB && b0 + lane < B) {
        const int bb = b0 + lane;
        const float mu = pick(am, lane) + J.bias[r];
        const float sg = softplus(pick(as, lane) + J.bias[r + Z]) + 1e-4f;
        J.mu[(size_t)bb * Z + r] = mu; J.sig[(size_t)bb * Z + r] = sg;
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
                                                       float* __restrict__ hout, int ldo, int H, int B) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= H) return;
    const int b0 = blockIdx.y * NB;
    const int in = na + nb;
    float ar[NB], az[NB], an[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) { ar[s] = 0.f; az[s] = 0.f; an[s] = 0.f; }
    const float* w0 = W_ih + (size_t)j * in; const float* w1 = W_ih + (size_t)(H + j) * in; const float* w2 = W_ih + (size_t)(2 * H + j) * in;
    dot_seg(w0, xa, na, lda, b0, B, lane, ar); dot_seg(w0 + na, xb, nb, ldb, b0, B, lane, ar);
    dot_seg(w1, xa, na, lda, b0, B, lane, az); dot_seg(w1 + na, xb, nb, ldb, b0, B, lane, az);
    dot_seg(w2, xa, na, lda, b0, B, lane, an); dot_seg(w2 + na, xb, nb, ldb, b0, B, lane, an);
    wave_reduce(ar); wave_reduce(az); wave_reduce(an);
    if (lane < NB && b0 + lane < B) {
        const int b = b0 + lane;
        const float* g = gh + (size_t)b * 3 * H;
        const float rg = sigmoidf((pick(ar, lane) + b_ih[j]) + g[j]);
        const float zg = sigmoidf((pick(az, lane) + b_ih[H + j]) + g[H + j]);
        const float ng = tanhf((pick(an, lane) + b_ih[2 * H + j]) + rg * g[2 * H + j]);
        const float hp = h[(size_t)b * ldh + j];
        hout[(size_t)b * ldo + j] = (hp - ng) * zg + ng;
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
    const float *qmu, *qsig, *pmu, *psig;   // posterior / prior params [B][Z] (KL) or null
    float* out_kp; int ldkp;           // best keypoints (K*4)
    float* out_z; int ldz;
    float* out_R; int ldR;             // best global rotations (K*9) or null
    int32_t* best; int ldbest;         // or null
    float* kl; float* rec; int ldstat; // per-sample sums (or null)
    int K, S, B, Z;
};

__global__ __launch_bounds__(256) void fk_kernel(FkArgs a) {
    extern __shared__ float sm[];
    const int K = a.K, S = a.S, b = blockIdx.x;
    float* Rl = sm;                    // [S][K][9]
    float* Rg = Rl + S * K * 9;        // [S][K][9]
    float* pos = Rg + S * K * 9;       // [S][K][3]
    float* dist = pos + S * K * 3;     // [S]
    __shared__ float red[256];
    __shared__ int best_s;
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
    // chain along the tree in priority order (geo_utils.py:16-25, hsvrnn_bvh.py:272-277): one thread per sample
    if ((int)threadIdx.x < S) {
        const int i = threadIdx.x;
        const float* rt = a.root + ((size_t)(i * a.B + b)) * a.ldr;
        const int root = a.order[0];
        for (int e = 0; e < 9; ++e) Rg[(i * K + root) * 9 + e] = Rl[(i * K + root) * 9 + e];
        pos[(i * K + root) * 3 + 0] = rt[0]; pos[(i * K + root) * 3 + 1] = rt[1]; pos[(i * K + root) * 3 + 2] = rt[2];
        for (int o = 1; o < K; ++o) {
            const int idx = a.order[o], par = a.parents[idx];
            const float* P = Rg + (i * K + par) * 9; const float* L = Rl + (i * K + idx) * 9;
            float* G = Rg + (i * K + idx) * 9;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) G[r * 3 + c] = (P[r * 3] * L[c] + P[r * 3 + 1] * L[3 + c]) + P[r * 3 + 2] * L[6 + c];
        }
        for (int o = 1; o < K; ++o) {
            const int idx = a.order[o], par = a.parents[idx];
            const float* G = Rg + (i * K + idx) * 9; const float* of = a.offset + ((size_t)b * K + idx) * 3;
            for (int r = 0; r < 3; ++r)
                pos[(i * K + idx) * 3 + r] = ((G[r * 3] * of[0] + G[r * 3 + 1] * of[1]) + G[r * 3 + 2] * of[2]) + pos[(i * K + par) * 3 + r];
        }
        float d = 0.f;
        if (a.obs) {
            const float* ob = a.obs + (size_t)b * a.ldobs;
            for (int k = 0; k < K; ++k) {
                for (int c = 0; c < 3; ++c) { float u = ob[k * 4 + c] - pos[(i * K + k) * 3 + c]; d += u * u; }
                float u = ob[k * 4 + 3] - (rt[3 + k] + 1.0f) * 0.5f; d += u * u;
            }
        }
        dist[i] = d;
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
struct StepBufs {
    float *hid_prior, *hid_post, *rh, *jh, *gh, *pmu, *psig, *qmu, *qsig, *z, *hr, *hj, *rootout, *rot;
};

void add_job(LinJobs& J, const LinearW& L, int col0, const float* xa, int na, int lda, const float* xb, int nb, int ldb,
             bool bias, const float* add, int ldadd, int add_mod, float* out, int ldo, int act, int batch) {
    LinJob& j = J.j[J.n];
    j.W = L.w; j.ldw = L.in; j.col0 = col0; j.xa = xa; j.na = na; j.lda = lda; j.xb = xb; j.nb = nb; j.ldb = ldb;
    j.bias = bias ? L.b : nullptr; j.add = add; j.ldadd = ldadd; j.add_mod = add_mod > 0 ? add_mod : 1;
    j.out = out; j.ldo = ldo; j.rows = L.out; j.act = act; j.batch = batch;
    J.start[J.n + 1] = J.start[J.n] + L.out;
    J.n++;
}

int pick_nb(int batch) { return batch >= 8 ? 8 : (batch >= 4 ? 4 : (batch >= 2 ? 2 : 1)); }

int launch_jobs(const LinJobs& J, hipStream_t s) {
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
               const float* gh, const float* h, int ldh, float* hout, int ldo, int H, int B, hipStream_t s) {
    const int nb = pick_nb(B);
    dim3 grid((H + 3) / 4, (B + nb - 1) / nb);
    if (nb == 1) hipLaunchKernelGGL((gru_rows_kernel<1>), grid, dim3(256), 0, s, W_ih, b_ih, xa, na, lda, xb, nb_, ldb, gh, h, ldh, hout, ldo, H, B);
    else if (nb == 2) hipLaunchKernelGGL((gru_rows_kernel<2>), grid, dim3(256), 0, s, W_ih, b_ih, xa, na, lda, xb, nb_, ldb, gh, h, ldh, hout, ldo, H, B);
    else if (nb == 4) hipLaunchKernelGGL((gru_rows_kernel<4>), grid, dim3(256), 0, s, W_ih, b_ih, xa, na, lda, xb, nb_, ldb, gh, h, ldh, hout, ldo, H, B);
    else hipLaunchKernelGGL((gru_rows_kernel<8>), grid, dim3(256), 0, s, W_ih, b_ih, xa, na, lda, xb, nb_, ldb, gh, h, ldh, hout, ldo, H, B);
    return nm_check_hip(hipGetLastError(), "gru launch");
}

StepBufs alloc_step(Arena& ws, int B, int S, int K, int Z, int H) {
    StepBufs b;
    b.hid_prior = ws.f((size_t)B * 128); b.hid_post = ws.f((size_t)B * 128); b.rh = ws.f((size_t)B * 128); b.jh = ws.f((size_t)B * 128);
    b.gh = ws.f((size_t)B * 3 * H);
    b.pmu = ws.f((size_t)B * Z); b.psig = ws.f((size_t)B * Z); b.qmu = ws.f((size_t)B * Z); b.qsig = ws.f((size_t)B * Z);
    b.z = ws.f((size_t)S * B * Z); b.hr = ws.f((size_t)S * B * 128); b.hj = ws.f((size_t)S * B * 128);
    b.rootout = ws.f((size_t)S * B * (3 + K)); b.rot = ws.f((size_t)S * B * 6 * K);
    return b;
}

struct StepIO {
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
    {   // 2. distribution parameters + samples
        DistJob jp{w.prior2.w, w.prior2.b, sb.hid_prior, sb.pmu, sb.psig, post ? nullptr : io.eps, sb.z, 1};
        DistJob jq{w.post2.w, w.post2.b, sb.hid_post, sb.qmu, sb.qsig, io.eps, sb.z, S};
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
        a.order = w.order; a.parents = w.parents;
        const bool kl = post && io.want_prior && io.kl;
        a.qmu = kl ? sb.qmu : nullptr; a.qsig = sb.qsig; a.pmu = sb.pmu; a.psig = sb.psig;
        a.out_kp = io.out_kp; a.ldkp = io.ldkp; a.out_z = io.out_z; a.ldz = io.ldz; a.out_R = io.out_R; a.ldR = io.ldR;
        a.best = io.best; a.ldbest = io.ldbest; a.kl = kl ? io.kl : nullptr; a.rec = io.rec; a.ldstat = io.ldstat;
        a.K = K; a.S = S; a.B = B; a.Z = Z;
        size_t lds = ((size_t)S * K * 21 + S) * sizeof(float);
        hipLaunchKernelGGL(fk_kernel, dim3(B), dim3(256), lds, s, a);
        if ((rc = nm_check_hip(hipGetLastError(), "fk launch"))) return rc;
    }
    if (io.hout) {   // 5. GRU
        if ((rc = launch_gru(w.w_ih, w.b_ih, io.out_kp, S4, io.ldkp, io.out_z, Z, io.ldz, sb.gh, io.h, io.ldh, io.hout, io.ldho, H, B, s))) return rc;
    }
    return NM_OK;
}

int ready(nm_ctx* c, const char* who, bool need_tree) {
    if (!c) { nm_set_error("%s: null ctx", who); return NM_ERR_ARG; }
    if (!c->has_weights) { nm_set_error("%s: nm_ctx_set_weights has not been called", who); return NM_ERR_STATE; }
    if (need_tree && !c->vrnn.has_tree) { nm_set_error("%s: nm_vrnn_set_tree has not been called (the reference builds it in encode())", who); return NM_ERR_STATE; }
    return nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
}

int max_fk_lds(nm_ctx* c, int S) {
    size_t lds = ((size_t)S * c->cfg.nkeypoints * 21 + S) * sizeof(float);
    if (lds > 60 * 1024) { nm_set_error("vrnn: S=%d samples exceed the FK kernel's LDS budget", S); return NM_ERR_UNSUPPORTED; }
    return NM_OK;
}

}  // namespace

extern "C" {

int nm_vrnn_set_tree(nm_ctx* c, const int32_t* parents, const int32_t* order) {
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
    rc = nm_check_hip(hipMemcpyAsync(c->vrnn.parents, c->vrnn.parents_h.data(), K * sizeof(int32_t), hipMemcpyHostToDevice, c->stream), "set_tree copy");
    if (rc) return rc;
    rc = nm_check_hip(hipMemcpyAsync(c->vrnn.order, c->vrnn.order_h.data(), K * sizeof(int32_t), hipMemcpyHostToDevice, c->stream), "set_tree copy");
    if (rc) return rc;
    rc = nm_check_hip(hipStreamSynchronize(c->stream), "set_tree sync");
    if (rc) return rc;
    c->vrnn.has_tree = true;
    return NM_OK;
}

int nm_vrnn_offsets(nm_ctx* c, const float* keypoints, int32_t B, int32_t T, float* offset) {
    int rc = ready(c, "vrnn_offsets", true);
    if (rc) return rc;
    if (!keypoints || !offset || B <= 0 || T <= 0) { nm_set_error("vrnn_offsets: bad argument"); return NM_ERR_ARG; }
    const int K = c->cfg.nkeypoints;
    if ((size_t)K * T * sizeof(float) > 60 * 1024) { nm_set_error("vrnn_offsets: T too large"); return NM_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(offsets_kernel, dim3(B), dim3(64), (size_t)K * T * sizeof(float), c->stream, keypoints, c->vrnn.offset_param,
                       c->vrnn.parents, T, K, offset);
    return nm_check_hip(hipGetLastError(), "offsets launch");
}

int nm_vrnn_encode(nm_ctx* c, const float* keypoints, const float* eps, int32_t B, int32_t T, int32_t S, float* kypt_recon,
                   float* R, float* z, float* h, float* scalars2, int32_t* best_idx) {
    int rc = ready(c, "vrnn_encode", true);
    if (rc) return rc;
    if (!keypoints || !eps || !kypt_recon || !R || !z || !h || !scalars2 || B <= 0 || T <= 0 || S <= 0) {
        nm_set_error("vrnn_encode: null / non-positive argument (eps must be supplied explicitly)"); return NM_ERR_ARG;
    }
    if ((rc = max_fk_lds(c, S))) return rc;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4;
    size_t need = ((size_t)B * (4 * 128 + 3 * H + 4 * Z + K * 3 + 2 * T) + (size_t)S * B * (Z + 256 + 3 + K + 6 * K)) * sizeof(float) + 64 * 256;
    if ((rc = nm_ctx_reserve(c, need))) return rc;
    c->ws.release(0);
    StepBufs sb = alloc_step(c->ws, B, S, K, Z, H);
    float* offset = c->ws.f((size_t)B * K * 3);
    float* kl = c->ws.f((size_t)B * T); float* rec = c->ws.f((size_t)B * T);
    if (c->ws.overflow) { nm_set_error("vrnn_encode: workspace overflow"); return NM_ERR_STATE; }
    if ((rc = nm_vrnn_offsets(c, keypoints, B, T, offset))) return rc;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3((H * B + 255) / 256), dim3(256), 0, c->stream, c->vrnn.h0, H, h, (T + 1) * H, B);
    for (int t = 0; t < T; ++t) {
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
        if ((rc = vrnn_step(c, sb, io, B, S))) return rc;
    }
    hipLaunchKernelGGL(vrnn_stats_kernel, dim3(1), dim3(256), 0, c->stream, kl, rec, B * T, Z, scalars2);
    return nm_check_hip(hipGetLastError(), "vrnn_encode");
}

int nm_vrnn_generate(nm_ctx* c, const float* keypoints_cond, const float* eps_post, const float* eps_prior, int32_t B,
                     int32_t Tcond, int32_t Ttot, int32_t S, float* out_cond, float* out_gen, float* h_last) {
    int rc = ready(c, "vrnn_generate", true);
    if (rc) return rc;
    if (!keypoints_cond || !eps_post || !out_cond || B <= 0 || Tcond <= 0 || Ttot < Tcond || S <= 0 || (Ttot > Tcond && (!eps_prior || !out_gen))) {
        nm_set_error("vrnn_generate: bad argument"); return NM_ERR_ARG;
    }
    if ((rc = max_fk_lds(c, S))) return rc;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4, Tg = Ttot - Tcond;
    size_t need = ((size_t)B * (4 * 128 + 3 * H + 4 * Z + K * 3 + 2 * H + Z) + (size_t)S * B * (Z + 256 + 3 + K + 6 * K)) * sizeof(float) + 64 * 256;
    if ((rc = nm_ctx_reserve(c, need))) return rc;
    c->ws.release(0);
    StepBufs sb = alloc_step(c->ws, B, S, K, Z, H);
    float* offset = c->ws.f((size_t)B * K * 3);
    float* hbuf[2] = {c->ws.f((size_t)B * H), c->ws.f((size_t)B * H)};
    float* zbuf = c->ws.f((size_t)B * Z);
    if (c->ws.overflow) { nm_set_error("vrnn_generate: workspace overflow"); return NM_ERR_STATE; }
    if ((rc = nm_vrnn_offsets(c, keypoints_cond, B, Tcond, offset))) return rc;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3((H * B + 255) / 256), dim3(256), 0, c->stream, c->vrnn.h0, H, hbuf[0], H, B);
    int cur = 0;
    for (int t = 0; t < Ttot; ++t) {
        StepIO io;
        const bool post = t < Tcond;
        io.h = hbuf[cur]; io.ldh = H;
        io.obs = post ? keypoints_cond + (size_t)t * S4 : nullptr; io.ldobs = Tcond * S4;
        io.eps = post ? eps_post + (size_t)t * S * B * Z : eps_prior + (size_t)(t - Tcond) * B * Z;
        io.offset = offset;
        io.out_kp = post ? out_cond + (size_t)t * S4 : out_gen + (size_t)(t - Tcond) * S4; io.ldkp = (post ? Tcond : Tg) * S4;
        io.out_z = zbuf; io.ldz = Z; io.out_R = nullptr; io.ldR = 0;
        io.best = nullptr; io.ldbest = 0; io.kl = nullptr; io.rec = nullptr; io.ldstat = 0;
        io.hout = hbuf[cur ^ 1]; io.ldho = H; io.want_prior = false;
        if ((rc = vrnn_step(c, sb, io, B, S))) return rc;
        cur ^= 1;
    }
    if (h_last) rc = nm_check_hip(hipMemcpyAsync(h_last, hbuf[cur], (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "generate: h_last");
    return rc;
}

int nm_vrnn_step(nm_ctx* c, int32_t posterior, const float* h_in, const float* kp_obs, const float* offset, const float* eps,
                 int32_t B, int32_t S, float* kp_out, float* z_out, float* h_out) {
    int rc = ready(c, "vrnn_step", true);
    if (rc) return rc;
    if (!h_in || !offset || !eps || !kp_out || !z_out || B <= 0 || (posterior && (!kp_obs || S <= 0))) {
        nm_set_error("vrnn_step: bad argument"); return NM_ERR_ARG;
    }
    if (!posterior) S = 1;
    if ((rc = max_fk_lds(c, S))) return rc;
    const int K = c->cfg.nkeypoints, Z = c->cfg.nlatent, H = c->cfg.nhidden, S4 = K * 4;
    size_t need = ((size_t)B * (4 * 128 + 3 * H + 4 * Z) + (size_t)S * B * (Z + 256 + 3 + K + 6 * K)) * sizeof(float) + 64 * 256;
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
}

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
                        int32_t* idx_out, float* dist_out) {
    if (!c || !rows || !target || !idx_out || B <= 0 || D <= 0) { nm_set_error("rows_argmin_dist: bad argument"); return NM_ERR_ARG; }
    hipLaunchKernelGGL(rows_argmin_kernel, dim3(1), dim3(256), 0, c->stream, rows, target, target_row_stride, B, D, idx_out, dist_out);
    return nm_check_hip(hipGetLastError(), "rows_argmin launch");
}

int nm_vrnn_mlp(nm_ctx* c, int32_t which, const float* x, int32_t B, float* y) {
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
}

int nm_vrnn_gru(nm_ctx* c, const float* x, const float* h, int32_t B, float* h_out) {
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
}

int nm_vrnn_fk(nm_ctx* c, const float* dec_in, const float* offset, int32_t B, float* kp, float* R) {
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
    a.order = w.order; a.parents = w.parents; a.qmu = a.qsig = a.pmu = a.psig = nullptr;
    a.out_kp = kp; a.ldkp = K * 4; a.out_z = nullptr; a.ldz = 0; a.out_R = R; a.ldR = K * 9;
    a.best = nullptr; a.ldbest = 0; a.kl = nullptr; a.rec = nullptr; a.ldstat = 0; a.K = K; a.S = 1; a.B = B; a.Z = Z;
    size_t lds = ((size_t)K * 21 + 1) * sizeof(float);
    hipLaunchKernelGGL(fk_kernel, dim3(B), dim3(256), lds, c->stream, a);
    return nm_check_hip(hipGetLastError(), "fk launch");
}

}  // extern "C"
