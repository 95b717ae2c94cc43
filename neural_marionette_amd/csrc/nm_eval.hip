// Evaluation metrics of the reference on the device:
//   voxel_chamfer_distance (utils/eval_utils.py:29-55) and the nearest-keypoint votes of semantic_scores (:59-90).
//
// Chamfer: the reference lists the occupied voxels of the ground truth and of the thresholded reconstruction and builds
// the full (N, M) squared-distance matrix per frame.  Both sets live on the same G^3 lattice, so the nearest-neighbour
// squared distance of every voxel is an exact Euclidean distance transform in integer index units: three separable
// min-plus passes of G steps each, O(G^4) per frame instead of O(N M), independent of how full the volume is, and exact
// (the float result is sum_int * (2/(G-1))^2 / count).
#include "nm_ctx.h"

namespace {

constexpr int EDT_INF = 1 << 28;

// pass along x: d[z][y][x] = min over occupied x' of (x - x')^2.  One block per (frame, z, y) row, one thread per x.
__global__ void edt_x_kernel(const float* __restrict__ vox, float thr, int G, int* __restrict__ d) {
    extern __shared__ int occ[];
    const size_t row = blockIdx.x;
    const int x = threadIdx.x;
    const float v = vox[row * G + x];
    occ[x] = (thr > 0.f) ? (v >= thr) : (v != 0.f);
    __syncthreads();
    int best = EDT_INF;
    for (int xp = 0; xp < G; ++xp)
        if (occ[xp]) { const int dd = (x - xp) * (x - xp); best = dd < best ? dd : best; }
    d[row * G + x] = best;
}

// pass along one of the other axes: out[i] = min over i' of in[i'] + (i - i')^2 with stride `stride` between i's.
// One block per line, one thread per position.
__global__ void edt_axis_kernel(const int* __restrict__ in, int G, int axis /*1: y, 2: z*/, int* __restrict__ out) {
    extern __shared__ int line[];
    // lines: axis 1 -> (frame, z, x), axis 2 -> (frame, y, x)
    const size_t li = blockIdx.x;
    const int i = threadIdx.x;
    size_t base; int stride;
    if (axis == 1) { const size_t fz = li / G; const int x = li % G; base = fz * G * G + x; stride = G; }
    else { const size_t f = li / ((size_t)G * G); const int yx = li % (G * G); base = f * (size_t)G * G * G + yx; stride = G * G; }
    line[i] = in[base + (size_t)i * stride];
    __syncthreads();
    int best = EDT_INF;
    for (int ip = 0; ip < G; ++ip) { const int c = line[ip] + (i - ip) * (i - ip); best = c < best ? c : best; }
    out[base + (size_t)i * stride] = best;
}

// per frame: sum of d_other over the occupied voxels of a set, and their count (integer atomics: deterministic)
__global__ void chamfer_sum_kernel(const float* __restrict__ vox, float thr, const int* __restrict__ d_other, int G3,
                                   unsigned long long* __restrict__ sums /*[frame][2]*/) {
    const int f = blockIdx.y;
    unsigned long long s = 0, n = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < G3; i += gridDim.x * blockDim.x) {
        const float v = vox[(size_t)f * G3 + i];
        const bool o = (thr > 0.f) ? (v >= thr) : (v != 0.f);
        if (o) { s += (unsigned long long)d_other[(size_t)f * G3 + i]; ++n; }
    }
    for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); n += __shfl_xor(n, off); }
    if ((threadIdx.x & 63) == 0 && n) { atomicAdd(&sums[2 * f], s); atomicAdd(&sums[2 * f + 1], n); }
}

__global__ void chamfer_final_kernel(const unsigned long long* __restrict__ s_gt, const unsigned long long* __restrict__ s_rec, int F,
                                     double scale2, double* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const double n1 = (double)s_gt[2 * f + 1], n2 = (double)s_rec[2 * f + 1];
    // an empty set has no nearest neighbour: the reference's min over an empty dimension raises; NaN here
    out[f] = (n1 > 0 && n2 > 0) ? ((double)s_gt[2 * f] / n1 + (double)s_rec[2 * f] / n2) * scale2 : __longlong_as_double(0x7ff8000000000000LL);
}

// semantic_scores: for every ground-truth keypoint the detected keypoint nearest to it (first index on ties), detections
// with intensity < 0.2 moved to (1e4, 1e4, 1e4) as the reference does; votes accumulated into counts[k'][k]
__global__ void semantic_kernel(const float* __restrict__ kypt, const float* __restrict__ gt, int BT, int K, int Kg,
                                int* __restrict__ closest, long long* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BT * Kg) return;
    const int bt = i / Kg, kg = i % Kg;
    const float gx = gt[(size_t)i * 3], gy = gt[(size_t)i * 3 + 1], gz = gt[(size_t)i * 3 + 2];
    float best = 0.f; int arg = 0;
    for (int k = 0; k < K; ++k) {
        const float* q = kypt + ((size_t)bt * K + k) * 4;
        const bool bad = q[3] < 0.2f;
        const float dx = gx - (bad ? 1e4f : q[0]), dy = gy - (bad ? 1e4f : q[1]), dz = gz - (bad ? 1e4f : q[2]);
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (k == 0 || d < best) { best = d; arg = k; }
    }
    closest[i] = arg;
    atomicAdd(reinterpret_cast<unsigned long long*>(&counts[(size_t)kg * K + arg]), 1ULL);
}

}  // namespace

extern "C" {

int nm_eval_voxel_chamfer(nm_ctx* c, const float* gt_vox, const float* recon, int32_t B, int32_t T, int32_t G, double* per_frame) try { NmScope nm_scope_(c);
    if (!c || !gt_vox || !recon || !per_frame || B <= 0 || T <= 0 || G < 2 || G > 1024) { nm_set_error("eval_voxel_chamfer: bad argument"); return NM_ERR_ARG; }
    int rc = nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
    if (rc) return rc;
    const int F = B * T;
    const size_t G3 = (size_t)G * G * G;
    // frames are processed in chunks so that the two distance fields + a scratch field stay under ~1 GB
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)F, ((size_t)1 << 28) / (G3 * 3)));
    const size_t need = (size_t)chunk * G3 * 3 * sizeof(int) + (size_t)F * 4 * sizeof(unsigned long long) + 4096;
    if ((rc = nm_ctx_reserve(c, need))) return rc;
    c->ws.release(0);
    int* da = static_cast<int*>(c->ws.alloc_bytes((size_t)chunk * G3 * sizeof(int)));
    int* db = static_cast<int*>(c->ws.alloc_bytes((size_t)chunk * G3 * sizeof(int)));
    int* dt = static_cast<int*>(c->ws.alloc_bytes((size_t)chunk * G3 * sizeof(int)));
    unsigned long long* s_gt = static_cast<unsigned long long*>(c->ws.alloc_bytes((size_t)F * 4 * sizeof(unsigned long long)));
    unsigned long long* s_rec = s_gt + 2 * (size_t)F;                     // one block: zeroed by the single memset below
    hipStream_t s = c->stream;
    if ((rc = nm_check_hip(hipMemsetAsync(s_gt, 0, (size_t)F * 4 * sizeof(unsigned long long), s), "memset"))) return rc;
    auto edt = [&](const float* vox, float thr, int nf, int* out) {      // out = exact squared distance to the nearest occupied voxel
        hipLaunchKernelGGL(edt_x_kernel, dim3((unsigned)((size_t)nf * G * G)), dim3(G), G * sizeof(int), s, vox, thr, G, out);
        hipLaunchKernelGGL(edt_axis_kernel, dim3((unsigned)((size_t)nf * G * G)), dim3(G), G * sizeof(int), s, out, G, 1, dt);
        hipLaunchKernelGGL(edt_axis_kernel, dim3((unsigned)((size_t)nf * G * G)), dim3(G), G * sizeof(int), s, dt, G, 2, out);
    };
    for (int f0 = 0; f0 < F; f0 += chunk) {
        const int nf = std::min(chunk, F - f0);
        const float* g = gt_vox + (size_t)f0 * G3; const float* r = recon + (size_t)f0 * G3;
        edt(g, 0.f, nf, da);                                               // distance to the ground-truth set
        edt(r, 0.5f, nf, db);                                              // distance to the reconstruction (recon >= 0.5)
        hipLaunchKernelGGL(chamfer_sum_kernel, dim3(64, nf), dim3(256), 0, s, g, 0.f, db, (int)G3, s_gt + 2 * (size_t)f0);
        hipLaunchKernelGGL(chamfer_sum_kernel, dim3(64, nf), dim3(256), 0, s, r, 0.5f, da, (int)G3, s_rec + 2 * (size_t)f0);
    }
    const double sc = 2.0 / (double)(G - 1);
    hipLaunchKernelGGL(chamfer_final_kernel, dim3((F + 63) / 64), dim3(64), 0, s, s_gt, s_rec, F, sc * sc, per_frame);
    return nm_check_hip(hipGetLastError(), "eval_voxel_chamfer launch");
} catch (...) { return nm_abi_catch("nm_eval_voxel_chamfer"); }

int nm_eval_semantic(nm_ctx* c, const float* keypoints, const float* gt_keypoints, int32_t BT, int32_t K, int32_t Kg,
                     int32_t* closest, int64_t* counts) try { NmScope nm_scope_(c);
    if (!c || !keypoints || !gt_keypoints || !closest || !counts || BT <= 0 || K <= 0 || Kg <= 0) { nm_set_error("eval_semantic: bad argument"); return NM_ERR_ARG; }
    int rc = nm_check_hip(hipSetDevice(c->cfg.device), "hipSetDevice");
    if (rc) return rc;
    hipLaunchKernelGGL(semantic_kernel, dim3((BT * Kg + 127) / 128), dim3(128), 0, c->stream, keypoints, gt_keypoints, BT, K, Kg, closest,
                       reinterpret_cast<long long*>(counts));
    return nm_check_hip(hipGetLastError(), "eval_semantic launch");
} catch (...) { return nm_abi_catch("nm_eval_semantic"); }

}  // extern "C"
