#include <hip/hip_runtime.h>
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ int dppmov(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, BANK, false); }
template <int OFF> __device__ __forceinline__ int xor_i(int x) {
    if constexpr (OFF == 1) return dppmov<0xB1>(x, x);
    else if constexpr (OFF == 2) return dppmov<0x4E>(x, x);
    else if constexpr (OFF == 4) { int r = dppmov<0x104, 0x5>(x, x); return dppmov<0x114, 0xA>(r, x); }
    else if constexpr (OFF == 8) return dppmov<0x128>(x, x);
    else if constexpr (OFF == 16) { auto p = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false); return (threadIdx.x & 16) ? (int)p[0] : (int)p[1]; }
    else { auto p = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false); return (threadIdx.x & 32) ? (int)p[0] : (int)p[1]; }
}
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[0 * 64 + l] = xor_i<1>(l); out[1 * 64 + l] = xor_i<2>(l); out[2 * 64 + l] = xor_i<4>(l);
    out[3 * 64 + l] = xor_i<8>(l); out[4 * 64 + l] = xor_i<16>(l); out[5 * 64 + l] = xor_i<32>(l);
}
int main() {
    int* d; hipMalloc(&d, 6 * 64 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); int h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0; const int offs[6] = {1, 2, 4, 8, 16, 32};
    for (int s = 0; s < 6; ++s) for (int l = 0; l < 64; ++l) if (h[s * 64 + l] != (l ^ offs[s])) { if (bad < 10) printf("off %d lane %d got %d\n", offs[s], l, h[s * 64 + l]); ++bad; }
    printf("bad %d\n", bad); return bad != 0;
}
