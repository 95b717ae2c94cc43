// Granule hand-off latency between two workgroups (ping-pong of one 8-byte {value, tag} granule), by store flavour and placement:
//   same XCD  + plain store (line stays in that XCD's L2)  + sc1 load (bypasses L1, L2-served)
//   same XCD  + sc1 store (write-through, drops the line)   + sc1 load
//   other XCD + sc1 store                                    + sc1 load
// Placement is read from HW_REG_XCC_ID, not assumed: 256 workgroups are launched, the first to arrive picks the pair.
// build: hipcc -O3 --offload-arch=gfx950 tools/calib/hop_latency.hip -o /tmp/hop_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID, bits [3:0]
__device__ __forceinline__ void st_plain(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc1(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u64 ld_sc1(const u64* p) { u64 v; asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }

// mode 0: same XCD plain store; 1: same XCD sc1 store; 2: other XCD sc1 store; 3: other XCD plain store (expected: never visible -> bounded spin reports it)
__global__ void pingpong(u64* g, int* claim, int mode, int iters, long long* out) {
    __shared__ int role;
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        role = -1;
        // first arriver fixes XCD A; role 0 = first block on A; role 1 = first other block on A (modes 0, 1) or first block NOT on A (modes 2, 3)
        int a = atomicCAS(&claim[0], -1, x);
        if (a == -1) a = x;
        const bool same = mode < 2;
        if (x == a) { const int k = atomicAdd(&claim[1], 1); if (k == 0) role = 0; else if (same && k == 1) role = 1; }
        else if (!same) { if (atomicAdd(&claim[2], 1) == 0) role = 1; }
    }
    __syncthreads();
    if (role < 0 || threadIdx.x != 0) return;
    u64* mine = g + (role ? 16 : 0); u64* theirs = g + (role ? 0 : 16);      // separate 128-B lines
    const bool plain = mode == 0 || mode == 3;
    long long t0 = 0;
    for (int i = 1; i <= iters; ++i) {
        if (i == 17) t0 = wall_clock64();
        if (role == 0) { if (plain) st_plain(mine, (u64)i); else st_sc1(mine, (u64)i); }
        long long spins = 0;
        while (ld_sc1(theirs) != (u64)i) { if (++spins > 2000000) { out[2] = i; return; } }
        if (role == 1) { if (plain) st_plain(mine, (u64)i); else st_sc1(mine, (u64)i); }
    }
    if (role == 0) { out[0] = wall_clock64() - t0; out[1] = iters - 16; }
}

int main() {
    u64* g; int* claim; long long* out;
    hipMalloc(&g, 4096); hipMalloc(&claim, 64); hipMalloc(&out, 64);
    const char* names[4] = {"same XCD, plain store + sc1 load", "same XCD, sc1 store + sc1 load", "other XCD, sc1 store + sc1 load", "other XCD, plain store + sc1 load"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            int h[3] = {-1, 0, 0}; long long z[3] = {0, 0, 0};
            hipMemset(g, 0, 4096); hipMemcpy(claim, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(out, z, sizeof(z), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 0, 0, g, claim, mode, 2016, out);
            hipDeviceSynchronize();
            hipMemcpy(z, out, sizeof(z), hipMemcpyDeviceToHost);
            if (z[2]) printf("%-36s: NOT VISIBLE (spin limit at iteration %lld)\n", names[mode], z[2]);
            else printf("%-36s: %.3f us per round trip = %.3f us per hop (wall_clock64 at 100 MHz)\n", names[mode], z[0] / 100.0 / z[1], z[0] / 200.0 / z[1]);
        }
    }
    return 0;
}
