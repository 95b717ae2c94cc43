// FETCH_SIZE calibration (verdict r5 item 9): streaming reads of a KNOWN byte count in the access patterns this library's kernels use, so
// that a rocprofv3 --pmc FETCH_SIZE pass gives the counter's factor per pattern (MI355X_MICROARCH.md states x2 for 16 B per lane, fully
// contiguous; the staging loads of the conv kernels are 16-byte pieces in 64- or 128-byte runs at the voxel stride).
//   pattern 0  contig16   : lane i reads 16 B at i * 16 (one dwordx4 per lane, 1 KiB per wave instruction)        - elementwise kernels, decoder tail
//   pattern 1  pair32     : lane i reads 32 B at i * 32 as two dwordx4 instructions (16-B pieces at a 32-B pitch) - conv_up2c / conv_pool staging
//   pattern 2  seg64_256  : 4 lanes read one 64-B run, runs 256 B apart (a 16-channel chunk of a 64-channel voxel) - conv_f16p2 producers, Cin = 64
//   pattern 3  seg64_128  : the same with runs 128 B apart (Cin = 32)                                             - conv_f16p producers
//   pattern 4  dword      : lane i reads 4 B at i * 4                                                             - reductions over NCDHW tensors
// Every pattern touches `bytes` DISTINCT bytes exactly once per launch (patterns 2 / 3 make (stride / 64) passes, one per chunk position, so
// the whole buffer is read; a pass's lines are touched again by the other passes from HBM or L2 - that re-touching is what the conv kernels do).
// build: hipcc -O3 --offload-arch=gfx950 tools/calib/fetch_calib.hip -o tools/calib/fetch_calib     run: fetch_calib <GiB> (default 2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(256) void read_kernel(const char* __restrict__ buf, size_t bytes, float* __restrict__ sink) {
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (P == 0) {
        for (size_t o = tid * 16; o < bytes; o += nthr * 16) acc += *reinterpret_cast<const f4*>(buf + o);
    } else if (P == 1) {
        for (size_t o = tid * 32; o < bytes; o += nthr * 32) { acc += *reinterpret_cast<const f4*>(buf + o); acc += *reinterpret_cast<const f4*>(buf + o + 16); }
    } else if (P == 2 || P == 3) {
        constexpr size_t S = P == 2 ? 256 : 128;
        const size_t nrun = bytes / S;                       // runs per pass
        for (size_t c = 0; c < S / 64; ++c)                  // pass c reads chunk c of every voxel
            for (size_t r = tid >> 2; r < nrun; r += nthr >> 2) acc += *reinterpret_cast<const f4*>(buf + r * S + c * 64 + (tid & 3) * 16);
    } else {
        for (size_t o = tid * 4; o < bytes; o += nthr * 4) acc[0] += *reinterpret_cast<const float*>(buf + o);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[0] = acc[0];       // (never true for the zero-filled buffer: keeps the loads)
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
    char* buf; float* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[5] = {"contig16", "pair32", "seg64_256", "seg64_128", "dword"};
    for (int rep = 0; rep < 2; ++rep)
        for (int p = 0; p < 5; ++p) {
            hipEventRecord(a);
            switch (p) {
                case 0: hipLaunchKernelGGL(read_kernel<0>, dim3(256 * 16), dim3(256), 0, 0, buf, bytes, sink); break;
                case 1: hipLaunchKernelGGL(read_kernel<1>, dim3(256 * 16), dim3(256), 0, 0, buf, bytes, sink); break;
                case 2: hipLaunchKernelGGL(read_kernel<2>, dim3(256 * 16), dim3(256), 0, 0, buf, bytes, sink); break;
                case 3: hipLaunchKernelGGL(read_kernel<3>, dim3(256 * 16), dim3(256), 0, 0, buf, bytes, sink); break;
                default: hipLaunchKernelGGL(read_kernel<4>, dim3(256 * 16), dim3(256), 0, 0, buf, bytes, sink); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0.f; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("pattern %d %-10s %.3f GB in %.3f ms = %.2f TB/s\n", p, names[p], bytes / 1e9, ms, bytes / 1e9 / ms);
        }
    return 0;
}
