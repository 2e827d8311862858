// Microbenchmark: the duration of ONE kernel as two hipEventRecord calls around its launch report it, and as the two events report it that
// hipExtLaunchKernelGGL binds to the dispatch itself (the kernel's own begin / end timestamps, what rocprofv3's kernel trace reads).
// Also: a second kernel bound to the same stop event (does the later binding win?) and an ext-bound start with a recorded stop.
// hipcc --offload-arch=gfx950 -O3 ext_events.hip -o ext_events.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>

__global__ void spin_kernel(uint32_t *out, int spin) {
    const long long t0 = wall_clock64();   // 100 MHz
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 64);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b, c, d, e;
    hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c); hipEventCreate(&d); hipEventCreate(&e);
    for (int spin : {0, 1000, 5000, 50000}) {   // bodies of ~0, 10, 50, 500 us
        double rec = 0, ext = 0, ext2 = 0, mixed = 0;
        const int n = 200;
        for (int i = 0; i < n + 20; ++i) {
            float ms = 0;
            hipEventRecord(a, s);
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, out, spin);
            hipEventRecord(b, s);
            hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, c, d, 0, out, spin);
            hipStreamSynchronize(s);
            if (i >= 20) {
                if (hipEventElapsedTime(&ms, a, b) == hipSuccess) rec += ms; else printf("rec failed\n");
                hipError_t r = hipEventElapsedTime(&ms, c, d);
                if (r == hipSuccess) ext += ms; else printf("ext failed: %s\n", hipGetErrorString(r));
            }
            // two kernels: start bound to the first, stop bound to both (the later binding should win)
            hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, c, d, 0, out, spin);
            hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, nullptr, d, 0, out, spin);
            hipStreamSynchronize(s);
            if (i >= 20) {
                hipError_t r = hipEventElapsedTime(&ms, c, d);
                if (r == hipSuccess) ext2 += ms; else printf("ext2 failed: %s\n", hipGetErrorString(r));
            }
            // ext-bound start, recorded stop behind a second kernel
            hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, c, nullptr, 0, out, spin);
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, out, spin);
            hipEventRecord(e, s);
            hipStreamSynchronize(s);
            if (i >= 20) {
                hipError_t r = hipEventElapsedTime(&ms, c, e);
                if (r == hipSuccess) mixed += ms; else printf("mixed failed: %s\n", hipGetErrorString(r));
            }
        }
        printf("spin %6d ticks: recorded events %8.2f us | bound to the dispatch %8.2f us | two kernels, stop rebound %8.2f us | bound start + recorded stop over two kernels %8.2f us\n",
               spin, rec / n * 1e3, ext / n * 1e3, ext2 / n * 1e3, mixed / n * 1e3);
    }
    return 0;
}
