// Microbenchmark: how long after a kernel's last store does the host learn of it -- through hipStreamSynchronize, and by polling a word of
// pinned host memory the kernel writes with a system-scope store.  hipcc --offload-arch=gfx950 -O3 sync_latency.hip -o sync_latency.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void flag_kernel(volatile uint32_t *flag, uint32_t v, int spin) {
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    __hip_atomic_store((uint32_t *)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    uint32_t *flag;
    hipHostMalloc(&flag, 64, hipHostMallocDefault);
    *flag = 0;
    hipStream_t s;
    hipStreamCreate(&s);
    using clk = std::chrono::steady_clock;
    for (int spin : {0, 20000, 200000}) {   // kernel bodies of ~0, ~10, ~100 us
        double t_sync = 0, t_poll = 0, t_poll_then_sync = 0;
        const int n = 2000;
        for (int i = 0; i < n; ++i) {   // (a) launch + hipStreamSynchronize
            auto a = clk::now();
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, (uint32_t)(i + 1), spin);
            hipStreamSynchronize(s);
            t_sync += std::chrono::duration<double, std::micro>(clk::now() - a).count();
        }
        *flag = 0;
        for (int i = 0; i < n; ++i) {   // (b) launch + poll the pinned word
            auto a = clk::now();
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, (uint32_t)(i + 1), spin);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (uint32_t)(i + 1)) {}
            t_poll += std::chrono::duration<double, std::micro>(clk::now() - a).count();
        }
        hipStreamSynchronize(s);
        *flag = 0;
        for (int i = 0; i < n; ++i) {   // (c) poll, then the synchronize the next call would need anyway before reusing buffers
            auto a = clk::now();
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, (uint32_t)(i + 1), spin);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (uint32_t)(i + 1)) {}
            auto b = clk::now();
            hipStreamSynchronize(s);
            t_poll += 0;
            t_poll_then_sync += std::chrono::duration<double, std::micro>(clk::now() - b).count();
            (void)a;
        }
        double t_evs = 0, t_evq = 0, t_sq = 0;
        hipEvent_t ev;
        hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        for (int i = 0; i < n; ++i) {   // (d) event record + hipEventSynchronize
            auto a = clk::now();
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, (uint32_t)(i + 1), spin);
            hipEventRecord(ev, s);
            hipEventSynchronize(ev);
            t_evs += std::chrono::duration<double, std::micro>(clk::now() - a).count();
        }
        for (int i = 0; i < n; ++i) {   // (e) event record + spinning on hipEventQuery
            auto a = clk::now();
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, (uint32_t)(i + 1), spin);
            hipEventRecord(ev, s);
            while (hipEventQuery(ev) == hipErrorNotReady) {}
            t_evq += std::chrono::duration<double, std::micro>(clk::now() - a).count();
        }
        for (int i = 0; i < n; ++i) {   // (f) spinning on hipStreamQuery
            auto a = clk::now();
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, (uint32_t)(i + 1), spin);
            while (hipStreamQuery(s) == hipErrorNotReady) {}
            t_sq += std::chrono::duration<double, std::micro>(clk::now() - a).count();
        }
        printf("   event record + hipEventSynchronize %.2f us, + hipEventQuery spin %.2f us, hipStreamQuery spin %.2f us\n", t_evs / n, t_evq / n, t_sq / n);
        printf("kernel body ~%d cycles: launch+synchronize %.2f us, launch+poll %.2f us, synchronize after the poll %.2f us\n", spin, t_sync / n, t_poll / n,
               t_poll_then_sync / n);
    }
    return 0;
}
