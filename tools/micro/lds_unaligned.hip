// Microbenchmark: what an LDS read costs by width and alignment when 64 lanes read at a ~74-byte stride (one lane per
// text line).  hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kIters = 256;

template <int kMode>
__global__ __launch_bounds__(256) void probe(const int *offs, uint64_t *out, long long *cycles) {
    __shared__ __attribute__((aligned(16))) uint8_t s[24 * 1024];
    for (int i = threadIdx.x; i < 24 * 1024 / 4; i += 256) reinterpret_cast<uint32_t *>(s)[i] = i * 2654435761u;
    __syncthreads();
    int o = offs[threadIdx.x];
    uint64_t acc = 0;
    const long long t0 = clock64();
#pragma unroll 8
    for (int k = 0; k < kIters; ++k) {
        const int a = (o + k * 8) % (24 * 1024 - 32);
        if (kMode == 0) { uint64_t v; __builtin_memcpy(&v, s + a, 8); acc ^= v; }                 // b64 at a byte offset
        if (kMode == 1) { uint64_t v; __builtin_memcpy(&v, s + (a & ~3), 8); acc ^= v; }          // b64, 4-byte aligned
        if (kMode == 2) { acc ^= *reinterpret_cast<const uint64_t *>(s + (a & ~7)); }              // b64, 8-byte aligned
        if (kMode == 3) { uint32_t v; __builtin_memcpy(&v, s + a, 4); acc ^= v; }                 // b32 at a byte offset
        if (kMode == 4) { acc ^= *reinterpret_cast<const uint32_t *>(s + (a & ~3)); }              // b32 aligned
        if (kMode == 5) { acc ^= s[a]; }                                                           // u8
        if (kMode == 6) { const uint4 v = *reinterpret_cast<const uint4 *>(s + (a & ~15)); acc ^= v.x ^ v.w; }  // b128 aligned
        if (kMode == 7) { const uint32_t *p = reinterpret_cast<const uint32_t *>(s + (a & ~3)); acc ^= (uint64_t)p[0] ^ p[1] ^ p[2]; }  // 3 aligned dwords
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    std::vector<int> h(256);
    for (int i = 0; i < 256; ++i) h[i] = i * 74 + (i * 7) % 5;
    int *d_off; uint64_t *d_out; long long *d_cyc;
    const int blocks = 256 * 6;
    hipMalloc(&d_off, 1024); hipMalloc(&d_out, blocks * 256 * 8); hipMalloc(&d_cyc, blocks * 8);
    hipMemcpy(d_off, h.data(), 1024, hipMemcpyHostToDevice);
    const char *names[] = {"b64 byte-offset", "b64 4B-aligned", "b64 8B-aligned", "b32 byte-offset", "b32 aligned", "u8", "b128 aligned", "3 x b32 aligned"};
    void (*ks[])(const int *, uint64_t *, long long *) = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>};
    for (int m = 0; m < 8; ++m) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(256), 0, 0, d_off, d_out, d_cyc);
        hipEventRecord(a);
        hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(256), 0, 0, d_off, d_out, d_cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // 6 blocks x 4 waves per CU share one LDS pipe: wave-level reads per CU = 24 * kIters
        const double cu_cycles = ms * 1e-3 * 2.4e9;
        printf("%-18s %.3f ms  ~%.1f LDS-pipe cycles per wave read (24 waves x %d reads per CU)\n", names[m], ms, cu_cycles / (24.0 * kIters), kIters);
    }
    return 0;
}
