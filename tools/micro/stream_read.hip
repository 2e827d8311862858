// Streaming-read structures over a 4 GB int32 column (round 6: what tile / workgroup shape reads fastest, as a floor for q5's count pass).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/stream_read.hip -o tools/micro/stream_read && tools/micro/stream_read
// one<W, ITERS>     : one tile of W x 4 x ITERS rows per workgroup of W threads, all loads issued up front (the count pass's shape: W = 256, ITERS = 8)
// persist<W, ITERS> : grid = CUs x k workgroups walking tiles b, b + G, ... (q7_max_kernel's shape)
#include <hip/hip_runtime.h>

#ifdef USE_LIB   // -DUSE_LIB -Iinclude -Lflock_amd -lflockgpu: the column is the library's NEXMark generator's (1087 s x 1e6 events/s), not the stand-in of fill()
#include "../../include/flockgpu.h"
#endif
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int W, int ITERS, bool NT>
__global__ __launch_bounds__(W) void one(const int32_t *__restrict__ a, int64_t n, unsigned long long *out) {
    const int64_t base = (int64_t)blockIdx.x * (W * 4 * ITERS) + threadIdx.x * 4;
    v4i v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const v4i *p = reinterpret_cast<const v4i *>(a + base + (int64_t)it * W * 4);
        v[it] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    int acc = 0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) acc += v[it].x ^ v[it].y ^ v[it].z ^ v[it].w;
    if (acc == 0x7fffffff) atomicAdd(out, 1ull);
}

template <int W, int ITERS>
__global__ __launch_bounds__(W) void persist(const int32_t *__restrict__ a, int64_t n_tiles, unsigned long long *out) {
    int acc = 0;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t base = t * (W * 4 * ITERS) + threadIdx.x * 4;
        v4i v[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) v[it] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(a + base + (int64_t)it * W * 4));
#pragma unroll
        for (int it = 0; it < ITERS; ++it) acc += v[it].x ^ v[it].y ^ v[it].z ^ v[it].w;
    }
    if (acc == 0x7fffffff) atomicAdd(out, 1ull);
}

// NEXMark-like keys: an id that grows with the row, half the rows on the "hot" id of their neighbourhood, the rest spread over ~110 ids
__global__ void fill(int32_t *a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        const int32_t front = (int32_t)(i / 13);
        a[i] = (h & 1) ? (front / 100) * 100 : front - (int32_t)((h >> 8) % 110);
    }
}

struct Tile { int64_t begin, lo, hi; int32_t seg, pad; };
__global__ void make_tiles(Tile *t, int64_t n_tiles, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tiles) t[i] = Tile{i * rows, i * rows, (i + 1) * rows, (int32_t)(i >> 9), 0};
}

// the count pass's front end: tile descriptor (dependent load), 8 x 16-byte loads, min / max over the workgroup (one barrier)
template <bool LDS_HIST, bool FLUSH>
__global__ __launch_bounds__(256) void front(const int32_t *__restrict__ a, const Tile *__restrict__ tiles, uint32_t *counters, unsigned long long *out) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[4096 + 64];
    __shared__ int32_t s_red[8];
    const Tile tr = tiles[blockIdx.x];
    v4i v[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) v[it] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(a + tr.begin + it * 1024 + threadIdx.x * 4));
    if (LDS_HIST) {
        uint4 *z = reinterpret_cast<uint4 *>(hist);
        for (int s = threadIdx.x; s < (4096 + 64) / 4; s += 256) z[s] = make_uint4(0, 0, 0, 0);
    }
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        mn = min(mn, min(min(v[it].x, v[it].y), min(v[it].z, v[it].w)));
        mx = max(mx, max(max(v[it].x, v[it].y), max(v[it].z, v[it].w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        s_red[threadIdx.x >> 6] = mn;
        s_red[4 + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
    mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const uint32_t span = (uint32_t)mx - (uint32_t)mn;
    if (!LDS_HIST) {
        if (span == 0x7fffffffu) atomicAdd(out, 1ull);
        return;
    }
    if (span >= 4096u) return;
    // (hot key handling as in q5_count_tile, simplified: the wave's first key)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int32_t k4[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        const int32_t hot = __builtin_amdgcn_readfirstlane(k4[0]) / 100 * 100;
        uint32_t hc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool is_hot = k4[j] == hot;
            hc += (uint32_t)__popcll(__ballot(is_hot));
            if (!is_hot) atomicAdd(&hist[(uint32_t)k4[j] - (uint32_t)mn], 1u);
        }
        if (lane == 0 && hc && (uint32_t)hot - (uint32_t)mn < 4096u) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], hc);
    }
    __syncthreads();
    if (!FLUSH) {
        if (hist[threadIdx.x] == 0x7fffffffu) atomicAdd(out, 1ull);
        return;
    }
    uint32_t *cnt = counters + (size_t)(tr.seg & 255) * (1u << 19);   // (a pane's counters: 512 tiles x ~630 ids fit 2^19)
    for (uint32_t s = threadIdx.x; s <= span; s += 256) {
        const uint32_t c = hist[s];
        if (c) __hip_atomic_fetch_add(&cnt[((uint32_t)mn + s) & ((1u << 19) - 1)], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The count pass's COMPUTE phase alone, as q5_count_tile has it (hot key in scalar registers, exec-masked LDS adds, flush), on keys made in
// registers (kLoad = false) or loaded (kLoad = true): what the CU needs per tile when no memory is waited for.
struct Pane { int64_t base; uint64_t cnt_off; uint32_t range, pad; };
template <bool kLoad, bool kFlush, bool kSpec = false, bool kPaneFirst = false, bool kL2 = false>
__global__ __launch_bounds__(256) void count_like(const int32_t *__restrict__ a, const Tile *__restrict__ tiles, uint32_t *counters, unsigned long long *out,
                                                  const uint64_t *__restrict__ spec = nullptr, const int32_t *__restrict__ pane_win_ptr = nullptr, const Pane *__restrict__ panes = nullptr) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[4096 + 64];
    __shared__ int32_t s_red[8];
    if (kSpec && spec && !spec[2]) return;
    const Tile tr = tiles[blockIdx.x];
    Pane pn{0, 0, 0, 0};
    if (kPaneFirst) {
        if (pane_win_ptr[tr.seg] == pane_win_ptr[tr.seg + 1]) return;
        pn = panes[tr.seg];
    }
    int32_t k[8][4];
    if (kLoad) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int64_t begin = kL2 ? (tr.begin & ((int64_t(64) << 13) - 1)) : tr.begin;   // kL2: every workgroup reads one of the first 64 tiles (2 MB: cache-resident)
            const v4i t = kL2 ? *reinterpret_cast<const v4i *>(a + begin + it * 1024 + threadIdx.x * 4)
                              : __builtin_nontemporal_load(reinterpret_cast<const v4i *>(a + begin + it * 1024 + threadIdx.x * 4));
            k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
        }
    } else {
#pragma unroll
        for (int it = 0; it < 8; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t i = tr.begin + it * 1024 + threadIdx.x * 4 + j;
                uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
                h ^= h >> 29;
                const int32_t front = (int32_t)(i / 13);
                k[it][j] = (h & 1) ? (front / 100) * 100 : front - (int32_t)((h >> 8) % 110);
            }
    }
    {
        uint4 *z = reinterpret_cast<uint4 *>(hist);
        for (int s = threadIdx.x; s < (4096 + 64) / 4; s += 256) z[s] = make_uint4(0, 0, 0, 0);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        mn = min(mn, min(min(k[it][0], k[it][1]), min(k[it][2], k[it][3])));
        mx = max(mx, max(max(k[it][0], k[it][1]), max(k[it][2], k[it][3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (lane == 0) {
        s_red[wave] = mn;
        s_red[4 + wave] = mx;
    }
    __syncthreads();
    mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const uint32_t span = (uint32_t)mx - (uint32_t)mn;
    if (span >= 4096u) return;
    int32_t hot = __builtin_amdgcn_readfirstlane(k[0][0]);
    uint32_t hot_cnt = 0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        uint64_t b0 = __ballot(k[it][0] == hot);
        if (__popcll((unsigned long long)b0) < 16) {
            if (hot_cnt) {
                if (lane == 0) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], hot_cnt);
                hot_cnt = 0;
            }
            const int32_t c1 = __builtin_amdgcn_readfirstlane(k[it][0]);
            const uint64_t m1 = __ballot(k[it][0] == c1);
            hot = c1;
            b0 = m1;
            if (__popcll((unsigned long long)m1) < 16 && ~m1) {
                const int l2 = __ffsll((unsigned long long)~m1) - 1;
                const int32_t c2 = __builtin_amdgcn_readlane(k[it][0], l2);
                const uint64_t m2 = __ballot(k[it][0] == c2);
                if (__popcll((unsigned long long)m2) > __popcll((unsigned long long)m1)) {
                    hot = c2;
                    b0 = m2;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool is_hot = k[it][j] == hot;
            const uint64_t b = (j == 0) ? b0 : __ballot(is_hot);
            hot_cnt += (uint32_t)__popcll((unsigned long long)b);
            if (!is_hot) atomicAdd(&hist[(uint32_t)k[it][j] - (uint32_t)mn], 1u);
        }
    }
    if (hot_cnt && lane == 0) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], hot_cnt);
    __syncthreads();
    if (!kFlush) {
        if (hist[threadIdx.x] == 0x7fffffffu) atomicAdd(out, 1ull);
        return;
    }
    uint32_t *cnt = counters + (size_t)(tr.seg & 255) * (1u << 19) + (kPaneFirst ? (uint32_t)pn.cnt_off : 0u);
    for (uint32_t s = threadIdx.x; s <= span; s += 256) {
        const uint32_t c = hist[s];
        if (c) __hip_atomic_fetch_add(&cnt[((uint32_t)mn + s) & ((1u << 19) - 1)], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <class F> static double time_ms(F f, int reps = 10) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const int64_t n = 1000046592;   // a multiple of 65536 rows near 1e9 (4.0 GB)
    int32_t *a;
    unsigned long long *out;
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&out, 8));
#ifdef USE_LIB
    {
        flockgpu_ctx *ctx = nullptr;
        if (flockgpu_ctx_create(0, nullptr, &ctx)) { printf("ctx: %s\n", flockgpu_last_error(ctx)); return 1; }
        flockgpu_nexmark_stream st{20260925ull, 0, 1000000, 1436918400000ull};
        uint64_t np, na, nb;
        flockgpu_nexmark_counts(&st, 0, 1088ull * 1000000, &np, &na, &nb);
        int32_t *gen = nullptr;
        CK(hipMalloc(&gen, nb * 4 + 64));
        if (flockgpu_nexmark_gen_bids(ctx, &st, 0, 1088ull * 1000000, gen, nullptr, nullptr, nullptr)) { printf("gen: %s\n", flockgpu_last_error(ctx)); return 1; }
        flockgpu_ctx_synchronize(ctx);
        CK(hipMemcpy(a, gen, n * 4, hipMemcpyDeviceToDevice));
        CK(hipFree(gen));
        printf("column: the library's NEXMark bids (%llu generated, %lld used)\n", (unsigned long long)nb, (long long)n);
    }
#else
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, a, n);
#endif
    const int64_t n_tiles = n / 8192;
    Tile *tiles;
    uint32_t *counters;
    CK(hipMalloc(&tiles, n_tiles * sizeof(Tile)));
    CK(hipMalloc(&counters, (size_t)256 << 21));
    CK(hipMemset(counters, 0, (size_t)256 << 21));
    uint64_t *spec;
    int32_t *pwp;
    Pane *panes;
    CK(hipMalloc(&spec, 64));
    CK(hipMalloc(&pwp, 4096 * 4));
    CK(hipMalloc(&panes, 4096 * sizeof(Pane)));
    {
        uint64_t hs[4] = {1, 1, 1, 0};
        CK(hipMemcpy(spec, hs, 32, hipMemcpyHostToDevice));
        int32_t hp[4096];
        for (int i = 0; i < 4096; ++i) hp[i] = i * 2;
        CK(hipMemcpy(pwp, hp, sizeof hp, hipMemcpyHostToDevice));
        CK(hipMemset(panes, 0, 4096 * sizeof(Pane)));
    }
    hipLaunchKernelGGL(make_tiles, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, 0, tiles, n_tiles, (int64_t)8192);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    auto report = [&](const char *name, double ms) { printf("%-28s %.4f ms  %.2f TB/s\n", name, ms, n * 4.0 / ms / 1e9); };
#define ONE(W, IT, NT) report("one<" #W "," #IT "," #NT ">", time_ms([&] { hipLaunchKernelGGL((one<W, IT, NT>), dim3((unsigned)(n / (W * 4 * IT))), dim3(W), 0, 0, a, n, out); }))
    for (int round = 0; round < 2; ++round) {
        report("count_like compute only", time_ms([&] { hipLaunchKernelGGL((count_like<false, false>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out); }));
        report("count_like compute + flush", time_ms([&] { hipLaunchKernelGGL((count_like<false, true>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out); }));
        report("count_like load + compute", time_ms([&] { hipLaunchKernelGGL((count_like<true, false>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out); }));
        report("count_like load+comp+flush", time_ms([&] { hipLaunchKernelGGL((count_like<true, true>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out); }));
        report("count_like cached loads+comp", time_ms([&] { hipLaunchKernelGGL((count_like<true, false, false, false, true>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out, spec, pwp, panes); }));
        report("count_like cached +comp+flush", time_ms([&] { hipLaunchKernelGGL((count_like<true, true, false, false, true>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out, spec, pwp, panes); }));
        report("  + spec_info check", time_ms([&] { hipLaunchKernelGGL((count_like<true, true, true, false>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out, spec, pwp, panes); }));
        report("  + pane loads first", time_ms([&] { hipLaunchKernelGGL((count_like<true, true, false, true>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out, spec, pwp, panes); }));
        report("  + both", time_ms([&] { hipLaunchKernelGGL((count_like<true, true, true, true>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out, spec, pwp, panes); }));
        report("front (desc + min/max)", time_ms([&] { hipLaunchKernelGGL((front<false, false>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out); }));
        report("front + LDS histogram", time_ms([&] { hipLaunchKernelGGL((front<true, false>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out); }));
        report("front + hist + flush", time_ms([&] { hipLaunchKernelGGL((front<true, true>), dim3((unsigned)n_tiles), dim3(256), 0, 0, a, tiles, counters, out); }));
        ONE(256, 8, true);
        ONE(256, 8, false);
        ONE(256, 4, true);
        ONE(256, 16, true);
        ONE(512, 8, true);
        ONE(512, 4, true);
        ONE(1024, 4, true);
        ONE(1024, 8, true);
        ONE(128, 8, true);
        ONE(64, 8, true);
        ONE(64, 16, true);
#define PER(W, IT, K) report("persist<" #W "," #IT "> x" #K, time_ms([&] { hipLaunchKernelGGL((persist<W, IT>), dim3((unsigned)(cus * K)), dim3(W), 0, 0, a, n / (W * 4 * IT), out); }))
        PER(256, 8, 8);
        PER(256, 8, 12);
        PER(256, 8, 16);
        PER(256, 4, 16);
        PER(512, 8, 4);
        PER(1024, 4, 2);
    }
    return 0;
}
