// Device-side take(): fixed-width gather and Utf8 gather (lengths -> chained scan -> byte copy).
#include "gather.hpp"

using namespace flockgpu;

namespace {

constexpr int kLenItems = 8;
constexpr int kLenTile = kBlock * kLenItems;  // 2048 rows per workgroup

__global__ __launch_bounds__(kBlock) void gather_i32_kernel(const int32_t *__restrict__ src,
                                                            const int32_t *__restrict__ rows, int64_t n,
                                                            int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = src[rows[i]];
}

// out_off[i + 1] = sum_{k <= i} len(rows[k]);  out_off[0] = 0.  Single pass chained scan.
__global__ __launch_bounds__(kBlock) void utf8_offsets_kernel(const int32_t *__restrict__ src_off,
                                                              const int32_t *__restrict__ rows, int64_t n,
                                                              uint64_t *status, uint32_t *err, int32_t n_tiles,
                                                              int32_t *__restrict__ out_off, uint64_t *total) {
    __shared__ uint64_t s_scan[2 * kWavesPerBlock];
    StripedScan sc;
#pragma unroll 1
    for (int32_t tile = (int32_t)blockIdx.x; tile < n_tiles; tile += (int32_t)gridDim.x) {
    const int64_t i0 = (int64_t)tile * kLenTile + (int64_t)threadIdx.x * kLenItems;
    uint32_t len[kLenItems];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        len[k] = 0;
        if (i0 + k < n) {
            const int32_t r = rows[i0 + k];
            len[k] = (uint32_t)(src_off[r + 1] - src_off[r]);
        }
        mine += len[k];
    }
    const uint32_t incl = wave_incl_scan_u32(mine);
    const uint32_t wave_total = __shfl(incl, 63, 64);
    uint64_t tile_base, tile_total;
    uint64_t pos = block_striped_offset(status, sc, tile, wave_total, s_scan, &tile_base, &tile_total, err) + (incl - mine);
    if (tile == 0 && threadIdx.x == 0) out_off[0] = 0;
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        pos += len[k];
        if (i0 + k < n) out_off[i0 + k + 1] = (int32_t)pos;
    }
    if (threadIdx.x == 0 && tile == n_tiles - 1) *total = tile_base + tile_total;
    }  // tile loop
}

// One lane per output value; short strings (NEXMark names / cities / states are <= 14 bytes).
__global__ __launch_bounds__(kBlock) void utf8_copy_kernel(const int32_t *__restrict__ src_off,
                                                           const uint8_t *__restrict__ src, const int32_t *__restrict__ rows,
                                                           int64_t n, const int32_t *__restrict__ out_off,
                                                           uint8_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int32_t r = rows[i];
        const int32_t b = src_off[r], e = src_off[r + 1];
        uint8_t *dst = out + out_off[i];
        for (int32_t k = 0; k < e - b; ++k) dst[k] = src[b + k];
    }
}


__global__ __launch_bounds__(kBlock) void scan_i32_kernel(int32_t *data, int64_t n, uint64_t *status, uint32_t *err,
                                                          int32_t n_tiles) {
    __shared__ uint64_t s_scan[2 * kWavesPerBlock];
    StripedScan sc;
#pragma unroll 1
    for (int32_t tile = (int32_t)blockIdx.x; tile < n_tiles; tile += (int32_t)gridDim.x) {
    const int64_t i0 = (int64_t)tile * kLenTile + (int64_t)threadIdx.x * kLenItems;
    uint32_t v[kLenItems], mine = 0;
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        v[k] = (i0 + k < n) ? (uint32_t)data[i0 + k] : 0u;
        mine += v[k];
    }
    const uint32_t incl = wave_incl_scan_u32(mine);
    const uint32_t wave_total = __shfl(incl, 63, 64);
    uint64_t tile_base, tile_total;
    uint64_t pos = block_striped_offset(status, sc, tile, wave_total, s_scan, &tile_base, &tile_total, err) + (incl - mine);
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        pos += v[k];
        if (i0 + k < n) data[i0 + k] = (int32_t)pos;
    }
    }  // tile loop
}


__global__ __launch_bounds__(kScanBlock) void tile_scan_kernel(const uint32_t *__restrict__ counts, int32_t n_tiles,
                                                              uint64_t *__restrict__ tile_base,
                                                              const int32_t *__restrict__ tile_first, int32_t n_seg,
                                                              int64_t *__restrict__ seg_out_off) {
    __shared__ uint64_t s_wave[kScanBlock / 64];
    __shared__ uint64_t s_carry;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int32_t t0 = 0; t0 < n_tiles; t0 += kScanBlock) {
        const int32_t t = t0 + (int32_t)threadIdx.x;
        uint64_t c = 0;
        if (t < n_tiles) {
            const uint4 w = *reinterpret_cast<const uint4 *>(counts + (size_t)t * kWavesPerBlock);
            c = (uint64_t)w.x + w.y + w.z + w.w;
        }
        const uint64_t incl = wave_incl_scan_u64(c);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint64_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (t < n_tiles) tile_base[t] = before + incl - c;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_base[n_tiles] = s_carry;
    __syncthreads();  // the block reads back its own global stores below (same CU, write-through L1)
    if (seg_out_off)
        for (int32_t s = threadIdx.x; s <= n_seg; s += kScanBlock) {
            const int32_t t = tile_first[s];  // empty segments share the next segment's first tile
            seg_out_off[s] = (int64_t)(t >= n_tiles ? s_carry : __hip_atomic_load(&tile_base[t], __ATOMIC_RELAXED,
                                                                                  __HIP_MEMORY_SCOPE_AGENT));
        }
}

}  // namespace

namespace flockgpu {

int launch_tile_scan(flockgpu_ctx *ctx, const uint32_t *counts, int32_t n_tiles, uint64_t *tile_base,
                            const int32_t *tile_first, int32_t n_seg, int64_t *seg_out_off) {
    {
        LaunchScope ls(ctx, "tile_scan_kernel");
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(kScanBlock), 0, ctx->stream, counts, n_tiles, tile_base,
                           tile_first, n_seg, seg_out_off);
    }
    return check_launch(ctx, "tile_scan_kernel");
}


int inclusive_scan_i32(flockgpu_ctx *ctx, const char *name, int32_t *data, int64_t n) {
    if (n <= 0) return FLOCKGPU_OK;
    const int64_t tiles = div_up(n, kLenTile);
    uint64_t *status = nullptr;
    FG_TRY(arena_get_t(ctx, name, (size_t)tiles + 2, &status));
    FG_HIP(ctx, hipMemsetAsync(status, 0, sizeof(uint64_t) * ((size_t)tiles + 2), ctx->stream));
    {
        unsigned grid = 1;
        FG_TRY(persistent_grid(ctx, scan_i32_kernel, "scan_i32_kernel", tiles, &grid));
        LaunchScope ls(ctx, "scan_i32_kernel");
        hipLaunchKernelGGL(scan_i32_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, data, n, status,
                           reinterpret_cast<uint32_t *>(status + tiles), (int32_t)tiles);
    }
    return check_launch(ctx, "scan_i32_kernel");
}

int gather_i32(flockgpu_ctx *ctx, const int32_t *src, const int32_t *rows, int64_t n, int32_t *out) {
    if (n <= 0) return FLOCKGPU_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n, kBlock), (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "gather_i32_kernel");
        hipLaunchKernelGGL(gather_i32_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, src, rows, n, out);
    }
    return check_launch(ctx, "gather_i32_kernel");
}

int gather_utf8(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 &src, const int32_t *rows, int64_t n,
                flockgpu_utf8 *out, int64_t *n_bytes) {
    const std::string k_off = std::string(name) + ".off", k_bytes = std::string(name) + ".bytes",
                      k_st = std::string(name) + ".scan";
    int32_t *o_off = nullptr;
    FG_TRY(arena_get_t(ctx, k_off.c_str(), (size_t)n + 1, &o_off));
    out->offsets = o_off;
    out->data = nullptr;
    *n_bytes = 0;
    if (n <= 0) {
        FG_HIP(ctx, hipMemsetAsync(o_off, 0, sizeof(int32_t), ctx->stream));
        uint8_t *o_b = nullptr;
        FG_TRY(arena_get_t(ctx, k_bytes.c_str(), 16, &o_b));
        out->data = o_b;
        return FLOCKGPU_OK;
    }
    const int64_t tiles = div_up(n, kLenTile);
    uint64_t *status = nullptr;
    FG_TRY(arena_get_t(ctx, k_st.c_str(), (size_t)tiles + 3, &status));  // + error word + total
    FG_HIP(ctx, hipMemsetAsync(status, 0, sizeof(uint64_t) * ((size_t)tiles + 3), ctx->stream));
    uint64_t *d_total = status + tiles + 1;
    {
        unsigned grid = 1;
        FG_TRY(persistent_grid(ctx, utf8_offsets_kernel, "utf8_offsets_kernel", tiles, &grid));
        LaunchScope ls(ctx, "utf8_offsets_kernel");
        hipLaunchKernelGGL(utf8_offsets_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, src.offsets, rows, n, status,
                           reinterpret_cast<uint32_t *>(status + tiles), (int32_t)tiles, o_off, d_total);
    }
    FG_TRY(check_launch(ctx, "utf8_offsets_kernel"));
    uint64_t *h_words = nullptr;  // [0] = error word, [1] = total
    FG_TRY(pinned_get_t(ctx, "gather_utf8.words", 2, &h_words));
    FG_HIP(ctx, hipMemcpyAsync(h_words, status + tiles, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((uint32_t)h_words[0]) return fail(ctx, FLOCKGPU_ERR_HIP, "%s: chained scan stalled", name);
    const uint64_t h_total = h_words[1];
    if (h_total > 0x7fffffffull)
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: gathered Utf8 column exceeds 2^31 bytes (Arrow Utf8 offsets are int32)", name);
    uint8_t *o_b = nullptr;
    FG_TRY(arena_get_t(ctx, k_bytes.c_str(), (size_t)h_total + 16, &o_b));
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n, kBlock), (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "utf8_copy_kernel");
        hipLaunchKernelGGL(utf8_copy_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, src.offsets, src.data, rows, n,
                           o_off, o_b);
    }
    FG_TRY(check_launch(ctx, "utf8_copy_kernel"));
    out->data = o_b;
    *n_bytes = (int64_t)h_total;
    return FLOCKGPU_OK;
}

}  // namespace flockgpu
