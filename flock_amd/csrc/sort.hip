// Stable LSD radix sort of (int32 key, uint32 value) pairs for gfx950 -- the "group the rows of every key together,
// arrival order kept" step of the session windows (sort.hpp).  HBM-bound integer work, no MFMA.
//
// One pass = one digit of up to 8 bits, count -> scan -> emit (no workgroup waits on another):
//   count : 4096-row tile -> LDS digit histogram -> hist[digit][tile]               (reads 4 B / row)
//   scan  : inclusive scan of the digit-major matrix (gather.hip)                   -> first output slot of every
//           (digit, tile) run
//   emit  : wave w of a tile owns rows [1024 w, 1024 (w+1)), lane = row % 64, so (wave, iteration, lane) IS arrival
//           order.  Per iteration the lanes holding the same digit find each other with one ballot per digit bit
//           (`m &= ~(ballot ^ my bit)`), the lowest lane of a group advances the wave's digit counter in LDS, and
//           rank = counter before + lanes of the group below me.  Keys and values are then regrouped by digit in LDS
//           and leave the tile as runs of consecutive addresses: reads 8 B / row, writes 8 B / row.
// The pass count adapts to the key range: ceil(bits / 8) passes of equal width.
#include "sort.hpp"

using namespace flockgpu;

namespace {

constexpr int kSortItems = 16;
constexpr int kSortTile = kBlock * kSortItems;      // 4096 rows
constexpr int kWaveRows = kSortTile / kWavesPerBlock;  // 1024 consecutive rows per wave
constexpr int kMaxDigits = 256;

__device__ __forceinline__ uint32_t digit_of(int32_t key, int32_t bias, int shift, uint32_t mask) {
    return (((uint32_t)key - (uint32_t)bias) >> shift) & mask;
}

__global__ __launch_bounds__(kBlock) void sort_count_kernel(const int32_t *__restrict__ keys, int64_t n, int32_t bias, int shift,
                                                            uint32_t mask, int32_t n_tiles, int32_t *__restrict__ hist) {
    __shared__ uint32_t s_h[kMaxDigits];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t tile_begin = (int64_t)blockIdx.x * kSortTile;
    int32_t k[kSortItems];
#pragma unroll
    for (int it = 0; it < kSortItems / 4; ++it) {
        const int64_t r0 = tile_begin + (int64_t)(it * kBlock + threadIdx.x) * 4;
        if (r0 + 4 <= n) {
            const int4 t = *reinterpret_cast<const int4 *>(keys + r0);
            k[it * 4] = t.x; k[it * 4 + 1] = t.y; k[it * 4 + 2] = t.z; k[it * 4 + 3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) k[it * 4 + j] = r0 + j < n ? keys[r0 + j] : 0;
        }
    }
#pragma unroll
    for (int it = 0; it < kSortItems / 4; ++it) {
        const int64_t r0 = tile_begin + (int64_t)(it * kBlock + threadIdx.x) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // a hot key puts the same digit in many lanes: the lanes that share the first lane's digit add once
            const bool valid = r0 + j < n;
            const uint32_t d = digit_of(k[it * 4 + j], bias, shift, mask);
            const uint32_t hot = __builtin_amdgcn_readfirstlane(d);
            const uint64_t b = __ballot(valid && d == hot);
            if (valid && d == hot) {
                if (mbcnt(b) == 0) atomicAdd(&s_h[hot], (uint32_t)__popcll((unsigned long long)b));
            } else if (valid) {
                atomicAdd(&s_h[d], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x <= mask) hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = (int32_t)s_h[threadIdx.x];
}

template <int kBits>  // width of this pass's digit
__global__ __launch_bounds__(kBlock) void sort_emit_kernel(const int32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                           int64_t n, int32_t bias, int shift, uint32_t mask, int32_t n_tiles,
                                                           const int32_t *__restrict__ hist_incl, int32_t *__restrict__ keys_out,
                                                           uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t s_wh[kWavesPerBlock][kMaxDigits];  // running digit counts of a wave, then its base inside the digit
    __shared__ uint32_t s_dig_off[kMaxDigits];             // tile-local position of the digit's first row
    __shared__ uint32_t s_glob[kMaxDigits];                // output position of the digit's first row of this tile
    __shared__ uint32_t s_wave_total[kWavesPerBlock];
    __shared__ int32_t s_keys[kSortTile];
    __shared__ uint32_t s_vals[kSortTile];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) s_wh[w][threadIdx.x] = 0;
    // (n < 2^31: row numbers in 32 bits -- a 64-bit clamp and address per load was a tenth of this kernel's VALU work)
    const uint32_t n32 = (uint32_t)n;
    const uint32_t tile_begin = blockIdx.x * (uint32_t)kSortTile;
    const uint32_t wave_begin = tile_begin + (uint32_t)wave * kWaveRows;
    int32_t k[kSortItems];
    uint32_t v[kSortItems], rank[kSortItems];
    if (vals_in) {
#pragma unroll
        for (int it = 0; it < kSortItems; ++it) {
            const uint32_t rc = min(wave_begin + it * 64 + lane, n32 - 1);  // clamped: no load under a per-row branch
            k[it] = keys_in[rc];
            v[it] = vals_in[rc];
        }
    } else {
#pragma unroll
        for (int it = 0; it < kSortItems; ++it) {
            const uint32_t rc = min(wave_begin + it * 64 + lane, n32 - 1);
            k[it] = keys_in[rc];
            v[it] = rc;
        }
    }
    __syncthreads();
    uint32_t *wh = s_wh[wave];  // this wave's running digit counts: written and read by this wave only, in program order
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        const bool valid = wave_begin + it * 64 + lane < n32;
        const uint32_t d = digit_of(k[it], bias, shift, mask);
        // the lanes that hold my digit: per digit bit, keep the lanes whose bit equals mine.  With t = 0 / ~0 for my bit the
        // lanes to keep are ~(ballot ^ t): one three-input boolean op per half of the mask (four VALU instructions a round,
        // where `m &= bit ? ballot : ~ballot` on a 64-bit mask compiled to nine -- this kernel is VALU-bound, not HBM-bound)
        const uint64_t live = __ballot(valid);
        uint32_t m_lo = (uint32_t)live, m_hi = (uint32_t)(live >> 32);
#pragma unroll
        for (int b = 0; b < kBits; ++b) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int32_t)d, (uint32_t)b, 1u);  // 0 or ~0: my bit b, sign-extended
            const uint64_t bal = __ballot(t != 0u);
            m_lo &= ~((uint32_t)bal ^ t);
            m_hi &= ~((uint32_t)(bal >> 32) ^ t);
        }
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
        const uint32_t before = __hip_atomic_load(&wh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        rank[it] = before + below;
        if (valid && below == 0)
            __hip_atomic_store(&wh[d], before + (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __syncthreads();
    {   // thread d: wave bases of digit d, tile-local and global start of the digit
        const uint32_t d = threadIdx.x;
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) {
            const uint32_t c = s_wh[w][d];
            s_wh[w][d] = total;
            total += c;
        }
        const uint32_t incl = wave_incl_scan_u32(total);
        if (lane == 63) s_wave_total[wave] = incl;
        __syncthreads();
        uint32_t off = incl - total;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) off += w < wave ? s_wave_total[w] : 0u;
        s_dig_off[d] = off;
        s_glob[d] = d <= mask ? (uint32_t)hist_incl[(size_t)d * n_tiles + blockIdx.x] - total : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        if (wave_begin + it * 64 + lane < n32) {
            const uint32_t d = digit_of(k[it], bias, shift, mask);
            const uint32_t lp = s_dig_off[d] + s_wh[wave][d] + rank[it];
            s_keys[lp] = k[it];
            s_vals[lp] = v[it];
        }
    }
    __syncthreads();
    const int32_t tile_n = (int32_t)min(n32 - tile_begin, (uint32_t)kSortTile);
    for (int32_t j = threadIdx.x; j < tile_n; j += kBlock) {
        const int32_t key = s_keys[j];
        const uint32_t d = digit_of(key, bias, shift, mask);
        const uint32_t pos = s_glob[d] + ((uint32_t)j - s_dig_off[d]);
        keys_out[pos] = key;
        vals_out[pos] = s_vals[j];
    }
}

__global__ __launch_bounds__(kBlock) void key_min_max_kernel(const int32_t *__restrict__ keys, int64_t n, int32_t *minmax) {
    __shared__ int32_t s_red[2 * kWavesPerBlock];
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
    for (int64_t r0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4; r0 < n; r0 += (int64_t)gridDim.x * kBlock * 4) {
        if (r0 + 4 <= n) {
            const int4 t = *reinterpret_cast<const int4 *>(keys + r0);
            mn = min(mn, min(min(t.x, t.y), min(t.z, t.w)));
            mx = max(mx, max(max(t.x, t.y), max(t.z, t.w)));
        } else {
            for (int64_t r = r0; r < n; ++r) {
                mn = min(mn, keys[r]);
                mx = max(mx, keys[r]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (lane_id() == 0) {
        s_red[threadIdx.x >> 6] = mn;
        s_red[kWavesPerBlock + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            mn = min(mn, s_red[w]);
            mx = max(mx, s_red[kWavesPerBlock + w]);
        }
        atomicMin(&minmax[0], mn);
        atomicMax(&minmax[1], mx);
    }
}

__global__ __launch_bounds__(kBlock) void sorted_key_offsets_kernel(const int32_t *__restrict__ sorted_key, int64_t n, int32_t n_keys,
                                                                    int64_t *__restrict__ off) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i > n) return;
    const int32_t prev = i > 0 ? sorted_key[i - 1] : -1;
    const int32_t cur = i < n ? sorted_key[i] : n_keys;
    for (int32_t k = prev + 1; k <= cur && k <= n_keys; ++k) off[k] = i;
}

__global__ void init_min_max_kernel(int32_t *minmax) {
    minmax[0] = 0x7fffffff;
    minmax[1] = (int32_t)0x80000000;
}

}  // namespace

namespace flockgpu {

int key_min_max(flockgpu_ctx *ctx, const int32_t *keys, int64_t n, int32_t *d_minmax) {
    hipLaunchKernelGGL(init_min_max_kernel, dim3(1), dim3(1), 0, ctx->stream, d_minmax);
    FG_TRY(check_launch(ctx, "init_min_max_kernel"));
    if (n <= 0) return FLOCKGPU_OK;
    if (reinterpret_cast<uintptr_t>(keys) & 15) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "sort: key column must be 16-byte aligned");
    const unsigned grid = (unsigned)std::min<int64_t>(div_up(n, (int64_t)kBlock * 4), (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "key_min_max_kernel");
        hipLaunchKernelGGL(key_min_max_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, keys, n, d_minmax);
    }
    return check_launch(ctx, "key_min_max_kernel");
}

int sorted_key_offsets(flockgpu_ctx *ctx, const int32_t *sorted_key, int64_t n, int32_t n_keys, int64_t *d_off) {
    {
        LaunchScope ls(ctx, "sorted_key_offsets_kernel");
        hipLaunchKernelGGL(sorted_key_offsets_kernel, dim3((unsigned)div_up(n + 1, kBlock)), dim3(kBlock), 0, ctx->stream, sorted_key, n,
                           n_keys, d_off);
    }
    return check_launch(ctx, "sorted_key_offsets_kernel");
}

int radix_sort_pairs(flockgpu_ctx *ctx, const char *name, const int32_t *keys, const uint32_t *vals, int64_t n, int32_t bias,
                     int bits, int32_t **out_keys, uint32_t **out_vals) {
    if (n >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: relations are limited to 2^31 rows per call", name);
    if (bits < 1) bits = 1;
    if (bits > 32) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: %d key bits", name, bits);
    if (n > 0 && (reinterpret_cast<uintptr_t>(keys) & 15))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: key column must be 16-byte aligned", name);
    const std::string base(name);
    int32_t *kb[2] = {nullptr, nullptr};
    uint32_t *vb[2] = {nullptr, nullptr};
    const size_t cap = (size_t)std::max<int64_t>(n, 1) + 4;
    FG_TRY(arena_get_t(ctx, (base + ".k0").c_str(), cap, &kb[0]));
    FG_TRY(arena_get_t(ctx, (base + ".v0").c_str(), cap, &vb[0]));
    const int passes = (bits + 7) / 8, width = (bits + passes - 1) / passes;
    if (passes > 1) {
        FG_TRY(arena_get_t(ctx, (base + ".k1").c_str(), cap, &kb[1]));
        FG_TRY(arena_get_t(ctx, (base + ".v1").c_str(), cap, &vb[1]));
    }
    // the last pass lands in buffer 0
    int dst = (passes - 1) & 1;
    *out_keys = kb[0];
    *out_vals = vb[0];
    if (n <= 0) return FLOCKGPU_OK;
    const int64_t tiles = div_up(n, kSortTile);
    int32_t *hist = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".hist").c_str(), (size_t)tiles << width, &hist));
    const int32_t *k_in = keys;
    const uint32_t *v_in = vals;
    for (int p = 0; p < passes; ++p, dst ^= 1) {
        const int shift = p * width, nb = std::min(width, bits - shift);
        const uint32_t mask = (1u << nb) - 1;
        {
            LaunchScope ls(ctx, "sort_count_kernel");
            hipLaunchKernelGGL(sort_count_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, k_in, n, bias, shift, mask,
                               (int32_t)tiles, hist);
        }
        FG_TRY(check_launch(ctx, "sort_count_kernel"));
        FG_TRY(inclusive_scan_i32(ctx, (base + ".scan").c_str(), hist, tiles << nb));
        {
            LaunchScope ls(ctx, "sort_emit_kernel");
            static constexpr decltype(&sort_emit_kernel<8>) kEmit[8] = {sort_emit_kernel<1>, sort_emit_kernel<2>, sort_emit_kernel<3>, sort_emit_kernel<4>,
                                                                        sort_emit_kernel<5>, sort_emit_kernel<6>, sort_emit_kernel<7>, sort_emit_kernel<8>};
            hipLaunchKernelGGL(kEmit[nb - 1], dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, k_in, v_in, n, bias, shift, mask,
                               (int32_t)tiles, hist, kb[dst], vb[dst]);
        }
        FG_TRY(check_launch(ctx, "sort_emit_kernel"));
        k_in = kb[dst];
        v_in = vb[dst];
    }
    return FLOCKGPU_OK;
}

}  // namespace flockgpu

extern "C" {

int flockgpu_group_rows_by_key(flockgpu_ctx *ctx, const int32_t *keys, int64_t rows, int32_t **out_keys, int32_t **out_rows) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!out_keys || !out_rows || rows < 0 || (rows > 0 && !keys)) return fail(ctx, FLOCKGPU_ERR_INVALID, "group_rows: null argument");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    int32_t *d_mm = nullptr, *h_mm = nullptr;
    FG_TRY(arena_get_t(ctx, "group_rows.minmax", 4, &d_mm));
    FG_TRY(pinned_get_t(ctx, "group_rows.minmax", 4, &h_mm));
    FG_TRY(key_min_max(ctx, keys, rows, d_mm));
    FG_TRY(publish_words(ctx, PublishList().add(h_mm, d_mm, 2)));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int bits = 1;
    int32_t bias = 0;
    if (rows > 0) {
        bias = h_mm[0];
        const uint64_t span = (uint64_t)((int64_t)h_mm[1] - (int64_t)h_mm[0]);
        while (bits < 32 && (span >> bits)) ++bits;
    }
    uint32_t *v = nullptr;
    FG_TRY(radix_sort_pairs(ctx, "group_rows", keys, nullptr, rows, bias, bits, out_keys, &v));
    *out_rows = reinterpret_cast<int32_t *>(v);
    return FLOCKGPU_OK;
}

}  // extern "C"
