// Device-side Yahoo Streaming Benchmark source: the two relations ysb.sql scans, straight into HBM.
// Restates flock/src/datasource/ysb/generator.rs:38-101 (campaign map: `campaigns` x `ads` UUID pairs; every event names a
// random ad and one of three event types) with the deviation class D1 of DESIGN.md: the reference draws UUIDs from the OS
// RNG (Uuid::new_v4) and picks ads through HashMap iteration order, i.e. its stream is not reproducible by construction;
// here every value is a pure function of (seed, index) so that the CPU oracle regenerates the same bytes.
//   uuid(tag, i) : hi = mix64(mix64(seed ^ tag * C1) + (i + 1) * C2), lo = mix64(hi ^ C3), RFC 4122 version-4 bits set,
//                  printed 8-4-4-4-12 in lower-case hex (36 bytes)
//   campaign row i : c_ad_id = uuid(1, i), campaign_id = uuid(2, i / ads)
//   event n        : ad_id = c_ad_id of row uni(draw(n, 0), campaigns * ads); event_type = {view, click, purchase}[uni(draw(n, 1), 3)]
#include "gather.hpp"

using namespace flockgpu;

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ uint64_t uni(uint64_t r, uint64_t n) { return ((r >> 32) * n) >> 32; }
__device__ __forceinline__ uint64_t draw(uint64_t seed, uint64_t n, uint32_t k) {
    return mix64(mix64(seed ^ (n * 0xD6E8FEB86659FD93ull)) + (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull);
}

__device__ __forceinline__ void write_uuid(uint64_t seed, uint64_t tag, uint64_t i, uint8_t *dst) {
    uint64_t hi = mix64(mix64(seed ^ (tag * 0xA24BAED4963EE407ull)) + (i + 1) * 0x9FB21C651E98DF25ull);
    uint64_t lo = mix64(hi ^ 0xC2B2AE3D27D4EB4Full);
    hi = (hi & ~0xF000ull) | 0x4000ull;                         // version 4
    lo = (lo & ~(3ull << 62)) | (2ull << 62);                   // variant 10
    int o = 0;
#pragma unroll
    for (int nib = 0; nib < 32; ++nib) {
        if (nib == 8 || nib == 12 || nib == 16 || nib == 20) dst[o++] = '-';
        const uint64_t v = nib < 16 ? hi : lo;
        const uint32_t d = (uint32_t)(v >> (60 - 4 * (nib & 15))) & 15u;
        dst[o++] = (uint8_t)(d < 10 ? '0' + d : 'a' + d - 10);
    }
}

__global__ __launch_bounds__(kBlock) void ysb_gen_campaigns_kernel(uint64_t seed, int64_t rows, int64_t ads, int32_t *ad_off,
                                                                   uint8_t *ad_bytes, int32_t *camp_off, uint8_t *camp_bytes) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        ad_off[0] = 0;
        camp_off[0] = 0;
    }
    if (i >= rows) return;
    ad_off[i + 1] = (int32_t)(36 * (i + 1));
    camp_off[i + 1] = (int32_t)(36 * (i + 1));
    write_uuid(seed, 1, (uint64_t)i, ad_bytes + 36 * i);
    write_uuid(seed, 2, (uint64_t)(i / ads), camp_bytes + 36 * i);
}

__device__ __constant__ char c_event_types[3][9] = {"view", "click", "purchase"};
__device__ __constant__ int32_t c_event_len[3] = {4, 5, 8};

// pass 1: ad_id (fixed width) + event_type lengths at et_off[j + 1]; pass 2 (after the scan): event_type bytes
__global__ __launch_bounds__(kBlock) void ysb_gen_events_kernel(uint64_t seed, uint64_t first, int64_t rows, uint64_t n_ads,
                                                                int32_t *ad_off, uint8_t *ad_bytes, int32_t *et_off) {
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < rows; j += (int64_t)gridDim.x * kBlock) {
        const uint64_t n = first + (uint64_t)j;
        if (j == 0) {
            ad_off[0] = 0;
            et_off[0] = 0;
        }
        ad_off[j + 1] = (int32_t)(36 * (j + 1));
        write_uuid(seed, 1, uni(draw(seed, n, 0), n_ads), ad_bytes + 36 * j);
        et_off[j + 1] = c_event_len[uni(draw(seed, n, 1), 3)];
    }
}

__global__ __launch_bounds__(kBlock) void ysb_gen_event_types_kernel(uint64_t seed, uint64_t first, int64_t rows,
                                                                     const int32_t *__restrict__ et_off, uint8_t *et_bytes) {
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < rows; j += (int64_t)gridDim.x * kBlock) {
        const uint32_t t = (uint32_t)uni(draw(seed, first + (uint64_t)j, 1), 3);
        uint8_t *dst = et_bytes + et_off[j];
        for (int c = 0; c < c_event_len[t]; ++c) dst[c] = (uint8_t)c_event_types[t][c];
    }
}

}  // namespace

extern "C" {

int flockgpu_ysb_gen_campaigns(flockgpu_ctx *ctx, uint64_t seed, int64_t n_campaigns, int64_t ads, int32_t *c_ad_id_off,
                               uint8_t *c_ad_id_bytes, int32_t *campaign_off, uint8_t *campaign_bytes) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (n_campaigns < 0 || ads < 1 || !c_ad_id_off || !c_ad_id_bytes || !campaign_off || !campaign_bytes)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "ysb_gen_campaigns: bad argument");
    const int64_t rows = n_campaigns * ads;
    if (rows * 36 >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "ysb_gen_campaigns: more than 2^31 bytes of ids");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    {
        LaunchScope ls(ctx, "ysb_gen_campaigns_kernel");
        hipLaunchKernelGGL(ysb_gen_campaigns_kernel, dim3((unsigned)std::max<int64_t>(1, div_up(rows, kBlock))), dim3(kBlock), 0,
                           ctx->stream, seed, rows, ads, c_ad_id_off, c_ad_id_bytes, campaign_off, campaign_bytes);
    }
    return check_launch(ctx, "ysb_gen_campaigns_kernel");
}

int flockgpu_ysb_gen_events(flockgpu_ctx *ctx, uint64_t seed, uint64_t first_event, int64_t n_events, int64_t n_ads,
                            int32_t *ad_id_off, uint8_t *ad_id_bytes, int32_t *event_type_off, uint8_t *event_type_bytes) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (n_events < 0 || n_ads < 1 || !ad_id_off || !ad_id_bytes || !event_type_off || !event_type_bytes)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "ysb_gen_events: bad argument");
    if (n_events * 36 >= (int64_t(1) << 31))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "ysb_gen_events: more than 2^31 bytes of ad ids per call (Arrow Utf8 offsets are int32)");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    if (n_events == 0) {
        FG_HIP(ctx, hipMemsetAsync(ad_id_off, 0, sizeof(int32_t), ctx->stream));
        FG_HIP(ctx, hipMemsetAsync(event_type_off, 0, sizeof(int32_t), ctx->stream));
        return FLOCKGPU_OK;
    }
    const unsigned grid = (unsigned)std::min<int64_t>(div_up(n_events, kBlock), (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "ysb_gen_events_kernel");
        hipLaunchKernelGGL(ysb_gen_events_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, seed, first_event, n_events, (uint64_t)n_ads,
                           ad_id_off, ad_id_bytes, event_type_off);
    }
    FG_TRY(check_launch(ctx, "ysb_gen_events_kernel"));
    FG_TRY(inclusive_scan_i32(ctx, "ysb.gen.scan", event_type_off + 1, n_events));
    {
        LaunchScope ls(ctx, "ysb_gen_event_types_kernel");
        hipLaunchKernelGGL(ysb_gen_event_types_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, seed, first_event, n_events,
                           event_type_off, event_type_bytes);
    }
    return check_launch(ctx, "ysb_gen_event_types_kernel");
}

}  // extern "C"
