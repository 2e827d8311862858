// Device-side NEXMark source: generates the columns the five plans scan straight into HBM.
// Restates flock/src/datasource/nexmark/event.rs:83-97,152-185,247-311,354-371 and the defaults of
// config.rs:121-157 with the documented deviations D1-D4: counter-based RNG
// draw(id, k) = mix64(mix64(seed ^ id*C1) + (k+1)*C2), integer fixed-point price, exact integer timestamps, one
// generator per stream.  Every row is computed from its event id alone, so rows map 1:1 to lanes (no scan
// needed for the fixed-width columns; person strings need one scan for the Utf8 offsets).
#include "gather.hpp"
#include "nexmark_exp2_table.h"

using namespace flockgpu;

namespace {

constexpr uint64_t kDenom = 50, kPersonProp = 1, kAuctionProp = 3, kBidProp = 46;
constexpr uint64_t kFirstAuctionId = 1000, kFirstPersonId = 1000, kFirstCategoryId = 10, kNumCategories = 5;
constexpr uint64_t kHotSellerRatio = 4, kHotAuctionRatio = 2, kHotBidderRatio = 4, kHotRatio2 = 100;
constexpr uint64_t kInFlightAuctions = 100, kActivePeople = 1000, kAuctionIdLead = 10, kPersonIdLead = 10;

__constant__ uint32_t c_exp2_q30[257] = NEXMARK_EXP2_TABLE_INIT;
// word lists of config.rs:145-157, fixed 16-byte cells: [len][bytes...]
__constant__ char c_states[6][4] = {"az", "ca", "id", "or", "wa", "wy"};
__constant__ char c_cities[10][16] = {"phoenix", "los angeles", "san francisco", "boise", "portland",
                                      "bend", "redmond", "seattle", "kent", "cheyenne"};
__constant__ uint8_t c_city_len[10] = {7, 11, 13, 5, 8, 4, 7, 7, 4, 8};
__constant__ char c_first[11][8] = {"peter", "paul", "luke", "john", "saul", "vicky", "kate", "julie", "sarah", "deiter", "walter"};
__constant__ uint8_t c_first_len[11] = {5, 4, 4, 4, 4, 5, 4, 5, 5, 6, 6};
__constant__ char c_last[9][8] = {"shultz", "abrams", "spencer", "white", "bartels", "walton", "smith", "jones", "noris"};
__constant__ uint8_t c_last_len[9] = {6, 6, 7, 5, 7, 6, 5, 5, 5};

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ uint64_t ev_base(uint64_t seed, uint64_t id) { return mix64(seed ^ (id * 0xD6E8FEB86659FD93ull)); }
__device__ __forceinline__ uint64_t draw(uint64_t base, uint32_t k) { return mix64(base + (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull); }
__device__ __forceinline__ uint64_t uni(uint64_t r, uint64_t n) { return ((r >> 32) * n) >> 32; }

__device__ __forceinline__ uint32_t price_from(uint64_t r) {
    const uint64_t F = 85605435163ull;  // floor(6 * log2(10) * 2^32)
    const uint64_t k = r >> 40;
    const uint64_t t = (k * F) >> 24;
    const uint32_t ip = (uint32_t)(t >> 32), fp = (uint32_t)t;
    const uint32_t idx = fp >> 24, rem = (fp >> 8) & 0xFFFFu;
    const uint64_t m = c_exp2_q30[idx] + ((((uint64_t)(c_exp2_q30[idx + 1] - c_exp2_q30[idx])) * rem) >> 16);
    return (uint32_t)((((100ull * m) << ip) + (1ull << 29)) >> 30);
}

__host__ __device__ __forceinline__ uint64_t person_last_id(uint64_t id) { return id / kDenom; }
__host__ __device__ __forceinline__ uint64_t auction_last_id(uint64_t id) {
    uint64_t epoch = id / kDenom, offset = id % kDenom;
    if (offset < kPersonProp) { epoch -= 1; offset = kAuctionProp - 1; }
    else if (offset >= kPersonProp + kAuctionProp) offset = kAuctionProp - 1;
    else offset -= kPersonProp;
    return epoch * kAuctionProp + offset;
}
__device__ __forceinline__ uint64_t person_next_id(uint64_t id, uint64_t r) {
    const uint64_t people = person_last_id(id) + 1;
    const uint64_t active = people < kActivePeople ? people : kActivePeople;
    return people - active + uni(r, active + kPersonIdLead);
}
__device__ __forceinline__ uint64_t auction_next_id(uint64_t id, uint64_t r) {
    const uint64_t max_a = auction_last_id(id);
    const uint64_t min_a = max_a < kInFlightAuctions ? 0 : max_a - kInFlightAuctions;
    return min_a + uni(r, max_a - min_a + 1 + kAuctionIdLead);
}

// number of events of each kind with id' < id
__host__ __device__ __forceinline__ uint64_t persons_before(uint64_t id) { return (id + kDenom - 1) / kDenom; }
__host__ __device__ __forceinline__ uint64_t auctions_before(uint64_t id) {
    const uint64_t o = id % kDenom;
    return (id / kDenom) * kAuctionProp + (o <= 1 ? 0 : (o - 1 < kAuctionProp ? o - 1 : kAuctionProp));
}
__host__ __device__ __forceinline__ uint64_t bids_before(uint64_t id) {
    const uint64_t o = id % kDenom;
    return (id / kDenom) * kBidProp + (o <= 4 ? 0 : o - 4);
}

__global__ __launch_bounds__(kBlock) void gen_bids_kernel(flockgpu_nexmark_stream s, uint64_t g0, uint64_t rows,
                                                          int32_t *__restrict__ auction, int32_t *__restrict__ bidder,
                                                          int32_t *__restrict__ price, int64_t *__restrict__ dt) {
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < rows; j += (uint64_t)gridDim.x * kBlock) {
        const uint64_t g = g0 + j;
        const uint64_t id = (g / kBidProp) * kDenom + 4 + g % kBidProp;
        const uint64_t base = ev_base(s.seed, id);
        if (auction) {
            const uint64_t a = uni(draw(base, 0), kHotAuctionRatio) > 0 ? (auction_last_id(id) / kHotRatio2) * kHotRatio2
                                                                        : auction_next_id(id, draw(base, 1));
            auction[j] = (int32_t)(a + kFirstAuctionId);
        }
        if (bidder) {
            const uint64_t b = uni(draw(base, 2), kHotBidderRatio) > 0 ? (person_last_id(id) / kHotRatio2) * kHotRatio2 + 1
                                                                       : person_next_id(id, draw(base, 3));
            bidder[j] = (int32_t)(b + kFirstPersonId);
        }
        if (price) price[j] = (int32_t)price_from(draw(base, 4));
        if (dt) dt[j] = (int64_t)(s.base_time + (id * 1000ull) / s.eps);
    }
}

__global__ __launch_bounds__(kBlock) void gen_auctions_kernel(flockgpu_nexmark_stream s, uint64_t g0, uint64_t rows,
                                                              int32_t *__restrict__ a_id, int32_t *__restrict__ seller,
                                                              int32_t *__restrict__ category) {
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < rows; j += (uint64_t)gridDim.x * kBlock) {
        const uint64_t g = g0 + j;
        const uint64_t id = (g / kAuctionProp) * kDenom + 1 + g % kAuctionProp;
        const uint64_t base = ev_base(s.seed, id);
        if (a_id) a_id[j] = (int32_t)(auction_last_id(id) + kFirstAuctionId);
        if (seller) {
            const uint64_t sel = uni(draw(base, 1), kHotSellerRatio) > 0 ? (person_last_id(id) / kHotRatio2) * kHotRatio2
                                                                         : person_next_id(id, draw(base, 2));
            seller[j] = (int32_t)(sel + kFirstPersonId);
        }
        if (category) category[j] = (int32_t)(kFirstCategoryId + uni(draw(base, 6), kNumCategories));
    }
}

// a_date_time / expires of the auctions (event.rs:297-310 next_length, restated like oracle/nexmark_gen.c)
__global__ __launch_bounds__(kBlock) void gen_auction_times_kernel(flockgpu_nexmark_stream s, uint64_t g0, uint64_t rows,
                                                                   int64_t *__restrict__ a_date_time, int64_t *__restrict__ expires) {
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < rows; j += (uint64_t)gridDim.x * kBlock) {
        const uint64_t g = g0 + j;
        const uint64_t id = (g / kAuctionProp) * kDenom + 1 + g % kAuctionProp;
        const uint64_t base = ev_base(s.seed, id);
        const uint64_t time = s.base_time + (id * 1000ull) / s.eps;
        const uint64_t events_for_auctions = (kInFlightAuctions * kDenom) / kAuctionProp;
        const uint64_t horizon = s.base_time + ((id + events_for_auctions) * 1000ull) / s.eps - time;
        const uint64_t span = horizon * 2 > 1 ? horizon * 2 : 1;
        if (a_date_time) a_date_time[j] = (int64_t)time;
        if (expires) expires[j] = (int64_t)(time + 1 + uni(draw(base, 5), span));
    }
}

// pass 1: ids + string lengths (written at offsets[j + 1]); pass 2 (after the scans): bytes
__global__ __launch_bounds__(kBlock) void gen_persons_len_kernel(flockgpu_nexmark_stream s, uint64_t g0, uint64_t rows,
                                                                 int32_t *__restrict__ p_id, int32_t *__restrict__ name_off,
                                                                 int32_t *__restrict__ city_off, int32_t *__restrict__ state_off) {
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < rows; j += (uint64_t)gridDim.x * kBlock) {
        const uint64_t id = (g0 + j) * kDenom;
        const uint64_t base = ev_base(s.seed, id);
        if (p_id) p_id[j] = (int32_t)(person_last_id(id) + kFirstPersonId);
        if (j == 0) {
            if (name_off) name_off[0] = 0;
            if (city_off) city_off[0] = 0;
            if (state_off) state_off[0] = 0;
        }
        if (name_off) name_off[j + 1] = c_first_len[uni(draw(base, 0), 11)] + 1 + c_last_len[uni(draw(base, 1), 9)];
        if (city_off) city_off[j + 1] = c_city_len[uni(draw(base, 2), 10)];
        if (state_off) state_off[j + 1] = 2 * (int32_t)(j + 1);  // fixed width: offsets are closed form
    }
}

__global__ __launch_bounds__(kBlock) void gen_persons_bytes_kernel(flockgpu_nexmark_stream s, uint64_t g0, uint64_t rows,
                                                                   const int32_t *__restrict__ name_off, uint8_t *__restrict__ name,
                                                                   const int32_t *__restrict__ city_off, uint8_t *__restrict__ city,
                                                                   uint8_t *__restrict__ state) {
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < rows; j += (uint64_t)gridDim.x * kBlock) {
        const uint64_t id = (g0 + j) * kDenom;
        const uint64_t base = ev_base(s.seed, id);
        if (name) {
            const uint32_t f = (uint32_t)uni(draw(base, 0), 11), l = (uint32_t)uni(draw(base, 1), 9);
            uint8_t *dst = name + name_off[j];
            const uint32_t fl = c_first_len[f], ll = c_last_len[l];
            for (uint32_t k = 0; k < fl; ++k) dst[k] = (uint8_t)c_first[f][k];
            dst[fl] = ' ';
            for (uint32_t k = 0; k < ll; ++k) dst[fl + 1 + k] = (uint8_t)c_last[l][k];
        }
        if (city) {
            const uint32_t c = (uint32_t)uni(draw(base, 2), 10);
            uint8_t *dst = city + city_off[j];
            for (uint32_t k = 0; k < c_city_len[c]; ++k) dst[k] = (uint8_t)c_cities[c][k];
        }
        if (state) {
            const uint32_t st = (uint32_t)uni(draw(base, 3), 6);
            state[2 * j] = (uint8_t)c_states[st][0];
            state[2 * j + 1] = (uint8_t)c_states[st][1];
        }
    }
}

inline unsigned grid_for(flockgpu_ctx *ctx, uint64_t rows) {
    const uint64_t b = (rows + kBlock - 1) / kBlock, cap = (uint64_t)ctx->num_cus * 16;
    return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

}  // namespace

extern "C" {

int flockgpu_nexmark_counts(const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1, uint64_t *n_person,
                            uint64_t *n_auction, uint64_t *n_bid) {
    if (!s || n1 < n0) return FLOCKGPU_ERR_INVALID;
    const uint64_t a = s->first_event_id + n0, b = s->first_event_id + n1;
    if (n_person) *n_person = persons_before(b) - persons_before(a);
    if (n_auction) *n_auction = auctions_before(b) - auctions_before(a);
    if (n_bid) *n_bid = bids_before(b) - bids_before(a);
    return FLOCKGPU_OK;
}

int flockgpu_nexmark_gen_bids(flockgpu_ctx *ctx, const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1, int32_t *auction,
                              int32_t *bidder, int32_t *price, int64_t *b_date_time) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!s || n1 < n0 || s->eps == 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "gen_bids: bad stream / range");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t g0 = bids_before(s->first_event_id + n0), rows = bids_before(s->first_event_id + n1) - g0;
    if (rows == 0) return FLOCKGPU_OK;
    {
        LaunchScope ls(ctx, "gen_bids_kernel");
        hipLaunchKernelGGL(gen_bids_kernel, dim3(grid_for(ctx, rows)), dim3(kBlock), 0, ctx->stream, *s, g0, rows, auction,
                           bidder, price, b_date_time);
    }
    return check_launch(ctx, "gen_bids_kernel");
}

int flockgpu_nexmark_gen_auctions(flockgpu_ctx *ctx, const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1,
                                  int32_t *a_id, int32_t *seller, int32_t *category) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!s || n1 < n0 || s->eps == 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "gen_auctions: bad stream / range");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t g0 = auctions_before(s->first_event_id + n0), rows = auctions_before(s->first_event_id + n1) - g0;
    if (rows == 0) return FLOCKGPU_OK;
    {
        LaunchScope ls(ctx, "gen_auctions_kernel");
        hipLaunchKernelGGL(gen_auctions_kernel, dim3(grid_for(ctx, rows)), dim3(kBlock), 0, ctx->stream, *s, g0, rows, a_id,
                           seller, category);
    }
    return check_launch(ctx, "gen_auctions_kernel");
}

int flockgpu_nexmark_gen_auction_times(flockgpu_ctx *ctx, const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1,
                                       int64_t *a_date_time, int64_t *expires) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!s || n1 < n0 || s->eps == 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "gen_auction_times: bad stream / range");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t g0 = auctions_before(s->first_event_id + n0), rows = auctions_before(s->first_event_id + n1) - g0;
    if (rows == 0) return FLOCKGPU_OK;
    {
        LaunchScope ls(ctx, "gen_auction_times_kernel");
        hipLaunchKernelGGL(gen_auction_times_kernel, dim3(grid_for(ctx, rows)), dim3(kBlock), 0, ctx->stream, *s, g0, rows,
                           a_date_time, expires);
    }
    return check_launch(ctx, "gen_auction_times_kernel");
}

int flockgpu_nexmark_gen_persons(flockgpu_ctx *ctx, const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1, int32_t *p_id,
                                 int32_t *name_off, uint8_t *name_bytes, int32_t *city_off, uint8_t *city_bytes,
                                 int32_t *state_off, uint8_t *state_bytes) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!s || n1 < n0 || s->eps == 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "gen_persons: bad stream / range");
    if ((name_bytes && !name_off) || (city_bytes && !city_off))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "gen_persons: bytes requested without offsets");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t g0 = persons_before(s->first_event_id + n0), rows = persons_before(s->first_event_id + n1) - g0;
    if (rows == 0) {
        if (name_off) FG_HIP(ctx, hipMemsetAsync(name_off, 0, sizeof(int32_t), ctx->stream));
        if (city_off) FG_HIP(ctx, hipMemsetAsync(city_off, 0, sizeof(int32_t), ctx->stream));
        if (state_off) FG_HIP(ctx, hipMemsetAsync(state_off, 0, sizeof(int32_t), ctx->stream));
        return FLOCKGPU_OK;
    }
    {
        LaunchScope ls(ctx, "gen_persons_len_kernel");
        hipLaunchKernelGGL(gen_persons_len_kernel, dim3(grid_for(ctx, rows)), dim3(kBlock), 0, ctx->stream, *s, g0, rows, p_id,
                           name_off, city_off, state_off);
    }
    FG_TRY(check_launch(ctx, "gen_persons_len_kernel"));
    if (name_off) FG_TRY(inclusive_scan_i32(ctx, "gen.scan_name", name_off + 1, (int64_t)rows));
    if (city_off) FG_TRY(inclusive_scan_i32(ctx, "gen.scan_city", city_off + 1, (int64_t)rows));
    {
        LaunchScope ls(ctx, "gen_persons_bytes_kernel");
        hipLaunchKernelGGL(gen_persons_bytes_kernel, dim3(grid_for(ctx, rows)), dim3(kBlock), 0, ctx->stream, *s, g0, rows,
                           name_off, name_bytes, city_off, city_bytes, state_bytes);
    }
    return check_launch(ctx, "gen_persons_bytes_kernel");
}

}  // extern "C"
