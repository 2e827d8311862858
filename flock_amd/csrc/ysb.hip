// Yahoo Streaming Benchmark for gfx950, per Tumbling(10 s) window
// (benchmarks/src/ysb/main.rs:91):
//   SELECT campaign_id, COUNT(*) FROM ad_event INNER JOIN campaign ON ad_id = c_ad_id WHERE event_type = 'view'
//   GROUP BY campaign_id                      (benchmarks/src/ysb/ysb.sql; stages flock/src/distributed_plan/planner.rs:298-346)
// A Utf8-KEYED join and group-by: both keys are 36-byte UUID strings (flock/src/datasource/ysb/event.rs:24-87).
//
// HBM-bound byte work, no MFMA.  The campaign table is small and static; the stream is the ad events:
//   dict   : DISTINCT campaign_id over the campaign rows (slot claim by string hash + full string compare): every campaign
//            row learns its group = the row that claimed its campaign_id
//   build  : multimap keyed c_ad_id (string hash -> slot {hash, head row}, chain through next[]), full compare on probe
//   pack   : the multimap re-laid as 64-byte slots {key bytes, head row, its group, its chain link}: one read per probe;
//            beside it a 4-byte TAG per slot (31 bits of the key's hash) -- the copy of the table a workgroup keeps in LDS
//   count  : per 2048-event tile, (1) `event_type = 'view'` for eight rows per lane, the tile's ad_id offsets parked in LDS
//            on the way, (2) the rows that passed compacted in LDS and probed two per lane: ad_id bytes, hash, the walk
//            to the slot carrying this hash's tag IN LDS (a divergent loop that never waits for memory), ONE 64-byte slot
//            read to compare the 36 bytes -> LDS histogram over the campaign rows, flushed per tile with one atomic per
//            touched group.  Strings are read through dword-aligned 16-byte loads, all loads of a value requested
//            together, 40 bytes on the fast path.  (Probing the global table directly -- 1.5 dependent L2 reads per
//            event inside the divergent loop -- cost 0.29 of the kernel's 0.78 ms; the LDS walk brought it to 0.58 ms.)
//   output : groups with a non-zero count per window, `take` of their campaign_id.
#include <algorithm>

#include "gather.hpp"

using namespace flockgpu;

namespace {

constexpr int kWords = 10;            // fast path: values of up to 40 bytes
constexpr uint32_t kEmptySlot = ~0u;
constexpr int kEvItems = 8;
constexpr int kEvTile = kBlock * kEvItems;  // 2048 events per workgroup; event  it*256 + tid  belongs to thread tid
constexpr int kHistGroups = 4095;           // campaign rows whose counts AND table tags (2 rows + 1 slots) a workgroup's LDS holds

// A Utf8 value as little-endian 32-bit words w[0 .. ceil(len/4)) (bytes past the end zeroed); len <= 4*kWords.
struct StrWords {
    uint32_t w[kWords];
    uint32_t len;
};

// `safe_end`: bytes of the column buffer that may be read (its length rounded up to a whole dword).  A value that lies at
// least 48 bytes before it is fetched with THREE 16-byte loads (dword-aligned vector loads) instead of eleven dword loads:
// one lane per row means every load instruction of a wave touches ~18 cache lines, so the instruction count is what the
// texture path pays for.  Values at the very end of the buffer take clamped dword loads (never past their last word).
__device__ __forceinline__ StrWords load_str(const uint8_t *__restrict__ data, int32_t b, uint32_t len, int64_t safe_end) {
    StrWords s;
    s.len = len;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + (uint32_t)b;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(addr & ~uintptr_t(3));
    const uint32_t sh = (uint32_t)(addr & 3) * 8;
    uint32_t a[kWords + 2];
    if ((int64_t)((uint32_t)b & ~3u) + 48 <= safe_end) {
        const uint4 v0 = *reinterpret_cast<const uint4 *>(p), v1 = *reinterpret_cast<const uint4 *>(p + 4),
                    v2 = *reinterpret_cast<const uint4 *>(p + 8);
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
        a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
        a[8] = v2.x; a[9] = v2.y; a[10] = v2.z; a[11] = v2.w;
    } else {
        const uint32_t last = len ? (uint32_t)(((addr & 3) + len - 1) >> 2) : 0u;
#pragma unroll
        for (int i = 0; i <= kWords; ++i) a[i] = p[min((uint32_t)i, last)];
    }
#pragma unroll
    for (int i = 0; i < kWords; ++i) {
        uint32_t v = __funnelshift_r(a[i], a[i + 1], sh);
        const uint32_t have = len > 4u * i ? len - 4u * i : 0u;  // bytes of this word inside the value
        v = have >= 4 ? v : (have ? (v & ((1u << (8 * have)) - 1)) : 0u);
        s.w[i] = v;
    }
    return s;
}

__device__ __forceinline__ uint32_t hash_words(const StrWords &s) {
    uint32_t h = 0x811C9DC5u ^ s.len;
#pragma unroll
    for (int i = 0; i < kWords; ++i) {
        h = (h ^ s.w[i]) * 0x9E3779B1u;
        h = (h << 13) | (h >> 19);
    }
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ bool same_words(const StrWords &a, const StrWords &b) {
    bool eq = a.len == b.len;
#pragma unroll
    for (int i = 0; i < kWords; ++i) eq = eq && a.w[i] == b.w[i];
    return eq;
}

__device__ __forceinline__ StrWords row_str(const flockgpu_utf8 &c, int64_t row, int64_t safe_end) {
    const int32_t b = c.offsets[row];
    return load_str(c.data, b, (uint32_t)(c.offsets[row + 1] - b), safe_end);
}

__device__ __forceinline__ uint32_t slot_for(uint32_t h, uint32_t cap) { return (uint32_t)(((uint64_t)h * cap) >> 32); }
// What the LDS copy of the table keeps of a slot's key: 31 bits of its hash (kEmptySlot = no key).  The slot index comes
// from the hash's high bits, so the low ones are the ones that tell two keys of one neighbourhood apart.
__device__ __forceinline__ uint32_t tag_of(uint32_t h) { return h & 0x7FFFFFFFu; }

// rep[r] = the campaign row that claimed r's campaign_id (DISTINCT campaign_id).
__global__ __launch_bounds__(kBlock) void ysb_dict_kernel(flockgpu_utf8 campaign_id, int32_t n, uint32_t *table, uint32_t cap,
                                                          int32_t *__restrict__ rep, uint32_t *err) {
    const int32_t r = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (r >= n) return;
    const int64_t safe_end = ((int64_t)campaign_id.offsets[n] + 3) & ~int64_t(3);
    const StrWords me = row_str(campaign_id, r, safe_end);
    if (me.len > 4u * kWords) {  // only the first 40 bytes are compared: longer values are not supported
        atomicOr(err, 2u);
        rep[r] = r;
        return;
    }
    uint32_t s = slot_for(hash_words(me), cap);
    for (uint32_t probe = 0; probe < cap; ++probe) {
        uint32_t cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmptySlot) {
            uint32_t expected = kEmptySlot;
            if (__hip_atomic_compare_exchange_strong(&table[s], &expected, (uint32_t)r, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                rep[r] = r;
                return;
            }
            cur = expected;
        }
        if (same_words(me, row_str(campaign_id, (int32_t)cur, safe_end))) {
            rep[r] = (int32_t)cur;
            return;
        }
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    atomicOr(err, 1u);
}

// Multimap keyed c_ad_id: slot = head row of the chain of rows with this string; next[] links the duplicates.
__global__ __launch_bounds__(kBlock) void ysb_build_kernel(flockgpu_utf8 c_ad_id, int32_t n, uint32_t *table, uint32_t cap,
                                                           int32_t *next, uint32_t *err) {
    const int32_t r = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (r >= n) return;
    const int64_t safe_end = ((int64_t)c_ad_id.offsets[n] + 3) & ~int64_t(3);
    const StrWords me = row_str(c_ad_id, r, safe_end);
    if (me.len > 4u * kWords) {
        atomicOr(err, 2u);
        next[r] = -1;
        return;
    }
    uint32_t s = slot_for(hash_words(me), cap);
    for (uint32_t probe = 0; probe < cap; ++probe) {
        uint32_t cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmptySlot) {
            next[r] = -1;
            uint32_t expected = kEmptySlot;
            if (__hip_atomic_compare_exchange_strong(&table[s], &expected, (uint32_t)r, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT))
                return;
            cur = expected;
        }
        // the slot is owned by a row: same string -> become the new head of its chain, else keep probing.  The owner's
        // string never changes (only rows with an EQUAL string replace it), so comparing with the current head is enough.
        if (same_words(me, row_str(c_ad_id, (int32_t)cur, safe_end))) {
            for (;;) {
                next[r] = (int32_t)cur;
                uint32_t expected = cur;
                if (__hip_atomic_compare_exchange_strong(&table[s], &expected, (uint32_t)r, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT))
                    return;
                cur = expected;
            }
        }
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    atomicOr(err, 1u);
}

struct EvLit {  // the literal of `event_type = lit`, up to 40 bytes
    uint32_t w[kWords];
    uint32_t len;
};

// One probe = one 64-byte read: the key of the slot's chain, its head row, that row's group and its chain link all sit
// in the slot, so the event side never chases  table -> offsets -> bytes -> next -> rep  (five dependent loads).
struct __align__(16) KeySlot {
    uint32_t w[kWords];
    uint32_t len;
    int32_t head;   // head campaign row of the chain with this c_ad_id, -1 = empty slot
    int32_t group;  // rep[head]
    int32_t link;   // next[head]
    uint32_t pad[2];
};
static_assert(sizeof(KeySlot) == 64, "one slot per 64 bytes");

__global__ __launch_bounds__(kBlock) void ysb_pack_kernel(flockgpu_utf8 c_ad_id, int32_t n, const uint32_t *__restrict__ table,
                                                          uint32_t cap, const int32_t *__restrict__ next,
                                                          const int32_t *__restrict__ rep, KeySlot *__restrict__ slots,
                                                          uint32_t *__restrict__ tags) {
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= cap) return;
    KeySlot k{};
    k.head = -1;
    uint32_t tag = kEmptySlot;
    const uint32_t cur = table[s];
    if (cur != kEmptySlot) {
        const int64_t safe_end = ((int64_t)c_ad_id.offsets[n] + 3) & ~int64_t(3);
        const StrWords me = row_str(c_ad_id, (int32_t)cur, safe_end);
#pragma unroll
        for (int i = 0; i < kWords; ++i) k.w[i] = me.w[i];
        k.len = me.len;
        k.head = (int32_t)cur;
        k.group = rep[cur];
        k.link = next[cur];
        tag = tag_of(hash_words(me));
    }
    slots[s] = k;
    tags[s] = tag;
}

// `event_type = lit` for one row.  kLitDwords = 2 / 4: a literal of up to 4 / 12 bytes (the YSB ones are 4-8) is
// compared through ONE load of that many dwords from the value's first dword on (adjacent rows share cache lines, and the
// fewer registers eight rows in flight take, the more waves a SIMD holds); 0: any literal, through the general word loader.
template <int kLitDwords>
__device__ __forceinline__ bool event_is(const flockgpu_utf8 &event_type, int64_t row, const EvLit &lit, int64_t safe_end) {
    const int32_t eb = event_type.offsets[row];
    const uint32_t elen = (uint32_t)(event_type.offsets[row + 1] - eb);
    if (kLitDwords > 0) {
        const uintptr_t addr = reinterpret_cast<uintptr_t>(event_type.data) + (uint32_t)eb;
        const uint32_t *p = reinterpret_cast<const uint32_t *>(addr & ~uintptr_t(3));
        const uint32_t sh = (uint32_t)(addr & 3) * 8;
        uint32_t a[4] = {0u, 0u, 0u, 0u};
        if ((int64_t)((uint32_t)eb & ~3u) + 4 * kLitDwords <= safe_end) {
            if (kLitDwords == 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(p);
                a[0] = v.x; a[1] = v.y;
            } else {
                const uint4 v = *reinterpret_cast<const uint4 *>(p);
                a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
            }
        } else {
            const uint32_t last = elen ? (uint32_t)(((addr & 3) + elen - 1) >> 2) : 0u;
#pragma unroll
            for (int i = 0; i < kLitDwords; ++i) a[i] = p[min((uint32_t)i, last)];
        }
        bool is = elen == lit.len;
#pragma unroll
        for (int i = 0; i < kLitDwords - 1; ++i) {
            uint32_t v = __funnelshift_r(a[i], a[i + 1], sh);
            const uint32_t have = elen > 4u * i ? elen - 4u * i : 0u;
            v = have >= 4 ? v : (have ? (v & ((1u << (8 * have)) - 1)) : 0u);
            is = is && v == lit.w[i];
        }
        return is;
    } else {
        if (elen != lit.len) return false;  // lit.len <= 40, so the loader never sees a longer value
        const StrWords ev = load_str(event_type.data, eb, elen, safe_end);
        bool is = true;
#pragma unroll
        for (int i = 0; i < kWords; ++i) is = is && ev.w[i] == lit.w[i];
        return is;
    }
}

// One probe of the 64-byte slot table in global memory from slot `s` on: the general walk (tables too large for LDS) and
// the continuation of the LDS walk after a tag that matched a different key.
template <bool kLdsHist>
__device__ __forceinline__ void probe_slots(const StrWords &key, uint32_t s, const KeySlot *__restrict__ slots, uint32_t cap,
                                            const int32_t *__restrict__ next, const int32_t *__restrict__ rep, uint32_t *s_hist,
                                            unsigned long long *wc) {
    for (uint32_t probe = 0; probe < cap; ++probe) {
        const uint4 *q = reinterpret_cast<const uint4 *>(slots + s);
        const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        if ((int32_t)q2.w < 0) return;  // head: empty slot
        const uint32_t *k = key.w;
        const bool eq = q0.x == k[0] && q0.y == k[1] && q0.z == k[2] && q0.w == k[3] && q1.x == k[4] && q1.y == k[5] &&
                        q1.z == k[6] && q1.w == k[7] && q2.x == k[8] && q2.y == k[9] && q2.z == key.len;
        if (eq) {
            int32_t g = (int32_t)q3.x;
            for (int32_t c = (int32_t)q3.y;; c = next[c]) {  // one output row per matching campaign row
                if (kLdsHist) atomicAdd(&s_hist[g], 1u);
                else atomicAdd(&wc[g], 1ull);
                if (c < 0) break;
                g = rep[c];
            }
            return;
        }
        s = (s + 1 == cap) ? 0 : s + 1;
    }
}

// counts[seg * n_camp + group] += matches.  kLds: the campaign side fits the workgroup's LDS (n_camp <= kHistGroups): the
// block pre-aggregates its counts there and keeps a copy of the table's tags (31 hash bits per slot) beside them.
// Two phases per 2048-event tile, so that neither runs under the other's divergence:
//   1. the filter, eight rows per lane, all loads independent of each other (clamped rows, no load under a branch); the
//      tile's ad_id offsets go to LDS in the same breath, so phase 2 starts at the bytes;
//   2. the rows that passed, compacted into an LDS list, taken two per lane at a time: ad_id bytes (three 16-byte loads),
//      hash, the walk to the slot with this hash's tag IN LDS (the divergent loop never waits for memory), ONE 64-byte
//      slot read to compare the whole key, LDS histogram.
template <bool kLds, int kLitDwords, int kPair, int kMinWaves>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kMinWaves))) void ysb_count_kernel(flockgpu_utf8 ad_id, flockgpu_utf8 event_type, SegTiles st, EvLit lit,
                                                           const KeySlot *__restrict__ slots, const uint32_t *__restrict__ tags,
                                                           uint32_t cap, const int32_t *__restrict__ next,
                                                           const int32_t *__restrict__ rep, int32_t n_camp, int64_t n_events,
                                                           unsigned long long *counts, uint32_t *err) {
    extern __shared__ uint32_t s_dyn[];  // kLds: [n_camp] counts, [cap] tags
    __shared__ uint16_t s_list[kEvTile];
    __shared__ int32_t s_off[kEvTile + 1];
    __shared__ uint32_t s_wave_total[kWavesPerBlock];
    uint32_t *s_hist = s_dyn, *s_tags = s_dyn + n_camp;
    const int64_t end_et = ((int64_t)event_type.offsets[n_events] + 3) & ~int64_t(3);
    const int64_t end_ad = ((int64_t)ad_id.offsets[n_events] + 3) & ~int64_t(3);
    const TileRange tr = locate_tile(st, (int32_t)blockIdx.x, kEvTile);
    unsigned long long *wc = counts + (size_t)tr.seg * n_camp;

    uint32_t mask = 0;
#pragma unroll
    for (int it = 0; it < kEvItems; ++it) {
        const int64_t r = tr.tile_begin + it * kBlock + threadIdx.x;
        const bool in = r >= tr.lo && r < tr.hi;
        const bool is = event_is<kLitDwords>(event_type, in ? r : tr.lo, lit, end_et);
        mask |= (uint32_t)(in && is) << it;
        s_off[it * kBlock + threadIdx.x] = ad_id.offsets[min(r, n_events)];
    }
    if (threadIdx.x == 0) s_off[kEvTile] = ad_id.offsets[min(tr.tile_begin + kEvTile, n_events)];
    if (kLds) {
        for (int s = threadIdx.x; s < n_camp; s += kBlock) s_hist[s] = 0;
        for (uint32_t s = threadIdx.x; s < cap; s += kBlock) s_tags[s] = tags[s];
    }
    const uint32_t cnt = (uint32_t)__popc(mask);
    const uint32_t incl = wave_incl_scan_u32(cnt);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) s_wave_total[wave] = incl;
    __syncthreads();
    uint32_t base = incl - cnt, n_pass = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        const uint32_t t = s_wave_total[w];
        base += w < wave ? t : 0u;
        n_pass += t;
    }
    for (uint32_t m = mask; m; m &= m - 1) s_list[base++] = (uint16_t)((__ffs(m) - 1) * kBlock + threadIdx.x);
    __syncthreads();

    for (uint32_t i0 = 0; i0 < n_pass; i0 += kPair * kBlock) {
        StrWords key[kPair];
        bool live[kPair];
#pragma unroll
        for (int u = 0; u < kPair; ++u) {
            const uint32_t i = i0 + u * kBlock + threadIdx.x;
            live[u] = i < n_pass;
            const uint32_t j = s_list[live[u] ? i : 0u];
            const int32_t ab = s_off[j];
            key[u] = load_str(ad_id.data, ab, (uint32_t)(s_off[j + 1] - ab), end_ad);
        }
        if (!kLds) {
#pragma unroll
            for (int u = 0; u < kPair; ++u) {
                if (!live[u]) continue;
                if (key[u].len > 4u * kWords) { atomicOr(err, 2u); continue; }
                probe_slots<false>(key[u], slot_for(hash_words(key[u]), cap), slots, cap, next, rep, s_hist, wc);
            }
            continue;
        }
        // the walk in LDS: the first slot from the hash's home on that holds this tag (a candidate) or nothing (no match)
        uint32_t at[kPair];
#pragma unroll
        for (int u = 0; u < kPair; ++u) {
            if (live[u] && key[u].len > 4u * kWords) { atomicOr(err, 2u); live[u] = false; }
            const uint32_t h = hash_words(key[u]), tag = tag_of(h);
            uint32_t s = slot_for(h, cap), t = kEmptySlot;
            if (live[u])
                while ((t = s_tags[s]) != tag && t != kEmptySlot) s = (s + 1 == cap) ? 0 : s + 1;  // cap > rows: an empty slot exists
            live[u] = live[u] && t == tag;
            at[u] = live[u] ? s : 0u;
        }
        uint4 q[kPair][4];
#pragma unroll
        for (int u = 0; u < kPair; ++u) {
            const uint4 *p = reinterpret_cast<const uint4 *>(slots + at[u]);
            q[u][0] = p[0]; q[u][1] = p[1]; q[u][2] = p[2]; q[u][3] = p[3];
        }
#pragma unroll
        for (int u = 0; u < kPair; ++u) {
            if (!live[u]) continue;
            const uint32_t *k = key[u].w;
            const uint4 q0 = q[u][0], q1 = q[u][1], q2 = q[u][2], q3 = q[u][3];
            const bool eq = q0.x == k[0] && q0.y == k[1] && q0.z == k[2] && q0.w == k[3] && q1.x == k[4] && q1.y == k[5] &&
                            q1.z == k[6] && q1.w == k[7] && q2.x == k[8] && q2.y == k[9] && q2.z == key[u].len;
            if (eq && (int32_t)q3.y < 0) atomicAdd(&s_hist[(int32_t)q3.x], 1u);  // the key of ONE campaign row: the usual case
            else probe_slots<true>(key[u], at[u], slots, cap, next, rep, s_hist, wc);  // a chain, or another key with this tag
        }
    }
    if (!kLds) return;
    __syncthreads();
    for (int s = threadIdx.x; s < n_camp; s += kBlock) {
        const uint32_t c = s_hist[s];
        if (c) atomicAdd(&wc[s], (unsigned long long)c);
    }
}

}  // namespace

extern "C" {

int flockgpu_ysb_campaign_counts(flockgpu_ctx *ctx, const flockgpu_ysb_event_cols *events, const flockgpu_windows *win,
                                 const flockgpu_ysb_campaign_cols *campaigns, const char *event_type_lit,
                                 flockgpu_ysb_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!events || !campaigns || !out || !event_type_lit || events->rows < 0 || campaigns->rows < 0)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "ysb: null argument");
    FG_TRY(check_windows(ctx, win, events->rows, "ysb"));
    if (events->rows > 0 && (!events->ad_id.offsets || !events->ad_id.data || !events->event_type.offsets || !events->event_type.data))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "ysb: null event column");
    if (campaigns->rows > 0 && (!campaigns->c_ad_id.offsets || !campaigns->c_ad_id.data || !campaigns->campaign_id.offsets ||
                                !campaigns->campaign_id.data))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "ysb: null campaign column");
    if (campaigns->rows >= (int64_t(1) << 30) || events->rows >= (int64_t(1) << 31))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "ysb: relations are limited to 2^31 / 2^30 rows per call");
    EvLit lit{};
    lit.len = (uint32_t)std::strlen(event_type_lit);
    if (lit.len > 4u * kWords) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "ysb: literal longer than %d bytes", 4 * kWords);
    for (uint32_t i = 0; i < lit.len; ++i) lit.w[i / 4] |= (uint32_t)(uint8_t)event_type_lit[i] << (8 * (i % 4));
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = win->n_windows;
    const int32_t n_camp = (int32_t)campaigns->rows;
    std::vector<int64_t> sb(n_win), se(n_win);
    for (int w = 0; w < n_win; ++w) {
        sb[w] = win->pane_row_offsets[win->win_pane_lo[w]];
        se[w] = win->pane_row_offsets[win->win_pane_hi[w]];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, "ysb", sb.data(), se.data(), n_win, kEvTile, &st));

    const uint32_t cap = (uint32_t)std::max<int64_t>(64, (int64_t)n_camp * 2 + 1);
    uint32_t *dict = nullptr, *table = nullptr, *d_err = nullptr;
    int32_t *rep = nullptr, *next = nullptr;
    FG_TRY(arena_get_t(ctx, "ysb.dict", (size_t)cap, &dict));
    FG_TRY(arena_get_t(ctx, "ysb.table", (size_t)cap, &table));
    FG_TRY(arena_get_t(ctx, "ysb.rep", (size_t)n_camp + 1, &rep));
    FG_TRY(arena_get_t(ctx, "ysb.next", (size_t)n_camp + 1, &next));
    FG_TRY(arena_get_t(ctx, "ysb.err", 4, &d_err));
    const size_t n_counts = (size_t)std::max(n_win, 1) * std::max(n_camp, 1);
    unsigned long long *d_counts = nullptr, *h_counts = nullptr;
    FG_TRY(arena_get_t(ctx, "ysb.counts", n_counts, &d_counts));
    FG_TRY(pinned_get_t(ctx, "ysb.counts", n_counts + 1, &h_counts));
    FG_HIP(ctx, hipMemsetAsync(dict, 0xFF, sizeof(uint32_t) * cap, ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(table, 0xFF, sizeof(uint32_t) * cap, ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(uint32_t), ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(unsigned long long) * n_counts, ctx->stream));
    if (n_camp > 0) {
        const unsigned gb = (unsigned)div_up(n_camp, kBlock);
        {
            LaunchScope ls(ctx, "ysb_dict_kernel");
            hipLaunchKernelGGL(ysb_dict_kernel, dim3(gb), dim3(kBlock), 0, ctx->stream, campaigns->campaign_id, n_camp, dict, cap, rep,
                               d_err);
        }
        FG_TRY(check_launch(ctx, "ysb_dict_kernel"));
        {
            LaunchScope ls(ctx, "ysb_build_kernel");
            hipLaunchKernelGGL(ysb_build_kernel, dim3(gb), dim3(kBlock), 0, ctx->stream, campaigns->c_ad_id, n_camp, table, cap, next,
                               d_err);
        }
        FG_TRY(check_launch(ctx, "ysb_build_kernel"));
    }
    if (st.n_tiles > 0 && n_camp > 0) {
        KeySlot *slots = nullptr;
        uint32_t *tags = nullptr;
        FG_TRY(arena_get_t(ctx, "ysb.slots", (size_t)cap, &slots));
        FG_TRY(arena_get_t(ctx, "ysb.tags", (size_t)cap, &tags));
        {
            LaunchScope ls(ctx, "ysb_pack_kernel");
            hipLaunchKernelGGL(ysb_pack_kernel, dim3((unsigned)div_up((int64_t)cap, kBlock)), dim3(kBlock), 0, ctx->stream,
                               campaigns->c_ad_id, n_camp, table, cap, next, rep, slots, tags);
        }
        FG_TRY(check_launch(ctx, "ysb_pack_kernel"));
        LaunchScope ls(ctx, "ysb_count_kernel");
        const bool lds = n_camp <= kHistGroups;
        const int lit_dwords = lit.len <= 4 ? 2 : lit.len <= 12 ? 4 : 0;
        const size_t shmem = lds ? sizeof(uint32_t) * ((size_t)n_camp + cap) : 0;
        // waves per SIMD the compiler is held to: measured on the 'view' literal, 6 waves (76 VGPRs, nothing spilled) beat the
        // 5 it picks by itself by 4 %, 8 (spilling) lose 15 %
        auto kernel = lds ? ysb_count_kernel<true, 0, 2, 6> : ysb_count_kernel<false, 0, 2, 6>;
        if (lit_dwords == 2) kernel = lds ? ysb_count_kernel<true, 2, 2, 6> : ysb_count_kernel<false, 2, 2, 6>;
        if (lit_dwords == 4) kernel = lds ? ysb_count_kernel<true, 4, 2, 5> : ysb_count_kernel<false, 4, 2, 5>;
        hipLaunchKernelGGL(kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), shmem, ctx->stream, events->ad_id, events->event_type, st,
                           lit, slots, tags, cap, next, rep, n_camp, events->rows, d_counts, d_err);
    }
    FG_TRY(check_launch(ctx, "ysb_count_kernel"));
    FG_HIP(ctx, hipMemcpyAsync(h_counts, d_counts, sizeof(unsigned long long) * n_counts, hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipMemcpyAsync(h_counts + n_counts, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t h_err = *reinterpret_cast<uint32_t *>(h_counts + n_counts);
    if (h_err & 2u) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "ysb: a key or event_type value is longer than %d bytes", 4 * kWords);
    if (h_err & 1u) return fail(ctx, FLOCKGPU_ERR_CAPACITY, "ysb: campaign table overflow");

    // groups with a non-zero count, window by window, in order of their representative campaign row
    std::vector<int64_t> &offs = ctx->host_i64["ysb.win_out_offsets"];
    offs.assign((size_t)n_win + 1, 0);
    std::vector<int32_t> rows;
    std::vector<uint64_t> cnts;
    for (int w = 0; w < n_win; ++w) {
        for (int32_t g = 0; g < n_camp; ++g) {
            const unsigned long long c = h_counts[(size_t)w * n_camp + g];
            if (c) {
                rows.push_back(g);
                cnts.push_back(c);
            }
        }
        offs[w + 1] = (int64_t)rows.size();
    }
    const size_t n_out = rows.size();
    int32_t *d_rows = nullptr, *h_rows = nullptr;
    uint64_t *d_cnt = nullptr, *h_cnt = nullptr;
    FG_TRY(arena_get_t(ctx, "ysb.out_rows", n_out + 1, &d_rows));
    FG_TRY(pinned_get_t(ctx, "ysb.out_rows", n_out + 1, &h_rows));
    FG_TRY(arena_get_t(ctx, "ysb.out_count", n_out + 1, &d_cnt));
    FG_TRY(pinned_get_t(ctx, "ysb.out_count", n_out + 1, &h_cnt));
    std::copy(rows.begin(), rows.end(), h_rows);
    std::copy(cnts.begin(), cnts.end(), h_cnt);
    if (n_out) {
        FG_HIP(ctx, hipMemcpyAsync(d_rows, h_rows, sizeof(int32_t) * n_out, hipMemcpyHostToDevice, ctx->stream));
        FG_HIP(ctx, hipMemcpyAsync(d_cnt, h_cnt, sizeof(uint64_t) * n_out, hipMemcpyHostToDevice, ctx->stream));
    }
    FG_TRY(gather_utf8(ctx, "ysb.out_campaign", campaigns->campaign_id, d_rows, (int64_t)n_out, &out->campaign_id, &out->campaign_bytes));
    out->count = d_cnt;
    out->win_out_offsets = offs.data();
    out->rows = (int64_t)n_out;
    return FLOCKGPU_OK;
}

}  // extern "C"
