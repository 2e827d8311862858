// Global-memory open-addressing hash tables used by the join / distinct kernels (q3, q8).
// One table region per window: region w = table + w * cap.  Slots are single naturally aligned 64-bit (or
// 32-bit) words updated with relaxed agent-scope atomics, so build kernels need no fences; the probe runs
// in a later kernel (kernel boundary = release/acquire).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace flockgpu {

constexpr uint32_t kFibHash = 0x9E3779B1u;
constexpr uint64_t kEmpty64 = ~0ull;   // {key = -1, row = -1}: row -1 is never stored
constexpr uint32_t kEmpty32 = ~0u;
// Linear probing is cut off after this many slots: at the load factors the host sizes for (<= 0.67) a longer
// run means the region is overloaded; the kernel raises its error flag and the host retries with a larger table.
constexpr uint32_t kMaxProbe = 2048;
__device__ __forceinline__ uint32_t probe_limit(uint32_t cap) { return cap < kMaxProbe ? cap : kMaxProbe; }

__device__ __forceinline__ uint32_t slot_of(uint32_t key, uint32_t cap) {
    return (uint32_t)(((uint64_t)(key * kFibHash) * cap) >> 32);
}
__device__ __forceinline__ uint64_t pack_kr(int32_t key, int32_t row) {
    return ((uint64_t)(uint32_t)key << 32) | (uint32_t)row;
}
__device__ __forceinline__ uint64_t ld64(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cas64(uint64_t *p, uint64_t &expected, uint64_t desired) {
    return __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
}

// Multimap insert: slot = {key, head row}; duplicates of a key are pushed in front of the chain `next[]`.
// Returns false when the region is full.
__device__ __forceinline__ bool multimap_insert(uint64_t *tab, uint32_t cap, int32_t *next, int32_t key, int32_t row) {
    uint32_t s = slot_of((uint32_t)key, cap);
#pragma unroll 1
    for (uint32_t probe = 0, lim = probe_limit(cap); probe < lim; ++probe) {
        uint64_t cur = ld64(&tab[s]);
        if (cur == kEmpty64) {
            if (cas64(&tab[s], cur, pack_kr(key, row))) {
                next[row] = -1;
                return true;
            }
        }
        while ((int32_t)(cur >> 32) == key && cur != kEmpty64) {
            // same key: become the new head, old head is our successor
            const int32_t old_head = (int32_t)(uint32_t)cur;
            next[row] = old_head;
            if (cas64(&tab[s], cur, pack_kr(key, row))) return true;
        }
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    return false;
}

// The same multimap with the chain's continuation marked IN the link: bit 31 of a slot's row field -- and of every next[] entry -- says
// "this row has a successor, next[row] holds it (with its own mark)".  A key that occurs once costs its readers no load of next[] at all
// (q3's hash path: 6e6 matches per call, each a dependent random load from an 80 MB array before) and its insert no store to it.
// Walk:  for (uint32_t c = field;; c = (uint32_t)next[c & kChainRow]) { row = c & kChainRow; ...; if (!(c & kChainMore)) break; }
constexpr uint32_t kChainMore = 0x80000000u, kChainRow = 0x7fffffffu;
__device__ __forceinline__ bool multimap_insert_marked(uint64_t *tab, uint32_t cap, int32_t *next, int32_t key, int32_t row) {
    uint32_t s = slot_of((uint32_t)key, cap);
#pragma unroll 1
    for (uint32_t probe = 0, lim = probe_limit(cap); probe < lim; ++probe) {
        uint64_t cur = ld64(&tab[s]);
        if (cur == kEmpty64) {
            if (cas64(&tab[s], cur, pack_kr(key, row))) return true;
        }
        while ((int32_t)(cur >> 32) == key && cur != kEmpty64) {
            // same key: become the new head, the old head field (row + its own mark) is our successor
            next[row] = (int32_t)(uint32_t)cur;
            if (cas64(&tab[s], cur, pack_kr(key, (int32_t)((uint32_t)row | kChainMore)))) return true;
        }
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    return false;
}

// Head row of `key`'s chain or -1.
__device__ __forceinline__ int32_t multimap_find(const uint64_t *tab, uint32_t cap, int32_t key) {
    uint32_t s = slot_of((uint32_t)key, cap);
#pragma unroll 1
    for (uint32_t probe = 0, lim = probe_limit(cap); probe < lim; ++probe) {
        const uint64_t cur = tab[s];
        if (cur == kEmpty64) return -1;
        if ((int32_t)(cur >> 32) == key) return (int32_t)(uint32_t)cur;
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    return -1;
}

// Set insert (insert-if-absent).  Returns 1 = inserted, 0 = already present, -1 = region full.
__device__ __forceinline__ int set_insert(uint64_t *tab, uint32_t cap, int32_t key, int32_t row) {
    uint32_t s = slot_of((uint32_t)key, cap);
#pragma unroll 1
    for (uint32_t probe = 0, lim = probe_limit(cap); probe < lim; ++probe) {
        uint64_t cur = ld64(&tab[s]);
        if (cur == kEmpty64) {
            if (cas64(&tab[s], cur, pack_kr(key, row))) return 1;
        }
        if ((int32_t)(cur >> 32) == key) return 0;
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    return -1;
}

}  // namespace flockgpu
