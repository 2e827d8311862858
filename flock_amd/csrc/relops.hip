// Generic relational operators over device columns (relops.hpp): the plan interpreter's fallback for plan nodes that no
// fused NEXMark pipeline covers -- the stage plans either side of a hash repartition.  One lane per row, 64-bit
// normalised keys, global-memory hash tables; exact for any input.
#include <algorithm>

#include "relops.hpp"

using namespace flockgpu;

namespace {

constexpr int64_t kEmptyKey = INT64_MIN;  // hash-table sentinel; the key INT64_MIN itself lives in the extra slot `cap`

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

inline unsigned grid_for(flockgpu_ctx *ctx, int64_t n) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up(n, kBlock), (int64_t)ctx->num_cus * 16));
}

__device__ __forceinline__ int64_t load_as_i64(const void *v, int32_t type, int64_t i) {
    return type == (int32_t)ColType::I32 ? (int64_t) static_cast<const int32_t *>(v)[i] : static_cast<const int64_t *>(v)[i];
}
// U64 columns compare as unsigned; everything else as signed
__device__ __forceinline__ bool cmp_i64(int64_t a, int64_t b, int32_t op, bool uns) {
    if (uns) {
        const uint64_t x = (uint64_t)a, y = (uint64_t)b;
        switch (op) {
            case 0: return x == y;
            case 1: return x != y;
            case 2: return x < y;
            case 3: return x <= y;
            case 4: return x > y;
            default: return x >= y;
        }
    }
    switch (op) {
        case 0: return a == b;
        case 1: return a != b;
        case 2: return a < b;
        case 3: return a <= b;
        case 4: return a > b;
        default: return a >= b;
    }
}

__global__ __launch_bounds__(kBlock) void widen_kernel(const void *__restrict__ v, int32_t type, int64_t n, int64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = load_as_i64(v, type, i);
}
__global__ __launch_bounds__(kBlock) void widen_u32_kernel(const uint32_t *__restrict__ in, int64_t n, uint64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = in[i];
}
// mask bytes -> flag words in the flag-tile geometry (scan.hpp): the lane's four consecutive rows are one 32-bit load
__global__ __launch_bounds__(kBlock) void mask_flag_kernel(const uint8_t *__restrict__ mask, int64_t n_rows, SegTiles st,
                                                           uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts) {
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const int64_t wbase = tr.tile_begin + flag_rel0();
    uint32_t flags = 0;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        const int64_t r0 = wbase + it * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + j;
            const bool on = r >= tr.lo && r < tr.hi && mask[r] != 0;
            flags |= (uint32_t)on << (it * 4 + j);
        }
    }
    store_flags_and_counts(flags, tile, flag_words, counts);
}

// ---- group by
__global__ __launch_bounds__(kBlock) void group_init_kernel(int64_t *__restrict__ tk, uint64_t *__restrict__ ta, int32_t *__restrict__ tf,
                                                            int64_t slots, uint32_t *__restrict__ err) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *err = 0u;   // (instead of a hipMemsetAsync of its own: ~10 us of host time each)
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < slots; i += (int64_t)gridDim.x * kBlock) {
        tk[i] = kEmptyKey;
        ta[i] = 0;
        tf[i] = 0x7fffffff;
    }
}
// slot of `key` in an open-addressing table of `cap` (power of two) slots, claiming an empty one; -1: table full
// (probing is cut off after kClaimProbes slots: in a table at most half full a longer run does not happen, and in one sized from a hint
// that turned out too small -- group_by_key64_n -- it must end in "full", not in a walk over the whole table per row)
constexpr uint64_t kClaimProbes = 4096;
constexpr int kJoinTotalSlots = 32;   // the probe passes' 64-bit pair total, spread over this many words (summed by the host)
__device__ __forceinline__ int64_t claim_slot(int64_t *tk, uint64_t cap, int64_t key) {
    if (key == kEmptyKey) return (int64_t)cap;
    uint64_t s = mix64((uint64_t)key) & (cap - 1);
    for (uint64_t probe = 0, lim = cap < kClaimProbes ? cap : kClaimProbes; probe < lim; ++probe) {
        int64_t cur = __hip_atomic_load(&tk[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmptyKey) {
            int64_t expected = kEmptyKey;
            if (__hip_atomic_compare_exchange_strong(&tk[s], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                return (int64_t)s;
            cur = expected;
        }
        if (cur == key) return (int64_t)s;
        s = (s + 1) & (cap - 1);
    }
    return -1;
}
__device__ __forceinline__ int64_t find_slot(const int64_t *tk, uint64_t cap, int64_t key) {
    if (key == kEmptyKey) return (int64_t)cap;
    uint64_t s = mix64((uint64_t)key) & (cap - 1);
    for (uint64_t probe = 0; probe < cap; ++probe) {
        const int64_t cur = tk[s];
        if (cur == kEmptyKey) return -1;
        if (cur == key) return (int64_t)s;
        s = (s + 1) & (cap - 1);
    }
    return -1;
}
__global__ __launch_bounds__(kBlock) void group_insert_kernel(const int64_t *__restrict__ keys, const uint64_t *__restrict__ values,
                                                              int32_t kind, int64_t n, int64_t *tk, uint64_t *ta, int32_t *tf, uint64_t cap,
                                                              uint32_t *err) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t key = keys[i];
        const int64_t s = claim_slot(tk, cap, key);
        if (s < 0) {
            atomicOr(err, 1u);
            continue;
        }
        if (s == (int64_t)cap) tk[s] = key;  // the dedicated slot of the sentinel key
        atomicMin(&tf[s], (int32_t)i);
        const uint64_t v = values ? values[i] : 1ull;
        if (kind == (int32_t)AggKind::SUM) atomicAdd(reinterpret_cast<unsigned long long *>(&ta[s]), (unsigned long long)v);
        else if (kind == (int32_t)AggKind::MAX) atomicMax(reinterpret_cast<unsigned long long *>(&ta[s]), (unsigned long long)v);
    }
}
// (the pass that runs behind the inserts also hands their error word to the host: `h_err` is pinned memory, read after the next wait)
__global__ __launch_bounds__(kBlock) void live_slot_mask_kernel(const int32_t *__restrict__ tf, int64_t slots, uint8_t *__restrict__ mask,
                                                                const uint32_t *__restrict__ err, uint32_t *__restrict__ h_err) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *h_err = *err;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < slots; i += (int64_t)gridDim.x * kBlock) mask[i] = tf[i] != 0x7fffffff;
}

struct AggSpecs {
    const void *values[kMaxGroupAggs];
    const uint8_t *valid[kMaxGroupAggs];   // null: every row valid
    int32_t op[kMaxGroupAggs];
    int32_t type[kMaxGroupAggs];
    int32_t n;
    uint32_t *seen;   // slots x n counters of valid contributions (null when no spec carries validity)
};
// doubles <-> unsigned keys of the same order (negative values: all bits flipped; others: the sign bit set)
__device__ __forceinline__ uint64_t f64_order_key(double d) {
    const uint64_t b = (uint64_t)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ uint64_t f64_from_order_key(uint64_t k) { return (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k; }
__device__ __forceinline__ uint64_t agg_identity(int32_t op) {
    switch (op) {
        case (int32_t)AggOp::MAX_S: return (uint64_t)INT64_MIN;
        case (int32_t)AggOp::MIN_S: return (uint64_t)INT64_MAX;
        case (int32_t)AggOp::MIN_U: return ~0ull;
        case (int32_t)AggOp::MIN_F64: return ~0ull;
        default: return 0;  // COUNT, SUM_INT, MAX_U, SUM_F64 (+0.0)
    }
}
__global__ __launch_bounds__(kBlock) void group_init_n_kernel(int64_t *__restrict__ tk, uint64_t *__restrict__ ta, int32_t *__restrict__ tf,
                                                              int64_t slots, AggSpecs sp, uint32_t *__restrict__ err) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *err = 0u;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < slots; i += (int64_t)gridDim.x * kBlock) {
        tk[i] = kEmptyKey;
        tf[i] = 0x7fffffff;
        for (int a = 0; a < sp.n; ++a) ta[i * sp.n + a] = agg_identity(sp.op[a]);
    }
}
// A row's value as the 64-bit pattern its accumulator merges (COUNT: 1; doubles as bits, or as order keys for MIN / MAX)
__device__ __forceinline__ uint64_t agg_row_value(int32_t op, const AggSpecs &sp, int a, int64_t i) {
    if (op == (int32_t)AggOp::COUNT) return 1ull;
    if (op == (int32_t)AggOp::SUM_F64) return (uint64_t)__double_as_longlong(static_cast<const double *>(sp.values[a])[i]);
    if (op == (int32_t)AggOp::MAX_F64 || op == (int32_t)AggOp::MIN_F64) return f64_order_key(static_cast<const double *>(sp.values[a])[i]);
    return (uint64_t)load_as_i64(sp.values[a], sp.type[a], i);
}
// Merges a partial result `v` (a row's value, a wave's reduction, a workgroup's LDS accumulator) into accumulator `acc`: sums and counts
// add, minima and maxima compare -- the same operation at every level, so a partial of partials is a partial.
__device__ __forceinline__ void agg_merge(uint64_t *acc, int32_t op, uint64_t v) {
    switch (op) {
        case (int32_t)AggOp::COUNT:
        case (int32_t)AggOp::SUM_INT: atomicAdd(reinterpret_cast<unsigned long long *>(acc), (unsigned long long)v); break;
        case (int32_t)AggOp::SUM_F64: atomicAdd(reinterpret_cast<double *>(acc), __longlong_as_double((long long)v)); break;
        case (int32_t)AggOp::MAX_F64:
        case (int32_t)AggOp::MAX_U: atomicMax(reinterpret_cast<unsigned long long *>(acc), (unsigned long long)v); break;
        case (int32_t)AggOp::MIN_F64:
        case (int32_t)AggOp::MIN_U: atomicMin(reinterpret_cast<unsigned long long *>(acc), (unsigned long long)v); break;
        case (int32_t)AggOp::MAX_S: atomicMax(reinterpret_cast<long long *>(acc), (long long)v); break;
        default: atomicMin(reinterpret_cast<long long *>(acc), (long long)v); break;   // MIN_S
    }
}
// The same merge between two plain values (the butterfly of a wave's reduction)
__device__ __forceinline__ uint64_t agg_combine(int32_t op, uint64_t x, uint64_t y) {
    switch (op) {
        case (int32_t)AggOp::COUNT:
        case (int32_t)AggOp::SUM_INT: return x + y;
        case (int32_t)AggOp::SUM_F64: return (uint64_t)__double_as_longlong(__longlong_as_double((long long)x) + __longlong_as_double((long long)y));
        case (int32_t)AggOp::MAX_F64:
        case (int32_t)AggOp::MAX_U: return x > y ? x : y;
        case (int32_t)AggOp::MIN_F64:
        case (int32_t)AggOp::MIN_U: return x < y ? x : y;
        case (int32_t)AggOp::MAX_S: return (int64_t)x > (int64_t)y ? x : y;
        default: return (int64_t)x < (int64_t)y ? x : y;
    }
}
__device__ __forceinline__ uint64_t wave_bcast_u64(uint64_t v, int src) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32);
}
__device__ __forceinline__ uint64_t wave_xor_u64(uint64_t v, int o) {
    return (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)v, o, 64) | ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64) << 32);
}

// GROUP BY with N aggregates into one open-addressing table, in three levels of the same merge (agg_merge):
//   wave      : half of NEXMark's bids name one auction, and 4.6e6 rows whose claim, first-row minimum and accumulator updates all land on
//               ONE slot serialise in L2 (2.7 ms per 9.2e6 bids, q5 on the generic operators).  Every wave looks at the key of its first
//               row: when at least kCombineMin of its 64 rows carry it, those rows are reduced inside the wave and one lane goes on for them;
//   workgroup : (kLds) a workgroup owns a contiguous run of rows and a small open-addressing table in LDS -- keys, first rows,
//               accumulators, counts of valid contributions -- that takes whatever finds a slot within a few probes (keys cluster in time:
//               an auction's bids, a bidder's session); what does not goes to the global table directly; the LDS table is flushed once;
//   global    : claim the key's slot, minimum of the first rows, merge.
constexpr int kCombineMin = 8;
constexpr int kLdsGroupProbes = 8;
struct GroupTable {
    int64_t *tk;
    uint64_t *ta;
    int32_t *tf;
    uint64_t cap;
    uint32_t *err;
    const uint8_t *key_valid;   // null: no NULL keys; else the rows whose key is NULL share ONE group, slot cap + 1 (slot cap: the key INT64_MIN)
};
__device__ __forceinline__ void group_update_global(const GroupTable &g, const AggSpecs &sp, int64_t key, bool key_null, int32_t first, const uint64_t *val,
                                                    const uint32_t *cnt) {
    const int64_t s = key_null ? (int64_t)g.cap + 1 : claim_slot(g.tk, g.cap, key);
    if (s < 0) {
        atomicOr(g.err, 1u);
        return;
    }
    if (s == (int64_t)g.cap) g.tk[s] = key;  // the dedicated slot of the sentinel key
    atomicMin(&g.tf[s], first);
    for (int a = 0; a < sp.n; ++a) {
        if (cnt[a] == 0) continue;   // (nothing but NULLs reached this partial)
        if (sp.valid[a]) atomicAdd(&sp.seen[s * sp.n + a], cnt[a]);
        agg_merge(&g.ta[s * sp.n + a], sp.op[a], val[a]);
    }
}
template <bool kLds>
__global__ __launch_bounds__(kBlock) void group_insert_n_kernel(const int64_t *__restrict__ keys, int64_t n, AggSpecs sp, GroupTable g, int64_t rows_per_wg,
                                                                uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_mem[];   // kLds: keys [S] | accumulators [S * n] | first rows [S] | valid counts [S * n]
    int64_t *s_key = reinterpret_cast<int64_t *>(s_mem);
    uint64_t *s_acc = s_mem + lds_slots;
    int32_t *s_first = reinterpret_cast<int32_t *>(s_acc + (size_t)lds_slots * sp.n);
    uint32_t *s_seen = reinterpret_cast<uint32_t *>(s_first + lds_slots);
    const int lane = lane_id();
    if (kLds) {
        for (uint32_t i = threadIdx.x; i < lds_slots; i += kBlock) {
            s_key[i] = kEmptyKey;
            s_first[i] = 0x7fffffff;
            for (int a = 0; a < sp.n; ++a) {
                s_acc[i * sp.n + a] = agg_identity(sp.op[a]);
                s_seen[i * sp.n + a] = 0;
            }
        }
        __syncthreads();
    }
    // a partial -- one row or a wave's reduction -- goes into the workgroup's table when it finds room there, else to the global one
    auto update = [&](int64_t key, bool key_null, int32_t first, const uint64_t *val, const uint32_t *cnt) {
        if (kLds && key != kEmptyKey && !key_null) {
            uint32_t s = (uint32_t)mix64((uint64_t)key) & (lds_slots - 1);
            for (int probe = 0; probe < kLdsGroupProbes; ++probe) {
                int64_t cur = __hip_atomic_load(&s_key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == kEmptyKey) {
                    int64_t expected = kEmptyKey;
                    if (__hip_atomic_compare_exchange_strong(&s_key[s], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) cur = key;
                    else cur = expected;
                }
                if (cur == key) {
                    atomicMin(&s_first[s], first);
                    for (int a = 0; a < sp.n; ++a) {
                        if (cnt[a] == 0) continue;
                        atomicAdd(&s_seen[s * sp.n + a], cnt[a]);
                        agg_merge(&s_acc[s * sp.n + a], sp.op[a], val[a]);
                    }
                    return;
                }
                s = (s + 1) & (lds_slots - 1);
            }
        }
        group_update_global(g, sp, key, key_null, first, val, cnt);
    };
    const int64_t lo = kLds ? (int64_t)blockIdx.x * rows_per_wg : 0, hi = kLds ? (lo + rows_per_wg < n ? lo + rows_per_wg : n) : n;
    const int64_t start = kLds ? lo + (threadIdx.x & ~63) : (int64_t)blockIdx.x * kBlock + (threadIdx.x & ~63);
    const int64_t stride = kLds ? kBlock : (int64_t)gridDim.x * kBlock;
    for (int64_t base = start; base < hi; base += stride) {   // (wave-uniform)
        // a table that filled up (sized from a hint the data outgrew) is known to every wave one load later: the pass is void, and its
        // remaining rows would each walk kClaimProbes occupied slots before giving up
        if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
        const int64_t i = base + lane;
        const bool live = i < hi;
        const bool knull = live && g.key_valid && !g.key_valid[i];
        const int64_t key = live && !knull ? keys[i] : 0;   // (a NULL's key bits mean nothing: all NULLs are one key)
        uint64_t val[kMaxGroupAggs];
        uint32_t cnt[kMaxGroupAggs];
        for (int a = 0; a < sp.n; ++a) {
            const bool ok = live && (!sp.valid[a] || sp.valid[a][i]);   // a NULL reaches no accumulator
            cnt[a] = ok ? 1u : 0u;
            val[a] = ok ? agg_row_value(sp.op[a], sp, a, i) : agg_identity(sp.op[a]);
        }
        const int64_t k0 = (int64_t)wave_bcast_u64((uint64_t)key, 0);   // (lane 0 is live: base < hi)
        const bool knull0 = __builtin_amdgcn_readlane((int)knull, 0) != 0;
        const uint64_t same = __ballot(live && key == k0 && knull == knull0);
        const bool combine = __popcll((unsigned long long)same) >= kCombineMin;
        const bool in_group = combine && ((same >> lane) & 1);
        if (combine) {   // the rows that share the first row's key: one partial, carried on by lane 0 (which holds the smallest row number)
            uint64_t gv[kMaxGroupAggs];
            uint32_t gc[kMaxGroupAggs];
            for (int a = 0; a < sp.n; ++a) {
                uint64_t v = in_group && cnt[a] ? val[a] : agg_identity(sp.op[a]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v = agg_combine(sp.op[a], v, wave_xor_u64(v, o));
                gv[a] = v;
                gc[a] = (uint32_t)__popcll((unsigned long long)__ballot(in_group && cnt[a]));
            }
            if (lane == 0) update(k0, knull0, (int32_t)i, gv, gc);
        }
        if (live && !in_group) update(key, knull, (int32_t)i, val, cnt);
    }
    if (!kLds) return;
    __syncthreads();
    if (__hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // (void pass: nothing worth flushing)
    for (uint32_t s = threadIdx.x; s < lds_slots; s += kBlock) {
        const int64_t key = s_key[s];
        if (key == kEmptyKey) continue;
        uint64_t val[kMaxGroupAggs];
        uint32_t cnt[kMaxGroupAggs];
        for (int a = 0; a < sp.n; ++a) {
            val[a] = s_acc[s * sp.n + a];
            cnt[a] = s_seen[s * sp.n + a];
        }
        group_update_global(g, sp, key, false, s_first[s], val, cnt);
    }
}
// key_valid[g] = 0 for the group that sits in slot `null_slot` (the NULL keys' group)
__global__ __launch_bounds__(kBlock) void key_valid_from_slot_kernel(const int32_t *__restrict__ slot_rows, int64_t n_groups, int32_t null_slot,
                                                                     uint8_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_groups; i += (int64_t)gridDim.x * kBlock) out[i] = slot_rows[i] != null_slot ? 1 : 0;
}
// out[a][g] = ta[slot_rows[g] * n + a]
struct AggOuts {
    uint64_t *out[kMaxGroupAggs];
    uint8_t *valid[kMaxGroupAggs];
};
__global__ __launch_bounds__(kBlock) void group_collect_n_kernel(const uint64_t *__restrict__ ta, const int32_t *__restrict__ slot_rows,
                                                                 int64_t n_groups, AggSpecs sp, AggOuts o) {
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * kBlock)
        for (int a = 0; a < sp.n; ++a) {
            const uint64_t v = ta[(int64_t)slot_rows[g] * sp.n + a];
            o.out[a][g] = (sp.op[a] == (int32_t)AggOp::MAX_F64 || sp.op[a] == (int32_t)AggOp::MIN_F64) ? f64_from_order_key(v) : v;
            if (o.valid[a]) o.valid[a][g] = sp.seen[(int64_t)slot_rows[g] * sp.n + a] ? 1 : 0;
        }
}
__global__ __launch_bounds__(kBlock) void gather_u8_kernel(const uint8_t *__restrict__ src, const int32_t *__restrict__ rows, int64_t n, uint8_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = src[rows[i]];
}
__global__ __launch_bounds__(kBlock) void replace_invalid_kernel(int64_t *__restrict__ keys, const uint8_t *__restrict__ valid, int64_t n, int64_t sentinel) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (!valid[i]) keys[i] = sentinel;
}
__global__ __launch_bounds__(kBlock) void valid_from_i64_kernel(const int64_t *__restrict__ keys, int64_t n, int64_t sentinel, uint8_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = keys[i] != sentinel ? 1 : 0;
}
__global__ __launch_bounds__(kBlock) void zero_u32_kernel(uint32_t *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = 0;
}
__global__ __launch_bounds__(kBlock) void pack_pair_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int64_t n,
                                                           int64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = (int64_t)(((uint64_t)(uint32_t)a[i] << 32) | (uint64_t)(uint32_t)b[i]);
}
__global__ __launch_bounds__(kBlock) void unpack_pair_kernel(const int64_t *__restrict__ k, int64_t n, int32_t *__restrict__ a,
                                                             int32_t *__restrict__ b) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        a[i] = (int32_t)((uint64_t)k[i] >> 32);
        b[i] = (int32_t)(uint32_t)k[i];
    }
}
__global__ __launch_bounds__(kBlock) void i64_to_f64_kernel(const int64_t *__restrict__ in, int64_t n, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (double)in[i];
}
__global__ __launch_bounds__(kBlock) void avg_finish_kernel(const double *__restrict__ sum, const uint64_t *__restrict__ count, int64_t n,
                                                            double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = sum[i] / (double)count[i];
}


// ---- column statistics: per-block minimum / maximum of an integer column (signed order, or unsigned for UInt64), folded on the host
__global__ __launch_bounds__(kBlock) void minmax_kernel(const void *__restrict__ v, int32_t type, int64_t n, int64_t *__restrict__ block_out) {
    __shared__ int64_t s_mn[kWavesPerBlock], s_mx[kWavesPerBlock];
    const bool uns = type == (int32_t)ColType::U64;
    const uint64_t flip = uns ? 0ull : (1ull << 63);   // both orders as unsigned order
    uint64_t mn = ~0ull, mx = 0ull;
    if (type == (int32_t)ColType::I32) {
        const int32_t *col = static_cast<const int32_t *>(v);
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
            const int4 t = *reinterpret_cast<const int4 *>(col + 4 * i);
            const int32_t lo = min(min(t.x, t.y), min(t.z, t.w)), hi = max(max(t.x, t.y), max(t.z, t.w));
            const uint64_t a = (uint64_t)(int64_t)lo ^ flip, b = (uint64_t)(int64_t)hi ^ flip;
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
            const uint64_t a = (uint64_t)(int64_t)col[4 * n4 + threadIdx.x] ^ flip;
            mn = a < mn ? a : mn;
            mx = a > mx ? a : mx;
        }
    } else {
        const int64_t *col = static_cast<const int64_t *>(v);
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
            const uint64_t a = (uint64_t)col[i] ^ flip;
            mn = a < mn ? a : mn;
            mx = a > mx ? a : mx;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = wave_xor_u64(mn, o), b = wave_xor_u64(mx, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (lane_id() == 0) {
        s_mn[threadIdx.x >> 6] = (int64_t)mn;
        s_mx[threadIdx.x >> 6] = (int64_t)mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            mn = (uint64_t)s_mn[w] < mn ? (uint64_t)s_mn[w] : mn;
            mx = (uint64_t)s_mx[w] > mx ? (uint64_t)s_mx[w] : mx;
        }
        block_out[blockIdx.x] = (int64_t)(mn ^ flip);
        block_out[gridDim.x + blockIdx.x] = (int64_t)(mx ^ flip);
    }
}

// ---- GROUP BY over a DENSE integer key (group_by_dense): the key's offset from the column minimum IS its slot -- no hashing, no claim,
// no probing.  One workgroup per 8192 consecutive rows:
//   * the tile's slots and their range; when the range fits kDenseBins (NEXMark: an 8192-bid tile names ~600 consecutive auctions; any
//     stream whose keys cluster in time) the tile is aggregated in LDS and flushed as one run of consecutive slots -- a few cache lines of
//     fire-and-forget atomics per wave instruction; a wider tile sends every row's update to the global table directly;
//   * the rows that carry the wave's most frequent slot (half of NEXMark's bids name one auction) never reach an atomic: counted by
//     ballot + popcount, their values folded per lane and reduced once per wave.
constexpr int kDenseBins = 2048;
constexpr int kDenseTile = 8192;
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
// kAcc: some accumulator besides the row count (COUNT-only GROUP BYs -- the reference's arch/ops/group-by.sql, q5's inner query -- run the
// instance without the accumulator paths' registers)
template <bool kKey64, bool kAcc>
__global__ __launch_bounds__(kBlock) void dense_group_kernel(const void *__restrict__ keys, int64_t n_rows, int64_t kmin, uint32_t range, AggSpecs sp,
                                                             uint32_t *__restrict__ g_cnt, uint64_t *__restrict__ g_acc, uint32_t *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_mem[];   // accumulators [n_acc][kDenseBins] | row counts [kDenseBins]
    uint64_t *s_acc = s_mem;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_mem + (size_t)(kAcc ? sp.n : 0) * kDenseBins);
    __shared__ uint32_t s_red[2][kWavesPerBlock];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int n_acc = kAcc ? sp.n : 0;
    for (uint32_t i = threadIdx.x; i < kDenseBins; i += kBlock) {
        s_cnt[i] = 0;
        for (int a = 0; a < n_acc; ++a) s_acc[(size_t)a * kDenseBins + i] = agg_identity(sp.op[a]);
    }
    // the tile's slots: wave w holds rows [t0 + w * 2048, + 2048), lane l of iteration it the four rows at + it * 256 + 4 l
    const int64_t t0 = (int64_t)blockIdx.x * kDenseTile + wave * (kDenseTile / kWavesPerBlock) + lane * 4;
    constexpr uint32_t kNone = 0xffffffffu;
    uint32_t s[kFlagIters][4];
    uint32_t mn = kNone, mx = 0, bad = 0;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        const int64_t r0 = t0 + it * 256;
        uint32_t d[4];      // key - kmin (mod 2^32)
        uint32_t wide = 0;  // bit j: the 64-bit difference does not fit 32 bits (64-bit keys only; an Int32 column's differences always do)
        if (kKey64) {
            uint64_t k[4];
            if (r0 + 4 <= n_rows) {
                const int4 a = *reinterpret_cast<const int4 *>(static_cast<const int64_t *>(keys) + r0), b = *reinterpret_cast<const int4 *>(static_cast<const int64_t *>(keys) + r0 + 2);
                k[0] = ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x;
                k[1] = ((uint64_t)(uint32_t)a.w << 32) | (uint32_t)a.z;
                k[2] = ((uint64_t)(uint32_t)b.y << 32) | (uint32_t)b.x;
                k[3] = ((uint64_t)(uint32_t)b.w << 32) | (uint32_t)b.z;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) k[j] = r0 + j < n_rows ? (uint64_t) static_cast<const int64_t *>(keys)[r0 + j] : (uint64_t)kmin;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t x = k[j] - (uint64_t)kmin;
                d[j] = (uint32_t)x;
                wide |= (uint32_t)((x >> 32) != 0) << j;
            }
        } else {
            const int32_t *col = static_cast<const int32_t *>(keys);
            int32_t k[4];
            if (r0 + 4 <= n_rows) {
                const int4 a = stream_load4(col + r0);
                k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) k[j] = r0 + j < n_rows ? col[r0 + j] : (int32_t)kmin;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = (uint32_t)k[j] - (uint32_t)(int32_t)kmin;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool live = r0 + j < n_rows;
            const bool in = d[j] < range && !((wide >> j) & 1u);
            bad |= (uint32_t)(live && !in);
            const uint32_t sl = live && in ? d[j] : kNone;
            s[it][j] = sl;
            mn = sl < mn ? sl : mn;
            mx = sl != kNone && sl > mx ? sl : mx;
        }
    }
    if (__ballot(bad != 0) && lane == 0) atomicOr(err, 1u);   // (a key outside the statistics the table was sized from: the call is void)
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    if (lane == 0) {
        s_red[0][wave] = mn;
        s_red[1][wave] = mx;
    }
    __syncthreads();   // (also: the bins are initialised)
    uint32_t tmin = s_red[0][0], tmax = s_red[1][0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) {
        tmin = s_red[0][w] < tmin ? s_red[0][w] : tmin;
        tmax = s_red[1][w] > tmax ? s_red[1][w] : tmax;
    }
    if (tmin == kNone) return;   // no live row in the tile (block-uniform)
    const bool in_lds = tmax - tmin < (uint32_t)kDenseBins;   // block-uniform
    // The wave's hot slot, kept across iterations and re-elected when it goes cold (NEXMark's hot auction moves on every hundred auctions:
    // a wave's 2048 rows see one or two of them).  Its rows never reach an atomic: counted by ballot + popcount, their values folded per
    // lane; when the slot changes -- and at the end -- the wave hands the partial over in one update.
    uint32_t hot = kNone, hot_cnt = 0;
    uint64_t hv[kMaxGroupAggs];
    for (int a = 0; a < n_acc; ++a) hv[a] = agg_identity(sp.op[a]);
    auto flush_hot = [&]() {   // (wave-uniform)
        if (!hot_cnt) return;
        for (int a = 0; a < n_acc; ++a) {
            uint64_t v = hv[a];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = agg_combine(sp.op[a], v, wave_xor_u64(v, o));
            hv[a] = v;
        }
        if (lane == 0) {
            if (in_lds) {
                atomicAdd(&s_cnt[hot - tmin], hot_cnt);
                for (int a = 0; a < n_acc; ++a) agg_merge(&s_acc[(size_t)a * kDenseBins + (hot - tmin)], sp.op[a], hv[a]);
            } else {
                atomicAdd(&g_cnt[hot], hot_cnt);
                for (int a = 0; a < n_acc; ++a) agg_merge(&g_acc[(size_t)a * range + hot], sp.op[a], hv[a]);
            }
        }
        hot_cnt = 0;
        for (int a = 0; a < n_acc; ++a) hv[a] = agg_identity(sp.op[a]);
    };
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        const int64_t r0 = t0 + it * 256;
        if (hot == kNone || __popcll((unsigned long long)__ballot(s[it][0] == hot)) < 8) {   // (wave-uniform) cold, or none yet: the best of three candidates
            flush_hot();
            hot = kNone;
            int best = 7;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)s[it][0], c * 21);
                const int n = cand == kNone ? 0 : __popcll((unsigned long long)__ballot(s[it][0] == cand));
                if (n > best) {
                    best = n;
                    hot = cand;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t sl = s[it][j];
            const bool live = sl != kNone;
            const bool is_hot = live && sl == hot;
            hot_cnt += (uint32_t)__popcll((unsigned long long)__ballot(is_hot));
            if (kAcc) {
                uint64_t val[kMaxGroupAggs];
                for (int a = 0; a < n_acc; ++a) val[a] = live ? agg_row_value(sp.op[a], sp, a, r0 + j) : agg_identity(sp.op[a]);
                if (is_hot)
                    for (int a = 0; a < n_acc; ++a) hv[a] = agg_combine(sp.op[a], hv[a], val[a]);
                if (live && !is_hot) {
                    if (in_lds) {
                        atomicAdd(&s_cnt[sl - tmin], 1u);
                        for (int a = 0; a < n_acc; ++a) agg_merge(&s_acc[(size_t)a * kDenseBins + (sl - tmin)], sp.op[a], val[a]);
                    } else {
                        atomicAdd(&g_cnt[sl], 1u);
                        for (int a = 0; a < n_acc; ++a) agg_merge(&g_acc[(size_t)a * range + sl], sp.op[a], val[a]);
                    }
                }
            } else if (live && !is_hot) {
                if (in_lds) atomicAdd(&s_cnt[sl - tmin], 1u);
                else atomicAdd(&g_cnt[sl], 1u);
            }
        }
    }
    flush_hot();
    if (!in_lds) return;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b <= tmax - tmin; b += kBlock) {
        const uint32_t c = s_cnt[b];
        if (!c) continue;
        atomicAdd(&g_cnt[tmin + b], c);
        for (int a = 0; a < n_acc; ++a) agg_merge(&g_acc[(size_t)a * range + tmin + b], sp.op[a], s_acc[(size_t)a * kDenseBins + b]);
    }
}
// live slots (count != 0) as flag words in the flag-tile geometry: a lane's four consecutive slots are one 16-byte load
__global__ __launch_bounds__(kBlock) void dense_live_flag_kernel(const uint32_t *__restrict__ cnt, int64_t n_slots, SegTiles st, uint32_t *__restrict__ flag_words,
                                                                 uint32_t *__restrict__ counts, const uint32_t *__restrict__ err, uint32_t *__restrict__ h_err) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *h_err = *err;   // (the count pass's error word, handed to the host's pinned copy by the pass behind it)
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const int64_t wbase = tr.tile_begin + flag_rel0();
    uint32_t flags = 0;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        const int64_t r0 = wbase + it * 256;
        if (r0 + 4 <= n_slots) {
            const uint4 t = *reinterpret_cast<const uint4 *>(cnt + r0);
            flags |= ((uint32_t)(t.x != 0) | ((uint32_t)(t.y != 0) << 1) | ((uint32_t)(t.z != 0) << 2) | ((uint32_t)(t.w != 0) << 3)) << (it * 4);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) flags |= (uint32_t)(r0 + j < n_slots && cnt[r0 + j] != 0) << (it * 4 + j);
        }
    }
    store_flags_and_counts(flags, tile, flag_words, counts);
}
struct DenseOuts {
    uint64_t *out[kMaxGroupAggs];
    int32_t acc_of[kMaxGroupAggs];   // -1: the row count itself (COUNT)
    int32_t n;
};
__global__ __launch_bounds__(kBlock) void dense_collect_kernel(const int32_t *__restrict__ slots, int64_t n_groups, int64_t kmin, uint32_t range,
                                                               const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ acc, int64_t *__restrict__ keys,
                                                               DenseOuts o) {
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * kBlock) {
        const uint32_t sl = (uint32_t)slots[g];
        keys[g] = (int64_t)((uint64_t)kmin + sl);
        for (int a = 0; a < o.n; ++a) o.out[a][g] = o.acc_of[a] < 0 ? (uint64_t)cnt[sl] : acc[(size_t)o.acc_of[a] * range + sl];
    }
}
__global__ __launch_bounds__(kBlock) void fill_u64_kernel(uint64_t *__restrict__ p, int64_t n, uint64_t v) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}

// ---- distinct (int32, Utf8)
__device__ __forceinline__ bool same_pair(const int32_t *key, const int32_t *off, const uint8_t *bytes, int32_t a, int32_t b) {
    if (key[a] != key[b]) return false;
    const int32_t la = off[a + 1] - off[a], lb = off[b + 1] - off[b];
    if (la != lb) return false;
    const uint8_t *pa = bytes + off[a], *pb = bytes + off[b];
    for (int32_t k = 0; k < la; ++k)
        if (pa[k] != pb[k]) return false;
    return true;
}
__global__ __launch_bounds__(kBlock) void fill_i32_kernel(int32_t *__restrict__ p, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}
// table[cap] (filled with -1 like the slots): stays negative while the keys are strictly increasing -- then every (key, text) pair is its own
// representative and nothing needs hashing (q8's persons arrive in id order: 27 us of byte-wise hashing per window for a DISTINCT that keeps every row)
__global__ __launch_bounds__(kBlock) void distinct_order_check_kernel(const int32_t *__restrict__ key, int64_t n, int32_t *flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) bad |= i > 0 && key[i] <= key[i - 1];
    if (__ballot(bad) && lane_id() == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(kBlock) void distinct_insert_kernel(const int32_t *__restrict__ key, const int32_t *__restrict__ off,
                                                                 const uint8_t *__restrict__ bytes, int64_t n, int32_t *table, uint64_t cap,
                                                                 uint8_t *__restrict__ is_rep) {
    if (table[cap] < 0) {   // (grid-uniform) strictly increasing keys: every row stays
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) is_rep[i] = 1;
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t h = 0xCBF29CE484222325ull ^ (uint64_t)(uint32_t)key[i];
        for (int32_t b = off[i]; b < off[i + 1]; ++b) h = (h ^ bytes[b]) * 0x100000001B3ull;
        uint64_t s = mix64(h) & (cap - 1);
        uint8_t rep = 0;
        for (uint64_t probe = 0; probe < cap; ++probe) {
            int32_t cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur < 0) {
                int32_t expected = -1;
                if (__hip_atomic_compare_exchange_strong(&table[s], &expected, (int32_t)i, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT)) {
                    rep = 1;
                    break;
                }
                cur = expected;
            }
            if (same_pair(key, off, bytes, cur, (int32_t)i)) break;
            s = (s + 1) & (cap - 1);
        }
        is_rep[i] = rep;
    }
}

// ---- Utf8 keys
__device__ __forceinline__ uint64_t hash_bytes(const uint8_t *p, int32_t len) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (int32_t b = 0; b < len; ++b) h = (h ^ p[b]) * 0x100000001B3ull;
    return mix64(h ^ (uint64_t)(uint32_t)len);
}
__device__ __forceinline__ bool same_bytes(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb) {
    if (la != lb) return false;
    for (int32_t k = 0; k < la; ++k)
        if (a[k] != b[k]) return false;
    return true;
}
__global__ __launch_bounds__(kBlock) void hash_utf8_kernel(const int32_t *__restrict__ off, const uint8_t *__restrict__ bytes, int64_t n,
                                                           int64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = (int64_t)hash_bytes(bytes + off[i], off[i + 1] - off[i]);
}
__global__ __launch_bounds__(kBlock) void utf8_codes_build_kernel(const int32_t *__restrict__ off, const uint8_t *__restrict__ bytes, int64_t n,
                                                                  int32_t *table, uint64_t cap, int64_t *__restrict__ codes) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint8_t *me = bytes + off[i];
        const int32_t len = off[i + 1] - off[i];
        uint64_t s = hash_bytes(me, len) & (cap - 1);
        int64_t code = i;
        for (uint64_t probe = 0; probe < cap; ++probe) {
            int32_t cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur < 0) {
                int32_t expected = -1;
                if (__hip_atomic_compare_exchange_strong(&table[s], &expected, (int32_t)i, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    break;
                cur = expected;
            }
            if (same_bytes(me, len, bytes + off[cur], off[cur + 1] - off[cur])) {
                code = cur;
                break;
            }
            s = (s + 1) & (cap - 1);
        }
        if (codes) codes[i] = code;
    }
}
__global__ __launch_bounds__(kBlock) void utf8_codes_probe_kernel(const int32_t *__restrict__ boff, const uint8_t *__restrict__ bbytes,
                                                                  const int32_t *__restrict__ table, uint64_t cap,
                                                                  const int32_t *__restrict__ off, const uint8_t *__restrict__ bytes, int64_t n,
                                                                  int64_t *__restrict__ codes) {
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
        const uint8_t *me = bytes + off[j];
        const int32_t len = off[j + 1] - off[j];
        uint64_t s = hash_bytes(me, len) & (cap - 1);
        int64_t code = -(j + 2);
        for (uint64_t probe = 0; probe < cap; ++probe) {
            const int32_t cur = table[s];
            if (cur < 0) break;
            if (same_bytes(me, len, bbytes + boff[cur], boff[cur + 1] - boff[cur])) {
                code = cur;
                break;
            }
            s = (s + 1) & (cap - 1);
        }
        codes[j] = code;
    }
}

// ---- max: per-workgroup maxima (signed or unsigned order), folded on the host
// block_out[b] = the block's maximum over its VALID rows, block_out[gridDim.x + b] = how many of them it saw
__global__ __launch_bounds__(kBlock) void max_kernel(const void *__restrict__ v, const uint8_t *__restrict__ valid, int32_t type, int64_t n,
                                                     int64_t *__restrict__ block_out) {
    __shared__ int64_t s_m[kWavesPerBlock];
    __shared__ int64_t s_c[kWavesPerBlock];
    const bool uns = type == (int32_t)ColType::U64;
    int64_t m = uns ? 0 : INT64_MIN, cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        if (valid && !valid[i]) continue;   // MAX ignores NULLs
        ++cnt;
        const int64_t x = load_as_i64(v, type, i);
        if (cmp_i64(x, m, 4, uns)) m = x;
    }
    cnt = (int64_t)wave_sum_u64((uint64_t)cnt);
    if (lane_id() == 0) s_c[threadIdx.x >> 6] = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t t = __shfl_xor(m, o, 64);
        if (cmp_i64(t, m, 4, uns)) m = t;
    }
    if (lane_id() == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w)
            if (cmp_i64(s_m[w], m, 4, uns)) m = s_m[w];
        block_out[blockIdx.x] = m;
        block_out[gridDim.x + blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    }
}

// mask[i] = 1 where position i of the sorted order begins a new key (HashDiff's partition boundaries)
__global__ __launch_bounds__(kBlock) void key_run_start_kernel(const int64_t *__restrict__ keys, const int32_t *__restrict__ order, int64_t n, uint8_t *__restrict__ mask) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        mask[i] = i == 0 || keys[order[i]] != keys[order[i - 1]] ? 1 : 0;
}

// ---- join on a hashed key (join_hashed): an open-addressing table of 16-byte slots -- the key, the build row that claimed the slot, the head of
// the chain of the key's FURTHER build rows -- so that a probe is ONE 16-byte read at a hashed position (plus the linear-probe steps of a table at
// most half full), and a build row of a key nobody else holds is one compare-and-swap and a plain store.  (Rounds 2-5 kept keys, chain heads and
// links in three arrays: a claim, an exchange and a link store per build row, three dependent reads per probe row, the chain walked twice.)
struct __attribute__((aligned(16))) JoinSlot {
    int64_t key;     // kEmptyKey: free (the key INT64_MIN itself lives in the extra slot `cap`, claimed through `first`)
    int32_t first;   // the build row whose insert claimed the slot (written by that insert alone); -1 while free
    int32_t head;    // chain of the key's other build rows (push front, links in next[]); -1: none -- the build keys seen so far are unique
};
__global__ __launch_bounds__(kBlock) void join_hash_init_kernel(JoinSlot *__restrict__ slots, int64_t n_slots) {
    const int4 e = make_int4((int32_t)(uint32_t)(uint64_t)kEmptyKey, (int32_t)((uint64_t)kEmptyKey >> 32), -1, -1);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * kBlock) reinterpret_cast<int4 *>(slots)[i] = e;
}
__global__ __launch_bounds__(kBlock) void join_hash_build_kernel(const void *__restrict__ keys, int32_t type, int64_t n, JoinSlot *slots, uint64_t cap,
                                                                 int32_t *__restrict__ next, uint32_t *err) {
    bool dup = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t key = load_as_i64(keys, type, i);
        uint64_t s = cap;
        bool mine = false, placed = true;
        if (key == kEmptyKey) {
            mine = atomicCAS(&slots[cap].first, -1, (int32_t)i) == -1;
            if (mine) continue;   // (`first` holds the row already)
        } else {
            placed = false;
            s = mix64((uint64_t)key) & (cap - 1);
            for (uint64_t probe = 0; probe < cap; ++probe) {
                // (the swap first, no look before it: in a table at most half full most home slots are free, and a failed swap returns what a
                // load would have -- one trip to the memory side per step instead of two)
                int64_t cur = kEmptyKey;
                if (__hip_atomic_compare_exchange_strong(&slots[s].key, &cur, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    mine = placed = true;
                    break;
                }
                if (cur == key) {
                    placed = true;
                    break;
                }
                s = (s + 1) & (cap - 1);
            }
        }
        if (!placed) {   // (cannot happen in a table at most half full)
            atomicOr(err, 1u);
            continue;
        }
        if (mine) slots[s].first = (int32_t)i;   // only the claiming insert writes here; read after the kernel boundary
        else {
            next[i] = atomicExch(&slots[s].head, (int32_t)i);  // push front: the chain is walked only after the kernel boundary
            dup = true;
        }
    }
    if (dup) atomicOr(err + 1, 1u);   // (err[1]: some key occurs twice on the build side -- the probe then counts and walks chains)
}
// the slot of `key`: its claiming row (>= 0) and chain head, or -1
__device__ __forceinline__ int32_t join_hash_find(const JoinSlot *__restrict__ slots, uint64_t cap, int64_t key, int4 v, uint64_t s, int32_t *head) {
    // v = the slot at s, already loaded (callers issue several rows' first loads together)
    for (uint64_t probe = 0; probe < cap; ++probe) {
        const int64_t k = (int64_t)((uint64_t)(uint32_t)v.x | ((uint64_t)(uint32_t)v.y << 32));
        if (k == key) {
            *head = v.w;
            return v.z;
        }
        if (k == kEmptyKey) return -1;
        s = (s + 1) & (cap - 1);
        v = reinterpret_cast<const int4 *>(slots)[s];
    }
    return -1;
}
__device__ __forceinline__ uint64_t join_hash_home(uint64_t cap, int64_t key) { return key == kEmptyKey ? cap : mix64((uint64_t)key) & (cap - 1); }
__device__ __forceinline__ int32_t join_hash_lookup(const JoinSlot *__restrict__ slots, uint64_t cap, int64_t key, int32_t *head) {
    const uint64_t s = join_hash_home(cap, key);
    const int4 v = reinterpret_cast<const int4 *>(slots)[s];
    if (key == kEmptyKey) {
        *head = v.w;
        return v.z;
    }
    return join_hash_find(slots, cap, key, v, s, head);
}
// ---- probe of a hashed table whose build keys are UNIQUE: as join_probe_unique_flag_kernel below -- the join is a filter of the probe side in the
// flag-tile geometry -- with the matching build row of every probe row stored beside the flags (match[row], coalesced), so that the build side's
// row list is one gather of `match` at the emitted rows instead of a second walk through the table.  A lane's four rows of one iteration have their
// slots requested together: four independent 16-byte reads in flight per lane before the first is looked at.
template <bool kI32>
__global__ __launch_bounds__(kBlock) void join_hash_probe_flag_kernel(const void *__restrict__ keys, int64_t n, SegTiles st, const JoinSlot *__restrict__ slots, uint64_t cap,
                                                                      uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts, int32_t *__restrict__ match) {
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const int64_t wbase = tr.tile_begin + flag_rel0();
    uint32_t flags = 0;
    int32_t a[kFlagIters][4];
    if (kI32) load_flag_tile(static_cast<const int32_t *>(keys), n, tr, a);
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        int64_t key[4];
        uint64_t home[4];
        int4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = wbase + it * 256 + j;
            key[j] = kI32 ? (int64_t)a[it][j] : (r < n ? static_cast<const int64_t *>(keys)[r] : 0);
            home[j] = join_hash_home(cap, key[j]);
            v[j] = reinterpret_cast<const int4 *>(slots)[home[j]];
        }
        int32_t m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = wbase + it * 256 + j;
            int32_t head;
            m[j] = key[j] == kEmptyKey ? v[j].z : join_hash_find(slots, cap, key[j], v[j], home[j], &head);
            if (r >= n) m[j] = -1;
            flags |= (uint32_t)(m[j] >= 0) << (it * 4 + j);
        }
        const int64_t r0 = wbase + it * 256;
        if (r0 + 4 <= n) *reinterpret_cast<int4 *>(match + r0) = make_int4(m[0], m[1], m[2], m[3]);   // (r0 is a multiple of 4, `match` 16-byte aligned)
        else
            for (int j = 0; j < 4; ++j)
                if (r0 + j < n) match[r0 + j] = m[j];
    }
    store_flags_and_counts(flags, tile, flag_words, counts);
}
__global__ __launch_bounds__(kBlock) void join_match_take_kernel(const int32_t *__restrict__ match, const int32_t *__restrict__ right, int64_t n_pairs, int32_t *__restrict__ left) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * kBlock) left[i] = match[right[i]];
}
// build keys that repeat: pairs per probe row counted (the claiming row + its chain), scanned, emitted -- a key's build rows in chain order
template <bool kEmit>
__global__ __launch_bounds__(kBlock) void join_hash_probe_kernel(const void *__restrict__ keys, int32_t type, int64_t n, const JoinSlot *__restrict__ slots, uint64_t cap,
                                                                 const int32_t *__restrict__ next, int32_t *__restrict__ counts, int32_t *__restrict__ out_left,
                                                                 int32_t *__restrict__ out_right, unsigned long long *__restrict__ total64) {
    unsigned long long mine = 0;   // count pass: the pair total in 64 bits (the 32-bit scan of `counts` wraps beyond 2^32 pairs)
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        int32_t head = -1;
        const int32_t first = join_hash_lookup(slots, cap, load_as_i64(keys, type, i), &head);
        int32_t c = 0;
        if (first >= 0) {
            c = 1;
            for (int32_t r = head; r >= 0; r = next[r]) ++c;
            if (kEmit) {
                int64_t at = (int64_t)counts[i] - c;   // (inclusive scan: this row's pairs end at counts[i])
                out_left[at] = first;
                out_right[at++] = (int32_t)i;
                for (int32_t r = head; r >= 0; r = next[r]) {
                    out_left[at] = r;
                    out_right[at++] = (int32_t)i;
                }
            }
        }
        if (!kEmit) {
            counts[i] = c;
            mine += (unsigned long long)c;
        }
    }
    if (!kEmit) {
        mine = wave_sum_u64(mine);
        if (lane_id() == 0 && mine) atomicAdd(total64 + (blockIdx.x & (kJoinTotalSlots - 1)), mine);   // (one of 32 words: a wave per 64 rows on ONE word is ~4.5 ns each, 22 us per 3e5 rows)
    }
}
// ---- join on a DENSE integer key (join_dense): the build side's key offsets from their minimum address the chain heads directly -- the
// multimap of join_hash_build_kernel without a key table, a claim or a probe walk.  Keys are read in their column's own type (no int64 copy).
__global__ __launch_bounds__(kBlock) void join_build_dense_kernel(const void *__restrict__ keys, int32_t type, int64_t n, int64_t kmin, uint32_t range,
                                                                  int32_t *head, int32_t *__restrict__ next, uint32_t *err) {
    bool dup = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t d = (uint64_t)load_as_i64(keys, type, i) - (uint64_t)kmin;
        if (d >= range) {   // (outside the statistics the table was sized from: the call is void)
            atomicOr(err, 1u);
            continue;
        }
        const int32_t before = atomicExch(&head[d], (int32_t)i);  // push front: the chain is walked only after the kernel boundary
        next[i] = before;
        if (before >= 0) dup = true;
    }
    if (dup) atomicOr(err + 1, 1u);   // (err[1]: some key occurs twice on the build side -- the probe then walks chains; unique keys take the flag-tile probe)
}
// ---- probe of a dense table whose build keys are UNIQUE (a primary key: auctions by id, persons by id -- arch/ops/join.sql, q3, q6): a probe row
// matches at most once, so the join is a FILTER of the probe side (flag = the key's slot holds a row) in the flag-tile geometry of the
// selecting kernels -- 16-byte key loads, a lane's 32 flags in one word, wave counts -- followed by the ordinary scan / emit of the flagged
// rows and one gather of the slots' rows for the build side.  Round 5's probe walked a chain per row twice (count pass, emit pass) behind
// 4-byte key loads: 0.08 / 0.10 of the HBM rate with 4x its algorithmic traffic on 2.3e7 x 1.5e6 rows.
template <bool kI32>
__global__ __launch_bounds__(kBlock) void join_probe_unique_flag_kernel(const void *__restrict__ keys, int64_t n, SegTiles st, int64_t kmin, uint32_t range,
                                                                        const int32_t *__restrict__ head, uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts) {
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const int64_t wbase = tr.tile_begin + flag_rel0();
    uint32_t flags = 0;
    if (kI32) {
        int32_t a[kFlagIters][4];
        load_flag_tile(static_cast<const int32_t *>(keys), n, tr, a);
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = wbase + it * 256 + j;
                const uint64_t d = (uint64_t)((int64_t)a[it][j] - kmin);
                const bool in = r < n && d < (uint64_t)range;
                flags |= (uint32_t)(in && head[in ? d : 0] >= 0) << (it * 4 + j);
            }
    } else {
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = wbase + it * 256 + j;
                const uint64_t d = r < n ? (uint64_t)static_cast<const int64_t *>(keys)[r] - (uint64_t)kmin : ~0ull;
                const bool in = d < (uint64_t)range;
                flags |= (uint32_t)(in && head[in ? d : 0] >= 0) << (it * 4 + j);
            }
    }
    store_flags_and_counts(flags, tile, flag_words, counts);
}
// left[i] = the build row in the slot of probe row right[i]'s key (every right[i] matched: the slot holds a row)
__global__ __launch_bounds__(kBlock) void join_unique_take_kernel(const void *__restrict__ keys, int32_t type, const int32_t *__restrict__ right, int64_t n_pairs, int64_t kmin,
                                                                  const int32_t *__restrict__ head, int32_t *__restrict__ left) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * kBlock)
        left[i] = head[(uint64_t)load_as_i64(keys, type, right[i]) - (uint64_t)kmin];
}
template <bool kEmit>
__global__ __launch_bounds__(kBlock) void join_probe_dense_kernel(const void *__restrict__ keys, int32_t type, int64_t n, int64_t kmin, uint32_t range,
                                                                  const int32_t *__restrict__ head, const int32_t *__restrict__ next,
                                                                  int32_t *__restrict__ counts, int32_t *__restrict__ out_left, int32_t *__restrict__ out_right,
                                                                  unsigned long long *__restrict__ total64) {
    unsigned long long mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t d = (uint64_t)load_as_i64(keys, type, i) - (uint64_t)kmin;
        int32_t c = 0;
        int32_t r = d < range ? head[d] : -1;
        if (kEmit) {
            if (r >= 0) {
                int32_t m = 0;
                for (int32_t q = r; q >= 0; q = next[q]) ++m;
                const int64_t at = (int64_t)counts[i] - m;   // (inclusive scan: this row's pairs end at counts[i])
                for (; r >= 0; r = next[r], ++c) {
                    out_left[at + c] = r;
                    out_right[at + c] = (int32_t)i;
                }
            }
        } else {
            for (; r >= 0; r = next[r]) ++c;
            counts[i] = c;
            mine += (unsigned long long)c;
        }
    }
    if (!kEmit) {
        mine = wave_sum_u64(mine);
        if (lane_id() == 0 && mine) atomicAdd(total64 + (blockIdx.x & (kJoinTotalSlots - 1)), mine);   // (one of 32 words: a wave per 64 rows on ONE word is ~4.5 ns each, 22 us per 3e5 rows)
    }
}
// ---- join, both sides small (join_tiny): ONE workgroup builds the smaller side's multimap in LDS -- keys, chain heads and the chain links
// themselves -- and walks the other side through it in chunks of its own size: matches counted, scanned across the workgroup and written
// at their final places in the same pass.  One launch and one host wait for a join of a stage plan's few thousand filtered rows (the
// global-table join is a table fill, a build, a count pass, a three-launch scan and an emit pass: at that size the launches ARE the cost,
// DESIGN section 3a).  Pairs come out ordered by probe row, a key's build rows in chain order -- as join_key64's.
constexpr int kTinyBuild = 4096, kTinySlots = 8192, kTinyThreads = 1024, kTinyProbe = 1 << 16;
__global__ __launch_bounds__(kTinyThreads) void join_tiny_kernel(const int64_t *__restrict__ bkeys, int32_t n_build, const int64_t *__restrict__ pkeys, int32_t n_probe,
                                                                 int32_t *__restrict__ out_build, int32_t *__restrict__ out_probe, uint32_t cap_pairs,
                                                                 unsigned long long *__restrict__ total) {
    __shared__ int64_t s_key[kTinySlots + 1];     // (+ 1: the slot of the key that doubles as the empty mark)
    __shared__ int32_t s_head[kTinySlots + 1];
    __shared__ int32_t s_next[kTinyBuild];
    __shared__ uint32_t s_wave[kTinyThreads / 64];
    for (int i = threadIdx.x; i <= kTinySlots; i += kTinyThreads) {
        s_key[i] = kEmptyKey;
        s_head[i] = -1;
    }
    __syncthreads();
    for (int32_t i = threadIdx.x; i < n_build; i += kTinyThreads) {
        const int64_t key = bkeys[i];
        uint32_t s = kTinySlots;
        if (key != kEmptyKey) {
            s = (uint32_t)mix64((uint64_t)key) & (kTinySlots - 1);
            for (;;) {   // at most half the slots are ever taken: an empty one is always reached
                int64_t cur = __hip_atomic_load(&s_key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == kEmptyKey) {
                    int64_t expected = kEmptyKey;
                    if (__hip_atomic_compare_exchange_strong(&s_key[s], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    cur = expected;
                }
                if (cur == key) break;
                s = (s + 1) & (kTinySlots - 1);
            }
        }
        s_next[i] = atomicExch(&s_head[s], i);
    }
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    unsigned long long running = 0;   // pairs of the chunks before this one (the same in every thread)
    for (int32_t base = 0; base < n_probe; base += kTinyThreads) {
        const int32_t i = base + (int32_t)threadIdx.x;
        int32_t head = -1;
        uint32_t c = 0;
        if (i < n_probe) {
            const int64_t key = pkeys[i];
            if (key == kEmptyKey) {
                head = s_head[kTinySlots];
            } else {
                uint32_t s = (uint32_t)mix64((uint64_t)key) & (kTinySlots - 1);
                for (;;) {
                    const int64_t cur = s_key[s];
                    if (cur == key) {
                        head = s_head[s];
                        break;
                    }
                    if (cur == kEmptyKey) break;
                    s = (s + 1) & (kTinySlots - 1);
                }
            }
            for (int32_t r = head; r >= 0; r = s_next[r]) ++c;
        }
        const uint32_t incl = wave_incl_scan_u32(c);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, chunk = 0;
#pragma unroll
        for (int w = 0; w < kTinyThreads / 64; ++w) {
            before += w < wave ? s_wave[w] : 0u;
            chunk += s_wave[w];
        }
        unsigned long long pos = running + before + (incl - c);
        if (pos + c <= cap_pairs)   // (beyond the buffers the host sized from its estimate: counted, not written -- the host repeats the call)
            for (int32_t r = head; r >= 0; r = s_next[r], ++pos) {
                out_build[pos] = r;
                out_probe[pos] = i;
            }
        running += chunk;
        __syncthreads();   // (s_wave is rewritten by the next chunk)
    }
    if (threadIdx.x == 0) *total = running;
}
__global__ __launch_bounds__(kBlock) void fold_key_kernel(const int64_t *__restrict__ keys, int64_t n, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t k = (uint64_t)keys[i];
        out[i] = (int32_t)(uint32_t)(k ^ (k >> 32));
    }
}

__global__ __launch_bounds__(kBlock) void narrow_kernel(const int64_t *__restrict__ in, int64_t n, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (int32_t)in[i];
}
__global__ __launch_bounds__(kBlock) void add_i32_kernel(int32_t *__restrict__ d, int64_t n, int32_t delta) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) d[i] += delta;
}

uint64_t pow2_at_least(uint64_t v) {
    uint64_t c = 1024;
    while (c < v) c <<= 1;
    return c;
}

#define RELOPS_LAUNCH(ctx, name, kernel, n, ...)                                                                      \
    do {                                                                                                              \
        {                                                                                                             \
            LaunchScope ls_((ctx), name);                                                                             \
            hipLaunchKernelGGL(kernel, dim3(grid_for((ctx), (n))), dim3(kBlock), 0, (ctx)->stream, __VA_ARGS__);      \
        }                                                                                                             \
        FG_TRY(check_launch((ctx), name));                                                                            \
    } while (0)

// ---- ORDER BY: order-preserving 64-bit sub-keys in the current row order, their range, the 32-bit digits a radix sort takes
// chunk: Utf8 only -- -1 = the value's length, c >= 0 = bytes [8c, 8c + 8) big-endian, zero padded
__device__ __forceinline__ uint64_t sort_norm(const void *__restrict__ v, const int32_t *__restrict__ off, int32_t type, int64_t r, int32_t chunk) {
    if (chunk == -2) return static_cast<const uint8_t *>(v)[r] ? 1u : 0u;   // a validity column: NULL = 0, valid = 1
    switch (type) {
        case (int32_t)ColType::I32: return (uint64_t)((int64_t) static_cast<const int32_t *>(v)[r]) ^ (uint64_t(1) << 63);
        case (int32_t)ColType::I64: return (uint64_t) static_cast<const int64_t *>(v)[r] ^ (uint64_t(1) << 63);
        case (int32_t)ColType::U64: return static_cast<const uint64_t *>(v)[r];
        case (int32_t)ColType::F64: {
            const uint64_t b = static_cast<const uint64_t *>(v)[r];
            return (b >> 63) ? ~b : (b | (uint64_t(1) << 63));
        }
        default: {
            const int32_t b = off[r], len = off[r + 1] - b;
            if (chunk < 0) return (uint64_t)(uint32_t)len;
            const uint8_t *p = static_cast<const uint8_t *>(v) + b;
            uint64_t k = 0;
            for (int i = 0; i < 8; ++i) {
                const int32_t at = chunk * 8 + i;
                k = (k << 8) | (at < len ? (uint64_t)p[at] : 0u);
            }
            return k;
        }
    }
}
__global__ __launch_bounds__(kBlock) void sort_norm_kernel(const void *__restrict__ v, const int32_t *__restrict__ off, const uint8_t *__restrict__ valid,
                                                           int32_t type, const int32_t *__restrict__ perm, int64_t n, int32_t chunk, int32_t descending,
                                                           uint64_t *__restrict__ out, uint64_t *__restrict__ minmax) {
    uint64_t mn = ~uint64_t(0), mx = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = perm ? perm[i] : i;
        // (the slot of a NULL holds an unspecified value: all NULLs tie on the value passes, so they keep their input order)
        uint64_t k = valid && chunk != -2 && !valid[r] ? 0u : sort_norm(v, off, type, r, chunk);
        if (descending) k = ~k;
        out[i] = k;
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    // the workgroup's range into ITS slot (minmax[2 b], minmax[2 b + 1]); the host takes the minimum and maximum over the slots.  (One atomicMin /
    // atomicMax pair per wave on two words: 3e4 same-address atomics at the memory side, 0.26 of the kernel's 0.28 ms per 9.8e6 rows -- 1.7 ms of
    // q6's 3.8 ms per execute, six such passes.)
    __shared__ uint64_t s_mm[2 * kWavesPerBlock];
    if (lane_id() == 0) {
        s_mm[threadIdx.x >> 6] = mn;
        s_mm[kWavesPerBlock + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) {
            mn = s_mm[w] < mn ? s_mm[w] : mn;
            mx = s_mm[kWavesPerBlock + w] > mx ? s_mm[kWavesPerBlock + w] : mx;
        }
        minmax[2 * blockIdx.x] = mn;
        minmax[2 * blockIdx.x + 1] = mx;
    }
}
// digits[i] = bits [shift, shift + 32) of (key[i] - base); `perm` re-orders the keys that were computed for an earlier order
__global__ __launch_bounds__(kBlock) void sort_digit_kernel(const uint64_t *__restrict__ key, const uint32_t *__restrict__ order, int64_t n,
                                                            uint64_t base, int32_t shift, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = (int32_t)(uint32_t)((key[order ? order[i] : i] - base) >> shift);
}
// digits[i] = bits [shift, shift + 32) of (norm(key[order ? order[i] : i]) - lo), norm = the integer's order-preserving unsigned form (complemented
// when descending): ORDER BY one integer column without NULLs skips the 64-bit normalised copy of the column (sort_norm_kernel: 12 bytes of
// traffic per row and a pass of its own)
__global__ __launch_bounds__(kBlock) void sort_int_digit_kernel(const void *__restrict__ v, int32_t type, const uint32_t *__restrict__ order, int64_t n,
                                                                int32_t descending, uint64_t lo, int32_t shift, int32_t *__restrict__ out) {
    const uint64_t flip = type == (int32_t)ColType::U64 ? 0ull : (uint64_t(1) << 63);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t k = (uint64_t)load_as_i64(v, type, order ? order[i] : i) ^ flip;
        if (descending) k = ~k;
        out[i] = (int32_t)(uint32_t)((k - lo) >> shift);
    }
}
__global__ __launch_bounds__(kBlock) void iota_i32_kernel(int32_t *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = (int32_t)i;
}
__global__ __launch_bounds__(kBlock) void compose_perm_kernel(const int32_t *__restrict__ perm, const uint32_t *__restrict__ order, int64_t n,
                                                              int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = perm ? perm[order[i]] : (int32_t)order[i];
}
__global__ __launch_bounds__(kBlock) void utf8_max_len_kernel(const int32_t *__restrict__ off, int64_t n, uint64_t *__restrict__ minmax) {
    uint64_t mx = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t l = (uint64_t)(uint32_t)(off[i + 1] - off[i]);
        mx = l > mx ? l : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t b = __shfl_xor(mx, o, 64);
        mx = b > mx ? b : mx;
    }
    // (a slot per workgroup, as sort_norm_kernel: the host folds them)
    __shared__ uint64_t s_mx[kWavesPerBlock];
    if (lane_id() == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) mx = s_mx[w] > mx ? s_mx[w] : mx;
        minmax[2 * blockIdx.x] = 0;
        minmax[2 * blockIdx.x + 1] = mx;
    }
}

// ---- ROW_NUMBER over runs of equal partition keys (row_number_runs)
struct RunKeys {
    const void *v[4];
    const uint8_t *valid[4];
    int32_t type[4];
    int32_t n;
};
__global__ __launch_bounds__(kBlock) void run_start_kernel(RunKeys k, int64_t n, int32_t *__restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        bool start = i == 0;
        for (int c = 0; c < k.n && !start; ++c) {
            const bool va = !k.valid[c] || k.valid[c][i], vb = !k.valid[c] || k.valid[c][i - 1];
            if (va != vb) start = true;
            else if (va) start = k.type[c] == (int32_t)ColType::I32 ? static_cast<const int32_t *>(k.v[c])[i] != static_cast<const int32_t *>(k.v[c])[i - 1]
                                                                    : static_cast<const uint64_t *>(k.v[c])[i] != static_cast<const uint64_t *>(k.v[c])[i - 1];
        }
        flag[i] = start ? 1 : 0;
    }
}
// run[i] = number of the run row i lies in (1-based: the inclusive scan of the start flags)
__global__ __launch_bounds__(kBlock) void run_first_kernel(const int32_t *__restrict__ run, int64_t n, int32_t *__restrict__ first) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (i == 0 || run[i] != run[i - 1]) first[run[i] - 1] = (int32_t)i;
}
__global__ __launch_bounds__(kBlock) void run_rank_kernel(const int32_t *__restrict__ run, const int32_t *__restrict__ first, int64_t n, uint64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (uint64_t)(i - first[run[i] - 1]) + 1;
}

}  // namespace

namespace flockgpu {

int widen_to_i64(flockgpu_ctx *ctx, const DevColumn &col, int64_t rows, int64_t *out) {
    if (col.type == ColType::UTF8 || col.type == ColType::F64) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "key column must be an integer column");
    if (rows <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "widen_kernel", widen_kernel, rows, col.values, (int32_t)col.type, rows, out);
    return FLOCKGPU_OK;
}

int sort_rows(flockgpu_ctx *ctx, const char *name, const SortKey *keys, int n_keys, int64_t rows, int32_t **out_rows, const int32_t **sorted_i32) {
    const std::string base(name);
    if (sorted_i32) *sorted_i32 = nullptr;
    if (rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: relations are limited to 2^31 rows", name);
    int32_t *perm[2] = {nullptr, nullptr};   // the order so far, ping-pong
    FG_TRY(arena_get_t(ctx, (base + ".perm0").c_str(), (size_t)rows + 4, &perm[0]));
    FG_TRY(arena_get_t(ctx, (base + ".perm1").c_str(), (size_t)rows + 4, &perm[1]));
    *out_rows = perm[0];
    if (rows == 0) return FLOCKGPU_OK;
    uint64_t *nk = nullptr, *d_mm = nullptr, *h_mm = nullptr;
    int32_t *digits = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".key").c_str(), (size_t)rows + 2, &nk));
    FG_TRY(arena_get_t(ctx, (base + ".digit").c_str(), (size_t)rows + 4, &digits));
    const unsigned grid = grid_for(ctx, rows);
    FG_TRY(arena_get_t(ctx, (base + ".minmax").c_str(), 2 * (size_t)grid + 2, &d_mm));   // (a slot per workgroup of sort_norm_kernel)
    FG_TRY(pinned_get_t(ctx, (base + ".minmax").c_str(), 2 * (size_t)grid + 2, &h_mm));
    // ORDER BY ONE integer column without NULLs -- `SELECT * FROM bid ORDER BY bidder`, the reference's arch/ops/sort.sql --: the column's exact
    // range (one 4-byte-per-row pass) sizes the radix sort, whose first pass reads the COLUMN itself when it is an ascending Int32 (bias =
    // minimum); other integer keys go through one digit pass per 32 bits.  No normalised 64-bit copy, no composition of permutations: the
    // sort's row numbers ARE the result.  (The general sequence below: 0.53 + 0.21 + 0.16 ms of such passes per 9.2e7 bids.)
    if (n_keys == 1 && !keys[0].col.valid && !keys[0].col.all_null &&
        (keys[0].col.type == ColType::I32 || keys[0].col.type == ColType::I64 || keys[0].col.type == ColType::U64)) {
        const SortKey &k = keys[0];
        int64_t mn = 0, mx = 0;
        FG_TRY(column_minmax(ctx, k.col, rows, &mn, &mx));
        const uint64_t flip = k.col.type == ColType::U64 ? 0ull : (uint64_t(1) << 63);
        uint64_t lo = (uint64_t)mn ^ flip, hi = (uint64_t)mx ^ flip;
        if (k.descending) {
            const uint64_t t = ~hi;
            hi = ~lo;
            lo = t;
        }
        const uint64_t span = hi - lo;
        int bits = 0;
        while (bits < 64 && (span >> bits)) ++bits;
        const uint32_t *order = nullptr;
        int32_t *sk = nullptr;
        uint32_t *sv = nullptr;
        if (bits == 0) {   // every row ties: the identity
            hipLaunchKernelGGL(iota_i32_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, perm[0], rows);
            FG_TRY(check_launch(ctx, "iota_i32_kernel"));
            return FLOCKGPU_OK;
        }
        if (k.col.type == ColType::I32 && !k.descending && bits <= 32) {
            FG_TRY(radix_sort_pairs(ctx, (base + ".lo").c_str(), static_cast<const int32_t *>(k.col.values), nullptr, rows, (int32_t)mn, bits, &sk, &sv));
            order = sv;
            if (sorted_i32) *sorted_i32 = sk;   // the column in sorted order: what the passes carried
        } else {
            for (int shift = 0; shift < bits; shift += 32) {
                {
                    LaunchScope ls(ctx, "sort_int_digit_kernel");
                    hipLaunchKernelGGL(sort_int_digit_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, k.col.values, (int32_t)k.col.type, order, rows,
                                       k.descending ? 1 : 0, lo, shift, digits);
                }
                FG_TRY(check_launch(ctx, "sort_int_digit_kernel"));
                FG_TRY(radix_sort_pairs(ctx, (base + (shift ? ".hi" : ".lo")).c_str(), digits, order, rows, 0, std::min(32, bits - shift), &sk, &sv));
                order = sv;
            }
        }
        *out_rows = reinterpret_cast<int32_t *>(const_cast<uint32_t *>(order));   // (row numbers below 2^31: the same bits)
        return FLOCKGPU_OK;
    }
    const int32_t *cur = nullptr;   // null: the identity
    int at = 0;                     // perm[at] receives the next order
    auto read_range = [&](unsigned slots = 1) -> int {   // h_mm[0 .. 1] = the range over `slots` (min, max) pairs
        FG_HIP(ctx, hipMemcpyAsync(h_mm, d_mm, 2 * (size_t)slots * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (unsigned b = 1; b < slots; ++b) {
            h_mm[0] = std::min(h_mm[0], h_mm[2 * b]);
            h_mm[1] = std::max(h_mm[1], h_mm[2 * b + 1]);
        }
        return FLOCKGPU_OK;
    };
    // one stable pass over a 64-bit sub-key of column `k` (chunk: Utf8 only)
    auto pass = [&](const SortKey &k, int32_t chunk) -> int {
        {
            LaunchScope ls(ctx, "sort_norm_kernel");
            hipLaunchKernelGGL(sort_norm_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, k.col.values, k.col.offsets, k.col.valid, (int32_t)k.col.type, cur, rows,
                               chunk, k.descending ? 1 : 0, nk, d_mm);
        }
        FG_TRY(check_launch(ctx, "sort_norm_kernel"));
        FG_TRY(read_range(grid));
        if (h_mm[0] > h_mm[1]) return FLOCKGPU_OK;   // (no rows)
        const uint64_t lo = h_mm[0], span = h_mm[1] - h_mm[0];
        if (span == 0) return FLOCKGPU_OK;   // every row ties on this sub-key
        int bits = 1;
        while (bits < 64 && (span >> bits)) ++bits;
        const uint32_t *order = nullptr;     // the order of this sub-key's rows after its lower 32 bits were sorted
        for (int shift = 0; shift < bits; shift += 32) {
            const int nb = std::min(32, bits - shift);
            hipLaunchKernelGGL(sort_digit_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, nk, order, rows, lo, shift, digits);
            FG_TRY(check_launch(ctx, "sort_digit_kernel"));
            int32_t *sk = nullptr;
            uint32_t *sv = nullptr;
            FG_TRY(radix_sort_pairs(ctx, (base + (shift ? ".hi" : ".lo")).c_str(), digits, order, rows, 0, nb, &sk, &sv));
            order = sv;
        }
        hipLaunchKernelGGL(compose_perm_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, cur, order, rows, perm[at]);
        FG_TRY(check_launch(ctx, "compose_perm_kernel"));
        cur = perm[at];
        at ^= 1;
        return FLOCKGPU_OK;
    };
    // where a key's NULLs go: one more stable pass on its validity byte, AFTER the value passes (more significant than the value)
    auto null_pass = [&](const SortKey &k) -> int {
        if (!k.col.valid) return FLOCKGPU_OK;
        SortKey v;
        v.col.type = ColType::I32;   // unused by the validity chunk
        v.col.values = k.col.valid;
        v.descending = !k.nulls_first;   // the pass orders NULL = 0 before valid = 1: ascending puts the NULLs first, "nulls last" inverts it
        return pass(v, -2);
    };
    for (int ki = n_keys - 1; ki >= 0; --ki) {
        const SortKey &k = keys[ki];
        if (k.col.all_null) continue;
        if (k.col.type != ColType::UTF8) {
            FG_TRY(pass(k, 0));
            FG_TRY(null_pass(k));
            continue;
        }
        hipLaunchKernelGGL(utf8_max_len_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, k.col.offsets, rows, d_mm);
        FG_TRY(check_launch(ctx, "utf8_max_len_kernel"));
        FG_TRY(read_range(grid));
        const int64_t max_len = (int64_t)h_mm[1];
        FG_TRY(pass(k, -1));                                            // length: decides between a string and its zero-padded twin
        for (int32_t c = (int32_t)((max_len + 7) / 8) - 1; c >= 0; --c) FG_TRY(pass(k, c));
        FG_TRY(null_pass(k));
    }
    if (!cur) {   // no key moved anything: the identity
        hipLaunchKernelGGL(iota_i32_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, perm[0], rows);
        FG_TRY(check_launch(ctx, "iota_i32_kernel"));
        cur = perm[0];
    }
    *out_rows = const_cast<int32_t *>(cur);
    return FLOCKGPU_OK;
}

int widen_u32_to_u64(flockgpu_ctx *ctx, const uint32_t *in, int64_t n, uint64_t *out) {
    if (n <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "widen_u32_kernel", widen_u32_kernel, n, in, n, out);
    return FLOCKGPU_OK;
}

int narrow_i64_to_i32(flockgpu_ctx *ctx, const int64_t *in, int64_t n, int32_t *out) {
    if (n <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "narrow_kernel", narrow_kernel, n, in, n, out);
    return FLOCKGPU_OK;
}

int add_i32(flockgpu_ctx *ctx, int32_t *data, int64_t n, int32_t delta) {
    if (n <= 0 || delta == 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "add_i32_kernel", add_i32_kernel, n, data, n, delta);
    return FLOCKGPU_OK;
}

int mask_to_rows(flockgpu_ctx *ctx, const char *name, const uint8_t *mask, int64_t rows, int32_t **out_rows, int64_t *n_out) {
    const std::string base = name;
    int32_t *o_rows = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".rows").c_str(), (size_t)std::max<int64_t>(rows, 0) + 4, &o_rows));
    *out_rows = o_rows;
    *n_out = 0;
    if (rows <= 0) return FLOCKGPU_OK;
    if (rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 2^31 rows", name);
    int64_t sb = 0, se = rows;
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, (base + ".tiles").c_str(), &sb, &se, 1, kFlagTile, &st));
    uint32_t *flags = nullptr, *counts = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".flags").c_str(), (size_t)st.n_tiles * kBlock + 4, &flags));
    FG_TRY(arena_get_t(ctx, (base + ".counts").c_str(), (size_t)st.n_tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, (base + ".base").c_str(), (size_t)st.n_tiles + 1, &tile_base));
    FG_TRY(pinned_get_t(ctx, (base + ".off").c_str(), 2, &h_off));
    pinned_pending(reinterpret_cast<uint64_t *>(h_off), 2);
    {
        LaunchScope ls(ctx, "mask_flag_kernel");
        hipLaunchKernelGGL(mask_flag_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, mask, rows, st, flags, counts);
    }
    FG_TRY(check_launch(ctx, "mask_flag_kernel"));
    if (st.n_tiles <= 2048) {   // (no scan launch for relations of up to 1.7e7 rows: gather.hpp, emit_flagged_rows_self)
        FG_TRY(emit_flagged_rows_self(ctx, st, flags, counts, o_rows, h_off));
    } else {
        FG_TRY(launch_tile_scan(ctx, counts, st.n_tiles, tile_base, st.tile_first, st.n_seg, h_off));   // (the scan writes its segment offsets straight into pinned memory)
        FG_TRY(emit_flagged_rows(ctx, st, flags, counts, tile_base, o_rows));
    }
    FG_TRY(wait_pinned(ctx, reinterpret_cast<const uint64_t *>(h_off), 2));
    *n_out = h_off[1];
    return FLOCKGPU_OK;
}

int gather_u8(flockgpu_ctx *ctx, const uint8_t *src, const int32_t *rows, int64_t n, uint8_t *out) {
    if (n <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "gather_u8_kernel", gather_u8_kernel, n, src, rows, n, out);
    return FLOCKGPU_OK;
}
int replace_invalid_i64(flockgpu_ctx *ctx, int64_t *keys, const uint8_t *valid, int64_t rows, int64_t sentinel) {
    if (rows <= 0 || !valid) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "replace_invalid_kernel", replace_invalid_kernel, rows, keys, valid, rows, sentinel);
    return FLOCKGPU_OK;
}
int valid_from_i64(flockgpu_ctx *ctx, const int64_t *keys, int64_t n, int64_t sentinel, uint8_t *out) {
    if (n <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "valid_from_i64_kernel", valid_from_i64_kernel, n, keys, n, sentinel, out);
    return FLOCKGPU_OK;
}

int take_column(flockgpu_ctx *ctx, const char *name, const DevColumn &src, const int32_t *rows, int64_t n, DevColumn *out) {
    *out = src;
    out->values = nullptr;
    out->offsets = nullptr;
    out->bytes = 0;
    out->valid = nullptr;
    if (src.valid) {
        uint8_t *v = nullptr;
        FG_TRY(arena_get_t(ctx, (std::string(name) + ".valid").c_str(), (size_t)std::max<int64_t>(n, 0) + 16, &v));
        FG_TRY(gather_u8(ctx, src.valid, rows, n, v));
        out->valid = v;
    }
    if (src.type == ColType::UTF8) {
        flockgpu_utf8 u{}, s{src.offsets, static_cast<const uint8_t *>(src.values)};
        int64_t nbytes = 0;
        FG_TRY(gather_utf8(ctx, name, s, rows, n, &u, &nbytes));
        out->values = u.data;
        out->offsets = u.offsets;
        out->bytes = nbytes;
        return FLOCKGPU_OK;
    }
    void *p = nullptr;
    FG_TRY(arena_get(ctx, (std::string(name) + ".val").c_str(), (size_t)std::max<int64_t>(n, 0) * col_width(src.type) + 16, &p));
    out->values = p;
    if (src.type == ColType::I32) return gather_i32(ctx, static_cast<const int32_t *>(src.values), rows, n, static_cast<int32_t *>(p));
    return gather_i64(ctx, static_cast<const int64_t *>(src.values), rows, n, static_cast<int64_t *>(p));
}

int group_by_key64(flockgpu_ctx *ctx, const char *name, const int64_t *keys, const uint64_t *values, AggKind kind, int64_t rows,
                   GroupResult *out) {
    *out = GroupResult{};
    const std::string base = name;
    if (rows >= (int64_t(1) << 30)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 2^30 rows in a generic GROUP BY", name);
    const uint64_t cap = pow2_at_least((uint64_t)std::max<int64_t>(rows, 1) * 2);
    const int64_t slots = (int64_t)cap + 1;
    int64_t *tk = nullptr;
    uint64_t *ta = nullptr;
    int32_t *tf = nullptr;
    uint8_t *live = nullptr;
    uint32_t *d_err = nullptr, *h_err = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".tk").c_str(), (size_t)slots, &tk));
    FG_TRY(arena_get_t(ctx, (base + ".ta").c_str(), (size_t)slots, &ta));
    FG_TRY(arena_get_t(ctx, (base + ".tf").c_str(), (size_t)slots, &tf));
    FG_TRY(arena_get_t(ctx, (base + ".live").c_str(), (size_t)slots + 16, &live));
    FG_TRY(arena_get_t(ctx, (base + ".err").c_str(), 4, &d_err));
    FG_TRY(pinned_get_t(ctx, (base + ".err").c_str(), 4, &h_err));
    RELOPS_LAUNCH(ctx, "group_init_kernel", group_init_kernel, slots, tk, ta, tf, slots, d_err);
    if (rows > 0)
        RELOPS_LAUNCH(ctx, "group_insert_kernel", group_insert_kernel, rows, keys, values, (int32_t)kind, rows, tk, ta, tf, cap, d_err);
    RELOPS_LAUNCH(ctx, "live_slot_mask_kernel", live_slot_mask_kernel, slots, tf, slots, live, d_err, h_err);
    int32_t *slot_rows = nullptr;
    int64_t n_groups = 0;
    FG_TRY(mask_to_rows(ctx, (base + ".sel").c_str(), live, slots, &slot_rows, &n_groups));  // synchronises
    if (*h_err) return fail(ctx, FLOCKGPU_ERR_CAPACITY, "%s: group table overflow", name);
    int64_t *ok = nullptr;
    uint64_t *oa = nullptr;
    int32_t *of = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".ok").c_str(), (size_t)n_groups + 2, &ok));
    FG_TRY(arena_get_t(ctx, (base + ".oa").c_str(), (size_t)n_groups + 2, &oa));
    FG_TRY(arena_get_t(ctx, (base + ".of").c_str(), (size_t)n_groups + 4, &of));
    FG_TRY(gather_i64(ctx, tk, slot_rows, n_groups, ok));
    FG_TRY(gather_i64(ctx, reinterpret_cast<const int64_t *>(ta), slot_rows, n_groups, reinterpret_cast<int64_t *>(oa)));
    FG_TRY(gather_i32(ctx, tf, slot_rows, n_groups, of));
    out->n_groups = n_groups;
    out->keys = ok;
    out->agg = oa;
    out->first_row = of;
    return FLOCKGPU_OK;
}

int group_by_key64_n(flockgpu_ctx *ctx, const char *name, const int64_t *keys, int64_t rows, const AggSpec *specs, int n_specs,
                     GroupResultN *out, const uint8_t *key_valid) {
    *out = GroupResultN{};
    const std::string base = name;
    if (n_specs < 0 || n_specs > kMaxGroupAggs) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than %d accumulators per group", name, kMaxGroupAggs);
    if (rows >= (int64_t(1) << 30)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 2^30 rows in a generic GROUP BY", name);
    AggSpecs sp{};
    sp.n = n_specs;
    bool track = false;
    for (int a = 0; a < n_specs; ++a) {
        sp.values[a] = specs[a].values;
        sp.valid[a] = specs[a].valid;
        track = track || specs[a].valid != nullptr;
        sp.op[a] = (int32_t)specs[a].op;
        sp.type[a] = (int32_t)specs[a].type;
        if (specs[a].op != AggOp::COUNT && !specs[a].values && rows > 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: accumulator without a value column", name);
        const bool is_f64 = specs[a].type == ColType::F64, is_text = specs[a].type == ColType::UTF8;
        const bool f64_op = specs[a].op == AggOp::SUM_F64 || specs[a].op == AggOp::MAX_F64 || specs[a].op == AggOp::MIN_F64;
        const bool bad = specs[a].op == AggOp::COUNT ? false : (f64_op ? !is_f64 : (is_f64 || is_text));
        if (bad) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: accumulator over a column of the wrong type", name);
    }
    const int width = std::max(n_specs, 1);
    // The table: two slots per row cannot overflow, but it is 2^25 slots of 20+ bytes for 9.2e6 rows that form 6e5 groups -- 0.2 ms to
    // initialise and every probe a miss.  A name that has been here before is sized for three slots per group it had then (a streaming host
    // sends window after window of the same shape); an overflow -- probing is cut off, claim_slot -- repeats the pass with four times the
    // slots, up to the two per row that always hold.
    const uint64_t full = pow2_at_least((uint64_t)std::max<int64_t>(rows, 1) * 2);
    std::vector<int64_t> &hint = ctx->host_i64[base + ".groups_hint"];   // {groups of the last call under this name + 1}
    uint64_t cap = hint.empty() || hint[0] <= 0 ? full : std::min(full, pow2_at_least((uint64_t)std::max<int64_t>((hint[0] - 1) * 3, 1024)));
    int64_t *tk = nullptr;
    uint64_t *ta = nullptr;
    int32_t *tf = nullptr;
    uint8_t *live = nullptr;
    uint32_t *d_err = nullptr, *h_err = nullptr;
    int32_t *slot_rows = nullptr;
    int64_t n_groups = 0;
    for (;;) {
        const int64_t slots = (int64_t)cap + 2;   // + the key INT64_MIN's slot + the NULL keys' slot
        FG_TRY(arena_get_t(ctx, (base + ".tk").c_str(), (size_t)slots, &tk));
        FG_TRY(arena_get_t(ctx, (base + ".tan").c_str(), (size_t)slots * (size_t)width, &ta));
        FG_TRY(arena_get_t(ctx, (base + ".tf").c_str(), (size_t)slots, &tf));
        FG_TRY(arena_get_t(ctx, (base + ".live").c_str(), (size_t)slots + 16, &live));
        FG_TRY(arena_get_t(ctx, (base + ".err").c_str(), 4, &d_err));
        FG_TRY(pinned_get_t(ctx, (base + ".err").c_str(), 4, &h_err));
        sp.seen = nullptr;
        if (track) {
            FG_TRY(arena_get_t(ctx, (base + ".seen").c_str(), (size_t)slots * (size_t)width, &sp.seen));
            RELOPS_LAUNCH(ctx, "zero_u32_kernel", zero_u32_kernel, slots * width, sp.seen, slots * (int64_t)width);
        }
        RELOPS_LAUNCH(ctx, "group_init_n_kernel", group_init_n_kernel, slots, tk, ta, tf, slots, sp, d_err);
        if (rows > 0) {
            // the workgroup-level table (LDS): 2048 slots for one or two accumulators, 1024 beyond; four rows per slot
            const GroupTable gt{tk, ta, tf, cap, d_err, key_valid};
            const uint32_t lds_slots = width <= 2 ? 2048u : 1024u;
            const size_t lds_bytes = (size_t)lds_slots * (8 + 8 * (size_t)sp.n + 4 + 4 * (size_t)sp.n);
            const int64_t rows_per_wg = (int64_t)lds_slots * 4;   // (NEXMark bids by auction: ~110 + 0.065 x rows distinct keys per run of rows; 16 / 8 / 4 rows per slot: 1.38 / 0.67 / 0.54 ms per 9.2e6 bids)
            LaunchScope ls_(ctx, "group_insert_n_kernel");
            if (rows >= rows_per_wg * 4 && sp.n > 0) {
                hipLaunchKernelGGL(group_insert_n_kernel<true>, dim3((unsigned)div_up(rows, rows_per_wg)), dim3(kBlock), lds_bytes, ctx->stream, keys, rows, sp, gt,
                                   rows_per_wg, lds_slots);
            } else {   // a few thousand rows: nothing to stage
                hipLaunchKernelGGL(group_insert_n_kernel<false>, dim3(grid_for(ctx, rows)), dim3(kBlock), 0, ctx->stream, keys, rows, sp, gt, (int64_t)0, 0u);
            }
        }
        FG_TRY(check_launch(ctx, "group_insert_n_kernel"));
        RELOPS_LAUNCH(ctx, "live_slot_mask_kernel", live_slot_mask_kernel, slots, tf, slots, live, d_err, h_err);
        FG_TRY(mask_to_rows(ctx, (base + ".sel").c_str(), live, slots, &slot_rows, &n_groups));  // synchronises
        if (!*h_err) break;
        if (cap >= full) return fail(ctx, FLOCKGPU_ERR_CAPACITY, "%s: group table overflow", name);
        cap = full;   // the hint was wrong: two slots per row always hold -- ONE more pass, not a ladder of x4 retries over every row
    }
    hint.assign(1, n_groups + 1);
    int64_t *ok = nullptr;
    int32_t *of = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".ok").c_str(), (size_t)n_groups + 2, &ok));
    FG_TRY(arena_get_t(ctx, (base + ".of").c_str(), (size_t)n_groups + 4, &of));
    FG_TRY(gather_i64(ctx, tk, slot_rows, n_groups, ok));
    FG_TRY(gather_i32(ctx, tf, slot_rows, n_groups, of));
    AggOuts o{};
    for (int a = 0; a < n_specs; ++a) {
        FG_TRY(arena_get_t(ctx, (base + ".oa" + std::to_string(a)).c_str(), (size_t)n_groups + 2, &o.out[a]));
        out->agg[a] = o.out[a];
        if (specs[a].valid) {
            FG_TRY(arena_get_t(ctx, (base + ".ov" + std::to_string(a)).c_str(), (size_t)n_groups + 16, &o.valid[a]));
            out->agg_valid[a] = o.valid[a];
        }
    }
    if (n_specs > 0 && n_groups > 0) RELOPS_LAUNCH(ctx, "group_collect_n_kernel", group_collect_n_kernel, n_groups, ta, slot_rows, n_groups, sp, o);
    if (key_valid) {
        if (cap + 1 >= (uint64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: group table too large for NULL keys", name);
        uint8_t *kv = nullptr;
        FG_TRY(arena_get_t(ctx, (base + ".okv").c_str(), (size_t)n_groups + 16, &kv));
        if (n_groups > 0) RELOPS_LAUNCH(ctx, "key_valid_from_slot_kernel", key_valid_from_slot_kernel, n_groups, slot_rows, n_groups, (int32_t)(cap + 1), kv);
        out->key_valid = kv;
    }
    out->n_groups = n_groups;
    out->keys = ok;
    out->first_row = of;
    return FLOCKGPU_OK;
}

int column_minmax(flockgpu_ctx *ctx, const DevColumn &col, int64_t rows, int64_t *mn, int64_t *mx) {
    *mn = *mx = 0;
    if (col.type == ColType::UTF8 || col.type == ColType::F64) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "column statistics need an integer column");
    if (rows <= 0) return FLOCKGPU_OK;
    const unsigned blocks = std::min<unsigned>(grid_for(ctx, rows / 4 + 1), 1024);
    int64_t *h = nullptr;
    FG_TRY(pinned_get_t(ctx, "relops.minmax", 2048, &h));
    {
        LaunchScope ls(ctx, "minmax_kernel");   // (the per-block results go straight into pinned memory: no copy call behind the kernel)
        hipLaunchKernelGGL(minmax_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, col.values, (int32_t)col.type, rows, h);
    }
    FG_TRY(check_launch(ctx, "minmax_kernel"));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const bool uns = col.type == ColType::U64;
    int64_t lo = h[0], hi = h[blocks];
    for (unsigned b = 1; b < blocks; ++b) {
        if (uns ? (uint64_t)h[b] < (uint64_t)lo : h[b] < lo) lo = h[b];
        if (uns ? (uint64_t)h[blocks + b] > (uint64_t)hi : h[blocks + b] > hi) hi = h[blocks + b];
    }
    *mn = lo;
    *mx = hi;
    return FLOCKGPU_OK;
}

bool dense_range_ok(int64_t kmin, int64_t kmax, int64_t rows, bool uns) {
    const uint64_t span = (uint64_t)kmax - (uint64_t)kmin;   // (mod 2^64: the true span for either signedness when kmin <= kmax in its order)
    if (uns ? (uint64_t)kmax < (uint64_t)kmin : kmax < kmin) return false;
    // a slot per key value: affordable while the range is within 16x the rows -- clearing and compacting the table is 4-12 bytes of STREAMING
    // per slot, a hash table's claim a random returning atomic per row; one of P = 8 hash partitions of a dense id column is 8x as wide as
    // its rows (the stage plans' joins and Final aggregates)
    return span < (uint64_t)std::max<int64_t>(int64_t(1) << 16, 16 * rows) && span < (uint64_t(1) << 31) - 1;
}

int group_by_dense(flockgpu_ctx *ctx, const char *name, const DevColumn &key, int64_t rows, int64_t kmin, int64_t kmax, const AggSpec *specs, int n_specs,
                   GroupResultN *out) {
    *out = GroupResultN{};
    const std::string base = name;
    if (n_specs < 0 || n_specs > kMaxGroupAggs) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than %d accumulators per group", name, kMaxGroupAggs);
    if (key.valid || (key.type != ColType::I32 && key.type != ColType::I64 && key.type != ColType::U64) || rows <= 0 || rows >= (int64_t(1) << 31) ||
        !dense_range_ok(kmin, kmax, rows, key.type == ColType::U64))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: not a dense integer key", name);
    const uint32_t range = (uint32_t)((uint64_t)kmax - (uint64_t)kmin + 1);
    AggSpecs sp{};
    DenseOuts o{};
    o.n = n_specs;
    for (int a = 0; a < n_specs; ++a) {
        if (specs[a].valid) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: the dense path takes arguments without NULLs", name);
        if (specs[a].op == AggOp::COUNT) {
            o.acc_of[a] = -1;
            continue;
        }
        if (specs[a].op == AggOp::SUM_F64 || specs[a].op == AggOp::MAX_F64 || specs[a].op == AggOp::MIN_F64 || !specs[a].values ||
            specs[a].type == ColType::F64 || specs[a].type == ColType::UTF8)
            return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: the dense path takes integer accumulators", name);
        o.acc_of[a] = sp.n;
        sp.values[sp.n] = specs[a].values;
        sp.op[sp.n] = (int32_t)specs[a].op;
        sp.type[sp.n] = (int32_t)specs[a].type;
        ++sp.n;
    }
    uint32_t *cnt = nullptr, *d_err = nullptr, *h_err = nullptr;
    uint64_t *acc = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".dcnt").c_str(), (size_t)range + 8, &cnt));
    FG_TRY(arena_get_t(ctx, (base + ".dacc").c_str(), (size_t)range * (size_t)std::max(sp.n, 1) + 2, &acc));
    FG_TRY(arena_get_t(ctx, (base + ".err").c_str(), 4, &d_err));
    FG_TRY(pinned_get_t(ctx, (base + ".err").c_str(), 4, &h_err));
    {   // the error word, the counters and every accumulator whose identity is zero (or all ones): one launch; the signed extremes keep theirs
        FillList fl;
        fl.add(d_err, 0u, 1).add(cnt, 0u, range);
        for (int a = 0; a < sp.n; ++a) {
            const uint64_t id = sp.op[a] == (int32_t)AggOp::MAX_S ? (uint64_t)INT64_MIN : sp.op[a] == (int32_t)AggOp::MIN_S ? (uint64_t)INT64_MAX : sp.op[a] == (int32_t)AggOp::MIN_U ? ~0ull : 0ull;
            if ((id == 0 || id == ~0ull) && fl.n < 4) fl.add(acc + (size_t)a * range, (uint32_t)id, (uint64_t)range * 2);
            else RELOPS_LAUNCH(ctx, "fill_u64_kernel", fill_u64_kernel, (int64_t)range, acc + (size_t)a * range, (int64_t)range, id);
        }
        FG_TRY(fill_words(ctx, fl));
    }
    {
        const size_t lds = (size_t)kDenseBins * (4 + 8 * (size_t)sp.n);
        const unsigned tiles = (unsigned)div_up(rows, kDenseTile);
        const bool k64 = key.type != ColType::I32;
        // four non-COUNT accumulators need 72 KB of the CU's 160 KB: above the 64 KB a launch gets without asking
        if (lds > 64 * 1024)
            FG_HIP(ctx, hipFuncSetAttribute(k64 ? reinterpret_cast<const void *>(&dense_group_kernel<true, true>) : reinterpret_cast<const void *>(&dense_group_kernel<false, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LaunchScope ls(ctx, "dense_group_kernel");
        if (sp.n == 0) {
            if (k64) hipLaunchKernelGGL((dense_group_kernel<true, false>), dim3(tiles), dim3(kBlock), lds, ctx->stream, key.values, rows, kmin, range, sp, cnt, acc, d_err);
            else hipLaunchKernelGGL((dense_group_kernel<false, false>), dim3(tiles), dim3(kBlock), lds, ctx->stream, key.values, rows, kmin, range, sp, cnt, acc, d_err);
        } else {
            if (k64) hipLaunchKernelGGL((dense_group_kernel<true, true>), dim3(tiles), dim3(kBlock), lds, ctx->stream, key.values, rows, kmin, range, sp, cnt, acc, d_err);
            else hipLaunchKernelGGL((dense_group_kernel<false, true>), dim3(tiles), dim3(kBlock), lds, ctx->stream, key.values, rows, kmin, range, sp, cnt, acc, d_err);
        }
    }
    FG_TRY(check_launch(ctx, "dense_group_kernel"));
    // the live slots, in key order
    int64_t sb = 0, se = (int64_t)range;
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, (base + ".dtiles").c_str(), &sb, &se, 1, kFlagTile, &st));
    uint32_t *flags = nullptr, *counts = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *h_off = nullptr;
    int32_t *slots = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".dflags").c_str(), (size_t)st.n_tiles * kBlock + 4, &flags));
    FG_TRY(arena_get_t(ctx, (base + ".dcounts").c_str(), (size_t)st.n_tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, (base + ".dbase").c_str(), (size_t)st.n_tiles + 1, &tile_base));
    FG_TRY(pinned_get_t(ctx, (base + ".doff").c_str(), 2, &h_off));
    pinned_pending(reinterpret_cast<uint64_t *>(h_off), 2);
    FG_TRY(arena_get_t(ctx, (base + ".dslots").c_str(), (size_t)std::min<int64_t>(range, rows) + 4, &slots));   // (at most one live slot per row)
    {
        LaunchScope ls(ctx, "dense_live_flag_kernel");
        hipLaunchKernelGGL(dense_live_flag_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, cnt, (int64_t)range, st, flags, counts, d_err, h_err);
    }
    FG_TRY(check_launch(ctx, "dense_live_flag_kernel"));
    if (st.n_tiles <= 2048) {
        FG_TRY(emit_flagged_rows_self(ctx, st, flags, counts, slots, h_off));
    } else {
        FG_TRY(launch_tile_scan(ctx, counts, st.n_tiles, tile_base, st.tile_first, st.n_seg, h_off));   // (segment offsets straight into pinned memory)
        FG_TRY(emit_flagged_rows(ctx, st, flags, counts, tile_base, slots));
    }
    FG_TRY(wait_pinned(ctx, reinterpret_cast<const uint64_t *>(h_off), 2));   // (h_err: written by dense_live_flag_kernel, a kernel earlier on the stream)
    if (*h_err) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: a key outside the column statistics [%lld, %lld] the table was sized from", name, (long long)kmin, (long long)kmax);
    const int64_t n_groups = h_off[1];
    int64_t *ok = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".ok").c_str(), (size_t)n_groups + 2, &ok));
    for (int a = 0; a < n_specs; ++a) {
        FG_TRY(arena_get_t(ctx, (base + ".oa" + std::to_string(a)).c_str(), (size_t)n_groups + 2, &o.out[a]));
        out->agg[a] = o.out[a];
    }
    if (n_groups > 0) RELOPS_LAUNCH(ctx, "dense_collect_kernel", dense_collect_kernel, n_groups, slots, n_groups, kmin, range, cnt, acc, ok, o);
    out->n_groups = n_groups;
    out->keys = ok;
    return FLOCKGPU_OK;
}

int pack_i32_pair(flockgpu_ctx *ctx, const int32_t *a, const int32_t *b, int64_t n, int64_t *out) {
    if (n <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "pack_pair_kernel", pack_pair_kernel, n, a, b, n, out);
    return FLOCKGPU_OK;
}
int unpack_i32_pair(flockgpu_ctx *ctx, const int64_t *keys, int64_t n, int32_t *a, int32_t *b) {
    if (n <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "unpack_pair_kernel", unpack_pair_kernel, n, keys, n, a, b);
    return FLOCKGPU_OK;
}
int i64_to_f64(flockgpu_ctx *ctx, const int64_t *in, int64_t n, double *out) {
    if (n <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "i64_to_f64_kernel", i64_to_f64_kernel, n, in, n, out);
    return FLOCKGPU_OK;
}
int avg_finish(flockgpu_ctx *ctx, const double *sum, const uint64_t *count, int64_t n, double *out) {
    if (n <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "avg_finish_kernel", avg_finish_kernel, n, sum, count, n, out);
    return FLOCKGPU_OK;
}

int distinct_i32_utf8(flockgpu_ctx *ctx, const char *name, const int32_t *key, const flockgpu_utf8 &text, int64_t rows, int32_t **out_rows,
                      int64_t *n_out) {
    const std::string base = name;
    if (rows >= (int64_t(1) << 30)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 2^30 rows in a generic DISTINCT", name);
    const uint64_t cap = pow2_at_least((uint64_t)std::max<int64_t>(rows, 1) * 2);
    int32_t *table = nullptr;
    uint8_t *rep = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".table").c_str(), (size_t)cap + 1, &table));   // (+ 1: the "keys are strictly increasing" mark)
    FG_TRY(arena_get_t(ctx, (base + ".rep").c_str(), (size_t)std::max<int64_t>(rows, 0) + 16, &rep));
    RELOPS_LAUNCH(ctx, "fill_i32_kernel", fill_i32_kernel, (int64_t)cap + 1, table, (int64_t)cap + 1, (int32_t)-1);
    if (rows > 0) {
        RELOPS_LAUNCH(ctx, "distinct_order_check_kernel", distinct_order_check_kernel, rows, key, rows, table + cap);
        RELOPS_LAUNCH(ctx, "distinct_insert_kernel", distinct_insert_kernel, rows, key, text.offsets, text.data, rows, table, cap, rep);
    }
    return mask_to_rows(ctx, (base + ".sel").c_str(), rep, rows, out_rows, n_out);
}

int hash_utf8_i64(flockgpu_ctx *ctx, const DevColumn &col, int64_t rows, int64_t *out) {
    if (col.type != ColType::UTF8) return fail(ctx, FLOCKGPU_ERR_INVALID, "hash_utf8: not a Utf8 column");
    if (rows <= 0) return FLOCKGPU_OK;
    RELOPS_LAUNCH(ctx, "hash_utf8_kernel", hash_utf8_kernel, rows, col.offsets, static_cast<const uint8_t *>(col.values), rows, out);
    return FLOCKGPU_OK;
}

int utf8_codes(flockgpu_ctx *ctx, const char *name, const DevColumn &build, int64_t n_build, int64_t *build_codes, const DevColumn *probe,
               int64_t n_probe, int64_t *probe_codes) {
    const std::string base = name;
    if (build.type != ColType::UTF8 || (probe && probe->type != ColType::UTF8)) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: not a Utf8 column", name);
    if (n_build >= (int64_t(1) << 30)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 2^30 rows in a Utf8 dictionary", name);
    const uint64_t cap = pow2_at_least((uint64_t)std::max<int64_t>(n_build, 1) * 2);
    int32_t *table = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".dict").c_str(), (size_t)cap, &table));
    RELOPS_LAUNCH(ctx, "fill_i32_kernel", fill_i32_kernel, (int64_t)cap, table, (int64_t)cap, (int32_t)-1);
    if (n_build > 0)
        RELOPS_LAUNCH(ctx, "utf8_codes_build_kernel", utf8_codes_build_kernel, n_build, build.offsets, static_cast<const uint8_t *>(build.values), n_build,
                      table, cap, build_codes);
    if (probe && probe_codes && n_probe > 0)
        RELOPS_LAUNCH(ctx, "utf8_codes_probe_kernel", utf8_codes_probe_kernel, n_probe, build.offsets, static_cast<const uint8_t *>(build.values), table,
                      cap, probe->offsets, static_cast<const uint8_t *>(probe->values), n_probe, probe_codes);
    return FLOCKGPU_OK;
}

int reduce_max(flockgpu_ctx *ctx, const DevColumn &col, int64_t rows, int64_t *out, int *any) {
    *out = 0;
    *any = rows > 0;
    if (col.type == ColType::UTF8 || col.type == ColType::F64) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "MAX needs an integer column");
    if (rows <= 0) return FLOCKGPU_OK;
    const unsigned blocks = std::min<unsigned>(grid_for(ctx, rows), 1024);
    int64_t *h = nullptr;
    FG_TRY(pinned_get_t(ctx, "relops.max", 2048, &h));
    {
        LaunchScope ls(ctx, "max_kernel");      // (per-block results straight into pinned memory)
        hipLaunchKernelGGL(max_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, col.values, col.valid, (int32_t)col.type, rows, h);
    }
    FG_TRY(check_launch(ctx, "max_kernel"));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const bool uns = col.type == ColType::U64;
    int64_t m = uns ? 0 : INT64_MIN, seen = 0;
    for (unsigned b = 0; b < blocks; ++b) {
        if (h[blocks + b] == 0) continue;   // (a block that saw no valid row holds the identity)
        seen += h[blocks + b];
        if (uns ? (uint64_t)h[b] > (uint64_t)m : h[b] > m) m = h[b];
    }
    *any = seen > 0;   // MAX over nothing but NULLs is NULL, as over no rows
    *out = seen > 0 ? m : 0;
    return FLOCKGPU_OK;
}

bool join_is_tiny(int64_t n_left, int64_t n_right) {
    return n_left > 0 && n_right > 0 && std::min(n_left, n_right) <= kTinyBuild && std::max(n_left, n_right) <= kTinyProbe;
}

int join_key64(flockgpu_ctx *ctx, const char *name, const int64_t *left, int64_t n_left, const int64_t *right, int64_t n_right,
               int32_t **left_rows, int32_t **right_rows, int64_t *n_pairs) {
    const std::string base = name;
    *n_pairs = 0;
    *left_rows = *right_rows = nullptr;
    // The hash table is built on the SMALLER side (DataFusion builds on the left; which side is hashed is unobservable in the pair
    // multiset): q5's final stage joins 6e5 (auction, num) groups with ONE row of MAX(num) on num = maxn -- built on the left, every
    // group with the same count chained onto one slot by compare-and-swap (1.39 ms of the staged q5's 2.1 ms of kernels).
    if (n_left > 4 * n_right && n_left > 4096) {
        const int rc = join_key64(ctx, (base + ".swapped").c_str(), right, n_right, left, n_left, right_rows, left_rows, n_pairs);
        return rc;
    }
    if (join_is_tiny(n_left, n_right)) {
        // both sides small: the whole join is one workgroup's work (join_tiny_kernel); the table goes on the smaller side
        const bool build_left = n_left <= n_right;
        const int64_t *bk = build_left ? left : right, *pk = build_left ? right : left;
        const int64_t nb = build_left ? n_left : n_right, np = build_left ? n_right : n_left;
        std::vector<int64_t> &hint = ctx->host_i64[base + ".pairs_hint"];   // {pairs of the last call under this name}
        uint64_t cap_pairs = (uint64_t)std::max<int64_t>(np + 1024, hint.empty() ? 0 : hint[0] + hint[0] / 4 + 1024);
        unsigned long long *h_tot = nullptr;   // (pinned: the kernel's one thread stores the total where the host reads it)
        FG_TRY(pinned_get_t(ctx, (base + ".tot64").c_str(), 2, &h_tot));
        int32_t *ob = nullptr, *op = nullptr;
        for (;;) {
            FG_TRY(arena_get_t(ctx, (base + ".ol").c_str(), (size_t)cap_pairs + 4, &ob));
            FG_TRY(arena_get_t(ctx, (base + ".or").c_str(), (size_t)cap_pairs + 4, &op));
            pinned_pending(reinterpret_cast<uint64_t *>(h_tot), 1);
            {
                LaunchScope ls(ctx, "join_tiny_kernel");
                hipLaunchKernelGGL(join_tiny_kernel, dim3(1), dim3(kTinyThreads), 0, ctx->stream, bk, (int32_t)nb, pk, (int32_t)np, ob, op,
                                   (uint32_t)std::min<uint64_t>(cap_pairs, 0x7fffffffu), h_tot);
            }
            FG_TRY(check_launch(ctx, "join_tiny_kernel"));
            FG_TRY(wait_pinned(ctx, reinterpret_cast<const uint64_t *>(h_tot), 1));
            if (h_tot[0] >= (1ull << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: join output of %llu rows exceeds 2^31", name, h_tot[0]);
            if (h_tot[0] <= cap_pairs) break;
            cap_pairs = h_tot[0];   // the estimate was too small: once more, with room for every pair
        }
        hint.assign(1, (int64_t)h_tot[0]);
        *left_rows = build_left ? ob : op;
        *right_rows = build_left ? op : ob;
        *n_pairs = (int64_t)h_tot[0];
        return FLOCKGPU_OK;
    }
    DevColumn cl, cr;
    cl.type = cr.type = ColType::I64;
    cl.values = left;
    cr.values = right;
    return join_hashed(ctx, name, cl, n_left, cr, n_right, left_rows, right_rows, n_pairs);
}

int join_hashed(flockgpu_ctx *ctx, const char *name, const DevColumn &left, int64_t n_left, const DevColumn &right, int64_t n_right,
                int32_t **left_rows, int32_t **right_rows, int64_t *n_pairs) {
    const std::string base = name;
    *n_pairs = 0;
    *left_rows = *right_rows = nullptr;
    auto is_int = [](const DevColumn &c) { return c.type == ColType::I32 || c.type == ColType::I64 || c.type == ColType::U64; };
    if (!is_int(left) || !is_int(right) || left.valid || right.valid || ((left.type == ColType::U64) != (right.type == ColType::U64)))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: join keys must be integer columns of one signedness without NULLs", name);
    if (n_left > 4 * n_right && n_left > 4096)   // the table goes on the smaller side (see join_key64)
        return join_hashed(ctx, (base + ".swapped").c_str(), right, n_right, left, n_left, right_rows, left_rows, n_pairs);
    if (n_left >= (int64_t(1) << 30) || n_right >= (int64_t(1) << 31))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: relation too large for the generic join", name);
    const uint64_t cap = pow2_at_least((uint64_t)std::max<int64_t>(n_left, 1) * 2);
    const int64_t n_slots = (int64_t)cap + 1;
    JoinSlot *slots = nullptr;
    int32_t *next = nullptr, *counts = nullptr, *ol = nullptr, *orr = nullptr;
    // the call's scalars, one block on the device and one in pinned memory: [0] error word, [1] duplicate build keys, [2..] the 64-bit pair total
    uint32_t *d_err = nullptr, *h_err = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".slots").c_str(), (size_t)n_slots, &slots));
    FG_TRY(arena_get_t(ctx, (base + ".next").c_str(), (size_t)std::max<int64_t>(n_left, 0) + 4, &next));
    FG_TRY(arena_get_t(ctx, (base + ".scalars").c_str(), 2 + 2 * (size_t)kJoinTotalSlots, &d_err));
    FG_TRY(pinned_get_t(ctx, (base + ".scalars").c_str(), 2 + 2 * (size_t)kJoinTotalSlots, &h_err));
    if (n_left <= 0 || n_right <= 0) {
        FG_TRY(arena_get_t(ctx, (base + ".ol").c_str(), 4, &ol));
        FG_TRY(arena_get_t(ctx, (base + ".or").c_str(), 4, &orr));
        *left_rows = ol;
        *right_rows = orr;
        return FLOCKGPU_OK;
    }
    unsigned long long *d_tot64 = reinterpret_cast<unsigned long long *>(d_err + 2), *h_tot64 = reinterpret_cast<unsigned long long *>(h_err + 2);
    FG_TRY(fill_words(ctx, FillList().add(d_err, 0u, 2 + 2 * kJoinTotalSlots)));
    RELOPS_LAUNCH(ctx, "join_hash_init_kernel", join_hash_init_kernel, n_slots, slots, n_slots);
    RELOPS_LAUNCH(ctx, "join_hash_build_kernel", join_hash_build_kernel, n_left, left.values, (int32_t)left.type, n_left, slots, cap, next, d_err);
    // Unique build keys (nothing says so before the build has run: this node's previous execute is the guess): the probe as a filter in the
    // flag-tile geometry.  The build reports duplicates in err[1]; a wrong guess repeats the probe the general way below.
    std::vector<int64_t> &dups_seen = ctx->host_i64[base + ".dups"];
    if (dups_seen.empty()) {
        int64_t sb = 0, se = n_right;
        SegTiles st;
        FG_TRY(build_seg_tiles(ctx, (base + ".ptiles").c_str(), &sb, &se, 1, kFlagTile, &st));
        uint32_t *flags = nullptr, *wcounts = nullptr;
        uint64_t *tile_base = nullptr;
        int64_t *h_off = nullptr;
        int32_t *match = nullptr;
        FG_TRY(arena_get_t(ctx, (base + ".pflags").c_str(), (size_t)st.n_tiles * kBlock + 4, &flags));
        FG_TRY(arena_get_t(ctx, (base + ".pcounts").c_str(), (size_t)st.n_tiles * kWavesPerBlock + 4, &wcounts));
        FG_TRY(arena_get_t(ctx, (base + ".pbase").c_str(), (size_t)st.n_tiles + 1, &tile_base));
        FG_TRY(arena_get_t(ctx, (base + ".match").c_str(), (size_t)n_right + 4, &match));
        FG_TRY(pinned_get_t(ctx, (base + ".poff").c_str(), 2, &h_off));
        FG_TRY(arena_get_t(ctx, (base + ".or").c_str(), (size_t)n_right + 4, &orr));   // (at most one pair per probe row)
        pinned_pending(reinterpret_cast<uint64_t *>(h_off), 2);
        {
            LaunchScope ls(ctx, "join_hash_probe_flag_kernel");
            if (right.type == ColType::I32)
                hipLaunchKernelGGL(join_hash_probe_flag_kernel<true>, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, right.values, n_right, st, slots, cap, flags, wcounts, match);
            else
                hipLaunchKernelGGL(join_hash_probe_flag_kernel<false>, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, right.values, n_right, st, slots, cap, flags, wcounts, match);
        }
        FG_TRY(check_launch(ctx, "join_hash_probe_flag_kernel"));
        pinned_pending32(h_err, 2);
        FG_TRY(publish_words(ctx, PublishList().add(h_err, d_err, 2)));
        if (st.n_tiles <= 2048) {
            FG_TRY(emit_flagged_rows_self(ctx, st, flags, wcounts, orr, h_off));
        } else {
            FG_TRY(launch_tile_scan(ctx, wcounts, st.n_tiles, tile_base, st.tile_first, st.n_seg, h_off));
            FG_TRY(emit_flagged_rows(ctx, st, flags, wcounts, tile_base, orr));
        }
        FG_TRY(wait_pinned(ctx, reinterpret_cast<const uint64_t *>(h_off), 2));
        FG_TRY(wait_pinned32(ctx, h_err, 2));
        if (h_err[0]) return fail(ctx, FLOCKGPU_ERR_CAPACITY, "%s: join table overflow", name);
        if (!h_err[1]) {
            const int64_t total = h_off[1];
            FG_TRY(arena_get_t(ctx, (base + ".ol").c_str(), (size_t)total + 4, &ol));
            if (total > 0) RELOPS_LAUNCH(ctx, "join_match_take_kernel", join_match_take_kernel, total, match, orr, total, ol);
            *left_rows = ol;
            *right_rows = orr;
            *n_pairs = total;
            return FLOCKGPU_OK;
        }
        dups_seen.assign(1, 1);   // duplicates on the build side: counts and chains from here on (the table and the links are built; the probe follows)
    }
    FG_TRY(arena_get_t(ctx, (base + ".counts").c_str(), (size_t)std::max<int64_t>(n_right, 0) + 4, &counts));
    RELOPS_LAUNCH(ctx, "join_hash_probe_kernel", join_hash_probe_kernel<false>, n_right, right.values, (int32_t)right.type, n_right, slots, cap, next, counts, (int32_t *)nullptr,
                  (int32_t *)nullptr, d_tot64);
    FG_TRY(inclusive_scan_i32(ctx, (base + ".scan").c_str(), counts, n_right));
    pinned_pending32(h_err, 2 + 2 * kJoinTotalSlots);
    FG_TRY(publish_words(ctx, PublishList().add(h_err, d_err, 2).add(h_tot64, d_tot64, 2 * kJoinTotalSlots)));
    FG_TRY(wait_pinned32(ctx, h_err, 2 + 2 * kJoinTotalSlots));
    for (int sl = 1; sl < kJoinTotalSlots; ++sl) h_tot64[0] += h_tot64[sl];
    if (h_err[0]) return fail(ctx, FLOCKGPU_ERR_CAPACITY, "%s: join table overflow", name);
    // the 64-bit total decides: the 32-bit inclusive scan of `counts` is only read when it cannot have wrapped
    if (h_tot64[0] >= (1ull << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: join output of %llu rows exceeds 2^31", name, h_tot64[0]);
    const int64_t total = (int64_t)h_tot64[0];
    FG_TRY(arena_get_t(ctx, (base + ".ol").c_str(), (size_t)total + 4, &ol));
    FG_TRY(arena_get_t(ctx, (base + ".or").c_str(), (size_t)total + 4, &orr));
    if (total > 0)
        RELOPS_LAUNCH(ctx, "join_hash_probe_kernel", join_hash_probe_kernel<true>, n_right, right.values, (int32_t)right.type, n_right, slots, cap, next, counts, ol, orr,
                      (unsigned long long *)nullptr);
    *left_rows = ol;
    *right_rows = orr;
    *n_pairs = total;
    return FLOCKGPU_OK;
}

int join_dense(flockgpu_ctx *ctx, const char *name, const DevColumn &left, int64_t n_left, int64_t kmin, int64_t kmax, const DevColumn &right, int64_t n_right,
               int32_t **left_rows, int32_t **right_rows, int64_t *n_pairs) {
    const std::string base = name;
    *n_pairs = 0;
    *left_rows = *right_rows = nullptr;
    auto is_int = [](const DevColumn &c) { return c.type == ColType::I32 || c.type == ColType::I64 || c.type == ColType::U64; };
    if (!is_int(left) || !is_int(right) || left.valid || right.valid || ((left.type == ColType::U64) != (right.type == ColType::U64)) ||
        !dense_range_ok(kmin, kmax, n_left, left.type == ColType::U64) || n_left >= (int64_t(1) << 31) || n_right >= (int64_t(1) << 31))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: not a dense integer join key", name);
    const uint32_t range = (uint32_t)((uint64_t)kmax - (uint64_t)kmin + 1);
    int32_t *head = nullptr, *next = nullptr, *counts = nullptr, *ol = nullptr, *orr = nullptr;
    uint32_t *d_err = nullptr, *h_err = nullptr;
    unsigned long long *d_tot64 = nullptr, *h_tot64 = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".dhead").c_str(), (size_t)range + 4, &head));
    FG_TRY(arena_get_t(ctx, (base + ".next").c_str(), (size_t)std::max<int64_t>(n_left, 0) + 4, &next));
    FG_TRY(arena_get_t(ctx, (base + ".counts").c_str(), (size_t)std::max<int64_t>(n_right, 0) + 4, &counts));
    FG_TRY(arena_get_t(ctx, (base + ".scalars").c_str(), 2 + 2 * (size_t)kJoinTotalSlots, &d_err));     // [0] error word, [2..] the 64-bit pair total in kJoinTotalSlots words
    FG_TRY(pinned_get_t(ctx, (base + ".scalars").c_str(), 2 + 2 * (size_t)kJoinTotalSlots, &h_err));
    d_tot64 = reinterpret_cast<unsigned long long *>(d_err + 2);
    h_tot64 = reinterpret_cast<unsigned long long *>(h_err + 2);
    if (n_left <= 0 || n_right <= 0) {
        FG_TRY(arena_get_t(ctx, (base + ".ol").c_str(), 4, &ol));
        FG_TRY(arena_get_t(ctx, (base + ".or").c_str(), 4, &orr));
        *left_rows = ol;
        *right_rows = orr;
        return FLOCKGPU_OK;
    }
    FG_TRY(fill_words(ctx, FillList().add(d_err, 0u, 2 + 2 * kJoinTotalSlots).add(head, 0xffffffffu, range)));   // scalars 0; chain heads -1 (empty): one launch
    RELOPS_LAUNCH(ctx, "join_build_dense_kernel", join_build_dense_kernel, n_left, left.values, (int32_t)left.type, n_left, kmin, range, head, next, d_err);
    // Unique build keys (nothing says so before the build has run: this node's previous execute is the guess): the probe as a filter in the
    // flag-tile geometry.  The build reports duplicates in err[1]; a wrong guess repeats the probe the general way below.
    std::vector<int64_t> &dups_seen = ctx->host_i64[base + ".dups"];
    if (dups_seen.empty()) {
        int64_t sb = 0, se = n_right;
        SegTiles st;
        FG_TRY(build_seg_tiles(ctx, (base + ".ptiles").c_str(), &sb, &se, 1, kFlagTile, &st));
        uint32_t *flags = nullptr, *wcounts = nullptr;
        uint64_t *tile_base = nullptr;
        int64_t *h_off = nullptr;
        FG_TRY(arena_get_t(ctx, (base + ".pflags").c_str(), (size_t)st.n_tiles * kBlock + 4, &flags));
        FG_TRY(arena_get_t(ctx, (base + ".pcounts").c_str(), (size_t)st.n_tiles * kWavesPerBlock + 4, &wcounts));
        FG_TRY(arena_get_t(ctx, (base + ".pbase").c_str(), (size_t)st.n_tiles + 1, &tile_base));
        FG_TRY(pinned_get_t(ctx, (base + ".poff").c_str(), 2, &h_off));
        FG_TRY(arena_get_t(ctx, (base + ".or").c_str(), (size_t)n_right + 4, &orr));   // (at most one pair per probe row)
        pinned_pending(reinterpret_cast<uint64_t *>(h_off), 2);
        {
            LaunchScope ls(ctx, "join_probe_unique_flag_kernel");
            if (right.type == ColType::I32)
                hipLaunchKernelGGL(join_probe_unique_flag_kernel<true>, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, right.values, n_right, st, kmin, range, head, flags, wcounts);
            else
                hipLaunchKernelGGL(join_probe_unique_flag_kernel<false>, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, right.values, n_right, st, kmin, range, head, flags, wcounts);
        }
        FG_TRY(check_launch(ctx, "join_probe_unique_flag_kernel"));
        pinned_pending32(h_err, 2);
        FG_TRY(publish_words(ctx, PublishList().add(h_err, d_err, 2)));
        if (st.n_tiles <= 2048) {
            FG_TRY(emit_flagged_rows_self(ctx, st, flags, wcounts, orr, h_off));
        } else {
            FG_TRY(launch_tile_scan(ctx, wcounts, st.n_tiles, tile_base, st.tile_first, st.n_seg, h_off));
            FG_TRY(emit_flagged_rows(ctx, st, flags, wcounts, tile_base, orr));
        }
        FG_TRY(wait_pinned(ctx, reinterpret_cast<const uint64_t *>(h_off), 2));
        FG_TRY(wait_pinned32(ctx, h_err, 2));
        if (h_err[0]) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: a build key outside the column statistics [%lld, %lld] the table was sized from", name, (long long)kmin, (long long)kmax);
        if (!h_err[1]) {
            const int64_t total = h_off[1];
            FG_TRY(arena_get_t(ctx, (base + ".ol").c_str(), (size_t)total + 4, &ol));
            if (total > 0)
                RELOPS_LAUNCH(ctx, "join_unique_take_kernel", join_unique_take_kernel, total, right.values, (int32_t)right.type, orr, total, kmin, head, ol);
            *left_rows = ol;
            *right_rows = orr;
            *n_pairs = total;
            return FLOCKGPU_OK;
        }
        dups_seen.assign(1, 1);   // duplicates on the build side: chains from here on (the table and the links are built; the probe follows)
    }
    RELOPS_LAUNCH(ctx, "join_probe_dense_kernel", join_probe_dense_kernel<false>, n_right, right.values, (int32_t)right.type, n_right, kmin, range, head, next, counts,
                  (int32_t *)nullptr, (int32_t *)nullptr, d_tot64);
    FG_TRY(inclusive_scan_i32(ctx, (base + ".scan").c_str(), counts, n_right));
    pinned_pending32(h_err, 2 + 2 * kJoinTotalSlots);
    FG_TRY(publish_words(ctx, PublishList().add(h_err, d_err, 2).add(h_tot64, d_tot64, 2 * kJoinTotalSlots)));
    FG_TRY(wait_pinned32(ctx, h_err, 2 + 2 * kJoinTotalSlots));
    for (int sl = 1; sl < kJoinTotalSlots; ++sl) h_tot64[0] += h_tot64[sl];
    if (h_err[0]) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: a build key outside the column statistics [%lld, %lld] the table was sized from", name, (long long)kmin, (long long)kmax);
    if (h_tot64[0] >= (1ull << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: join output of %llu rows exceeds 2^31", name, h_tot64[0]);
    const int64_t total = (int64_t)h_tot64[0];
    FG_TRY(arena_get_t(ctx, (base + ".ol").c_str(), (size_t)total + 4, &ol));
    FG_TRY(arena_get_t(ctx, (base + ".or").c_str(), (size_t)total + 4, &orr));
    if (total > 0)
        RELOPS_LAUNCH(ctx, "join_probe_dense_kernel", join_probe_dense_kernel<true>, n_right, right.values, (int32_t)right.type, n_right, kmin, range, head, next, counts, ol,
                      orr, (unsigned long long *)nullptr);
    *left_rows = ol;
    *right_rows = orr;
    *n_pairs = total;
    return FLOCKGPU_OK;
}

int key_run_starts(flockgpu_ctx *ctx, const char *name, const int64_t *keys, const int32_t *order, int64_t rows, int32_t **starts, int64_t *n_keys) {
    const std::string base = name;
    *starts = nullptr;
    *n_keys = 0;
    if (rows <= 0) return FLOCKGPU_OK;
    uint8_t *mask = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".mask").c_str(), (size_t)rows + 16, &mask));
    RELOPS_LAUNCH(ctx, "key_run_start_kernel", key_run_start_kernel, rows, keys, order, rows, mask);
    return mask_to_rows(ctx, (base + ".sel").c_str(), mask, rows, starts, n_keys);
}

int row_number_runs(flockgpu_ctx *ctx, const char *name, const DevColumn *cols, int n_cols, int64_t rows, uint64_t *out) {
    if (n_cols < 0 || n_cols > 4) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: PARTITION BY more than four columns", name);
    if (rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 2^31 rows", name);
    if (rows <= 0) return FLOCKGPU_OK;
    RunKeys k{};
    k.n = n_cols;
    for (int c = 0; c < n_cols; ++c) {
        if (cols[c].type == ColType::UTF8) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: PARTITION BY a Utf8 column", name);
        k.v[c] = cols[c].values;
        k.valid[c] = cols[c].valid;
        k.type[c] = (int32_t)cols[c].type;
    }
    const std::string base = name;
    int32_t *run = nullptr, *first = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".run").c_str(), (size_t)rows + 4, &run));
    FG_TRY(arena_get_t(ctx, (base + ".first").c_str(), (size_t)rows + 4, &first));
    RELOPS_LAUNCH(ctx, "run_start_kernel", run_start_kernel, rows, k, rows, run);
    FG_TRY(inclusive_scan_i32(ctx, (base + ".scan").c_str(), run, rows));
    RELOPS_LAUNCH(ctx, "run_first_kernel", run_first_kernel, rows, run, rows, first);
    RELOPS_LAUNCH(ctx, "run_rank_kernel", run_rank_kernel, rows, run, first, rows, out);
    return FLOCKGPU_OK;
}

int partition_rows_key64(flockgpu_ctx *ctx, const char *name, const int64_t *keys, int64_t rows, int32_t n_parts, int32_t **out_rows,
                         std::vector<int64_t> *part_offsets) {
    const std::string base = name;
    part_offsets->assign((size_t)n_parts + 1, 0);
    *out_rows = nullptr;
    int32_t *folded = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".fold").c_str(), (size_t)std::max<int64_t>(rows, 0) + 4, &folded));
    if (rows > 0) RELOPS_LAUNCH(ctx, "fold_key_kernel", fold_key_kernel, rows, keys, rows, folded);
    int64_t off[2] = {0, rows};
    int32_t lo = 0, hi = 1;
    const flockgpu_windows w{off, 1, &lo, &hi, 1};
    flockgpu_partition_result r{};
    FG_TRY(flockgpu_partition_by_key(ctx, folded, rows, &w, n_parts, &r));
    // the partition result is overwritten by the next partition call on this ctx: keep a copy under this node's name
    int32_t *keep = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".prow").c_str(), (size_t)std::max<int64_t>(r.rows, 0) + 4, &keep));
    if (r.rows > 0) FG_HIP(ctx, hipMemcpyAsync(keep, r.row, sizeof(int32_t) * (size_t)r.rows, hipMemcpyDeviceToDevice, ctx->stream));
    for (int32_t p = 0; p <= n_parts; ++p) (*part_offsets)[p] = r.part_win_offsets[p];
    *out_rows = keep;
    return FLOCKGPU_OK;
}

}  // namespace flockgpu
