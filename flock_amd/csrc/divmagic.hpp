// Remainder of an unsigned 32-bit value by a divisor known on the host, as one 32x32 -> high-32 multiply plus a few adds
// and shifts (Granlund / Montgomery "division by invariant integers", the branch-full round-up variant): what the generic
// predicate kernel (pred.hip) does for `CAST(col AS Int64) % m` on an Int32 column -- the FilterExec of the reference's
// arch/ops/filter.sql and q2 (flock/src/distributed_plan/planner.rs:120-124) -- instead of a 64-bit hardware-less division
// (~150 instructions per row on gfx950, which would make a 4-byte-per-row filter compute-bound).
// Plain C++ (no HIP types): tests/cpp/divmagic_test.cpp checks it against `%` with g++ on the CPU.
#pragma once
#include <cstdint>

#ifndef FLOCKGPU_HD
#ifdef __HIPCC__
#define FLOCKGPU_HD __host__ __device__ __forceinline__
#else
#define FLOCKGPU_HD inline
#endif
#endif

namespace flockgpu {

struct UMod32 {
    uint32_t d = 1;      // the divisor, 1 <= d < 2^32
    uint32_t magic = 0;  // 0: d is a power of two
    uint32_t shift = 0;
    uint32_t add = 0;    // the 33-bit multiplier case
};

inline UMod32 umod32_make(uint32_t d) {
    UMod32 m;
    m.d = d ? d : 1;
    uint32_t k = 31;
    while (!((m.d >> k) & 1u)) --k;  // floor(log2 d)
    m.shift = k;
    if ((m.d & (m.d - 1)) == 0) return m;  // power of two: magic = 0
    const uint64_t two = uint64_t(1) << (32 + k);
    uint64_t pm = two / m.d;
    const uint64_t rem = two - pm * m.d;
    const uint64_t e = m.d - rem;
    if (e < (uint64_t(1) << k)) {
        m.add = 0;
    } else {  // one more bit of multiplier: 2 * pm (+ 1 when the doubled remainder reaches d)
        pm *= 2;
        const uint64_t twice = rem * 2;
        if (twice >= m.d) pm += 1;
        m.add = 1;
    }
    m.magic = (uint32_t)(pm + 1);
    return m;
}

FLOCKGPU_HD uint32_t umod32_apply(uint32_t n, const UMod32 &m) {
    uint32_t q;
    if (m.magic == 0) {
        q = n >> m.shift;
    } else {
        const uint32_t hi = (uint32_t)(((uint64_t)n * m.magic) >> 32);
        q = m.add ? ((((n - hi) >> 1) + hi) >> m.shift) : (hi >> m.shift);
    }
    return n - q * m.d;
}

// Truncated remainder of a signed 32-bit value by m (|m| = mm.d): the sign of the dividend, as Rust's / Arrow's `%`.
FLOCKGPU_HD int32_t smod32_apply(int32_t x, const UMod32 &mm) {
    const uint32_t a = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
    const uint32_t r = umod32_apply(a, mm);
    return x < 0 ? -(int32_t)r : (int32_t)r;
}

}  // namespace flockgpu
