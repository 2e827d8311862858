// Stable grouping of rows by an int32 key (sort.hip): LSD radix passes, each a count -> scan -> emit over 4096-row
// tiles.  It is what the reference's `repartition(.., HashDiff([key], distinct_keys))` produces for the session and
// global windows (flock-function/src/aws/window/session.rs:242-250): every key's rows together, arrival order kept.
#pragma once
#include "gather.hpp"

namespace flockgpu {

// Sorts the n pairs (keys[i], vals ? vals[i] : i) by  keys[i] - bias  (an unsigned value below 2^bits), stable.
// The results live in the ctx arena under `name` (double buffers `name.k0/k1/v0/v1`); the inputs are not written.
int radix_sort_pairs(flockgpu_ctx *ctx, const char *name, const int32_t *keys, const uint32_t *vals, int64_t n, int32_t bias,
                     int bits, int32_t **out_keys, uint32_t **out_vals);

// Minimum and maximum of keys[0 .. n) into d_minmax[0], d_minmax[1] (device, initialised by the call).
int key_min_max(flockgpu_ctx *ctx, const int32_t *keys, int64_t n, int32_t *d_minmax);

// d_off[k] = first position of the ascending `sorted_key` (values in [0, n_keys]) whose key is >= k, k = 0 .. n_keys.
int sorted_key_offsets(flockgpu_ctx *ctx, const int32_t *sorted_key, int64_t n, int32_t n_keys, int64_t *d_off);

}  // namespace flockgpu
