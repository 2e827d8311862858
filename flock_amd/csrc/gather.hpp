// take() for fixed-width and Utf8 columns on the device (the `take` calls DataFusion's HashJoinExec issues
// to materialise join output).  Utf8 needs lengths -> exclusive scan -> byte copy.
#pragma once
#include "scan.hpp"

namespace flockgpu {

int gather_i32(flockgpu_ctx *ctx, const int32_t *src, const int32_t *rows, int64_t n, int32_t *out);

// Gathers `n` Utf8 values.  out_off (n + 1 entries) and out_bytes live in the ctx arena under `name`;
// *n_bytes receives the total byte count (host value; the call synchronises the stream).
int gather_utf8(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 &src, const int32_t *rows, int64_t n,
                flockgpu_utf8 *out, int64_t *n_bytes);

}  // namespace flockgpu

namespace flockgpu {
// In-place inclusive scan of n int32 values (single-pass chained scan); `name` keys the scan-state arena buffer.
int inclusive_scan_i32(flockgpu_ctx *ctx, const char *name, int32_t *data, int64_t n);
}  // namespace flockgpu
