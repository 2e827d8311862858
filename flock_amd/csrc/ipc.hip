// Arrow IPC record-batch BODY assembly on the device: the reference turns every output
// batch into Arrow Flight data (`flight_data_from_arrow_batch`, flock/src/transmute.rs:155-170,190-205) whose body is the
// batch's buffers laid end to end, each padded to 8 bytes (arrow IpcWriteOptions::default()).  With the columns resident in
// HBM that concatenation is one batched copy, so the host receives the finished body with ONE D2H transfer and only has to
// write the (tiny) flatbuffer header next to it (flock_amd/payload.py); compression stays on the CPU as in the reference.
#include <algorithm>

#include "gather.hpp"

using namespace flockgpu;

namespace {

constexpr int kChunk = 16 * 1024;  // bytes per workgroup

struct PackDesc {
    const uint8_t *src;
    int64_t dst_off;    // multiple of 8
    int64_t bytes;      // payload bytes; the padding up to the next multiple of 8 is zero-filled
    int64_t first_chunk;
};

// chunk c of the flattened (buffer, chunk) list: binary search for its buffer, copy 16 KiB
__global__ __launch_bounds__(kBlock) void ipc_pack_kernel(const PackDesc *__restrict__ desc, int32_t n_desc, uint8_t *__restrict__ dst) {
    const int64_t c = blockIdx.x;
    int32_t lo = 0, hi = n_desc;
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (desc[mid].first_chunk <= c) lo = mid;
        else hi = mid;
    }
    const PackDesc d = desc[lo];
    const int64_t b0 = (c - d.first_chunk) * kChunk;
    const int64_t padded = (d.bytes + 7) & ~int64_t(7);
    const int64_t b1 = std::min<int64_t>(b0 + kChunk, padded);
    const bool aligned = ((reinterpret_cast<uintptr_t>(d.src) | (uintptr_t)d.dst_off) & 15) == 0;
    if (aligned) {
        for (int64_t o = b0 + (int64_t)threadIdx.x * 16; o < b1; o += kBlock * 16) {
            if (o + 16 <= d.bytes) {
                *reinterpret_cast<uint4 *>(dst + d.dst_off + o) = *reinterpret_cast<const uint4 *>(d.src + o);
            } else {
                for (int64_t i = o; i < std::min<int64_t>(o + 16, b1); ++i) dst[d.dst_off + i] = i < d.bytes ? d.src[i] : 0;
            }
        }
    } else {
        for (int64_t i = b0 + threadIdx.x; i < b1; i += kBlock) dst[d.dst_off + i] = i < d.bytes ? d.src[i] : 0;
    }
}

}  // namespace

extern "C" {

int flockgpu_ipc_pack_body(flockgpu_ctx *ctx, const flockgpu_ipc_buffer *buffers, int32_t n_buffers, uint8_t *out, int64_t out_capacity,
                           int64_t *out_bytes) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!out_bytes || n_buffers < 0 || (n_buffers > 0 && !buffers)) return fail(ctx, FLOCKGPU_ERR_INVALID, "ipc_pack: null argument");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<PackDesc> d;
    int64_t off = 0, chunks = 0;
    for (int32_t i = 0; i < n_buffers; ++i) {
        if (buffers[i].bytes < 0 || (buffers[i].bytes > 0 && !buffers[i].data)) return fail(ctx, FLOCKGPU_ERR_INVALID, "ipc_pack: bad buffer %d", i);
        if (buffers[i].bytes > 0) {
            d.push_back(PackDesc{static_cast<const uint8_t *>(buffers[i].data), off, buffers[i].bytes, chunks});
            chunks += div_up((buffers[i].bytes + 7) & ~int64_t(7), kChunk);
        }
        off += (buffers[i].bytes + 7) & ~int64_t(7);
    }
    *out_bytes = off;
    if (!out) return FLOCKGPU_OK;  // size query
    if (out_capacity < off) return fail(ctx, FLOCKGPU_ERR_CAPACITY, "ipc_pack: body needs %lld bytes, %lld given", (long long)off, (long long)out_capacity);
    if (reinterpret_cast<uintptr_t>(out) & 7) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "ipc_pack: output must be 8-byte aligned");
    if (d.empty()) return FLOCKGPU_OK;
    if (chunks > 0x7fffffff) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "ipc_pack: body too large for one call");
    PackDesc *d_desc = nullptr, *h_desc = nullptr;
    FG_TRY(arena_get_t(ctx, "ipc.desc", d.size(), &d_desc));
    FG_TRY(pinned_get_t(ctx, "ipc.desc", d.size(), &h_desc));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the pinned descriptors of the previous call may still be in flight
    std::copy(d.begin(), d.end(), h_desc);
    FG_HIP(ctx, hipMemcpyAsync(d_desc, h_desc, sizeof(PackDesc) * d.size(), hipMemcpyHostToDevice, ctx->stream));
    {
        LaunchScope ls(ctx, "ipc_pack_kernel");
        hipLaunchKernelGGL(ipc_pack_kernel, dim3((unsigned)chunks), dim3(kBlock), 0, ctx->stream, d_desc, (int32_t)d.size(), out);
    }
    return check_launch(ctx, "ipc_pack_kernel");
}

}  // extern "C"
