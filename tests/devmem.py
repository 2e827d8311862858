"""Where the GPU tests' hand-made input columns live.  Normally: a torch tensor.  With FLOCK_TEST_GUARDED=1 (tests/test_gpu_guard.py
re-runs the direct-call tests that way, in a process of its own): memory that ENDS where mapped address space ends
(flockgpu_malloc_guarded), so a kernel that reads or writes past the end of an input column faults instead of landing in whatever
the allocator put next to it."""
import os

import numpy as np

_GUARD_CTX = None


def guarded() -> bool:
    return os.environ.get("FLOCK_TEST_GUARDED", "") not in ("", "0")


def dev(a):
    a = np.ascontiguousarray(a)
    if guarded():
        global _GUARD_CTX
        if _GUARD_CTX is None:
            from flock_amd import GpuContext
            _GUARD_CTX = GpuContext(0)          # owns the guarded allocations until the process ends
        return _GUARD_CTX.guarded(a)
    import torch
    return torch.from_numpy(a).cuda()


def guard_stream(g, max_rows=30_000_000):
    """Under FLOCK_TEST_GUARDED=1: every device column of a generated stream (bids / auctions / persons, Utf8 offsets and bytes)
    moved into guarded memory -- same values, so the oracle comparisons of the generator-based tests hold as they are.  Streams beyond
    `max_rows` rows stay where the generator put them (the copy goes through the host)."""
    if not guarded():
        return g
    from flock_amd import DeviceUtf8
    rels = [getattr(g, k, None) for k in ("bids", "auctions", "persons")]
    if sum(int(getattr(r, "rows", 0) or 0) for r in rels if r is not None) > max_rows:
        return g
    for r in rels:
        if r is None:
            continue
        for name, v in list(vars(r).items()):
            if isinstance(v, DeviceUtf8):
                off = v.offsets.cpu().numpy()
                setattr(r, name, DeviceUtf8(dev(off), dev(v.data.cpu().numpy()[: int(off[-1])])))   # (the bytes end with the last value)
            elif hasattr(v, "data_ptr") and hasattr(v, "cpu"):
                setattr(r, name, dev(v.cpu().numpy()))
    return g
