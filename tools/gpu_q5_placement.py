#!/usr/bin/env python3
"""gpurun experiment: does q5_count_kernel's duration depend on WHERE the 4 GB `auction` column sits in HBM?
The same column is cloned to K addresses (the allocations in between keep them apart), the query runs N times on each copy
with only q5_count_kernel bracketed by HIP events; then the copies are visited again in reverse order (drift vs placement).
Output: one line per (pass, copy): device address, address bits 21..35, average / min / max kernel ms."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from flock_amd import Bids, GpuContext, NEXMarkSource, query_window  # noqa: E402


def main():
    k, reps = int(os.environ.get("COPIES", "8")), int(os.environ.get("REPS", "6"))
    ctx = GpuContext(0)
    w = query_window(5)
    stream = NEXMarkSource(1087, 1_000_000, w, seed=20260925).generate_data(ctx, relations=("bid",), bid_columns=("auction",))
    sched = stream.window_schedule("bid", w)
    base = stream.bids.auction
    copies, spacers = [base], []
    for i in range(k - 1):
        spacers.append(torch.empty((i + 1) * 37 * 1024 * 1024 + 4096 * (i + 3), dtype=torch.uint8, device=base.device))  # odd gaps
        copies.append(base.clone())
    torch.cuda.synchronize()
    out = []

    def measure(tag, i):
        b = Bids(auction=copies[i], rows=stream.bids.rows)
        for _ in range(2):
            ctx.q5_hot_items(b, sched)
        times = []
        for _ in range(reps):
            ctx.profile_reset(); ctx.profile_only("q5_count_kernel"); ctx.profile(True)
            ctx.q5_hot_items(b, sched)
            torch.cuda.synchronize()
            st = ctx.profile_read()["q5_count_kernel"]
            times.append(st["total_ms"] / st["launches"])
        ctx.profile(False); ctx.profile_only(None)
        addr = copies[i].data_ptr()
        rec = {"pass": tag, "copy": i, "addr": hex(addr), "addr_mod_4GiB_MiB": round((addr % (1 << 32)) / 2**20, 2),
               "avg_ms": round(sum(times) / len(times), 4), "min_ms": round(min(times), 4), "max_ms": round(max(times), 4)}
        out.append(rec)
        print(json.dumps(rec), flush=True)

    for i in range(k):
        measure("forward", i)
    for i in reversed(range(k)):
        measure("reverse", i)
    # the same copy many times in a row: drift inside one process
    for r in range(4):
        measure(f"again{r}", 0)
    a = [r["avg_ms"] for r in out]
    print(json.dumps({"summary": {"min_avg": min(a), "max_avg": max(a), "spread_pct": round(100 * (max(a) - min(a)) / min(a), 2)}}))


if __name__ == "__main__":
    main()
