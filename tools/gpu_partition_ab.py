"""gpurun helper: `flockgpu_partition_by_key` (RepartitionExec Hash([key], n), shuffle.hip) at 1 .. 64 destinations on ONE GPU -- the stage-0 side of the
exchange as 2 / 4 / 8 ranks would run it.  NEXMark-shaped input: 8e7 (auction, count) pairs of 216 hopping windows' Partial groups stand in as
8e7 keys over 216 windows; with --payload the two 4-byte columns of q5's exchange ride in the emit pass (what comm.hip asks for).
Prints per destination count: ms per call (host clock around 10 calls) and the LaunchScope averages of the count / emit kernels.

    python tools/gpu_partition_ab.py [--rows 80000000] [--windows 216] [--parts 1,2,4,8,16,64]
"""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=80_000_000)
    ap.add_argument("--windows", type=int, default=216)
    ap.add_argument("--parts", default="1,2,4,8,16,64")
    ap.add_argument("--calls", type=int, default=10)
    ap.add_argument("--check", action="store_true", help="compare every destination count with the numpy restatement (first 2e6 rows)")
    a = ap.parse_args()
    import torch
    from flock_amd import GpuContext, WindowSchedule

    ctx = GpuContext(0)
    g = torch.Generator(device="cuda").manual_seed(7)
    keys = torch.randint(0, 2**31 - 1, (a.rows,), dtype=torch.int32, device="cuda", generator=g)
    offs = np.linspace(0, a.rows, a.windows + 1).astype(np.int64)
    offs[1:-1] += 3                                                  # unaligned window starts
    sched = WindowSchedule(offs, np.arange(a.windows), np.arange(1, a.windows + 1))
    out = {}
    for n_parts in [int(x) for x in a.parts.split(",")]:
        for _ in range(3):
            ctx.partition_by_key_raw(keys, sched, n_parts)
        ctx.profile(True)
        ctx.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.calls):
            ctx.partition_by_key_raw(keys, sched, n_parts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.calls
        prof = ctx.profile_read()
        ctx.profile(False)
        k = {name: round(v["total_ms"] / max(v["launches"], 1), 4) for name, v in prof.items()}
        # algorithmic bytes: the count pass reads the keys and leaves a destination byte per row (n > 1), the emit pass reads that byte and writes the row number
        cb, eb = (0, a.rows * 4) if n_parts == 1 else (a.rows * 5, a.rows * 5)
        frac = lambda nbytes, name: round(nbytes / (max(k.get(name, 0.0), 1e-9) * 1e-3) / 8e12, 4)
        out[n_parts] = {"ms_per_call": round(ms, 4), "kernels_ms": k, "count_frac_of_8TBs": frac(cb, "partition_count_kernel"), "emit_frac_of_8TBs": frac(eb, "partition_emit_kernel")}
        print(n_parts, json.dumps(out[n_parts]), flush=True)
    if a.check:
        sys.path.insert(0, "tests")
        from test_distributed import NumpyOps
        n = min(a.rows, 2_000_000)
        offs2 = np.array([0, 5, 5, 8200, 8200 + 8192 * 3, n // 2 + 1, n])
        s2 = WindowSchedule(offs2, np.array([0, 1, 2, 3, 5]), np.array([1, 2, 3, 5, 6]))
        kh = keys[:n].cpu()
        for n_parts in [int(x) for x in a.parts.split(",")]:
            rows, counts = ctx.partition_by_key(keys[:n].clone(), s2, n_parts)
            wr, wc = NumpyOps().partition(kh, s2, n_parts)
            ok = np.array_equal(counts, wc) and np.array_equal(rows.cpu().numpy(), wr.numpy())
            print("check", n_parts, "ok" if ok else "MISMATCH", flush=True)
            if not ok:
                sys.exit(1)
    ctx.close()


if __name__ == "__main__":
    main()
