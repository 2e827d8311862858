#!/usr/bin/env python3
"""bench.py -- NEXMark rows/s of the MI355X hot path, one JSON line on rank 0.

Step = one pass of the hot path over one batch of synthetic input already resident in HBM: every
window of the query's schedule (benchmarks/src/nexmark/main.rs:115-123) over `seconds` x `eps`
generated events, executed through the C ABI (include/flockgpu.h).

Workload at N=1 (BASELINE.json configs[3], the largest single-GPU configuration the metric is quoted
on): q5 hot-items over 1.0e9 synthetic bids (1087 s x 1e6 events/s, Hopping(10 s, 5 s) -> 216 windows).
With N ranks every rank owns its own slice of the global event stream (first_event_id = rank * events):
NEXMark windows are independent units, so the path shards with no data-path collective ("weak").
q2 / q3 / q8 (BASELINE.json configs[1], [2], [4]) are reported alongside in "also".

roofline: dominant kernel's ALGORITHMIC bytes (q5 = 4 B per bid with pane
sharing) / its average launch duration measured with HIP events on the launch stream inside the
timed region; peak = 8 TB/s HBM3E (MI355X_MICROARCH.md).  cpu_baseline: the scalar C oracle (a port:
the Rust/DataFusion reference cannot be built here) on a bounded sample of the same windows.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0

# query -> (dominant kernel, algorithmic bytes per input row of that kernel's relation)
DOMINANT = {
    5: ("q5_count_kernel", 4.0, "bid"),        # auction column, each bid read once (pane sharing)
    2: ("q2_flag_kernel", 4.0, "bid"),         # the filter pass proper: auction column once (price / output: q2_emit_kernel)
    3: ("q3_probe_flag_kernel", 8.0, "auction"),    # seller + category per auction row (filter/probe phase)
    8: ("q8_sellers_bitmap_kernel", 4.0, "auction"),  # seller per auction row
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--query", type=int, default=5, choices=[2, 3, 5, 8])
    ap.add_argument("--seconds", type=int, default=0, help="epochs of synthetic events per rank (0 = BASELINE config)")
    ap.add_argument("--eps", type=int, default=1_000_000)
    ap.add_argument("--no-also", action="store_true", help="skip the q2/q3/q8 side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-windows", type=int, default=10)
    return ap.parse_args()


DEFAULT_SECONDS = {5: 1087, 2: 109, 3: 100, 8: 1000}   # 1e9 bids / 1e8 bids / 1e8 events / 1e9 events


def relations_for(q):
    return {2: ("bid",), 5: ("bid",), 3: ("auction", "person"), 8: ("auction", "person")}[q]


def input_rows(q, stream):
    if q in (2, 5):
        return stream.bids.rows
    return stream.auctions.rows + stream.persons.rows


def make_stream(ctx, q, seconds, eps, rank):
    from flock_amd import NEXMarkSource, query_window
    src = NEXMarkSource(seconds, eps, query_window(q), seed=20260925, first_event_id=rank * seconds * eps)
    cols = ("auction", "price") if q == 2 else ("auction",)
    return src.generate_data(ctx, relations=relations_for(q), bid_columns=cols)


def run_steps(ctx, q, stream, steps, warmup, barrier):
    import torch
    from flock_amd import run_query
    res = None
    for _ in range(warmup):
        res = run_query(ctx, q, stream)
    ctx.profile_reset()
    ctx.profile(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = run_query(ctx, q, stream)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    stats = ctx.profile_read()
    ctx.profile(False)
    return dt, stats, res


def roofline(q, stats, stream, res):
    name, bpr, rel = DOMINANT[q]
    st = stats.get(name)
    if not st or not st["launches"]:
        return None
    rows = {"bid": lambda: stream.bids.rows, "auction": lambda: stream.auctions.rows}[rel]()
    alg_bytes = bpr * rows
    avg_ms = st["total_ms"] / st["launches"]
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic = None
    prof = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch, measured separately
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get(name)
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
            "algorithmic_bytes_per_launch": int(alg_bytes), "launches": st["launches"],
            "kernels_ms": {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in stats.items()}}


def cpu_baseline(q, stream, n_windows):
    """Scalar C oracle (kind = "port") on the first windows of the same workload, host cores of this box."""
    import numpy as np
    import oracle
    from flock_amd import query_window
    w = query_window(q)
    if q == 5:
        sched = stream.window_schedule("bid", w)
        n_windows = min(n_windows, sched.n_windows)
        lo0, _ = sched.window_rows(0)
        _, hi1 = sched.window_rows(n_windows - 1)
        host = stream.bids.auction[lo0:hi1].cpu().numpy()
        t0 = time.perf_counter()
        for i in range(n_windows):
            lo, hi = sched.window_rows(i)
            oracle.q5_hot_items(host[lo - lo0:hi - lo0])
        dt = time.perf_counter() - t0
        rows = hi1 - lo0
        sample = f"first {n_windows} Hopping(10,5) windows = {rows} bids (each bid counted once)"
    else:
        return None
    return {"value": round(rows / dt, 1), "unit": "rows/s", "cores": 1, "kind": "port", "sample": sample,
            "seconds": round(dt, 2), "host_cores_available": os.cpu_count()}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: flock_amd has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        tok = torch.zeros(1, device=f"cuda:{local}")

        def barrier():
            dist.all_reduce(tok)
            torch.cuda.synchronize()
    else:
        def barrier():
            pass

    from flock_amd import GpuContext
    ctx = GpuContext(local)
    q = args.query
    seconds = args.seconds or DEFAULT_SECONDS[q]
    stream = make_stream(ctx, q, seconds, args.eps, rank)
    dt, stats, res = run_steps(ctx, q, stream, args.steps, args.warmup, barrier)
    rows = input_rows(q, stream)
    if world > 1:
        t = torch.tensor([dt, float(rows)], dtype=torch.float64, device=f"cuda:{local}")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt_max, rows_all = float(tmax[0]), float(t[1])
    else:
        dt_max, rows_all = dt, float(rows)

    out = None
    if rank == 0:
        from flock_amd import query_window
        w = query_window(q)
        out = {
            "metric": "NEXMark rows/sec per node (q3 join, q5 agg)", "value": round(rows_all * args.steps / dt_max, 1),
            "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"NEXMark q{q} {w.kind}({w.size},{w.hop}) over {seconds} s x {args.eps} events/s per GPU",
                       "query": f"q{q}", "input_rows_per_gpu": int(rows), "windows_per_gpu": res.n_windows,
                       "parallelism": f"window-sharded x{world} (no data-path collective)",
                       "result_rows": int(res.rows)},
            "roofline": roofline(q, stats, stream, res),
        }
    if rank == 0:
        out["cpu_baseline"] = cpu_baseline(q, stream, args.cpu_windows) if (world == 1 and not args.no_cpu) else None
    # side measurements (N = 1 only): the other BASELINE configs, each with its own roofline
    if rank == 0 and world == 1 and not args.no_also:
        also = {}
        del stream, res
        torch.cuda.empty_cache()
        for q2 in (2, 3, 8, 5):
            if q2 == q:
                continue
            try:
                s2 = make_stream(ctx, q2, DEFAULT_SECONDS[q2], args.eps, 0)
                d2, st2, r2 = run_steps(ctx, q2, s2, max(2, min(args.steps, 3)), 1, lambda: None)
                steps2 = max(2, min(args.steps, 3))
                also[f"q{q2}"] = {"value": round(input_rows(q2, s2) * steps2 / d2, 1), "unit": "rows/s",
                                  "ms_per_step": round(d2 / steps2 * 1e3, 3), "input_rows": int(input_rows(q2, s2)),
                                  "windows": r2.n_windows, "result_rows": int(r2.rows),
                                  "seconds_of_events": DEFAULT_SECONDS[q2], "roofline": roofline(q2, st2, s2, r2)}
                del s2, r2
                torch.cuda.empty_cache()
            except Exception as e:  # a side measurement must never hide the headline
                also[f"q{q2}"] = {"error": str(e)}
        out["also"] = also
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
