#!/usr/bin/env python3
"""Mints tests/golden/next_hashes.json: frozen per-window fingerprints of the two "next" workloads whose oracles had no
second implementation or no frozen answer (YSB and q11 were "parity unpinned"):

  * YSB (benchmarks/src/ysb/ysb.sql, Tumbling(10 s)): every window's {campaign_id: COUNT(*)} computed by the scalar C oracle
    (oracle_ysb_campaign_counts) AND by Arrow C++ (pyarrow compute / Acero: filter, hash join, group_by) -- and, at the small
    sizes, by the literal Python dict walk too.  The reference pins YSB the same way: distributed == local engine
    (flock/src/launcher/aws/mod.rs:681-844).
  * q11 (q11.sql under Window::Session(10 s)): every epoch's closed sessions computed by the whole-column numpy restatement
    AND by the Arrow formulation (oracle.q11_user_sessions_arrow) -- and, at the small sizes, by the literal session walk
    (flock-function/src/aws/window/session.rs, line by line).
A line is written only where all implementations agree.  Run here: `python tools/make_next_goldens.py [--only small]`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa  # noqa: F401  (first import on the main thread)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "next_hashes.json")

# (seed, eps, seconds, campaigns, ads): the last one is what bench.py's ysb_next runs
YSB = [(1, 1000, 30, 100, 10), (9, 20_000, 20, 7, 3), (4, 200_000, 10, 1000, 10), (20260925, 1_000_000, 50, 100, 10)]
# (seed, eps, seconds, timeout): the last one is what bench.py's q11_next runs
Q11 = [(3, 400, 40, 10), (4, 2000, 25, 2), (5, 150, 60, 5), (20260926, 1_000_000, 109, 10)]


def ysb_key(seed, eps, seconds, campaigns, ads):
    return f"ysb/seed={seed}/eps={eps}/seconds={seconds}/campaigns={campaigns}/ads={ads}"


def q11_key(seed, eps, seconds, timeout):
    return f"q11/seed={seed}/eps={eps}/seconds={seconds}/timeout={timeout}"


def mint_ysb(seed, eps, seconds, campaigns, ads, threads=8):
    t0 = time.time()
    c_ad, camp = oracle.ysb_campaigns(seed, campaigns, ads)
    groups = oracle.ysb_campaign_groups(camp)
    n_win = seconds // 10
    small = eps * seconds <= 2_000_000

    def one(w):
        ad, et = oracle.ysb_events(seed, w * 10 * eps, 10 * eps, campaigns * ads)
        a = oracle.ysb_campaign_counts_c(ad, et, c_ad, camp, groups=groups)
        b = oracle.ysb_campaign_counts_arrow(ad, et, c_ad, camp)
        assert a == b, (w, len(a), len(b))
        if small:
            assert a == oracle.ysb_campaign_counts(ad, et, c_ad, camp), w
        return oracle.ysb_fingerprint(a), sum(a.values())
    with ThreadPoolExecutor(min(threads, max(n_win, 1))) as pool:
        res = list(pool.map(one, range(n_win)))
    print(f"{ysb_key(seed, eps, seconds, campaigns, ads)}: {n_win} windows, {time.time() - t0:.1f} s", file=sys.stderr)
    return {"windows": n_win, "joined_events": int(sum(r[1] for r in res)), "fingerprints": [r[0] for r in res]}


def mint_q11(seed, eps, seconds, timeout):
    t0 = time.time()
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    n = eps * seconds
    chunk = 10 * eps
    parts = [s.bids(a, min(n, a + chunk), columns=("bidder", "b_date_time")) for a in range(0, n, chunk)]
    bidder = np.concatenate([p["bidder"] for p in parts])
    ts = np.concatenate([p["b_date_time"] for p in parts])
    off = np.array([s.counts(0, e * eps)[2] for e in range(seconds + 1)], np.int64)
    a = oracle.q11_user_sessions_columnar(bidder, ts, off, timeout, oracle.BASE_TIME)
    b = oracle.q11_user_sessions_arrow(bidder, ts, off, timeout, oracle.BASE_TIME)
    fa, fb = oracle.q11_fingerprints(a), oracle.q11_fingerprints(b)
    assert fa == fb, [t for t in range(seconds) if fa[t] != fb[t]][:5]
    if n <= 200_000:   # the literal walk (a Python loop over every bidder of every epoch)
        walk = oracle.q11_user_sessions(bidder, ts, off, timeout, oracle.BASE_TIME)
        for t, d in enumerate(walk):
            keys = sorted(d)
            fw = oracle.multiset_fingerprint([np.array(keys, np.int32), np.array([d[k][0] for k in keys], np.int64), np.array([d[k][1] for k in keys], np.int64),
                                              np.array([d[k][2] for k in keys], np.int64)]) if keys else "0:0000000000000000"
            assert fw == fa[t], t
    print(f"{q11_key(seed, eps, seconds, timeout)}: {seconds} epochs, {len(a[1])} rows, {time.time() - t0:.1f} s", file=sys.stderr)
    return {"epochs": seconds, "result_rows": int(len(a[1])), "fingerprints": fa}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["small", "all"], default="all")
    args = ap.parse_args()
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for cfg in YSB:
        if args.only == "small" and cfg[1] * cfg[2] > 5_000_000:
            continue
        out[ysb_key(*cfg)] = mint_ysb(*cfg)
    for cfg in Q11:
        if args.only == "small" and cfg[1] * cfg[2] > 5_000_000:
            continue
        out[q11_key(*cfg)] = mint_q11(*cfg)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
