#!/usr/bin/env python3
"""Mints tests/golden/nexmark_hashes.json: the row-multiset fingerprint of every window's result for
(query, seed, eps, seconds), frozen only where TWO independent CPU implementations agree
(the reference's NEXMark tests only print, so it pins nothing itself):

  * oracle/nexmark_ops.c (the scalar C restatement the HIP kernels are compared with), and
  * pyarrow compute / acero (Arrow C++: filter, group_by count_all, hash join) on the same generated columns.

Run here (CPU container): `python tools/make_nexmark_goldens.py [--only small]`.  The BASELINE-size entries take a
few minutes and ~10 GB of RAM.  The tests never regenerate this file: a simultaneous drift of generator + oracle +
kernel shows up as a mismatch against it.

Fingerprint = oracle.multiset_fingerprint(columns of the query's OUTPUT SCHEMA, qN_plan.fmt:1), one per window,
windows as flock-function/src/aws/window/{elementwise,tumbling,hopping}.rs cut them.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa            # imported on the main thread: first import on a worker thread crashed later pools
import pyarrow.compute as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "nexmark_hashes.json")

SMALL = [(1, 1000, 3), (7, 5000, 4), (42, 50_000, 12), (5, 1_000_000, 2)]
# (seed, eps, seconds): BASELINE.json configs as bench.py runs them (seed 20260925)
CONFIGS = {
    1: SMALL,
    2: SMALL + [(20260925, 1_000_000, 109)],
    3: SMALL + [(20260925, 1_000_000, 100), (20260925, 1_000_000, 1000)],
    5: [(1, 1000, 23), (7, 5000, 30), (42, 50_000, 20), (20260925, 1_000_000, 1087)],
    8: [(1, 1000, 23), (7, 5000, 30), (42, 50_000, 20), (20260925, 1_000_000, 1000)],
    # SURVEY.md section 8(f) "next" queries at the sizes bench.py runs them (q7: 1e9 bids; q4 / q9: 300 s = 2.76e8 bids)
    7: [(1, 1000, 23), (7, 5000, 30), (42, 50_000, 20), (20260925, 1_000_000, 1087)],
    9: [(1, 1000, 3), (7, 5000, 4), (42, 50_000, 6), (20260925, 1_000_000, 300)],
    4: [(1, 1000, 3), (7, 5000, 4), (42, 50_000, 6), (20260925, 1_000_000, 300)],
    13: [(1, 1000, 3), (7, 5000, 4), (42, 50_000, 12), (20260925, 1_000_000, 1087)],
}
WINDOW = {1: ("elementwise",), 2: ("elementwise",), 3: ("elementwise",), 5: ("hopping", 10, 5), 8: ("tumbling", 10), 7: ("tumbling", 10),
          9: ("elementwise",), 4: ("elementwise",), 13: ("elementwise",)}


def key(q, seed, eps, seconds):
    return f"q{q}/seed={seed}/eps={eps}/seconds={seconds}"


def windows_of(q, seconds):
    w = WINDOW[q]
    if w[0] == "elementwise":
        return oracle.elementwise_windows(seconds)
    if w[0] == "tumbling":
        return oracle.tumbling_windows(seconds, w[1])
    return oracle.hopping_windows(seconds, w[1], w[2])


def generate(seed, eps, seconds, relations, threads):
    """The whole stream's columns + per-epoch row offsets per relation, generated in parallel epoch chunks."""
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    chunk = max(1, (seconds + threads * 4 - 1) // (threads * 4))
    spans = [(e, min(seconds, e + chunk)) for e in range(0, seconds, chunk)]

    def one(span):
        n0, n1 = span[0] * eps, span[1] * eps
        out = {}
        if "bid" in relations:
            out["bid"] = s.bids(n0, n1, columns=relations["bid"])
        if "auction" in relations:
            a = s.auctions(n0, n1)
            out["auction"] = {k: a[k] for k in relations["auction"]}
        if "person" in relations:
            p = s.persons(n0, n1)
            out["person"] = {k: p[k] for k in relations["person"]}
        return out

    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(one, spans))
    cols = {}
    for rel in relations:
        cols[rel] = {}
        for k in relations[rel]:
            vs = [p[rel][k] for p in parts]
            if isinstance(vs[0], oracle.Utf8):
                lens = np.concatenate([np.diff(v.offsets) for v in vs])
                off = np.zeros(len(lens) + 1, np.int64)
                np.cumsum(lens, out=off[1:])
                assert off[-1] < 2**31
                cols[rel][k] = oracle.Utf8(off.astype(np.int32), np.concatenate([v.data for v in vs]))
            else:
                cols[rel][k] = np.concatenate(vs)
    offs = {}
    for rel in relations:
        idx = {"person": 0, "auction": 1, "bid": 2}[rel]
        per_epoch = np.array([s.counts(e * eps, (e + 1) * eps)[idx] for e in range(seconds)], np.int64) if seconds <= 4096 else None
        offs[rel] = np.concatenate(([0], np.cumsum(per_epoch)))
    return cols, offs


# ---------------------------------------------------------------- per-window results, twice
def _pa():
    return pa, pc


def q1_window(c, lo, hi):
    pa, pc = _pa()
    b = {k: v[lo:hi] for k, v in c["bid"].items()}
    price = oracle.q1_project(b["price"])
    alt = pc.multiply(pa.scalar(0.908, pa.float64()), pc.cast(pa.array(b["price"]), pa.float64())).to_numpy()
    assert price.tobytes() == alt.tobytes()
    return oracle.multiset_fingerprint([b["auction"], b["bidder"], price, b["b_date_time"]])


def q2_window(c, lo, hi):
    pa, pc = _pa()
    a, p = c["bid"]["auction"][lo:hi], c["bid"]["price"][lo:hi]
    oa, op = oracle.q2_filter(a, p)
    a64 = pc.cast(pa.array(a), pa.int64())
    mask = pc.equal(pc.subtract(a64, pc.multiply(pc.divide(a64, 123), 123)), 0)      # truncated remainder
    assert np.array_equal(pc.filter(pa.array(a), mask).to_numpy(), oa) and np.array_equal(pc.filter(pa.array(p), mask).to_numpy(), op)
    return oracle.multiset_fingerprint([oa, op])


def q3_window(c, alo, ahi, plo, phi):
    pa, pc = _pa()
    au = {k: v[alo:ahi] for k, v in c["auction"].items()}
    pe = {k: (v.slice(plo, phi) if isinstance(v, oracle.Utf8) else v[plo:phi]) for k, v in c["person"].items()}
    ar, pr = oracle.q3_join(au["seller"], au["category"], pe["p_id"], pe["state"])
    fp = oracle.multiset_fingerprint([(pe["name"], pr), (pe["city"], pr), (pe["state"], pr), (au["a_id"], ar)])
    ta = pa.table({k: au[k] for k in ("a_id", "seller", "category")}).filter(pc.equal(pc.field("category"), 10))
    tp = pa.table({"p_id": pe["p_id"], "prow": np.arange(len(pe["p_id"])),
                   "state": pa.Array.from_buffers(pa.utf8(), len(pe["state"]), [None, pa.py_buffer(pe["state"].offsets), pa.py_buffer(pe["state"].data)])})
    tp = tp.filter(pc.is_in(pc.field("state"), pa.array(["or", "id", "ca"])))
    j = ta.join(tp, keys="seller", right_keys="p_id", join_type="inner")
    prow = j["prow"].to_numpy()
    alt = oracle.multiset_fingerprint([(pe["name"], prow), (pe["city"], prow), (pe["state"], prow), j["a_id"].to_numpy()])
    assert fp == alt, (fp, alt)
    return fp


def q5_window(c, lo, hi):
    pa, pc = _pa()
    a = c["bid"]["auction"][lo:hi]
    oa, on = oracle.q5_hot_items(a)
    t = pa.table({"auction": a}).group_by("auction", use_threads=False).aggregate([([], "count_all")])
    if len(a):
        mx = pc.max(t["count_all"]).as_py()
        t = t.filter(pc.equal(pc.field("count_all"), mx))
    alt = sorted(zip(t["auction"].to_pylist(), t["count_all"].to_pylist()))
    assert alt == sorted(zip(oa.tolist(), on.tolist()))
    return oracle.multiset_fingerprint([oa, on.astype(np.int64)])


def q8_window(c, alo, ahi, plo, phi):
    pa, pc = _pa()
    seller = c["auction"]["seller"][alo:ahi]
    p_id, name = c["person"]["p_id"][plo:phi], c["person"]["name"].slice(plo, phi)
    rows = oracle.q8_join(p_id, name, seller)
    fp = oracle.multiset_fingerprint([(p_id, rows), (name, rows)])
    nm = pa.Array.from_buffers(pa.utf8(), len(name), [None, pa.py_buffer(name.offsets), pa.py_buffer(name.data)])
    tp = pa.table({"p_id": p_id, "name": nm}).group_by(["p_id", "name"], use_threads=False).aggregate([])
    ts = pa.table({"seller": seller}).group_by("seller", use_threads=False).aggregate([])
    j = tp.join(ts, keys="p_id", right_keys="seller", join_type="inner")
    jn = j["name"].combine_chunks()
    if len(jn):
        b = jn.buffers()
        off = np.frombuffer(b[1], np.int32, len(jn) + 1, jn.offset * 4)
        jname = oracle.Utf8((off - off[0]).astype(np.int32), np.frombuffer(b[2], np.uint8)[off[0]:off[-1]])
    else:
        jname = oracle.Utf8(np.zeros(1, np.int32), np.zeros(0, np.uint8))
    alt = oracle.multiset_fingerprint([j["p_id"].to_numpy(), jname])
    assert fp == alt, (fp, alt)
    return fp


def q7_window(c, lo, hi):
    pa, pc = _pa()
    b = {k: v[lo:hi] for k, v in c["bid"].items()}
    rows = oracle.q7_highest_bid(b["price"])
    if hi > lo:
        mask = pc.equal(pa.array(b["price"]), pc.max(pa.array(b["price"])))
        assert np.array_equal(np.flatnonzero(mask.to_numpy(zero_copy_only=False)), rows)
    return oracle.multiset_fingerprint([(b["auction"], rows), (b["price"], rows), (b["bidder"], rows), (b["b_date_time"], rows)])   # q7.sql's column order


def side_input_of(seed, eps, seconds, stride=6007):
    """The bounded side input flock_amd.nexmark.synthetic_side_input builds for a generated stream (the reference reads q13's table from
    a user-supplied CSV in S3, benchmarks/src/nexmark/main.rs:44,353-361): every `stride`-th auction id the bids can name, value = 7 key + 1."""
    n_auctions = oracle.NexmarkStream(seed=seed, eps=eps).counts(0, seconds * eps)[1]
    first = 1000
    key = np.arange(first - first % stride + stride, first + n_auctions + stride, stride, dtype=np.int32)
    return key, (key * 7 + 1).astype(np.int32)


def q13_window(c, lo, hi):
    pa, pc = _pa()
    b = {k: v[lo:hi] for k, v in c["bid"].items()}
    key, value = c["side"]
    br, sr = oracle.q13_side_join(b["auction"], key)
    fp = oracle.multiset_fingerprint([(b["auction"], br), (b["bidder"], br), (b["price"], br), (b["b_date_time"], br), (value, sr)])
    tb = pa.table({"auction": b["auction"], "row": np.arange(hi - lo)})
    j = tb.join(pa.table({"key": key, "value": value}), keys="auction", right_keys="key", join_type="inner")
    rows = j["row"].to_numpy()
    alt = oracle.multiset_fingerprint([(b["auction"], rows), (b["bidder"], rows), (b["price"], rows), (b["b_date_time"], rows), j["value"].to_numpy()])
    assert fp == alt, (fp, alt)
    return fp


def _winning_bids_pa(a, b):
    """Q of q4.sql / q9.sql through pyarrow: join, BETWEEN, MAX GROUP BY (a_id, category)."""
    pa, pc = _pa()
    ta = pa.table({"a_id": a["a_id"], "category": a["category"], "a_date_time": a["a_date_time"], "expires": a["expires"]})
    tb = pa.table({"auction": b["auction"], "price": b["price"], "b_date_time": b["b_date_time"], "row": np.arange(len(b["price"]))})
    j = ta.join(tb, keys="a_id", right_keys="auction", join_type="inner")
    j = j.filter(pc.and_(pc.greater_equal(j["b_date_time"], j["a_date_time"]), pc.less_equal(j["b_date_time"], j["expires"])))
    return tb, j.group_by(["a_id", "category"], use_threads=False).aggregate([("price", "max")])


def q9_window(c, alo, ahi, blo, bhi):
    a = {k: v[alo:ahi] for k, v in c["auction"].items()}
    b = {k: v[blo:bhi] for k, v in c["bid"].items()}
    rows = oracle.q9_winning_bids(a["a_id"], a["a_date_time"], a["expires"], b["auction"], b["price"], b["b_date_time"])
    tb, q = _winning_bids_pa(a, b)
    back = tb.join(q.select(["a_id", "price_max"]), keys=["auction", "price"], right_keys=["a_id", "price_max"], join_type="inner")
    assert sorted(back["row"].to_pylist()) == rows.tolist()
    return oracle.multiset_fingerprint([(b["auction"], rows), (b["bidder"], rows), (b["price"], rows), (b["b_date_time"], rows)])


def q4_window(c, alo, ahi, blo, bhi):
    a = {k: v[alo:ahi] for k, v in c["auction"].items()}
    b = {k: v[blo:bhi] for k, v in c["bid"].items()}
    cats, avg = oracle.q4_avg_final_by_category(a["a_id"], a["category"], a["a_date_time"], a["expires"], b["auction"], b["price"], b["b_date_time"])
    _, q = _winning_bids_pa(a, b)
    q4 = q.group_by("category", use_threads=False).aggregate([("price_max", "mean")]).sort_by("category")
    assert cats.tolist() == q4["category"].to_pylist() and avg.tobytes() == q4["price_max_mean"].to_numpy().astype(np.float64).tobytes()
    return oracle.multiset_fingerprint([cats, avg])   # (the Float64 column by its bits)


def mint(q, seed, eps, seconds, threads):
    rel = {1: {"bid": ("auction", "bidder", "price", "b_date_time")}, 2: {"bid": ("auction", "price")},
           3: {"auction": ("a_id", "seller", "category"), "person": ("p_id", "name", "city", "state")},
           5: {"bid": ("auction",)}, 8: {"auction": ("seller",), "person": ("p_id", "name")},
           7: {"bid": ("auction", "bidder", "price", "b_date_time")}, 13: {"bid": ("auction", "bidder", "price", "b_date_time")},
           9: {"auction": ("a_id", "category", "a_date_time", "expires"), "bid": ("auction", "bidder", "price", "b_date_time")},
           4: {"auction": ("a_id", "category", "a_date_time", "expires"), "bid": ("auction", "bidder", "price", "b_date_time")}}[q]
    t0 = time.time()
    c, offs = generate(seed, eps, seconds, rel, threads)
    if q == 13:
        c["side"] = side_input_of(seed, eps, seconds)
    wins = windows_of(q, seconds)

    def one(w):
        e0, e1 = w
        if q in (1, 2, 5, 7, 13):
            lo, hi = int(offs["bid"][e0]), int(offs["bid"][e1])
            return {1: q1_window, 2: q2_window, 5: q5_window, 7: q7_window, 13: q13_window}[q](c, lo, hi)
        if q in (4, 9):
            return (q4_window if q == 4 else q9_window)(c, int(offs["auction"][e0]), int(offs["auction"][e1]), int(offs["bid"][e0]), int(offs["bid"][e1]))
        span = (int(offs["auction"][e0]), int(offs["auction"][e1]), int(offs["person"][e0]), int(offs["person"][e1]))
        return (q3_window if q == 3 else q8_window)(c, *span)

    with ThreadPoolExecutor(threads) as pool:
        fps = list(pool.map(one, wins))
    rows = sum(int(f.split(":")[0]) for f in fps)
    print(f"{key(q, seed, eps, seconds)}: {len(wins)} windows, {rows} result rows, {time.time() - t0:.1f} s", file=sys.stderr)
    return {"windows": len(wins), "result_rows": rows, "fingerprints": fps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["small", "all"], default="all")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--query", type=int, default=0)
    args = ap.parse_args()
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for q, cfgs in CONFIGS.items():
        if args.query and q != args.query:
            continue
        for seed, eps, seconds in cfgs:
            if args.only == "small" and eps * seconds > 5_000_000:
                continue
            out[key(q, seed, eps, seconds)] = mint(q, seed, eps, seconds, args.threads)
            with open(OUT, "w") as f:
                json.dump(out, f, indent=0, sort_keys=True)
                f.write("\n")


if __name__ == "__main__":
    main()
