"""tests/golden/nexmark_hashes.json (minted by tools/make_nexmark_goldens.py from oracle == pyarrow agreement,
SURVEY.md section 8c) against (CPU) the oracle re-run on the small configurations and (GPU, -m gpu) the HIP path on EVERY
window of EVERY configuration, BASELINE.json's full sizes included: q2 1e8 bids (109 windows), q3 1e8 / 1e9 events
(100 / 1000 windows), q5 1e9 bids (216 windows), q8 1e9 events (100 windows); the "next" queries q7 (1e9 bids, 108 windows) and
q4 / q9 (300 s = 2.76e8 bids, 300 windows each) at the sizes bench.py runs them."""
import json
import os
import sys

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "nexmark_hashes.json")))


def _parse(k):
    q, seed, eps, seconds = k.split("/")
    return int(q[1:]), int(seed.split("=")[1]), int(eps.split("=")[1]), int(seconds.split("=")[1])


SMALL = [k for k in sorted(GOLDEN) if _parse(k)[2] * _parse(k)[3] <= 5_000_000]
ALL = sorted(GOLDEN, key=lambda k: (_parse(k)[2] * _parse(k)[3], k))


def test_golden_file_covers_the_baseline_configs():
    for k, windows in (("q2/seed=20260925/eps=1000000/seconds=109", 109), ("q3/seed=20260925/eps=1000000/seconds=100", 100),
                       ("q3/seed=20260925/eps=1000000/seconds=1000", 1000), ("q5/seed=20260925/eps=1000000/seconds=1087", 216),
                       ("q8/seed=20260925/eps=1000000/seconds=1000", 100), ("q7/seed=20260925/eps=1000000/seconds=1087", 108),
                       ("q9/seed=20260925/eps=1000000/seconds=300", 300), ("q4/seed=20260925/eps=1000000/seconds=300", 300),
                       ("q13/seed=20260925/eps=1000000/seconds=1087", 1087)):
        assert GOLDEN[k]["windows"] == windows == len(GOLDEN[k]["fingerprints"]), k
        assert GOLDEN[k]["result_rows"] > 0


@pytest.mark.parametrize("k", SMALL)
def test_oracle_reproduces_the_small_goldens(k):
    """Re-mints the small entries in memory (oracle AND pyarrow, asserted equal inside) and compares with the frozen file."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_nexmark_goldens as m
    q, seed, eps, seconds = _parse(k)
    got = m.mint(q, seed, eps, seconds, threads=4)
    assert got == GOLDEN[k]


def test_fingerprint_is_order_free_and_column_order_sensitive():
    u = oracle.Utf8(np.array([0, 2, 4, 6], np.int32), np.frombuffer(b"orcaid", np.uint8).copy())
    a = np.array([1, 2, 3], np.int32)
    perm = np.array([2, 0, 1])
    assert oracle.multiset_fingerprint([u, a]) == oracle.multiset_fingerprint([(u, perm), a[perm]])
    assert oracle.multiset_fingerprint([u, a]) != oracle.multiset_fingerprint([u, a[perm]])
    assert oracle.multiset_fingerprint([a, a[perm]]) != oracle.multiset_fingerprint([a[perm], a])
    assert oracle.multiset_fingerprint([a[:0]]) == "0:0000000000000000"


# ------------------------------------------------------------------ GPU: the HIP path against every frozen window
def _segment_fingerprints(row_hash, off):
    out = []
    with np.errstate(over="ignore"):
        cs = np.concatenate(([np.uint64(0)], np.cumsum(row_hash, dtype=np.uint64)))
    for w in range(len(off) - 1):
        lo, hi = int(off[w]), int(off[w + 1])
        out.append(f"{hi - lo}:{(int(cs[hi]) - int(cs[lo])) & 0xFFFFFFFFFFFFFFFF:016x}")
    return out


def _u(pair, n):
    off, data = pair
    return oracle.Utf8(off[: n + 1], data)


def hip_fingerprints(ctx, q, seed, eps, seconds):
    """Per-window fingerprints of the HIP path's OUTPUT COLUMNS (C ABI -> host copies), hashed with the checker's hash."""
    import torch
    from flock_amd import NEXMarkSource, query_window, run_query
    all4 = ("auction", "bidder", "price", "b_date_time")
    rel = {1: ("bid",), 2: ("bid",), 5: ("bid",), 7: ("bid",), 13: ("bid",), 3: ("auction", "person"), 8: ("auction", "person"), 9: ("bid", "auction"), 4: ("bid", "auction")}[q]
    cols = {1: all4, 2: ("auction", "price"), 5: ("auction",), 7: all4, 13: all4, 9: all4, 4: ("auction", "price", "b_date_time")}.get(q, ("auction",))
    g = NEXMarkSource(seconds, eps, query_window(q), seed=seed).generate_data(ctx, relations=rel, bid_columns=cols, auction_times=q in (4, 9))
    out = run_query(ctx, q, g)
    if q == 1:
        off = g.window_schedule("bid").pane_row_offsets
        b = g.bids
        h = oracle.row_hashes([b.auction.cpu().numpy(), b.bidder.cpu().numpy(), out.cpu().numpy(), b.b_date_time.cpu().numpy()])
    elif q == 2:
        a, p, off = out.to_host()
        h = oracle.row_hashes([a, p])
    elif q == 5:
        a, n, off = out.to_host()
        assert n.dtype == np.uint64                          # q5_plan.fmt:1 `num: UInt64`
        h = oracle.row_hashes([a, n.astype(np.int64)])
    elif q == 7:
        o = out.to_host()
        off = o["offsets"]
        h = oracle.row_hashes([o["auction"], o["price"], o["bidder"], o["b_date_time"]]) if len(o["price"]) else np.zeros(0, np.uint64)
    elif q == 13:
        o = out.to_host()
        off = o["offsets"]
        h = oracle.row_hashes([o["auction"], o["bidder"], o["price"], o["b_date_time"], o["value"]]) if len(o["price"]) else np.zeros(0, np.uint64)
    elif q == 9:
        o = out.to_host()
        off = o["offsets"]
        h = oracle.row_hashes([o["auction"], o["bidder"], o["price"], o["b_date_time"]]) if len(o["price"]) else np.zeros(0, np.uint64)
    elif q == 4:
        o = out.to_host()
        off = o["offsets"]
        h = oracle.row_hashes([o["category"], o["avg"]]) if len(o["avg"]) else np.zeros(0, np.uint64)
    elif q == 3:
        o = out.to_host()
        off, n = o["offsets"], len(o["a_id"])
        h = oracle.row_hashes([_u(o["name"], n), _u(o["city"], n), _u(o["state"], n), o["a_id"]]) if n else np.zeros(0, np.uint64)
    else:
        o = out.to_host()
        off, n = o["offsets"], len(o["p_id"])
        h = oracle.row_hashes([o["p_id"], _u(o["name"], n)]) if n else np.zeros(0, np.uint64)
    del g, out
    torch.cuda.empty_cache()
    return _segment_fingerprints(h if h is not None else np.zeros(0, np.uint64), off)


@pytest.fixture(scope="module")
def ctx():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", ALL)
def test_hip_path_reproduces_every_frozen_window(ctx, k):
    q, seed, eps, seconds = _parse(k)
    got = hip_fingerprints(ctx, q, seed, eps, seconds)
    want = GOLDEN[k]["fingerprints"]
    assert len(got) == len(want) == GOLDEN[k]["windows"]
    bad = [w for w in range(len(want)) if got[w] != want[w]]
    assert not bad, f"{k}: {len(bad)} windows differ, first {bad[:5]}: {[(got[w], want[w]) for w in bad[:3]]}"


BASELINE_SHAPES = [(2, 1_000_000, 109), (3, 1_000_000, 100), (3, 1_000_000, 1000), (5, 1_000_000, 1087), (8, 1_000_000, 1000)]


@pytest.mark.gpu
@pytest.mark.parametrize("q,eps,seconds", BASELINE_SHAPES)
def test_hip_path_equals_the_oracle_on_every_window_of_an_unfrozen_seed(ctx, q, eps, seconds):
    """BASELINE.json's full sizes on a seed the golden file does not hold: the host generates the stream with the C oracle's
    generator, runs the threaded C oracle (and pyarrow beside it) on EVERY window, and the HIP path -- fed by the device
    generator -- must reproduce every window's row multiset.  Covers the device generator at full size as well."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_nexmark_goldens as m
    seed = 77
    want = m.mint(q, seed, eps, seconds, threads=min(32, os.cpu_count() or 8))
    got = hip_fingerprints(ctx, q, seed, eps, seconds)
    assert len(got) == want["windows"]
    bad = [w for w in range(len(got)) if got[w] != want["fingerprints"][w]]
    assert not bad, f"q{q}: {len(bad)} of {len(got)} windows differ, first {bad[:5]}"
