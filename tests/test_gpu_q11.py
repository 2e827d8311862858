"""q11 (user sessions) on the GPU vs the literal session walk of the oracle, and the stable
grouping of rows by key underneath it vs numpy's stable sort."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
BASE = 1_436_918_400_000


@pytest.fixture(scope="module")
def ctx():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _dev(a):
    from devmem import dev
    return dev(a)          # (a torch tensor; guarded memory under FLOCK_TEST_GUARDED=1: tests/test_gpu_guard.py)


@pytest.mark.parametrize("n,lo,hi,seed", [(1, 5, 6, 0), (63, -3, 4, 1), (4096, 0, 1, 2), (4097, 1000, 1256, 3), (100_003, -70_000, 70_000, 4),
                                          (1_000_000, 1000, 21_000, 5), (300_000, -2**31, 2**31 - 1, 6), (50_000, 0, 2**24, 7)])
def test_group_rows_by_key_is_a_stable_sort(ctx, n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    keys = rng.integers(lo, hi, n, dtype=np.int64, endpoint=(hi == 2**31 - 1)).astype(np.int32)
    if n > 1000:   # hot keys: long runs of equal digits inside a wave
        keys[rng.integers(0, n, n // 3)] = keys[0]
    k, rows = ctx.group_rows_by_key(_dev(keys))
    want = np.argsort(keys, kind="stable")
    assert np.array_equal(rows, want.astype(np.int32))
    assert np.array_equal(k, keys[want])


def _as_dicts(out):
    off = out["offsets"]
    return [{int(out["bidder"][i]): (int(out["bid_count"][i]), int(out["start_time"][i]), int(out["end_time"][i]))
             for i in range(off[t], off[t + 1])} for t in range(len(off) - 1)]


def _bids(ctx, bidder, ts):
    from flock_amd import Bids
    return Bids(None, _dev(bidder), None, _dev(ts), len(bidder))


@pytest.mark.parametrize("seed,n_epochs,per_epoch,n_bidders,timeout,jitter", [
    (0, 12, 40, 9, 2, 0), (1, 30, 250, 400, 3, 0), (2, 20, 600, 150, 1, 2500), (3, 25, 100, 300, 4, 9000), (5, 40, 150, 600, 10, 500),
    (6, 15, 3000, 2500, 2, 0), (7, 6, 20_000, 30_000, 1, 1500)])
def test_q11_sessions_match_the_walk(ctx, seed, n_epochs, per_epoch, n_bidders, timeout, jitter):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, per_epoch + 1, n_epochs)
    counts[rng.integers(0, n_epochs)] = 0     # an epoch without bids still runs the time-out check
    off = np.r_[0, np.cumsum(counts)].astype(np.int64)
    ep = np.repeat(np.arange(n_epochs), counts)
    ms = np.concatenate([np.sort(rng.integers(0, 1000, c)) for c in counts]) if off[-1] else np.zeros(0, np.int64)
    late = rng.integers(0, jitter + 1, off[-1]) if jitter else 0
    ts = BASE + np.maximum(0, ep * 1000 + ms - late).astype(np.int64)
    bidder = rng.integers(1000, 1000 + n_bidders, off[-1]).astype(np.int32)
    bidder[rng.integers(0, max(off[-1], 1), off[-1] // 4)] = 1000   # a hot bidder
    want = oracle.q11_user_sessions(bidder, ts, off, timeout, BASE)
    res = ctx.q11_user_sessions(_bids(ctx, bidder, ts), off, timeout, BASE)
    out = res.to_host()
    assert _as_dicts(out) == want
    assert res.rows == sum(len(d) for d in want) and res.sessions_total >= res.rows
    for t in range(n_epochs):   # rows of an epoch are ordered by bidder
        assert np.all(np.diff(out["bidder"][out["offsets"][t]:out["offsets"][t + 1]]) > 0)


def test_q11_edge_cases(ctx):
    from flock_amd import FlockGpuError, _ffi
    # hand-worked cases of tests/test_oracle_q11.py
    rows = [(0, 7, 100), (0, 8, 200), (0, 9, 300), (0, 7, 900), (1, 7, 1500), (3, 9, 3500), (5, 8, 5100), (6, 1, 6000)]
    bidder = np.array([r[1] for r in rows], np.int32)
    ts = np.array([BASE + r[2] for r in rows], np.int64)
    off = np.searchsorted(np.array([r[0] for r in rows]), np.arange(8)).astype(np.int64)
    got = _as_dicts(ctx.q11_user_sessions(_bids(ctx, bidder, ts), off, 2, BASE).to_host())
    assert got == oracle.q11_user_sessions(bidder, ts, off, 2, BASE) and got[4] == {7: (3, BASE + 100, BASE + 1500)}
    # out-of-order data: two sessions of one bidder closed in one epoch are one row
    rows = [(0, 5, 2000), (0, 6, 20), (2, 6, 30), (4, 5, 4000), (4, 5, 100)]
    bidder = np.array([r[1] for r in rows], np.int32)
    ts = np.array([BASE + r[2] for r in rows], np.int64)
    off = np.searchsorted(np.array([r[0] for r in rows]), np.arange(6)).astype(np.int64)
    res = ctx.q11_user_sessions(_bids(ctx, bidder, ts), off, 1, BASE)
    got = _as_dicts(res.to_host())
    assert got == oracle.q11_user_sessions(bidder, ts, off, 1, BASE) and got[4] == {5: (3, BASE + 100, BASE + 4000)}
    assert res.sessions_total == 3 and res.rows == 2
    # a run that starts inside the relation (aligned), empty runs, zero epochs
    pad = np.r_[np.full(8, 99, np.int32), bidder]
    res = ctx.q11_user_sessions(_bids(ctx, pad, np.r_[np.zeros(8, np.int64), ts]), off + 8, 1, BASE)
    assert _as_dicts(res.to_host()) == got
    empty = ctx.q11_user_sessions(_bids(ctx, bidder, ts), np.zeros(4, np.int64), 1, BASE)
    assert empty.rows == 0 and empty.offsets().tolist() == [0, 0, 0, 0]
    assert ctx.q11_user_sessions(_bids(ctx, bidder, ts), np.zeros(1, np.int64), 1, BASE).rows == 0
    with pytest.raises(FlockGpuError) as e:
        ctx.q11_user_sessions(_bids(ctx, bidder, ts), np.array([0, 3, 2, 4]), 1, BASE)
    assert e.value.code == _ffi.ERR_INVALID
    with pytest.raises(FlockGpuError) as e:
        ctx.q11_user_sessions(_bids(ctx, bidder, ts), np.array([0, 9]), 1, BASE)
    assert e.value.code == _ffi.ERR_INVALID


def test_q11_generated_stream(ctx):
    """The generator's bids under the benchmark's Session(10 s) window, through run_query."""
    from flock_amd.nexmark import NEXMarkSource, Window, run_query
    src = NEXMarkSource(40, 20_000, Window.session(10), seed=3)
    stream = src.generate_data(ctx, relations=("bid",))
    res = run_query(ctx, 11, stream)
    bidder = stream.bids.bidder.cpu().numpy()
    ts = stream.bids.b_date_time.cpu().numpy()
    off = stream.epoch_row_offsets("bid")
    got = res.to_host()
    w_off, w_b, w_c, w_mn, w_mx = oracle.q11_user_sessions_columnar(bidder, ts, off, 10, BASE)
    assert np.array_equal(got["offsets"], w_off) and res.rows > 1000
    assert np.array_equal(got["bidder"], w_b) and np.array_equal(got["bid_count"], w_c)
    assert np.array_equal(got["start_time"], w_mn) and np.array_equal(got["end_time"], w_mx)
    assert int(got["bid_count"].sum()) <= len(bidder)


def _columnar_equal(res, want):
    got = res.to_host()
    w_off, w_b, w_c, w_mn, w_mx = want
    assert np.array_equal(got["offsets"], w_off)
    assert np.array_equal(got["bidder"], w_b) and np.array_equal(got["bid_count"], w_c)
    assert np.array_equal(got["start_time"], w_mn) and np.array_equal(got["end_time"], w_mx)


@pytest.mark.parametrize("seed,shape", [(0, "far"), (1, "far_first"), (2, "extremes"), (3, "edge_of_span"), (4, "many_epochs")])
def test_q11_times_and_epochs_outside_the_sorted_payload(ctx, seed, shape):
    """Round 5: the sort carries (epoch, b_date_time - reference) as its 32-bit payload (q11.hip, q11_pack_kernel).  Runs whose times do
    not fit that span around the first row's time -- one row is enough --, or whose epoch count leaves the time too few bits, take the
    row-number payload and the gather instead; a time exactly on either end of the span still packs.  Same rows either way."""
    rng = np.random.default_rng(seed)
    n_epochs, per = 24, 700
    counts = rng.integers(1, per + 1, n_epochs)
    off = np.r_[0, np.cumsum(counts)].astype(np.int64)
    n = int(off[-1])
    ep = np.repeat(np.arange(n_epochs), counts)
    ts = BASE + ep * 1000 + np.concatenate([np.sort(rng.integers(0, 1000, c)) for c in counts]).astype(np.int64)
    bidder = rng.integers(50, 400, n).astype(np.int32)
    if shape == "far":                    # a few late rows, ~2 years ahead: they are sessions of their own (never closed inside the run)
        ts[rng.integers(1, n, 5)] += 1 << 36
    elif shape == "far_first":            # the reference row itself is the odd one
        ts[0] += 1 << 40
    elif shape == "extremes":             # the ends of the int64 range next to ordinary times
        ts[n // 2] = np.iinfo(np.int64).max
        ts[n // 3] = 0
    elif shape == "edge_of_span":         # 24 epochs -> 5 epoch bits, 27 time bits: first time - 2^26 and first time + 2^26 - 1 both fit
        ts[n // 2] = ts[0] - (1 << 26)
        ts[n // 3] = ts[0] + (1 << 26) - 1
    elif shape == "many_epochs":          # 2^21 + 3 epochs, nearly all empty: 22 epoch bits leave 10 for the time
        big = (1 << 21) + 3
        where = np.sort(rng.choice(big, n_epochs, replace=False))
        c2 = np.zeros(big, np.int64)
        c2[where] = counts
        off = np.r_[0, np.cumsum(c2)].astype(np.int64)
        ts = BASE + np.repeat(where, counts) * 1000 + (ts - BASE) % 1000
    timeout = 3
    want = oracle.q11_user_sessions_columnar(bidder, ts, off, timeout, BASE)
    ctx.profile_reset()
    ctx.profile(True)
    try:
        _columnar_equal(ctx.q11_user_sessions(_bids(ctx, bidder, ts), off, timeout, BASE), want)
        ran = ctx.profile_read()
    finally:
        ctx.profile(False)
    gathered = "q11_gather_kernel" in ran
    assert gathered == (shape != "edge_of_span") and ("q11_pack_kernel" in ran) == (shape != "many_epochs"), sorted(ran)
    if shape in ("far", "extremes"):      # the columnar restatement against the literal walk on exactly these rows
        walk = oracle.q11_user_sessions(bidder, ts, off, timeout, BASE)
        assert [sorted(d) for d in walk] == [want[1][want[0][t]:want[0][t + 1]].tolist() for t in range(n_epochs)]
