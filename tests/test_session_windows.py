"""Session windows kept on the host the way the reference's launcher keeps them (flock-function/src/aws/window/session.rs:64-321;
`flock_amd/session.py`) and q11 through the plan ABI: per epoch, the sessions the epoch closes -> `collect` on the plan of q11.sql
(tests/golden/plans/q11.json) -> the rows of the oracle's literal walk, which are also the rows of the whole-run device path
(`flockgpu_q11_user_sessions`)."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

import oracle

BASE = 1436918400_000
PLANS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plans")


def _stream(seed, n_epochs, per_epoch, n_bidders, jitter):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, per_epoch + 1, n_epochs)
    counts[rng.integers(0, n_epochs)] = 0     # an epoch without events still runs the time-out check
    off = np.r_[0, np.cumsum(counts)].astype(np.int64)
    ep = np.repeat(np.arange(n_epochs), counts)
    ms = np.concatenate([np.sort(rng.integers(0, 1000, c)) for c in counts]) if off[-1] else np.zeros(0, np.int64)
    late = rng.integers(0, jitter + 1, off[-1]) if jitter else 0
    ts = BASE + np.maximum(0, ep * 1000 + ms - late).astype(np.int64)
    bidder = rng.integers(1000, 1000 + n_bidders, off[-1]).astype(np.int32)
    bidder[rng.integers(0, max(off[-1], 1), off[-1] // 4)] = 1000   # a hot bidder
    auction = rng.integers(0, 50, off[-1]).astype(np.int32)
    epochs = [pa.record_batch([pa.array(auction[a:b]), pa.array(bidder[a:b]), pa.array(auction[a:b]), pa.array(ts[a:b]).cast(pa.timestamp("ms"))],
                              names=["auction", "bidder", "price", "b_date_time"]) if b > a else None for a, b in zip(off[:-1], off[1:])]
    return bidder, ts, off, epochs


CASES = [(0, 12, 40, 9, 2, 0), (1, 30, 250, 400, 3, 0), (2, 20, 600, 150, 1, 2500), (3, 25, 100, 300, 4, 9000), (5, 40, 150, 600, 10, 500)]


@pytest.mark.parametrize("seed,n_epochs,per_epoch,n_bidders,timeout,jitter", CASES)
def test_closed_sessions_per_epoch_are_the_walks(seed, n_epochs, per_epoch, n_bidders, timeout, jitter):
    """The bookkeeping alone (no GPU): the events handed out per epoch, grouped by bidder, are the oracle's rows."""
    from flock_amd.session import SessionWindows
    bidder, ts, off, epochs = _stream(seed, n_epochs, per_epoch, n_bidders, jitter)
    want = oracle.q11_user_sessions(bidder, ts, off, timeout, BASE)
    w = SessionWindows(timeout, base_time_ms=BASE)
    for t, rb in enumerate(epochs):
        got = {}
        for b in w.add_epoch(t, rb):
            k, tt = b.column("bidder").to_numpy(), b.column("b_date_time").cast(pa.int64()).to_numpy()
            for kk in np.unique(k):
                m = k == kk
                c, mn, mx = got.get(int(kk), (0, 2 ** 62, 0))
                got[int(kk)] = (c + int(m.sum()), min(mn, int(tt[m].min())), max(mx, int(tt[m].max())))
        assert got == want[t], t
        # only what OPEN sessions still point into is held (never one slot per epoch forever), and a batch that is mostly
        # dead is compacted: at most 4x the live rows (+ small batches) stay pinned
        live = sum(len(rows) for pieces, _ in w._open.values() for _, rows in pieces)
        assert (w.open_sessions > 0 or w.held_batches == 0) and w.held_rows <= 4 * live + 256 * w.held_batches
    assert sum(len(d) for d in want) > 0


def test_q11_plan_is_an_executable_operator_tree():
    from flock_amd.runtime import explain
    text = explain(json.load(open(os.path.join(PLANS, "q11.json"))))
    assert text.splitlines()[0] == "Project [bidder:Int32, bid_count:UInt64, start_time:Timestamp(ms), end_time:Timestamp(ms)]"
    assert "Aggregate(FinalPartitioned)" in text and "Aggregate(Partial)" in text and "Scan(bid)" in text


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_epochs,per_epoch,n_bidders,timeout,jitter", CASES[1:4])
def test_q11_through_the_plan_abi(seed, n_epochs, per_epoch, n_bidders, timeout, jitter):
    import torch
    from flock_amd import Bids, GpuContext
    from flock_amd.runtime import ExecutionContext
    from flock_amd.session import launch_session_query
    bidder, ts, off, epochs = _stream(seed, n_epochs, per_epoch, n_bidders, jitter)
    want = oracle.q11_user_sessions(bidder, ts, off, timeout, BASE)
    gpu = GpuContext(0)
    ctx = ExecutionContext([json.load(open(os.path.join(PLANS, "q11.json")))], gpu=gpu)
    try:
        per_epoch_batches = launch_session_query(ctx, epochs, timeout, base_time_ms=BASE)
        assert len(per_epoch_batches) == n_epochs
        for t, batches in enumerate(per_epoch_batches):
            got = {}
            for rb in batches:
                assert rb.schema.names == ["bidder", "bid_count", "start_time", "end_time"] and rb.schema.field("start_time").type == pa.timestamp("ms")
                for b, c, s, e in zip(rb["bidder"].to_pylist(), rb["bid_count"].to_pylist(), rb["start_time"].cast(pa.int64()).to_pylist(),
                                      rb["end_time"].cast(pa.int64()).to_pylist()):
                    assert b not in got
                    got[b] = (c, s, e)
            assert got == want[t], t
        # the whole run on the device (the batched-window ABI): the same rows
        dev = gpu.q11_user_sessions(Bids(None, torch.from_numpy(bidder).cuda(), None, torch.from_numpy(ts).cuda(), len(bidder)), off, timeout, BASE).to_host()
        o = dev["offsets"]
        for t in range(n_epochs):
            rows = {int(dev["bidder"][i]): (int(dev["bid_count"][i]), int(dev["start_time"][i]), int(dev["end_time"][i])) for i in range(o[t], o[t + 1])}
            assert rows == want[t], t
    finally:
        ctx.close()
        gpu.close()
