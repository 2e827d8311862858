"""q11 oracle: the literal session walk (oracle.q11_user_sessions, following flock-function/src/aws/window/session.rs line by
line) against hand-worked cases, and the whole-column restatement used as bench.py's CPU baseline against the walk."""
import numpy as np
import pytest

import oracle

BASE = 1_436_918_400_000


def _bids(rows):
    """rows: (epoch, bidder, ms after BASE) in arrival order -> columns + epoch offsets"""
    n_epochs = max(r[0] for r in rows) + 1
    rows = sorted(rows, key=lambda r: r[0])  # stable: arrival order inside an epoch kept
    bidder = np.array([r[1] for r in rows], np.int32)
    ts = np.array([BASE + r[2] for r in rows], np.int64)
    off = np.searchsorted(np.array([r[0] for r in rows]), np.arange(n_epochs + 1)).astype(np.int64)
    return bidder, ts, off


def _columnar_as_dicts(res):
    off, who, cnt, mn, mx = res
    return [{int(who[i]): (int(cnt[i]), int(mn[i]), int(mx[i])) for i in range(off[t], off[t + 1])} for t in range(len(off) - 1)]


def test_hand_worked_sessions():
    # timeout 2 s.  bidder 7: bids in epochs 0,1 (one session, last bid at second 1) -> clock 1+2+1 = 4: closed in epoch 4.
    # bidder 8: epoch 0 and epoch 5; its first session times out in epoch 3; the second one is still open at the end.
    # bidder 9: epochs 0, 3 with the epoch-3 bid 3.2 s after the last one: whole seconds 3 - 0 = 3 > 2, but the time-out
    #           check of epoch 3 runs AFTER the partition is added; the session had clock 3, 3 < 3 is false, so the gap
    #           test closes it in epoch 3 and the new session (last second 3, clock 6) stays open (7 epochs: 0..6 -> closes in 6)
    rows = [(0, 7, 100), (0, 8, 200), (0, 9, 300), (0, 7, 900), (1, 7, 1500), (3, 9, 3500), (5, 8, 5100), (6, 1, 6000)]
    bidder, ts, off = _bids(rows)
    got = oracle.q11_user_sessions(bidder, ts, off, 2, BASE)
    assert got[0] == {} and got[1] == {} and got[2] == {}
    assert got[3] == {8: (1, BASE + 200, BASE + 200), 9: (1, BASE + 300, BASE + 300)}
    assert got[4] == {7: (3, BASE + 100, BASE + 1500)}
    assert got[5] == {}
    assert got[6] == {9: (1, BASE + 3500, BASE + 3500)}
    assert _columnar_as_dicts(oracle.q11_user_sessions_columnar(bidder, ts, off, 2, BASE)) == got


def test_two_sessions_of_a_bidder_closed_in_one_epoch_are_one_row():
    # timeout 1 s.  Bidder 5's epoch-0 bid carries second 2 (ahead of the clock): its time-out check first fires in
    # epoch 2 + 1 + 1 = 4.  In epoch 4 the bidder's partition starts with second 4 (gap 2 > 1: the old session is handed
    # out) and ENDS with a late bid of second 0, so the epoch-4 check hands the new session out as well: q11.sql groups
    # both by bidder.  Bidder 6: seconds 0 (epoch 0) and 0 (epoch 2, late): no gap, not timed out before epoch 2 -> one
    # session of two bids, closed by the epoch-2 check.
    rows = [(0, 5, 2000), (0, 6, 20), (2, 6, 30), (4, 5, 4000), (4, 5, 100)]
    bidder, ts, off = _bids(rows)
    got = oracle.q11_user_sessions(bidder, ts, off, 1, BASE)
    assert got == [{}, {}, {6: (2, BASE + 20, BASE + 30)}, {}, {5: (3, BASE + 100, BASE + 4000)}]
    assert _columnar_as_dicts(oracle.q11_user_sessions_columnar(bidder, ts, off, 1, BASE)) == got


@pytest.mark.parametrize("seed,n_epochs,per_epoch,n_bidders,timeout,jitter", [
    (0, 12, 40, 9, 2, 0), (1, 30, 25, 40, 3, 0), (2, 20, 60, 15, 1, 2500), (3, 25, 10, 30, 4, 9000), (4, 8, 0, 3, 2, 0),
    (5, 40, 15, 60, 10, 500)])
def test_columnar_restatement_matches_the_walk(seed, n_epochs, per_epoch, n_bidders, timeout, jitter):
    rng = np.random.default_rng(seed)
    rows = []
    for t in range(n_epochs):
        k = int(rng.integers(0, per_epoch + 1)) if per_epoch else 0
        ms = np.sort(rng.integers(0, 1000, k))
        for i in range(k):
            late = int(rng.integers(0, jitter + 1)) if jitter else 0
            rows.append((t, int(rng.integers(100, 100 + n_bidders)), max(0, t * 1000 + int(ms[i]) - late)))
    if not rows:
        rows = [(n_epochs - 1, 100, 0)]
    bidder, ts, off = _bids(rows)
    off = np.r_[off, np.full(n_epochs + 1 - len(off), off[-1])] if len(off) < n_epochs + 1 else off
    want = oracle.q11_user_sessions(bidder, ts, off, timeout, BASE)
    assert _columnar_as_dicts(oracle.q11_user_sessions_columnar(bidder, ts, off, timeout, BASE)) == want
    if n_bidders >= 40 and jitter == 0:
        assert sum(len(d) for d in want) > 3


def _q11_per_bidder_then_acero(bidder, ts, off, timeout_s, base_ms):
    """An independent formulation: the launcher part as a PER-BIDDER event simulation (the walk above goes
    epoch by epoch over all bidders, the columnar restatement compares neighbours of one sorted array), and q11.sql itself --
    GROUP BY bidder: COUNT(*), MIN(b_date_time), MAX(b_date_time) -- evaluated by Arrow C++'s hash aggregation over the rows each
    epoch hands out."""
    import pyarrow as pa
    n_epochs = len(off) - 1
    ep = np.searchsorted(off, np.arange(off[-1]), side="right") - 1
    handed = [[] for _ in range(n_epochs)]                    # per epoch: row arrays of the sessions closed there
    base_s = base_ms // 1000
    tbl = pa.table({"bidder": pa.array(bidder), "row": pa.array(np.arange(len(bidder), dtype=np.int64))})
    groups = tbl.group_by("bidder", use_threads=False).aggregate([("row", "list")])
    for rows in groups["row_list"].to_pylist():
        rows = np.asarray(rows, np.int64)                      # arrival order
        parts = np.split(rows, np.flatnonzero(np.diff(ep[rows])) + 1)
        open_rows, open_ep = None, None
        for p in parts:
            e = int(ep[p[0]])
            if open_rows is not None:
                last_s = int(ts[open_rows[-1]]) // 1000
                t_close = max(open_ep, last_s - base_s + timeout_s + 1)   # first epoch whose time-out check fires
                if t_close < e:
                    handed[t_close].append(open_rows)
                    open_rows = None
                elif int(ts[p[0]]) // 1000 - last_s > timeout_s:
                    handed[e].append(open_rows)
                    open_rows = None
            open_rows = p if open_rows is None else np.concatenate([open_rows, p])
            open_ep = e
        if open_rows is not None:
            t_close = max(open_ep, int(ts[open_rows[-1]]) // 1000 - base_s + timeout_s + 1)
            if t_close <= n_epochs - 1:
                handed[t_close].append(open_rows)
    out = []
    for e in range(n_epochs):
        if not handed[e]:
            out.append({})
            continue
        r = np.concatenate(handed[e])
        t = pa.table({"bidder": pa.array(bidder[r]), "b_date_time": pa.array(ts[r])})
        g = t.group_by("bidder").aggregate([("b_date_time", "count"), ("b_date_time", "min"), ("b_date_time", "max")])
        out.append({int(b): (int(c), int(mn), int(mx)) for b, c, mn, mx in zip(g["bidder"].to_pylist(), g["b_date_time_count"].to_pylist(),
                                                                              g["b_date_time_min"].to_pylist(), g["b_date_time_max"].to_pylist())})
    return out


@pytest.mark.parametrize("seed,eps,seconds,timeout", [(3, 400, 40, 10), (4, 2000, 25, 2), (5, 150, 60, 5)])
def test_q11_walk_equals_an_independent_pyarrow_formulation_on_nexmark_bids(seed, eps, seconds, timeout):
    """Seeded NEXMark bids (the generator's hot / cold bidders and its event-time order), ElementWise epochs of one second."""
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    n = eps * seconds
    cols = s.bids(0, n, columns=("bidder", "b_date_time"))
    off = np.array([s.counts(0, e * eps)[2] for e in range(seconds + 1)], np.int64)
    base = int(cols["b_date_time"][0]) // 1000 * 1000
    want = oracle.q11_user_sessions(cols["bidder"], cols["b_date_time"], off, timeout, base)
    got = _q11_per_bidder_then_acero(cols["bidder"], cols["b_date_time"], off, timeout, base)
    assert got == want
    assert sum(len(d) for d in want) > 10
    assert _columnar_as_dicts(oracle.q11_user_sessions_columnar(cols["bidder"], cols["b_date_time"], off, timeout, base)) == want


def test_q11_independent_formulation_on_late_and_early_bids():
    rng = np.random.default_rng(11)
    rows = []
    for t in range(30):
        for _ in range(int(rng.integers(0, 30))):
            rows.append((t, int(rng.integers(100, 130)), max(0, t * 1000 + int(rng.integers(-4000, 3000)))))
    bidder, ts, off = _bids(rows)
    off = np.r_[off, np.full(31 - len(off), off[-1])] if len(off) < 31 else off
    for timeout in (1, 3):
        assert _q11_per_bidder_then_acero(bidder, ts, off, timeout, BASE) == oracle.q11_user_sessions(bidder, ts, off, timeout, BASE)
