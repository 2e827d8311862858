"""q6 (benchmarks/src/nexmark/query/q6.sql, q6_plan.fmt -- SURVEY.md section 8(f): the remaining join / aggregate query, the one that needs
WindowAggExec): the whole-column numpy oracle (oracle.q6_avg_price_by_seller) against the same plan walked operator by operator over Python rows
(oracle/generic_ops.py: nexmark_q6 = join, BETWEEN, sort, ROW_NUMBER per auction, = 1, sort, ROW_NUMBER per seller, <= 10, AVG) and against hand-worked
rows.  The reference's own q6 tests (flock/src/datasource/nexmark/queries/q6_v2.rs, q6_v3.rs) only print."""
import numpy as np
import pytest

import oracle
from oracle import generic_ops as g


def _tables(seed, eps, seconds):
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    n = eps * seconds
    au, bi = s.auctions(0, n), s.bids(0, n)
    auction = {k: [int(x) for x in au[k]] for k in ("a_id", "a_date_time", "expires", "seller")}
    bid = {k: [int(x) for x in bi[k]] for k in ("auction", "price", "b_date_time")}
    return au, bi, auction, bid


def test_row_number_runs_by_hand():
    t = {"k": [5, 5, 5, None, None, 7, 5], "v": [1, 2, 3, 4, 5, 6, 7]}
    w = g.window_row_number(t, ["k"], "rn")
    assert list(w) == ["rn", "k", "v"] and w["rn"] == [1, 2, 3, 1, 2, 1, 1]      # runs, not groups: the trailing 5 starts again
    assert g.window_row_number(t, [], "rn")["rn"] == [1, 2, 3, 4, 5, 6, 7]


def test_q6_by_hand():
    # auctions 1 (seller 9) and 2 (seller 9) and 3 (seller 4); bids outside [a_date_time, expires] do not count
    auction = {"a_id": [1, 2, 3], "a_date_time": [0, 0, 0], "expires": [100, 100, 50], "seller": [9, 9, 4]}
    bid = {"auction": [1, 1, 2, 3, 3, 2, 8], "price": [10, 30, 7, 99, 5, 50, 1], "b_date_time": [5, 6, 7, 60, 8, 101, 9]}
    got = g.nexmark_q6(auction, bid)
    assert dict(zip(got["seller"], got["AVG(R.price)"])) == {9: (30 + 7) / 2, 4: 5.0}
    s, a = oracle.q6_avg_price_by_seller(auction["a_id"], auction["a_date_time"], auction["expires"], auction["seller"], bid["auction"], bid["price"], bid["b_date_time"])
    assert s.tolist() == [4, 9] and a.tolist() == [5.0, 18.5]
    # only the LAST winner of seller 9 (by b_date_time): auction 2's bid at time 7
    s, a = oracle.q6_avg_price_by_seller(auction["a_id"], auction["a_date_time"], auction["expires"], auction["seller"], bid["auction"], bid["price"], bid["b_date_time"], last=1)
    assert dict(zip(s.tolist(), a.tolist())) == {4: 5.0, 9: 7.0}


@pytest.mark.parametrize("seed,eps,seconds", [(1, 2000, 3), (7, 5000, 2)])
def test_q6_numpy_equals_the_operator_walk_on_nexmark_events(seed, eps, seconds):
    au, bi, auction, bid = _tables(seed, eps, seconds)
    walk = g.nexmark_q6(auction, bid)
    s, a = oracle.q6_avg_price_by_seller(au["a_id"], au["a_date_time"], au["expires"], au["seller"], bi["auction"], bi["price"], bi["b_date_time"])
    assert len(s) > 20 and sorted(zip(walk["seller"], walk["AVG(R.price)"])) == list(zip(s.tolist(), a.tolist()))
