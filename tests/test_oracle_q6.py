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


@pytest.mark.parametrize("seed,eps,seconds", [(2, 4000, 3), (9, 20_000, 2)])
def test_q6_numpy_equals_an_independent_pandas_formulation(seed, eps, seconds):
    """A third engine: pandas (merge, boolean filter, stable sort_values + groupby().cumcount() as ROW_NUMBER, groupby().mean()) over the same events
    -- the way the other queries' oracles are cross-checked against pyarrow / Acero (tests/test_oracle_cross.py)."""
    pd = pytest.importorskip("pandas")
    au, bi, _, _ = _tables(seed, eps, seconds)
    a = pd.DataFrame({k: np.asarray(au[k]) for k in ("a_id", "a_date_time", "expires", "seller")})
    b = pd.DataFrame({k: np.asarray(bi[k]) for k in ("auction", "price", "b_date_time")})
    j = a.merge(b, left_on="a_id", right_on="auction", how="inner", sort=False)
    j = j[(j.b_date_time >= j.a_date_time) & (j.b_date_time <= j.expires)]
    j = j.sort_values(["a_id", "price"], ascending=[True, False], kind="stable")
    q = j[j.groupby("a_id").cumcount() == 0]
    q = q.sort_values(["seller", "b_date_time"], ascending=[True, False], kind="stable")
    r = q[q.groupby("seller").cumcount() < 10]
    want = r.groupby("seller")["price"].mean()
    s, avg = oracle.q6_avg_price_by_seller(au["a_id"], au["a_date_time"], au["expires"], au["seller"], bi["auction"], bi["price"], bi["b_date_time"])
    # (ties: pandas' merge keeps the left rows' order and, within one, the right rows' -- the (auction row, bid row) order of the numpy restatement)
    assert len(s) > 50 and s.tolist() == want.index.tolist()
    assert np.allclose(avg, want.to_numpy(), rtol=0, atol=1e-9) and np.array_equal(avg, want.to_numpy())
