"""Pins the scalar C oracle (oracle/nexmark_ops.c) on seeded NEXMark windows against
 (i) oracle/generic_ops.py (itself pinned to the reference's operator goldens) and
 (ii) pyarrow compute/acero as an independent Arrow-native engine."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import oracle
from oracle import generic_ops as g

SEEDS = [(1, 1000, 3), (7, 5000, 2), (42, 20000, 1)]  # (seed, eps, seconds)


def _tables(seed, eps, n0, n1):
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    b, a, p = s.bids(n0, n1), s.auctions(n0, n1), s.persons(n0, n1)
    bid = {k: v.tolist() for k, v in b.items()}
    auc = {k: v.tolist() for k, v in a.items() if not isinstance(v, oracle.Utf8)}
    per = {k: (v.to_pylist() if isinstance(v, oracle.Utf8) else v.tolist()) for k, v in p.items()}
    return (b, a, p), (bid, auc, per)


def _sorted(rows):
    return sorted(rows)


@pytest.mark.parametrize("seed,eps,seconds", SEEDS)
def test_q1_q2(seed, eps, seconds):
    for e in range(seconds):
        (b, _, _), (bid, _, _) = _tables(seed, eps, e * eps, (e + 1) * eps)
        # q1: exact f64 bits
        got = oracle.q1_project(b["price"])
        want = np.array(g.nexmark_q1(bid)["price"], np.float64)
        assert got.tobytes() == want.tobytes()
        pa_out = pc.multiply(pa.scalar(0.908, pa.float64()), pc.cast(pa.array(b["price"]), pa.float64()))
        assert got.tobytes() == pa_out.to_numpy().tobytes()
        # q2: exact, input order
        oa, op = oracle.q2_filter(b["auction"], b["price"])
        w = g.nexmark_q2(bid)
        assert oa.tolist() == w["auction"] and op.tolist() == w["price"]
        a64 = pc.cast(pa.array(b["auction"]), pa.int64())
        rem = pc.subtract(a64, pc.multiply(pc.divide(a64, 123), 123))
        mask = pc.equal(rem, 0)
        assert pc.filter(pa.array(b["auction"]), mask).to_numpy().tolist() == oa.tolist()
        assert pc.filter(pa.array(b["price"]), mask).to_numpy().tolist() == op.tolist()


@pytest.mark.parametrize("seed,eps,seconds", SEEDS)
def test_q3(seed, eps, seconds):
    total = 0
    for e in range(seconds):
        (_, a, p), (_, auc, per) = _tables(seed, eps, e * eps, (e + 1) * eps)
        ar, pr = oracle.q3_join(a["seller"], a["category"], p["p_id"], p["state"])
        names, cities, states = p["name"].to_pylist(), p["city"].to_pylist(), p["state"].to_pylist()
        got = [(names[j], cities[j], states[j], int(a["a_id"][i])) for i, j in zip(ar, pr)]
        want = g.rows(g.nexmark_q3(auc, per))
        assert got == want  # same probe order: right rows in order, left insertion order
        # pyarrow / acero as second opinion (multiset)
        ta = pa.table({k: a[k] for k in ("a_id", "seller", "category")}).filter(pc.equal(pc.field("category"), 10))
        tp = pa.table({"p_id": p["p_id"], "name": names, "city": cities, "state": states})
        tp = tp.filter(pc.is_in(pc.field("state"), pa.array(["or", "id", "ca"])))
        j = ta.join(tp, keys="seller", right_keys="p_id", join_type="inner")
        pa_rows = list(zip(j["name"].to_pylist(), j["city"].to_pylist(), j["state"].to_pylist(), j["a_id"].to_pylist()))
        assert _sorted(pa_rows) == _sorted(got)
        total += len(got)
    assert total > 0


@pytest.mark.parametrize("seed,eps,seconds", SEEDS)
def test_q5(seed, eps, seconds):
    (b, _, _), (bid, _, _) = _tables(seed, eps, 0, seconds * eps)
    oa, on = oracle.q5_hot_items(b["auction"])
    assert on.dtype == np.uint64 and oa.dtype == np.int32
    w = g.nexmark_q5(bid)
    assert _sorted(zip(oa.tolist(), on.tolist())) == _sorted(zip(w["auction"], w["num"]))
    t = pa.table({"auction": b["auction"]}).group_by("auction").aggregate([([], "count_all")])
    mx = pc.max(t["count_all"]).as_py()
    t = t.filter(pc.equal(pc.field("count_all"), mx))
    assert _sorted(zip(t["auction"].to_pylist(), t["count_all"].to_pylist())) == _sorted(zip(oa.tolist(), on.tolist()))
    # the AuctionBids sub-query itself
    k, c = oracle.count_by_key(b["auction"])
    full = g.hash_aggregate_exec({"auction": bid["auction"]}, ["auction"], [("num", "count", None)])
    assert k.tolist() == full["auction"] and c.tolist() == full["num"]


@pytest.mark.parametrize("seed,eps,seconds", SEEDS)
def test_q8(seed, eps, seconds):
    (_, a, p), (_, auc, per) = _tables(seed, eps, 0, seconds * eps)
    rows = oracle.q8_join(p["p_id"], p["name"], a["seller"])
    names = p["name"].to_pylist()
    got = [(int(p["p_id"][r]), names[r]) for r in rows]
    want = g.rows(g.nexmark_q8(per, auc))
    assert _sorted(got) == _sorted(want)
    assert len(got) > 0


def test_q8_duplicate_persons_collapse():
    # GROUP BY p_id, name must collapse exact duplicates but keep same-id/different-name rows
    p_id = np.array([5, 5, 5, 6, 7], np.int32)
    name = oracle.Utf8(np.array([0, 1, 2, 3, 4, 5], np.int32), np.frombuffer(b"aabcd", np.uint8).copy())
    seller = np.array([5, 7, 7, 9], np.int32)
    rows = oracle.q8_join(p_id, name, seller)
    assert rows.tolist() == [0, 2, 4]
    want = g.nexmark_q8({"p_id": p_id.tolist(), "name": name.to_pylist()}, {"seller": seller.tolist()})
    assert sorted(g.rows(want)) == [(5, "a"), (5, "b"), (7, "d")]


def test_q3_duplicate_build_keys_emit_every_pair():
    seller = np.array([1, 1, 2, 1], np.int32)
    category = np.array([10, 10, 10, 11], np.int32)
    p_id = np.array([1, 2, 1], np.int32)
    state = oracle.Utf8(np.array([0, 2, 4, 6], np.int32), np.frombuffer(b"orcaOR", np.uint8).copy())
    ar, pr = oracle.q3_join(seller, category, p_id, state)
    assert list(zip(ar.tolist(), pr.tolist())) == [(0, 0), (1, 0), (2, 1)]  # 'OR' != 'or' (bytewise)


def test_empty_windows():
    e = np.empty(0, np.int32)
    assert len(oracle.q1_project(e)) == 0
    assert len(oracle.q2_filter(e, e)[0]) == 0
    assert len(oracle.q5_hot_items(e)[0]) == 0
    st = oracle.Utf8(np.zeros(1, np.int32), np.empty(0, np.uint8))
    assert len(oracle.q3_join(e, e, e, st)[0]) == 0
    assert len(oracle.q8_join(e, st, e)) == 0


# ---------------------------------------------------------------- "next" queries (SURVEY.md section 8 f): q4, q7, q9
@pytest.mark.parametrize("seed,eps,seconds", SEEDS)
def test_q4_q7_q9_against_pyarrow(seed, eps, seconds):
    """The numpy restatements of q4 / q7 / q9 against pyarrow's join / filter / group_by (independent engine)."""
    total9 = 0
    for e in range(seconds):
        (b, a, _), _ = _tables(seed, eps, e * eps, (e + 1) * eps)
        # q7: rows reaching MAX(price)
        rows = oracle.q7_highest_bid(b["price"])
        mx = pc.max(pa.array(b["price"])).as_py()
        assert rows.tolist() == [i for i, p in enumerate(b["price"].tolist()) if p == mx]
        # inner query Q through pyarrow: join, BETWEEN, MAX GROUP BY
        ta = pa.table({"a_id": a["a_id"], "category": a["category"], "a_date_time": a["a_date_time"], "expires": a["expires"]})
        tb = pa.table({"auction": b["auction"], "price": b["price"], "b_date_time": b["b_date_time"],
                       "row": np.arange(len(b["price"]))})
        j = ta.join(tb, keys="a_id", right_keys="auction", join_type="inner")
        j = j.filter(pc.and_(pc.greater_equal(j["b_date_time"], j["a_date_time"]), pc.less_equal(j["b_date_time"], j["expires"])))
        q = j.group_by(["a_id", "category"]).aggregate([("price", "max")])
        # q9: bids whose (auction, price) = (id, final)
        back = tb.join(q.select(["a_id", "price_max"]), keys=["auction", "price"], right_keys=["a_id", "price_max"], join_type="inner")
        want9 = sorted(back["row"].to_pylist())
        got9 = oracle.q9_winning_bids(a["a_id"], a["a_date_time"], a["expires"], b["auction"], b["price"], b["b_date_time"])
        assert got9.tolist() == want9
        total9 += len(want9)
        # q4: AVG(final) GROUP BY category -- Float64 bits
        q4 = q.group_by("category").aggregate([("price_max", "mean")]).sort_by("category")
        cats, avg = oracle.q4_avg_final_by_category(a["a_id"], a["category"], a["a_date_time"], a["expires"], b["auction"],
                                                    b["price"], b["b_date_time"])
        assert cats.tolist() == q4["category"].to_pylist()
        assert avg.tobytes() == q4["price_max_mean"].to_numpy().astype(np.float64).tobytes()
    assert total9 > 0
