"""q6 through the plan ABI (benchmarks/src/nexmark/query/q6.sql, q6_plan.fmt): the one NEXMark query that needs WindowAggExec.  `window_agg_exec` with
ROW_NUMBER() is a device operator of the generic path since round 5 (relops.hpp: row_number_runs, over the sort_exec the planner puts underneath); the
rows equal the oracle's (oracle.q6_avg_price_by_seller = the numpy restatement, checked against the operator walk in tests/test_oracle_q6.py).
Ties: the reference's sort is not stable, so which of two EQUAL top bids of an auction wins there is unspecified; here (and in the oracle) the sort is
stable and the join's pair order decides -- the test streams are checked to hold no such tie, so the comparison does not lean on it."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

import oracle
from oracle import generic_ops as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAN = os.path.join(ROOT, "tests", "golden", "plans", "q6.json")
TS = pa.timestamp("ms")


def test_q6_plan_parses_with_its_two_windows():
    from flock_amd.runtime import explain
    txt = explain(json.load(open(PLAN)))
    assert txt.count("Window(ROW_NUMBER PARTITION BY") == 2 and "PARTITION BY a_id" in txt and "PARTITION BY seller" in txt
    assert txt.splitlines()[0].startswith("Project [seller:Int32, AVG(R.price):Float64]")


def test_other_window_functions_are_refused_by_name():
    from flock_amd import FlockGpuError
    from flock_amd.runtime import explain
    plan = json.load(open(PLAN))
    node = plan
    while node.get("execution_plan") != "window_agg_exec":
        node = node["input"]
    node["window_expr"][0].update({"fun": "Rank", "name": "RANK() PARTITION BY [#Q.seller]"})
    with pytest.raises(FlockGpuError) as e:
        explain(plan)
    assert "rank" in str(e.value).lower() and "ROW_NUMBER" in str(e.value)


def _batches(seed, eps, seconds):
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    n = eps * seconds
    au, bi = s.auctions(0, n), s.bids(0, n)
    # the generator's bids tie (several bids per millisecond, a few dozen price points per auction); the comparison must not lean on how ties fall,
    # so times become unique per bid (x 1000 + row, the auctions' ranges scaled along) and prices nearly so (checked by _no_ties)
    row = np.arange(len(bi["auction"]), dtype=np.int64)
    bi = dict(bi, b_date_time=bi["b_date_time"].astype(np.int64) * 1000 + row % 1000, price=((bi["price"].astype(np.int64) % (1 << 20)) * 2048 + row % 2048).astype(np.int32))
    au = dict(au, a_date_time=au["a_date_time"].astype(np.int64) * 1000, expires=au["expires"].astype(np.int64) * 1000 + 999)
    auction = pa.record_batch([pa.array(au["a_id"]), pa.array(au["a_date_time"]).cast(TS), pa.array(au["expires"]).cast(TS), pa.array(au["seller"])],
                              names=["a_id", "a_date_time", "expires", "seller"])
    bid = pa.record_batch([pa.array(bi["auction"]), pa.array(bi["price"]), pa.array(bi["b_date_time"]).cast(TS)], names=["auction", "price", "b_date_time"])
    return au, bi, auction, bid


def _no_ties(au, bi):
    """No auction with two equal top bids inside its time range, no seller with two winners at one time: the result does not depend on a tie."""
    ar, br = oracle._auction_bid_pairs(au["a_id"], au["a_date_time"], au["expires"], bi["auction"], bi["b_date_time"])
    key = np.stack([np.asarray(au["a_id"], np.int64)[ar], -np.asarray(bi["price"], np.int64)[br]], axis=1)
    o = np.lexsort((key[:, 1], key[:, 0]))
    k = key[o]
    first = np.r_[True, k[1:, 0] != k[:-1, 0]]
    second_equal = (~first[1:]) & first[:-1] & (k[1:, 1] == k[:-1, 1])
    w = o[first]
    st = np.stack([np.asarray(au["seller"], np.int64)[ar][w], np.asarray(bi["b_date_time"], np.int64)[br][w]], axis=1)
    return not second_equal.any() and len(np.unique(st, axis=0)) == len(st)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,eps,seconds,chunk", [(3, 20_000, 2, 7000), (11, 100_000, 3, 100_000), (5, 2000, 1, 300)])
def test_q6_through_the_plan_abi(seed, eps, seconds, chunk):
    from flock_amd import GpuContext
    from flock_amd.runtime import ExecutionContext, collect
    au, bi, auction, bid = _batches(seed, eps, seconds)
    assert _no_ties(au, bi)
    gpu = GpuContext(0)
    ctx = ExecutionContext([json.load(open(PLAN))], gpu=gpu)
    split = lambda rb: [rb.slice(a, chunk) for a in range(0, rb.num_rows, chunk)]
    for _ in range(2):   # twice on one plan: hints, cached statistics and arenas of the first run are in play the second time
        out = collect(ctx, [[split(auction)], [split(bid)]])[0]
        got = sorted((r for b in out for r in zip(b["seller"].to_pylist(), b["AVG(R.price)"].to_pylist())))
        s, a = oracle.q6_avg_price_by_seller(au["a_id"], au["a_date_time"], au["expires"], au["seller"], bi["auction"], bi["price"], bi["b_date_time"])
        assert len(s) > 10 and got == list(zip(s.tolist(), a.tolist()))
    assert out[0].schema.field("AVG(R.price)").type == pa.float64() and out[0].schema.field("seller").type == pa.int32()
    ctx.close()
    gpu.close()


@pytest.mark.gpu
def test_row_number_over_runs_with_nulls_and_two_keys():
    """window_agg_exec on its own: PARTITION BY (i, l) over a sorted input with NULL keys, against oracle/generic_ops.py: window_row_number; the window
    column comes first; a filter on it (rank <= 2) keeps the first two rows of every run."""
    from flock_amd import GpuContext
    from flock_amd.runtime import ExecutionContext, collect
    from test_plan_round5 import F, NAMES, batches, binary, cast, col, lit, pyrows, scan, table
    r = np.random.default_rng(21)
    t = table(5000, r, null_p=0.2)
    t["i"] = [None if x is None else x % 7 for x in t["i"]]
    t["l"] = [None if x is None else x % 3 for x in t["l"]]
    srt = {"execution_plan": "sort_exec", "input": scan(), "expr": [{"expr": col("i"), "options": {"descending": False, "nulls_first": False}},
                                                                   {"expr": col("l"), "options": {"descending": True, "nulls_first": True}},
                                                                   {"expr": col("j"), "options": {"descending": False, "nulls_first": False}}]}
    win = {"execution_plan": "window_agg_exec", "input": srt, "window_expr": [{"fun": "RowNumber", "name": "rn", "partition_by": [col("i"), col("l")], "order_by": []}]}
    pred = binary(cast({"physical_expr": "column", "name": "rn", "index": 0}, "Int64"), "LtEq", lit("Int64", 2))
    gpu = GpuContext(0)
    want_sorted = g.sort_exec(t, [("i", False, False), ("l", True, True), ("j", False, False)])
    want = g.window_row_number(want_sorted, ["i", "l"], "rn")
    for plan, expect in ((win, want), ({"execution_plan": "filter_exec", "predicate": pred, "input": win}, g.filter_exec(want, lambda x: x["rn"] <= 2))):
        ctx = ExecutionContext([plan], gpu=gpu)
        rb = collect(ctx, [[batches(t, 1700)]])[0][0]
        ctx.close()
        assert rb.schema.names[0] == "rn" and rb.schema.field("rn").type == pa.uint64()
        assert pyrows(rb) == g.rows(expect)
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("instances,on_device", [(None, False), (1, True)])
def test_q6_stage_by_stage_equals_the_whole_plan(instances, on_device):
    """The reference's distributed mode cuts q6 at its hash repartitions (flock/src/distributed_plan/stage.rs:269-367 -> flock_amd.stages.build_query_dag:
    seven stages); every ROW_NUMBER() partition lies inside one hash partition of its key, so the stages' union is the whole plan's result -- with 8
    function instances per stage over host batches, and with one instance per stage and the stage boundary in HBM."""
    from flock_amd import GpuContext
    from flock_amd import stages as S
    au, bi, auction, bid = _batches(3, 20_000, 2)
    gpu = GpuContext(0)
    kw = {} if instances is None else {"instances": instances, "share_sources": True, "on_device": on_device}
    run = S.StagedRun(gpu, S.build_query_dag(json.load(open(PLAN))), **kw)
    assert len(run.stages) == 7
    for _ in range(2):
        out = run.run({"auction": auction, "bid": bid})
        got = sorted((r for b in out for r in zip(b["seller"].to_pylist(), b["AVG(R.price)"].to_pylist())))
        s, a = oracle.q6_avg_price_by_seller(au["a_id"], au["a_date_time"], au["expires"], au["seller"], bi["auction"], bi["price"], bi["b_date_time"])
        assert got == list(zip(s.tolist(), a.tolist()))
    run.close()
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", [0, 10])
def test_pass_through_queries_q0_and_q10(q):
    """benchmarks/src/nexmark/query/q0.sql (`SELECT * FROM bid`) and q10.sql (the four bid columns by name): one projection over the scan (q0_plan.fmt,
    q10_plan.fmt).  The batches that come back hold the rows that went in, in order, with the schema's types (Timestamp(Millisecond) stays one)."""
    from flock_amd import GpuContext
    from flock_amd.runtime import ExecutionContext, collect
    s = oracle.NexmarkStream(seed=2, eps=30_000)
    bi = s.bids(0, 60_000)
    bid = pa.record_batch([pa.array(bi["auction"]), pa.array(bi["bidder"]), pa.array(bi["price"]), pa.array(bi["b_date_time"]).cast(TS)], names=["auction", "bidder", "price", "b_date_time"])
    gpu = GpuContext(0)
    ctx = ExecutionContext([json.load(open(os.path.join(ROOT, "tests", "golden", "plans", f"q{q}.json")))], gpu=gpu)
    out = collect(ctx, [[[bid.slice(0, 20_000), bid.slice(20_000)]]])[0]
    got = pa.Table.from_batches(out).combine_chunks().to_batches()[0]
    assert got.schema.names == bid.schema.names and got.schema.field("b_date_time").type == TS
    assert all(got[c].equals(bid[c]) for c in bid.schema.names)       # (the plan's fields are NOT NULL, pyarrow's inferred ones nullable: columns, not batches)
    ctx.close()
    gpu.close()
