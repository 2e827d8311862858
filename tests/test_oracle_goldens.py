"""Pins oracle/generic_ops.py against every operator-level golden vector the
reference's own tests hold at the ExecutionContext::execute boundary
(SURVEY.md section 8c).  Values transcribed from the cited reference tests."""
from oracle import generic_ops as g

# flock/src/runtime/context.rs:437-472 == flock/src/launcher/local.rs:190-215 (same 8-row table)
T8 = {
    "c1": [90, 90, 91, 101, 92, 102, 93, 103],
    "c2": [92.1, 93.2, 95.3, 96.4, 98.5, 99.6, 100.7, 101.8],
    "c3": ["a", "a", "d", "b", "b", "d", "c", "c"],
    "c4": [33, 1, 54, 33, 12, 75, 2, 87],
    "c5": ["rapport", "pedantic", "mimesis", "haptic", "baksheesh", "amok", "devious", "c"],
    "neg": [-90, -90, -91, -101, -92, -102, -93, -103],
}


def test_filter_group_max_min_sort_context_rs_493():
    # SELECT MAX(c1), MIN(c2), c3 FROM test WHERE c2 < 99 GROUP BY c3 ORDER BY c3   (context.rs:477-503)
    f = g.filter_exec(T8, lambda r: r["c2"] < 99.0)
    a = g.hash_aggregate_exec(f, ["c3"], [("MAX(test.c1)", "max", "c1"), ("MIN(test.c2)", "min", "c2")])
    s = g.sort_exec(a, [("c3", False)])
    assert g.rows({k: s[k] for k in ("MAX(test.c1)", "MIN(test.c2)", "c3")}) == [
        (90, 92.1, "a"), (101, 96.4, "b"), (91, 95.3, "d")]


def test_utf8_join_sort_limit_context_rs_579():
    # SELECT a, b, d FROM t1 JOIN t2 ON a = c ORDER BY a ASC LIMIT 3
    # (context.rs:579-589 == distributed_plan/stage.rs:937-947 == driver/funcgen/dag.rs:1012-1022
    #  == launcher/aws/mod.rs:317-327)
    t1 = {"a": ["a", "b", "c", "d"], "b": [1, 10, 10, 100]}
    t2 = {"c": ["a", "b", "c", "d"], "d": [1, 10, 10, 100]}
    j = g.hash_join_inner(t1, t2, [("a", "c")])
    assert list(j) == ["a", "b", "c", "d"]  # left cols ++ right cols
    out = g.limit_exec(g.sort_exec({k: j[k] for k in ("a", "b", "d")}, [("a", False)]), 3)
    assert g.rows(out) == [("a", 1, 1), ("b", 10, 10), ("c", 10, 10)]


def test_min_avg_count_local_rs_223():
    # SELECT MIN(c1), AVG(c4), COUNT(c3) FROM test_table  -> 90 | 37.125 | 8   (launcher/local.rs:171-231)
    a = g.hash_aggregate_exec(T8, [], [("min", "min", "c1"), ("avg", "avg", "c4"), ("count", "count", "c3")])
    assert g.rows(a) == [(90, 37.125, 8)]


def _batches(n):
    return [{"c0": [1, 2, 3, 4, 5, 6, 7, 8]} for _ in range(n)]


def test_coalesce_batches_transmute_rs_298():
    # 10 x 8 rows coalesced at 20 -> 24, 24, 24, 8   (transmute.rs:298-318)
    out = g.coalesce_batches(_batches(10), 20)
    assert [g.num_rows(b) for b in out] == [24, 24, 24, 8]


def test_round_robin_shapes_transmute_rs_320():
    # 1 -> 4: 13,13,12,12 (transmute.rs:320-336); 3 -> 1: 150 (:338-352); 3 -> 5: 30 each (:354-372)
    assert [len(p) for p in g.repartition_round_robin([_batches(50)], 4)] == [13, 13, 12, 12]
    assert [len(p) for p in g.repartition_round_robin([_batches(50)] * 3, 1)] == [150]
    assert [len(p) for p in g.repartition_round_robin([_batches(50)] * 3, 5)] == [30] * 5


def test_hash_repartition_preserves_rows_transmute_rs_374():
    # Hash([c0], 8) over 3 x 50 batches of 8 rows: 8 partitions, 8*50*3 rows   (transmute.rs:374-393)
    out = g.repartition_hash([_batches(50)] * 3, "c0", 8)
    assert len(out) == 8
    assert sum(g.num_rows(b) for p in out for b in p) == 8 * 50 * 3


def test_ungrouped_aggregate_over_empty_input_yields_one_row():
    # q5 on an empty window: MAX -> NULL, inner join on num = maxn yields nothing
    assert g.rows(g.nexmark_q5({"auction": []})) == []
    a = g.hash_aggregate_exec({"num": []}, [], [("maxn", "max", "num")])
    assert g.rows(a) == [(None,)]
