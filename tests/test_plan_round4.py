"""Round-4 additions to the plan-level ABI (include/flockgpu_plan.h), each against the oracle:
  * SortExec / GlobalLimitExec on the device (reference goldens end in them: context.rs:471,549; stage.rs:337 cuts at sort_exec;
    launcher/aws/mod.rs:350 `ORDER BY a_id`)
  * the device-side pane ring for hopping windows (window/hopping.rs:52-74 re-sends every window whole)
  * asynchronous execute (one tokio task per plan, context.rs:172-191)
  * the hash-placement guard for shuffling stages (shuffle_writer.rs:106-128; actor.rs:425-543)."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

import oracle
from oracle import generic_ops as g
from test_plan_boundary import PLANS, TS, _auction_batches, _bid_batches, _person_batches, _plan, _utf8


@pytest.fixture(scope="module")
def gpu():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _field(name, dt, nullable=False):
    return {"data_type": dt, "dict_id": 0, "dict_is_ordered": False, "name": name, "nullable": nullable}


_TS = {"Timestamp": ["Millisecond", None]}
_SORT_FIELDS = [_field("i", "Int32"), _field("l", "Int64"), _field("u", "UInt64"), _field("f", "Float64"), _field("s", "Utf8"), _field("t", _TS)]


def _sort_plan(keys, limit=None):
    scan = {"execution_plan": "memory_exec", "schema": {"fields": _SORT_FIELDS, "metadata": {}}, "projection": list(range(len(_SORT_FIELDS)))}
    names = [f["name"] for f in _SORT_FIELDS]
    plan = {"execution_plan": "sort_exec", "input": scan,
            "expr": [{"expr": {"physical_expr": "column", "name": k, "index": names.index(k)}, "options": {"descending": d, "nulls_first": d}}
                     for k, d in keys]}
    return plan if limit is None else {"execution_plan": "global_limit_exec", "limit": limit, "input": plan}


def _sort_table(n, seed):
    r = np.random.default_rng(seed)
    words = [b"", b"a", b"a\x00", b"ab", b"abcdefgh", b"abcdefgh\x00", b"abcdefghi", b"abcdefghijklmnopq", b"abcdefghijklmnopr", b"zz", b"\xc3\xa9t\xc3\xa9",
             b"Walton Abrams", b"Walton Abramson", b"b" * 40, b"b" * 39 + b"a"]
    return {
        "i": r.integers(-5, 5, n).astype(np.int32) * np.int32(400_000_000),          # both signs, the full 32-bit span, many ties
        "l": r.integers(-2**62, 2**62, n, dtype=np.int64) // np.int64(r.integers(1, 1 << 40)),
        "u": r.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + r.integers(0, 2, n).astype(np.uint64),
        "f": np.where(r.random(n) < 0.3, np.round(r.normal(0, 3, n)), r.normal(0, 1e6, n)) + 0.0,   # ties, both signs (no -0.0 / NaN)
        "s": [words[k] for k in r.integers(0, len(words), n)],
        "t": r.integers(1_436_918_400_000, 1_436_918_400_000 + 50, n).astype(np.int64),
    }


def _sort_batch(t):
    f = t["f"].copy()
    f[f == 0] = 0.0
    return pa.record_batch([pa.array(t["i"]), pa.array(t["l"]), pa.array(t["u"]), pa.array(f), pa.array(t["s"], pa.binary()).cast(pa.string()),
                            pa.array(t["t"]).cast(TS)], names=["i", "l", "u", "f", "s", "t"])


def _rows_in_order(rb):
    cols = []
    for c in rb.schema.names:
        col = rb[c]
        if pa.types.is_timestamp(col.type):
            col = col.cast(pa.int64())
        if pa.types.is_string(col.type):
            col = col.cast(pa.binary())
        cols.append(col.to_pylist())
    return list(zip(*cols))


# ------------------------------------------------------------------ CPU
def test_sort_and_limit_parse_into_the_operator_tree():
    from flock_amd.runtime import explain
    text = explain(_sort_plan([("s", False), ("i", True)], limit=7))
    assert text.splitlines()[0].startswith("Limit(7)") and "Sort(s ASC, i DESC)" in text.splitlines()[1]
    # ORDER BY a numeric expression: its value as a column under the sort, dropped again above it (round 5, tests/test_plan_round5b.py); an
    # expression without a numeric type (a comparison) is handed back, never mis-executed
    from flock_amd import FlockGpuError, _ffi
    ok = _sort_plan([("i", False)])
    key = ok["expr"][0]["expr"]
    ok["expr"][0]["expr"] = {"physical_expr": "binary_expr", "op": "Plus", "left": key, "right": key}
    assert [l.split()[0].split("(")[0] for l in explain(ok).splitlines()[:3]] == ["Project", "Sort", "Project"]
    bad = _sort_plan([("i", False)])
    bad["expr"][0]["expr"] = {"physical_expr": "binary_expr", "op": "Lt", "left": key, "right": key}
    with pytest.raises(FlockGpuError) as e:
        explain(bad)
    assert e.value.code == _ffi.ERR_UNSUPPORTED and "ORDER BY" in str(e.value)


def test_partition_scheme_is_named_and_checked():
    import ctypes as C
    from flock_amd import _ffi
    from flock_amd.runtime import partition_scheme
    lib = _ffi.load()
    name = partition_scheme()
    assert name.startswith("flockgpu/") and lib.flockgpu_plan_check_partition_scheme(name.encode()) == _ffi.OK
    assert lib.flockgpu_plan_check_partition_scheme(b"datafusion/ahash-0000") == _ffi.ERR_UNSUPPORTED
    assert lib.flockgpu_plan_check_partition_scheme(None) == _ffi.ERR_UNSUPPORTED


# ------------------------------------------------------------------ GPU: ORDER BY / LIMIT
@pytest.mark.gpu
@pytest.mark.parametrize("keys,limit", [([("i", False)], None), ([("i", True)], 10), ([("l", False)], None), ([("u", True)], None),
                                        ([("f", False)], None), ([("f", True), ("i", False)], 33), ([("s", False)], None), ([("s", True)], 5),
                                        ([("t", False), ("s", False), ("u", False)], None), ([("i", False), ("s", True), ("l", False)], 1000),
                                        ([("s", False)], 0)])
@pytest.mark.parametrize("n,seed", [(1, 1), (700, 2), (20_000, 3)])
def test_order_by_and_limit_equal_the_oracle_row_for_row(gpu, keys, limit, n, seed):
    """Stable ORDER BY over every column type, ASC / DESC, multi-key, with and without LIMIT: the rows come back in exactly the
    order the oracle's stable lexicographic sort gives (ties in input order)."""
    from flock_amd.runtime import ExecutionContext, collect
    t = _sort_table(n, seed)
    ctx = ExecutionContext([_sort_plan(keys, limit)], gpu=gpu)
    rb = collect(ctx, [[[_sort_batch(t)]]])[0][0]
    ctx.close()
    table = {k: (v.tolist() if hasattr(v, "tolist") else list(v)) for k, v in t.items()}
    want = g.sort_exec(table, keys)
    if limit is not None:
        want = g.limit_exec(want, limit)
    assert rb.schema.names == ["i", "l", "u", "f", "s", "t"]
    assert _rows_in_order(rb) == g.rows(want)


@pytest.mark.gpu
def test_empty_input_sorts_to_nothing(gpu):
    from flock_amd.runtime import ExecutionContext, collect
    ctx = ExecutionContext([_sort_plan([("s", False), ("i", True)], limit=3)], gpu=gpu)
    assert collect(ctx, [[[]]])[0][0].num_rows == 0
    ctx.close()


@pytest.mark.gpu
def test_q3_order_by_a_id_through_the_plan_path(gpu):
    """launcher/aws/mod.rs:340-351: the reference's distributed == local differential runs q3 with `ORDER BY a_id ASC`; a_id is unique
    per result row, so the order is fully determined: row for row against the oracle."""
    from flock_amd.runtime import ExecutionContext, collect
    s = oracle.NexmarkStream(seed=17, eps=40_000)
    n = 160_000
    ctx = ExecutionContext([open(os.path.join(PLANS, "q3_sorted.json")).read()], name="q3-sorted", gpu=gpu)
    assert "fused q3" in ctx.plans[0].description and ctx.plans[0].description.startswith("Sort(a_id ASC)")
    rb = collect(ctx, [[_person_batches(s, 0, n, n)], [_auction_batches(s, 0, n, n)]])[0][0]
    ctx.close()
    a, p = s.auctions(0, n), s.persons(0, n)
    ar, pr = oracle.q3_join(a["seller"], a["category"], p["p_id"], p["state"])
    names, cities, states = _utf8(p["name"]).to_pylist(), _utf8(p["city"]).to_pylist(), _utf8(p["state"]).to_pylist()
    want = sorted(((names[j], cities[j], states[j], int(a["a_id"][i])) for i, j in zip(ar.tolist(), pr.tolist())), key=lambda r: r[3])
    got = list(zip(rb["name"].to_pylist(), rb["city"].to_pylist(), rb["state"].to_pylist(), rb["a_id"].to_pylist()))
    assert got == want and len(got) > 100


@pytest.mark.gpu
def test_sorted_stage_plans_run_stage_by_stage(gpu):
    """The stages build_query_dag cuts out of `... ORDER BY a LIMIT 3` (stage.rs:337: a cut at sort_exec) all execute: the join
    stage's partitions feed the sort stage, whose rows are context.rs:579-587 in order."""
    from flock_amd.runtime import ExecutionContext, collect
    from flock_amd.stages import build_query_dag
    t1 = pa.record_batch([pa.array(["a", "b", "c", "d"]), pa.array([1, 10, 10, 100], pa.int32())], names=["a", "b"])
    t2 = pa.record_batch([pa.array(["a", "b", "c", "d"]), pa.array([1, 10, 10, 100], pa.int32())], names=["c", "d"])
    from test_stage_plans import run_staged
    stages = build_query_dag(json.load(open(os.path.join(PLANS, "golden_join_sorted.json"))))
    assert len(stages) == 4                                               # two repartition plans, the join, sort + limit (stage.rs:776-901)
    got, _, _ = run_staged(gpu, stages, {"t1": t1, "t2": t2}, chunks=1)
    assert len(got) == 1
    assert list(zip(got[0]["a"].to_pylist(), got[0]["b"].to_pylist(), got[0]["d"].to_pylist())) == [("a", 1, 1), ("b", 10, 10), ("c", 10, 10)]


# ------------------------------------------------------------------ GPU: the pane ring
def _q5_rows(rb):
    return sorted(zip(rb["auction"].to_pylist(), rb["num"].to_pylist()))


@pytest.mark.gpu
@pytest.mark.parametrize("generic_only", [False, True])
@pytest.mark.parametrize("size,hop", [(4, 2), (6, 2), (3, 3)])
def test_q5_pane_ring_equals_whole_window_feeds_equals_oracle(gpu, generic_only, size, hop):
    """hopping(size, hop) q5 through `collect`: with the ring every pane is fed ONCE and the window results equal those of the
    reference's protocol (every window re-sent whole, hopping.rs:52-74) and the oracle -- on every window, incl. the ramp-up windows
    of fewer panes and an empty pane.  generic_only: the rows ring (any plan); else q5's state ring (a pane is counted once)."""
    from flock_amd.runtime import ExecutionContext, collect
    eps, seconds = 8_000, 16
    s = oracle.NexmarkStream(seed=31, eps=eps)
    ppw, n_panes = size // hop, seconds // hop
    pane_batches = [_bid_batches(s, p * hop * eps, (p + 1) * hop * eps, 5_000) for p in range(n_panes)]
    pane_host = [s.bids(p * hop * eps, (p + 1) * hop * eps)["auction"] for p in range(n_panes)]
    pane_batches[3], pane_host[3] = [], np.zeros(0, np.int32)            # a pane in which nothing arrived
    ring = ExecutionContext([_plan(5)], name="q5-ring", gpu=gpu, generic_only=generic_only)
    whole = ExecutionContext([_plan(5)], name="q5-whole", gpu=gpu, generic_only=generic_only)
    ring.open_window_ring(ppw)
    for p in range(n_panes):
        rb = collect(ring, [[pane_batches[p]]], pane=p)[0][0]
        lo = max(0, p - ppw + 1)
        first, held, _ = ring.plans[0].ring_state()                     # (after clean_data_sources: the oldest pane of a full ring is gone)
        assert first + held == p + 1 and held == (ppw - 1 if p + 1 >= ppw else p + 1)
        window = [b for q in range(lo, p + 1) for b in pane_batches[q]]
        ref = collect(whole, [[window]])[0][0]
        host = np.concatenate(pane_host[lo:p + 1]) if p + 1 > lo else np.zeros(0, np.int32)
        oa, on = oracle.q5_hot_items(host) if len(host) else (np.zeros(0, np.int32), np.zeros(0, np.uint64))
        assert _q5_rows(rb) == _q5_rows(ref) == sorted(zip(oa.tolist(), on.tolist())), (p, lo)
        assert rb.schema == ref.schema
    ring.close_window_ring()
    # after ring_close the plan feeds whole windows again
    rb = collect(ring, [[pane_batches[0]]])[0][0]
    oa, on = oracle.q5_hot_items(pane_host[0])
    assert _q5_rows(rb) == sorted(zip(oa.tolist(), on.tolist()))
    ring.close()
    whole.close()


@pytest.mark.gpu
def test_ring_retires_the_oldest_pane_and_counts_panes(gpu):
    from flock_amd.runtime import ExecutionContext, collect
    s = oracle.NexmarkStream(seed=2, eps=2_000)
    ctx = ExecutionContext([_plan(5)], gpu=gpu)
    ctx.open_window_ring(3)
    assert ctx.plans[0].ring_state() == (0, 0, 3)
    for p in range(10, 16):                                                # pane ids need not start at 0
        collect(ctx, [[_bid_batches(s, (p - 10) * 2_000, (p - 9) * 2_000, 2_000)]], pane=p)
        first, held, ppw = ctx.plans[0].ring_state()
        assert ppw == 3 and first + held == p + 1 and held == min(2, p - 9)   # (after clean_data_sources: the oldest of a full ring is gone)
    ctx.close()


@pytest.mark.gpu
def test_ring_refuses_out_of_order_and_skipped_panes(gpu):
    from flock_amd import FlockGpuError, _ffi
    from flock_amd.runtime import ExecutionContext
    s = oracle.NexmarkStream(seed=3, eps=2_000)
    b = lambda p: [[_bid_batches(s, p * 2_000, (p + 1) * 2_000, 2_000)]]
    ctx = ExecutionContext([_plan(5)], gpu=gpu)
    with pytest.raises(ValueError):
        ctx.feed_data_sources(b(0), pane=0)                                 # no ring open
    ctx.open_window_ring(2)
    with pytest.raises(ValueError):
        ctx.feed_data_sources(b(0))                                         # a ring is open: panes only
    ctx.feed_data_sources(b(4), pane=4)
    ctx.feed_data_sources(b(5), pane=5)
    want = _q5_rows(ctx.execute()[0][0])
    for bad in (3, 4, 7):                                                   # older than the ring, closed pane, skipped pane
        with pytest.raises(FlockGpuError) as e:
            ctx.feed_data_sources(b(bad), pane=bad)
        assert e.value.code == _ffi.ERR_INVALID and "pane" in str(e.value)
    with pytest.raises(FlockGpuError) as e:                                 # the successor needs room: the window must end first
        ctx.feed_data_sources(b(6), pane=6)
    assert e.value.code == _ffi.ERR_INVALID and "full" in str(e.value)
    assert _q5_rows(ctx.execute()[0][0]) == want                            # a refused feed left the ring as it was
    # a feed the plan refuses (wrong column type) also leaves the ring as it was -- even when it would have opened a new pane
    ctx.clean_data_sources()
    wrong = pa.record_batch([pa.array([1, 2], pa.int64())], names=["auction"])
    with pytest.raises(FlockGpuError):
        ctx.feed_data_sources([[[wrong]]], pane=6)
    assert ctx.plans[0].ring_state() == (5, 1, 2)
    ctx.feed_data_sources(b(6), pane=6)
    host = s.bids(5 * 2_000, 7 * 2_000)["auction"]
    oa, on = oracle.q5_hot_items(host)
    assert _q5_rows(ctx.execute()[0][0]) == sorted(zip(oa.tolist(), on.tolist()))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("generic_only", [False, True])
def test_prefetched_panes_equal_fed_panes(gpu, generic_only):
    """flockgpu_plan_prefetch_pane: pane p + 1 crosses PCIe into side buffers while window p executes, and is appended device to device when
    its turn comes.  Same windows as feeding every pane the ordinary way, on q5's state ring and on the rows ring (generic operators);
    panes of different sizes, an empty pane in between, pageable batches cut into several pieces."""
    from flock_amd.runtime import ExecutionContext, collect
    s = oracle.NexmarkStream(seed=12, eps=3_000)
    sizes = [3_000, 9_000, 0, 4_500, 30_000, 3_000, 3_000]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    pane = lambda p: [[_bid_batches(s, int(starts[p]), int(starts[p + 1]), 2_500)]] if sizes[p] else [[[]]]
    plain = ExecutionContext([_plan(5)], name="q5-plain", gpu=gpu, generic_only=generic_only)
    ahead = ExecutionContext([_plan(5)], name="q5-ahead", gpu=gpu, generic_only=generic_only)
    plain.open_window_ring(2)
    ahead.open_window_ring(2)
    ahead.feed_data_sources(pane(0), pane=0)
    for p in range(len(sizes)):
        want = collect(plain, pane(p), pane=p)[0][0]
        if p + 1 < len(sizes) and sizes[p + 1]:
            ahead.prefetch_data_sources(pane(p + 1), pane=p + 1)          # ... on its way while window p executes
        got = ahead.execute()[0][0]
        ahead.clean_data_sources()
        assert _q5_rows(got) == _q5_rows(want), p
        host = s.bids(int(starts[max(p - 1, 0)]), int(starts[p + 1]))["auction"]
        if len(host):
            oa, on = oracle.q5_hot_items(host)
            assert _q5_rows(got) == sorted(zip(oa.tolist(), on.tolist())), p
        if p + 1 < len(sizes):
            if sizes[p + 1]:
                ahead.feed_data_sources(None, pane=p + 1)
            else:
                ahead.feed_data_sources(pane(p + 1), pane=p + 1)
    plain.close()
    ahead.close()


@pytest.mark.gpu
def test_prefetch_refusals_leave_the_ring_as_it_was(gpu):
    from flock_amd import FlockGpuError, _ffi
    from flock_amd.runtime import ExecutionContext
    s = oracle.NexmarkStream(seed=13, eps=2_000)
    b = lambda p: [[_bid_batches(s, p * 2_000, (p + 1) * 2_000, 2_000)]]
    ctx = ExecutionContext([_plan(5)], gpu=gpu)
    with pytest.raises(ValueError):
        ctx.prefetch_data_sources(b(0), pane=0)                             # no ring open
    ctx.open_window_ring(2)
    ctx.feed_data_sources(b(0), pane=0)
    with pytest.raises(FlockGpuError) as e:
        ctx.prefetch_data_sources(b(3), pane=3)                             # not the next pane
    assert e.value.code == _ffi.ERR_INVALID and "next" in str(e.value)
    ctx.prefetch_data_sources(b(1), pane=1)
    with pytest.raises(FlockGpuError) as e:
        ctx.plans[0].prefetch(0, b(1)[0][0], 1)                             # one pane at a time
    assert e.value.code == _ffi.ERR_INVALID
    with pytest.raises(ValueError):
        ctx.feed_data_sources(b(1), pane=1)                                 # the prefetched pane is fed from its side buffers
    want0 = _q5_rows(ctx.execute()[0][0])
    ctx.clean_data_sources()
    ctx.feed_data_sources(None, pane=1)
    host = s.bids(0, 4_000)["auction"]
    oa, on = oracle.q5_hot_items(host)
    assert _q5_rows(ctx.execute()[0][0]) == sorted(zip(oa.tolist(), on.tolist())) and want0
    ctx.clean_data_sources()
    # NULLs (and Utf8 columns) are not prefetched: the pane is kept for the ordinary feed -- which, on q5's state ring, refuses these NULLs
    # and leaves the ring as it was
    nulls = pa.record_batch([pa.array([1, None, 3, 1, 1], pa.int32())], names=["auction"])
    ctx.prefetch_data_sources([[[nulls]]], pane=2)
    assert ctx._pre["moving"] == [] and len(ctx._pre["later"]) == 1
    with pytest.raises(FlockGpuError) as e:
        ctx.feed_data_sources(None, pane=2)
    assert e.value.code == _ffi.ERR_UNSUPPORTED and ctx.plans[0].ring_state() == (1, 1, 2)
    ctx.prefetch_data_sources(b(2), pane=2)
    ctx.feed_data_sources(None, pane=2)
    assert ctx.plans[0].ring_state() == (1, 2, 2)
    ctx.clean_data_sources()
    ctx.prefetch_data_sources(b(3), pane=3)
    ctx.close_window_ring()                                                 # a prefetch that was never fed is dropped with the ring
    ctx.close()


@pytest.mark.gpu
def test_two_relation_utf8_plan_through_the_rows_ring(gpu):
    """q8 (persons with a Utf8 column + auctions) over Hopping(3 panes): the generic rows ring keeps both relations' panes on the
    device (Utf8 offsets rebased when the oldest pane goes) and every window equals the whole-window feed and the oracle."""
    from flock_amd.runtime import ExecutionContext, collect
    eps, n_panes, ppw = 30_000, 7, 3
    s = oracle.NexmarkStream(seed=8, eps=eps)
    pers = [_person_batches(s, p * eps, (p + 1) * eps, eps) for p in range(n_panes)]
    aucs = [_auction_batches(s, p * eps, (p + 1) * eps, 11_000) for p in range(n_panes)]
    ring = ExecutionContext([_plan(8)], name="q8-ring", gpu=gpu)
    whole = ExecutionContext([_plan(8)], name="q8-whole", gpu=gpu)
    ring.open_window_ring(ppw)
    total = 0
    for p in range(n_panes):
        rb = collect(ring, [[pers[p]], [aucs[p]]], pane=p)[0][0]
        lo = max(0, p - ppw + 1)
        ref = collect(whole, [[[b for q in range(lo, p + 1) for b in pers[q]]], [[b for q in range(lo, p + 1) for b in aucs[q]]]])[0][0]
        hp, ha = s.persons(lo * eps, (p + 1) * eps), s.auctions(lo * eps, (p + 1) * eps)
        rows = oracle.q8_join(hp["p_id"], hp["name"], ha["seller"])
        names = _utf8(hp["name"]).to_pylist()
        want = sorted((int(hp["p_id"][r]), names[r]) for r in rows)
        got = sorted(zip(rb["p_id"].to_pylist(), rb["name"].to_pylist()))
        assert got == sorted(zip(ref["p_id"].to_pylist(), ref["name"].to_pylist())) == want, p
        total += len(want)
    assert total > 50
    ring.close()
    whole.close()


@pytest.mark.gpu
def test_large_batches_without_nulls_behind_a_small_one_with_nulls(gpu):
    """Validity is materialised from the first batch that holds a NULL on; batches fed afterwards without any NULL are valid by convention and
    filled in when the leaf is scanned -- for which the validity buffer must have room for every row of the leaf, not only for the rows
    of the last batch that held a NULL (it had not: a 20 000-row batch behind a 50-row one was an invalid memset)."""
    from flock_amd.runtime import ExecutionContext, collect
    aggs = [("count", "v", "UInt64"), ("max", "v", "Int64"), ("count", None, "UInt64")]
    small = _null_table(50, 41)
    r = np.random.default_rng(41)
    big = {"k": [int(x) for x in r.integers(-3, 6, 20_000)], "v": [int(x) for x in r.integers(-50, 50, 20_000)],
           "f": [float(x) for x in np.round(r.normal(0, 10, 20_000))], "s": ["s1"] * 20_000}
    ctx = ExecutionContext([_agg_plan(aggs)], gpu=gpu)
    rb = collect(ctx, [[_null_batches(small, 50) + _null_batches(big, 20_000) + _null_batches(big, 7_000)]])[0][0]
    ctx.close()
    t = {c: small[c] + big[c] + big[c] for c in ("k", "v", "f", "s")}
    want = g.hash_aggregate_exec(t, ["k"], [("%s(%s)" % (fn.upper(), col or "UInt8(1)"), fn, col) for fn, col, _ in aggs])
    key = lambda r: (r[0] is None, r[0] or 0)
    assert sorted(_pyrows(rb), key=key) == sorted(g.rows(want), key=key)


def _group_plan(key, aggs):
    """_agg_plan with the GROUP BY column as a parameter (Partial -> Hash([key]) -> FinalPartitioned)."""
    def expr(fn, col, dt):
        arg = _c(col) if col else {"physical_expr": "literal", "value": {"UInt8": 1}}
        return {"aggregate_expr": fn, "name": "%s(%s)" % (fn.upper(), col or "UInt8(1)"), "data_type": dt, "nullable": True, "expr": arg}
    ae = [expr(*a) for a in aggs]
    part = {"execution_plan": "hash_aggregate_exec", "mode": "Partial", "group_expr": [[_c(key), key]], "aggr_expr": ae, "input": _scan(),
            "input_schema": {"fields": _NF, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}
    rep = {"execution_plan": "repartition_exec", "input": part, "partitioning": {"Hash": [[{"physical_expr": "column", "name": key, "index": 0}], 4]}}
    return {"execution_plan": "hash_aggregate_exec", "mode": "FinalPartitioned", "group_expr": [[{"physical_expr": "column", "name": key, "index": 0}, key]],
            "aggr_expr": ae, "input": {"execution_plan": "coalesce_batches_exec", "input": rep, "target_batch_size": 4096},
            "input_schema": {"fields": _NF, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(20))
def test_group_by_at_random(gpu, seed):
    """The generic GROUP BY (wave reduction of a skewed key -> LDS table per workgroup -> global table sized from the node's last run) over
    seeded random tables: key column of every supported type, any set of aggregates that fits the four accumulators, NULLs from none to
    nearly all in keys and values, one hot key or none, from a dozen rows to past the size where the LDS level switches on, several
    executes per plan so that the table's sizing hint sees the groups shrink and multiply.  Values are integral, so sums and averages are
    exact whatever the order of the additions."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(1000 + seed)
    key = ["k", "v", "s"][seed % 3]
    pool = [("count", None, "UInt64"), ("count", "v" if key != "v" else "k", "UInt64"), ("sum", "v" if key != "v" else "k", "Int64"),
            ("max", "v" if key != "v" else "k", "Int64" if key != "v" else "Int32"), ("min", "f", "Float64"), ("max", "f", "Float64"), ("avg", "k" if key != "k" else "v", "Float64")]
    aggs, used = [], 0
    for i in r.permutation(len(pool)):
        cost = 2 if pool[i][0] == "avg" else 1
        if used + cost <= 4 and r.random() < 0.7:
            aggs.append(pool[i])
            used += cost
    if not aggs:
        aggs = [pool[0]]
    ctx = ExecutionContext([_group_plan(key, aggs)], gpu=gpu)
    for n in [int(x) for x in r.choice([12, 700, 40_000, 140_000], 3)]:
        n_keys = int(r.choice([3, 200, max(4, n // 2)]))
        null_p = float(r.choice([0.0, 0.1, 0.9]))
        hot = float(r.choice([0.0, 0.6]))
        kk = r.integers(-n_keys // 2, n_keys, n)
        kk[r.random(n) < hot] = 1
        nul = lambda col, p: [None if r.random() < p else x for x in col]
        t = {"k": nul([int(x) for x in (kk if key == "k" else r.integers(-40, 40, n))], null_p if key == "k" else null_p / 2),
             "v": nul([int(x) * (10**12 if key == "v" else 1) for x in (kk if key == "v" else r.integers(-10**6, 10**6, n))], null_p if key == "v" else null_p / 2),
             "f": nul([float(x) for x in np.round(r.normal(0, 100, n))], null_p / 2),
             "s": nul(["" if x == 0 else "key%d" % x for x in (kk if key == "s" else r.integers(0, 5, n))], null_p if key == "s" else 0.0)}
        rb = collect(ctx, [[_null_batches(t, int(r.choice([n, max(1, n // 3)])))]])[0][0]
        want = g.hash_aggregate_exec(t, [key], [("%s(%s)" % (fn.upper(), col or "UInt8(1)"), fn, col) for fn, col, _ in aggs])
        order = lambda row: (row[0] is None, row[0] if row[0] is not None else 0) if key != "s" else (row[0] is None, row[0] or "")
        assert sorted(_pyrows(rb), key=order) == sorted(g.rows(want), key=order), (seed, key, aggs, n, n_keys, null_p, hot)
    ctx.close()


@pytest.mark.gpu
def test_rows_ring_carries_validity(gpu):
    """A GROUP BY with NULL keys and NULL values over Hopping(3 panes) through the rows ring: the validity bytes of the held panes stay with
    their rows when the oldest pane goes (some panes hold no NULL at all: their rows are valid by convention, before and after the shift);
    every window equals the oracle over the rows of the panes it holds."""
    from flock_amd.runtime import ExecutionContext, collect
    aggs = [("count", "v", "UInt64"), ("max", "v", "Int64"), ("min", "f", "Float64"), ("count", None, "UInt64")]
    ring = ExecutionContext([_agg_plan(aggs)], name="agg-ring", gpu=gpu)
    ring.open_window_ring(3)
    panes = []
    for p in range(7):
        t = _null_table(400 + 37 * p, 100 + p)
        if p in (2, 5):                                                # panes without a single NULL
            r = np.random.default_rng(p)
            t = {"k": [int(x) for x in r.integers(-3, 6, 300)], "v": [int(x) for x in r.integers(-50, 50, 300)],
                 "f": [float(x) for x in np.round(r.normal(0, 10, 300))], "s": ["s1"] * 300}
        panes.append(t)
        rb = collect(ring, [[_null_batches(t, 150)]], pane=p)[0][0]
        held = panes[max(0, p - 2): p + 1]
        window = {c: [x for q in held for x in q[c]] for c in ("k", "v", "f", "s")}
        want = g.hash_aggregate_exec(window, ["k"], [("%s(%s)" % (fn.upper(), col or "UInt8(1)"), fn, col) for fn, col, _ in aggs])
        key = lambda r: (r[0] is None, r[0] or 0)
        assert sorted(_pyrows(rb), key=key) == sorted(g.rows(want), key=key), p
    ring.close()


@pytest.mark.gpu
def test_prefetched_pane_behind_a_pane_with_nulls_on_the_rows_ring(gpu):
    """ADVICE r4 (plan.hip, feed_pane's prefetch append): a small pane WITH NULLs fed the ordinary way leaves validity bytes on its columns;
    the next, much larger pane holds no NULL and arrives by prefetch -- its rows are appended device to device, and the scan then fills
    "valid" up to the leaf's row count.  The validity buffer has to have grown with the values (it used to be written past its end)."""
    from flock_amd.runtime import ExecutionContext
    aggs = [("count", "v", "UInt64"), ("max", "v", "Int64"), ("count", None, "UInt64")]
    ring = ExecutionContext([_agg_plan(aggs)], name="agg-ring-pre", gpu=gpu)
    ring.open_window_ring(2)
    panes = [_null_table(50, 7)]
    r = np.random.default_rng(8)
    for n in (20_000, 45_000):
        panes.append({"k": [int(x) for x in r.integers(-3, 6, n)], "v": [int(x) for x in r.integers(-50, 50, n)],
                      "f": [float(x) for x in np.round(r.normal(0, 10, n))], "s": ["s1"] * n})
    ring.feed_data_sources([[_null_batches(panes[0], 50)]], pane=0)
    for p in range(len(panes)):
        if p + 1 < len(panes):
            ring.prefetch_data_sources([[_null_batches(panes[p + 1], 7_000)]], pane=p + 1)
        rb = ring.execute()[0][0]
        ring.clean_data_sources()
        held = panes[max(0, p - 1): p + 1]
        window = {c: [x for q in held for x in q[c]] for c in ("k", "v", "f", "s")}
        want = g.hash_aggregate_exec(window, ["k"], [("%s(%s)" % (fn.upper(), col or "UInt8(1)"), fn, col) for fn, col, _ in aggs])
        key = lambda r: (r[0] is None, r[0] or 0)
        assert sorted(_pyrows(rb), key=key) == sorted(g.rows(want), key=key), p
        if p + 1 < len(panes):
            ring.feed_data_sources(None, pane=p + 1)
    ring.close()


# ------------------------------------------------------------------ GPU: asynchronous execute
@pytest.mark.gpu
def test_plans_on_their_own_contexts_execute_side_by_side(gpu):
    """context.rs:172-191: every plan of a function runs on its own task.  Two plans (q5's two halves would be; here q5 and q2 over the
    same bids) on two GpuContexts: execute() starts both, then joins them -- same batches as one after the other."""
    from flock_amd import GpuContext
    from flock_amd.runtime import ExecutionContext
    s = oracle.NexmarkStream(seed=5, eps=50_000)
    bids = _bid_batches(s, 0, 200_000, 40_000)
    g2 = GpuContext(0, own_stream=True)
    p2 = json.loads(_plan(2))
    p2["input"]["input"]["predicate"]["left"]["right"]["value"] = {"Int64": 7}      # auction % 7 = 0: rows at any stream size
    both = ExecutionContext([p2, _plan(5)], name="two", gpus=[g2, gpu])
    serial = ExecutionContext([p2, _plan(5)], name="two-serial", gpu=gpu)
    for ctx in (both, serial):
        ctx.feed_data_sources([[bids], [bids]])      # (q2's leaf takes the first source, q5's first `bid` leaf the second)
    a, b = both.execute(), serial.execute()
    assert both._concurrent() and not serial._concurrent()
    assert _q5_rows(a[1][0]) == _q5_rows(b[1][0]) and a[0][0].equals(b[0][0]) and a[0][0].num_rows > 0
    # wait without a started call, and a second start while one is in flight, are argument errors
    from flock_amd import FlockGpuError, _ffi
    with pytest.raises(FlockGpuError) as e:
        both.plans[0].wait()
    assert e.value.code == _ffi.ERR_INVALID
    both.plans[1].execute_async()
    with pytest.raises(FlockGpuError):
        both.plans[1].execute_async()
    with pytest.raises(FlockGpuError):
        both.plans[1].execute()
    assert _q5_rows(both.plans[1].wait()) == _q5_rows(b[1][0])
    both.close()
    serial.close()
    g2.close()


@pytest.mark.gpu
def test_batched_calls_async_equal_sync(gpu):
    """flockgpu_q{3,5,8}_*_async + flockgpu_ctx_wait: two contexts with one call each in flight give what the synchronous calls give."""
    from flock_amd import GpuContext, NEXMarkSource, Window, run_query
    g2 = GpuContext(0, own_stream=True)
    src = NEXMarkSource(20, 30_000, Window.hopping(10, 5), seed=12)
    data = src.generate_data(gpu)
    sched = data.window_schedule("bid")
    want = gpu.q5_hot_items(data.bids, sched).to_host()
    p1 = gpu.q5_hot_items_async(data.bids, sched)
    p2 = g2.q5_hot_items_async(data.bids, sched)
    for got in (p1.wait().to_host(), p2.wait().to_host()):
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
    sa, sp = data.window_schedule("auction", Window.element_wise()), data.window_schedule("person", Window.element_wise())
    w3 = gpu.q3_join(data.auctions, sa, data.persons, sp).to_host()
    p1 = gpu.q3_join_async(data.auctions, sa, data.persons, sp)
    p2 = g2.q3_join_async(data.auctions, sa, data.persons, sp)
    for got in (p1.wait().to_host(), p2.wait().to_host()):
        assert all(np.array_equal(got[k], w3[k]) for k in ("a_id", "auction_row", "person_row", "offsets")) and len(w3["a_id"]) > 0
    ta, tp = data.window_schedule("auction", Window.tumbling(10)), data.window_schedule("person", Window.tumbling(10))
    w8 = gpu.q8_join(data.persons, tp, data.auctions, ta).to_host()
    p2 = g2.q8_join_async(data.persons, tp, data.auctions, ta)
    got = p2.wait().to_host()
    assert all(np.array_equal(got[k], w8[k]) for k in ("p_id", "offsets")) and len(w8["p_id"]) > 0
    g2.close()


# ------------------------------------------------------------------ GPU: hash placement of shuffling stages
@pytest.mark.gpu
def test_mixed_hash_placement_is_rejected_not_silently_wrong(gpu):
    """A join stage whose two inputs were placed by DIFFERENT hashes -- the auctions by libflockgpu's stage 0, the persons by another
    engine (the oracle's repartition_hash, standing in for DataFusion's ahash) -- would meet only part of its pairs in each partition.
    The library's partitions carry their scheme as Arrow schema metadata and the consuming stage refuses the mix at feed time."""
    from flock_amd import FlockGpuError, _ffi
    from flock_amd.runtime import ExecutionContext, collect, partition_scheme
    from flock_amd.stages import build_query_dag
    s = oracle.NexmarkStream(seed=19, eps=60_000)
    n = 120_000
    stages = build_query_dag(json.loads(_plan(3)))
    shuffling = [st for st in stages if st.is_shuffling]
    join = stages[-1]
    outs = []
    for st in shuffling:
        ctx = ExecutionContext([st.plan], gpu=gpu)
        rel = _person_batches(s, 0, n, n) if "person" in ctx.plans[0].inputs else _auction_batches(s, 0, n, n)
        outs.append(collect(ctx, [[rel]]))                                   # [partition][batch]
        ctx.close()
    tag = outs[0][0][0].schema.metadata
    assert tag and tag[b"flockgpu.partition_scheme"].decode() == partition_scheme()
    P = len(outs[0])
    jc = ExecutionContext([join.plan], gpu=gpu)
    # all GPU-placed: partition p of both relations joins; the union over p is the whole join
    total = 0
    for p in range(P):
        total += collect(jc, [[outs[0][p]], [outs[1][p]]])[0][0].num_rows
    a, pe = s.auctions(0, n), s.persons(0, n)
    ar, _ = oracle.q3_join(a["seller"], a["category"], pe["p_id"], pe["state"])
    assert total == len(ar) > 0
    # one side re-placed by another engine's hash (untagged batches): refused, nothing joined
    other = outs[1][0][0].replace_schema_metadata(None)
    with pytest.raises(FlockGpuError) as e:
        jc.feed_data_sources([[outs[0][0]], [[other]]])
    assert e.value.code == _ffi.ERR_INVALID and "placement" in str(e.value)
    jc.clean_data_sources()
    # every producer on the other engine (all untagged) is consistent again -- and a foreign tag is named
    untagged = [[b.replace_schema_metadata(None) for b in part] for part in (outs[0][0], outs[1][0])]
    jc.feed_data_sources([[untagged[0]], [untagged[1]]])
    jc.clean_data_sources()
    foreign = outs[0][0][0].replace_schema_metadata({"flockgpu.partition_scheme": "datafusion/ahash-0000"})
    with pytest.raises(FlockGpuError) as e:
        jc.feed_data_sources([[[foreign]]])
    assert "datafusion/ahash-0000" in str(e.value)
    jc.close()


# ------------------------------------------------------------------ GPU: NULLs (validity bitmaps in, validity bitmaps out)
_NF = [_field("k", "Int32", True), _field("v", "Int64", True), _field("f", "Float64", True), _field("s", "Utf8", True)]


def _scan(fields=None):
    fields = fields or _NF
    return {"execution_plan": "memory_exec", "schema": {"fields": fields, "metadata": {}}, "projection": list(range(len(fields)))}


def _c(name, fields=None):
    names = [f["name"] for f in (fields or _NF)]
    return {"physical_expr": "column", "name": name, "index": names.index(name)}


def _null_table(n, seed, null_every=(3, 4, 5, 7)):
    r = np.random.default_rng(seed)
    k = [None if i % null_every[0] == 1 else int(x) for i, x in enumerate(r.integers(-3, 6, n))]
    v = [None if i % null_every[1] == 2 else int(x) for i, x in enumerate(r.integers(-50, 50, n))]
    f = [None if i % null_every[2] == 0 else float(x) for i, x in enumerate(np.round(r.normal(0, 10, n)))]
    s = [None if i % null_every[3] == 3 else "s%d" % x for i, x in enumerate(r.integers(0, 9, n))]
    if n > 20:
        for i in range(n):                      # key 5: every value NULL -> MAX / MIN / SUM / AVG of that group are NULL, COUNT(v) is 0
            if k[i] == 5:
                v[i], f[i] = None, None
    return {"k": k, "v": v, "f": f, "s": s}


def _null_batches(t, chunk):
    n = len(t["k"])
    out = []
    for a in range(0, n, chunk):
        out.append(pa.record_batch([pa.array(t["k"][a:a + chunk], pa.int32()), pa.array(t["v"][a:a + chunk], pa.int64()),
                                    pa.array(t["f"][a:a + chunk], pa.float64()), pa.array(t["s"][a:a + chunk], pa.string())], names=["k", "v", "f", "s"]))
    return out


def _agg_plan(aggs):
    """Partial -> Hash([k]) -> FinalPartitioned GROUP BY k with `aggs` = [(fn, column or None, data_type)], as DataFusion plans a GROUP BY."""
    def expr(fn, col, dt):
        arg = _c(col) if col else {"physical_expr": "literal", "value": {"UInt8": 1}}
        name = "%s(%s)" % (fn.upper(), col or "UInt8(1)")
        return {"aggregate_expr": fn, "name": name, "data_type": dt, "nullable": True, "expr": arg}
    ae = [expr(*a) for a in aggs]
    part = {"execution_plan": "hash_aggregate_exec", "mode": "Partial", "group_expr": [[_c("k"), "k"]], "aggr_expr": ae, "input": _scan(),
            "input_schema": {"fields": _NF, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}
    rep = {"execution_plan": "repartition_exec", "input": part, "partitioning": {"Hash": [[{"physical_expr": "column", "name": "k", "index": 0}], 4]}}
    return {"execution_plan": "hash_aggregate_exec", "mode": "FinalPartitioned", "group_expr": [[{"physical_expr": "column", "name": "k", "index": 0}, "k"]],
            "aggr_expr": ae, "input": {"execution_plan": "coalesce_batches_exec", "input": rep, "target_batch_size": 4096},
            "input_schema": {"fields": _NF, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}


def _pyrows(rb):
    return list(zip(*[rb[c].to_pylist() for c in rb.schema.names]))


@pytest.mark.gpu
@pytest.mark.parametrize("n,chunk", [(40, 40), (5_000, 1_300), (60_000, 60_000)])
def test_aggregates_skip_nulls_and_group_null_keys(gpu, n, chunk):
    """COUNT(col) counts the non-NULL values, MIN / MAX / SUM / AVG skip NULLs and are NULL over nothing but NULLs,
    COUNT(*) counts rows, NULL group keys form one group -- validity bitmaps in (only some batches hold NULLs), validity bitmaps out."""
    from flock_amd.runtime import ExecutionContext, collect
    t = _null_table(n, 7)
    for aggs in ([("count", "v", "UInt64"), ("max", "v", "Int64"), ("min", "f", "Float64"), ("count", None, "UInt64")],
                 [("avg", "v", "Float64"), ("sum", "v", "Int64")]):
        ctx = ExecutionContext([_agg_plan(aggs)], gpu=gpu)
        rb = collect(ctx, [[_null_batches(t, chunk)]])[0][0]
        ctx.close()
        want = g.hash_aggregate_exec(t, ["k"], [("%s(%s)" % (fn.upper(), col or "UInt8(1)"), fn, col) for fn, col, _ in aggs])
        key = lambda r: (r[0] is None, r[0])
        assert sorted(_pyrows(rb), key=key) == sorted(g.rows(want), key=key)
        assert rb.schema.names == list(want)
        if n > 20:
            assert any(r[0] is None for r in _pyrows(rb)) and any(r[1] is None or r[1] == 0 for r in _pyrows(rb))


@pytest.mark.gpu
@pytest.mark.parametrize("n,chunk", [(60, 60), (9_000, 2_100)])
def test_null_utf8_group_keys_form_one_group(gpu, n, chunk):
    """GROUP BY a Utf8 column that holds NULLs: the NULLs are ONE group -- not the empty string's, which the table also holds -- and the
    group's key comes back NULL."""
    from flock_amd.runtime import ExecutionContext, collect
    t = _null_table(n, 19)
    t["s"] = [None if s is None else ("" if s == "s3" else s) for s in t["s"]]          # real empty strings next to the NULLs
    aggs = [("count", None, "UInt64"), ("max", "v", "Int64"), ("count", "v", "UInt64")]
    def expr(fn, col, dt):
        arg = _c(col) if col else {"physical_expr": "literal", "value": {"UInt8": 1}}
        return {"aggregate_expr": fn, "name": "%s(%s)" % (fn.upper(), col or "UInt8(1)"), "data_type": dt, "nullable": True, "expr": arg}
    ae = [expr(*a) for a in aggs]
    part = {"execution_plan": "hash_aggregate_exec", "mode": "Partial", "group_expr": [[_c("s"), "s"]], "aggr_expr": ae, "input": _scan(),
            "input_schema": {"fields": _NF, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}
    rep = {"execution_plan": "repartition_exec", "input": part, "partitioning": {"Hash": [[{"physical_expr": "column", "name": "s", "index": 0}], 4]}}
    plan = {"execution_plan": "hash_aggregate_exec", "mode": "FinalPartitioned", "group_expr": [[{"physical_expr": "column", "name": "s", "index": 0}, "s"]],
            "aggr_expr": ae, "input": {"execution_plan": "coalesce_batches_exec", "input": rep, "target_batch_size": 4096},
            "input_schema": {"fields": _NF, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}
    ctx = ExecutionContext([plan], gpu=gpu)
    rb = collect(ctx, [[_null_batches(t, chunk)]])[0][0]
    ctx.close()
    want = g.hash_aggregate_exec(t, ["s"], [("%s(%s)" % (fn.upper(), col or "UInt8(1)"), fn, col) for fn, col, _ in aggs])
    key = lambda r: (r[0] is None, r[0] or "")
    assert sorted(_pyrows(rb), key=key) == sorted(g.rows(want), key=key)
    assert sum(r[0] is None for r in _pyrows(rb)) == 1 and any(r[0] == "" for r in _pyrows(rb))


@pytest.mark.gpu
@pytest.mark.parametrize("n,chunk", [(50, 50), (7_000, 1_900), (80_000, 80_000)])
def test_null_int64_group_keys_form_one_group(gpu, n, chunk):
    """GROUP BY an Int64 column that holds NULLs: no 64-bit value is free to stand for NULL, so the GROUP BY keeps the NULL keys in a slot
    of their own (relops.hip: key validity into group_by_key64_n) -- next to the key INT64_MIN, which has its own, too."""
    from flock_amd.runtime import ExecutionContext, collect
    t = _null_table(n, 29)
    t["v"] = [None if v is None else (-2**63 if v == 7 else (2**63 - 1 if v == 8 else v % 11)) for v in t["v"]]     # few groups, both extremes among them
    aggs = [("count", None, "UInt64"), ("max", "k", "Int32"), ("count", "k", "UInt64"), ("min", "f", "Float64")]
    def expr(fn, col, dt):
        arg = _c(col) if col else {"physical_expr": "literal", "value": {"UInt8": 1}}
        return {"aggregate_expr": fn, "name": "%s(%s)" % (fn.upper(), col or "UInt8(1)"), "data_type": dt, "nullable": True, "expr": arg}
    ae = [expr(*a) for a in aggs]
    part = {"execution_plan": "hash_aggregate_exec", "mode": "Partial", "group_expr": [[_c("v"), "v"]], "aggr_expr": ae, "input": _scan(),
            "input_schema": {"fields": _NF, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}
    rep = {"execution_plan": "repartition_exec", "input": part, "partitioning": {"Hash": [[{"physical_expr": "column", "name": "v", "index": 0}], 4]}}
    plan = {"execution_plan": "hash_aggregate_exec", "mode": "FinalPartitioned", "group_expr": [[{"physical_expr": "column", "name": "v", "index": 0}, "v"]],
            "aggr_expr": ae, "input": {"execution_plan": "coalesce_batches_exec", "input": rep, "target_batch_size": 4096},
            "input_schema": {"fields": _NF, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}
    ctx = ExecutionContext([plan], gpu=gpu)
    rb = collect(ctx, [[_null_batches(t, chunk)]])[0][0]
    ctx.close()
    want = g.hash_aggregate_exec(t, ["v"], [("%s(%s)" % (fn.upper(), col or "UInt8(1)"), fn, col) for fn, col, _ in aggs])
    key = lambda r: (r[0] is None, r[0] or 0)
    assert sorted(_pyrows(rb), key=key) == sorted(g.rows(want), key=key)
    assert sum(r[0] is None for r in _pyrows(rb)) == 1
    if n > 1000:
        assert any(r[0] == -2**63 for r in _pyrows(rb)) and any(r[0] == 2**63 - 1 for r in _pyrows(rb))


@pytest.mark.gpu
def test_group_table_sized_from_the_last_call_regrows_when_the_groups_multiply(gpu):
    """The generic GROUP BY sizes its table for three slots per group of the plan's PREVIOUS execute (relops.hip group_by_key64_n): the same
    plan sees 7 groups, then 90 000 (the hinted table overflows; the pass is repeated with more slots, twice), then 7 again, then skewed
    keys (most rows share one key: the wave-level reduction of that key) -- every answer equals the oracle."""
    from flock_amd.runtime import ExecutionContext, collect
    aggs = [("count", None, "UInt64"), ("sum", "v", "Int64"), ("max", "v", "Int64"), ("min", "f", "Float64")]
    ctx = ExecutionContext([_agg_plan(aggs)], gpu=gpu)
    r = np.random.default_rng(3)
    for n, n_keys, hot in ((3_000, 7, 0.0), (200_000, 90_000, 0.0), (3_000, 7, 0.0), (150_000, 5_000, 0.7), (150_000, 5_000, 0.0)):
        k = r.integers(0, n_keys, n)
        k[r.random(n) < hot] = 42
        t = {"k": [int(x) for x in k], "v": [int(x) for x in r.integers(-1000, 1000, n)], "f": [float(x) for x in np.round(r.normal(0, 50, n))], "s": ["x"] * n}
        rb = collect(ctx, [[_null_batches(t, 70_000)]])[0][0]
        want = g.hash_aggregate_exec(t, ["k"], [("%s(%s)" % (fn.upper(), col or "UInt8(1)"), fn, col) for fn, col, _ in aggs])
        assert sorted(_pyrows(rb)) == sorted(g.rows(want)), (n, n_keys, hot)
        assert rb.num_rows == len(set(t["k"]))
    ctx.close()


@pytest.mark.gpu
def test_filter_projection_and_sort_carry_nulls(gpu):
    """A NULL comparison keeps no row -- under OR, too (NULL OR TRUE is TRUE) --, NULLs in projected columns come back as NULLs, and
    ORDER BY places them where the plan's SortOptions say."""
    from flock_amd.runtime import ExecutionContext, collect
    t = _null_table(3_000, 11)
    lit = lambda kind, x: {"physical_expr": "literal", "value": {kind: x}}
    cast = lambda e, ty: {"physical_expr": "cast_expr", "expr": e, "cast_type": ty}
    pred = {"physical_expr": "binary_expr", "op": "Or",
            "left": {"physical_expr": "binary_expr", "op": "Lt", "left": _c("v"), "right": lit("Int64", 5)},
            "right": {"physical_expr": "binary_expr", "op": "Gt", "left": _c("f"), "right": lit("Float64", 0.5)}}
    filt = {"execution_plan": "filter_exec", "predicate": pred, "input": _scan()}
    proj = {"execution_plan": "projection_exec", "input": filt, "schema": {"fields": [_NF[3], _NF[1], _NF[0]], "metadata": {}},
            "expr": [[_c("s"), "s"], [_c("v"), "v"], [_c("k"), "k"]]}
    for desc, nulls_first in ((False, False), (False, True), (True, True), (True, False)):
        plan = {"execution_plan": "sort_exec", "input": proj,
                "expr": [{"expr": {"physical_expr": "column", "name": "v", "index": 1}, "options": {"descending": desc, "nulls_first": nulls_first}},
                         {"expr": {"physical_expr": "column", "name": "s", "index": 0}, "options": {"descending": False, "nulls_first": False}}]}
        ctx = ExecutionContext([plan], gpu=gpu)
        rb = collect(ctx, [[_null_batches(t, 700)]])[0][0]
        ctx.close()
        kept = g.filter_exec(t, lambda r: (None if r["v"] is None else r["v"] < 5) or (None if r["f"] is None else r["f"] > 0.5) or None)
        # (python's `or` over None: None or True -> True, None or False -> False/None -> not True: dropped, as SQL's three-valued OR)
        want = g.sort_exec({"s": kept["s"], "v": kept["v"], "k": kept["k"]}, [("v", desc, nulls_first), ("s", False, False)])
        assert _pyrows(rb) == g.rows(want) and rb.num_rows > 500
        assert rb["v"].null_count > 0 and rb["s"].null_count > 0 and rb["k"].null_count > 0


@pytest.mark.gpu
def test_join_with_null_keys_and_null_payloads(gpu):
    """Inner hash join: NULL keys never match (their rows are left out at feed), NULLs in the other columns travel to the output."""
    from flock_amd.runtime import ExecutionContext, collect
    left, right = _null_table(2_000, 3), _null_table(1_500, 4, null_every=(5, 3, 4, 2))
    rf = [_field("k2", "Int32", True), _field("v2", "Int64", True), _field("f2", "Float64", True), _field("s2", "Utf8", True)]
    side = lambda scan, key, fields: {"execution_plan": "coalesce_batches_exec", "target_batch_size": 4096,
                                      "input": {"execution_plan": "repartition_exec", "input": scan, "partitioning": {"Hash": [[_c(key, fields)], 4]}}}
    plan = {"execution_plan": "hash_join_exec", "left": side(_scan(), "k", _NF), "right": side(_scan(rf), "k2", rf), "join_type": "Inner", "mode": "Partitioned",
            "on": [[_c("k"), _c("k2", rf)]], "schema": {"fields": _NF + rf, "metadata": {}}}
    rbs = [pa.record_batch([b[c] for c in b.schema.names], names=["k2", "v2", "f2", "s2"]) for b in _null_batches(right, 400)]
    ctx = ExecutionContext([plan], gpu=gpu)
    rb = collect(ctx, [[_null_batches(left, 900)], [rbs]])[0][0]
    ctx.close()
    want = g.hash_join_inner(left, {"k2": right["k"], "v2": right["v"], "f2": right["f"], "s2": right["s"]}, [("k", "k2")])
    key = lambda r: tuple((x is None, x) for x in r)
    assert sorted(_pyrows(rb), key=key) == sorted(g.rows(want), key=key) and rb.num_rows > 1000
    assert rb["v"].null_count > 0 and rb["s"].null_count > 0 and rb["f2"].null_count > 0 and rb["k"].null_count == 0 and rb["k2"].null_count == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_inner_join_at_random(gpu, seed):
    """The generic inner hash join over seeded random tables: Int32 or Utf8 keys, duplicates on both sides (every pair comes out), NULL keys
    (never match), NULLs in the columns carried along, empty and one-sided inputs, a few hundred to tens of thousands of rows, the same
    plan executed three times with different shapes."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(2000 + seed)
    on = "k" if seed % 2 == 0 else "s"
    rf = [_field("k2", "Int32", True), _field("v2", "Int64", True), _field("f2", "Float64", True), _field("s2", "Utf8", True)]
    side = lambda scan, key, fields: {"execution_plan": "coalesce_batches_exec", "target_batch_size": 4096,
                                      "input": {"execution_plan": "repartition_exec", "input": scan, "partitioning": {"Hash": [[_c(key, fields)], 4]}}}
    plan = {"execution_plan": "hash_join_exec", "left": side(_scan(), on, _NF), "right": side(_scan(rf), on + "2", rf), "join_type": "Inner", "mode": "Partitioned",
            "on": [[_c(on), _c(on + "2", rf)]], "schema": {"fields": _NF + rf, "metadata": {}}}
    ctx = ExecutionContext([plan], gpu=gpu)

    def table(n, n_keys, null_p):
        nul = lambda col, p: [None if r.random() < p else x for x in col]
        kk = r.integers(0, max(n_keys, 1), n)
        return {"k": nul([int(x) for x in kk], null_p), "v": nul([int(x) for x in r.integers(-10**9, 10**9, n)], null_p / 2),
                "f": nul([float(x) for x in np.round(r.normal(0, 100, n))], null_p / 2),
                "s": nul(["" if x == 0 else "name-%d" % x for x in (kk if on == "s" else r.integers(0, 7, n))], null_p if on == "s" else null_p / 2)}
    for _ in range(3):
        nl, nr = int(r.choice([0, 300, 4_000, 30_000])), int(r.choice([1, 500, 9_000]))
        n_keys = int(r.choice([5, 400, 20_000]))
        if nl * nr // max(n_keys, 1) > 3_000_000:      # (keep the oracle's nested loops in seconds)
            n_keys = 20_000
        left, right = table(nl, n_keys, float(r.choice([0.0, 0.2]))), table(nr, n_keys, float(r.choice([0.0, 0.3])))
        lb = _null_batches(left, max(1, nl // 2)) if nl else [pa.record_batch([pa.array([], pa.int32()), pa.array([], pa.int64()), pa.array([], pa.float64()), pa.array([], pa.string())], names=["k", "v", "f", "s"])]
        rbs = [pa.record_batch([b[c] for c in b.schema.names], names=["k2", "v2", "f2", "s2"]) for b in _null_batches(right, max(1, nr // 3))]
        rb = collect(ctx, [[lb], [rbs]])[0][0]
        want = g.hash_join_inner(left, {"k2": right["k"], "v2": right["v"], "f2": right["f"], "s2": right["s"]}, [(on, on + "2")])
        key = lambda row: tuple((x is None, x if x is not None else 0) if not isinstance(x, str) else (False, x) for x in row)
        assert sorted(_pyrows(rb), key=key) == sorted(g.rows(want), key=key), (seed, on, nl, nr, n_keys)
    ctx.close()

