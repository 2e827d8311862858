"""Round-6 additions to the plan path (include/flockgpu_plan.h), each against the oracle:
  * comparisons whose right side is not a literal (`a % 3 = b`, `a = b % 3`, `a % 2 = b % 2`): the one-pass predicate program has no leaf
    for them and must hand them to the general evaluator instead of comparing against a literal that is not there;
  * the dense GROUP BY with four non-COUNT accumulators (72 KB of LDS per workgroup: above the 64 KB a launch gets without asking)."""
import json

import numpy as np
import pytest

from oracle import generic_ops as g
from test_plan_round5 import F, NAMES, _agg_plan, batches, binary, cast, col, lit, pyrows, scan, table
from test_plan_round5b import TYPES, norm


@pytest.fixture(scope="module")
def gpu():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _mod(e, m, ty="Int64"):
    return binary(e, "Modulo", lit(ty, m))


@pytest.mark.gpu
def test_a_remainder_or_a_column_against_a_non_literal_goes_to_the_general_evaluator(gpu):
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(606)
    t = table(9000, r, null_p=0.1)
    i64, j64 = cast(col("i"), "Int64"), cast(col("j"), "Int64")
    preds = [binary(_mod(j64, 7), "Eq", i64),                                   # a % 7 = b   (i is in [-40, 400))
             binary(i64, "Eq", _mod(j64, 401)),                                 # a = b % 401   (i is in [-40, 400))
             binary(_mod(i64, 2), "Eq", _mod(j64, 2)),                          # a % 2 = b % 2
             binary(_mod(col("i"), 5, "Int32"), "Lt", col("i")),                # a % 5 < a
             binary(col("l"), "GtEq", _mod(col("l"), 1000)),                    # l >= l % 1000
             binary(binary(_mod(i64, 3), "Eq", _mod(j64, 3)), "And", binary(col("i"), "Gt", lit("Int32", 10)))]
    for pred in preds:
        ctx = ExecutionContext([{"execution_plan": "filter_exec", "predicate": pred, "input": scan()}], gpu=gpu)
        gpu.profile_reset()
        gpu.profile(True)
        try:
            rb = collect(ctx, [[batches(t, 3000)]])[0][0]
            ran = gpu.profile_read()
        finally:
            gpu.profile(False)
            ctx.close()
        want = g.rows(g.filter_by_typed_expr(t, pred, TYPES))
        assert norm(pyrows(rb)) == norm(want) and 0 < len(want) < 9000, json.dumps(pred)
        assert "valprog_kernel" in ran and "pred_flag_kernel" not in ran, (json.dumps(pred), sorted(ran))


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["j", "l"])
def test_dense_group_by_with_four_accumulators(gpu, key):
    """SUM / MIN / MAX / SUM over a dense key: four 8-byte accumulator planes + the row counts = 72 KB of LDS per workgroup."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(77)
    n = 60_000
    t = table(n, r, null_p=0.0)
    t[key] = [int(x) for x in (1000 + np.sort(r.integers(0, 5000, n)))]
    other = "l" if key == "j" else "j"
    aggs = [("sum", "i", "Int64"), ("min", other, "Int64" if other == "l" else "Int32"), ("max", "i", "Int32"), ("sum", other, "Int64")]   # (kMaxGroupAggs = 4 accumulators per GROUP BY)
    plan = _agg_plan(key, aggs)
    ctx = ExecutionContext([plan], gpu=gpu)
    gpu.profile_reset()
    gpu.profile(True)
    try:
        rb = collect(ctx, [[batches(t, 20_000)]])[0][0]
        ran = gpu.profile_read()
    finally:
        gpu.profile(False)
        ctx.close()
    assert "dense_group_kernel" in ran, sorted(ran)
    want = g.hash_aggregate_exec(t, [key], [("%s(%s)" % (fn.upper(), c or "UInt8(1)"), fn, c) for fn, c, _ in aggs])
    assert sorted(pyrows(rb)) == sorted(g.rows(want))


@pytest.mark.gpu
def test_division_and_remainder_by_literals_at_the_edges(gpu):
    """Round 6: `x / c` and `x % c` with a literal c run as a multiply-high by a host-made reciprocal (valprog.hpp, ValBuilder::fuse_immediate).
    Every divisor class -- powers of two, the 64- and 65-bit multiplier forms, negative divisors, INT_MIN as the divisor, UInt64 divisors above
    2^63 -- over dividends at the edges of their types and around multiples of the divisor, against the oracle's truncating division."""
    from flock_amd.runtime import ExecutionContext, collect
    from test_plan_round5b import projection
    r = np.random.default_rng(64)
    n = 4096
    t = table(n, r, null_p=0.1)
    div32 = [1, 2, 3, 7, -7, 10, 100, 123, 641, 65_536, 65_537, 2**31 - 1, -2**31, -3]
    div64 = [1, 2, 3, 7, -7, 10, 1000, 2**32, 2**32 + 1, 2**33 + 9, 2**62, 2**62 + 1, 2**63 - 1, -2**63, -1_000_003, 6_700_417]
    divu = [1, 2, 3, 10, 2**32 + 1, 2**63, 2**63 + 5, 2**64 - 1, 2**64 - 59]
    edge32 = [0, 1, -1, 2**31 - 1, -2**31 + 1, -2**31, 65_535, 65_536, -65_537]
    edge64 = [0, 1, -1, 2**63 - 1, -2**63 + 1, -2**63, 2**32, -2**32 - 1, 2**62, 2**62 + 1]
    edgeu = [0, 1, 2**63 - 1, 2**63, 2**63 + 5, 2**64 - 1, 2**64 - 60, 2**32]
    for k, d in enumerate(div32):
        for m in (-2, -1, 0, 1, 2, 1000):
            edge32.append(max(-2**31 + 1, min(2**31 - 1, m * d + int(r.integers(-1, 2)))))
    for k, d in enumerate(div64):
        for m in (-3, -1, 0, 1, 2, 77):
            edge64.append(max(-2**63 + 1, min(2**63 - 1, m * d + int(r.integers(-1, 2)))))
    t["j"][:len(edge32)] = edge32
    t["l"][:len(edge64)] = edge64
    t["u"][:len(edgeu)] = edgeu
    exprs = []
    for d in div32:
        if d != -1:
            exprs += [(binary(col("j"), "Divide", lit("Int32", d)), "q32_%d" % d), (binary(col("j"), "Modulo", lit("Int32", d)), "r32_%d" % d)]
    for d in div64:
        exprs += [(binary(col("l"), "Divide", lit("Int64", d)), "q64_%d" % d), (binary(col("l"), "Modulo", lit("Int64", d)), "r64_%d" % d)]
    for d in divu:
        exprs += [(binary(col("u"), "Divide", lit("UInt64", d)), "qu_%d" % d), (binary(col("u"), "Modulo", lit("UInt64", d)), "ru_%d" % d)]
    # -2^31 / -2^31 and -2^63 / -2^63 are 1, no overflow; only a divisor of -1 overflows (tests/test_plan_round5b.py)
    for lo in range(0, len(exprs), 6):   # (a projection per six expressions: every one its own program)
        part = exprs[lo:lo + 6]
        ctx = ExecutionContext([projection(part)], gpu=gpu)
        try:
            rb = collect(ctx, [[batches(t, n)]])[0][0]
        finally:
            ctx.close()
        assert norm(pyrows(rb)) == norm(g.rows(g.project_typed(t, part, TYPES))), [nm for _, nm in part]


def _hash_diff_plan(key, n, below=None):
    return {"execution_plan": "repartition_exec", "input": below or scan(), "partitioning": {"HashDiff": [[col(key)], n]}}


def test_hash_diff_partitioning_parses_and_the_oracle_gives_every_key_its_partition():
    """`Partitioning::HashDiff(exprs, n)` (the fork's variant: flock-function/src/aws/window/session.rs:252) is part of the plan dialect since
    round 6; the oracle's restatement: one partition per distinct key, input order inside."""
    from flock_amd.runtime import explain
    t = {"i": [3, 1, 3, 2, 1, 3], "x": [10, 11, 12, 13, 14, 15]}
    parts = g.repartition_hash_diff(t, "i", 4)
    assert [p["i"] for p in parts] == [[1, 1], [2], [3, 3, 3], []] and [p["x"] for p in parts] == [[11, 14], [13], [10, 12, 15], []]
    with pytest.raises(ValueError):
        g.repartition_hash_diff(t, "i", 2)
    txt = explain(_hash_diff_plan("i", 4))
    assert "HashDiff, 4" in txt and "not supported" not in txt, txt


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["i", "j", "l"])
def test_hash_diff_repartition_gives_every_distinct_key_a_partition(gpu, key):
    """The session / global window launchers repartition a window's rows by `HashDiff(key, COUNT(DISTINCT key))` (session.rs:236-253): every
    distinct key one partition, rows in input order inside it.  Also under a FilterExec (the composed send order), and with more partitions
    named than keys exist (empty partitions behind) or fewer (refused)."""
    from flock_amd import FlockGpuError, _ffi
    from flock_amd.runtime import ExecutionContext
    r = np.random.default_rng(252)
    n = 30_000
    t = table(n, r, null_p=0.0)
    t[key] = [int(x) for x in r.choice(np.array([-7, 0, 5, 11, 2**20, 2**20 + 1, 99]) if key != "l" else np.array([-2**40, -1, 0, 3, 2**35, 2**35 + 7]), n)]
    distinct = len(set(t[key]))

    def run(plan):
        ctx = ExecutionContext([plan], gpu=gpu)
        try:
            ctx.feed_data_sources([[batches(t, 7_000)]])
            assert ctx.is_shuffling()
            return [pyrows(b[0]) for b in ctx.execute_partitioned()[0]]
        finally:
            ctx.close()
    for n_parts in (distinct, distinct + 3):
        got = run(_hash_diff_plan(key, n_parts))
        want = [g.rows(p) for p in g.repartition_hash_diff(t, key, n_parts)]
        assert len(got) == n_parts and sorted(map(tuple, (norm(p) for p in got))) == sorted(map(tuple, (norm(p) for p in want)))
        assert sum(len(p) for p in got) == n and all(len({row[NAMES.index(key)] for row in p}) <= 1 for p in got)
    pred = binary(col("j" if key != "j" else "i"), "Gt", lit("Int32", 0))
    got = run(_hash_diff_plan(key, distinct, {"execution_plan": "filter_exec", "predicate": pred, "input": scan()}))
    kept = g.filter_by_expr(t, pred)
    want = [g.rows(p) for p in g.repartition_hash_diff(kept, key, distinct)]
    assert sorted(map(tuple, (norm(p) for p in got))) == sorted(map(tuple, (norm(p) for p in want)))
    with pytest.raises(FlockGpuError) as e:
        run(_hash_diff_plan(key, distinct - 1))
    assert e.value.code == _ffi.ERR_INVALID and "distinct keys" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("ktypes", [("Int32", "Int32"), ("Int64", "Int64"), ("Int32", "Int64")])
@pytest.mark.parametrize("nl,nr", [(9_000, 200_000), (200_000, 9_000), (70_000, 70_000)])
def test_inner_join_on_spread_keys_unique_build_side_then_duplicates(gpu, ktypes, nl, nr):
    """The hashed join (relops.hpp join_hashed: 16-byte slots, unique build keys probed as a filter in the flag-tile geometry): a primary key
    spread over the whole key type on the smaller side (INT64_MIN, the table's free mark, among the Int64 keys), probe rows with and
    without a partner; then the SAME plan instance fed a build side that repeats keys -- the unique guess is wrong and the call must fall back
    to counted chains -- and once more (the guess is remembered)."""
    import pyarrow as pa
    from flock_amd.runtime import ExecutionContext, collect
    from test_plan_round5 import _field, _join_plan
    r = np.random.default_rng(hash((ktypes, nl, nr)) % 2**31)
    lt, rt = ktypes
    lf = [_field("a", lt, False), _field("x", "Int32", False)]
    rf = [_field("b", rt, False), _field("y", "Int64", False)]
    narrow = "Int32" in ktypes
    lo, hi = (-2**31, 2**31) if narrow else (-2**63, 2**63)
    n_small, n_big = min(nl, nr), max(nl, nr)
    pk = r.choice(np.arange(-n_small, n_small, dtype=np.int64), n_small, replace=False) * ((hi - lo) // (2 * n_small) - 1)   # unique, both signs, the full width
    if not narrow:
        pk[0] = -2**63
    fk = np.where(r.random(n_big) < 0.7, r.choice(pk, n_big), r.integers(lo, hi, n_big))
    pa_t = lambda t: pa.int32() if t == "Int32" else pa.int64()

    def tables(small, big):
        ka, kb = (small, big) if nl <= nr else (big, small)
        left = {"a": [int(v) for v in ka], "x": [int(v) for v in r.integers(-9, 9, len(ka))]}
        right = {"b": [int(v) for v in kb], "y": [int(v) for v in r.integers(-2**40, 2**40, len(kb))]}
        lb = [pa.record_batch([pa.array(left["a"], pa_t(lt)), pa.array(left["x"], pa.int32())], names=["a", "x"])]
        rb = [pa.record_batch([pa.array(right["b"], pa_t(rt)), pa.array(right["y"], pa.int64())], names=["b", "y"])]
        return left, right, lb, rb
    ctx = ExecutionContext([_join_plan(lf, rf, "a", "b")], gpu=gpu)
    gpu.profile_reset()
    gpu.profile(True)
    try:
        left, right, lb, rb = tables(pk, fk)
        out = collect(ctx, [[lb], [rb]])[0][0]
        ran = gpu.profile_read()
        assert "join_hash_probe_flag_kernel" in ran and "join_hash_probe_kernel" not in ran, sorted(ran)
        want = g.hash_join_inner(left, right, [("a", "b")])
        assert sorted(pyrows(out)) == sorted(g.rows(want)) and 0 < out.num_rows < n_big
        dup = pk.copy()
        dup[1:n_small // 3] = dup[n_small // 3:2 * (n_small // 3) - 1]        # a third of the build keys twice
        for _ in range(2):
            left, right, lb, rb = tables(dup, fk)
            gpu.profile_reset()
            out = collect(ctx, [[lb], [rb]])[0][0]
            assert "join_hash_probe_kernel" in gpu.profile_read()
            want = g.hash_join_inner(left, right, [("a", "b")])
            assert sorted(pyrows(out)) == sorted(g.rows(want)) and out.num_rows > 0
    finally:
        gpu.profile(False)
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["descriptions", "mixed", "threshold", "huge"])
def test_take_of_long_text_values(gpu, shape):
    """A filter's take of a Utf8 column whose values average more than the short-value kernel's stage holds (gather.hip, utf8_emit_long_kernel:
    16-byte output chunks made from realigned source chunks): descriptions of 50-99 bytes; a mix of NULLs, empty, one-to-five-byte and 200-400-byte
    values (chunks with three and more pieces, output windows beyond one round of the chunk map); values just above the threshold; values of 2-3 KB."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(hash(shape) % 2**31)
    n = {"descriptions": 30_000, "mixed": 20_000, "threshold": 30_000, "huge": 3_000}[shape]
    t = table(n, r, null_p=0.1)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz0123456789 -"))

    def text(lo, hi):
        return "".join(r.choice(letters, int(r.integers(lo, hi + 1))))
    if shape == "descriptions":
        t["s"] = [text(50, 99) for _ in range(n)]
    elif shape == "mixed":
        t["s"] = [None if x < 0.1 else "" if x < 0.25 else text(1, 5) if x < 0.6 else text(200, 400) for x in r.random(n)]
    elif shape == "threshold":
        t["s"] = [text(17, 23) for _ in range(n)]
    else:
        t["s"] = [text(2000, 3000) for _ in range(n)]
    pred = binary(col("j"), "Gt", lit("Int32", -2**29))     # ~5 of 8 rows
    ctx = ExecutionContext([{"execution_plan": "filter_exec", "predicate": pred, "input": scan()}], gpu=gpu)
    gpu.profile_reset()
    gpu.profile(True)
    try:
        rb = collect(ctx, [[batches(t, 7_000)]])[0][0]
        ran = gpu.profile_read()
    finally:
        gpu.profile(False)
        ctx.close()
    want = g.rows(g.filter_by_expr(t, pred))
    assert norm(pyrows(rb)) == norm(want) and 0 < len(want) < n
    assert "utf8_emit_long_kernel" in ran, sorted(ran)


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["increasing", "repeats", "shuffled"])
@pytest.mark.parametrize("under_join", [False, True])
def test_distinct_int32_utf8_pairs_in_and_out_of_key_order(gpu, order, under_join):
    """`SELECT DISTINCT j, s` (q8's `DISTINCT p_id, name`): strictly increasing keys keep every row without hashing (relops.hip,
    distinct_order_check_kernel); equal neighbours -- the same pair twice, or one key under two strings -- and keys in no order go through
    the hash set.  Alone, and under a join that takes the pairs' strings itself (plan.hip exec_lazy)."""
    from flock_amd.runtime import ExecutionContext, collect
    import pyarrow as pa
    from test_plan_round5 import _field
    r = np.random.default_rng(hash((order, under_join)) % 2**31)
    n = 20_000
    keys = np.cumsum(r.integers(1, 4, n)).astype(np.int64) - 7
    if order == "repeats":
        keys[1::3] = keys[0::3][:len(keys[1::3])]                 # every third row repeats its predecessor's key
    names = ["n%d" % (int(k) % 97) for k in keys]
    if order == "repeats":
        names = [nm if i % 6 else nm + "'" for i, nm in enumerate(names)]   # ... some of them under another string
    if order == "shuffled":
        perm = r.permutation(n)
        keys, names = keys[perm], [names[i] for i in perm]
        keys = np.concatenate([keys, keys[:500]])                  # and exact duplicates far apart
        names = names + names[:500]
    lf = [_field("j", "Int32", False), _field("s", "Utf8", False)]
    c = lambda nme, i: {"physical_expr": "column", "name": nme, "index": i}
    sc = lambda fs: {"execution_plan": "memory_exec", "schema": {"fields": fs, "metadata": {}}, "projection": list(range(len(fs)))}
    ge = [[c("j", 0), "j"], [c("s", 1), "s"]]
    part = {"execution_plan": "hash_aggregate_exec", "mode": "Partial", "group_expr": ge, "aggr_expr": [], "input": sc(lf), "input_schema": {"fields": lf, "metadata": {}},
            "schema": {"fields": lf, "metadata": {}}}
    rep = {"execution_plan": "repartition_exec", "input": part, "partitioning": {"Hash": [[c("j", 0), c("s", 1)], 4]}}
    dist = {"execution_plan": "hash_aggregate_exec", "mode": "FinalPartitioned", "group_expr": ge, "aggr_expr": [], "input": rep, "input_schema": {"fields": lf, "metadata": {}},
            "schema": {"fields": lf, "metadata": {}}}
    left_rb = [pa.record_batch([pa.array([int(k) for k in keys], pa.int32()), pa.array(names, pa.string())], names=["j", "s"])]
    pairs = sorted(set(zip((int(k) for k in keys), names)))
    if not under_join:
        ctx = ExecutionContext([dist], gpu=gpu)
        try:
            out = collect(ctx, [[left_rb]])[0][0]
        finally:
            ctx.close()
        assert sorted(pyrows(out)) == [list(p) for p in pairs] or sorted(map(tuple, pyrows(out))) == pairs
        return
    rf = [_field("b", "Int32", False), _field("y", "Int64", False)]
    kb = r.choice(np.concatenate([keys, keys + 1_000_000]), 30_000)
    right = {"b": [int(x) for x in kb], "y": [int(x) for x in r.integers(-9, 9, len(kb))]}
    right_rb = [pa.record_batch([pa.array(right["b"], pa.int32()), pa.array(right["y"], pa.int64())], names=["b", "y"])]
    side = lambda inner, k, fs: {"execution_plan": "coalesce_batches_exec", "target_batch_size": 4096,
                                 "input": {"execution_plan": "repartition_exec", "input": inner, "partitioning": {"Hash": [[c(k, 0)], 4]}}}
    join = {"execution_plan": "hash_join_exec", "left": side(dist, "j", lf), "right": side(sc(rf), "b", rf), "join_type": "Inner", "mode": "Partitioned",
            "on": [[c("j", 0), c("b", 0)]], "schema": {"fields": lf + rf, "metadata": {}}}
    ctx = ExecutionContext([join], gpu=gpu)
    try:
        for _ in range(2):   # (twice: the second execute may put the table on the other side)
            out = collect(ctx, [[left_rb], [right_rb]])[0][0]
            want = g.hash_join_inner({"j": [p[0] for p in pairs], "s": [p[1] for p in pairs]}, right, [("j", "b")])
            assert sorted(map(tuple, pyrows(out))) == sorted(map(tuple, g.rows(want))) and out.num_rows > 0
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ktype", ["Int32", "Int64"])
@pytest.mark.parametrize("spread", [1, 7, 1_000_003])
def test_group_by_without_aggregates_is_distinct(gpu, ktype, spread):
    """`SELECT DISTINCT seller` (q8.dag's second aggregate: a GROUP BY with no aggregate): dense keys take the perfect-hash GROUP BY with no
    accumulator, keys spread wider than their rows the hash table; negative keys, every key many times, one key once."""
    from flock_amd.runtime import ExecutionContext, collect
    import pyarrow as pa
    from test_plan_round5 import _field
    r = np.random.default_rng(hash((ktype, spread)) % 2**31)
    n = 50_000
    if ktype == "Int32":
        spread = min(spread, 200_000)          # (9001 x 200000 still fits an Int32)
    keys = (r.integers(-3_000, 9_000, n) * spread).astype(np.int64)
    keys[17] = 9_001 * spread
    f = [_field("k", ktype, False), _field("x", "Int32", False)]
    c = {"physical_expr": "column", "name": "k", "index": 0}
    sc = {"execution_plan": "memory_exec", "schema": {"fields": f, "metadata": {}}, "projection": [0, 1]}
    out_f = [f[0]]
    part = {"execution_plan": "hash_aggregate_exec", "mode": "Partial", "group_expr": [[c, "k"]], "aggr_expr": [], "input": sc, "input_schema": {"fields": f, "metadata": {}},
            "schema": {"fields": out_f, "metadata": {}}}
    rep = {"execution_plan": "repartition_exec", "input": part, "partitioning": {"Hash": [[c], 4]}}
    plan = {"execution_plan": "hash_aggregate_exec", "mode": "FinalPartitioned", "group_expr": [[c, "k"]], "aggr_expr": [], "input": rep, "input_schema": {"fields": out_f, "metadata": {}},
            "schema": {"fields": out_f, "metadata": {}}}
    rb = [pa.record_batch([pa.array([int(v) for v in keys], pa.int32() if ktype == "Int32" else pa.int64()), pa.array([1] * n, pa.int32())], names=["k", "x"])]
    ctx = ExecutionContext([plan], gpu=gpu)
    gpu.profile_reset()
    gpu.profile(True)
    try:
        out = collect(ctx, [[rb]])[0][0]
        ran = gpu.profile_read()
    finally:
        gpu.profile(False)
        ctx.close()
    assert sorted(v[0] for v in pyrows(out)) == sorted(set(int(v) for v in keys))
    assert ("dense_group_kernel" in ran) == (spread <= 7), sorted(ran)


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [None, 1_000_000])
def test_order_by_one_int32_key_over_a_million_rows(gpu, limit):
    """arch/ops/sort.sql's shape at a size where its takes change form (gather.hpp gather_fixed_packed: above 2^20 rows the fixed-width columns are
    interleaved into 16-byte records and taken record by record; the sorted key column comes out of the radix passes themselves): 1.3e6 rows,
    `ORDER BY k` -- every column of every row against numpy's stable sort, ties in input order; with and without a LIMIT."""
    from flock_amd.runtime import ExecutionContext, collect
    import pyarrow as pa
    from test_plan_round5 import _field
    r = np.random.default_rng(1311 + (limit or 0))
    n = 1_300_000
    cols = {"a": r.integers(-2**31, 2**31 - 1, n).astype(np.int32), "k": r.integers(-5_000, 60_000, n).astype(np.int32),
            "p": r.integers(0, 10_000_000, n).astype(np.int32), "t": r.integers(-2**60, 2**60, n).astype(np.int64), "u": r.integers(0, 2**62, n).astype(np.int64)}
    f = [_field("a", "Int32", False), _field("k", "Int32", False), _field("p", "Int32", False), _field("t", "Int64", False), _field("u", "Int64", False)]
    scan_ = {"execution_plan": "memory_exec", "schema": {"fields": f, "metadata": {}}, "projection": list(range(len(f)))}
    plan = {"execution_plan": "sort_exec", "input": scan_, "expr": [{"expr": {"physical_expr": "column", "name": "k", "index": 1}, "options": {"descending": False, "nulls_first": False}}]}
    if limit:
        plan = {"execution_plan": "global_limit_exec", "limit": limit, "input": plan}
    rb = [pa.record_batch([pa.array(cols[c]) for c in ("a", "k", "p", "t", "u")], names=["a", "k", "p", "t", "u"])]
    ctx = ExecutionContext([plan], gpu=gpu)
    gpu.profile_reset()
    gpu.profile(True)
    try:
        out = collect(ctx, [[rb]])[0][0]
        ran = gpu.profile_read()
    finally:
        gpu.profile(False)
        ctx.close()
    order = np.argsort(cols["k"], kind="stable")[:limit]
    assert out.num_rows == len(order)
    for i, c in enumerate(("a", "k", "p", "t", "u")):
        assert np.array_equal(out.column(i).to_numpy(), cols[c][order]), c
    assert ("gather_records_kernel" in ran) == (limit is None), sorted(ran)   # (a LIMIT that keeps under half the rows takes them the plain way)
