"""Stage plans and `execute_partitioned`: the reference's distributed mode cuts a query plan into
stage plans (flock/src/distributed_plan/stage.rs:269-367; shapes pinned as text in flock/src/distributed_plan/planner.rs:
148-171 and playground/src/distributed_plan/nexmark/q{3,5,8}.dag), runs a stage that ends in a hash repartition with
`execute_partitioned` (flock/src/runtime/context.rs:197-216,328-337; flock-function/src/aws/actor.rs:60-66) and routes
partition j to member j of the next function group (actor.rs:425-543).

CPU: the splitters reproduce the stage shapes the reference asserts, and every stage plan is executable by the engine.
GPU: stage-0 o exchange o stage-1 ... over `runtime.collect` == the whole-query plan == the oracle, with P real partitions."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

import oracle
from oracle import generic_ops as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANS = os.path.join(ROOT, "tests", "golden", "plans")
TS = pa.timestamp("ms")


def _plan(q):
    return json.load(open(os.path.join(PLANS, f"q{q}.json")))


def _count(plan, what):
    text = json.dumps(plan)
    return text.count(what)


# ------------------------------------------------------------------ CPU
def test_stage_splitter_reproduces_the_reference_q3_stage_shapes():
    """planner.rs:173-198 asserts these counts on the stages of q3 (and prints the trees, planner.rs:148-171)."""
    from flock_amd.stages import build_query_dag
    stages = build_query_dag(_plan(3))
    assert len(stages) == 3
    stage0 = [s for s in stages if s.is_shuffling]          # the two plans of "Stage 0"
    stage1 = stages[-1]
    assert len(stage0) == 2 and not stage1.is_shuffling and sorted(stage1.inputs) == [0, 1]
    both = [s.plan for s in stage0]
    assert sum(_count(p, '"Hash"') for p in both) == 2
    assert sum(_count(p, '"RoundRobinBatch"') for p in both) == 2
    assert sum(_count(p, '"coalesce_batches_exec"') for p in both) == 4
    assert sum(_count(p, '"filter_exec"') for p in both) == 2
    assert sum(_count(p, '"memory_exec"') for p in both) == 2
    assert _count(stage1.plan, '"projection_exec"') == 1 and _count(stage1.plan, '"coalesce_batches_exec"') == 1
    assert _count(stage1.plan, '"memory_exec"') == 2 and _count(stage1.plan, '"hash_join_exec"') == 1


@pytest.mark.parametrize("q,n_dag,n_fine", [(3, 3, 3), (5, 3, 6), (8, 3, 5)])
def test_every_stage_plan_is_executable(q, n_dag, n_fine):
    """Both split rules: stage.rs (cuts at final aggregates / joins) and the playground's ShuffleWriter stages (a cut at every
    hash repartition and every gather: q5.dag lists 6 distinct sub-plans of this kind, q8.dag 5, q3.dag 3)."""
    from flock_amd.runtime import explain
    from flock_amd.stages import build_query_dag, split_at_repartitions
    for rule, n in ((build_query_dag, n_dag), (split_at_repartitions, n_fine)):
        stages = rule(_plan(q))
        assert len(stages) == n
        assert not stages[-1].is_shuffling
        if rule is build_query_dag or q != 5:        # (q5's Partial MAX stage sits under a gather: it is not shuffled)
            assert all(s.is_shuffling for s in stages[:-1])
        for s in stages:
            text = explain(s.plan)                            # raises FlockGpuError when a node is not supported
            assert text.startswith("Repartition(Hash, 8)") == s.is_shuffling
        fed = sorted(i for s in stages for i in s.inputs if i is not None)
        assert fed == list(range(len(stages) - 1))           # every stage but the root feeds exactly one leaf


def test_dag_nodes_and_edges_equal_the_counts_stage_rs_asserts():
    """a join's left and right sub-plans are ONE node of the reference's QueryDag (stage.rs:330-334).  Node / edge counts
    of stage.rs's own tests: aggregate 2 / 1 (stage.rs:605-606), join 2 nodes (stage.rs:682, :726), sort + limit over a join 3 / 2
    (stage.rs:776-901)."""
    from flock_amd.stages import build_query_dag, dag_edge_count, dag_nodes
    agg = build_query_dag(json.load(open(os.path.join(PLANS, "golden_aggregate.json"))))
    assert len(dag_nodes(agg)) == 2 and dag_edge_count(agg) == 1
    join_plan = json.load(open(os.path.join(PLANS, "golden_join.json")))
    join = build_query_dag(join_plan)
    assert len(join) == 3 and dag_nodes(join) == [[0, 1], [2]] and dag_edge_count(join) == 1
    sorted_join = {"execution_plan": "global_limit_exec", "limit": 3,
                   "input": {"execution_plan": "sort_exec", "expr": [], "input": join_plan}}
    st = build_query_dag(sorted_join)
    assert len(st) == 4 and len(dag_nodes(st)) == 3 and dag_edge_count(st) == 2
    assert dag_nodes(st)[0] == [0, 1] and all(j is None for i in (0, 1) for j in st[i].inputs)     # the two base-fed plans share node 0
    for q in (3, 5, 8):
        assert [len(g) for g in dag_nodes(build_query_dag(_plan(q)))] == [2, 1]


def test_partial_count_stage_uses_the_fused_kernel():
    from flock_amd.runtime import explain
    from flock_amd.stages import split_at_repartitions
    s0 = split_at_repartitions(_plan(5))[0]
    assert "fused Partial COUNT" in explain(s0.plan)


# ------------------------------------------------------------------ GPU
def _utf8(u):
    return pa.StringArray.from_buffers(len(u), pa.py_buffer(u.offsets.tobytes()), pa.py_buffer(u.data.tobytes()))


def _relations(seed, eps, n):
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    b, a, p = s.bids(0, n), s.auctions(0, n), s.persons(0, n)
    bid = pa.record_batch([pa.array(b["auction"]), pa.array(b["bidder"]), pa.array(b["price"]), pa.array(b["b_date_time"]).cast(TS)],
                          names=["auction", "bidder", "price", "b_date_time"])
    auc = pa.record_batch([pa.array(a["a_id"]), pa.array(a["seller"]), pa.array(a["category"])], names=["a_id", "seller", "category"])
    per = pa.record_batch([pa.array(p["p_id"]), _utf8(p["name"]), _utf8(p["city"]), _utf8(p["state"])], names=["p_id", "name", "city", "state"])
    host = {"bid": {k: v.tolist() for k, v in b.items()},
            "auction": {k: a[k].tolist() for k in ("a_id", "seller", "category")},
            "person": {"p_id": p["p_id"].tolist(), "name": p["name"].to_pylist(), "city": p["city"].to_pylist(), "state": p["state"].to_pylist()}}
    return {"bid": bid, "auction": auc, "person": per}, host


def _rows(batches):
    out = []
    for rb in batches:
        cols = [rb[c].to_pylist() for c in rb.schema.names]
        out.extend(zip(*cols))
    return sorted(out)


def run_staged(gpu, stages, relations, chunks=2):
    """The distributed run of a stage DAG in one process: a shuffling stage is executed with execute_partitioned (through
    `collect`) by `chunks` producers, each over its slice of the base relation; partition j of every producer goes to
    invocation j of the consuming stage, as actor.rs:425-543 routes it.  Returns (batches of the root stage, partition
    sizes seen per shuffling stage)."""
    from flock_amd.runtime import ExecutionContext, collect
    outputs, sizes = {}, {}
    for i, st in enumerate(stages):
        ctx = ExecutionContext([st.plan], name=f"stage-{i}", gpu=gpu)
        assert ctx.is_shuffling() == st.is_shuffling
        feeders = [j for j in st.inputs if j is not None]
        base = any(j is None for j in st.inputs)
        if base and feeders:
            raise AssertionError("a stage fed by both a base relation and another stage")
        invocations = []
        if base:
            for c in range(chunks):                               # `chunks` source functions, each with a slice of the rows
                src = []
                for rb in relations.values():
                    lo, hi = rb.num_rows * c // chunks, rb.num_rows * (c + 1) // chunks
                    src.append([[rb.slice(lo, hi - lo)]])
                invocations.append(src)
        else:
            parts = max(len(outputs[j]) for j in feeders)
            for p in range(parts):
                src = []
                for j in feeders:
                    o = outputs[j]
                    # a non-partitioned producer sends everything to every member (it has one member in these plans)
                    src.append([o[p] if len(o) == parts else [b for part in o for b in part]])
                invocations.append(src)
        result = None
        for src in invocations:
            out = collect(ctx, src)
            if st.is_shuffling:
                assert len(out) == ctx.plans[0].partitions == 8      # Hash(.., 8) in the fixtures
                result = result or [[] for _ in out]
                for p, batches in enumerate(out):
                    result[p].extend(batches)
            else:
                assert len(out) == 1
                result = result or [[]]
                result[0].extend(out[0])
        outputs[i] = result
        if st.is_shuffling:
            sizes[i] = [sum(b.num_rows for b in part) for part in result]
        ctx.close()
    return [b for part in outputs[len(stages) - 1] for b in part], sizes, outputs


@pytest.fixture(scope="module")
def gpu():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _whole(gpu, q, relations):
    from flock_amd.runtime import ExecutionContext, collect
    ctx = ExecutionContext([_plan(q)], gpu=gpu)
    out = collect(ctx, [[[rb]] for rb in relations.values()])
    ctx.close()
    return out[0]


def _oracle_rows(q, host):
    if q == 3:
        return sorted(g.rows(g.nexmark_q3(host["auction"], host["person"])))
    if q == 5:
        r = g.nexmark_q5(host["bid"])
        return sorted(zip(r["auction"], r["num"]))
    return sorted(g.rows(g.nexmark_q8(host["person"], host["auction"])))


@pytest.mark.gpu
@pytest.mark.parametrize("rule", ["build_query_dag", "split_at_repartitions"])
@pytest.mark.parametrize("q,seed,eps,n", [(3, 3, 50_000, 150_000), (5, 5, 20_000, 100_000), (8, 8, 50_000, 200_000), (5, 1, 100, 50)])
def test_staged_execution_equals_whole_plan_equals_oracle(gpu, rule, q, seed, eps, n):
    from flock_amd import stages as S
    relations, host = _relations(seed, eps, n)
    stages = getattr(S, rule)(_plan(q))
    # base-fed stages of the fine split are filters / Partial aggregates: any number of source functions may each run them
    # over a slice of the window.  stage.rs keeps Partial -> Final inside one stage plan, which only one producer may run.
    got, sizes, outputs = run_staged(gpu, stages, relations, chunks=2 if rule == "split_at_repartitions" else 1)
    want = _oracle_rows(q, host)
    assert _rows(got) == want                                              # stage-0 o exchange o stage-1 == oracle
    assert _rows(_whole(gpu, q, relations)) == want                        # == the whole-query plan (fused pipeline)
    if n > 1000:
        assert len(want) > 0
        # real partitions: a shuffle of more than a handful of rows reaches all 8 destinations
        big = [i for i in sizes if sum(sizes[i]) > 1000]
        assert big and all(c > 0 for i in big for c in sizes[i]), sizes
    # equal keys meet in one partition: no key of a shuffling stage's output appears in two partitions
    for i, st in enumerate(stages):
        if not st.is_shuffling:
            continue
        key = st.plan["input"]["partitioning"]["Hash"][0][0]["name"]
        seen = {}
        for p, part in enumerate(outputs[i]):
            for b in part:
                for k in set(b[key].to_pylist()):
                    assert seen.setdefault(k, p) == p, (i, key, k)


@pytest.mark.gpu
@pytest.mark.parametrize("q,seed,eps,n", [(3, 3, 50_000, 150_000), (5, 5, 20_000, 100_000), (8, 8, 50_000, 200_000)])
def test_one_function_instance_may_host_several_partitions(gpu, q, seed, eps, n):
    """`StagedRun(instances=k)`: partition p of a shuffle goes to instance p % k of the consuming stage, which feeds all it gets
    into one execute.  Same rows as one instance per partition (and as the oracle) for k = 1 and for a k that does not divide 8;
    and with the source stages sharing one device copy of a relation (`share_sources`: q5's two subplans both scan `bid`); and with
    the stages' results handed over in HBM (`on_device`: flockgpu_plan_execute_retain / flockgpu_plan_feed_from)."""
    from flock_amd import stages as S
    relations, host = _relations(seed, eps, n)
    want = _oracle_rows(q, host)
    assert len(want) > 0
    for k, share, dev in ((0, False, False), (1, False, False), (3, False, False), (1, True, False), (1, False, True), (1, True, True)):
        run = S.StagedRun(gpu, S.build_query_dag(_plan(q)), instances=k, share_sources=share, on_device=dev)
        try:
            src = {name: relations[name] for name in (["bid"] if q == 5 else (["person", "auction"] if q == 8 else ["auction", "person"]))}
            assert _rows(run.run(src)) == want, (k, share, dev)
            assert _rows(run.run(src)) == want, (k, share, dev)     # the plans are reusable: a second window through the same instances
        finally:
            run.close()


@pytest.mark.gpu
def test_a_stage_reads_its_producers_results_on_the_device(gpu):
    """flockgpu_plan_execute_retain / flockgpu_plan_feed_from on q3's three stage plans: the join stage finds each of its two inputs
    among the producers by column names, a producer without a retained result or with the wrong columns is refused without a trace,
    and a second window through the same plans gives the second window's rows."""
    from flock_amd import FlockGpuError, _ffi
    from flock_amd import stages as S
    from flock_amd.runtime import ExecutionContext
    st = S.build_query_dag(_plan(3))
    persons, auctions, join = (ExecutionContext([x.plan], gpu=gpu) for x in st)
    try:
        for seed, n in ((3, 150_000), (4, 90_000)):
            relations, host = _relations(seed, 50_000, n)
            persons.feed_data_sources([[[relations["person"]]]])
            auctions.feed_data_sources([[[relations["auction"]]]])
            with pytest.raises(FlockGpuError) as e:                     # nothing retained yet
                join.plans[0].feed_from(0, persons.plans[0])
            assert e.value.code == _ffi.ERR_INVALID
            assert persons.execute_retain()[0] > 0 and auctions.execute_retain()[0] > 0
            assert not join.plans[0].feed_from(0, persons.plans[0]) or not join.plans[0].feed_from(1, persons.plans[0])   # one of the two leaves wants the auctions
            join.clean_data_sources()
            join.feed_from([persons, auctions])
            got = _rows(join.execute()[0])
            assert got == _oracle_rows(3, host) and len(got) > 0
            for c in (join, auctions, persons):
                c.clean_data_sources()
            with pytest.raises(FlockGpuError):                          # a reset drops the retained result
                join.plans[0].feed_from(0, persons.plans[0])
    finally:
        for c in (persons, auctions, join):
            c.close()


@pytest.mark.gpu
def test_a_retained_result_survives_another_plans_execute(gpu):
    """q5's two source stage plans both run the fused Partial COUNT, whose output buffers belong to the context, not to a plan: the
    counts stage keeps ITS result while the MAX stage executes the same pipeline over OTHER bids afterwards.  The join stage must see
    the first window's counts against the second window's maximum."""
    from flock_amd import stages as S
    from flock_amd.runtime import ExecutionContext
    st = S.build_query_dag(_plan(5))
    mx, counts, join = (ExecutionContext([x.plan], gpu=gpu) for x in st)     # stage 0: MAX(num); stage 1: (auction, num); stage 2: join
    try:
        rel_a, _ = _relations(5, 20_000, 100_000)
        rng = np.random.default_rng(3)
        top = int(np.unique(rel_a["bid"]["auction"].to_numpy(), return_counts=True)[1].max())
        # window B: a handful of auctions, its maximum count chosen to be a count that occurs in window A
        a_vals, a_cnt = np.unique(rel_a["bid"]["auction"].to_numpy(), return_counts=True)
        target = int(np.sort(a_cnt)[len(a_cnt) // 2])
        b_auction = np.concatenate([np.full(target, 7, np.int32), np.full(max(target - 1, 1), 8, np.int32)])
        rel_b = pa.record_batch([pa.array(b_auction), pa.array(b_auction), pa.array(b_auction), pa.array(np.zeros(len(b_auction), np.int64)).cast(TS)],
                                names=["auction", "bidder", "price", "b_date_time"])
        counts.feed_data_sources([[[rel_a["bid"]]]])
        counts.execute_retain()
        mx.feed_data_sources([[[rel_b]]])
        mx.execute_retain()
        join.feed_from([counts, mx])
        got = _rows(join.execute()[0])
        want = sorted((int(a), int(c)) for a, c in zip(a_vals, a_cnt) if c == target)
        assert got == want and len(want) > 0 and target < top
    finally:
        for c in (join, mx, counts):
            c.clean_data_sources()
            c.close()


@pytest.mark.gpu
def test_two_plans_read_one_upload(gpu):
    """flockgpu_plan_feed_shared: q5's stage plans both scan `bid.auction`; the second reads the first one's device copy.  A plan
    that reads a column the donor never uploaded is refused (nothing changes, it is fed its own copy), a shared leaf cannot be
    appended to, and after the donor's next window the borrower sees the new rows only through a new share."""
    from flock_amd import FlockGpuError, _ffi
    from flock_amd import stages as S
    from flock_amd.runtime import ExecutionContext
    relations, host = _relations(5, 20_000, 100_000)
    st = S.build_query_dag(_plan(5))
    a, b = ExecutionContext([st[0].plan], gpu=gpu), ExecutionContext([st[1].plan], gpu=gpu)
    wants_price = ExecutionContext([_plan(2)], gpu=gpu)          # q2 reads bid.auction AND bid.price
    try:
        src = [[[relations["bid"]]]]
        b.feed_data_sources(src)
        own = _rows(x for part in b.execute_partitioned()[0] for x in part)
        b.clean_data_sources()
        a.feed_data_sources(src)
        assert b.share_data_sources(a)
        with pytest.raises(FlockGpuError) as e:                  # no appending to a relation that is not this plan's
            b.plans[0].feed(0, [relations["bid"]])
        assert e.value.code == _ffi.ERR_INVALID
        assert not wants_price.share_data_sources(a)             # the donor uploaded `auction` only
        shared = _rows(x for part in b.execute_partitioned()[0] for x in part)
        assert shared == own and len(own) > 0
        a.clean_data_sources()
        b.clean_data_sources()
        half = relations["bid"].slice(0, 50_000)                 # the next window: fewer rows
        a.feed_data_sources([[[half]]])
        assert b.share_data_sources(a)
        small = sum(x.num_rows for part in b.execute_partitioned()[0] for x in part)
        b.clean_data_sources()
        b.feed_data_sources([[[half]]])
        assert small == sum(x.num_rows for part in b.execute_partitioned()[0] for x in part) and 0 < small < len(own)
    finally:
        for c in (a, b, wants_price):
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 2, 3, 5, 7, 8, 13])
def test_generic_operators_equal_the_fused_pipelines(gpu, q, monkeypatch):
    """Every whole-query plan once through its fused pipeline and once with FLOCKGPU_PLAN_GENERIC_ONLY (relops.hip only)."""
    from flock_amd.runtime import ExecutionContext, collect
    relations, _ = _relations(11, 30_000, 120_000)
    key = np.arange(1000, 3000, 7, dtype=np.int32)
    side = pa.record_batch([pa.array(key), pa.array((key * 5).astype(np.int32))], names=["key", "value"])
    src = [[[rb]] for rb in relations.values()] + [[[side]]]

    def run(generic_only=False):
        ctx = ExecutionContext([_plan(q)], gpu=gpu, generic_only=generic_only)
        text = ctx.plans[0].description
        out = collect(ctx, src)[0]
        ctx.close()
        return text, out
    fused_text, fused = run()
    generic_text, generic = run(generic_only=True)
    assert "fused" not in generic_text and (q == 1 or "fused" in fused_text)
    assert fused[0].schema == generic[0].schema
    assert _rows(fused) == _rows(generic) and fused[0].num_rows > 0


@pytest.mark.gpu
def test_look_alike_projections_come_back_in_the_plans_shape(gpu):
    """q2 / q3 look-alikes with a swapped, reduced or aliased root projection."""
    from flock_amd.runtime import ExecutionContext, collect
    relations, host = _relations(2, 20_000, 60_000)
    src = [[[rb]] for rb in relations.values()]
    a, p = np.array(host["bid"]["auction"], np.int32), np.array(host["bid"]["price"], np.int32)
    wa, wp = oracle.q2_filter(a, p)
    p2 = _plan(2)
    p2["expr"] = [p2["expr"][1], p2["expr"][0]]
    p2["schema"]["fields"] = p2["schema"]["fields"][::-1]
    ctx = ExecutionContext([p2], gpu=gpu)
    rb = collect(ctx, src)[0][0]
    assert rb.schema.names == ["price", "auction"] and rb["price"].to_numpy().tolist() == wp.tolist() and rb["auction"].to_numpy().tolist() == wa.tolist()
    ctx.close()
    p2["expr"] = [[p2["expr"][0][0], "cost"]]
    ctx = ExecutionContext([p2], gpu=gpu)
    rb = collect(ctx, src)[0][0]
    assert rb.schema.names == ["cost"] and rb.num_columns == 1 and rb["cost"].to_numpy().tolist() == wp.tolist()
    ctx.close()
    p3 = _plan(3)
    p3["expr"] = p3["expr"][::-1]
    ctx = ExecutionContext([p3], gpu=gpu)
    rb = collect(ctx, src)[0][0]
    assert rb.schema.names == ["a_id", "state", "city", "name"]
    want = sorted((r[3], r[2], r[1], r[0]) for r in g.rows(g.nexmark_q3(host["auction"], host["person"])))
    assert _rows([rb]) == want and len(want) > 0
    ctx.close()


@pytest.mark.gpu
def test_simple_select_shape_and_pinned_feed(gpu):
    """The shape of the reference's simple_select.json (a projection of an Int64 column), fed once from pageable and once
    from pinned host memory (flockgpu_host_alloc): same rows; appended feeds accumulate."""
    import ctypes as C
    from flock_amd import _ffi
    from flock_amd.runtime import ExecutionContext, collect
    lib = _ffi.load()
    plan = open(os.path.join(PLANS, "simple_select.json")).read()
    values = np.arange(-50_000, 3_000_000, 7, dtype=np.int64)
    pageable = pa.record_batch([pa.array(values)], names=["c1"])
    ptr = C.c_void_p()
    assert lib.flockgpu_host_alloc(values.nbytes, C.byref(ptr)) == 0
    C.memmove(ptr, values.ctypes.data, values.nbytes)
    buf = pa.foreign_buffer(ptr.value, values.nbytes)
    pinned = pa.record_batch([pa.Array.from_buffers(pa.int64(), len(values), [None, buf])], names=["c1"])
    ctx = ExecutionContext([plan], gpu=gpu)
    a = collect(ctx, [[[pageable]]])[0][0]
    b = collect(ctx, [[[pinned]]])[0][0]
    assert a.schema.names == ["c1"] and a["c1"].to_numpy().tolist() == values.tolist() and a.equals(b)
    c = collect(ctx, [[[pageable.slice(0, 1000), pinned.slice(1000, 5000)], [pageable.slice(6000)]]])[0][0]
    assert c["c1"].to_numpy().tolist() == values.tolist()
    ctx.close()
    del pinned, buf, b, c
    assert lib.flockgpu_host_free(ptr) == 0
