"""q4 / q9 / YSB through the plan-level ABI: the physical plans the reference pins
as text (flock/src/distributed_plan/planner.rs:218-256 q4, :298-346 YSB; benchmarks/src/nexmark/query/q9_plan.fmt) parse,
split into the reference's stages, and run -- as one plan (fused pipeline), on the generic operators alone, and stage by
stage with real hash partitions -- to the oracle's rows."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

import oracle
from test_stage_plans import PLANS, TS, _rows, _utf8, run_staged


def _plan(name):
    return json.load(open(os.path.join(PLANS, f"{name}.json")))


# ------------------------------------------------------------------ CPU: shapes
def test_q4_and_ysb_split_into_the_reference_stages():
    """planner.rs:218-256: q4 = 4 stages (stage 0 holds the two repartitions of the base relations), :298-346: YSB = 3."""
    from flock_amd.runtime import explain
    from flock_amd.stages import build_query_dag, stage_levels
    s4 = build_query_dag(_plan("q4"))
    assert len(s4) == 5 and max(stage_levels(s4)) + 1 == 4
    text = [json.dumps(s.plan) for s in s4]
    lv = stage_levels(s4)
    stage1 = [t for t, l in zip(text, lv) if l == 1]
    assert len(stage1) == 1 and stage1[0].count('"memory_exec"') == 2 and '"hash_join_exec"' in stage1[0] and '"Partial"' in stage1[0]
    assert '"filter_exec"' in stage1[0] and stage1[0].count('"Hash"') == 1
    stage2 = [t for t, l in zip(text, lv) if l == 2][0]
    assert '"FinalPartitioned"' in stage2 and '"Partial"' in stage2 and stage2.count('"projection_exec"') == 2 and '"avg"' in stage2
    stage3 = [t for t, l in zip(text, lv) if l == 3][0]
    assert stage3.count('"projection_exec"') == 1 and '"FinalPartitioned"' in stage3 and stage3.count('"memory_exec"') == 1
    sy = build_query_dag(_plan("ysb"))
    assert len(sy) == 4 and max(stage_levels(sy)) + 1 == 3
    for st in s4 + sy + build_query_dag(_plan("q9")):
        explain(st.plan)                                          # every stage plan is executable (raises otherwise)


def test_whole_plans_are_recognised():
    import ctypes as C
    from flock_amd import _ffi
    from flock_amd.runtime import explain
    lib = _ffi.load()
    for name, number, what in (("q4", 4, "fused q4"), ("q9", 9, "fused q9"), ("ysb", 100, "fused YSB")):
        text = json.dumps(_plan(name)).encode()
        got = C.c_int(-1)
        assert lib.flockgpu_plan_recognise(text, len(text), C.byref(got)) == _ffi.OK and got.value == number
        assert what in explain(text.decode())


def test_look_alikes_are_not_taken_for_q4_q9():
    """A different bound column, a third aggregate or a different second join key keeps the plan on the generic operators."""
    from flock_amd.runtime import explain
    p = _plan("q9")
    p["input"]["input"]["on"][1][0]["name"], p["input"]["input"]["on"][1][0]["index"] = "bidder", 1      # bidder = final
    assert "fused q9" not in explain(p)
    text = json.dumps(_plan("q4")).replace('"LtEq"', '"Lt"')                                              # b_date_time < expires
    assert "fused q4" not in explain(text) and "generic" in explain(text)


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gpu():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _nexmark(seed, eps, n, shuffle_auctions=False):
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    b, a = s.bids(0, n), s.auctions(0, n)
    if shuffle_auctions:
        order = np.random.default_rng(seed).permutation(len(a["a_id"]))
        a = {k: v[order] for k, v in a.items()}
    bid = pa.record_batch([pa.array(b["auction"]), pa.array(b["bidder"]), pa.array(b["price"]), pa.array(b["b_date_time"]).cast(TS)],
                          names=["auction", "bidder", "price", "b_date_time"])
    auc = pa.record_batch([pa.array(a["a_id"]), pa.array(a["a_date_time"]).cast(TS), pa.array(a["expires"]).cast(TS), pa.array(a["seller"]),
                           pa.array(a["category"])], names=["a_id", "a_date_time", "expires", "seller", "category"])
    return {"bid": bid, "auction": auc}, a, b


def _want(q, a, b):
    args = (a["a_date_time"], a["expires"], b["auction"], b["price"], b["b_date_time"])
    if q == "q9":
        rows = oracle.q9_winning_bids(a["a_id"], *args)
        return sorted(zip(b["auction"][rows].tolist(), b["bidder"][rows].tolist(), b["price"][rows].tolist(), b["b_date_time"][rows].tolist()))
    cats, avg = oracle.q4_avg_final_by_category(a["a_id"], a["category"], *args)
    return sorted(zip(cats.tolist(), avg.tolist()))


def _got(batches):
    out = []
    for rb in batches:
        cols = []
        for c in rb.schema.names:
            col = rb[c]
            cols.append(col.cast(pa.int64()).to_pylist() if pa.types.is_timestamp(col.type) else col.to_pylist())
        out.extend(zip(*cols))
    return sorted(out)


def _collect_whole(gpu, plan, sources, generic_only=False):
    from flock_amd.runtime import ExecutionContext, collect
    ctx = ExecutionContext([plan], gpu=gpu, generic_only=generic_only)
    text = ctx.plans[0].description
    out = collect(ctx, [[[rb]] for rb in sources.values()])[0]
    ctx.close()
    return text, out


@pytest.mark.gpu
@pytest.mark.parametrize("q", ["q9", "q4"])
@pytest.mark.parametrize("seed,eps,n", [(3, 20_000, 100_000), (8, 200, 150), (5, 300_000, 900_000)])
def test_whole_plan_fused_and_generic_equal_the_oracle(gpu, q, seed, eps, n, monkeypatch):
    relations, a, b = _nexmark(seed, eps, n)
    want = _want(q, a, b)
    text, out = _collect_whole(gpu, _plan(q), relations)
    assert f"fused {q}" in text
    assert _got(out) == want                                      # AVG: Float64 equality, bit for bit
    if n > 1000:
        assert len(want) > 0
    text, out = _collect_whole(gpu, _plan(q), relations, generic_only=True)   # FLOCKGPU_PLAN_GENERIC_ONLY
    assert "fused" not in text and _got(out) == want


@pytest.mark.gpu
@pytest.mark.parametrize("q", ["q9", "q4"])
def test_unsorted_auction_ids_fall_back_to_the_generic_operators(gpu, q):
    """The fused q4 / q9 kernels need dense, increasing auction ids inside a batch; the plan path must not: a shuffled auction
    relation still gives the oracle's rows (through relops.hip, picked at run time)."""
    relations, a, b = _nexmark(21, 10_000, 60_000, shuffle_auctions=True)
    text, out = _collect_whole(gpu, _plan(q), relations)
    assert f"fused {q}" in text and _got(out) == _want(q, a, b) and len(out[0]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("rule", ["build_query_dag", "split_at_repartitions"])
@pytest.mark.parametrize("q", ["q9", "q4"])
def test_staged_q4_q9_equal_the_oracle(gpu, q, rule):
    from flock_amd import stages as S
    relations, a, b = _nexmark(13, 20_000, 120_000)
    stages = getattr(S, rule)(_plan(q))
    got, sizes, _ = run_staged(gpu, stages, relations, chunks=2 if rule == "split_at_repartitions" else 1)
    assert _got(got) == _want(q, a, b)
    assert all(c > 0 for i in sizes if sum(sizes[i]) > 1000 for c in sizes[i]), sizes


def _ysb(seed, n_events, campaigns, ads):
    c_ad, camp = oracle.ysb_campaigns(seed, campaigns, ads)
    ad, et = oracle.ysb_events(seed, 0, n_events, campaigns * ads)
    ev = pa.record_batch([_utf8(ad), _utf8(et)], names=["ad_id", "event_type"])
    cp = pa.record_batch([_utf8(c_ad), _utf8(camp)], names=["c_ad_id", "campaign_id"])
    want = oracle.ysb_campaign_counts(ad, et, c_ad, camp)
    return {"ad_event": ev, "campaign": cp}, sorted((k.decode(), int(v)) for k, v in want.items())


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_events,campaigns,ads", [(1, 30_000, 100, 10), (9, 200_000, 7, 3), (2, 40, 3, 2)])
def test_ysb_whole_plan_fused_generic_and_staged(gpu, seed, n_events, campaigns, ads, monkeypatch):
    from flock_amd import stages as S
    relations, want = _ysb(seed, n_events, campaigns, ads)
    text, out = _collect_whole(gpu, _plan("ysb"), relations)
    assert "fused YSB" in text and _rows(out) == want and (len(want) > 0 or n_events < 100)
    for rule in ("build_query_dag", "split_at_repartitions"):
        got, sizes, _ = run_staged(gpu, getattr(S, rule)(_plan("ysb")), relations, chunks=2 if rule == "split_at_repartitions" else 1)
        assert _rows(got) == want, rule
    text, out = _collect_whole(gpu, _plan("ysb"), relations, generic_only=True)
    assert "fused" not in text and _rows(out) == want
