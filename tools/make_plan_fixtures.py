#!/usr/bin/env python3
"""Writes tests/golden/plans/*.json: the physical plans of the five NEXMark target queries (and q7, q13) in the
serde_json dialect of the reference's DataFusion fork.

The fork's serialiser cannot be run here (no Rust toolchain), so the plans are AUTHORED from
  * the grammar of the checked-in fixtures flock/src/tests/data/plan/{simple_select,aggregate,join}.json
    (tag keys "execution_plan" / "physical_expr", field names, "partitioning": {"Hash": [[exprs], n]} ...), and
  * the operator trees + expressions the reference pins as text:
      q1, q2  flock/src/distributed_plan/planner.rs:86-125
      q3      flock/src/distributed_plan/planner.rs:148-171
      q5, q8  playground/src/distributed_plan/nexmark/q5.dag, q8.dag (stages re-joined into one plan)
    with target_partitions = 8 (flock/src/configs/flock.toml:113) and target_batch_size = 4096.
Newer fork revisions print columns as `Column { name, index }` (planner.rs:153): both "name" and "index" are emitted.
"""
import json
import os

P = 8
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "plans")


def field(name, dt, nullable=False):
    return {"data_type": dt, "dict_id": 0, "dict_is_ordered": False, "name": name, "nullable": nullable}


TS = {"Timestamp": ["Millisecond", None]}
BID = [field("auction", "Int32"), field("bidder", "Int32"), field("price", "Int32"), field("b_date_time", TS)]
AUCTION = [field("a_id", "Int32"), field("item_name", "Utf8"), field("description", "Utf8"), field("initial_bid", "Int32"),
           field("reserve", "Int32"), field("a_date_time", TS), field("expires", TS), field("seller", "Int32"),
           field("category", "Int32")]
PERSON = [field("p_id", "Int32"), field("name", "Utf8"), field("email_address", "Utf8"), field("credit_card", "Utf8"),
          field("city", "Utf8"), field("state", "Utf8"), field("p_date_time", TS)]


def schema(fields, name=None):
    return {"fields": fields, "metadata": {"name": name} if name else {}}


def col(name, index):
    return {"physical_expr": "column", "name": name, "index": index}


def lit(kind, v):
    return {"physical_expr": "literal", "value": {kind: v}}


def cast(e, t):
    return {"physical_expr": "cast_expr", "expr": e, "cast_type": t}


def binary(l, op, r):
    return {"physical_expr": "binary_expr", "left": l, "op": op, "right": r}


def memory(fields, projection, name):
    return {"execution_plan": "memory_exec", "schema": schema(fields, name), "projection": projection}


def rr(inp):
    return {"execution_plan": "repartition_exec", "input": inp, "partitioning": {"RoundRobinBatch": P}}


def hashp(inp, exprs):
    return {"execution_plan": "repartition_exec", "input": inp, "partitioning": {"Hash": [exprs, P]}}


def coalesce(inp):
    return {"execution_plan": "coalesce_batches_exec", "input": inp, "target_batch_size": 4096}


def proj(inp, exprs, fields):
    return {"execution_plan": "projection_exec", "expr": [[e, n] for e, n in exprs], "input": inp, "schema": schema(fields)}


def filt(inp, pred):
    return {"execution_plan": "filter_exec", "predicate": pred, "input": inp}


def agg(inp, mode, group, aggrs, in_fields, out_fields):
    return {"execution_plan": "hash_aggregate_exec", "mode": mode, "group_expr": [[e, n] for e, n in group],
            "aggr_expr": aggrs, "input": inp, "input_schema": schema(in_fields), "schema": schema(out_fields),
            "output_rows": {"metric_type": "Counter", "value": 0}}


def join(left, right, on, fields):
    return {"execution_plan": "hash_join_exec", "left": left, "right": right,
            "on": [[col(l, li), col(r, ri)] for (l, li), (r, ri) in on], "join_type": "Inner", "mode": "Partitioned",
            "random_state": {"k0": 0, "k1": 0, "k2": 0, "k3": 0}, "schema": schema(fields)}


def q0():
    # benchmarks/src/nexmark/query/q0.sql `SELECT * FROM bid` and q10.sql `SELECT auction, bidder, price, b_date_time FROM bid` (q0_plan.fmt / q10_plan.fmt:
    # one Projection over the scan of all four columns): the pass-through queries -- what they measure in the reference is the source and the sink
    return proj(rr(memory(BID, [0, 1, 2, 3], "bid")), [(col(f["name"], i), f["name"]) for i, f in enumerate(BID)], BID)


def q1():
    # ProjectionExec: expr=[auction@0, bidder@1, 0.908 * CAST(price@2 AS Float64) as price, b_date_time@3]
    #   RepartitionExec: RoundRobinBatch(8) <- MemoryExec          (planner.rs:90-92)
    out = [field("auction", "Int32"), field("bidder", "Int32"), field("price", "Float64"), field("b_date_time", TS)]
    return proj(rr(memory(BID, [0, 1, 2, 3], "bid")),
                [(col("auction", 0), "auction"), (col("bidder", 1), "bidder"),
                 (binary(lit("Float64", 0.908), "Multiply", cast(col("price", 2), "Float64")), "price"),
                 (col("b_date_time", 3), "b_date_time")], out)


def q2():
    # ProjectionExec [auction@0, price@1] <- CoalesceBatches(4096) <- FilterExec: CAST(auction@0 AS Int64) % 123 = 0
    #   <- RepartitionExec: RoundRobinBatch(8) <- MemoryExec       (planner.rs:120-124)
    f = [field("auction", "Int32"), field("price", "Int32")]
    pred = binary(binary(cast(col("auction", 0), "Int64"), "Modulo", lit("Int64", 123)), "Eq", lit("Int64", 0))
    return proj(coalesce(filt(rr(memory(BID, [0, 2], "bid")), pred)), [(col("auction", 0), "auction"), (col("price", 1), "price")], f)


def q3():
    # planner.rs:152-171
    af = [field("a_id", "Int32"), field("seller", "Int32"), field("category", "Int32")]
    pf = [field("p_id", "Int32"), field("name", "Utf8"), field("city", "Utf8"), field("state", "Utf8")]
    left = coalesce(hashp(coalesce(filt(rr(memory(AUCTION, [0, 7, 8], "auction")),
                                        binary(cast(col("category", 2), "Int64"), "Eq", lit("Int64", 10)))), [col("seller", 1)]))
    st = lambda s: binary(col("state", 3), "Eq", lit("Utf8", s))
    right = coalesce(hashp(coalesce(filt(rr(memory(PERSON, [0, 1, 4, 5], "person")),
                                         binary(binary(st("or"), "Or", st("id")), "Or", st("ca")))), [col("p_id", 0)]))
    j = join(left, right, [(("seller", 1), ("p_id", 0))], af + pf)
    out = [field("name", "Utf8"), field("city", "Utf8"), field("state", "Utf8"), field("a_id", "Int32")]
    return proj(coalesce(j), [(col("name", 4), "name"), (col("city", 5), "city"), (col("state", 6), "state"), (col("a_id", 0), "a_id")], out)


def count_by_auction():
    inp = [field("auction", "Int32")]
    part = [field("auction", "Int32"), field("COUNT(UInt8(1))[count]", "UInt64", True)]
    fin = [field("auction", "Int32"), field("COUNT(UInt8(1))", "UInt64", True)]
    cnt = [{"aggregate_expr": "count", "name": "COUNT(UInt8(1))", "data_type": "UInt64", "nullable": True,
            "expr": lit("UInt8", 1)}]
    partial = agg(rr(memory(BID, [0], "bid")), "Partial", [(col("auction", 0), "auction")], cnt, inp, part)
    return agg(coalesce(hashp(partial, [col("auction", 0)])), "FinalPartitioned", [(col("auction", 0), "auction")], cnt, inp, fin)


def q5():
    # playground/src/distributed_plan/nexmark/q5.dag
    num = [field("auction", "Int32"), field("num", "UInt64", True)]
    left = proj(proj(count_by_auction(), [(col("auction", 0), "auction"), (col("COUNT(UInt8(1))", 1), "num")], num),
                [(col("auction", 0), "auction"), (col("num", 1), "num")], num)
    only_num = [field("num", "UInt64", True)]
    counts = proj(proj(count_by_auction(), [(col("COUNT(UInt8(1))", 1), "num")], only_num), [(col("num", 0), "num")], only_num)
    mx = [{"aggregate_expr": "max", "name": "MAX(CountBids.num)", "data_type": "UInt64", "nullable": True, "expr": col("num", 0)}]
    mxf = [field("MAX(CountBids.num)", "UInt64", True)]
    partial = agg(counts, "Partial", [], mx, only_num, [field("MAX(CountBids.num)[max]", "UInt64", True)])
    final = agg({"execution_plan": "coalesce_partitions_exec", "input": partial}, "Final", [], mx, only_num, mxf)
    maxn = [field("maxn", "UInt64", True)]
    right = proj(proj(final, [(col("MAX(CountBids.num)", 0), "maxn")], maxn), [(col("maxn", 0), "maxn")], maxn)
    j = join(coalesce(hashp(left, [col("num", 1)])), coalesce(hashp(right, [col("maxn", 0)])), [(("num", 1), ("maxn", 0))], num + maxn)
    return proj(coalesce(j), [(col("auction", 0), "auction"), (col("num", 1), "num")], num)


def q8():
    # playground/src/distributed_plan/nexmark/q8.dag
    pf = [field("p_id", "Int32"), field("name", "Utf8")]
    grp_p = [(col("p_id", 0), "p_id"), (col("name", 1), "name")]
    p_partial = agg(rr(memory(PERSON, [0, 1], "person")), "Partial", grp_p, [], pf, pf)
    p_final = agg(coalesce(hashp(p_partial, [col("p_id", 0), col("name", 1)])), "FinalPartitioned", grp_p, [], pf, pf)
    left = proj(proj(p_final, grp_p, pf), grp_p, pf)
    sf = [field("seller", "Int32")]
    grp_s = [(col("seller", 0), "seller")]
    s_partial = agg(rr(memory(AUCTION, [7], "auction")), "Partial", grp_s, [], sf, sf)
    s_final = agg(coalesce(hashp(s_partial, [col("seller", 0)])), "FinalPartitioned", grp_s, [], sf, sf)
    right = proj(proj(s_final, grp_s, sf), grp_s, sf)
    j = join(coalesce(hashp(left, [col("p_id", 0)])), coalesce(hashp(right, [col("seller", 0)])), [(("p_id", 0), ("seller", 0))], pf + sf)
    return proj(coalesce(j), grp_p, pf)


def q7():
    # benchmarks/src/nexmark/query/q7.sql + q7_plan.fmt (SURVEY.md section 8(f) "next" query):
    # Projection [auction, price, bidder, b_date_time] <- HashJoin(price = maxprice)
    #   left : bid                       right: Projection maxprice <- MAX(price) (Partial -> CoalescePartitions -> Final) <- bid
    mx = [{"aggregate_expr": "max", "name": "MAX(bid.price)", "data_type": "Int32", "nullable": True, "expr": col("price", 2)}]
    partial = agg(rr(memory(BID, [0, 1, 2, 3], "bid")), "Partial", [], mx, BID, [field("MAX(bid.price)[max]", "Int32", True)])
    final = agg({"execution_plan": "coalesce_partitions_exec", "input": partial}, "Final", [], mx, BID,
                [field("MAX(bid.price)", "Int32", True)])
    maxp = [field("maxprice", "Int32", True)]
    right = proj(proj(final, [(col("MAX(bid.price)", 0), "maxprice")], maxp), [(col("maxprice", 0), "maxprice")], maxp)
    left = coalesce(hashp(rr(memory(BID, [0, 1, 2, 3], "bid")), [col("price", 2)]))
    j = join(left, coalesce(hashp(right, [col("maxprice", 0)])), [(("price", 2), ("maxprice", 0))], BID + maxp)
    out = [field("auction", "Int32"), field("price", "Int32"), field("bidder", "Int32"), field("b_date_time", TS)]
    return proj(coalesce(j), [(col("auction", 0), "auction"), (col("price", 2), "price"), (col("bidder", 1), "bidder"),
                              (col("b_date_time", 3), "b_date_time")], out)


def q11():
    # benchmarks/src/nexmark/query/q11.sql over the bids of the session windows closed in an epoch (flock-function/src/aws/window/session.rs:
    # 187-321 sends them; the plan below is what `physical_plan(q11.sql)` is for a DataFusion of this fork's vintage, in the dialect and
    # two-phase aggregate shape of q5.dag): Projection <- FinalPartitioned <- Hash([bidder], 8) <- Partial <- RoundRobin <- MemoryExec
    inp = [field("bidder", "Int32"), field("b_date_time", TS)]
    names = ("COUNT(UInt8(1))", "MIN(bid.b_date_time)", "MAX(bid.b_date_time)")
    aggs = [{"aggregate_expr": "count", "name": names[0], "data_type": "UInt64", "nullable": True, "expr": lit("UInt8", 1)},
            {"aggregate_expr": "min", "name": names[1], "data_type": TS, "nullable": True, "expr": col("b_date_time", 1)},
            {"aggregate_expr": "max", "name": names[2], "data_type": TS, "nullable": True, "expr": col("b_date_time", 1)}]
    part = [field("bidder", "Int32"), field(names[0] + "[count]", "UInt64", True), field(names[1] + "[min]", TS, True), field(names[2] + "[max]", TS, True)]
    fin = [field("bidder", "Int32"), field(names[0], "UInt64", True), field(names[1], TS, True), field(names[2], TS, True)]
    grp = [(col("bidder", 0), "bidder")]
    partial = agg(rr(memory(BID, [1, 3], "bid")), "Partial", grp, aggs, inp, part)
    final = agg(coalesce(hashp(partial, [col("bidder", 0)])), "FinalPartitioned", grp, aggs, inp, fin)
    out = [field("bidder", "Int32"), field("bid_count", "UInt64", True), field("start_time", TS, True), field("end_time", TS, True)]
    return proj(final, [(col("bidder", 0), "bidder"), (col(names[0], 1), "bid_count"), (col(names[1], 2), "start_time"), (col(names[2], 3), "end_time")], out)


def q13():
    # benchmarks/src/nexmark/query/q13.sql + q13_plan.fmt (SURVEY.md section 8(f) "next" query):
    # Projection [auction, bidder, price, b_date_time, value] <- HashJoin(auction = key)
    #   left : bid (hash-repartitioned on auction)      right: side_input [key, value] (hash-repartitioned on key)
    side = [field("key", "Int32"), field("value", "Int32")]
    left = coalesce(hashp(rr(memory(BID, [0, 1, 2, 3], "bid")), [col("auction", 0)]))
    right = coalesce(hashp(rr(memory(side, [0, 1], "side_input")), [col("key", 0)]))
    j = join(left, right, [(("auction", 0), ("key", 0))], BID + side)
    out = BID + [field("value", "Int32")]
    return proj(coalesce(j), [(col("auction", 0), "auction"), (col("bidder", 1), "bidder"), (col("price", 2), "price"),
                              (col("b_date_time", 3), "b_date_time"), (col("value", 5), "value")], out)


def winning_bids(with_category):
    """Q of q4.sql / q9.sql: MAX(price) per auction over the bids placed while the auction was open.
    flock/src/distributed_plan/planner.rs:218-256 (q4 stages 0-2; q9 is the same sub-plan without `category`):
      HashAggregateExec FinalPartitioned gby=[a_id(, category)] aggr=[MAX(bid.price)] <- Hash([a_id(, category)]) <- Partial
        <- FilterExec b_date_time@6 >= a_date_time@1 AND b_date_time@6 <= expires@2
        <- HashJoinExec Partitioned on=[(a_id@0, auction@0)] <- Hash([a_id]) <- auction ; Hash([auction]) <- bid"""
    if with_category:
        af = [AUCTION[0], AUCTION[5], AUCTION[6], AUCTION[8]]
        a_scan, b_scan, bf = memory(AUCTION, [0, 5, 6, 8], "auction"), memory(BID, [0, 2, 3], "bid"), [BID[0], BID[2], BID[3]]
    else:
        af = [AUCTION[0], AUCTION[5], AUCTION[6]]
        a_scan, b_scan, bf = memory(AUCTION, [0, 5, 6], "auction"), memory(BID, [0, 1, 2, 3], "bid"), BID
    jf = af + bf
    ix = {f["name"]: i for i, f in enumerate(jf)}
    j = join(coalesce(hashp(rr(a_scan), [col("a_id", 0)])), coalesce(hashp(rr(b_scan), [col("auction", 0)])),
             [(("a_id", 0), ("auction", 0))], jf)
    bdt = col("b_date_time", ix["b_date_time"])
    between = binary(binary(bdt, "GtEq", col("a_date_time", 1)), "And", binary(bdt, "LtEq", col("expires", 2)))
    grp = [(col("a_id", 0), "a_id")] + ([(col("category", 3), "category")] if with_category else [])
    gf = [field("a_id", "Int32")] + ([field("category", "Int32")] if with_category else [])
    mx = [{"aggregate_expr": "max", "name": "MAX(bid.price)", "data_type": "Int32", "nullable": True, "expr": col("price", ix["price"])}]
    partial = agg(coalesce(filt(coalesce(j), between)), "Partial", grp, mx, jf, gf + [field("MAX(bid.price)[max]", "Int32", True)])
    keys = [col("a_id", 0)] + ([col("category", 1)] if with_category else [])
    fgrp = [(col("a_id", 0), "a_id")] + ([(col("category", 1), "category")] if with_category else [])
    return agg(coalesce(hashp(partial, keys)), "FinalPartitioned", fgrp, mx, jf, gf + [field("MAX(bid.price)", "Int32", True)])


def q4():
    # benchmarks/src/nexmark/query/q4.sql; flock/src/distributed_plan/planner.rs:218-256 (stages re-joined into one plan)
    fin = [field("final", "Int32", True), field("category", "Int32")]
    q = proj(proj(winning_bids(True), [(col("MAX(bid.price)", 2), "final"), (col("category", 1), "category")], fin),
             [(col("final", 0), "final"), (col("category", 1), "category")], fin)
    avg = [{"aggregate_expr": "avg", "name": "AVG(Q.final)", "data_type": "Float64", "nullable": True, "expr": col("final", 0)}]
    part = [field("category", "Int32"), field("AVG(Q.final)[count]", "UInt64", True), field("AVG(Q.final)[sum]", "Float64", True)]
    out = [field("category", "Int32"), field("AVG(Q.final)", "Float64", True)]
    partial = agg(q, "Partial", [(col("category", 1), "category")], avg, fin, part)
    final = agg(coalesce(hashp(partial, [col("category", 0)])), "FinalPartitioned", [(col("category", 0), "category")], avg, fin, out)
    return proj(final, [(col("category", 0), "category"), (col("AVG(Q.final)", 1), "AVG(Q.final)")], out)


def q9():
    # benchmarks/src/nexmark/query/q9.sql + q9_plan.fmt: bid JOIN Q ON auction = id AND price = final
    idf = [field("id", "Int32"), field("final", "Int32", True)]
    q = proj(proj(winning_bids(False), [(col("a_id", 0), "id"), (col("MAX(bid.price)", 1), "final")], idf),
             [(col("id", 0), "id"), (col("final", 1), "final")], idf)
    left = coalesce(hashp(rr(memory(BID, [0, 1, 2, 3], "bid")), [col("auction", 0), col("price", 2)]))
    right = coalesce(hashp(q, [col("id", 0), col("final", 1)]))
    j = join(left, right, [(("auction", 0), ("id", 0)), (("price", 2), ("final", 1))], BID + idf)
    return proj(coalesce(j), [(col("auction", 0), "auction"), (col("bidder", 1), "bidder"), (col("price", 2), "price"),
                              (col("b_date_time", 3), "b_date_time")], BID)


def window_row_number(inp, partition, order, name):
    """window_agg_exec with one ROW_NUMBER() (q6_plan.fmt: WindowAggr windowExpr=[[ROW_NUMBER() PARTITION BY [..] ORDER BY [..]]]) over the sort_exec the
    physical planner puts underneath (PARTITION BY keys, then ORDER BY keys).  The serde shape of the fork's WindowAggExec is not in the reference
    tree: this dialect is authored here (a `window_expr` list of {fun, name, partition_by, order_by}) and accepted with tolerance by plan_ir.hpp."""
    srt = sort_by(inp, [(p, False) for p in partition] + list(order))
    return {"execution_plan": "window_agg_exec", "input": srt,
            "window_expr": [{"window_expr": "built_in_window_expr", "fun": "RowNumber", "name": name, "partition_by": list(partition),
                             "order_by": [{"expr": k, "options": {"descending": bool(d), "nulls_first": bool(d)}} for k, d in order]}]}


def q6():
    # benchmarks/src/nexmark/query/q6.sql + q6_plan.fmt (a logical plan; the physical operators as the planner lays them out for the other queries)
    af = [AUCTION[0], AUCTION[5], AUCTION[6], AUCTION[7]]                  # a_id, a_date_time, expires, seller
    bf = [BID[0], BID[2], BID[3]]                                          # auction, price, b_date_time
    jf = af + bf
    j = join(coalesce(hashp(rr(memory(AUCTION, [0, 5, 6, 7], "auction")), [col("a_id", 0)])), coalesce(hashp(rr(memory(BID, [0, 2, 3], "bid")), [col("auction", 0)])),
             [(("a_id", 0), ("auction", 0))], jf)
    bdt = col("b_date_time", 6)
    between = binary(binary(bdt, "GtEq", col("a_date_time", 1)), "And", binary(bdt, "LtEq", col("expires", 2)))
    rn1 = "ROW_NUMBER() PARTITION BY [#auction.a_id] ORDER BY [#bid.price DESC NULLS FIRST]"
    w1 = window_row_number(coalesce(filt(coalesce(j), between)), [col("a_id", 0)], [(col("price", 5), True)], rn1)     # [rn, a_id, a_date_time, expires, seller, auction, price, b_date_time]
    top = filt(w1, binary(cast(col(rn1, 0), "Int64"), "Eq", lit("Int64", 1)))
    qf = [field("seller", "Int32"), field("a_id", "Int32"), field("price", "Int32"), field("b_date_time", TS), field("price_rank", "UInt64", True)]
    q = proj(top, [(col("seller", 4), "seller"), (col("a_id", 1), "a_id"), (col("price", 6), "price"), (col("b_date_time", 7), "b_date_time"), (col(rn1, 0), "price_rank")], qf)
    q = sort_by(q, [(col("a_id", 1), False), (col("price", 2), True)])                                                     # the subquery's own ORDER BY
    qf2 = [qf[0], qf[2], qf[3], qf[4]]
    q = proj(q, [(col("seller", 0), "seller"), (col("price", 2), "price"), (col("b_date_time", 3), "b_date_time"), (col("price_rank", 4), "price_rank")], qf2)
    rn2 = "ROW_NUMBER() PARTITION BY [#Q.seller] ORDER BY [#Q.b_date_time DESC NULLS FIRST]"
    w2 = window_row_number(coalesce(hashp(q, [col("seller", 0)])), [col("seller", 0)], [(col("b_date_time", 2), True)], rn2)   # [rn, seller, price, b_date_time, price_rank]
    recent = filt(w2, binary(cast(col(rn2, 0), "Int64"), "LtEq", lit("Int64", 10)))
    rf = [field("seller", "Int32"), field("price", "Int32"), field("time_rank", "UInt64", True)]
    r = proj(proj(recent, [(col("seller", 1), "seller"), (col("price", 2), "price"), (col(rn2, 0), "time_rank")], rf),
             [(col("seller", 0), "seller"), (col("price", 1), "price"), (col("time_rank", 2), "time_rank")], rf)
    avg = [{"aggregate_expr": "avg", "name": "AVG(R.price)", "data_type": "Float64", "nullable": True, "expr": col("price", 1)}]
    part = [field("seller", "Int32"), field("AVG(R.price)[count]", "UInt64", True), field("AVG(R.price)[sum]", "Float64", True)]
    out = [field("seller", "Int32"), field("AVG(R.price)", "Float64", True)]
    partial = agg(r, "Partial", [(col("seller", 0), "seller")], avg, rf, part)
    final = agg(coalesce(hashp(partial, [col("seller", 0)])), "FinalPartitioned", [(col("seller", 0), "seller")], avg, rf, out)
    return proj(final, [(col("seller", 0), "seller"), (col("AVG(R.price)", 1), "AVG(R.price)")], out)


AD_EVENT = [field("user_id", "Utf8"), field("page_id", "Utf8"), field("ad_id", "Utf8"), field("ad_type", "Utf8"),
            field("event_type", "Utf8"), field("event_time", TS), field("ip_address", "Utf8")]
CAMPAIGN = [field("c_ad_id", "Utf8"), field("campaign_id", "Utf8")]


def ysb():
    # benchmarks/src/ysb/ysb.sql; flock/src/distributed_plan/planner.rs:298-346 (stages re-joined into one plan)
    ev = [AD_EVENT[2], AD_EVENT[4]]
    left = coalesce(hashp(coalesce(filt(rr(memory(AD_EVENT, [2, 4], "ysb_ad_events")), binary(col("event_type", 1), "Eq", lit("Utf8", "view")))),
                          [col("ad_id", 0)]))
    right = coalesce(hashp(rr(memory(CAMPAIGN, [0, 1], "ysb_campaigns")), [col("c_ad_id", 0)]))
    jf = ev + CAMPAIGN
    j = {"execution_plan": "hash_join_exec", "left": left, "right": right, "on": [[col("ad_id", 0), col("c_ad_id", 0)]],
         "join_type": "Inner", "mode": "Partitioned", "random_state": {"k0": 0, "k1": 0, "k2": 0, "k3": 0}, "schema": schema(jf)}
    cnt = [{"aggregate_expr": "count", "name": "COUNT(UInt8(1))", "data_type": "UInt64", "nullable": True, "expr": lit("UInt8", 1)}]
    part = [field("campaign_id", "Utf8"), field("COUNT(UInt8(1))[count]", "UInt64", True)]
    out = [field("campaign_id", "Utf8"), field("COUNT(UInt8(1))", "UInt64", True)]
    partial = agg(coalesce(j), "Partial", [(col("campaign_id", 3), "campaign_id")], cnt, jf, part)
    final = agg(coalesce(hashp(partial, [col("campaign_id", 0)])), "FinalPartitioned", [(col("campaign_id", 0), "campaign_id")], cnt, jf, out)
    return proj(final, [(col("campaign_id", 0), "campaign_id"), (col("COUNT(UInt8(1))", 1), "COUNT(UInt8(1))")], out)


def name_col(name):
    # older fork revisions serialise a column by name only (the dialect of flock/src/tests/data/plan/*.json)
    return {"physical_expr": "column", "name": name}


def golden_aggregate():
    """The plan of the reference's operator-level golden at this boundary (flock/src/runtime/context.rs:430-503:
    `SELECT MAX(c1), MIN(c2), c3 FROM test WHERE c2 < 99 GROUP BY c3 [ORDER BY c3]`), without the final sort, in the shape and
    dialect of the reference's aggregate.json fixture: Projection <- FinalPartitioned <- Hash([c3], 8) <- Partial <- Filter
    (c2 < TRY_CAST(99 AS Float64)) <- RoundRobin <- MemoryExec [0, 1, 2]."""
    f = [field("c1", "Int64"), field("c2", "Float64"), field("c3", "Utf8")]
    aggs = [{"aggregate_expr": "max", "data_type": "Int64", "expr": name_col("c1"), "name": "MAX(c1)", "nullable": True},
            {"aggregate_expr": "min", "data_type": "Float64", "expr": name_col("c2"), "name": "MIN(c2)", "nullable": True}]
    pred = {"physical_expr": "binary_expr", "left": name_col("c2"), "op": "Lt",
            "right": {"physical_expr": "try_cast_expr", "cast_type": "Float64", "expr": lit("Int64", 99)}}
    part = [field("c3", "Utf8"), field("MAX(c1)[max]", "Int64", True), field("MIN(c2)[min]", "Float64", True)]
    out = [field("c3", "Utf8"), field("MAX(c1)", "Int64", True), field("MIN(c2)", "Float64", True)]
    scan = {"execution_plan": "memory_exec", "schema": schema(f), "projection": [0, 1, 2]}
    grp = [[name_col("c3"), "c3"]]
    partial = {"execution_plan": "hash_aggregate_exec", "mode": "Partial", "group_expr": grp, "aggr_expr": aggs, "input": coalesce(filt(rr(scan), pred)),
               "input_schema": schema(f), "schema": schema(part), "output_rows": {"metric_type": "Counter", "value": 0}}
    final = {"execution_plan": "hash_aggregate_exec", "mode": "FinalPartitioned", "group_expr": grp, "aggr_expr": aggs,
             "input": coalesce({"execution_plan": "repartition_exec", "input": partial, "partitioning": {"Hash": [[name_col("c3")], P]}}),
             "input_schema": schema(f), "schema": schema(out), "output_rows": {"metric_type": "Counter", "value": 0}}
    pf = [field("MAX(c1)", "Int64", True), field("MIN(c2)", "Float64", True), field("c3", "Utf8")]
    return {"execution_plan": "projection_exec", "expr": [[name_col("MAX(c1)"), "MAX(c1)"], [name_col("MIN(c2)"), "MIN(c2)"], [name_col("c3"), "c3"]],
            "input": final, "schema": schema(pf)}


def golden_join():
    """The plan of flock/src/runtime/context.rs:505-589 (`SELECT a, b, d FROM t1 JOIN t2 ON a = c [ORDER BY a LIMIT 3]`) below
    its sort + limit, in the shape and dialect of the reference's join.json: Projection <- HashJoin on [["a", "c"]] (Utf8 keys,
    bare names) <- Hash([a], 8) <- t1 ; Hash([c], 8) <- t2."""
    t1 = [field("a", "Utf8"), field("b", "Int32")]
    t2 = [field("c", "Utf8"), field("d", "Int32")]
    side = lambda f, k: coalesce({"execution_plan": "repartition_exec", "partitioning": {"Hash": [[name_col(k)], P]},
                                  "input": rr({"execution_plan": "memory_exec", "schema": schema(f), "projection": [0, 1]})})
    j = {"execution_plan": "hash_join_exec", "left": side(t1, "a"), "right": side(t2, "c"), "on": [["a", "c"]], "join_type": "Inner",
         "mode": "Partitioned", "random_state": {"k0": 0, "k1": 0, "k2": 0, "k3": 0}, "schema": schema(t1 + t2)}
    return {"execution_plan": "projection_exec", "expr": [[name_col("a"), "a"], [name_col("b"), "b"], [name_col("d"), "d"]], "input": coalesce(j),
            "schema": schema([t1[0], t1[1], t2[1]])}


def sort_by(inp, keys):
    """sort_exec in the dialect of the reference's join.json: [{"expr": column, "options": {"descending", "nulls_first"}}]."""
    return {"execution_plan": "sort_exec", "input": inp,
            "expr": [{"expr": k, "options": {"descending": bool(desc), "nulls_first": bool(desc)}} for k, desc in keys],
            "output_rows": {"metric_type": "Counter", "value": 0}, "sort_time_nanos": {"metric_type": "TimeNanos", "value": 0}}


def golden_aggregate_sorted():
    """context.rs:471 whole: `... GROUP BY c3 ORDER BY c3` = SortExec over the merged partitions of golden_aggregate()."""
    return sort_by({"execution_plan": "merge_exec", "input": golden_aggregate()}, [(name_col("c3"), False)])


def golden_join_sorted():
    """context.rs:544-551 whole (the shape of the reference's join.json): GlobalLimit 3 <- Sort [a ASC] <- Merge <- golden_join()."""
    return {"execution_plan": "global_limit_exec", "limit": 3,
            "input": sort_by({"execution_plan": "merge_exec", "input": golden_join()}, [(name_col("a"), False)])}


def q3_sorted():
    """launcher/aws/mod.rs:340-351: q3 with `ORDER BY a_id ASC` (the reference's distributed == local differential recipe)."""
    return sort_by({"execution_plan": "coalesce_partitions_exec", "input": q3()}, [(col("a_id", 3), False)])


def arch_groupby():
    """flock-function/src/aws/arch/ops/group-by.sql: `SELECT auction, Count(*) FROM bid GROUP BY auction` -- the two-phase aggregate of
    q5's inner query under its projection."""
    out = [field("auction", "Int32"), field("COUNT(UInt8(1))", "UInt64", True)]
    return proj(count_by_auction(), [(col("auction", 0), "auction"), (col("COUNT(UInt8(1))", 1), "COUNT(UInt8(1))")], out)


def arch_join():
    """flock-function/src/aws/arch/ops/join.sql: `SELECT * FROM auction INNER JOIN bid ON a_id = auction` -- HashJoinExec Partitioned,
    both sides hash-repartitioned on the key, every column of both relations in the output (two Utf8 columns among them)."""
    left = coalesce(hashp(rr(memory(AUCTION, list(range(9)), "auction")), [col("a_id", 0)]))
    right = coalesce(hashp(rr(memory(BID, [0, 1, 2, 3], "bid")), [col("auction", 0)]))
    j = join(left, right, [(("a_id", 0), ("auction", 0))], AUCTION + BID)
    return proj(coalesce(j), [(col(f["name"], i), f["name"]) for i, f in enumerate(AUCTION + BID)], AUCTION + BID)


def arch_sort():
    """flock-function/src/aws/arch/ops/sort.sql: `SELECT * FROM bid ORDER BY bidder`."""
    return sort_by({"execution_plan": "coalesce_partitions_exec", "input": memory(BID, [0, 1, 2, 3], "bid")}, [(col("bidder", 1), False)])


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in (("golden_aggregate", golden_aggregate), ("golden_join", golden_join), ("golden_aggregate_sorted", golden_aggregate_sorted),
                     ("golden_join_sorted", golden_join_sorted), ("q3_sorted", q3_sorted)):
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fn(), f, indent=1, sort_keys=True)
            f.write("\n")
    # the reference's operator harness (flock-function/src/aws/arch/ops/*.sql, source.rs:25-65); filter.sql is q2's statement
    for name, fn in (("arch_filter", q2), ("arch_groupby", arch_groupby), ("arch_join", arch_join), ("arch_sort", arch_sort)):
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fn(), f, indent=1, sort_keys=True)
            f.write("\n")
    for name, fn in (("q0", q0), ("q10", q0), ("q1", q1), ("q2", q2), ("q3", q3), ("q5", q5), ("q8", q8), ("q7", q7), ("q13", q13), ("q4", q4), ("q9", q9), ("q11", q11), ("ysb", ysb), ("q6", q6)):
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fn(), f, indent=1, sort_keys=True)
            f.write("\n")
    # shapes of the reference's own fixtures (flock/src/tests/data/plan/simple_select.json, join.json), authored here in the
    # same dialect (not copies of the reference files): a pure projection of an Int64 column -- executable by the generic
    # operators --, the same under sort + limit (device operators since round 4), and a plan the engine must hand back as
    # UNSUPPORTED (a LEFT join)
    simple = proj(rr(memory([field("c1", "Int64")], [0], None)), [(col("c1", 0), "c1")], [field("c1", "Int64")])
    with open(os.path.join(OUT, "simple_select.json"), "w") as f:
        json.dump(simple, f, indent=1, sort_keys=True)
        f.write("\n")
    sort_limit = {"execution_plan": "global_limit_exec", "limit": 3,
                  "input": {"execution_plan": "sort_exec", "input": simple,
                            "expr": [{"expr": col("c1", 0), "options": {"descending": False, "nulls_first": False}}]}}
    with open(os.path.join(OUT, "sort_limit.json"), "w") as f:
        json.dump(sort_limit, f, indent=1, sort_keys=True)
        f.write("\n")
    left_join = golden_join()
    left_join["input"]["input"]["join_type"] = "Left"
    with open(os.path.join(OUT, "unsupported_left_join.json"), "w") as f:
        json.dump(left_join, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
