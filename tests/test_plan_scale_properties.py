"""Round-6 operators at sizes no row-by-row oracle finishes in seconds, checked through properties that do not depend on the size (the tier's
rule for full-size parity): the hashed join's pair count and key / payload checksums, ORDER BY's sortedness, stability and column checksums,
the long-text take's byte count and per-row lengths -- each against numpy on the same seeded inputs."""
import numpy as np
import pyarrow as pa
import pytest

from test_plan_round5 import _field, _join_plan


@pytest.fixture(scope="module")
def gpu():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_hashed_join_of_twenty_million_rows_by_its_checksums(gpu):
    """2e7 probe rows against 1.5e6 unique build keys spread over the Int32 range (arch/ops/join.sql's sizes, ids scrambled): every probe row whose key
    is a build key pairs exactly once; the output's row count, the sums of both key columns (equal, row by row) and of both payload columns are numpy's."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(2026)
    nb, npr = 1_500_000, 20_000_000
    build = (np.arange(nb, dtype=np.uint32) * np.uint32(2654435761)).view(np.int32)          # a bijection of 0 .. nb - 1: unique, full range
    x = r.integers(-1000, 1000, nb).astype(np.int32)
    pick = r.integers(0, 2 * nb, npr)                                                            # half the probe rows name a build key
    probe = (pick.astype(np.uint32) * np.uint32(2654435761)).view(np.int32)
    y = r.integers(-2**40, 2**40, npr).astype(np.int64)
    hit = pick < nb
    lf = [_field("a", "Int32", False), _field("x", "Int32", False)]
    rf = [_field("b", "Int32", False), _field("y", "Int64", False)]
    lb = [pa.record_batch([pa.array(build), pa.array(x)], names=["a", "x"])]
    rb = [pa.record_batch([pa.array(probe), pa.array(y)], names=["b", "y"])]
    ctx = ExecutionContext([_join_plan(lf, rf, "a", "b")], gpu=gpu)
    gpu.profile_reset()
    gpu.profile(True)
    try:
        out = collect(ctx, [[lb], [rb]])[0][0]
        ran = gpu.profile_read()
    finally:
        gpu.profile(False)
        ctx.close()
    assert "join_hash_probe_flag_kernel" in ran, sorted(ran)
    assert out.num_rows == int(hit.sum())
    a, xo, b, yo = (out.column(i).to_numpy() for i in range(4))
    assert np.array_equal(a, b)                                                                  # every pair joins equal keys
    assert int(a.astype(np.int64).sum()) == int(probe[hit].astype(np.int64).sum())
    assert int(yo.sum()) == int(y[hit].sum()) and int(xo.astype(np.int64).sum()) == int(x[pick[hit]].astype(np.int64).sum())
    assert np.array_equal(b, probe[hit]) and np.array_equal(yo, y[hit])                        # pairs come in probe order


@pytest.mark.gpu
def test_order_by_of_thirty_million_rows_is_sorted_stable_and_complete(gpu):
    """`SELECT * ORDER BY k` over 3e7 rows (sort.sql's shape): the key column ascends, rows of equal keys keep their input order (the row-number column
    ascends inside every run), and every column's sum is the input's -- a permutation."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(77)
    n = 30_000_000
    k = r.integers(1000, 2_000_000, n).astype(np.int32)
    rown = np.arange(n, dtype=np.int32)
    p = r.integers(0, 10_000_000, n).astype(np.int32)
    t = r.integers(0, 2**50, n).astype(np.int64)
    f = [_field("rown", "Int32", False), _field("k", "Int32", False), _field("p", "Int32", False), _field("t", "Int64", False)]
    scan_ = {"execution_plan": "memory_exec", "schema": {"fields": f, "metadata": {}}, "projection": [0, 1, 2, 3]}
    plan = {"execution_plan": "sort_exec", "input": scan_, "expr": [{"expr": {"physical_expr": "column", "name": "k", "index": 1}, "options": {"descending": False, "nulls_first": False}}]}
    rb = [pa.record_batch([pa.array(rown), pa.array(k), pa.array(p), pa.array(t)], names=["rown", "k", "p", "t"])]
    ctx = ExecutionContext([plan], gpu=gpu)
    try:
        out = collect(ctx, [[rb]])[0][0]
    finally:
        ctx.close()
    ro, ko, po, to = (out.column(i).to_numpy() for i in range(4))
    assert out.num_rows == n and np.all(np.diff(ko.astype(np.int64)) >= 0)
    same = np.diff(ko) == 0
    assert np.all(np.diff(ro.astype(np.int64))[same] > 0)                                          # stable: ties in input order
    assert np.array_equal(ko, k[ro]) and np.array_equal(po, p[ro]) and np.array_equal(to, t[ro])   # every row carries its own columns
    assert int(ro.astype(np.int64).sum()) == n * (n - 1) // 2


@pytest.mark.gpu
def test_long_text_take_of_five_million_values_by_lengths_and_bytes(gpu):
    """A filter's take of 5e6 of 1e7 text values of 40-110 bytes (0.75 GB in): the output's offsets are the running sum of the kept rows' lengths and its
    bytes are the kept values' bytes, end to end."""
    from flock_amd.runtime import ExecutionContext, collect
    from test_plan_round5 import binary, lit
    r = np.random.default_rng(5)
    n = 10_000_000
    lens = r.integers(40, 111, n).astype(np.int32)
    off = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=off[1:])
    data = r.integers(97, 123, int(off[-1]), dtype=np.uint8)
    j = r.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    text = pa.StringArray.from_buffers(n, pa.py_buffer(off.tobytes()), pa.py_buffer(data.tobytes()))
    f = [_field("j", "Int32", False), _field("s", "Utf8", False)]
    scan_ = {"execution_plan": "memory_exec", "schema": {"fields": f, "metadata": {}}, "projection": [0, 1]}
    pred = binary({"physical_expr": "column", "name": "j", "index": 0}, "Gt", lit("Int32", 0))
    ctx = ExecutionContext([{"execution_plan": "filter_exec", "predicate": pred, "input": scan_}], gpu=gpu)
    gpu.profile_reset()
    gpu.profile(True)
    try:
        out = collect(ctx, [[[pa.record_batch([pa.array(j), text], names=["j", "s"])]]])[0][0]
        ran = gpu.profile_read()
    finally:
        gpu.profile(False)
        ctx.close()
    assert "utf8_emit_long_kernel" in ran, sorted(ran)
    keep = np.nonzero(j > 0)[0]
    assert out.num_rows == len(keep) and np.array_equal(out.column(0).to_numpy(), j[keep])
    s = out.column(1)
    bufs = s.buffers()
    got_off = np.frombuffer(bufs[1], np.int32, len(keep) + 1, s.offset * 4)
    assert np.array_equal(np.diff(got_off), lens[keep])
    got = np.frombuffer(bufs[2], np.uint8, int(got_off[-1] - got_off[0]), int(got_off[0]))
    want = np.concatenate([data[off[i]:off[i + 1]] for i in keep[:2000]])                          # the first values byte by byte ...
    assert np.array_equal(got[:len(want)], want)
    idx = np.repeat(off[keep], lens[keep]) + (np.arange(int(lens[keep].sum())) - np.repeat(np.cumsum(lens[keep]) - lens[keep], lens[keep]))
    assert np.array_equal(got, data[idx])                                                         # ... and all of them
