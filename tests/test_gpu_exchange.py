"""GPU side of the key-partitioned exchange: the HIP partition / take kernels against the numpy restatement used by the
CPU gloo tests (tests/test_distributed.py), and the exchange-mode queries end to end through RCCL on one rank."""
import socket

import numpy as np
import pytest

import oracle
from test_distributed import NumpyOps, _strs, mix32, part_of  # noqa: F401  (same stand-ins as the CPU tests)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def world1():
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def _dev(a):
    from devmem import dev
    return dev(a)          # (a torch tensor; guarded memory under FLOCK_TEST_GUARDED=1: tests/test_gpu_guard.py)


def _utf8(u):
    from flock_amd import DeviceUtf8
    data = u.data if len(u.data) >= 16 else np.concatenate([u.data, np.zeros(16 - len(u.data), np.uint8)])
    return DeviceUtf8(_dev(u.offsets), _dev(data))


@pytest.mark.parametrize("n_parts", [1, 2, 3, 5, 8, 9, 13, 37, 64])   # (groups of eight destinations per trip: whole, ragged, one past a group)
def test_partition_matches_restatement(ctx, n_parts):
    import torch
    from flock_amd import WindowSchedule
    rng = np.random.default_rng(n_parts)
    n = 200_003
    keys = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    keys[1000:60_000] = 4242                                     # hot key: one destination gets a burst
    offs = np.array([0, 5, 5, 8200, 8200 + 8192 * 3, 150_001, n])  # ragged, unaligned, empty, multi-tile windows
    sched = WindowSchedule(offs, np.array([0, 1, 2, 3, 5]), np.array([1, 2, 3, 5, 6]))   # window 3 spans two panes
    rows, counts = ctx.partition_by_key(_dev(keys), sched, n_parts)
    want_rows, want_counts = NumpyOps().partition(torch.from_numpy(keys), sched, n_parts)
    assert np.array_equal(counts, want_counts)
    assert np.array_equal(rows.cpu().numpy(), want_rows.numpy())


def test_partition_empty_and_errors(ctx):
    from flock_amd import FlockGpuError, WindowSchedule
    rows, counts = ctx.partition_by_key(_dev(np.zeros(4, np.int32))[:0], WindowSchedule(np.array([0, 0]), np.array([0]), np.array([1])), 4)
    assert rows.numel() == 0 and counts.tolist() == [[0], [0], [0], [0]]
    with pytest.raises(FlockGpuError):
        ctx.partition_by_key(_dev(np.zeros(16, np.int32)), WindowSchedule.single(16), 0)
    with pytest.raises(FlockGpuError):
        ctx.partition_by_key(_dev(np.zeros(16, np.int32)), WindowSchedule.single(16), 65)


def test_take_matches_numpy(ctx):
    rng = np.random.default_rng(5)
    src32 = rng.integers(-2**31, 2**31 - 1, 100_000, dtype=np.int64).astype(np.int32)
    src64 = rng.integers(-2**62, 2**62, 100_000, dtype=np.int64)
    rows = rng.integers(0, 100_000, 250_001).astype(np.int32)
    assert np.array_equal(ctx.take(_dev(src32), _dev(rows)).cpu().numpy(), src32[rows])
    assert np.array_equal(ctx.take(_dev(src64), _dev(rows)).cpu().numpy(), src64[rows])
    # Utf8 incl. empty strings, strings longer than the 16-byte fast path and longer than a staging buffer
    vals = [b"", b"a", b"or", b"x" * 13, b"y" * 16, b"z" * 17, b"w" * 40, bytes(range(1, 200)), b"q" * 30_000]
    pick = rng.integers(0, len(vals), 5000)
    strs = [vals[i] for i in pick]
    off = np.concatenate(([0], np.cumsum([len(s) for s in strs]))).astype(np.int32)
    col = oracle.Utf8(off, np.frombuffer(b"".join(strs), np.uint8).copy())
    take_rows = rng.integers(0, 5000, 12_345).astype(np.int32)
    got = ctx.take_utf8(_utf8(col), _dev(take_rows))
    want = oracle.take_utf8(col, take_rows)
    assert np.array_equal(got.offsets.cpu().numpy(), want.offsets)
    assert np.array_equal(got.data.cpu().numpy()[: len(want.data)], want.data)
    empty = ctx.take_utf8(_utf8(col), _dev(take_rows)[:0])
    assert empty.offsets.cpu().numpy().tolist() == [0]
    lens = _dev(np.diff(want.offsets).astype(np.int32))
    assert np.array_equal(ctx.offsets_from_lengths(lens).cpu().numpy(), want.offsets)


def test_exchange_queries_one_rank(ctx, world1):
    """q3 / q5 / q8 through partition -> take -> RCCL all_to_all -> regroup -> local operator, world size 1."""
    from flock_amd import NEXMarkSource, Window
    from exchange_model import q3_exchange, q5_exchange, q8_exchange
    seed, eps, seconds = 21, 20_000, 20
    g = NEXMarkSource(seconds, eps, Window.tumbling(10), seed=seed).generate_data(ctx)
    host = oracle.NexmarkStream(seed=seed, eps=eps)
    n = eps * seconds
    au, pe, bi = host.auctions(0, n), host.persons(0, n), host.bids(0, n, columns=("auction",))["auction"]

    o8 = q8_exchange(ctx, g.persons, g.window_schedule("person"), g.auctions, g.window_schedule("auction")).to_host()
    sp, sa = g.window_schedule("person"), g.window_schedule("auction")
    names = _strs(*o8["name"])
    for w in range(2):
        (plo, phi), (alo, ahi) = sp.window_rows(w), sa.window_rows(w)
        rows = oracle.q8_join(pe["p_id"][plo:phi], pe["name"].slice(plo, phi), au["seller"][alo:ahi])
        want = sorted(zip(pe["p_id"][plo:phi][rows].tolist(), _strs(pe["name"].offsets, pe["name"].data, [plo + r for r in rows])))
        sl = slice(o8["offsets"][w], o8["offsets"][w + 1])
        assert sorted(zip(o8["p_id"][sl].tolist(), names[sl])) == want and want

    ew = Window.element_wise()
    o3 = q3_exchange(ctx, g.auctions, g.window_schedule("auction", ew), g.persons, g.window_schedule("person", ew)).to_host()
    sa, sp = g.window_schedule("auction", ew), g.window_schedule("person", ew)
    nm, ci, stt = _strs(*o3["name"]), _strs(*o3["city"]), _strs(*o3["state"])
    total = 0
    for w in range(seconds):
        (alo, ahi), (plo, phi) = sa.window_rows(w), sp.window_rows(w)
        ar, pr = oracle.q3_join(au["seller"][alo:ahi], au["category"][alo:ahi], pe["p_id"][plo:phi], pe["state"].slice(plo, phi))
        rows = [plo + int(r) for r in pr]
        want = sorted(zip(_strs(pe["name"].offsets, pe["name"].data, rows), _strs(pe["city"].offsets, pe["city"].data, rows),
                          _strs(pe["state"].offsets, pe["state"].data, rows), au["a_id"][alo:ahi][ar].tolist()))
        sl = slice(o3["offsets"][w], o3["offsets"][w + 1])
        assert sorted(zip(nm[sl], ci[sl], stt[sl], o3["a_id"][sl].tolist())) == want
        total += len(want)
    assert total > 0

    hop = Window.hopping(10, 5)
    sb = g.window_schedule("bid", hop)
    r5 = q5_exchange(ctx, g.bids, sb)
    for w in range(sb.n_windows):
        lo, hi = sb.window_rows(w)
        oa, on = oracle.q5_hot_items(bi[lo:hi])
        sl = slice(r5.offsets[w], r5.offsets[w + 1])
        assert sorted(zip(r5.auction[sl].tolist(), r5.num[sl].tolist())) == sorted(zip(oa.tolist(), on.tolist()))
        assert int(r5.win_max[w]) == int(on[0])


def test_scans_beyond_one_pass(ctx):
    """More than 16384 tiles: the tile scan runs chunk by chunk (chunk-local bases, scan of the chunk totals, fix-up)."""
    import torch
    from flock_amd import WindowSchedule
    rng = np.random.default_rng(9)
    lens = rng.integers(0, 21, 40_000_000).astype(np.int32)            # 19.5k scan tiles of 2048 values
    got = ctx.offsets_from_lengths(_dev(lens)).cpu().numpy()
    want = np.concatenate(([0], np.cumsum(lens, dtype=np.int64))).astype(np.int32)
    assert np.array_equal(got, want)
    n = 20_000_003                                                     # 2442 tiles x 8 destinations = 19.5k pseudo-tiles
    keys = rng.integers(0, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    offs = np.array([0, 7_000_001, 7_000_001, n])
    sched = WindowSchedule(offs, np.arange(3), np.arange(1, 4))
    rows, counts = ctx.partition_by_key(_dev(keys), sched, 8)
    want_rows, want_counts = NumpyOps().partition(torch.from_numpy(keys), sched, 8)
    assert np.array_equal(counts, want_counts) and np.array_equal(rows.cpu().numpy(), want_rows.numpy())


# ------------------------------------------------------------------ q5.dag's two aggregation stages around the repartition
def _q5_cases(rng, n):
    return {
        "generator": oracle.NexmarkStream(seed=11, eps=20_000).bids(0, 50 * 20_000, columns=("auction",))["auction"][:n],
        "uniform_narrow": rng.integers(1000, 9000, n).astype(np.int32),          # wider than the LDS histogram: direct path
        "uniform_int32": rng.integers(-2**31, 2**31 - 1, n).astype(np.int32),    # hash tables only
        "few_keys": rng.integers(0, 7, n).astype(np.int32),
        "sorted_runs": (np.arange(n) // 13 - 5).astype(np.int32),
    }


@pytest.mark.parametrize("case", ["generator", "uniform_narrow", "uniform_int32", "few_keys", "sorted_runs"])
def test_q5_partial_then_weighted_equals_hot_items(ctx, case):
    """partial counts per pane == np.unique per pane; weighted hot items over those groups (all panes, hopping windows) ==
    hot items over the rows; and the same with the groups of two row-stripes concatenated pane by pane (what a partition
    receives from two sources: a key may arrive twice inside a pane)."""
    import torch
    from flock_amd import Bids, WindowSchedule
    rng = np.random.default_rng(7)
    n = 600_000
    a = _q5_cases(rng, n)[case]
    n = len(a)
    pane_off = np.array([0, 3, 3, 100_001, 250_000, 250_000 + 8192 * 20, n])
    lo, hi = np.array([0, 1, 2, 3, 0], np.int32), np.array([2, 3, 4, 6, 6], np.int32)   # overlapping windows, one over everything
    sched = WindowSchedule(pane_off, lo, hi)
    key, cnt, poff = ctx.q5_partial_counts(Bids(auction=_dev(a), rows=n), sched)
    k, c = key.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32)
    for p in range(len(pane_off) - 1):
        wk, wc = np.unique(a[pane_off[p]:pane_off[p + 1]], return_counts=True)
        got = sorted(zip(k[poff[p]:poff[p + 1]].tolist(), c[poff[p]:poff[p + 1]].tolist()))
        assert got == list(zip(wk.tolist(), wc.tolist())), (case, p)
    want = ctx.q5_hot_items(Bids(auction=_dev(a), rows=n), sched)
    wa, wn, woff = want.to_host()
    got = ctx.q5_hot_items_weighted(key, cnt, WindowSchedule(poff, lo, hi))
    ga, gn, goff = got.to_host()
    assert np.array_equal(ga, wa) and np.array_equal(gn, wn) and np.array_equal(goff, woff)
    assert np.array_equal(got.win_max(), want.win_max()) and np.array_equal(got.win_groups(), want.win_groups())
    for w in range(len(lo)):   # ... and against the oracle
        oa, on = oracle.q5_hot_items(a[pane_off[lo[w]]:pane_off[hi[w]]])
        o = np.argsort(oa)   # ties: the engine orders a window's winners by auction
        assert np.array_equal(ga[goff[w]:goff[w + 1]], oa[o]) and np.array_equal(gn[goff[w]:goff[w + 1]], on[o]), (case, w)
    # two stripes of every pane, aggregated separately, concatenated pane by pane
    mid = (pane_off[:-1] + pane_off[1:]) // 2
    idx0 = np.concatenate([np.arange(pane_off[p], mid[p]) for p in range(len(mid))])
    idx1 = np.concatenate([np.arange(mid[p], pane_off[p + 1]) for p in range(len(mid))])
    parts = []
    for idx, off in ((idx0, np.r_[0, np.cumsum(mid - pane_off[:-1])]), (idx1, np.r_[0, np.cumsum(pane_off[1:] - mid)])):
        kk, cc, po = ctx.q5_partial_counts(Bids(auction=_dev(a[idx]), rows=len(idx)), WindowSchedule(off, lo, hi))
        parts.append((kk.cpu().numpy(), cc.cpu().numpy(), po))
    cat_k = np.concatenate([np.r_[parts[0][0][parts[0][2][p]:parts[0][2][p + 1]], parts[1][0][parts[1][2][p]:parts[1][2][p + 1]]]
                            for p in range(len(mid))])
    cat_c = np.concatenate([np.r_[parts[0][1][parts[0][2][p]:parts[0][2][p + 1]], parts[1][1][parts[1][2][p]:parts[1][2][p + 1]]]
                            for p in range(len(mid))])
    cat_off = parts[0][2] + parts[1][2]
    g2 = ctx.q5_hot_items_weighted(_dev(cat_k.astype(np.int32)), _dev(cat_c.astype(np.int32)), WindowSchedule(cat_off, lo, hi))
    ga2, gn2, goff2 = g2.to_host()
    assert np.array_equal(ga2, wa) and np.array_equal(gn2, wn) and np.array_equal(goff2, woff), case


def test_q5_exchange_world1_uses_partial_groups(ctx, world1):
    """q5_exchange end to end on one rank through RCCL: identical to the single-GPU operator."""
    from flock_amd import Bids, NEXMarkSource, Window
    from exchange_model import q5_exchange
    w = Window.hopping(10, 5)
    g = NEXMarkSource(40, 30_000, w, seed=5).generate_data(ctx, relations=("bid",), bid_columns=("auction",))
    sched = g.window_schedule("bid", w)
    shard = q5_exchange(ctx, g.bids, sched)
    a, n, off = ctx.q5_hot_items(g.bids, sched).to_host()
    assert np.array_equal(shard.auction, a) and np.array_equal(shard.num, n) and np.array_equal(shard.offsets, off)


def test_q5_weighted_rejects_counts_that_overflow_32_bits(ctx):
    """rows of the weighted entry carry a count each, so a window can reach 2^32 with few rows; the 32-bit counters
    (and the count half of the packed table slots) would wrap -- the call must say UNSUPPORTED instead."""
    from flock_amd import FlockGpuError, WindowSchedule, _ffi
    n = 64
    key = _dev(np.arange(1000, 1000 + n, dtype=np.int32))
    big = _dev(np.full(n, 2**27, np.int32))                                  # (uint32 counts travel as int32 bits)
    sched = WindowSchedule(np.array([0, n // 2, n]), np.array([0, 0], np.int32), np.array([1, 2], np.int32))   # window 1 = everything: 2^33
    with pytest.raises(FlockGpuError) as e:
        ctx.q5_hot_items_weighted(key, big, sched)
    assert e.value.code == _ffi.ERR_UNSUPPORTED and "2^32" in str(e.value)
    ok = ctx.q5_hot_items_weighted(key, big, WindowSchedule(np.array([0, 16]), np.array([0], np.int32), np.array([1], np.int32)))   # 16 * 2^27 = 2^31
    a, cnt, off = ok.to_host()
    assert len(a) == 16 and (cnt == 2**27).all()


@pytest.mark.parametrize("n_parts", [1, 4, 8])
def test_partition_of_two_hundred_million_rows_by_its_properties(ctx, n_parts):
    """RepartitionExec Hash([key], n) at the size of the BASELINE exchange (2e8 keys over 216 hopping-pane windows; one destination, the
    destination-by-destination write-out and the one-pass write-out), checked through what does not depend on the size: every group (destination,
    window) holds rows of its window only, in ascending order (so no row twice), whose keys hash to the destination; the group sizes are the
    histogram of the destinations per window and add up to the rows."""
    import torch
    from flock_amd import WindowSchedule
    n, n_win = 200_000_003, 216
    g = torch.Generator(device="cuda").manual_seed(20 + n_parts)
    keys = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
    offs = np.linspace(0, n, n_win + 1).astype(np.int64)
    offs[1:-1] += 1                                                       # unaligned window starts
    sched = WindowSchedule(offs, np.arange(n_win), np.arange(1, n_win + 1))
    rows, counts = ctx.partition_by_key(keys, sched, n_parts)
    assert rows.numel() == n and int(counts.sum()) == n
    # the destinations, restated in torch (test_distributed.mix32 / part_of)
    x = keys.to(torch.int64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    part = (x * n_parts) >> 32
    del x
    win_of_row = torch.bucketize(torch.arange(n, device="cuda"), torch.from_numpy(offs[1:]).cuda(), right=True)
    want = torch.bincount(part * n_win + win_of_row, minlength=n_parts * n_win).reshape(n_parts, n_win).cpu().numpy()
    assert np.array_equal(counts, want)
    group_of_pos = torch.repeat_interleave(torch.arange(n_parts * n_win, device="cuda"), torch.from_numpy(counts.reshape(-1)).cuda())
    r64 = rows.to(torch.int64)
    assert bool((part[r64] == group_of_pos // n_win).all())               # every row sits with its destination ...
    assert bool((win_of_row[r64] == group_of_pos % n_win).all())          # ... and its window
    same = group_of_pos[1:] == group_of_pos[:-1]
    assert bool((r64[1:][same] > r64[:-1][same]).all())                   # input order inside a group: ascending, hence no row twice


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 8191, 8192, 8193, 20_000, 32_767, 32_768, 32_769, 70_001])
def test_inclusive_scan_around_the_one_launch_limit(ctx, n):
    """`flockgpu_inclusive_scan_i32` through `offsets_from_lengths`: inputs up to 32768 values take one workgroup walking 8192 values a round
    (gather.hip scan_small_kernel), larger ones the three-launch scan -- sizes either side of a round, of a wave and of the limit."""
    rng = np.random.default_rng(n)
    lens = rng.integers(0, 50, n).astype(np.int32)
    got = ctx.offsets_from_lengths(_dev(lens)).cpu().numpy()
    assert np.array_equal(got, np.concatenate(([0], np.cumsum(lens, dtype=np.int64))).astype(np.int32))
