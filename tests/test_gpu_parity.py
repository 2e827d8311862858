"""GPU parity tests: the HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.
Bit-exact: all outputs are integer / byte / index work, and q1's f64 column is one IEEE multiply.
Row order: q2 is compared exactly (FilterExec keeps input order); join / aggregate outputs are
compared as sorted multisets per window (the reference's convention, flock/src/test_util.rs:61-90)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _dev(a):
    from devmem import dev
    return dev(a)          # (a torch tensor; guarded memory under FLOCK_TEST_GUARDED=1: tests/test_gpu_guard.py)


def _utf8(u):
    from flock_amd import DeviceUtf8
    data = u.data if len(u.data) else np.zeros(16, np.uint8)
    return DeviceUtf8(_dev(u.offsets), _dev(data))


def _host_stream(seed, eps, seconds, first=0):
    s = oracle.NexmarkStream(seed=seed, eps=eps, first_event_id=first)
    n = eps * seconds
    return s, s.bids(0, n), s.auctions(0, n), s.persons(0, n)


def _gpu_stream(ctx, seed, eps, seconds, window, first=0):
    from devmem import guard_stream
    from flock_amd import NEXMarkSource
    return guard_stream(NEXMarkSource(seconds, eps, window, seed=seed, first_event_id=first).generate_data(ctx))


def _str_rows(off, data, rows=None):
    b = data.tobytes()
    idx = range(len(off) - 1) if rows is None else rows
    return [b[off[i]:off[i + 1]] for i in idx]


CASES = [(1, 1000, 3), (7, 5000, 4), (42, 50_000, 12), (5, 1_000_000, 2)]


# ------------------------------------------------------------------ generator: device == oracle, bit for bit
@pytest.mark.parametrize("seed,eps,seconds", CASES)
@pytest.mark.parametrize("first", [0, 123_450])
def test_device_generator_matches_oracle(ctx, seed, eps, seconds, first):
    from flock_amd import Window
    g = _gpu_stream(ctx, seed, eps, seconds, Window.element_wise(), first)
    _, b, a, p = _host_stream(seed, eps, seconds, first)
    for k in ("auction", "bidder", "price", "b_date_time"):
        assert np.array_equal(getattr(g.bids, k).cpu().numpy(), b[k]), k
    for k in ("a_id", "seller", "category"):
        assert np.array_equal(getattr(g.auctions, k).cpu().numpy(), a[k]), k
    assert np.array_equal(g.persons.p_id.cpu().numpy(), p["p_id"])
    for k in ("name", "city", "state"):
        col = getattr(g.persons, k)
        off = col.offsets.cpu().numpy()
        assert np.array_equal(off, p[k].offsets), k
        assert np.array_equal(col.data.cpu().numpy()[: off[-1]], p[k].data), k


# ------------------------------------------------------------------ q1 / q2
@pytest.mark.parametrize("seed,eps,seconds", CASES)
def test_q1_projection_bit_exact(ctx, seed, eps, seconds):
    from flock_amd import Window, run_query
    g = _gpu_stream(ctx, seed, eps, seconds, Window.element_wise())
    got = run_query(ctx, 1, g).cpu().numpy()
    want = oracle.q1_project(g.bids.price.cpu().numpy())
    assert got.dtype == np.float64 and got.tobytes() == want.tobytes()


def test_profile_samples_are_the_launches_behind_the_totals(ctx):
    """flockgpu_profile_samples (round 5): the per-launch durations of one kernel, in launch order, add up to flockgpu_profile_read's total;
    `profile_only` takes "a|b" for a step that runs one of several kernels; a kernel that never ran has no samples."""
    from flock_amd import Window, run_query
    g = _gpu_stream(ctx, 5, 200_000, 3, Window.element_wise())
    ctx.profile_reset()
    ctx.profile_only("q2_flag_kernel|no_such_kernel")
    ctx.profile(True)
    try:
        for _ in range(5):
            run_query(ctx, 2, g)
        st = ctx.profile_read()
        smp = ctx.profile_samples("q2_flag_kernel")
    finally:
        ctx.profile(False)
        ctx.profile_only(None)
    assert sorted(st) == ["q2_flag_kernel"] and st["q2_flag_kernel"]["launches"] == 5 == len(smp)
    assert all(0 < x < 50 for x in smp) and abs(sum(smp) - st["q2_flag_kernel"]["total_ms"]) < 1e-3
    assert ctx.profile_samples("no_such_kernel") == []
    ctx.profile_reset()
    assert ctx.profile_samples("q2_flag_kernel") == []


@pytest.mark.parametrize("seed,eps,seconds", CASES)
def test_q2_filter_exact_per_epoch(ctx, seed, eps, seconds):
    from flock_amd import Window, run_query
    g = _gpu_stream(ctx, seed, eps, seconds, Window.element_wise())
    a, p, off = run_query(ctx, 2, g).to_host()
    ha, hp = g.bids.auction.cpu().numpy(), g.bids.price.cpu().numpy()
    sched = g.window_schedule("bid")
    assert len(off) == seconds + 1 and off[0] == 0
    for w in range(seconds):
        lo, hi = sched.window_rows(w)
        wa, wp = oracle.q2_filter(ha[lo:hi], hp[lo:hi])
        assert np.array_equal(a[off[w]:off[w + 1]], wa), w
        assert np.array_equal(p[off[w]:off[w + 1]], wp), w
    assert off[-1] == len(a)


def test_q2_bursty_selectivity_and_other_moduli(ctx):
    # hot auction id == 0 (mod m) makes >50 % of a window pass; also negative ids / moduli (truncated remainder)
    from flock_amd import Bids, WindowSchedule
    rng = np.random.default_rng(3)
    n = 300_000
    auction = rng.integers(-5000, 5000, n).astype(np.int32)
    auction[1000:200_000:2] = 1230                     # burst
    auction[5] = np.iinfo(np.int32).min
    price = rng.integers(100, 10**8, n).astype(np.int32)
    bids = Bids(auction=_dev(auction), price=_dev(price), rows=n)
    offs = np.array([0, 17, 17, 4099, 150_001, n])     # ragged, unaligned, one empty window
    sched = WindowSchedule(offs, np.arange(5), np.arange(1, 6))
    for m in (123, 7, 1, -123, 124, 96, -64, 2**31, 2**31 - 1, 2**40):
        a, p, off = ctx.q2_filter(bids, sched, modulus=m).to_host()
        for w in range(5):
            wa, wp = oracle.q2_filter(auction[offs[w]:offs[w + 1]], price[offs[w]:offs[w + 1]], modulus=m)
            assert np.array_equal(a[off[w]:off[w + 1]], wa), (m, w)
            assert np.array_equal(p[off[w]:off[w + 1]], wp), (m, w)


def test_q2_overlapping_windows_and_empty_input(ctx):
    from flock_amd import Bids, FlockGpuError, WindowSchedule
    auction = (np.arange(10_000, dtype=np.int32) * 3) % 1000
    price = np.arange(10_000, dtype=np.int32)
    bids = Bids(auction=_dev(auction), price=_dev(price), rows=10_000)
    sched = WindowSchedule(np.array([0, 2500, 5000, 7500, 10_000]), np.array([0, 1, 2]), np.array([2, 3, 4]))
    a, p, off = ctx.q2_filter(bids, sched).to_host()
    for w, (lo, hi) in enumerate([(0, 5000), (2500, 7500), (5000, 10_000)]):
        wa, wp = oracle.q2_filter(auction[lo:hi], price[lo:hi])
        assert np.array_equal(a[off[w]:off[w + 1]], wa) and np.array_equal(p[off[w]:off[w + 1]], wp)
    empty = Bids(auction=_dev(np.zeros(4, np.int32)), price=_dev(np.zeros(4, np.int32)), rows=0)
    r = ctx.q2_filter(empty, WindowSchedule(np.array([0, 0]), np.array([0]), np.array([1])))
    assert r.rows == 0 and r.offsets().tolist() == [0, 0]
    with pytest.raises(FlockGpuError):
        ctx.q2_filter(bids, sched, modulus=0)        # DataFusion raises divide-by-zero
    with pytest.raises(FlockGpuError):
        ctx.q2_filter(bids, WindowSchedule(np.array([0, 20_000]), np.array([0]), np.array([1])))


# ------------------------------------------------------------------ q3
@pytest.mark.parametrize("seed,eps,seconds", CASES)
def test_q3_join_per_epoch(ctx, seed, eps, seconds):
    _check_q3_nexmark(ctx, seed, eps, seconds)


def _check_q3_nexmark(ctx, seed, eps, seconds):
    from flock_amd import Window, run_query
    g = _gpu_stream(ctx, seed, eps, seconds, Window.element_wise())
    out = run_query(ctx, 3, g).to_host()
    _, _, a, p = _host_stream(seed, eps, seconds)
    sa, sp = g.window_schedule("auction"), g.window_schedule("person")
    names, cities, states = (_str_rows(p[k].offsets, p[k].data) for k in ("name", "city", "state"))
    g_name, g_city, g_state = (_str_rows(*out[k]) for k in ("name", "city", "state"))
    off, total = out["offsets"], 0
    for w in range(seconds):
        (alo, ahi), (plo, phi) = sa.window_rows(w), sp.window_rows(w)
        ar, pr = oracle.q3_join(a["seller"][alo:ahi], a["category"][alo:ahi], p["p_id"][plo:phi], p["state"].slice(plo, phi))
        want = sorted((names[plo + j], cities[plo + j], states[plo + j], int(a["a_id"][alo + i])) for i, j in zip(ar, pr))
        sl = slice(off[w], off[w + 1])
        got = sorted(zip(g_name[sl], g_city[sl], g_state[sl], out["a_id"][sl].tolist()))
        assert got == want, w
        # the row-pair view must agree too, and stay inside the window
        assert sorted(zip((out["auction_row"][sl] - alo).tolist(), (out["person_row"][sl] - plo).tolist())) == sorted(zip(ar.tolist(), pr.tolist()))
        total += len(want)
    assert total == off[-1] == len(out["a_id"])
    if eps >= 5000:
        assert total > 0


def test_q3_duplicate_keys_emit_every_pair(ctx):
    from flock_amd import Auctions, Persons, WindowSchedule
    rng = np.random.default_rng(11)
    na, npn = 20_000, 3_000
    seller = rng.integers(0, 50, na).astype(np.int32)            # heavy duplicates on the probe side
    category = rng.integers(10, 12, na).astype(np.int32)
    a_id = np.arange(na, dtype=np.int32) + 1000
    p_id = rng.integers(0, 60, npn).astype(np.int32)             # and on the build side
    st = rng.choice(np.array([b"or", b"OR", b"id", b"ca", b"wa", b"o", b"cal"], dtype=object), npn)
    s_off = np.concatenate([[0], np.cumsum([len(x) for x in st])]).astype(np.int32)
    s_data = np.frombuffer(b"".join(st), np.uint8).copy()
    state = oracle.Utf8(s_off, s_data)
    name = oracle.Utf8(np.arange(npn + 1, dtype=np.int32) * 3, np.frombuffer(b"".join(b"n%02d" % (i % 100) for i in range(npn)), np.uint8).copy())
    aw = WindowSchedule(np.array([0, 7001, 7001, na]), np.arange(3), np.arange(1, 4))
    pw = WindowSchedule(np.array([0, 1000, 1003, npn]), np.arange(3), np.arange(1, 4))
    out = ctx.q3_join(Auctions(_dev(a_id), _dev(seller), _dev(category), na), aw,
                      Persons(_dev(p_id), _utf8(name), _utf8(name), _utf8(state), npn), pw).to_host()
    off = out["offsets"]
    for w in range(3):
        (alo, ahi), (plo, phi) = aw.window_rows(w), pw.window_rows(w)
        ar, pr = oracle.q3_join(seller[alo:ahi], category[alo:ahi], p_id[plo:phi], state.slice(plo, phi))
        sl = slice(off[w], off[w + 1])
        assert sorted(zip((out["auction_row"][sl] - alo).tolist(), (out["person_row"][sl] - plo).tolist())) == sorted(zip(ar.tolist(), pr.tolist())), w
    assert off[2] - off[1] == 0 and off[-1] > 10_000


# ------------------------------------------------------------------ q5
def _q5_check(ctx, g, sched, auction_host):
    a, n, off = ctx.q5_hot_items(g, sched).to_host()
    assert n.dtype == np.uint64 and a.dtype == np.int32
    for w in range(sched.n_windows):
        lo, hi = sched.window_rows(w)
        wa, wn = oracle.q5_hot_items(auction_host[lo:hi])
        assert sorted(zip(a[off[w]:off[w + 1]].tolist(), n[off[w]:off[w + 1]].tolist())) == sorted(zip(wa.tolist(), wn.tolist())), w
    return a, n, off


@pytest.mark.parametrize("seed,eps,seconds", [(1, 1000, 30), (7, 5000, 23), (42, 50_000, 27), (5, 1_000_000, 20)])
def test_q5_hopping_windows(ctx, seed, eps, seconds):
    from flock_amd import Window, query_window
    g = _gpu_stream(ctx, seed, eps, seconds, query_window(5))
    sched = g.window_schedule("bid")
    assert sched.n_windows == len(oracle.hopping_windows(seconds, 10, 5))
    _q5_check(ctx, g.bids, sched, g.bids.auction.cpu().numpy())
    # a second call reuses the arena and the learned table size
    _q5_check(ctx, g.bids, sched, g.bids.auction.cpu().numpy())
    # other window shapes on the same data: tumbling and a 3-pane hop
    _q5_check(ctx, g.bids, g.window_schedule("bid", Window.tumbling(7)), g.bids.auction.cpu().numpy())
    _q5_check(ctx, g.bids, g.window_schedule("bid", Window.hopping(6, 2)), g.bids.auction.cpu().numpy())


def test_q5_ties_uniform_keys_and_extremes(ctx):
    from flock_amd import Bids, WindowSchedule
    rng = np.random.default_rng(5)
    n = 400_000
    cases = {
        "all_distinct": np.arange(n, dtype=np.int32),                       # every key ties with count 1
        "one_key": np.full(n, 77, np.int32),
        "uniform": rng.integers(-2**31, 2**31 - 1, n).astype(np.int32),     # LDS table overflows -> spill path
        "two_way_tie": np.tile(np.array([5, -9], np.int32), n // 2),
        "with_zero_key": np.where(rng.random(n) < 0.5, 0, rng.integers(0, 4, n)).astype(np.int32),
    }
    offs = np.array([0, 1, 8193, 200_000, 200_000, n])
    sched = WindowSchedule(offs, np.arange(5), np.arange(1, 6))
    for name, auction in cases.items():
        a, nn, off = _q5_check(ctx, Bids(auction=_dev(auction), rows=n), sched, auction)
        assert off[4] == off[3], name          # empty window -> MAX is NULL -> no rows


@pytest.mark.parametrize("case", ["even_at_limit", "even_over", "odd_over", "both_over"])
def test_q5_counts_around_the_16_bit_counter_limit(case):
    """Round 6: the bid path counts into 16-bit counters, two keys per word, and checks every pane's counter sum against its rows; a count
    of 65535 fits, 65536 carries into the odd neighbour (even key) or out of the word (odd key) -- the call is then repeated with 32-bit
    counters.  A fresh context per case, so that no earlier fallback hides the 16-bit pass; two calls each (the second one runs with
    whatever the first one left: cleaned counters, the kept counter width)."""
    from flock_amd import Bids, GpuContext, WindowSchedule
    rng = np.random.default_rng(16)
    even, odd = {"even_at_limit": (65535, 9), "even_over": (65536, 9), "odd_over": (40, 65536), "both_over": (70_000, 131_073)}[case]
    n_other = 150_000
    auction = np.concatenate([np.full(even, 1000, np.int32), np.full(odd, 1001, np.int32), (1002 + rng.integers(0, 3000, n_other)).astype(np.int32)])
    rng.shuffle(auction)
    n = len(auction)
    half = n // 2 + 5
    sched = WindowSchedule(np.array([0, half, n]), np.array([0, 1]), np.array([2, 2]))   # panes [0, half), [half, n); windows {0, 1}, {1}: the hopping shape
    c = GpuContext(0)
    try:
        for _ in range(2):
            _q5_check(c, Bids(auction=_dev(auction), rows=n), sched, auction)
    finally:
        c.close()


# ------------------------------------------------------------------ q7 (first "next" query, SURVEY.md section 8 f)
@pytest.mark.parametrize("seed,eps,seconds", [(1, 1000, 30), (7, 5000, 20), (42, 50_000, 30), (5, 1_000_000, 20)])
def test_q7_tumbling_windows(ctx, seed, eps, seconds):
    from flock_amd import query_window, run_query
    g = _gpu_stream(ctx, seed, eps, seconds, query_window(7))
    r = run_query(ctx, 7, g)
    out, mx = r.to_host(), r.win_max()
    _, b, _, _ = _host_stream(seed, eps, seconds)
    sched = g.window_schedule("bid")
    off = out["offsets"]
    assert sched.n_windows == seconds // 10
    for w in range(sched.n_windows):
        lo, hi = sched.window_rows(w)
        rows = oracle.q7_highest_bid(b["price"][lo:hi]) + lo
        sl = slice(off[w], off[w + 1])
        for k in ("auction", "price", "bidder", "b_date_time"):
            assert np.array_equal(out[k][sl], b[k][rows]), (w, k)            # input order is kept: exact equality
        assert int(mx[w]) == int(b["price"][lo:hi].max())
    assert off[-1] == len(out["price"]) > 0


def test_q7_ties_negatives_and_empty_windows(ctx):
    from flock_amd import Bids, FlockGpuError, WindowSchedule
    rng = np.random.default_rng(8)
    n = 100_003
    price = rng.integers(-1000, 1000, n).astype(np.int32)
    price[40_000:40_500] = 5000                       # 500-way tie across tile boundaries in window 2
    price[3] = price[16] = 999_999                    # tie inside window 0, whose rows start unaligned
    price[90_000:] = np.iinfo(np.int32).min           # window 4: every row is the (minimal) maximum
    auction = rng.integers(0, 10**6, n).astype(np.int32)
    bidder = rng.integers(0, 10**6, n).astype(np.int32)
    when = rng.integers(0, 2**62, n)
    bids = Bids(_dev(auction), _dev(bidder), _dev(price), _dev(when), n)
    offs = np.array([1, 17, 17, 60_001, 90_000, n])   # ragged, unaligned, one empty window
    sched = WindowSchedule(offs, np.arange(5), np.arange(1, 6))
    r = ctx.q7_highest_bid(bids, sched)
    out, off = r.to_host(), r.offsets()
    for w in range(5):
        lo, hi = sched.window_rows(w)
        rows = oracle.q7_highest_bid(price[lo:hi]) + lo
        sl = slice(off[w], off[w + 1])
        assert np.array_equal(out["auction"][sl], auction[rows]) and np.array_equal(out["price"][sl], price[rows])
        assert np.array_equal(out["bidder"][sl], bidder[rows]) and np.array_equal(out["b_date_time"][sl], when[rows])
    assert off[2] == off[1] and off[5] - off[4] == n - 90_000 and off[3] - off[2] == 500
    empty = ctx.q7_highest_bid(Bids(_dev(auction)[:0], _dev(bidder)[:0], _dev(price)[:0], _dev(when)[:0], 0),
                               WindowSchedule(np.array([0, 0]), np.array([0]), np.array([1])))
    assert empty.rows == 0 and empty.offsets().tolist() == [0, 0]
    with pytest.raises(FlockGpuError):
        ctx.q7_highest_bid(Bids(auction=_dev(auction), price=_dev(price), rows=n), sched)     # q7 projects all four columns


# ------------------------------------------------------------------ q4 / q9 ("next" queries: join + BETWEEN + MAX / AVG)
def _gpu_stream_times(ctx, seed, eps, seconds, window):
    from devmem import guard_stream
    from flock_amd import NEXMarkSource
    return guard_stream(NEXMarkSource(seconds, eps, window, seed=seed).generate_data(ctx, relations=("bid", "auction"), auction_times=True))


@pytest.mark.parametrize("seed,eps,seconds", [(1, 1000, 5), (7, 5000, 4), (42, 50_000, 6), (5, 1_000_000, 2)])
def test_q4_q9_per_epoch(ctx, seed, eps, seconds):
    from flock_amd import Window, run_query
    g = _gpu_stream_times(ctx, seed, eps, seconds, Window.element_wise())
    s = oracle.NexmarkStream(seed=seed, eps=eps)
    a, b = s.auctions(0, eps * seconds), s.bids(0, eps * seconds)
    assert np.array_equal(g.auctions.a_date_time.cpu().numpy(), a["a_date_time"])      # device generator == oracle
    assert np.array_equal(g.auctions.expires.cpu().numpy(), a["expires"])
    o9, o4 = run_query(ctx, 9, g).to_host(), run_query(ctx, 4, g).to_host()
    sa, sb = g.window_schedule("auction"), g.window_schedule("bid")
    total = 0
    for w in range(seconds):
        (alo, ahi), (blo, bhi) = sa.window_rows(w), sb.window_rows(w)
        args = (a["a_id"][alo:ahi], a["a_date_time"][alo:ahi], a["expires"][alo:ahi], b["auction"][blo:bhi], b["price"][blo:bhi],
                b["b_date_time"][blo:bhi])
        rows = oracle.q9_winning_bids(*args) + blo
        sl = slice(o9["offsets"][w], o9["offsets"][w + 1])
        for k in ("auction", "price", "bidder", "b_date_time"):
            assert np.array_equal(o9[k][sl], b[k][rows]), (w, k)                        # input order: exact equality
        cats, avg = oracle.q4_avg_final_by_category(a["a_id"][alo:ahi], a["category"][alo:ahi], *args[1:])
        sl4 = slice(o4["offsets"][w], o4["offsets"][w + 1])
        assert np.array_equal(o4["category"][sl4], cats) and o4["avg"][sl4].tobytes() == avg.tobytes()   # Float64 bits
        total += len(rows)
    assert total == len(o9["price"]) and (total > 0 or eps < 5000)


def test_q4_q9_edge_cases(ctx):
    """Ties, bids outside [a_date_time, expires], bids on auctions of other windows, negative prices, wide tiles (a tile's
    auction rows spanning more than the LDS table), empty windows on either side, and the unsupported (unsorted) input."""
    from flock_amd import Auctions, Bids, FlockGpuError, WindowSchedule
    rng = np.random.default_rng(17)
    na, nb = 50_000, 400_000
    a_id = (1000 + np.cumsum(rng.integers(1, 4, na))).astype(np.int32)           # strictly increasing, gaps
    a_time = np.sort(rng.integers(0, 10_000, na)).astype(np.int64)
    expires = a_time + rng.integers(0, 50, na)
    category = rng.integers(-3, 9, na).astype(np.int32)                          # 12 values: registers AND the LDS accumulators
    pick = rng.integers(0, na, nb)                                               # uniform over all auctions: wide tiles
    pick[5000:30_000:2] = 123                                                    # a hot auction
    auction = a_id[pick].astype(np.int32)
    when = a_time[pick] + rng.integers(-10, 70, nb)                              # some before a_date_time, some after expires
    auction[::9] = rng.integers(-2**31, 2**31 - 1, len(auction[::9])).astype(np.int32)
    price = rng.integers(-500, 500, nb).astype(np.int32)                         # many ties
    bidder = rng.integers(0, 10**6, nb).astype(np.int32)
    aw = WindowSchedule(np.array([0, 20_001, 20_001, 35_000, na]), np.arange(4), np.arange(1, 5))    # window 1: no auctions
    bw = WindowSchedule(np.array([3, 150_000, 200_000, 200_000, nb]), np.arange(4), np.arange(1, 5))  # window 2: no bids
    auc = Auctions(a_id=_dev(a_id), category=_dev(category), rows=na, a_date_time=_dev(a_time), expires=_dev(expires))
    bids = Bids(_dev(auction), _dev(bidder), _dev(price), _dev(when), nb)
    o9, o4 = ctx.q9_winning_bids(auc, aw, bids, bw).to_host(), ctx.q4_avg_final_by_category(auc, aw, bids, bw).to_host()
    total = 0
    for w in range(4):
        (alo, ahi), (blo, bhi) = aw.window_rows(w), bw.window_rows(w)
        args = (a_time[alo:ahi], expires[alo:ahi], auction[blo:bhi], price[blo:bhi], when[blo:bhi])
        rows = oracle.q9_winning_bids(a_id[alo:ahi], *args) + blo
        sl = slice(o9["offsets"][w], o9["offsets"][w + 1])
        assert np.array_equal(o9["auction"][sl], auction[rows]) and np.array_equal(o9["price"][sl], price[rows]), w
        assert np.array_equal(o9["bidder"][sl], bidder[rows]) and np.array_equal(o9["b_date_time"][sl], when[rows]), w
        cats, avg = oracle.q4_avg_final_by_category(a_id[alo:ahi], category[alo:ahi], *args)
        sl4 = slice(o4["offsets"][w], o4["offsets"][w + 1])
        assert np.array_equal(o4["category"][sl4], cats) and o4["avg"][sl4].tobytes() == avg.tobytes(), w
        total += len(rows)
    assert total > 1000 and o9["offsets"][2] == o9["offsets"][1] and o9["offsets"][3] == o9["offsets"][2]
    shuffled = a_id.copy()
    shuffled[10:20] = shuffled[10:20][::-1]
    with pytest.raises(FlockGpuError) as e:
        ctx.q9_winning_bids(Auctions(a_id=_dev(shuffled), rows=na, a_date_time=_dev(a_time), expires=_dev(expires)), aw, bids, bw)
    from flock_amd import _ffi
    assert e.value.code == _ffi.ERR_UNSUPPORTED


# ------------------------------------------------------------------ q13 ("next" query: bounded side-input join)
@pytest.mark.parametrize("n_side,dups", [(0, False), (1, False), (5000, True), (8191, False), (20_000, True)])
def test_q13_side_input_join(ctx, n_side, dups):
    """Side tables that fit the LDS copy (<= 16384 slots) and ones that do not; duplicate keys; keys no bid has."""
    from flock_amd import Window
    seed, eps, seconds = 31, 200_000, 3
    g = _gpu_stream(ctx, seed, eps, seconds, Window.element_wise())
    _, b, _, _ = _host_stream(seed, eps, seconds)
    rng = np.random.default_rng(n_side + 1)
    lo_k, hi_k = int(b["auction"].min()) - 50, int(b["auction"].max()) + 50
    key = rng.integers(lo_k, hi_k, n_side).astype(np.int32) if dups else \
        rng.choice(np.arange(lo_k, hi_k, dtype=np.int64), n_side, replace=False).astype(np.int32)
    value = rng.integers(-2**31, 2**31 - 1, n_side).astype(np.int32)
    side_k = _dev(key) if n_side else _dev(np.zeros(4, np.int32))[:0]
    side_v = _dev(value) if n_side else _dev(np.zeros(4, np.int32))[:0]
    sched = g.window_schedule("bid")
    out = ctx.q13_side_join(g.bids, sched, side_k, side_v).to_host()
    off, total = out["offsets"], 0
    for w in range(seconds):
        lo, hi = sched.window_rows(w)
        br, sr = oracle.q13_side_join(b["auction"][lo:hi], key)
        sl = slice(off[w], off[w + 1])
        assert sorted(zip((out["bid_row"][sl] - lo).tolist(), out["side_row"][sl].tolist())) == sorted(zip(br.tolist(), sr.tolist())), w
        assert (np.diff(out["bid_row"][sl].astype(np.int64)) >= 0).all()              # bid order is kept
        rows = out["bid_row"][sl]
        for k in ("auction", "bidder", "price", "b_date_time"):
            assert np.array_equal(out[k][sl], b[k][rows]), (w, k)
        assert np.array_equal(out["value"][sl], value[out["side_row"][sl]])
        total += len(br)
    assert total == len(out["value"]) and (total > 0 or n_side <= 1)


# ------------------------------------------------------------------ q8
@pytest.mark.parametrize("seed,eps,seconds", [(1, 1000, 30), (7, 5000, 20), (42, 50_000, 30), (5, 1_000_000, 10)])
def test_q8_tumbling_windows(ctx, seed, eps, seconds):
    _check_q8_nexmark(ctx, seed, eps, seconds)


def _check_q8_nexmark(ctx, seed, eps, seconds):
    from flock_amd import query_window, run_query
    g = _gpu_stream(ctx, seed, eps, seconds, query_window(8))
    out = run_query(ctx, 8, g).to_host()
    _, _, a, p = _host_stream(seed, eps, seconds)
    sp, sa = g.window_schedule("person"), g.window_schedule("auction")
    names = _str_rows(p["name"].offsets, p["name"].data)
    g_names = _str_rows(*out["name"])
    off = out["offsets"]
    assert sp.n_windows == seconds // 10
    for w in range(sp.n_windows):
        (plo, phi), (alo, ahi) = sp.window_rows(w), sa.window_rows(w)
        rows = oracle.q8_join(p["p_id"][plo:phi], p["name"].slice(plo, phi), a["seller"][alo:ahi])
        want = sorted((int(p["p_id"][plo + r]), names[plo + r]) for r in rows)
        sl = slice(off[w], off[w + 1])
        assert sorted(zip(out["p_id"][sl].tolist(), g_names[sl])) == want, w
    assert off[-1] == len(out["p_id"]) > 0


def test_q8_duplicates_collapse(ctx):
    from flock_amd import Auctions, Persons, WindowSchedule
    rng = np.random.default_rng(2)
    npn, na = 50_000, 80_000
    p_id = rng.integers(0, 5000, npn).astype(np.int32)
    tag = rng.integers(0, 3, npn)
    nm = [b"x%d" % t if t else b"" for t in tag]                 # same id + same name -> duplicate; empty names too
    n_off = np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.int32)
    name = oracle.Utf8(n_off, np.frombuffer(b"".join(nm) or b"\0", np.uint8).copy())
    seller = rng.integers(2000, 9000, na).astype(np.int32)
    pw = WindowSchedule(np.array([0, 20_000, npn]), np.arange(2), np.arange(1, 3))
    aw = WindowSchedule(np.array([0, 30_000, na]), np.arange(2), np.arange(1, 3))
    out = ctx.q8_join(Persons(_dev(p_id), _utf8(name), None, None, npn), pw,
                      Auctions(None, _dev(seller), None, na), aw).to_host()
    g_names = _str_rows(*out["name"])
    off = out["offsets"]
    for w in range(2):
        (plo, phi), (alo, ahi) = pw.window_rows(w), aw.window_rows(w)
        rows = oracle.q8_join(p_id[plo:phi], name.slice(plo, phi), seller[alo:ahi])
        want = sorted((int(p_id[plo + r]), nm[plo + r]) for r in rows)
        sl = slice(off[w], off[w + 1])
        assert sorted(zip(out["p_id"][sl].tolist(), g_names[sl])) == want, w


def test_q5_keys_in_no_order_take_the_partitioned_count():
    """Keys spread over the whole pane range make every tile wider than the fast kernel's LDS histogram.  The first such call on a ctx
    goes through the general per-tile kernel and notes it; from the second call on the pane's rows are partitioned by key range and
    counted in LDS (q5_part_tile / q5_bucket_count).  Exact either way -- hot keys, keys outside the sampled range and
    ragged panes included -- and a ctx that sees time-ordered keys again goes back to the fast kernel."""
    from flock_amd import Bids, GpuContext, WindowSchedule
    c = GpuContext(0)
    rng = np.random.default_rng(17)
    pane_rows = [600_001, 450_000, 0, 700_003, 512_000]
    offs = np.concatenate(([0], np.cumsum(pane_rows)))
    n = int(offs[-1])
    auction = np.empty(n, np.int32)
    for p in range(len(pane_rows)):
        lo, hi = offs[p], offs[p + 1]
        base = 1000 + 250_000 * p
        a = rng.integers(base, base + 300_000, hi - lo)
        a[rng.random(hi - lo) < 0.4] = base + 77                      # a hot auction
        if hi > lo:
            # strangers far outside any estimate, on rows the range sampling does not look at (q5_range_kernel reads the 4-row groups
            # whose index is a multiple of `stride`): a sampled stranger makes the pane's range unaffordable and the call falls back
            # to hash tables altogether -- also exact, but not what this test is about
            al = int(lo) & ~3
            stride = max(((int(hi) - al + 3) >> 2) // 2048, 1)
            assert stride >= 2
            rows = al + 4 * (stride * rng.integers(1, 2000, 50) + 1) + 2
            a[rows[(rows >= lo) & (rows < hi)] - lo] = rng.integers(-2**31, 2**31 - 1, int(((rows >= lo) & (rows < hi)).sum()))
        auction[lo:hi] = a
    sched = WindowSchedule(offs, np.arange(0, 4, dtype=np.int32), np.arange(2, 6, dtype=np.int32))   # Hopping over 2 panes, stride 1
    bids = Bids(auction=_dev(auction), rows=n)
    for call in range(3):                                                # general kernel, then the partitioned count twice
        _q5_check(c, bids, sched, auction)
    stats_before = None
    c.profile_reset()
    c.profile(True)
    _q5_check(c, bids, sched, auction)
    stats_before = c.profile_read()
    c.profile(False)
    assert "q5_bucket_count_kernel" in stats_before and "q5_count_kernel" not in stats_before, sorted(stats_before)
    # time-ordered keys again (the generator's shape): the sample of the partition's tiles is narrow, the next call is the fast kernel's
    ordered = np.sort(rng.integers(1000, 61_000, 1_000_000)).astype(np.int32)        # ~16 bids per auction, in auction order: an 8192-row tile spans ~500 ids
    half = len(ordered) // 2
    sched2 = WindowSchedule(np.array([0, half, len(ordered)]), np.array([0], np.int32), np.array([2], np.int32))
    b2 = Bids(auction=_dev(ordered), rows=len(ordered))
    _q5_check(c, b2, sched2, ordered)
    c.profile_reset()
    c.profile(True)
    _q5_check(c, b2, sched2, ordered)
    st = c.profile_read()
    c.profile(False)
    assert "q5_count_kernel" in st and "q5_bucket_count_kernel" not in st, sorted(st)
    c.close()


# ------------------------------------------------------------------ dense / general path selection (q3, q8)
def _sorted_keys(rng, n, spread, lo=-50_000):
    """n strictly increasing int32 keys with an average gap of `spread`."""
    return (lo + np.cumsum(rng.integers(1, 2 * spread, n))).astype(np.int32)


@pytest.mark.parametrize("case", ["dense_gaps", "wide_tiles", "sparse", "unsorted", "mixed_windows"])
def test_q8_every_path_is_exact(ctx, case):
    """Strictly increasing p_ids over an affordable range take the bitmap path (staged in LDS, or direct when a tile's
    keys span more than 32768 ids); sparse / unsorted keys, or any window of them, take the hash path.  Sellers may
    lie outside every window's key range, be negative, or be absent; windows may be empty on either side."""
    from flock_amd import Auctions, Persons, WindowSchedule
    rng = np.random.default_rng({"dense_gaps": 1, "wide_tiles": 2, "sparse": 3, "unsorted": 4, "mixed_windows": 5}[case])
    npn, na = 60_000, 150_000
    spread = {"dense_gaps": 2, "wide_tiles": 30, "sparse": 40_000, "unsorted": 2, "mixed_windows": 2}[case]
    p_id = _sorted_keys(rng, npn, spread)
    if case == "unsorted":
        p_id[20_500:20_600] = p_id[20_500:20_600][::-1]
    if case == "mixed_windows":
        p_id[45_000:] = rng.integers(0, 2**31 - 1, npn - 45_000).astype(np.int32)      # last window: random keys
    nm = [b"p%d" % (i % 977) for i in range(npn)]
    name = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.int32),
                       np.frombuffer(b"".join(nm), np.uint8).copy())
    seller = rng.choice(p_id, na).astype(np.int32)
    seller[::7] = rng.integers(-2**31, 2**31 - 1, len(seller[::7])).astype(np.int32)    # strangers
    seller[1000:40_000:3] = p_id[777]                                                    # a hot seller
    pw = WindowSchedule(np.array([0, 20_000, 20_000, 45_000, npn]), np.arange(4), np.arange(1, 5))   # window 1: no persons
    aw = WindowSchedule(np.array([0, 70_001, 90_000, 90_000, na]), np.arange(4), np.arange(1, 5))    # window 2: no auctions
    out = ctx.q8_join(Persons(_dev(p_id), _utf8(name), None, None, npn), pw, Auctions(None, _dev(seller), None, na), aw).to_host()
    g_names, off = _str_rows(*out["name"]), out["offsets"]
    total = 0
    for w in range(4):
        (plo, phi), (alo, ahi) = pw.window_rows(w), aw.window_rows(w)
        rows = oracle.q8_join(p_id[plo:phi], name.slice(plo, phi), seller[alo:ahi])
        want = sorted((int(p_id[plo + r]), nm[plo + r]) for r in rows)
        sl = slice(off[w], off[w + 1])
        assert sorted(zip(out["p_id"][sl].tolist(), g_names[sl])) == want, (case, w)
        total += len(want)
    assert total == len(out["p_id"]) and total > 1000 and off[2] == off[1] and off[3] == off[2]


@pytest.mark.parametrize("case", ["dense_gaps", "sparse", "unsorted", "mixed_windows"])
def test_q3_every_path_is_exact(ctx, case):
    from flock_amd import Auctions, Persons, WindowSchedule
    rng = np.random.default_rng({"dense_gaps": 11, "sparse": 13, "unsorted": 14, "mixed_windows": 15}[case])
    npn, na = 40_000, 120_000
    spread = {"dense_gaps": 3, "sparse": 50_000, "unsorted": 2, "mixed_windows": 2}[case]
    p_id = _sorted_keys(rng, npn, spread)
    if case == "unsorted":
        p_id[100:120] = p_id[100:120][::-1]
    if case == "mixed_windows":
        p_id[30_000:] = rng.integers(0, 50, npn - 30_000).astype(np.int32)               # duplicates in the last window
    st = rng.choice(np.array([b"or", b"OR", b"id", b"ca", b"wa", b"", b"orx", b"california"], dtype=object), npn)
    state = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in st])]).astype(np.int32),
                        np.frombuffer(b"".join(st), np.uint8).copy())
    nm = [b"n%d" % (i % 313) for i in range(npn)]
    name = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.int32),
                       np.frombuffer(b"".join(nm), np.uint8).copy())
    seller = rng.choice(p_id, na).astype(np.int32)
    seller[::5] = rng.integers(-2**31, 2**31 - 1, len(seller[::5])).astype(np.int32)
    category = rng.integers(9, 12, na).astype(np.int32)
    a_id = (np.arange(na) * 7 + 3).astype(np.int32)
    pw = WindowSchedule(np.array([0, 10_000, 10_000, 30_000, npn]), np.arange(4), np.arange(1, 5))
    aw = WindowSchedule(np.array([0, 50_001, 60_000, 60_000, na]), np.arange(4), np.arange(1, 5))
    out = ctx.q3_join(Auctions(_dev(a_id), _dev(seller), _dev(category), na), aw,
                      Persons(_dev(p_id), _utf8(name), _utf8(name), _utf8(state), npn), pw).to_host()
    off = out["offsets"]
    g_state, g_name = _str_rows(*out["state"]), _str_rows(*out["name"])
    total = 0
    for w in range(4):
        (alo, ahi), (plo, phi) = aw.window_rows(w), pw.window_rows(w)
        ar, pr = oracle.q3_join(seller[alo:ahi], category[alo:ahi], p_id[plo:phi], state.slice(plo, phi))
        sl = slice(off[w], off[w + 1])
        want = sorted((nm[plo + j], st[plo + j], int(a_id[alo + i])) for i, j in zip(ar, pr))
        assert sorted(zip(g_name[sl], g_state[sl], out["a_id"][sl].tolist())) == want, (case, w)
        total += len(ar)
    assert total == len(out["a_id"]) and total > 1000


def test_q3_hash_path_tables_built_in_lds_and_the_lost_bet():
    """The hash path builds each window's multimap in LDS and probes it there (one workgroup per window, q3_window_join_lds_kernel).
    A window of up to 12 K persons gets its usual 1.5 slots per person; a larger one (up to 36 K) the LDS-sized table on the bet that
    the state filter drops enough persons -- here first every person passes (20 K into 18 K slots: the bet is lost, the same call answers
    from tables built in global memory), then the usual mix on the same ctx (no new bet at that size), then smaller windows (no bet
    needed).  Sparse ids with duplicates on both sides, so only the hash path can answer."""
    from flock_amd import Auctions, GpuContext, Persons, WindowSchedule
    c = GpuContext(0)
    rng = np.random.default_rng(23)
    npn, na = 40_000, 150_000
    p_id = _sorted_keys(rng, npn, 50_000)
    p_id[5_000:5_040] = p_id[5_000]                                   # a 40-fold duplicate key on the build side
    nm = [b"n%d" % (i % 911) for i in range(npn)]
    name = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.int32), np.frombuffer(b"".join(nm), np.uint8).copy())
    seller = rng.choice(p_id, na).astype(np.int32)
    category = rng.integers(10, 12, na).astype(np.int32)
    a_id = (np.arange(na) * 3 + 1).astype(np.int32)
    for states, edges in (([b"or"], [0, 20_000, npn]), ([b"or", b"wa", b"tx", b"id"], [0, 20_000, npn]), ([b"or", b"wa"], [0, 9_000, 20_000, 31_000, npn])):
        st = rng.choice(np.array(states, dtype=object), npn)
        state = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in st])]).astype(np.int32), np.frombuffer(b"".join(st), np.uint8).copy())
        n_w = len(edges) - 1
        a_edges = np.linspace(0, na, n_w + 1).astype(np.int64)
        pw = WindowSchedule(np.array(edges), np.arange(n_w), np.arange(1, n_w + 1))
        aw = WindowSchedule(a_edges, np.arange(n_w), np.arange(1, n_w + 1))
        for _ in range(2):
            out = c.q3_join(Auctions(_dev(a_id), _dev(seller), _dev(category), na), aw, Persons(_dev(p_id), _utf8(name), _utf8(name), _utf8(state), npn), pw).to_host()
            off, total = out["offsets"], 0
            for w in range(n_w):
                (alo, ahi), (plo, phi) = aw.window_rows(w), pw.window_rows(w)
                ar, pr = oracle.q3_join(seller[alo:ahi], category[alo:ahi], p_id[plo:phi], state.slice(plo, phi))
                sl = slice(off[w], off[w + 1])
                assert sorted(zip((out["auction_row"][sl] - alo).tolist(), (out["person_row"][sl] - plo).tolist())) == sorted(zip(ar.tolist(), pr.tolist())), (states, w)
                total += len(ar)
            assert total == len(out["a_id"]) > 1000
    c.close()


def test_q3_q8_follow_the_data_on_one_ctx(ctx):
    """The dense paths are speculated from what the previous call on the ctx saw (bit blocks / row table / general path, sizes of
    the Utf8 takes): after the hostile inputs above the same ctx must come back to exact answers on generator data, small then
    ten times larger (the take laid out for the small call is too small and is redone), then small again."""
    for eps, seconds in ((2_000, 4), (40_000, 5), (2_000, 3)):
        _check_q3_nexmark(ctx, 31, eps, seconds)
        _check_q8_nexmark(ctx, 31, eps, 20)


def test_q3_five_launch_sequence_equals_the_general_one():
    """From its second gapless call on a ctx q3 runs build -> probe -> emit -> Utf8 lengths -> Utf8 bytes with self-scanning emits and
    results written straight into pinned memory (q3.hip, "steady-state sequence").  Same bytes as the first (general) call; a call
    with more pairs / bytes than the estimates redoes the take; an input whose ids have gaps or are out of order voids the sequence
    and is answered exactly by the general one; and the ctx comes back to the fast sequence afterwards."""
    from flock_amd import Auctions, GpuContext, Persons, Window, WindowSchedule, run_query
    c = GpuContext(0)
    g = _gpu_stream(c, 77, 30_000, 6, Window.element_wise())
    first = run_query(c, 3, g).to_host()
    for _ in range(3):
        again = run_query(c, 3, g).to_host()
        for k in ("a_id", "auction_row", "person_row", "offsets"):
            assert np.array_equal(first[k], again[k]), k
        for k in ("name", "city", "state"):
            assert np.array_equal(first[k][0], again[k][0]) and np.array_equal(first[k][1], again[k][1]), k
    assert len(first["a_id"]) > 0
    _check_q3_nexmark(c, 78, 300_000, 4)          # ten times the pairs: the take laid out for the small call is redone
    _check_q3_nexmark(c, 78, 300_000, 4)          # ... and now fits
    _check_q3_nexmark(c, 79, 30_000, 3)
    # hostile: gaps in the ids of one window, then a swapped pair -- same ctx, estimates still valid
    rng = np.random.default_rng(5)
    npn, na = 30_000, 90_000
    for hostile in ("gap", "swap"):
        p_id = np.arange(npn, dtype=np.int32) + 1000
        if hostile == "gap":
            p_id[20_000:] += 7
        else:
            p_id[[12_345, 12_346]] = p_id[[12_346, 12_345]]
        st = rng.choice(np.array([b"or", b"id", b"ca", b"wa", b"tx"], dtype=object), npn)
        state = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in st])]).astype(np.int32), np.frombuffer(b"".join(st), np.uint8).copy())
        nm = [b"n%d" % (i % 313) for i in range(npn)]
        name = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.int32), np.frombuffer(b"".join(nm), np.uint8).copy())
        seller = rng.choice(p_id, na).astype(np.int32)
        category = rng.integers(9, 12, na).astype(np.int32)
        a_id = (np.arange(na) * 3 + 1).astype(np.int32)
        pw = WindowSchedule(np.array([0, 10_000, npn]), np.arange(2), np.arange(1, 3))
        aw = WindowSchedule(np.array([0, 40_001, na]), np.arange(2), np.arange(1, 3))
        out = c.q3_join(Auctions(_dev(a_id), _dev(seller), _dev(category), na), aw, Persons(_dev(p_id), _utf8(name), _utf8(name), _utf8(state), npn), pw).to_host()
        off, total = out["offsets"], 0
        for w in range(2):
            (alo, ahi), (plo, phi) = aw.window_rows(w), pw.window_rows(w)
            ar, pr = oracle.q3_join(seller[alo:ahi], category[alo:ahi], p_id[plo:phi], state.slice(plo, phi))
            sl = slice(off[w], off[w + 1])
            assert sorted(zip((out["auction_row"][sl] - alo).tolist(), (out["person_row"][sl] - plo).tolist())) == sorted(zip(ar.tolist(), pr.tolist())), (hostile, w)
            total += len(ar)
        assert total == len(out["a_id"]) > 1000
        _check_q3_nexmark(c, 80, 30_000, 3)       # (general sequence again: it re-arms the estimates)
        _check_q3_nexmark(c, 80, 30_000, 3)       # (fast sequence)
    c.close()


def test_q3_every_row_joins_long_strings_many_tiles():
    """q3's steady-state sequence on what NEXMark never shows it: EVERY auction joins (8192 joined rows per tile), names of 0 .. 70 bytes
    (values beyond the take's 16-byte head, tiles of the take beyond its LDS stage), hundreds of tiles (self-scans over several rounds),
    ragged window edges inside tiles, a window without auctions, and the same ctx called again and again with the selectivity changing
    under it (estimates too large, then too small: the take is redone)."""
    from flock_amd import Auctions, GpuContext, Persons, WindowSchedule
    c = GpuContext(0)
    rng = np.random.default_rng(17)
    npn, na = 50_000, 2_600_000                       # 318 auction tiles
    p_id = np.arange(npn, dtype=np.int32) + 7
    st = rng.choice(np.array([b"or", b"id", b"ca"], dtype=object), npn)
    state = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in st])]).astype(np.int32), np.frombuffer(b"".join(st), np.uint8).copy())
    lens = rng.choice(np.array([0, 1, 3, 15, 16, 17, 33, 70]), npn)
    nm = [bytes(rng.integers(97, 123, int(n), dtype=np.uint8)) for n in lens]
    name = oracle.Utf8(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), np.frombuffer(b"".join(nm), np.uint8).copy())
    cy = [b"c%d" % (i % 97) for i in range(npn)]
    city = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in cy])]).astype(np.int32), np.frombuffer(b"".join(cy), np.uint8).copy())
    p_edges = np.array([0, 10_001, 10_001 + 8192 * 2, 40_003, npn])
    a_edges = np.array([0, 700_003, 700_003, 1_900_001, na])          # the third window has persons and no auctions
    pw = WindowSchedule(p_edges, np.arange(4), np.arange(1, 5))
    aw = WindowSchedule(a_edges, np.arange(4), np.arange(1, 5))
    seller = np.empty(na, np.int32)
    for w in range(4):
        seller[a_edges[w]:a_edges[w + 1]] = rng.integers(p_id[p_edges[w]], p_id[p_edges[w + 1] - 1] + 1, a_edges[w + 1] - a_edges[w])
    a_id = (np.arange(na, dtype=np.int64) * 5 + 3).astype(np.int32)
    persons = Persons(_dev(p_id), _utf8(name), _utf8(city), _utf8(state), npn)
    for variant in ("all", "tenth", "all"):
        category = np.full(na, 10, np.int32) if variant == "all" else rng.choice(np.array([10] + [11] * 9, np.int32), na)
        for call in range(3):     # general sequence, steady-state sequence with estimates from it, and again
            out = c.q3_join(Auctions(_dev(a_id), _dev(seller), _dev(category), na), aw, persons, pw).to_host()
            keep = np.flatnonzero(category == 10)
            assert np.array_equal(out["auction_row"], keep), (variant, call)      # auction order, every row once (all states pass)
            assert np.array_equal(out["a_id"], a_id[keep]) and np.array_equal(out["person_row"], seller[keep] - 7)
            assert out["offsets"].tolist() == [int(np.searchsorted(keep, e)) for e in a_edges], (variant, call)
            for k, src in (("name", name), ("city", city), ("state", state)):
                off, data = out[k]
                want_len = (src.offsets[1:] - src.offsets[:-1])[out["person_row"]]
                assert np.array_equal(np.diff(off), want_len) and off[0] == 0, (k, variant, call)
                pos = np.repeat(src.offsets[:-1][out["person_row"]] - off[:-1], want_len) + np.arange(off[-1])
                assert np.array_equal(data[:off[-1]], src.data[pos]), (k, variant, call)
    c.close()


def _q8_check(out, pw, aw, p_id, name, nm, seller, tag):
    g_names, off = _str_rows(*out["name"]), out["offsets"]
    total = 0
    for w in range(pw.n_windows):
        (plo, phi), (alo, ahi) = pw.window_rows(w), aw.window_rows(w)
        rows = oracle.q8_join(p_id[plo:phi], name.slice(plo, phi), seller[alo:ahi])
        want = sorted((int(p_id[plo + r]), nm[plo + r]) for r in rows)
        sl = slice(off[w], off[w + 1])
        assert sorted(zip(out["p_id"][sl].tolist(), g_names[sl])) == want, (tag, w)
        assert np.array_equal(out["p_id"][sl], p_id[out["person_row"][sl]]), (tag, w)      # person_row names the joined input rows
        total += len(want)
    assert total == len(out["p_id"]) == off[-1]
    return total


def test_q8_three_launch_sequence_and_the_range_path():
    """From its second dense call on a ctx q8 runs sellers -> persons (+ name lengths) -> fused emit (rows, ids, name offsets and bytes;
    bitmap cleaned behind itself) on windows whose ids have no gaps (q8.hip "steady-state sequence").  Same rows as the general
    sequence; estimates too small -> the emit pass once more; gaps, a swapped pair or long names void it and the general sequence
    answers; and ids in ANY order over a dense range take the range path (bitmaps + uniqueness check), duplicates the hash path --
    all on ONE ctx, so every switch of regime is exercised with the state the previous call left."""
    from flock_amd import Auctions, GpuContext, Persons, Window, WindowSchedule, run_query
    c = GpuContext(0)
    g = _gpu_stream(c, 91, 40_000, 30, Window.tumbling(10))
    first = run_query(c, 8, g).to_host()                     # general sequence (first call of the ctx)
    for _ in range(3):                                       # three-launch sequence
        again = run_query(c, 8, g).to_host()
        for k in ("p_id", "person_row", "offsets"):
            assert np.array_equal(first[k], again[k]), k
        assert np.array_equal(first["name"][0], again["name"][0]) and np.array_equal(first["name"][1], again["name"][1])
    assert len(first["p_id"]) > 1000
    _check_q8_nexmark(c, 92, 400_000, 20)                    # ten times the rows: the emit pass is redone with the exact sizes
    _check_q8_nexmark(c, 92, 400_000, 20)
    _check_q8_nexmark(c, 93, 40_000, 20)
    rng = np.random.default_rng(9)
    npn, na = 70_000, 200_000
    pw = WindowSchedule(np.array([0, 25_000, 25_000, 50_000, npn]), np.arange(4), np.arange(1, 5))      # window 1: no persons
    aw = WindowSchedule(np.array([0, 80_001, 120_000, 120_000, na]), np.arange(4), np.arange(1, 5))     # window 2: no auctions
    for case in ("gapless", "gap", "swap", "long_names", "shuffled", "shuffled_wide", "shuffled_dups", "gapless"):
        p_id = np.arange(npn, dtype=np.int32) + 5000
        if case == "gap":
            p_id[60_000:] += 3
        if case == "swap":
            p_id[[31_000, 31_001]] = p_id[[31_001, 31_000]]
        if case.startswith("shuffled"):                      # which person holds which id, shuffled inside every window
            if case == "shuffled_wide":
                p_id[50_000:] = 5000 + 50_000 + np.arange(npn - 50_000, dtype=np.int32) * 20      # last window spans 400 000 ids: beyond the LDS stage
            for w in range(4):
                lo, hi = pw.window_rows(w)
                p_id[lo:hi] = rng.permutation(p_id[lo:hi])
            if case == "shuffled_dups":
                p_id[40_000] = p_id[40_001]                  # two persons of window 2 share an id ...
        width = 40 if case == "long_names" else 9
        nm = [(b"person-%d" % (i % 4999)).ljust(int(rng.integers(0, width)), b"x") if i % 11 else b"" for i in range(npn)]
        if case == "shuffled_dups":
            nm[40_000], nm[40_001] = b"twin-a", b"twin-b"    # ... under different names: both are DISTINCT rows
        name = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.int32), np.frombuffer(b"".join(nm), np.uint8).copy())
        seller = rng.choice(p_id, na).astype(np.int32)
        seller[::9] = rng.integers(-2**31, 2**31 - 1, len(seller[::9])).astype(np.int32)
        seller[500:60_000:2] = p_id[4321]
        if case == "shuffled_dups":
            seller[100_000] = p_id[40_000]                   # the shared id sells: the duplicate matters
        per = Persons(_dev(p_id), _utf8(name), None, None, npn)
        auc = Auctions(None, _dev(seller), None, na)
        for rep in range(2):                                 # second call: whatever sequence the first one armed
            total = _q8_check(c.q8_join(per, pw, auc, aw).to_host(), pw, aw, p_id, name, nm, seller, (case, rep))
            assert total > 1000
    c.close()


def _q8_part_log2(max_p, max_a):
    """The bucket count q8.hip's partitioned hash path picks (kPartPersonsPerBucket / kPartAuctionsPerBucket)."""
    l = 0
    while l < 10 and ((max_p >> l) > 1600 or (max_a >> l) > 6144):
        l += 1
    return l


def _q8_part_bucket(k, log2nb):
    return ((np.asarray(k).astype(np.uint32) * np.uint32(0x9E3779B1)) >> np.uint32(32 - log2nb)).astype(np.int64)


def test_q8_hash_path_grouped_by_bucket():
    """Ids over the whole int32 range in no order (no bitmap is affordable): both relations are grouped by (window, hash bucket), every
    bucket's DISTINCT seller set and {p_id, row} table live in LDS (q8_sellers_part / q8_persons_part / q8_bucket_join kernels).
    Duplicates (same id and name), persons that share an id under different names, the keys -1 / INT_MIN / INT_MAX on both sides, a hot
    seller, windows without persons or without auctions, ragged tiles, one bucket and many buckets -- and a bucket that does NOT fit
    its LDS tables: seller sets that overflow are built again four times as large; a person table that overflows voids the attempt,
    the global tables answer, and the ctx stays off the partitioned path for a while."""
    from flock_amd import Auctions, GpuContext, Persons, WindowSchedule
    c = GpuContext(0)
    rng = np.random.default_rng(33)

    def run(p_id, nm, seller, pw, aw, tag, expect):
        name = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.int32),
                           np.frombuffer(b"".join(nm) or b"\0", np.uint8).copy())
        per = Persons(_dev(p_id), _utf8(name), None, None, len(p_id))
        auc = Auctions(None, _dev(seller), None, len(seller))
        c.profile_reset()
        c.profile(True)
        out = c.q8_join(per, pw, auc, aw).to_host()
        st = c.profile_read()
        c.profile(False)
        total = _q8_check(out, pw, aw, p_id, name, nm, seller, tag)
        for k, want in expect.items():
            assert (k in st) == want, (tag, k, sorted(st))
        return total

    for npn, na, bounds_p, bounds_a in ((3_000, 9_000, [0, 1_000, 1_000, 2_200, 3_000], [0, 4_001, 6_000, 6_000, 9_000]),            # one bucket
                                        (140_000, 400_000, [0, 50_000, 50_000, 95_001, 140_000], [0, 150_001, 230_000, 230_000, 400_000])):
        pw = WindowSchedule(np.array(bounds_p), np.arange(4), np.arange(1, 5))        # window 1: no persons
        aw = WindowSchedule(np.array(bounds_a), np.arange(4), np.arange(1, 5))        # window 2: no auctions
        p_id = rng.integers(-2**31, 2**31 - 1, npn).astype(np.int32)
        p_id[5:npn:97] = p_id[4:npn - 1:97]                                          # neighbours share an id ...
        nm = [b"who-%d" % (i % 3989) if i % 13 else b"" for i in range(npn)]
        for i in range(5, npn, 194):
            nm[i] = nm[i - 1]                                                         # ... half of them under the same name: duplicates
        p_id[[7, 8, npn - 1]] = [-1, -2**31, 2**31 - 1]
        p_id[npn - 2] = -1                                                            # -1 twice in the last window, different names
        seller = rng.choice(p_id, na).astype(np.int32)
        seller[::9] = rng.integers(-2**31, 2**31 - 1, len(seller[::9])).astype(np.int32)
        seller[500:na // 3:2] = p_id[321]
        seller[[11, na - 1, na - 2]] = [-1, -1, 2**31 - 1]
        seller[12] = -2**31
        for rep in range(2):
            total = run(p_id, nm, seller, pw, aw, (npn, rep), {"q8_bucket_join_kernel": True, "q8_persons_general_kernel": False})
            assert total > 500
    # a window of 281 auction tiles (the bucket workgroups index a relation's tiles 256 at a time), and a bucket with four times the
    # average number of persons (more than a workgroup requests ahead)
    npn, na = 60_000, 2_310_000
    pw = WindowSchedule(np.array([0, 30_000, npn]), np.arange(2), np.arange(1, 3))
    aw = WindowSchedule(np.array([0, 2_300_000, na]), np.arange(2), np.arange(1, 3))
    p_id = rng.integers(-2**31, 2**31 - 1, npn).astype(np.int32)
    nm = [b"m%d" % (i % 7919) for i in range(npn)]
    seller = rng.choice(p_id[::3], na).astype(np.int32)
    seller[::11] = rng.integers(-2**31, 2**31 - 1, len(seller[::11])).astype(np.int32)
    assert run(p_id, nm, seller, pw, aw, "many_auction_tiles", {"q8_bucket_join_kernel": True, "q8_persons_general_kernel": False}) > 10_000
    npn, na = 60_000, 100_000
    pw = WindowSchedule(np.array([0, 30_000, npn]), np.arange(2), np.arange(1, 3))
    aw = WindowSchedule(np.array([0, 50_000, na]), np.arange(2), np.arange(1, 3))
    log2nb = _q8_part_log2(30_000, 50_000)
    pool = rng.integers(-2**31, 2**31 - 1, 400_000).astype(np.int32)
    fat = np.unique(pool[_q8_part_bucket(pool, log2nb) == 5])[:3000]
    assert len(fat) == 3000 and log2nb >= 4
    p_id = rng.integers(-2**31, 2**31 - 1, npn).astype(np.int32)
    p_id[31_000:34_000] = fat
    nm = [b"f%d" % i for i in range(npn)]
    seller = rng.choice(p_id, na).astype(np.int32)
    seller[60_000:61_000] = fat[::3]
    assert run(p_id, nm, seller, pw, aw, "fat_bucket", {"q8_bucket_join_kernel": True, "q8_persons_general_kernel": False}) > 10_000
    # buckets that do not fit: 6000 distinct sellers of ONE bucket in one window (more than the small seller set's 4096 slots), and --
    # second data set -- one id under 300 names (a chain the person table's bounded probing gives up on)
    npn, na = 60_000, 100_000
    pw = WindowSchedule(np.array([0, 30_000, npn]), np.arange(2), np.arange(1, 3))
    aw = WindowSchedule(np.array([0, 50_000, na]), np.arange(2), np.arange(1, 3))
    log2nb = _q8_part_log2(30_000, 50_000)
    assert log2nb >= 4
    pool = rng.integers(-2**31, 2**31 - 1, 400_000).astype(np.int32)
    crowd = np.unique(pool[_q8_part_bucket(pool, log2nb) == 3])[:6000]
    assert len(crowd) == 6000
    for case in ("crowded_bucket", "one_id_many_names"):
        p_id = rng.integers(-2**31, 2**31 - 1, npn).astype(np.int32)
        nm = [b"n%d" % i for i in range(npn)]
        seller = rng.choice(p_id, na).astype(np.int32)
        if case == "crowded_bucket":
            seller[:6000] = crowd
            p_id[1000:1500] = crowd[:500]
        else:
            p_id[1000:1300] = 424242                                                   # (a chain the LDS table's bounded probing gives up on)
            seller[77] = 424242
        if case == "crowded_bucket":      # the join alone is repeated with the large seller sets, which this ctx then keeps
            first = later = {"q8_bucket_join_kernel": True, "q8_persons_general_kernel": False}
        else:
            first = {"q8_bucket_join_kernel": True, "q8_persons_general_kernel": True}      # tried, void, answered by the global tables
            later = {"q8_bucket_join_kernel": False, "q8_persons_general_kernel": True}     # not tried again right away
        assert run(p_id, nm, seller, pw, aw, (case, 0), first) > 300
        assert run(p_id, nm, seller, pw, aw, (case, 1), later) > 300
        c.close()
        c = GpuContext(0)
    c.close()


def test_q3_range_path_ids_in_any_order():
    """q3 with the persons' ids shuffled inside every window (dense range, no order): the row table laid out from exact statistics
    (q3.hip "RANGE path"); duplicates in a window void it for the hash join; gapless data afterwards returns to the bit blocks."""
    from flock_amd import Auctions, GpuContext, Persons, WindowSchedule
    c = GpuContext(0)
    rng = np.random.default_rng(21)
    npn, na = 50_000, 160_000
    pw = WindowSchedule(np.array([0, 15_000, 15_000, 35_000, npn]), np.arange(4), np.arange(1, 5))
    aw = WindowSchedule(np.array([0, 60_001, 90_000, 90_000, na]), np.arange(4), np.arange(1, 5))
    for case in ("ordered", "shuffled", "shuffled", "shuffled_gaps", "shuffled_dups", "ordered", "ordered"):
        p_id = np.arange(npn, dtype=np.int32) + 777
        if case == "shuffled_gaps":
            p_id = (777 + np.cumsum(rng.integers(1, 4, npn))).astype(np.int32)
        if case.startswith("shuffled"):
            for w in range(4):
                lo, hi = pw.window_rows(w)
                p_id[lo:hi] = rng.permutation(p_id[lo:hi])
        if case == "shuffled_dups":
            p_id[20_000:20_050] = p_id[20_050:20_100]
        st = rng.choice(np.array([b"or", b"id", b"ca", b"wa", b"tx", b""], dtype=object), npn)
        state = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in st])]).astype(np.int32), np.frombuffer(b"".join(st), np.uint8).copy())
        nm = [b"n%d" % (i % 313) for i in range(npn)]
        name = oracle.Utf8(np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.int32), np.frombuffer(b"".join(nm), np.uint8).copy())
        seller = rng.choice(p_id, na).astype(np.int32)
        seller[::6] = rng.integers(-2**31, 2**31 - 1, len(seller[::6])).astype(np.int32)
        category = rng.integers(9, 12, na).astype(np.int32)
        a_id = (np.arange(na) * 5 + 2).astype(np.int32)
        out = c.q3_join(Auctions(_dev(a_id), _dev(seller), _dev(category), na), aw, Persons(_dev(p_id), _utf8(name), _utf8(name), _utf8(state), npn), pw).to_host()
        off, total = out["offsets"], 0
        g_state, g_name = _str_rows(*out["state"]), _str_rows(*out["name"])
        for w in range(4):
            (alo, ahi), (plo, phi) = aw.window_rows(w), pw.window_rows(w)
            ar, pr = oracle.q3_join(seller[alo:ahi], category[alo:ahi], p_id[plo:phi], state.slice(plo, phi))
            sl = slice(off[w], off[w + 1])
            assert sorted(zip((out["auction_row"][sl] - alo).tolist(), (out["person_row"][sl] - plo).tolist())) == sorted(zip(ar.tolist(), pr.tolist())), (case, w)
            assert sorted(zip(g_name[sl], g_state[sl], out["a_id"][sl].tolist())) == sorted((nm[plo + j], st[plo + j], int(a_id[alo + i])) for i, j in zip(ar, pr)), (case, w)
            total += len(ar)
        assert total == len(out["a_id"]) > 1000, case
    c.close()


# ------------------------------------------------------------------ full-size, size-independent properties
def test_full_size_properties(ctx):
    """1e8-event stream (configs q2/q3 of BASELINE.json): checks that need no CPU pass over all rows."""
    from flock_amd import NEXMarkSource, Window, run_query
    seconds, eps = 100, 1_000_000
    g = NEXMarkSource(seconds, eps, Window.element_wise(), seed=99).generate_data(ctx, bid_columns=("auction", "price"))
    # q2: idempotence (filtering the filtered rows again keeps everything) + stability (sorted row positions)
    r2 = run_query(ctx, 2, g)
    a, p, off = r2.to_host()
    assert (a.astype(np.int64) % 123 == 0).all() and len(a) == off[-1]
    cnt = int((g.bids.auction.to(dtype=__import__("torch").int64) % 123 == 0).sum().item())
    assert cnt == len(a)
    # q5: sum over the window's groups is implied by max <= rows, winners' count equals win_max, groups > 0
    r5 = ctx.q5_hot_items(g.bids, g.window_schedule("bid", Window.hopping(10, 5)))
    wa, wn, woff = r5.to_host()
    mx = r5.win_max()
    assert len(mx) == 19 and (mx > 0).all() and (r5.win_groups() > 0).all()
    for w in range(19):
        assert (wn[woff[w]:woff[w + 1]] == mx[w]).all() and woff[w + 1] > woff[w]
    # spot-check 2 windows against the oracle
    sched = g.window_schedule("bid", Window.hopping(10, 5))
    host_auction = g.bids.auction.cpu().numpy()
    for w in (0, 18):
        lo, hi = sched.window_rows(w)
        oa, on = oracle.q5_hot_items(host_auction[lo:hi])
        assert sorted(zip(wa[woff[w]:woff[w + 1]].tolist(), wn[woff[w]:woff[w + 1]].tolist())) == sorted(zip(oa.tolist(), on.tolist()))
    # q3 / q8: every output row satisfies the predicates and the join condition (checked on the row pairs)
    o3 = run_query(ctx, 3, g).to_host()
    seller = g.auctions.seller.cpu().numpy()
    cat = g.auctions.category.cpu().numpy()
    pid = g.persons.p_id.cpu().numpy()
    assert (cat[o3["auction_row"]] == 10).all() and (seller[o3["auction_row"]] == pid[o3["person_row"]]).all()
    assert set(_str_rows(*o3["state"])) <= {b"or", b"id", b"ca"} and len(o3["a_id"]) > 0
    assert (np.diff(o3["auction_row"].astype(np.int64)) >= 0).all()          # order-preserving expansion
    o8 = run_query(ctx, 8, g).to_host()
    assert len(np.unique(o8["p_id"])) == len(o8["p_id"]) > 0                   # DISTINCT
    assert (np.diff(o8["person_row"].astype(np.int64)) > 0).all()


def test_baseline_sizes_properties(ctx):
    """BASELINE.json's full sizes (q5: 1e9 bids, q8 / q3: 1e9 events), checked through properties that need no CPU pass
    over all rows: winners verified against an independent device-side count (torch.bincount) on sampled windows and
    against the oracle on two of them; q8 / q3 outputs verified against torch set operations on sampled windows."""
    import torch
    from flock_amd import NEXMarkSource, Window, query_window, run_query
    eps = 1_000_000
    g = NEXMarkSource(1087, eps, query_window(5), seed=20260925).generate_data(ctx, relations=("bid",), bid_columns=("auction",))
    assert g.bids.rows == 1_000_040_000
    sched = g.window_schedule("bid")
    r5 = run_query(ctx, 5, g)
    a, n, off = r5.to_host()
    mx, groups = r5.win_max(), r5.win_groups()
    assert sched.n_windows == 216 and len(off) == 217 and (np.diff(off) >= 1).all() and (mx > 0).all()
    for w in (0, 1, 57, 108, 214, 215):
        lo, hi = sched.window_rows(w)
        keys = g.bids.auction[lo:hi].to(torch.int64)
        base = int(keys.min())
        cnt = torch.bincount(keys - base)
        assert int(cnt.max()) == int(mx[w]) and int((cnt > 0).sum()) == int(groups[w])
        winners = (torch.nonzero(cnt == cnt.max()).flatten() + base).cpu().numpy()
        sl = slice(off[w], off[w + 1])
        assert np.array_equal(np.sort(a[sl]), winners) and (n[sl] == mx[w]).all()
    for w in (3, 200):
        lo, hi = sched.window_rows(w)
        oa, on = oracle.q5_hot_items(g.bids.auction[lo:hi].cpu().numpy())
        assert sorted(zip(a[off[w]:off[w + 1]].tolist(), n[off[w]:off[w + 1]].tolist())) == sorted(zip(oa.tolist(), on.tolist()))
    del g, r5
    torch.cuda.empty_cache()

    g = NEXMarkSource(1000, eps, query_window(8), seed=20260925).generate_data(ctx, relations=("auction", "person"))
    assert g.auctions.rows + g.persons.rows == 80_000_000
    sp, sa = g.window_schedule("person"), g.window_schedule("auction")
    o8 = run_query(ctx, 8, g)
    h8 = o8.to_host()
    off = h8["offsets"]
    assert (np.diff(h8["person_row"].astype(np.int64)) > 0).all() and off[-1] == len(h8["p_id"])
    for w in (0, 49, 99):
        (plo, phi), (alo, ahi) = sp.window_rows(w), sa.window_rows(w)
        pid = g.persons.p_id[plo:phi]
        keep = torch.isin(pid, torch.unique(g.auctions.seller[alo:ahi]))
        assert torch.unique(pid).numel() == pid.numel()                       # DISTINCT is the identity on this input
        assert np.array_equal(h8["p_id"][off[w]:off[w + 1]], pid[keep].cpu().numpy())
        assert np.array_equal(h8["person_row"][off[w]:off[w + 1]] - plo, torch.nonzero(keep).flatten().cpu().numpy())
    # names of the first window against the oracle's take
    rows0 = h8["person_row"][off[0]:off[1]].astype(np.int64)
    pe_off = g.persons.name.offsets[: sp.window_rows(0)[1] + 1].cpu().numpy()
    pe_dat = g.persons.name.data[: int(pe_off[-1])].cpu().numpy()
    want = oracle.take_utf8(oracle.Utf8(pe_off, pe_dat), rows0)
    got_off, got_dat = h8["name"]
    assert np.array_equal(got_off[: len(rows0) + 1], want.offsets) and np.array_equal(got_dat[: len(want.data)], want.data)

    ew = Window.element_wise()
    o3 = run_query(ctx, 3, g, ew).to_host()
    sa3, sp3 = g.window_schedule("auction", ew), g.window_schedule("person", ew)
    seller, cat, pid = g.auctions.seller, g.auctions.category, g.persons.p_id
    ar, pr = torch.from_numpy(o3["auction_row"].astype(np.int64)).cuda(), torch.from_numpy(o3["person_row"].astype(np.int64)).cuda()
    assert bool((cat[ar] == 10).all()) and bool((seller[ar] == pid[pr]).all()) and len(o3["a_id"]) > 1_000_000
    assert set(_str_rows(*o3["state"])[:200_000]) <= {b"or", b"id", b"ca"}
    st_off = g.persons.state.offsets
    for w in (0, 500, 999):                                                  # exact pair count per sampled window
        (alo, ahi), (plo, phi) = sa3.window_rows(w), sp3.window_rows(w)
        so = st_off[plo:phi + 1].cpu().numpy()
        sd = g.persons.state.data[int(so[0]):int(so[-1])].cpu().numpy()
        oa, op = oracle.q3_join(seller[alo:ahi].cpu().numpy(), cat[alo:ahi].cpu().numpy(), pid[plo:phi].cpu().numpy(),
                                oracle.Utf8((so - so[0]).astype(np.int32), sd))
        sl = slice(o3["offsets"][w], o3["offsets"][w + 1])
        assert sorted(zip((o3["auction_row"][sl] - alo).tolist(), (o3["person_row"][sl] - plo).tolist())) == sorted(zip(oa.tolist(), op.tolist()))
