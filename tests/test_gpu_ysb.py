"""Yahoo Streaming Benchmark on the GPU vs the CPU restatement: the device generator
regenerates the oracle's bytes, and the Utf8-keyed join + group-by is exact -- per window, as {campaign_id: count}."""
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
    data = u.data if len(u.data) >= 16 else np.concatenate([u.data, np.zeros(16 - len(u.data), np.uint8)])
    return DeviceUtf8(_dev(u.offsets), _dev(data))


def _col(strings):
    off = np.concatenate(([0], np.cumsum([len(s) for s in strings]))).astype(np.int32)
    return oracle.Utf8(off, np.frombuffer(b"".join(strings) or b"\0", np.uint8).copy()[: int(off[-1])] if off[-1] else np.zeros(0, np.uint8))


def _result_dicts(out):
    off, data = out["campaign_id"]
    b = data.tobytes()
    names = [b[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    wo = out["offsets"]
    return [dict(zip(names[wo[w]:wo[w + 1]], out["count"][wo[w]:wo[w + 1]].tolist())) for w in range(len(wo) - 1)]


@pytest.mark.parametrize("seed,eps,seconds,campaigns,ads", [(1, 1000, 30, 100, 10), (9, 20_000, 20, 7, 3), (4, 200_000, 10, 1000, 10)])
def test_ysb_generator_and_query(ctx, seed, eps, seconds, campaigns, ads):
    from flock_amd.ysb import YSBSource, run_ysb
    g = YSBSource(seconds, eps, campaigns=campaigns, ads=ads, seed=seed).generate_data(ctx)
    c_ad, camp = oracle.ysb_campaigns(seed, campaigns, ads)
    ad, et = oracle.ysb_events(seed, 0, seconds * eps, campaigns * ads)
    for dev, host in ((g.c_ad_id, c_ad), (g.campaign_id, camp), (g.ad_id, ad), (g.event_type, et)):   # generator: device == oracle
        assert np.array_equal(dev.offsets.cpu().numpy(), host.offsets)
        assert np.array_equal(dev.data.cpu().numpy()[: len(host.data)], host.data)
    got = _result_dicts(run_ysb(ctx, g).to_host())
    assert len(got) == seconds // 10
    for w in range(seconds // 10):
        lo, hi = w * 10 * eps, (w + 1) * 10 * eps
        want = oracle.ysb_campaign_counts(ad.slice(lo, hi), et.slice(lo, hi), c_ad, camp)
        assert got[w] == want and len(want) > 0, w


def test_ysb_duplicates_unknown_ads_and_odd_strings(ctx):
    """Duplicate c_ad_id rows (every match counts), one campaign_id shared by far-apart rows, ads no campaign has, empty and
    maximum-length (40-byte) keys, an event_type that only shares a prefix with the literal, ragged / empty windows; and
    the unsupported case (a 41-byte key)."""
    from flock_amd import FlockGpuError, WindowSchedule, _ffi
    from flock_amd.ysb import campaign_counts
    rng = np.random.default_rng(3)
    keys = [b"", b"k", b"x" * 40, b"ad-0001", b"ad-0002", b"ad-0002", b"AD-0002", b"y" * 39 + b"a", b"y" * 39 + b"b"]
    camps = [b"c-empty", b"c1", b"c-long", b"c1", b"c2", b"c3", b"c2", b"", b"c1"]
    c_ad, camp = _col(keys), _col(camps)
    pool = keys + [b"nobody", b"x" * 39, b"ad-0003"]
    n = 30_000
    ev_keys = [pool[i] for i in rng.integers(0, len(pool), n)]
    ev_types = [[b"view", b"click", b"purchase", b"vie", b"views", b""][i] for i in rng.integers(0, 6, n)]
    ad, et = _col(ev_keys), _col(ev_types)
    offs = np.array([0, 5, 5, 12_345, n])
    sched = WindowSchedule(offs, np.arange(4), np.arange(1, 5))
    got = _result_dicts(campaign_counts(ctx, _utf8(ad), _utf8(et), n, sched, _utf8(c_ad), _utf8(camp), len(keys)).to_host())
    for w in range(4):
        want = oracle.ysb_campaign_counts(ad.slice(offs[w], offs[w + 1]), et.slice(offs[w], offs[w + 1]), c_ad, camp)
        assert got[w] == want, w
    assert got[1] == {} and sum(got[3].values()) > 1000 and b"" in got[3]
    for lit in ("click", "", "purchase"):
        got = _result_dicts(campaign_counts(ctx, _utf8(ad), _utf8(et), n, sched, _utf8(c_ad), _utf8(camp), len(keys), lit).to_host())
        assert got[3] == oracle.ysb_campaign_counts(ad.slice(offs[3], n), et.slice(offs[3], n), c_ad, camp, lit.encode())
    with pytest.raises(FlockGpuError) as e:
        campaign_counts(ctx, _utf8(ad), _utf8(et), n, sched, _utf8(_col([b"z" * 41])), _utf8(_col([b"c"])), 1)
    assert e.value.code == _ffi.ERR_UNSUPPORTED
    empty = campaign_counts(ctx, _utf8(ad), _utf8(et), 0, WindowSchedule(np.array([0, 0]), np.array([0]), np.array([1])), _utf8(c_ad),
                            _utf8(camp), len(keys))
    assert empty.rows == 0 and empty.offsets().tolist() == [0, 0]


@pytest.mark.parametrize("n_camp_rows", [50, 9000])
def test_ysb_long_literals_and_wide_campaign_tables(ctx, n_camp_rows):
    """Literals beyond the 12-byte single-load filter (13, 20 and 40 bytes, and values that differ from them only in the
    last byte), with the campaign table both inside and beyond what the LDS histogram holds (8192 rows)."""
    from flock_amd import WindowSchedule
    from flock_amd.ysb import campaign_counts
    rng = np.random.default_rng(n_camp_rows)
    keys = [b"ad-%06d" % i for i in range(n_camp_rows)]
    camps = [b"campaign-%04d" % (i % 997) for i in range(n_camp_rows)]
    c_ad, camp = _col(keys), _col(camps)
    types = [b"impression-13", b"impression-14", b"impression-served-ok", b"impression-served-no", b"t" * 40, b"t" * 39 + b"u", b"view"]
    n = 40_000
    ad = _col([keys[i] if i < n_camp_rows else b"ad-unknown" for i in rng.integers(0, n_camp_rows + n_camp_rows // 4 + 1, n)])
    et = _col([types[i] for i in rng.integers(0, len(types), n)])
    offs = np.array([0, 17_001, n])
    sched = WindowSchedule(offs, np.arange(2), np.arange(1, 3))
    for lit in (types[0], types[2], types[4], b"view"):
        got = _result_dicts(campaign_counts(ctx, _utf8(ad), _utf8(et), n, sched, _utf8(c_ad), _utf8(camp), n_camp_rows,
                                            lit.decode()).to_host())
        for w in range(2):
            want = oracle.ysb_campaign_counts(ad.slice(offs[w], offs[w + 1]), et.slice(offs[w], offs[w + 1]), c_ad, camp, lit)
            assert got[w] == want and sum(want.values()) > 1000, (lit, w)


def _hash_words(keys):
    """ysb.hip's `hash_words` over equal-length byte keys (<= 40 bytes), vectorised: the test needs two DIFFERENT keys that
    the device table cannot tell apart by hash."""
    n, ln = len(keys), len(keys[0])
    buf = np.zeros((n, 40), np.uint8)
    buf[:, :ln] = np.frombuffer(b"".join(keys), np.uint8).reshape(n, ln)
    w = buf.view("<u4").astype(np.uint64)
    m = np.uint64(0xFFFFFFFF)
    h = np.full(n, 0x811C9DC5 ^ ln, np.uint64)
    for i in range(10):
        h = ((h ^ w[:, i]) * np.uint64(0x9E3779B1)) & m
        h = ((h << np.uint64(13)) | (h >> np.uint64(19))) & m
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & m
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & m
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def test_ysb_two_ads_with_one_hash(ctx):
    """Two ad ids whose 32-bit hashes are EQUAL: same home slot, same tag in the workgroup's LDS copy of the table, so the
    walk stops at whichever of them sits first and only the comparison of the whole key tells them apart (the probe then
    carries on in the global table).  Counts of both, and of a third ad that is in no campaign, must still be exact."""
    from flock_amd import WindowSchedule
    from flock_amd.ysb import campaign_counts
    keys = [b"ad-%033d" % i for i in range(400_000)]
    h = _hash_words(keys)
    order = np.argsort(h, kind="stable")
    same = np.nonzero(h[order][1:] == h[order][:-1])[0]
    assert len(same) > 0, "no 32-bit collision among 4e5 keys (expected ~18)"
    twins = [(keys[order[i]], keys[order[i + 1]]) for i in same[:3]]
    rng = np.random.default_rng(5)
    for a, b in twins:
        others = [b"ad-%033d" % i for i in range(500_000, 500_040)]
        c_keys = [a, b] + others
        camps = [b"campaign-a", b"campaign-b"] + [b"campaign-%02d" % (i % 7) for i in range(len(others))]
        pool = c_keys + [b"ad-%033d" % 999_999_999]
        n = 30_000
        ad = _col([pool[i] for i in rng.integers(0, len(pool), n)])
        et = _col([[b"view", b"click"][i] for i in rng.integers(0, 2, n)])
        offs = np.array([0, 11_111, n])
        sched = WindowSchedule(offs, np.arange(2), np.arange(1, 3))
        c_ad, camp = _col(c_keys), _col(camps)
        got = _result_dicts(campaign_counts(ctx, _utf8(ad), _utf8(et), n, sched, _utf8(c_ad), _utf8(camp), len(c_keys)).to_host())
        for w in range(2):
            want = oracle.ysb_campaign_counts(ad.slice(offs[w], offs[w + 1]), et.slice(offs[w], offs[w + 1]), c_ad, camp)
            assert got[w] == want and want[b"campaign-a"] > 50 and want[b"campaign-b"] > 50, (a, b, w)
