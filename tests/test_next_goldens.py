"""tests/golden/next_hashes.json (tools/make_next_goldens.py): frozen per-window answers of YSB and q11, minted only where the
scalar oracle, Arrow C++ (pyarrow / Acero) and -- at the small sizes -- the literal Python restatement all agree.  CPU: the small
entries are re-minted and compared; the three YSB implementations and the three q11 implementations are checked against each other.
GPU (-m gpu): the HIP path reproduces EVERY frozen window, the sizes bench.py runs (5e7 ad events, 1e8 bids) included."""
import json
import os
import sys

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "next_hashes.json")))


def _kv(k):
    parts = k.split("/")
    return parts[0], {p.split("=")[0]: int(p.split("=")[1]) for p in parts[1:]}


KEYS = sorted(GOLDEN, key=lambda k: (_kv(k)[1]["eps"] * _kv(k)[1]["seconds"], k))
SMALL = [k for k in KEYS if _kv(k)[1]["eps"] * _kv(k)[1]["seconds"] <= 5_000_000]


def test_golden_file_covers_the_bench_sizes():
    assert GOLDEN["ysb/seed=20260925/eps=1000000/seconds=50/campaigns=100/ads=10"]["windows"] == 5
    assert GOLDEN["q11/seed=20260926/eps=1000000/seconds=109/timeout=10"]["epochs"] == 109
    for k in KEYS:
        assert len(GOLDEN[k]["fingerprints"]) == GOLDEN[k].get("windows", GOLDEN[k].get("epochs")) > 0, k


@pytest.mark.parametrize("k", SMALL)
def test_small_entries_are_reproduced_by_every_implementation(k):
    import make_next_goldens as m
    kind, a = _kv(k)
    got = m.mint_ysb(a["seed"], a["eps"], a["seconds"], a["campaigns"], a["ads"], threads=2) if kind == "ysb" else m.mint_q11(a["seed"], a["eps"], a["seconds"], a["timeout"])
    assert got == GOLDEN[k]


def test_ysb_c_twin_and_arrow_agree_with_the_dict_walk_on_hostile_rows():
    """Duplicate c_ad_id rows (every match counts), one campaign_id on far-apart rows, unknown ads, empty / long keys, event types that
    only share a prefix with the literal."""
    def col(strings):
        off = np.concatenate(([0], np.cumsum([len(s) for s in strings]))).astype(np.int32)
        return oracle.Utf8(off, np.frombuffer(b"".join(strings), np.uint8).copy() if off[-1] else np.zeros(0, np.uint8))
    rng = np.random.default_rng(3)
    keys = [b"", b"k", b"x" * 40, b"ad-0001", b"ad-0002", b"ad-0002", b"AD-0002", b"y" * 39 + b"a", b"y" * 39 + b"b"]
    camps = [b"c-empty", b"c1", b"c-long", b"c1", b"c2", b"c3", b"c2", b"", b"c1"]
    pool = keys + [b"nobody", b"x" * 39, b"ad-0003"]
    n = 20_000
    ad = col([pool[i] for i in rng.integers(0, len(pool), n)])
    et = col([[b"view", b"click", b"purchase", b"vie", b"views", b""][i] for i in rng.integers(0, 6, n)])
    c_ad, camp = col(keys), col(camps)
    want = oracle.ysb_campaign_counts(ad, et, c_ad, camp)
    assert oracle.ysb_campaign_counts_c(ad, et, c_ad, camp) == want == oracle.ysb_campaign_counts_arrow(ad, et, c_ad, camp)
    assert len(want) == 6 and oracle.ysb_campaign_counts_c(ad.slice(0, 0), et.slice(0, 0), c_ad, camp) == {}
    assert oracle.ysb_fingerprint(want) != oracle.ysb_fingerprint({k: v + (k == b"c1") for k, v in want.items()})


@pytest.mark.parametrize("seed,n_epochs,per_epoch,n_bidders,timeout,jitter", [(0, 12, 40, 9, 2, 0), (2, 20, 60, 15, 1, 2500), (3, 25, 10, 30, 4, 9000), (5, 40, 15, 60, 10, 500)])
def test_q11_arrow_formulation_equals_the_walk(seed, n_epochs, per_epoch, n_bidders, timeout, jitter):
    base = 1_436_918_400_000
    rng = np.random.default_rng(seed)
    rows = []
    for t in range(n_epochs):
        ms = np.sort(rng.integers(0, 1000, int(rng.integers(0, per_epoch + 1))))
        for v in ms:
            late = int(rng.integers(0, jitter + 1)) if jitter else 0
            rows.append((t, int(rng.integers(100, 100 + n_bidders)), max(0, t * 1000 + int(v) - late)))
    bidder = np.array([r[1] for r in rows], np.int32)
    ts = np.array([base + r[2] for r in rows], np.int64)
    off = np.searchsorted(np.array([r[0] for r in rows]), np.arange(n_epochs + 1)).astype(np.int64)
    walk = oracle.q11_user_sessions(bidder, ts, off, timeout, base)
    o, who, cnt, mn, mx = oracle.q11_user_sessions_arrow(bidder, ts, off, timeout, base)
    got = [{int(who[i]): (int(cnt[i]), int(mn[i]), int(mx[i])) for i in range(o[t], o[t + 1])} for t in range(n_epochs)]
    assert got == walk
    assert oracle.q11_fingerprints((o, who, cnt, mn, mx)) == oracle.q11_fingerprints(oracle.q11_user_sessions_columnar(bidder, ts, off, timeout, base))


# ------------------------------------------------------------------ GPU: the HIP path against every frozen window
@pytest.fixture(scope="module")
def ctx():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", KEYS)
def test_hip_path_reproduces_every_frozen_window(ctx, k):
    import torch
    kind, a = _kv(k)
    if kind == "ysb":
        from flock_amd.ysb import YSBSource, run_ysb
        g = YSBSource(a["seconds"], a["eps"], campaigns=a["campaigns"], ads=a["ads"], seed=a["seed"]).generate_data(ctx)
        out = run_ysb(ctx, g).to_host()
        off, data = out["campaign_id"]
        b = data.tobytes()
        names = [b[off[i]:off[i + 1]] for i in range(len(off) - 1)]
        wo = out["offsets"]
        got = [oracle.ysb_fingerprint(dict(zip(names[wo[w]:wo[w + 1]], out["count"][wo[w]:wo[w + 1]].tolist()))) for w in range(len(wo) - 1)]
    else:
        from flock_amd.nexmark import NEXMarkSource, Window, run_query
        w = Window.session(a["timeout"])
        g = NEXMarkSource(a["seconds"], a["eps"], w, seed=a["seed"]).generate_data(ctx, relations=("bid",), bid_columns=("bidder", "b_date_time"))
        o = run_query(ctx, 11, g, w).to_host()
        got = oracle.q11_fingerprints((o["offsets"], o["bidder"], o["bid_count"], o["start_time"], o["end_time"]))
    del g
    torch.cuda.empty_cache()
    want = GOLDEN[k]["fingerprints"]
    assert len(got) == len(want)
    bad = [w for w in range(len(want)) if got[w] != want[w]]
    assert not bad, f"{k}: {len(bad)} windows differ, first {bad[:5]}: {[(got[w], want[w]) for w in bad[:3]]}"
