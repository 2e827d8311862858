"""Generator restatement checks (shape, not RNG bits -- SURVEY.md appendix B)."""
import numpy as np

import oracle


def test_event_counts_match_reference_assertions():
    # flock/src/datasource/nexmark/nexmark.rs:427-453: 10 000 eps x 1 s -> exactly 10 000 events,
    # 10 000 eps x 10 s -> 100 000; mix 1:3:46 (config.rs:135-138)
    s = oracle.NexmarkStream(eps=10_000)
    assert s.counts(0, 10_000) == (200, 600, 9200)
    assert sum(s.counts(0, 100_000)) == 100_000
    # ragged ranges
    assert s.counts(3, 53) == (1, 3, 46)
    assert s.counts(1, 4) == (0, 3, 0)


def test_ids_dense_and_typed():
    s = oracle.NexmarkStream(seed=3, eps=5000)
    p, a, b = s.persons(0, 5000), s.auctions(0, 5000), s.bids(0, 5000)
    assert p["p_id"].tolist() == list(range(1000, 1100))          # event.rs:154, first_person_id
    assert a["a_id"].tolist() == list(range(1000, 1300))          # event.rs:261
    assert set(np.unique(a["category"]).tolist()) <= {10, 11, 12, 13, 14}
    assert set(p["state"].to_pylist()) <= {"az", "ca", "id", "or", "wa", "wy"}
    assert b["price"].min() >= 100 and b["price"].max() <= 100_000_000   # event.rs:53-55
    assert b["auction"].dtype == np.int32 and b["b_date_time"].dtype == np.int64
    # timestamps: exactly eps events per 1-s epoch (deviation D3)
    ts = b["b_date_time"] - oracle.BASE_TIME
    assert ts.min() >= 0 and ts.max() < 1000


def test_hot_key_skew():
    s = oracle.NexmarkStream(seed=9, eps=100_000)
    b = s.bids(0, 100_000)
    hot = (b["auction"] - 1000) % 100 == 0
    assert 0.45 < hot.mean() < 0.56                                  # hot_auction_ratio = 2 (event.rs:355-359)
    hotb = (b["bidder"] - 1000) % 100 == 1
    assert 0.70 < hotb.mean() < 0.80                                 # hot_bidder_ratio = 4 (event.rs:360-364)
    a = s.auctions(0, 100_000)
    assert 0.70 < ((a["seller"] - 1000) % 100 == 0).mean() < 0.80    # hot_seller_ratio = 4 (event.rs:255-259)


def test_slices_concatenate():
    s = oracle.NexmarkStream(seed=5, eps=2000)
    whole = s.bids(0, 6000)
    parts = [s.bids(0, 1234), s.bids(1234, 4000), s.bids(4000, 6000)]
    for k in whole:
        assert np.array_equal(whole[k], np.concatenate([p[k] for p in parts]))
    # shard = slice of the global stream selected by first_event_id
    sh = oracle.NexmarkStream(seed=5, eps=2000, first_event_id=4000).bids(0, 2000)
    assert np.array_equal(sh["auction"], parts[2]["auction"])
    assert np.array_equal(sh["price"], parts[2]["price"])
