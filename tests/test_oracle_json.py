"""JSON-lines oracle (the reference's epoch buffers and `event_bytes_to_batch`): the lines the oracle writes for generated
events decode back to the same columns, and an independent decoder (Arrow C++'s JSON reader, pyarrow.json) agrees."""
import io

import numpy as np
import pytest

import oracle

SCHEMAS = {
    "bid": [("auction", "int32"), ("bidder", "int32"), ("price", "int32"), ("b_date_time", "int64")],
    "auction": [("a_id", "int32"), ("item_name", "utf8"), ("description", "utf8"), ("initial_bid", "int32"), ("reserve", "int32"),
                ("a_date_time", "int64"), ("expires", "int64"), ("seller", "int32"), ("category", "int32")],
    "person": [("p_id", "int32"), ("name", "utf8"), ("email_address", "utf8"), ("credit_card", "utf8"), ("city", "utf8"),
               ("state", "utf8"), ("p_date_time", "int64")],
}


def _events(relation, n_events=4000, seed=3):
    s = oracle.NexmarkStream(seed=seed, eps=1000)
    return {"bid": lambda: s.bids(0, n_events), "auction": lambda: s.auctions(0, n_events, strings=True),
            "person": lambda: s.persons(0, n_events, filler=True)}[relation]()


def _same(a, b):
    if isinstance(a, oracle.Utf8):
        return np.array_equal(a.offsets, b.offsets) and np.array_equal(a.data[: a.offsets[-1]], b.data[: b.offsets[-1]])
    return np.array_equal(a, b)


@pytest.mark.parametrize("relation", ["bid", "auction", "person"])
def test_lines_round_trip_and_agree_with_arrow(relation):
    pj = pytest.importorskip("pyarrow.json")
    cols = _events(relation)
    text = oracle.nexmark_json_lines(relation, cols)
    assert text.count(b"\n") == len(cols[SCHEMAS[relation][0][0]]) and b" " not in text.split(b"\n")[0].split(b'"')[0]
    got = oracle.json_lines_decode(text, SCHEMAS[relation])
    for name, _ in SCHEMAS[relation]:
        assert _same(got[name], cols[name]), name
    table = pj.read_json(io.BytesIO(text))
    for name, t in SCHEMAS[relation]:
        col = table.column(name).to_pylist()
        if t == "utf8":
            b, off = got[name].data.tobytes(), got[name].offsets
            assert col == [b[off[i]:off[i + 1]].decode() for i in range(len(col))], name
        else:
            assert col == got[name].tolist(), name
