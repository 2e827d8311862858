"""`Arena` (flock_amd/arena.py) against the reference's own test of it (flock/src/runtime/arena/mod.rs:283-324): the 37 UK
cities in 8 batches of <= 5 rows, one payload per batch, NotReady x 7 then Ready, every sequence number set, `take` returns the
8 payloads' batches.  The CPU half drives the bookkeeping with hand-made payloads; the GPU half goes through to_payload /
to_record_batch (device-side IPC body assembly) and checks the reference's Flight size golden for the whole table
(payload.rs:288-309: 1856 bytes) on real bytes."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

from flock_amd import payload as P
from flock_amd.arena import Arena, HashAggregateStatus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UK = json.load(open(os.path.join(ROOT, "tests", "golden", "uk_cities.json")))


def _uuid(i, n):
    return P.Uuid("SX72HzqFz1Qij4bP-1024-7", i, n)        # UuidBuilder::new_with_ts(function, 1024, n): qid = code-ts-random


def test_arena_bookkeeping_like_the_reference_test():
    arena = Arena()
    n = 8
    for i in range(n):
        pay = P.Payload(data=[P.DataFrame(b"h%d" % i, b"b%d" % i)], schema=b"s", uuid=_uuid(i + 1, n), encoding=P.Encoding("None"))
        status = arena.collect(pay)
        assert status == (HashAggregateStatus.NotReady if i < n - 1 else HashAggregateStatus.Ready)
        assert arena.is_complete(("SX72HzqFz1Qij4bP-1024-7", 0)) == (i == n - 1)
    wid = ("SX72HzqFz1Qij4bP-1024-7", 0)
    w = arena[wid]
    assert w.size == 8 and len(w.r1_flight_data) == 8 and all(arena.get_bitmap(wid).is_set(i + 1) for i in range(8))
    assert not arena.get_bitmap(wid).is_set(0) and arena.get_bitmap(("no exists", 0)) is None
    # a re-delivered payload is recognised by its sequence number and ignored
    again = P.Payload(data=[P.DataFrame(b"x", b"y")], schema=b"s", uuid=_uuid(3, n), encoding=P.Encoding("None"))
    assert arena.collect(again) == HashAggregateStatus.Processed and len(arena[wid].r1_flight_data) == 8
    # another shuffle id is another window; a window of one payload is ready at once
    solo = P.Payload(data=[P.DataFrame(b"x", b"y")], schema=b"s", uuid=_uuid(1, 1), encoding=P.Encoding("None"), shuffle_id=5)
    assert arena.collect(solo) == HashAggregateStatus.Ready and ("SX72HzqFz1Qij4bP-1024-7", 5) in arena
    with pytest.raises(AssertionError):
        arena.collect(P.Payload(uuid=_uuid(2, 9)))                      # seq_len differs inside a window (mod.rs:186)
    assert arena.take(None, ("no exists", 0)) == [[], []]


@pytest.mark.gpu
def test_arena_take_and_the_flight_size_golden_on_real_bytes():
    import torch
    from flock_amd import DeviceUtf8, GpuContext
    ctx = GpuContext(0)

    def batch(lo, hi):
        city = pa.array(UK["city"][lo:hi])
        off = np.frombuffer(city.buffers()[1], np.int32)[: hi - lo + 1].copy()
        data = np.concatenate([np.frombuffer(city.buffers()[2], np.uint8)[: off[-1]], np.zeros(16, np.uint8)])
        return P.DeviceBatch([("city", "utf8"), ("lat", "float64"), ("lng", "float64")],
                             [DeviceUtf8(torch.from_numpy(off).cuda(), torch.from_numpy(data).cuda()),
                              torch.tensor(UK["lat"][lo:hi], dtype=torch.float64).cuda(), torch.tensor(UK["lng"][lo:hi], dtype=torch.float64).cuda()], hi - lo)
    # payload.rs:288-309: the whole table as ONE batch -> header + body = 1856 bytes, now on the bytes actually written
    header, body = P.batch_to_flight_data(ctx, batch(0, 37))
    assert len(header) + len(body) == 1856
    # arena/mod.rs:283-324: 8 batches of 5 rows (the last has 2), one payload each
    arena, n = Arena(), 8
    for i in range(n):
        pay = P.to_payload(ctx, [batch(5 * i, min(5 * i + 5, 37))], [], _uuid(i + 1, n), False)
        pay = P.Payload.from_json(pay.to_json())                         # over the wire
        assert arena.collect(pay) == (HashAggregateStatus.NotReady if i < n - 1 else HashAggregateStatus.Ready)
    wid = ("SX72HzqFz1Qij4bP-1024-7", 0)
    got = arena.take(ctx, wid)
    assert len(got) == 1 and len(got[0]) == 8 and wid not in arena
    cities, lat = [], []
    for (b,) in got[0]:
        off = b.columns[0].offsets.cpu().numpy()
        raw = b.columns[0].data.cpu().numpy()[: off[-1]].tobytes()
        cities += [raw[off[k]:off[k + 1]].decode() for k in range(b.rows)]
        lat += b.columns[1].cpu().numpy().tolist()
    assert cities == UK["city"] and lat == UK["lat"]
    assert len(arena.take(ctx, ("no exists", 0))[0]) == 0
    ctx.close()
