"""`Arena` -- the window-session store of a function that aggregates the partial payloads of a window before it runs its
plan (flock/src/runtime/arena/mod.rs:47-241).  Same surface and argument meaning:

    arena = Arena()
    status = arena.collect(payload)         # HashAggregateStatus.{NotReady, Ready, Processed}      (mod.rs:180-232)
    arena.is_complete(window_id)            #                                                       (mod.rs:163-167)
    arena.get_bitmap(window_id)             # which sequence numbers arrived                        (mod.rs:158-160)
    r1, r2 = arena.take(ctx, window_id)     # [payload][batch] per relation, decoded onto the device (mod.rs:89-155)

A window is (query id, shuffle id) (`Payload::get_window_id`, flock/src/runtime/payload.rs:200-202); the payloads of one window
carry `Uuid{qid, seq_num in 1..=seq_len, seq_len}` (`UuidBuilder`, payload.rs:46-84): a sequence number that is already set is a
re-delivery and is ignored, the window is ready when `seq_len` distinct payloads are in.  `take` hands the Flight data of both
relations to `Payload.to_record_batch`'s decoder -- here the body goes to HBM once and the columns are views into it
(flock_amd/payload.py) -- and forgets the window.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from .payload import DataFrame, Encoding, Payload, Uuid, flight_data_to_batch, schema_from_bytes, _kind_of

WindowId = Tuple[str, int]


class HashAggregateStatus(enum.Enum):
    Processed = "Processed"     # this payload was collected before
    Ready = "Ready"             # the window is complete
    NotReady = "NotReady"


class Bitmap:
    """Bit per sequence number (flock/src/runtime/arena/bitmap.rs): `Bitmap::new(seq_len + 1)`, numbers start at 1."""

    def __init__(self, size: int):
        # bitmap.rs:35-47: the byte count is rounded up to a multiple of 64, and BOTH accessors assert `i < capacity` in bits --
        # one bound for `set` and `is_set`, so a sequence number in the slack is recognised when it is delivered again
        self.size = size
        n_bytes = (size + 7) // 8
        self._bits = bytearray((n_bytes + 63) // 64 * 64)

    @property
    def capacity(self) -> int:
        return len(self._bits) << 3

    def set(self, i: int):
        assert 0 <= i < self.capacity, f"bit {i} outside a bitmap of {self.capacity} bits"     # bitmap.rs:58
        self._bits[i >> 3] |= 1 << (i & 7)

    def is_set(self, i: int) -> bool:
        assert 0 <= i < self.capacity, f"bit {i} outside a bitmap of {self.capacity} bits"     # bitmap.rs:52
        return bool(self._bits[i >> 3] >> (i & 7) & 1)


@dataclass
class WindowSession:
    size: int
    r1_flight_data: List[List[DataFrame]] = field(default_factory=list)
    r1_schema: bytes = b""
    r2_flight_data: List[List[DataFrame]] = field(default_factory=list)
    r2_schema: bytes = b""
    bitmap: Optional[Bitmap] = None
    encoding: Encoding = field(default_factory=Encoding)


def window_id_of(payload: Payload) -> WindowId:
    return payload.uuid.qid, int(payload.shuffle_id or 0)


class Arena(Dict[WindowId, WindowSession]):
    def get_bitmap(self, window_id: WindowId) -> Optional[Bitmap]:
        w = self.get(window_id)
        return w.bitmap if w else None

    def is_complete(self, window_id: WindowId) -> bool:
        w = self.get(window_id)
        return bool(w) and w.size == len(w.r1_flight_data)

    def collect(self, payload: Payload) -> HashAggregateStatus:
        uuid: Uuid = payload.uuid
        wid = window_id_of(payload)
        w = self.get(wid)
        if w is None:
            w = WindowSession(uuid.seq_len, [payload.data], payload.schema, [payload.data2], payload.schema2,
                              Bitmap(uuid.seq_len + 1), payload.encoding)
            w.bitmap.set(uuid.seq_num)
            self[wid] = w
            return HashAggregateStatus.Ready if uuid.seq_len == 1 else HashAggregateStatus.NotReady
        if uuid.seq_len != w.size:
            raise AssertionError("payloads of one window disagree on seq_len")       # mod.rs:186 `assert!`
        if w.bitmap.is_set(uuid.seq_num):
            return HashAggregateStatus.Processed
        w.r1_flight_data.append(payload.data)
        w.r2_flight_data.append(payload.data2)
        w.bitmap.set(uuid.seq_num)
        return HashAggregateStatus.Ready if w.size == len(w.r1_flight_data) else HashAggregateStatus.NotReady

    def take(self, ctx, window_id: WindowId):
        """[relation][payload][batch] as device batches; an unknown window gives [[], []] (mod.rs:152-154)."""
        w = self.pop(window_id, None)
        if w is None:
            return [[], []]
        if not w.r1_schema:
            raise ValueError("Record batches are empty.")                            # mod.rs:66-70

        def side(frames_per_payload, schema_bytes):
            schema = schema_from_bytes(schema_bytes)
            fields = [(f.name, _kind_of(f.type)) for f in schema]
            return [[flight_data_to_batch(ctx, w.encoding.decompress(f.header), w.encoding.decompress(f.body), fields) for f in frames]
                    for frames in frames_per_payload if frames]
        out = [side(w.r1_flight_data, w.r1_schema)]
        if w.r2_schema:
            out.append(side(w.r2_flight_data, w.r2_schema))
        return out
