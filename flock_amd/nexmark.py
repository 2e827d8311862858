"""NEXMark data source + window launchers on the device.

Mirrors the reference's source side of the hot path:
  * `NEXMarkSource::new(seconds, threads, eps, window)` / `generate_data`
    (flock/src/datasource/nexmark/nexmark.rs:291-389) -> :class:`NEXMarkSource`, whose
    `generate_data` fills HBM-resident columns through `flockgpu_nexmark_gen_*`;
  * `create_nexmark_source` (benchmarks/src/nexmark/main.rs:115-123) fixes the window per query;
  * `window::{elementwise,tumbling,hopping}::launch_tasks`
    (flock-function/src/aws/window/*.rs) -> :func:`window_schedule`, and `run_query`, which executes
    every window of the schedule in a few batched launches instead of one `collect` per window.
The source function forces one generator per stream (nexmark/source.rs:44-48); `first_event_id`
selects the slice of the global stream a shard (GPU rank) owns.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from math import gcd
from typing import Optional, Tuple

import numpy as np

from . import _ffi
from .engine import Auctions, Bids, DeviceUtf8, GpuContext, Persons, WindowSchedule

BASE_TIME = 1_436_918_400_000  # config.rs:20


@dataclass(frozen=True)
class Window:
    """stream/window.rs:80-113 (only the variants the five target queries use)."""
    kind: str
    size: int = 1
    hop: int = 1

    @staticmethod
    def element_wise():
        return Window("elementwise", 1, 1)

    @staticmethod
    def tumbling(seconds: int):
        return Window("tumbling", seconds, seconds)

    @staticmethod
    def hopping(size: int, hop: int):
        return Window("hopping", size, hop)

    @staticmethod
    def session(timeout_seconds: int):
        """stream/window.rs:137 `session_window`: size = the inactivity timeout."""
        return Window("session", timeout_seconds, timeout_seconds)


def query_window(query_number: int) -> Window:  # noqa: C901
    """benchmarks/src/nexmark/main.rs:115-123."""
    if query_number in (0, 1, 2, 3, 4, 6, 9, 10, 13):
        return Window.element_wise()
    if query_number == 5:
        return Window.hopping(10, 5)
    if query_number in (7, 8):
        return Window.tumbling(10)
    if query_number == 11:
        return Window.session(10)
    raise NotImplementedError(f"window of q{query_number} (global, processing time) is outside the hot-path scope")


def window_epochs(window: Window, seconds: int):
    """[(first_epoch, end_epoch)] per window, exactly as the launchers iterate:
    elementwise.rs:46, tumbling.rs:55-57, hopping.rs:54-57 (partial trailing windows are dropped)."""
    if window.kind == "elementwise":
        return [(e, e + 1) for e in range(seconds)]
    if window.kind == "tumbling":
        return [(t * window.size, (t + 1) * window.size) for t in range(seconds // window.size)]
    if window.kind == "hopping":
        out = []
        for t in range(0, seconds, window.hop):
            if t + window.size > seconds:
                break
            out.append((t, t + window.size))
        return out
    raise ValueError(window.kind)


@dataclass
class NEXMarkStream:
    """Device-resident events of one generator, columnar, in event order (epoch-contiguous)."""
    source: "NEXMarkSource"
    bids: Optional[Bids]
    auctions: Optional[Auctions]
    persons: Optional[Persons]

    def epoch_row_offsets(self, relation: str) -> np.ndarray:
        """Row offset of the first event of every epoch 0..seconds for `relation`."""
        s = self.source
        return s.epoch_row_offsets(relation)

    def window_schedule(self, relation: str, window: Optional[Window] = None) -> WindowSchedule:
        s = self.source
        window = window or s.window
        cache = self.__dict__.setdefault("_schedules", {})
        if (relation, window) not in cache:
            cache[(relation, window)] = self._build_schedule(relation, window)
        return cache[(relation, window)]

    def _build_schedule(self, relation: str, window: Window) -> WindowSchedule:
        s = self.source
        epochs = window_epochs(window, s.seconds)
        off = self.epoch_row_offsets(relation)
        pane = gcd(window.size, window.hop) if window.kind == "hopping" else window.size
        n_panes = s.seconds // pane
        pane_off = off[np.arange(n_panes + 1) * pane]
        lo = np.array([a // pane for a, _ in epochs], np.int32)
        hi = np.array([b // pane for _, b in epochs], np.int32)
        return WindowSchedule(pane_off, lo, hi)


class NEXMarkSource:
    """`NEXMarkSource::new(seconds, threads = 1, events_per_second, window)` on the device."""

    def __init__(self, seconds: int, events_per_second: int, window: Window, seed: int = 0, first_event_id: int = 0):
        if events_per_second <= 0 or seconds < 0:
            raise ValueError("seconds >= 0 and events_per_second > 0 required")
        self.seconds, self.eps, self.window = seconds, events_per_second, window
        self.seed, self.first_event_id = seed, first_event_id

    def _stream(self) -> _ffi.NexmarkStream:
        return _ffi.NexmarkStream(self.seed, self.first_event_id, self.eps, BASE_TIME)

    def counts(self, n0: int, n1: int) -> Tuple[int, int, int]:
        lib = _ffi.load()
        p, a, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
        s = self._stream()
        rc = lib.flockgpu_nexmark_counts(C.byref(s), n0, n1, C.byref(p), C.byref(a), C.byref(b))
        if rc != _ffi.OK:
            raise ValueError("bad event range")
        return p.value, a.value, b.value

    def epoch_row_offsets(self, relation: str) -> np.ndarray:
        idx = {"person": 0, "auction": 1, "bid": 2}[relation]
        cache = self.__dict__.setdefault("_epoch_offsets", {})
        if relation not in cache:
            # closed form: events [0, e * eps) hold c(e) rows of the relation
            out = np.empty(self.seconds + 1, np.int64)
            for e in range(self.seconds + 1):
                out[e] = self.counts(0, e * self.eps)[idx]
            cache[relation] = out
        return cache[relation]

    def generate_data(self, ctx: GpuContext, relations=("bid", "auction", "person"),
                      bid_columns=("auction", "bidder", "price", "b_date_time"), auction_times=False) -> NEXMarkStream:
        """`generate_data` (nexmark.rs:357-389) straight into HBM columns."""
        import torch
        dev = f"cuda:{ctx.device}"
        lib = _ffi.load()
        n1 = self.seconds * self.eps
        np_, na, nb = self.counts(0, n1)
        s = self._stream()
        bids = auctions = persons = None
        ptr = lambda t: None if t is None else t.data_ptr()
        if "bid" in relations:
            col = lambda name, dt: torch.empty(nb, dtype=dt, device=dev) if name in bid_columns else None
            bids = Bids(col("auction", torch.int32), col("bidder", torch.int32), col("price", torch.int32),
                        col("b_date_time", torch.int64), nb)
            ctx._check(lib.flockgpu_nexmark_gen_bids(ctx._h, C.byref(s), 0, n1, ptr(bids.auction), ptr(bids.bidder),
                                                     ptr(bids.price), ptr(bids.b_date_time)))
        if "auction" in relations:
            auctions = Auctions(*(torch.empty(na, dtype=torch.int32, device=dev) for _ in range(3)), na)
            ctx._check(lib.flockgpu_nexmark_gen_auctions(ctx._h, C.byref(s), 0, n1, ptr(auctions.a_id),
                                                         ptr(auctions.seller), ptr(auctions.category)))
            if auction_times:   # q4 / q9 scan [a_id, a_date_time, expires (, category)] (q9_plan.fmt)
                auctions.a_date_time = torch.empty(na, dtype=torch.int64, device=dev)
                auctions.expires = torch.empty(na, dtype=torch.int64, device=dev)
                ctx._check(lib.flockgpu_nexmark_gen_auction_times(ctx._h, C.byref(s), 0, n1, ptr(auctions.a_date_time),
                                                                  ptr(auctions.expires)))
        if "person" in relations:
            mk = lambda width: DeviceUtf8(torch.empty(np_ + 1, dtype=torch.int32, device=dev),
                                          torch.empty(max(np_ * width, 16), dtype=torch.uint8, device=dev))
            persons = Persons(torch.empty(np_, dtype=torch.int32, device=dev), mk(14), mk(13), mk(2), np_)
            ctx._check(lib.flockgpu_nexmark_gen_persons(
                ctx._h, C.byref(s), 0, n1, ptr(persons.p_id), ptr(persons.name.offsets), ptr(persons.name.data),
                ptr(persons.city.offsets), ptr(persons.city.data), ptr(persons.state.offsets), ptr(persons.state.data)))
        ctx.synchronize()
        return NEXMarkStream(self, bids, auctions, persons)


# Arrow schemas of the three relations as the json::Reader sees them (flock/src/datasource/nexmark/event.rs:130-149,
# 220-245, 336-352): usize -> Int32, Epoch -> Timestamp(ms) (Int64 here), String -> Utf8.
NEXMARK_JSON_SCHEMAS = {
    "bid": [("auction", "int32"), ("bidder", "int32"), ("price", "int32"), ("b_date_time", "int64")],
    "auction": [("a_id", "int32"), ("item_name", "utf8"), ("description", "utf8"), ("initial_bid", "int32"), ("reserve", "int32"),
                ("a_date_time", "int64"), ("expires", "int64"), ("seller", "int32"), ("category", "int32")],
    "person": [("p_id", "int32"), ("name", "utf8"), ("email_address", "utf8"), ("credit_card", "utf8"), ("city", "utf8"),
               ("state", "utf8"), ("p_date_time", "int64")],
}


def event_bytes_to_columns(ctx: GpuContext, text, relation: str):
    """`event_bytes_to_batch(&event.<relation>, NEXMARK_<RELATION>, ..)` (nexmark.rs:180-205) on the device: the epoch's
    serde_json lines (uint8 device tensor) -> the relation's columns.  Returns ({name: column}, rows)."""
    return ctx.json_lines_decode(text, NEXMARK_JSON_SCHEMAS[relation])


def synthetic_side_input(ctx: GpuContext, stream: NEXMarkStream, stride: int = 6007):
    """A bounded q13 side input for a generated stream: key = every `stride`-th auction id the stream's bids can name,
    value = a function of the key.  (The reference reads the table from a user-supplied CSV in S3,
    benchmarks/src/nexmark/main.rs:44,353-361; its content is not part of the repository.)"""
    import torch
    n_auctions = stream.source.counts(0, stream.source.seconds * stream.source.eps)[1]
    first = 1000 + (stream.source.counts(0, 0)[1] if stream.source.first_event_id == 0 else 0)
    key = torch.arange(first - first % stride + stride, first + n_auctions + stride, stride, dtype=torch.int32,
                       device=f"cuda:{ctx.device}")
    return key, key * 7 + 1


def run_query(ctx: GpuContext, query_number: int, stream: NEXMarkStream, window: Optional[Window] = None, side_input=None):
    """Executes every window of the query's schedule (the per-invocation batch loop of
    flock-function: window launcher + `actor::collect`) and returns the engine's result object."""
    window = window or query_window(query_number)
    if query_number == 1:
        return ctx.q1_project(stream.bids)
    if query_number == 2:
        return ctx.q2_filter(stream.bids, stream.window_schedule("bid", window))
    if query_number == 3:
        return ctx.q3_join(stream.auctions, stream.window_schedule("auction", window), stream.persons,
                           stream.window_schedule("person", window))
    if query_number == 5:
        return ctx.q5_hot_items(stream.bids, stream.window_schedule("bid", window))
    if query_number in (4, 9):
        fn = ctx.q4_avg_final_by_category if query_number == 4 else ctx.q9_winning_bids
        return fn(stream.auctions, stream.window_schedule("auction", window), stream.bids, stream.window_schedule("bid", window))
    if query_number == 13:
        if side_input is None:
            side_input = stream.__dict__.setdefault("_side_input", synthetic_side_input(ctx, stream))
        return ctx.q13_side_join(stream.bids, stream.window_schedule("bid", window), *side_input)
    if query_number == 7:
        return ctx.q7_highest_bid(stream.bids, stream.window_schedule("bid", window))
    if query_number == 11:   # the session launcher's whole walk over the epochs (window/session.rs:185-262)
        if window.kind != "session":
            raise ValueError("q11 runs under Window::Session")
        return ctx.q11_user_sessions(stream.bids, stream.epoch_row_offsets("bid"), window.size, BASE_TIME)
    if query_number == 8:
        return ctx.q8_join(stream.persons, stream.window_schedule("person", window), stream.auctions,
                           stream.window_schedule("auction", window))
    raise NotImplementedError(f"q{query_number} is outside the hot-path scope (SURVEY.md section 8)")


def run_query_async(ctx: GpuContext, query_number: int, stream: NEXMarkStream, window: Optional[Window] = None):
    """`run_query` handed to the context's worker thread (flockgpu_q{3,5,8}_*_async): returns at once with a pending call whose
    `.wait()` gives the result object.  One call in flight per context; calls on different contexts overlap on the GPU -- the
    reference's one tokio task per plan (flock/src/runtime/context.rs:172-191)."""
    window = window or query_window(query_number)
    if query_number == 3:
        return ctx.q3_join_async(stream.auctions, stream.window_schedule("auction", window), stream.persons, stream.window_schedule("person", window))
    if query_number == 5:
        return ctx.q5_hot_items_async(stream.bids, stream.window_schedule("bid", window))
    if query_number == 8:
        return ctx.q8_join_async(stream.persons, stream.window_schedule("person", window), stream.auctions, stream.window_schedule("auction", window))
    raise NotImplementedError(f"q{query_number} has no asynchronous entry point")
