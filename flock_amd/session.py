"""Session windows on the host, the way the reference's launcher keeps them (flock-function/src/aws/window/session.rs:64-321), with the
query of every epoch's closed sessions executed through the plan ABI (`flockgpu_plan_*`).

The reference walks the epochs one by one: the epoch's events are split into one partition per key (HashDiff repartition, :243-248), a
partition joins the key's open session unless its FIRST event lies more than `timeout` whole seconds after the session's LAST event -- then
the session is closed and a new one starts (`add_partitions_to_session_windows`, :64-134) --, after that every open session whose last
event is more than `timeout` whole seconds older than the epoch clock  BASE_TIME / 1000 + epoch  is closed
(`find_timeout_session_windows`, :144-178), and the sessions closed in the epoch are sent to the query function (:262-321).  This module is
that bookkeeping over Arrow batches (row numbers per key, nothing copied until a session closes) and the hand-over to an
`ExecutionContext`; `flockgpu_q11_user_sessions` (include/flockgpu.h) is the same walk for a whole run of epochs on the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

from .nexmark import BASE_TIME


class SessionWindows:
    """Open sessions per key.  `add_epoch` returns the events of the sessions that the epoch closes, as Arrow batches."""

    def __init__(self, timeout_seconds: int, key: str = "bidder", time: str = "b_date_time", base_time_ms: int = BASE_TIME):
        self.timeout, self.key, self.time, self.base_s = int(timeout_seconds), key, time, base_time_ms // 1000
        # key -> [pieces, last event's whole second]; a piece = (batch id, row numbers) into self._batches
        self._open: Dict[int, list] = {}
        # epoch batches that some open session still points into: id -> [batch, pieces pointing into it].  A batch leaves the dict
        # with its last piece, so a long-running stream holds what its OPEN sessions need and nothing else (a list with
        # one slot per epoch forever, rescanned every epoch)
        # (entry = [batch, pieces, live rows, keys with a piece in it]; a batch of which under a quarter is still live -- a hot key that
        # never times out keeps touching every epoch -- is compacted: its live rows move into a small batch of their own)
        self._batches: Dict[int, list] = {}
        self._next_batch = 0

    def _rows_of(self, pieces) -> list:
        out = []
        for b, rows in pieces:
            entry = self._batches[b]
            out.append(entry[0].take(rows))
            entry[1] -= 1
            entry[2] -= len(rows)
            if entry[1] == 0:
                del self._batches[b]
        return out

    def _compact(self):
        import pyarrow as pa
        for b in [b for b, e in self._batches.items() if e[0].num_rows > 256 and e[2] * 4 < e[0].num_rows]:
            batch, _, _, keys = self._batches.pop(b)
            takes, at, moved = [], 0, []
            for k in keys:
                cur = self._open.get(k)
                if cur is None:
                    continue
                for i, (pb, rows) in enumerate(cur[0]):
                    if pb == b:
                        takes.append(rows)
                        moved.append((cur[0], i, at, len(rows)))
                        at += len(rows)
            if not takes:
                continue
            nb = self._next_batch
            self._next_batch += 1
            self._batches[nb] = [batch.take(pa.array(np.concatenate(takes))), len(moved), at, set(keys)]
            for pieces, i, lo, n in moved:
                pieces[i] = (nb, np.arange(lo, lo + n))

    def add_epoch(self, epoch: int, batch) -> list:
        """`batch`: the epoch's events (a pyarrow RecordBatch, or None for an epoch without events).  Returns the closed sessions'
        events in the order the reference collects them: sessions a new partition displaced (in partition order), then the timed-out ones."""
        import pyarrow as pa
        closed: list = []
        if batch is not None and batch.num_rows:
            bi = self._next_batch
            self._next_batch += 1
            entry = self._batches[bi] = [batch, 0, 0, set()]
            keys = batch.column(self.key).to_numpy(zero_copy_only=False)
            secs = batch.column(self.time).cast(pa.int64()).to_numpy(zero_copy_only=False) // 1000
            order = np.argsort(keys, kind="stable")           # one partition per key, arrival order inside it
            bounds = np.flatnonzero(np.r_[True, keys[order][1:] != keys[order][:-1], True])
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                rows = order[lo:hi]
                k = int(keys[rows[0]])
                cur = self._open.get(k)
                if cur is not None and int(secs[rows[0]]) - cur[1] > self.timeout:      # session.rs:118-124
                    closed.append(self._open.pop(k)[0])
                    cur = None
                if cur is None:
                    cur = self._open.setdefault(k, [[], 0])
                cur[0].append((bi, rows))
                entry[1] += 1
                entry[2] += len(rows)
                entry[3].add(k)
                cur[1] = int(secs[rows[-1]])
        now = self.base_s + epoch                                                       # session.rs:163-170
        for k in [k for k, (_, last) in self._open.items() if now - last > self.timeout]:
            closed.append(self._open.pop(k)[0])
        out = [b for pieces in closed for b in self._rows_of(pieces)]
        self._compact()
        return out

    @property
    def held_rows(self) -> int:
        """Rows of the epoch batches (and compacted remainders) still held."""
        return sum(e[0].num_rows for e in self._batches.values())

    @property
    def held_batches(self) -> int:
        """Epoch batches still referenced by an open session."""
        return len(self._batches)

    @property
    def open_sessions(self) -> int:
        return len(self._open)


def launch_session_query(ctx, epochs: List[Optional[object]], timeout_seconds: int, key: str = "bidder", time: str = "b_date_time",
                         base_time_ms: int = BASE_TIME) -> List[list]:
    """`launch_tasks` (session.rs:187-321) in one process: `epochs[t]` = the events of epoch t; every epoch's closed sessions go through
    `collect(ctx, ...)` -- feed_data_sources -> execute -> clean_data_sources on the plan ABI -- together (one function invocation per
    epoch).  Returns, per epoch, the result batches ([] when the epoch closes nothing).  Sessions still open after the last epoch are
    never sent, as in the reference."""
    from .runtime import collect
    windows = SessionWindows(timeout_seconds, key, time, base_time_ms)
    out = []
    for t, batch in enumerate(epochs):
        closed = windows.add_epoch(t, batch)
        out.append(collect(ctx, [[closed]])[0] if closed else [])
    return out
