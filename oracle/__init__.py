"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/liboracle.so`` (the scalar C restatement of Flock's
NEXMark generator and of the DataFusion operators behind q1/q2/q3/q5/q8).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  ``flock_amd`` never does.

Reference call sites restated: flock/src/datasource/nexmark/{event,config,generator}.rs
(generator) and flock-function/src/aws/actor.rs:54-79 (`collect`, one window per call).
NEXMark result parity is *unpinned* by the reference's own tests (they only print);
see oracle/nexmark_ops.c for how this oracle is pinned transitively.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
BASE_TIME = 1_436_918_400_000  # config.rs:20


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in ("nexmark_gen.c", "nexmark_ops.c", "nexmark_exp2_table.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Stream(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_event_id", C.c_uint64), ("eps", C.c_uint64), ("base_time", C.c_uint64)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        u64, vp = C.c_uint64, C.c_void_p
        _lib.oracle_nexmark_counts.argtypes = [u64, u64, u64, vp, vp, vp]
        _lib.oracle_nexmark_counts.restype = None
        _lib.oracle_nexmark_gen_bids.argtypes = [vp, u64, u64, vp, vp, vp, vp]
        _lib.oracle_nexmark_gen_bids.restype = u64
        _lib.oracle_nexmark_gen_auctions.argtypes = [vp, u64, u64] + [vp] * 11
        _lib.oracle_nexmark_gen_auctions.restype = u64
        _lib.oracle_nexmark_gen_persons.argtypes = [vp, u64, u64] + [vp] * 12
        _lib.oracle_nexmark_gen_persons.restype = u64
        _lib.oracle_q1_project.argtypes = [vp, u64, vp]
        _lib.oracle_q1_project.restype = None
        _lib.oracle_q2_filter.argtypes = [vp, vp, u64, C.c_int64, vp, vp]
        _lib.oracle_q2_filter.restype = u64
        _lib.oracle_q3_join.argtypes = [vp, vp, u64, C.c_int64, vp, vp, vp, u64, vp, C.c_int, vp, vp]
        _lib.oracle_q3_join.restype = u64
        _lib.oracle_q5_hot_items.argtypes = [vp, u64, vp, vp, u64]
        _lib.oracle_q5_hot_items.restype = u64
        _lib.oracle_count_by_key.argtypes = [vp, u64, vp, vp, u64]
        _lib.oracle_count_by_key.restype = u64
        _lib.oracle_q8_join.argtypes = [vp, vp, vp, u64, vp, u64, vp]
        _lib.oracle_q8_join.restype = u64
        _lib.oracle_take_utf8.argtypes = [vp, vp, vp, u64, vp, vp]
        _lib.oracle_take_utf8.restype = u64
        _lib.oracle_hash_utf8_rows.argtypes = [vp, vp, vp, u64, vp]
        _lib.oracle_hash_utf8_rows.restype = None
        _lib.oracle_ysb_campaign_counts.argtypes = [vp, vp, vp, vp, u64, C.c_char_p, u64, vp, vp, vp, u64, vp]
        _lib.oracle_ysb_campaign_counts.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class Utf8:
    """Arrow Utf8 column: int32 offsets (rows + 1) + bytes."""

    offsets: np.ndarray
    data: np.ndarray

    def __len__(self):
        return len(self.offsets) - 1

    def to_pylist(self):
        b = self.data.tobytes()
        o = self.offsets
        return [b[o[i]:o[i + 1]].decode() for i in range(len(self))]

    def slice(self, lo, hi):
        o = self.offsets[lo:hi + 1]
        return Utf8((o - o[0]).astype(np.int32), self.data[o[0]:o[-1]])


# ---------------------------------------------------------------- generator
@dataclass
class NexmarkStream:
    """One generator's event stream (threads = 1 per source function,
    flock-function/src/aws/nexmark/source.rs:44-48)."""

    seed: int = 0
    first_event_id: int = 0
    eps: int = 1000
    base_time: int = BASE_TIME

    def _c(self):
        return _Stream(self.seed, self.first_event_id, self.eps, self.base_time)

    def counts(self, n0, n1):
        p, a, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().oracle_nexmark_counts(self.first_event_id, n0, n1, C.byref(p), C.byref(a), C.byref(b))
        return p.value, a.value, b.value

    def bids(self, n0, n1, columns=("auction", "bidder", "price", "b_date_time")):
        _, _, nb = self.counts(n0, n1)
        out = {
            "auction": np.empty(nb, np.int32) if "auction" in columns else None,
            "bidder": np.empty(nb, np.int32) if "bidder" in columns else None,
            "price": np.empty(nb, np.int32) if "price" in columns else None,
            "b_date_time": np.empty(nb, np.int64) if "b_date_time" in columns else None,
        }
        s = self._c()
        rows = lib().oracle_nexmark_gen_bids(C.byref(s), n0, n1, *(_p(out[k]) for k in ("auction", "bidder", "price", "b_date_time")))
        assert rows == nb
        return {k: v for k, v in out.items() if v is not None}

    def auctions(self, n0, n1, strings=False):
        _, na, _ = self.counts(n0, n1)
        cols = {k: np.empty(na, np.int32) for k in ("a_id", "initial_bid", "reserve", "seller", "category")}
        cols["a_date_time"] = np.empty(na, np.int64)
        cols["expires"] = np.empty(na, np.int64)
        io = ib = do = db = None
        if strings:
            io, ib = np.empty(na + 1, np.int32), np.empty(max(na * 19, 1), np.uint8)
            do, db = np.empty(na + 1, np.int32), np.empty(max(na * 99, 1), np.uint8)
        s = self._c()
        rows = lib().oracle_nexmark_gen_auctions(
            C.byref(s), n0, n1, _p(cols["a_id"]), _p(cols["initial_bid"]), _p(cols["reserve"]), _p(cols["a_date_time"]),
            _p(cols["expires"]), _p(cols["seller"]), _p(cols["category"]), _p(io), _p(ib), _p(do), _p(db))
        assert rows == na
        if strings:
            cols["item_name"] = Utf8(io, ib[: io[-1]].copy())
            cols["description"] = Utf8(do, db[: do[-1]].copy())
        return cols

    def persons(self, n0, n1, filler=False):
        np_, _, _ = self.counts(n0, n1)
        cols = {"p_id": np.empty(np_, np.int32), "p_date_time": np.empty(np_, np.int64)}
        no, nb = np.empty(np_ + 1, np.int32), np.empty(max(np_ * 14, 1), np.uint8)
        co, cb = np.empty(np_ + 1, np.int32), np.empty(max(np_ * 13, 1), np.uint8)
        so, sb = np.empty(np_ + 1, np.int32), np.empty(max(np_ * 2, 1), np.uint8)
        eo = eb = cco = ccb = None
        if filler:
            eo, eb = np.empty(np_ + 1, np.int32), np.empty(max(np_ * 15, 1), np.uint8)
            cco, ccb = np.empty(np_ + 1, np.int32), np.empty(max(np_ * 19, 1), np.uint8)
        s = self._c()
        rows = lib().oracle_nexmark_gen_persons(
            C.byref(s), n0, n1, _p(cols["p_id"]), _p(cols["p_date_time"]), _p(no), _p(nb), _p(eo), _p(eb),
            _p(cco), _p(ccb), _p(co), _p(cb), _p(so), _p(sb))
        assert rows == np_
        cols["name"] = Utf8(no, nb[: no[-1]].copy())
        cols["city"] = Utf8(co, cb[: co[-1]].copy())
        cols["state"] = Utf8(so, sb[: so[-1]].copy())
        if filler:
            cols["email_address"] = Utf8(eo, eb[: eo[-1]].copy())
            cols["credit_card"] = Utf8(cco, ccb[: cco[-1]].copy())
        return cols


# ---------------------------------------------------------------- operators (one window per call)
def q1_project(price: np.ndarray) -> np.ndarray:
    price = np.ascontiguousarray(price, np.int32)
    out = np.empty(len(price), np.float64)
    lib().oracle_q1_project(_p(price), len(price), _p(out))
    return out


def q2_filter(auction: np.ndarray, price: np.ndarray, modulus: int = 123):
    auction = np.ascontiguousarray(auction, np.int32)
    price = np.ascontiguousarray(price, np.int32)
    oa, op = np.empty(len(auction), np.int32), np.empty(len(auction), np.int32)
    m = lib().oracle_q2_filter(_p(auction), _p(price), len(auction), modulus, _p(oa), _p(op))
    return oa[:m].copy(), op[:m].copy()


def q3_join(seller, category, p_id, state: Utf8, category_lit=10, state_lits=("or", "id", "ca")):
    """Returns (auction_row, person_row) int64 pairs in DataFusion probe order."""
    seller = np.ascontiguousarray(seller, np.int32)
    category = np.ascontiguousarray(category, np.int32)
    p_id = np.ascontiguousarray(p_id, np.int32)
    lits = (C.c_char_p * len(state_lits))(*[s.encode() for s in state_lits])
    args = (_p(seller), _p(category), len(seller), category_lit, _p(p_id), _p(state.offsets), _p(state.data),
            len(p_id), C.cast(lits, C.c_void_p), len(state_lits))
    n = lib().oracle_q3_join(*args, None, None)
    ar, pr = np.empty(n, np.int64), np.empty(n, np.int64)
    n2 = lib().oracle_q3_join(*args, _p(ar), _p(pr))
    assert n2 == n
    return ar, pr


def q5_hot_items(auction: np.ndarray):
    """(auction Int32, num UInt64) rows with num == MAX(num); ties kept."""
    auction = np.ascontiguousarray(auction, np.int32)
    cap = 1024
    while True:
        oa, on = np.empty(cap, np.int32), np.empty(cap, np.uint64)
        n = lib().oracle_q5_hot_items(_p(auction), len(auction), _p(oa), _p(on), cap)
        if n <= cap:
            return oa[:n].copy(), on[:n].copy()
        cap = int(n)


def count_by_key(key: np.ndarray):
    key = np.ascontiguousarray(key, np.int32)
    n = lib().oracle_count_by_key(_p(key), len(key), None, None, 0)
    ok, oc = np.empty(n, np.int32), np.empty(n, np.uint64)
    lib().oracle_count_by_key(_p(key), len(key), _p(ok), _p(oc), n)
    return ok, oc


def q7_highest_bid(price: np.ndarray) -> np.ndarray:
    """Row indices of the q7 output for one window, in input order: bid JOIN (SELECT MAX(price) ...) ON price = maxprice
    (benchmarks/src/nexmark/query/q7.sql, q7_plan.fmt).  Every row reaching the maximum is returned (inner join on
    equality); MAX over an empty window is NULL, so the join is empty."""
    price = np.ascontiguousarray(price, np.int32)
    if len(price) == 0:
        return np.zeros(0, np.int64)
    return np.nonzero(price == price.max())[0].astype(np.int64)


def q13_side_join(b_auction, side_key):
    """(bid_row, side_row) pairs of  bid JOIN side_input ON auction = key  for one window, ordered by bid row
    (benchmarks/src/nexmark/query/q13.sql); duplicate keys on the side input produce one pair each."""
    side_key = np.asarray(side_key, np.int64)
    b_auction = np.asarray(b_auction, np.int64)
    order = np.argsort(side_key, kind="stable")
    keys = side_key[order]
    lo, hi = np.searchsorted(keys, b_auction, "left"), np.searchsorted(keys, b_auction, "right")
    n = hi - lo
    bid_row = np.repeat(np.arange(len(b_auction)), n)
    pos = np.arange(int(n.sum())) - np.repeat(np.cumsum(n) - n, n) + np.repeat(lo, n)
    return bid_row.astype(np.int64), order[pos].astype(np.int64)


def _auction_bid_pairs(a_id, a_date_time, expires, b_auction, b_date_time):
    """(auction_row, bid_row) pairs of  auction INNER JOIN bid ON a_id = auction  WHERE b_date_time BETWEEN a_date_time AND
    expires  (q4.sql / q9.sql inner query; any number of duplicate keys on either side)."""
    a_id = np.asarray(a_id, np.int64)
    b_auction = np.asarray(b_auction, np.int64)
    order = np.argsort(a_id, kind="stable")
    keys = a_id[order]
    lo, hi = np.searchsorted(keys, b_auction, "left"), np.searchsorted(keys, b_auction, "right")
    n = hi - lo
    bid_row = np.repeat(np.arange(len(b_auction)), n)
    pos = np.arange(int(n.sum())) - np.repeat(np.cumsum(n) - n, n) + np.repeat(lo, n)
    auc_row = order[pos]
    when = np.asarray(b_date_time, np.int64)[bid_row]
    keep = (when >= np.asarray(a_date_time, np.int64)[auc_row]) & (when <= np.asarray(expires, np.int64)[auc_row])
    return auc_row[keep], bid_row[keep]


def _group_max(keys: np.ndarray, values: np.ndarray):
    """(distinct keys ascending, MAX(values) per key)."""
    uniq, inv = np.unique(keys, return_inverse=True, axis=0)
    best = np.full(len(uniq), np.iinfo(np.int64).min, np.int64)
    np.maximum.at(best, inv.reshape(-1), values)
    return uniq, best


def q9_winning_bids(a_id, a_date_time, expires, b_auction, b_price, b_date_time) -> np.ndarray:
    """Bid rows (input order) of q9 for one window (benchmarks/src/nexmark/query/q9.sql, q9_plan.fmt):
    Q = MAX(price) GROUP BY a_id over the filtered join; result = bids with (auction, price) = (Q.id, Q.final) --
    the outer join does not repeat the BETWEEN."""
    ar, br = _auction_bid_pairs(a_id, a_date_time, expires, b_auction, b_date_time)
    b_price, b_auction = np.asarray(b_price, np.int64), np.asarray(b_auction, np.int64)
    if len(ar) == 0:
        return np.zeros(0, np.int64)
    ids, final = _group_max(np.asarray(a_id, np.int64)[ar], b_price[br])
    pos = np.searchsorted(ids, b_auction)
    pos[pos == len(ids)] = 0
    keep = (ids[pos] == b_auction) & (final[pos] == b_price)
    return np.nonzero(keep)[0].astype(np.int64)


def q6_avg_price_by_seller(a_id, a_date_time, expires, seller, b_auction, b_price, b_date_time, last=10):
    """(seller Int32, AVG(price) Float64) rows of q6 for one window, ordered by seller (benchmarks/src/nexmark/query/q6.sql, q6_plan.fmt): per auction the
    winning bid -- ROW_NUMBER() OVER (PARTITION BY a_id ORDER BY price DESC) = 1 among the bids placed while the auction was open; equal top
    prices: the first in (auction row, bid row) order --, per seller the `last` winners with the latest b_date_time (ROW_NUMBER() OVER (PARTITION BY seller
    ORDER BY b_date_time DESC) <= last; equal times: the earlier auction id first), AVG over their prices as Float64 sum / UInt64 count.
    Whole-column numpy; oracle/generic_ops.py: nexmark_q6 walks the same plan operator by operator (tests/test_oracle_q6.py)."""
    ar, br = _auction_bid_pairs(a_id, a_date_time, expires, b_auction, b_date_time)
    if len(ar) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.float64)
    a_id, seller = np.asarray(a_id, np.int64), np.asarray(seller, np.int64)
    price, when = np.asarray(b_price, np.int64)[br], np.asarray(b_date_time, np.int64)[br]
    o = np.lexsort((-price, a_id[ar]))                       # stable: a_id ASC, price DESC, then pair order
    first = np.r_[True, a_id[ar][o][1:] != a_id[ar][o][:-1]]
    w = o[first]                                             # the winning pair of every auction, by a_id
    ws, wp, wt = seller[ar][w], price[w], when[w]
    o2 = np.lexsort((-wt, ws))                               # stable: seller ASC, b_date_time DESC, then a_id order
    s2 = ws[o2]
    start = np.flatnonzero(np.r_[True, s2[1:] != s2[:-1]])
    rank = np.arange(len(s2)) - np.repeat(start, np.diff(np.r_[start, len(s2)])) + 1
    keep = rank <= last
    ks, kp = s2[keep], wp[o2][keep]
    sellers, st = np.unique(ks, return_index=True)
    sums = np.add.reduceat(kp, st)
    counts = np.diff(np.r_[st, len(ks)])
    return sellers.astype(np.int32), sums.astype(np.float64) / counts.astype(np.float64)


def q4_avg_final_by_category(a_id, category, a_date_time, expires, b_auction, b_price, b_date_time):
    """(category Int32, AVG(final) Float64) rows of q4 for one window, ordered by category (q4.sql; stages in
    flock/src/distributed_plan/planner.rs:218-256).  Inner groups are (a_id, category); AVG = Float64 sum / UInt64 count
     -- the sums here are integers far below 2^53, so the division is the only rounding."""
    ar, br = _auction_bid_pairs(a_id, a_date_time, expires, b_auction, b_date_time)
    if len(ar) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.float64)
    a_id, category, b_price = np.asarray(a_id, np.int64), np.asarray(category, np.int64), np.asarray(b_price, np.int64)
    groups, final = _group_max(np.stack([category[ar], a_id[ar]], axis=1), b_price[br])   # sorted by (category, a_id)
    cats, start = np.unique(groups[:, 0], return_index=True)
    sums = np.add.reduceat(final, start)
    counts = np.diff(np.append(start, len(final)))
    return cats.astype(np.int32), sums.astype(np.float64) / counts.astype(np.float64)


def q8_join(p_id, name: Utf8, seller):
    """Row indices (into the window's person rows) of the output, in row order."""
    p_id = np.ascontiguousarray(p_id, np.int32)
    seller = np.ascontiguousarray(seller, np.int32)
    out = np.empty(len(p_id), np.int64)
    n = lib().oracle_q8_join(_p(p_id), _p(name.offsets), _p(name.data), len(p_id), _p(seller), len(seller), _p(out))
    return out[:n].copy()


def take_utf8(col: Utf8, rows: np.ndarray) -> Utf8:
    rows = np.ascontiguousarray(rows, np.int64)
    off = np.empty(len(rows) + 1, np.int32)
    nbytes = lib().oracle_take_utf8(_p(col.offsets), _p(col.data), _p(rows), len(rows), _p(off), None)
    data = np.empty(max(nbytes, 1), np.uint8)
    lib().oracle_take_utf8(_p(col.offsets), _p(col.data), _p(rows), len(rows), _p(off), _p(data))
    return Utf8(off, data[:nbytes])


# ---------------------------------------------------------------- result fingerprints (tests/golden/nexmark_hashes.json)
def _fmix64(x):
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def column_hashes(col, rows=None) -> np.ndarray:
    """64-bit hash per value of one column (optionally of `col.take(rows)`): integers as their int64 value, Utf8 by bytes."""
    if isinstance(col, Utf8):
        n = len(col) if rows is None else len(rows)
        out = np.empty(n, np.uint64)
        r = None if rows is None else np.ascontiguousarray(rows, np.int64)
        off, data = np.ascontiguousarray(col.offsets, np.int32), np.ascontiguousarray(col.data, np.uint8)
        lib().oracle_hash_utf8_rows(_p(off), _p(data) if len(data) else None, _p(r), n, _p(out))
        return out
    v = np.asarray(col)
    if rows is not None:
        v = v[np.asarray(rows, np.int64)]
    if v.dtype.kind == "f":
        v = np.ascontiguousarray(v, np.float64).view(np.int64)       # the BITS of a Float64 column
    with np.errstate(over="ignore"):
        return _fmix64(v.astype(np.int64).view(np.uint64) + np.uint64(0x9E3779B97F4A7C15))


def row_hashes(columns) -> np.ndarray:
    """Hash per result row: the per-column hashes chained left to right (column order matters, row order does not).
    An entry is a column, or (column, rows) for `column.take(rows)` -- join outputs name rows of two relations."""
    r = None
    for c in columns:
        h = column_hashes(*c) if isinstance(c, tuple) else column_hashes(c)
        r = _fmix64((np.uint64(0x243F6A8885A308D3) if r is None else r) ^ h)
    return r


def multiset_fingerprint(columns) -> str:
    """'rows:sum64' of one window's result rows -- equal for equal row multisets whatever the order."""
    h = row_hashes(columns)
    with np.errstate(over="ignore"):
        return f"{len(h)}:{int(h.sum(dtype=np.uint64)):016x}"


# ---------------------------------------------------------------- window schedules
def elementwise_windows(seconds):
    """One window per 1-s epoch (flock-function/src/aws/window/elementwise.rs:46)."""
    return [(e, e + 1) for e in range(seconds)]


def tumbling_windows(seconds, size):
    """flock-function/src/aws/window/tumbling.rs:55-57."""
    return [(t * size, t * size + size) for t in range(seconds // size)]


def hopping_windows(seconds, size, hop):
    """Only full windows (flock-function/src/aws/window/hopping.rs:54-57)."""
    out = []
    for t in range(0, seconds, hop):
        if t + size > seconds:
            break
        out.append((t, t + size))
    return out


# ---------------------------------------------------------------- Yahoo Streaming Benchmark (SURVEY.md section 8 f, rank 4)
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x):
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _ysb_uuid(seed: int, tag: int, idx) -> np.ndarray:
    """36-byte version-4 UUID strings uuid(tag, i) of flock_amd/csrc/ysb_gen.hip, one row of bytes per index."""
    idx = np.asarray(idx, np.uint64)
    with np.errstate(over="ignore"):
        hi = _mix64(_mix64(np.uint64(seed) ^ (np.uint64(tag) * np.uint64(0xA24BAED4963EE407))) + (idx + np.uint64(1)) * np.uint64(0x9FB21C651E98DF25))
        lo = _mix64(hi ^ np.uint64(0xC2B2AE3D27D4EB4F))
    hi = (hi & ~np.uint64(0xF000)) | np.uint64(0x4000)
    lo = (lo & ~(np.uint64(3) << np.uint64(62))) | (np.uint64(2) << np.uint64(62))
    out = np.empty((len(idx), 36), np.uint8)
    hexd = np.frombuffer(b"0123456789abcdef", np.uint8)
    o = 0
    for nib in range(32):
        if nib in (8, 12, 16, 20):
            out[:, o] = ord("-")
            o += 1
        v = hi if nib < 16 else lo
        out[:, o] = hexd[((v >> np.uint64(60 - 4 * (nib & 15))) & np.uint64(15)).astype(np.int64)]
        o += 1
    return out


def _ysb_draw(seed: int, n, k: int):
    with np.errstate(over="ignore"):
        return _mix64(_mix64(np.uint64(seed) ^ (np.asarray(n, np.uint64) * np.uint64(0xD6E8FEB86659FD93))) + np.uint64((k + 1) * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF))


def _ysb_uni(r, n: int):
    return ((r >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)


def ysb_campaigns(seed: int, n_campaigns: int, ads: int):
    """(c_ad_id, campaign_id) Utf8 columns of the campaign table (generator.rs:46-56 restated, see ysb_gen.hip)."""
    rows = n_campaigns * ads
    i = np.arange(rows, dtype=np.uint64)
    off = (np.arange(rows + 1) * 36).astype(np.int32)
    return Utf8(off, _ysb_uuid(seed, 1, i).reshape(-1)), Utf8(off.copy(), _ysb_uuid(seed, 2, i // np.uint64(ads)).reshape(-1))


def ysb_events(seed: int, first: int, n: int, n_ads: int):
    """(ad_id, event_type) Utf8 columns of ad events [first, first + n) (generator.rs:76-96 restated)."""
    idx = np.arange(first, first + n, dtype=np.uint64)
    ad = _ysb_uuid(seed, 1, _ysb_uni(_ysb_draw(seed, idx, 0), n_ads))
    kinds = [b"view", b"click", b"purchase"]
    t = _ysb_uni(_ysb_draw(seed, idx, 1), 3).astype(np.int64)
    lens = np.array([4, 5, 8])[t]
    et_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    et = np.frombuffer(b"".join(kinds[k] for k in t.tolist()), np.uint8).copy() if n else np.zeros(0, np.uint8)
    return Utf8((np.arange(n + 1) * 36).astype(np.int32), ad.reshape(-1)), Utf8(et_off, et)


def ysb_campaign_counts(ad_id: Utf8, event_type: Utf8, c_ad_id: Utf8, campaign_id: Utf8, lit: bytes = b"view"):
    """{campaign_id bytes: COUNT(*)} of ysb.sql for one window (benchmarks/src/ysb/ysb.sql): filter event_type = lit,
    inner join ad_id = c_ad_id (every matching campaign row counts), group by campaign_id."""
    def rows(u):
        b = u.data.tobytes()
        return [b[u.offsets[i]:u.offsets[i + 1]] for i in range(len(u))]
    by_ad = {}
    for a, c in zip(rows(c_ad_id), rows(campaign_id)):
        by_ad.setdefault(a, []).append(c)
    out = {}
    for a, t in zip(rows(ad_id), rows(event_type)):
        if t == lit:
            for c in by_ad.get(a, ()):
                out[c] = out.get(c, 0) + 1
    return out


def _utf8_rows(u: "Utf8"):
    b = u.data.tobytes()
    return [b[u.offsets[i]:u.offsets[i + 1]] for i in range(len(u))]


def ysb_campaign_groups(campaign_id: "Utf8"):
    """(group number per campaign row, the groups' campaign_id bytes): rows with equal campaign_id share a number."""
    names, group = {}, np.empty(len(campaign_id), np.int32)
    for r, c in enumerate(_utf8_rows(campaign_id)):
        group[r] = names.setdefault(c, len(names))
    return group, list(names)


def ysb_campaign_counts_c(ad_id: "Utf8", event_type: "Utf8", c_ad_id: "Utf8", campaign_id: "Utf8", lit: bytes = b"view", groups=None):
    """ysb.sql for one window through the scalar C twin (oracle_ysb_campaign_counts): same {campaign_id bytes: COUNT(*)} as the
    dict walk above, at C speed (ctypes releases the GIL: bench.py runs one window per thread).  `groups`: ysb_campaign_groups(campaign_id)
    when the caller runs many windows against one campaign table."""
    group, names = groups if groups is not None else ysb_campaign_groups(campaign_id)
    counts = np.zeros(max(len(names), 1), np.uint64)
    ad_off, et_off = np.ascontiguousarray(ad_id.offsets, np.int32), np.ascontiguousarray(event_type.offsets, np.int32)
    ad_data, et_data = np.ascontiguousarray(ad_id.data), np.ascontiguousarray(event_type.data)
    c_off, c_data = np.ascontiguousarray(c_ad_id.offsets, np.int32), np.ascontiguousarray(c_ad_id.data)
    rc = lib().oracle_ysb_campaign_counts(_p(ad_off), _p(ad_data), _p(et_off), _p(et_data), len(ad_id), lit, len(lit), _p(c_off), _p(c_data),
                                          _p(np.ascontiguousarray(group)), len(c_ad_id), _p(counts))
    if rc != 0:
        raise MemoryError("oracle_ysb_campaign_counts")
    return {names[g]: int(counts[g]) for g in range(len(names)) if counts[g]}


def ysb_campaign_counts_arrow(ad_id: "Utf8", event_type: "Utf8", c_ad_id: "Utf8", campaign_id: "Utf8", lit: bytes = b"view"):
    """The same window through Arrow C++ (pyarrow compute / Acero: filter, hash join, group_by count_all) -- the independent second
    engine the YSB goldens are minted against (the reference's own differential YSB test compares its distributed run with a local
    DataFusion run, flock/src/launcher/aws/mod.rs:681-844; DataFusion is not buildable here)."""
    import pyarrow as pa
    import pyarrow.compute as pc

    def arr(u, binary=True):
        return pa.Array.from_buffers(pa.binary(), len(u), [None, pa.py_buffer(np.ascontiguousarray(u.offsets, np.int32)), pa.py_buffer(np.ascontiguousarray(u.data))])
    ev = pa.table({"ad_id": arr(ad_id), "event_type": arr(event_type)})
    ev = ev.filter(pc.equal(ev["event_type"], pa.scalar(lit, pa.binary())))
    ca = pa.table({"c_ad_id": arr(c_ad_id), "campaign_id": arr(campaign_id)})
    j = ev.join(ca, keys="ad_id", right_keys="c_ad_id", join_type="inner")
    g = j.group_by("campaign_id", use_threads=False).aggregate([([], "count_all")])
    return dict(zip(g["campaign_id"].to_pylist(), g["count_all"].to_pylist()))


def ysb_fingerprint(counts: dict) -> str:
    """Order-free fingerprint of one window's {campaign_id: count} (rows:sum of fmix64 row hashes, as multiset_fingerprint)."""
    names = sorted(counts)
    if not names:
        return "0:0000000000000000"
    off = np.concatenate(([0], np.cumsum([len(c) for c in names]))).astype(np.int32)
    col = Utf8(off, np.frombuffer(b"".join(names), np.uint8).copy() if off[-1] else np.zeros(0, np.uint8))
    return multiset_fingerprint([col, np.array([counts[c] for c in names], np.int64)])


# ---- q11: user sessions ---------------------------------------------------------------------------------------------
def q11_user_sessions(bidder, b_date_time, epoch_row_offsets, timeout_s, base_time_ms):
    """The session launcher's walk, literally (flock-function/src/aws/window/session.rs), then q11.sql over what every
    epoch closes.  Returns one dict per epoch: {bidder: (bid_count, start_time, end_time)}.

    :242-250  the epoch's bids are split into one partition per distinct bidder (arrival order kept)
    :64-134   add_partitions_to_session_windows: a partition joins the bidder's open session unless the whole second of its
              FIRST bid is more than `timeout` after the whole second of the session's LAST bid -- then the old session is
              handed out and the partition starts a new one
    :144-178  find_timeout_session_windows: afterwards every open session whose last bid's whole second is more than
              `timeout` behind  BASE_TIME/1000 + epoch  is handed out
    :263-310  the sessions handed out in one epoch go to the query together (coalesce_windows): q11.sql groups by bidder,
              so two sessions of one bidder closed in the same epoch are one row
    Sessions still open after the last epoch are never handed out.  Timestamps are >= 0 (Rust's `/` truncates)."""
    bidder = np.asarray(bidder)
    ts = np.asarray(b_date_time, dtype=np.int64)
    off = np.asarray(epoch_row_offsets, dtype=np.int64)
    windows = {}  # bidder -> list of partitions (row-number arrays)
    out = []
    for t in range(len(off) - 1):
        lo, hi = int(off[t]), int(off[t + 1])
        b = bidder[lo:hi]
        order = np.argsort(b, kind="stable")
        sb = b[order]
        starts = np.flatnonzero(np.r_[True, sb[1:] != sb[:-1]]) if hi > lo else np.zeros(0, np.int64)
        ends = np.r_[starts[1:], len(sb)] if hi > lo else starts
        closed = []
        for s0, e0 in zip(starts.tolist(), ends.tolist()):
            rows = lo + order[s0:e0]
            key = int(sb[s0])
            if key in windows:
                last_s = int(ts[windows[key][-1][-1]]) // 1000
                if int(ts[rows[0]]) // 1000 - last_s > timeout_s:
                    closed.append(windows.pop(key))
            windows.setdefault(key, []).append(rows)
        clock = base_time_ms // 1000 + t
        for key in [k for k, parts in windows.items() if clock - int(ts[parts[-1][-1]]) // 1000 > timeout_s]:
            closed.append(windows.pop(key))
        res = {}
        for parts in closed:
            rows = np.concatenate(parts)
            key, c, mn, mx = int(bidder[rows[0]]), len(rows), int(ts[rows].min()), int(ts[rows].max())
            if key in res:
                c0, mn0, mx0 = res[key]
                c, mn, mx = c + c0, min(mn, mn0), max(mx, mx0)
            res[key] = (c, mn, mx)
        out.append(res)
    return out


def q11_user_sessions_columnar(bidder, b_date_time, epoch_row_offsets, timeout_s, base_time_ms):
    """The same result through whole-column numpy passes (the CPU baseline of bench.py; checked against the literal walk
    above in tests/test_oracle_q11.py): every decision of the walk only compares two neighbouring partitions of one bidder.
    Returns (epoch_out_offsets, bidder, bid_count, start_time, end_time), rows of an epoch ordered by bidder."""
    bidder = np.asarray(bidder)
    ts = np.asarray(b_date_time, dtype=np.int64)
    off = np.asarray(epoch_row_offsets, dtype=np.int64)
    n_epochs, lo, hi = len(off) - 1, int(off[0]), int(off[-1])
    n = hi - lo
    empty = (np.zeros(n_epochs + 1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.uint64), np.zeros(0, np.int64), np.zeros(0, np.int64))
    if n == 0:
        return empty
    order = np.argsort(bidder[lo:hi], kind="stable")
    k, t = bidder[lo:hi][order], ts[lo:hi][order]
    ep = np.searchsorted(off - lo, order, side="right") - 1
    sec = t // 1000
    clock = np.maximum(ep, sec - base_time_ms // 1000 + timeout_s + 1)  # first epoch in which the time-out check fires
    same = (k[:-1] == k[1:]) & ((ep[:-1] == ep[1:]) | ((clock[:-1] >= ep[1:]) & (sec[1:] - sec[:-1] <= timeout_s)))
    starts = np.r_[0, np.flatnonzero(~same) + 1]
    ends = np.r_[starts[1:], n]
    last = ends - 1
    nxt = np.minimum(ends, n - 1)
    follows = (ends < n) & (k[nxt] == k[last])
    close = np.where(follows, np.minimum(clock[last], ep[nxt]), np.where(clock[last] <= n_epochs - 1, clock[last], -1))
    cnt = (ends - starts).astype(np.uint64)
    mn, mx, who = np.minimum.reduceat(t, starts), np.maximum.reduceat(t, starts), k[starts]
    absorbed = np.r_[False, (who[1:] == who[:-1]) & (close[1:] == close[:-1]) & (close[1:] >= 0)]
    head = np.flatnonzero(~absorbed)
    cnt, mn, mx = np.add.reduceat(cnt, head), np.minimum.reduceat(mn, head), np.maximum.reduceat(mx, head)
    who, close = who[head], close[head]
    keep = close >= 0
    o = np.argsort(close[keep], kind="stable")
    out_off = np.searchsorted(close[keep][o], np.arange(n_epochs + 1), side="left").astype(np.int64)
    return out_off, who[keep][o].astype(np.int32), cnt[keep][o], mn[keep][o], mx[keep][o]


def q11_user_sessions_arrow(bidder, b_date_time, epoch_row_offsets, timeout_s, base_time_ms):
    """q11 a third way, with Arrow C++ doing the relational work (pyarrow compute / Acero): rows sorted by (bidder, arrival), a session
    id per row from a running sum of "a new session starts here", `group_by(session) -> COUNT / MIN / MAX` and a second
    `group_by(close epoch, bidder)` for q11.sql's own GROUP BY over what one epoch hands out.  The break and close rules are
    session.rs:64-178 as stated at q11_user_sessions; this is the engine the q11 goldens are minted against (with the literal walk).
    Returns the same tuple as q11_user_sessions_columnar."""
    import pyarrow as pa
    import pyarrow.compute as pc
    off = np.asarray(epoch_row_offsets, dtype=np.int64)
    n_epochs, lo, hi = len(off) - 1, int(off[0]), int(off[-1])
    n = hi - lo
    if n == 0:
        return (np.zeros(n_epochs + 1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.uint64), np.zeros(0, np.int64), np.zeros(0, np.int64))
    row = np.arange(n, dtype=np.int64)
    ep = (np.searchsorted(off - lo, row, side="right") - 1).astype(np.int64)
    t = pa.table({"bidder": np.asarray(bidder)[lo:hi], "ts": np.asarray(b_date_time, np.int64)[lo:hi], "ep": ep, "row": row})
    t = t.take(pc.sort_indices(t, sort_keys=[("bidder", "ascending"), ("row", "ascending")]))
    sec = pc.divide(t["ts"], 1000)                                       # (timestamps >= 0: truncation = floor)
    clock = pc.max_element_wise(t["ep"], pc.add(pc.subtract(sec, base_time_ms // 1000), timeout_s + 1))   # first epoch whose time-out check fires
    k, e = t["bidder"].combine_chunks(), t["ep"].combine_chunks()
    sec, clock = sec.combine_chunks(), clock.combine_chunks()
    prev = lambda a: a.slice(0, n - 1)
    nxt = lambda a: a.slice(1, n - 1)
    joins = pc.and_(pc.equal(prev(k), nxt(k)),
                    pc.or_(pc.equal(prev(e), nxt(e)),
                           pc.and_(pc.greater_equal(prev(clock), nxt(e)), pc.less_equal(pc.subtract(nxt(sec), prev(sec)), timeout_s))))
    starts = pa.concat_arrays([pa.array([1], pa.int64()), pc.cast(pc.invert(joins), pa.int64())])
    sid = pc.cumulative_sum(starts)
    t = t.append_column("sid", sid).append_column("clock", clock)
    g = t.group_by("sid", use_threads=False).aggregate([("bidder", "min"), ("ts", "min"), ("ts", "max"), ([], "count_all"), ("clock", "max"), ("ep", "max"),
                                                         ("row", "max")]).sort_by("sid")
    who = g["bidder_min"].to_numpy()
    m = len(who)
    # a session closes when its bidder's NEXT session starts (that epoch) or when the time-out check fires, whichever is first;
    # the clock of a session = the clock of its LAST row (rows of a session are time-ordered within the walk: take the last by arrival)
    last_row = g["row_max"].to_numpy()
    order_rows = t["row"].to_numpy()
    pos_of = np.empty(n, np.int64)
    pos_of[order_rows] = np.arange(n)
    last_clock = clock.to_numpy()[pos_of[last_row]]
    first_ep_next = np.r_[t["ep"].to_numpy()[np.flatnonzero(starts.to_numpy())][1:], -1]
    follows = np.r_[who[1:] == who[:-1], False]
    close = np.where(follows, np.minimum(last_clock, first_ep_next), np.where(last_clock <= n_epochs - 1, last_clock, -1))
    s = pa.table({"close": close, "bidder": who, "cnt": g["count_all"].to_numpy(), "mn": g["ts_min"].to_numpy(), "mx": g["ts_max"].to_numpy()})
    s = s.filter(pc.greater_equal(s["close"], 0))
    r = s.group_by(["close", "bidder"], use_threads=False).aggregate([("cnt", "sum"), ("mn", "min"), ("mx", "max")]).sort_by([("close", "ascending"), ("bidder", "ascending")])
    c = r["close"].to_numpy()
    out_off = np.searchsorted(c, np.arange(n_epochs + 1), side="left").astype(np.int64)
    return (out_off, r["bidder"].to_numpy().astype(np.int32), r["cnt_sum"].to_numpy().astype(np.uint64), r["mn_min"].to_numpy().astype(np.int64),
            r["mx_max"].to_numpy().astype(np.int64))


def q11_fingerprints(result) -> list:
    """One fingerprint per epoch of a (epoch_out_offsets, bidder, bid_count, start_time, end_time) result."""
    out_off, who, cnt, mn, mx = result
    fps = []
    for t in range(len(out_off) - 1):
        sl = slice(int(out_off[t]), int(out_off[t + 1]))
        fps.append(multiset_fingerprint([np.asarray(who[sl], np.int32), np.asarray(cnt[sl]).astype(np.int64), np.asarray(mn[sl], np.int64), np.asarray(mx[sl], np.int64)]))
    return fps


# ---- JSON lines <-> columns (the reference's event buffers and `event_bytes_to_batch`) ------------------------------------
def nexmark_json_lines(relation: str, cols: dict) -> bytes:
    """The epoch buffer the reference's generator builds (flock/src/datasource/nexmark/generator.rs:79-93):
    serde_json::to_vec of every event + b"\n"; struct field order of event.rs:103-117 / 189-207 / 315-323, compact
    separators, integers as decimal literals, Epoch newtype as its integer."""
    import json
    order = {"bid": ["auction", "bidder", "price", "b_date_time"],
             "auction": ["a_id", "item_name", "description", "initial_bid", "reserve", "a_date_time", "expires", "seller", "category"],
             "person": ["p_id", "name", "email_address", "credit_card", "city", "state", "p_date_time"]}[relation]
    n = len(cols[order[0]])
    strs = {}
    for k in order:
        if isinstance(cols[k], Utf8):
            b, off = cols[k].data.tobytes(), cols[k].offsets
            strs[k] = [b[off[i]:off[i + 1]].decode() for i in range(n)]
    out = []
    for i in range(n):
        obj = {k: (strs[k][i] if k in strs else int(cols[k][i])) for k in order}
        out.append(json.dumps(obj, separators=(",", ":"), ensure_ascii=False).encode())
    return b"".join(line + b"\n" for line in out)


def json_lines_decode(data: bytes, fields):
    """`event_bytes_to_batch` (flock/src/transmute.rs:255-266: arrow json::Reader over the buffer, one object per line):
    fields = [(name, "int32" | "int64" | "utf8")] -> {name: np.ndarray | Utf8}.  Python's json module is the decoder."""
    import json
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    objs = [json.loads(line) for line in lines]
    out = {}
    for name, t in fields:
        if t == "utf8":
            vals = [o[name].encode() for o in objs]
            off = np.zeros(len(vals) + 1, np.int32)
            np.cumsum([len(v) for v in vals], out=off[1:])
            out[name] = Utf8(off, np.frombuffer(b"".join(vals), np.uint8).copy())
        else:
            for o in objs:
                if not isinstance(o[name], int) or isinstance(o[name], bool):
                    raise ValueError(f"{name}: integer literal expected")
            out[name] = np.array([o[name] for o in objs], np.int32 if t == "int32" else np.int64)
    return out
