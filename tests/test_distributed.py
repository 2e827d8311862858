"""N > 1 path on CPU: world-size-2 `gloo` runs of the key-partitioned exchange (tests/exchange_model.py).

The exchange logic (counts all-to-all, per-column all_to_all_single with uneven splits, regrouping into
window-major order, window bookkeeping, q5's all_reduce(MAX)) is the product code; the three device steps it calls
(`partition`, `take`, `offsets_from_lengths`) are HIP kernels on a GPU and are replaced here by the numpy stand-ins
below, which restate the same contract (include/flockgpu.h "key-partitioned exchange").  The per-window operators
on the received rows are played by the CPU oracle.  The GPU twins of these tests (HIP partition / take against these
same stand-ins, and the exchange end to end on one rank) are in tests/test_gpu_exchange.py.
"""
import os
import socket
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from flock_amd.engine import DeviceUtf8, WindowSchedule


# ---------------------------------------------------------------- numpy stand-ins for the HIP steps
def mix32(x):
    x = np.asarray(x).astype(np.uint32).astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def part_of(keys, n_parts):
    return ((mix32(keys) * np.uint64(n_parts)) >> np.uint64(32)).astype(np.int64)


class NumpyOps:
    """Same contract as exchange_model.GpuOps, on CPU tensors."""

    def partition(self, keys, schedule: WindowSchedule, n_parts: int):
        k = keys.numpy()
        rows, counts = [], np.zeros((n_parts, schedule.n_windows), np.int64)
        for p in range(n_parts):
            for w in range(schedule.n_windows):
                lo, hi = schedule.window_rows(w)
                r = np.arange(lo, hi)[part_of(k[lo:hi], n_parts) == p]
                counts[p, w] = len(r)
                rows.append(r)
        return torch.from_numpy(np.concatenate(rows).astype(np.int32) if rows else np.zeros(0, np.int32)), counts

    def take(self, col, rows):
        idx = rows.to(torch.int64)
        if isinstance(col, DeviceUtf8):
            t = oracle.take_utf8(oracle.Utf8(col.offsets.numpy(), col.data.numpy()), idx.numpy())
            data = t.data if len(t.data) >= 16 else np.concatenate([t.data, np.zeros(16 - len(t.data), np.uint8)])
            return DeviceUtf8(torch.from_numpy(t.offsets.copy()), torch.from_numpy(data.copy()))
        return col[idx]

    def offsets_from_lengths(self, lengths):
        off = torch.zeros(lengths.numel() + 1, dtype=torch.int32)
        off[1:] = torch.cumsum(lengths, 0)
        return off

    def q5_partial(self, auction, schedule: WindowSchedule):
        a = auction.numpy()
        po = schedule.pane_row_offsets
        keys, cnts, off = [], [], [0]
        for p in range(len(po) - 1):
            k, c = np.unique(a[po[p]:po[p + 1]], return_counts=True)
            keys.append(k.astype(np.int32))
            cnts.append(c.astype(np.int32))
            off.append(off[-1] + len(k))
        cat = lambda xs: torch.from_numpy(np.concatenate(xs) if xs else np.zeros(0, np.int32))
        return cat(keys), cat(cnts), np.array(off, np.int64)

    def q5_weighted(self, auction, count, schedule: WindowSchedule):
        a, c = auction.numpy(), count.numpy().astype(np.uint32)
        oa, on, off, mx = [], [], [0], []
        for w in range(schedule.n_windows):
            lo, hi = schedule.window_rows(w)
            k, inv = np.unique(a[lo:hi], return_inverse=True)
            tot = np.zeros(len(k), np.uint64)
            np.add.at(tot, inv, c[lo:hi].astype(np.uint64))
            m = tot.max() if len(tot) else 0
            sel = tot == m if m else np.zeros(len(k), bool)
            oa.append(k[sel].astype(np.int32))
            on.append(tot[sel])
            off.append(off[-1] + int(sel.sum()))
            mx.append(int(m))
        return (np.concatenate(oa) if oa else np.zeros(0, np.int32), np.concatenate(on) if on else np.zeros(0, np.uint64),
                np.array(off, np.int64), np.array(mx, np.uint64))


# ---------------------------------------------------------------- harness
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn, ret):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        fn(rank, world)
        dist.barrier()
        dist.destroy_process_group()
        ret[rank] = "ok"
    except Exception:
        ret[rank] = traceback.format_exc()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert not p.is_alive(), "distributed worker hung"
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"


def _stripe(lo, hi, rank, world):
    """Rows [lo, hi) of a window striped into `world` contiguous pieces; this rank's piece."""
    n = hi - lo
    return lo + n * rank // world, lo + n * (rank + 1) // world


def _utf8_t(u):
    data = u.data if len(u.data) >= 16 else np.concatenate([u.data, np.zeros(16 - len(u.data), np.uint8)])
    return DeviceUtf8(torch.from_numpy(np.ascontiguousarray(u.offsets)), torch.from_numpy(np.ascontiguousarray(data)))


def _strs(off, data, rows=None):
    b = bytes(np.asarray(data).tobytes())
    idx = range(len(off) - 1) if rows is None else rows
    return [b[off[i]:off[i + 1]] for i in idx]


SEED, EPS, SECONDS = 11, 4000, 20


def _local_rows(offsets_fn, windows, rank, world):
    """Concatenated stripes of every window + the local schedule (one pane per window)."""
    pieces, off = [], [0]
    for (e0, e1) in windows:
        lo, hi = offsets_fn(e0), offsets_fn(e1)
        a, b = _stripe(lo, hi, rank, world)
        pieces.append(np.arange(a, b))
        off.append(off[-1] + (b - a))
    rows = np.concatenate(pieces) if pieces else np.zeros(0, np.int64)
    n = len(windows)
    return rows, WindowSchedule(np.array(off), np.arange(n), np.arange(1, n + 1))


# ---------------------------------------------------------------- tests
def _q8_rank(rank, world):
    from exchange_model import shuffle_relation
    s = oracle.NexmarkStream(seed=SEED, eps=EPS)
    n = EPS * SECONDS
    au, pe = s.auctions(0, n), s.persons(0, n)
    wins = oracle.tumbling_windows(SECONDS, 10)
    a_off = lambda e: s.counts(0, e * EPS)[1]
    p_off = lambda e: s.counts(0, e * EPS)[0]
    a_rows, a_sched = _local_rows(a_off, wins, rank, world)
    p_rows, p_sched = _local_rows(p_off, wins, rank, world)
    ops = NumpyOps()
    a_cols, ra = shuffle_relation(ops, {"seller": torch.from_numpy(au["seller"][a_rows])}, "seller", a_sched)
    p_cols, rp = shuffle_relation(
        ops, {"p_id": torch.from_numpy(pe["p_id"][p_rows]), "name": _utf8_t(oracle.take_utf8(pe["name"], p_rows))}, "p_id", p_sched)
    # every received key belongs to this rank
    assert (part_of(a_cols["seller"].numpy(), world) == rank).all() and (part_of(p_cols["p_id"].numpy(), world) == rank).all()
    mine = []
    name = oracle.Utf8(p_cols["name"].offsets.numpy(), p_cols["name"].data.numpy())
    for w in range(len(wins)):
        (alo, ahi), (plo, phi) = ra.window_rows(w), rp.window_rows(w)
        pid = p_cols["p_id"].numpy()[plo:phi]
        out = oracle.q8_join(pid, name.slice(plo, phi), a_cols["seller"].numpy()[alo:ahi])
        nm = _strs(name.offsets, name.data, rows=[plo + r for r in out])
        mine.append(sorted(zip(pid[out].tolist(), nm)))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    for w, (e0, e1) in enumerate(wins):
        (alo, ahi), (plo, phi) = (a_off(e0), a_off(e1)), (p_off(e0), p_off(e1))
        rows = oracle.q8_join(pe["p_id"][plo:phi], pe["name"].slice(plo, phi), au["seller"][alo:ahi])
        nm = _strs(pe["name"].offsets, pe["name"].data, rows=[plo + r for r in rows])
        want = sorted(zip(pe["p_id"][plo:phi][rows].tolist(), nm))
        got = sorted(sum((gathered[r][w] for r in range(world)), []))
        assert got == want and len(want) > 0, f"q8 window {w}"


def test_q8_join_shuffle_world2():
    _run(_q8_rank)


def _q3_rank(rank, world):
    from exchange_model import shuffle_relation
    s = oracle.NexmarkStream(seed=SEED + 1, eps=EPS)
    n = EPS * 6
    au, pe = s.auctions(0, n), s.persons(0, n)
    wins = oracle.elementwise_windows(6)
    a_off = lambda e: s.counts(0, e * EPS)[1]
    p_off = lambda e: s.counts(0, e * EPS)[0]
    a_rows, a_sched = _local_rows(a_off, wins, rank, world)
    p_rows, p_sched = _local_rows(p_off, wins, rank, world)
    ops = NumpyOps()
    a_cols, ra = shuffle_relation(ops, {k: torch.from_numpy(au[k][a_rows]) for k in ("a_id", "seller", "category")}, "seller", a_sched)
    p_in = {"p_id": torch.from_numpy(pe["p_id"][p_rows])}
    for k in ("name", "city", "state"):
        p_in[k] = _utf8_t(oracle.take_utf8(pe[k], p_rows))
    p_cols, rp = shuffle_relation(ops, p_in, "p_id", p_sched)
    u = {k: oracle.Utf8(p_cols[k].offsets.numpy(), p_cols[k].data.numpy()) for k in ("name", "city", "state")}
    mine = []
    for w in range(len(wins)):
        (alo, ahi), (plo, phi) = ra.window_rows(w), rp.window_rows(w)
        ar, pr = oracle.q3_join(a_cols["seller"].numpy()[alo:ahi], a_cols["category"].numpy()[alo:ahi],
                                p_cols["p_id"].numpy()[plo:phi], u["state"].slice(plo, phi))
        rows = [plo + int(r) for r in pr]
        mine.append(sorted(zip(_strs(u["name"].offsets, u["name"].data, rows), _strs(u["city"].offsets, u["city"].data, rows),
                               _strs(u["state"].offsets, u["state"].data, rows), a_cols["a_id"].numpy()[alo:ahi][ar].tolist())))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    total = 0
    for w, (e0, e1) in enumerate(wins):
        (alo, ahi), (plo, phi) = (a_off(e0), a_off(e1)), (p_off(e0), p_off(e1))
        ar, pr = oracle.q3_join(au["seller"][alo:ahi], au["category"][alo:ahi], pe["p_id"][plo:phi], pe["state"].slice(plo, phi))
        rows = [plo + int(r) for r in pr]
        want = sorted(zip(_strs(pe["name"].offsets, pe["name"].data, rows), _strs(pe["city"].offsets, pe["city"].data, rows),
                          _strs(pe["state"].offsets, pe["state"].data, rows), au["a_id"][alo:ahi][ar].tolist()))
        got = sorted(sum((gathered[r][w] for r in range(world)), []))
        assert got == want, f"q3 window {w}"
        total += len(want)
    assert total > 0


def test_q3_join_shuffle_world2():
    _run(_q3_rank)


def _q5_rank(rank, world):
    """q5_exchange itself (partial groups -> hash repartition -> weighted final -> all_reduce(MAX)) with numpy stand-ins
    for the device steps."""
    from flock_amd import Bids
    from exchange_model import q5_exchange
    s = oracle.NexmarkStream(seed=SEED + 2, eps=EPS)
    b = s.bids(0, EPS * SECONDS, columns=("auction",))["auction"]
    b_off = lambda e: s.counts(0, e * EPS)[2]
    wins = oracle.hopping_windows(SECONDS, 10, 5)
    panes = [(e, e + 5) for e in range(0, SECONDS, 5)]
    rows, pane_sched = _local_rows(b_off, panes, rank, world)
    sched = WindowSchedule(pane_sched.pane_row_offsets, np.array([e0 // 5 for e0, _ in wins], np.int32),
                           np.array([e1 // 5 for _, e1 in wins], np.int32))
    local = torch.from_numpy(b[rows])
    shard = q5_exchange(None, Bids(auction=local, rows=len(rows)), sched, ops=NumpyOps())
    gathered = [None] * world
    dist.all_gather_object(gathered, [sorted(zip(shard.auction[shard.offsets[w]:shard.offsets[w + 1]].tolist(),
                                                 shard.num[shard.offsets[w]:shard.offsets[w + 1]].tolist()))
                                      for w in range(len(wins))])
    for w, (e0, e1) in enumerate(wins):
        oa, on = oracle.q5_hot_items(b[b_off(e0):b_off(e1)])
        got = sorted(sum((gathered[r][w] for r in range(world)), []))
        assert got == sorted(zip(oa.tolist(), on.tolist())), f"q5 window {w}"
        assert int(shard.win_max[w]) == int(on[0])


def test_q5_repartition_and_global_max_world2():
    _run(_q5_rank)


def _merge_rank(rank, world):
    from exchange_model import q5_merge_window_winners
    # window-sharded mode: rank r owns windows [3r, 3r + 3) with r + 1 winner rows each
    a = np.arange(3 * (rank + 1), dtype=np.int32) + 100 * rank
    n = (np.arange(3 * (rank + 1)) + 7).astype(np.uint64)
    off = np.arange(4) * (rank + 1)
    m = q5_merge_window_winners(a, n, off)
    assert m.offsets.tolist() == [0, 1, 2, 3, 5, 7, 9]
    assert m.auction.tolist() == list(range(3)) + [100 + i for i in range(6)]
    assert m.num.tolist() == [7, 8, 9] + [7 + i for i in range(6)]


def test_window_sharded_merge_world2():
    _run(_merge_rank)


def test_regroup_index_orders_by_window_then_source():
    from exchange_model import regroup_index
    recv = np.array([[2, 0, 1], [1, 3, 0]])                     # [source][window]
    idx, win_off = regroup_index(recv, "cpu")
    # received buffer: s0w0 s0w0 s0w2 | s1w0 s1w1 s1w1 s1w1
    assert idx.tolist() == [0, 1, 3, 4, 5, 6, 2] and win_off.tolist() == [0, 3, 6, 7]


def _chunked_rank(rank, world):
    import exchange_model as D
    D._MAX_PEER_BYTES = 64            # 16 int32 per peer and round: forces several rounds with ragged tails
    rng = np.random.default_rng(rank)
    send_splits = np.array([37, 5]) if rank == 0 else np.array([0, 50])
    recv_splits = np.array([37, 0]) if rank == 0 else np.array([5, 50])
    send = torch.from_numpy(rng.integers(0, 1000, int(send_splits.sum())).astype(np.int32))
    got = D._all_to_all(send, send_splits, recv_splits, None)
    everything = [None] * world
    dist.all_gather_object(everything, (send.tolist(), send_splits.tolist()))
    want = []
    for src in range(world):
        data, splits = everything[src]
        off = int(np.sum(splits[:rank]))
        want += data[off: off + splits[rank]]
    assert got.tolist() == want


def test_all_to_all_is_cut_into_rounds_world2():
    _run(_chunked_rank)
