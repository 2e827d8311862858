"""TEST MODEL of the in-library exchange (flock_amd/csrc/comm.hip) on `torch.distributed` collectives -- not product code: no product
path imports it (it lived in the package until round 5).  It states the exchange protocol in Python so that the protocol's logic --
split sizes, regrouping, window bookkeeping -- runs at world size 2 on `gloo` here (tests/test_distributed.py), and so that the HIP
partition / take kernels can be checked bit for bit against its numpy stand-ins (tests/test_gpu_exchange.py).

Multi-GPU execution of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference scales the path in two ways (SURVEY.md section 8 e), both mirrored here:

1. **Window sharding** -- whole windows are routed to function-group members by consistent hash of the window id
   (flock-function/src/aws/window/tumbling.rs:144-151, hopping.rs:82-88).  Rank r simply owns its own slice of the
   event stream (`NEXMarkSource(first_event_id=...)`); there is no data-path collective.  This is what `bench.py
   --gpus N` measures ("scaling": "weak").

2. **Key-partitioned exchange** -- the distributed plans are `filter / partial-agg -> RepartitionExec Hash([key], n)
   -> join / final-agg` (flock/src/distributed_plan/planner.rs:152-171, playground/.../nexmark/q{3,5,8}.dag), moved
   between Lambdas as an all-to-all of Arrow payloads (flock-function/src/aws/actor.rs:425-543).  Here every window
   is striped across the ranks; `shuffle_relation` routes every row to rank `part(key)`:
       partition (HIP: count -> scan -> emit)  ->  take into send order (HIP)  ->  counts all-to-all
       ->  one `all_to_all_single` per column buffer  ->  take into (window, source) order (HIP)
   after which the unchanged single-GPU operators (`q3_join`, `q8_join`, `q5_hot_items`) run on the received rows;
   the result stays sharded by key.  Messages: one buffer per column per step (xGMI is point-to-point, 7 links per
   GPU: few large transfers, never one per window).  q5 shuffles PANES (each bid moves once although it belongs to
   two hopping windows), aggregates the keys it owns, and needs one more collective: `all_reduce(MAX)` of the
   per-window maxima (8 bytes per window) before the `num = maxn` filter.  As in q5.dag the repartition is preceded by
   `HashAggregateExec: mode=Partial`: what crosses the fabric are the (auction, count) groups of every pane, not the bids.

The local device operations are injected (`LocalOps`) so that the exchange logic itself -- split sizes, regrouping,
window bookkeeping -- is covered by world-size-2 `gloo` tests on CPU tensors (tests/test_distributed.py), with the
test supplying numpy stand-ins for the HIP kernels.  The product path (`GpuOps`) has no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple, Union

import numpy as np

from flock_amd.engine import Auctions, DeviceUtf8, GpuContext, Persons, WindowSchedule


def _torch():
    import torch
    return torch


def _dist():
    import torch.distributed as dist
    return dist


Column = Union["object", DeviceUtf8]


class GpuOps:
    """The HIP implementations of the local steps of a shuffle (include/flockgpu.h "key-partitioned exchange")."""

    def __init__(self, ctx: GpuContext):
        self.ctx = ctx
        self._slot = 0

    def partition(self, keys, schedule: WindowSchedule, n_parts: int):
        return self.ctx.partition_by_key(keys, schedule, n_parts)

    def take(self, col: Column, rows):
        if isinstance(col, DeviceUtf8):
            self._slot = (self._slot + 1) % 16
            return self.ctx.take_utf8(col, rows, slot=self._slot)
        return self.ctx.take(col, rows)

    def offsets_from_lengths(self, lengths):
        return self.ctx.offsets_from_lengths(lengths)

    def q5_partial(self, auction, schedule: WindowSchedule):
        """(auction, count, pane_out_offsets): the groups of every pane (q5.dag HashAggregateExec mode=Partial)."""
        from flock_amd.engine import Bids
        return self.ctx.q5_partial_counts(Bids(auction=auction, rows=int(auction.numel())), schedule)

    def q5_weighted(self, auction, count, schedule: WindowSchedule):
        """(auction, num, offsets, win_max) host arrays: FinalPartitioned + MAX + join over (auction, count) rows."""
        r = self.ctx.q5_hot_items_weighted(auction, count, schedule)
        a, n, off = r.to_host()
        return a, n, off, r.win_max()


_MAX_PEER_BYTES = 1 << 30   # RCCL / c10d transfers above 2 GiB per peer arrive corrupted: stay well below


def _all_to_all(send, send_splits, recv_splits, group):
    """One variable-size all-to-all of a 1-D buffer (RCCL `ncclSend/ncclRecv` group under the hood), cut into
    rounds of at most 1 GiB per peer.  Round k moves elements [k*C, (k+1)*C) of every (source, destination) run; C is
    a constant, so both ends derive the same per-round splits from their own counts, and the number of rounds is
    agreed with one all_reduce(MAX)."""
    torch = _torch()
    dist = _dist()
    send_splits = np.asarray(send_splits, np.int64)
    recv_splits = np.asarray(recv_splits, np.int64)
    out = torch.empty(int(recv_splits.sum()), dtype=send.dtype, device=send.device)
    send = send.contiguous()
    chunk = max(1, _MAX_PEER_BYTES // send.element_size())
    biggest = torch.tensor([max(int(send_splits.max(initial=0)), int(recv_splits.max(initial=0)))], dtype=torch.int64,
                           device=send.device)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(biggest.item()) // chunk))
    if rounds == 1:
        dist.all_to_all_single(out, send, [int(x) for x in recv_splits], [int(x) for x in send_splits], group=group)
        return out
    s_off = np.concatenate(([0], np.cumsum(send_splits)))
    r_off = np.concatenate(([0], np.cumsum(recv_splits)))
    for k in range(rounds):
        s_len = np.clip(send_splits - k * chunk, 0, chunk)
        r_len = np.clip(recv_splits - k * chunk, 0, chunk)
        s_buf = torch.cat([send[int(s_off[d] + k * chunk): int(s_off[d] + k * chunk + s_len[d])] for d in range(len(s_len))])
        r_buf = torch.empty(int(r_len.sum()), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(r_buf, s_buf, [int(x) for x in r_len], [int(x) for x in s_len], group=group)
        pos = 0
        for src in range(len(r_len)):
            out[int(r_off[src] + k * chunk): int(r_off[src] + k * chunk + r_len[src])] = r_buf[pos: pos + int(r_len[src])]
            pos += int(r_len[src])
    return out


def exchange_counts(counts: np.ndarray, device, group=None) -> np.ndarray:
    """counts[d][w] = rows this rank sends to rank d for window w  ->  recv[s][w] = rows rank s sends to this rank."""
    torch = _torch()
    g, w = counts.shape
    send = torch.from_numpy(np.ascontiguousarray(counts, np.int64)).to(device).reshape(-1)
    out = torch.empty_like(send)
    _dist().all_to_all_single(out, send, [w] * g, [w] * g, group=group)
    return out.cpu().numpy().reshape(g, w)


def regroup_index(recv_counts: np.ndarray, device):
    """The received buffer is source-major ([source][window] runs); the operators want each window contiguous.
    Returns (index, window_row_offsets): out[i] = recv[index[i]] is window-major with sources in rank order."""
    torch = _torch()
    g, w = recv_counts.shape
    src_off = np.zeros(g * w + 1, np.int64)
    np.cumsum(recv_counts.reshape(-1), out=src_off[1:])
    starts = src_off[:-1].reshape(g, w).T.reshape(-1)          # run start, in (window, source) order
    lens = recv_counts.T.reshape(-1)
    out_off = np.zeros(g * w + 1, np.int64)
    np.cumsum(lens, out=out_off[1:])
    total = int(out_off[-1])
    shift = torch.from_numpy(starts - out_off[:-1]).to(device)
    idx = torch.arange(total, dtype=torch.int64, device=device) + torch.repeat_interleave(
        shift, torch.from_numpy(lens).to(device), output_size=total)
    win_off = out_off[::g].copy()                              # window w starts at out_off[w * g]
    return idx.to(torch.int32), win_off


def shuffle_relation(ops, columns: Dict[str, Column], key: str, schedule: WindowSchedule, group=None
                     ) -> Tuple[Dict[str, Column], WindowSchedule]:
    """RepartitionExec Hash([key], world) for one relation: every row of every window moves to rank part(key).
    Returns the received columns (window-major, sources in rank order inside a window) and their schedule."""
    torch = _torch()
    dist = _dist()
    world = dist.get_world_size(group)
    device = columns[key].device
    rows, counts = ops.partition(columns[key], schedule, world)           # counts[world][n_windows]
    recv_counts = exchange_counts(counts, device, group)
    send_splits, recv_splits = counts.sum(axis=1), recv_counts.sum(axis=1)
    index, win_off = regroup_index(recv_counts, device)
    send_off = np.concatenate(([0], np.cumsum(send_splits)))
    out: Dict[str, Column] = {}
    for name, col in columns.items():
        sent = ops.take(col, rows)
        if isinstance(col, DeviceUtf8):
            lengths = sent.offsets[1:] - sent.offsets[:-1]
            bounds = sent.offsets[torch.from_numpy(send_off).to(device)].cpu().numpy().astype(np.int64)
            byte_send = np.diff(bounds)
            byte_recv = exchange_counts(byte_send.reshape(world, 1), device, group).reshape(-1)
            r_len = _all_to_all(lengths, send_splits, recv_splits, group)
            r_bytes = _all_to_all(sent.data[: int(bounds[-1])], byte_send, byte_recv, group)
            if r_bytes.numel() < 16:
                r_bytes = torch.cat([r_bytes, torch.zeros(16 - r_bytes.numel(), dtype=torch.uint8, device=device)])
            received = DeviceUtf8(ops.offsets_from_lengths(r_len), r_bytes)
        else:
            received = _all_to_all(sent, send_splits, recv_splits, group)
        out[name] = ops.take(received, index)
    n_win = schedule.n_windows
    return out, WindowSchedule(win_off, np.arange(n_win), np.arange(1, n_win + 1))


# ------------------------------------------------------------------ the exchange-mode queries
def q8_exchange(ctx: GpuContext, persons: Persons, person_windows: WindowSchedule, auctions: Auctions,
                auction_windows: WindowSchedule, group=None, ops=None):
    """q8 with the join shuffle of q8.dag: both sides hash-partitioned on the join key (p_id / seller), then the
    local DISTINCT + join.  This rank returns the result rows whose p_id it owns."""
    ops = ops or GpuOps(ctx)
    p_cols, p_sched = shuffle_relation(ops, {"p_id": persons.p_id, "name": persons.name}, "p_id", person_windows, group)
    a_cols, a_sched = shuffle_relation(ops, {"seller": auctions.seller}, "seller", auction_windows, group)
    p = Persons(p_id=p_cols["p_id"], name=p_cols["name"], rows=int(p_cols["p_id"].numel()))
    a = Auctions(seller=a_cols["seller"], rows=int(a_cols["seller"].numel()))
    return ctx.q8_join(p, p_sched, a, a_sched)


def q3_exchange(ctx: GpuContext, auctions: Auctions, auction_windows: WindowSchedule, persons: Persons,
                person_windows: WindowSchedule, group=None, ops=None, category: int = 10, states=("or", "id", "ca")):
    """q3 with the join shuffle of planner.rs:152-171: auctions partitioned on seller, persons on p_id."""
    ops = ops or GpuOps(ctx)
    a_cols, a_sched = shuffle_relation(
        ops, {"a_id": auctions.a_id, "seller": auctions.seller, "category": auctions.category}, "seller", auction_windows, group)
    p_cols, p_sched = shuffle_relation(
        ops, {"p_id": persons.p_id, "name": persons.name, "city": persons.city, "state": persons.state}, "p_id",
        person_windows, group)
    a = Auctions(a_id=a_cols["a_id"], seller=a_cols["seller"], category=a_cols["category"], rows=int(a_cols["a_id"].numel()))
    p = Persons(p_id=p_cols["p_id"], name=p_cols["name"], city=p_cols["city"], state=p_cols["state"],
                rows=int(p_cols["p_id"].numel()))
    return ctx.q3_join(a, a_sched, p, p_sched, category=category, states=states)


@dataclass
class Q5Shard:
    """This rank's share of the q5 answer in exchange mode: winners among the auctions it owns (host arrays)."""
    auction: np.ndarray   # int32
    num: np.ndarray       # uint64
    offsets: np.ndarray   # int64, n_windows + 1
    win_max: np.ndarray   # uint64, n_windows: the GLOBAL maximum per window


def global_window_max(local_max: np.ndarray, device, group=None) -> np.ndarray:
    """all_reduce(MAX) of the per-window maxima (MAX over the FinalPartitioned partitions, q5.dag)."""
    torch = _torch()
    t = torch.from_numpy(np.ascontiguousarray(local_max).astype(np.int64)).to(device)
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX, group=group)
    return t.cpu().numpy().astype(np.uint64)


def keep_global_winners(auction, num, offsets, local_max, gmax):
    """Rows of windows whose local maximum is the global one survive the `num = maxn` join."""
    keep_win = (local_max == gmax) & (gmax > 0)
    per_win = np.diff(offsets)
    keep_row = np.repeat(keep_win, per_win)
    new_off = np.concatenate(([0], np.cumsum(np.where(keep_win, per_win, 0))))
    return auction[keep_row], num[keep_row], new_off


def q5_exchange(ctx: GpuContext, bids, windows: WindowSchedule, group=None, ops=None) -> Q5Shard:
    """q5 as q5.dag runs it: HashAggregateExec(Partial) on this rank's rows, pane by pane -> the (auction, count) groups
    hash-partitioned on `auction` (each group moves once although its pane belongs to two hopping windows; a bid's
    4 bytes never cross the fabric, only its group's 8) -> FinalPartitioned COUNT / MAX / filter over the owned keys ->
    all_reduce(MAX) across the partitions."""
    ops = ops or GpuOps(ctx)
    keys, cnts, pane_off = ops.q5_partial(bids.auction, windows)
    n_panes = len(pane_off) - 1
    pane_sched = WindowSchedule(pane_off, np.arange(n_panes), np.arange(1, n_panes + 1))
    cols, recv = shuffle_relation(ops, {"auction": keys, "count": cnts}, "auction", pane_sched, group)
    recv_sched = WindowSchedule(recv.pane_row_offsets, windows.win_pane_lo, windows.win_pane_hi)
    a, n, off, local_max = ops.q5_weighted(cols["auction"], cols["count"], recv_sched)
    gmax = global_window_max(local_max, cols["auction"].device, group)
    a, n, off = keep_global_winners(a, n, off, local_max, gmax)
    return Q5Shard(a, n, off, gmax)


@dataclass
class Q5Merged:
    """Global q5 answer assembled from per-rank partial results (host arrays, identical on every rank)."""
    auction: np.ndarray   # int32
    num: np.ndarray       # uint64
    offsets: np.ndarray   # int64, n_windows + 1


def q5_merge_window_winners(local_auction: np.ndarray, local_num: np.ndarray, local_offsets: np.ndarray, group=None,
                            device="cpu") -> Q5Merged:
    """Final step of a WINDOW-sharded q5 when one rank must hold the whole answer: every rank owns complete windows
    (mode 1), so the merge is an all-gather of the few winner rows per window, concatenated in rank order."""
    torch = _torch()
    dist = _dist()
    world = dist.get_world_size(group)
    n = torch.tensor([len(local_auction), len(local_offsets) - 1], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [s.cpu().numpy() for s in sizes]
    rows_max, win_max = max(int(s[0]) for s in sizes), max(int(s[1]) for s in sizes)

    def gather(a, dtype, cap):
        buf = torch.zeros(cap, dtype=dtype, device=device)
        buf[: len(a)] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        outs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf, group=group)
        return [o.cpu().numpy() for o in outs]

    ga = gather(local_auction.astype(np.int32), torch.int32, rows_max)
    gn = gather(local_num.astype(np.int64), torch.int64, rows_max)
    go = gather(np.diff(local_offsets).astype(np.int64), torch.int64, win_max)
    auction = np.concatenate([ga[r][: int(sizes[r][0])] for r in range(world)])
    num = np.concatenate([gn[r][: int(sizes[r][0])] for r in range(world)]).astype(np.uint64)
    per_win = np.concatenate([go[r][: int(sizes[r][1])] for r in range(world)])
    return Q5Merged(auction, num, np.concatenate(([0], np.cumsum(per_win))))
