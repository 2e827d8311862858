"""Host-side driver of libflockgpu: the GPU twin of the reference's per-window operator execution.

One :class:`GpuContext` = one HIP stream + device arena (the unit `actor::collect`
would own, flock-function/src/aws/actor.rs:54-79).  Column data lives in HBM as torch
tensors (torch is only the allocator / stream provider here); every operator call goes
through the C ABI of include/flockgpu.h.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import FlockGpuError


def _torch():
    import torch
    return torch


def _cached_ffi(obj, fields, make):
    """The ctypes view of a column container, rebuilt only when one of its columns (or the row count) was replaced: building it costs a
    `data_ptr()` per column and a ctypes cast per host array -- tens of microseconds per call, as much as the kernels of a small batch
    (q3 at 1e8 events: ~60 us), and a host in the reference's own language has none of it."""
    def ident(v):   # device columns by address (an id() can be reused by a new tensor), host arrays by identity
        if v is None:
            return 0
        if isinstance(v, DeviceUtf8):
            return (v.offsets.data_ptr(), v.data.data_ptr())
        return v.data_ptr() if hasattr(v, "data_ptr") else id(v)
    key = tuple(ident(getattr(obj, f)) for f in fields) + (getattr(obj, "rows", None),)
    hit = obj.__dict__.get("_ffi_cache")
    if hit is None or hit[0] != key:
        hit = (key, make())
        obj.__dict__["_ffi_cache"] = hit
    return hit[1]


# ------------------------------------------------------------------ device column containers
@dataclass
class DeviceUtf8:
    """Arrow Utf8 column in HBM: int32 offsets (rows + 1) + uint8 bytes."""
    offsets: "object"
    data: "object"

    def ffi(self) -> _ffi.Utf8:
        return _ffi.Utf8(self.offsets.data_ptr(), self.data.data_ptr())


@dataclass
class Bids:
    auction: "object" = None
    bidder: "object" = None
    price: "object" = None
    b_date_time: "object" = None
    rows: int = 0

    def ffi(self) -> _ffi.BidCols:
        p = lambda t: None if t is None else t.data_ptr()
        return _cached_ffi(self, ("auction", "bidder", "price", "b_date_time"),
                           lambda: _ffi.BidCols(p(self.auction), p(self.bidder), p(self.price), p(self.b_date_time), self.rows))


@dataclass
class Auctions:
    a_id: "object" = None
    seller: "object" = None
    category: "object" = None
    rows: int = 0
    a_date_time: "object" = None   # q4 / q9 only
    expires: "object" = None       # q4 / q9 only

    def ffi_times(self) -> _ffi.AuctionTimeCols:
        p = lambda t: None if t is None else t.data_ptr()
        return _ffi.AuctionTimeCols(p(self.a_id), p(self.category), p(self.a_date_time), p(self.expires), self.rows)

    def ffi(self) -> _ffi.AuctionCols:
        p = lambda t: None if t is None else t.data_ptr()
        return _cached_ffi(self, ("a_id", "seller", "category"), lambda: _ffi.AuctionCols(p(self.a_id), p(self.seller), p(self.category), self.rows))


@dataclass
class Persons:
    p_id: "object" = None
    name: Optional[DeviceUtf8] = None
    city: Optional[DeviceUtf8] = None
    state: Optional[DeviceUtf8] = None
    rows: int = 0

    def ffi(self) -> _ffi.PersonCols:
        z = _ffi.Utf8(None, None)
        u = lambda c: z if c is None else c.ffi()
        return _cached_ffi(self, ("p_id", "name", "city", "state"),
                           lambda: _ffi.PersonCols(None if self.p_id is None else self.p_id.data_ptr(), u(self.name), u(self.city), u(self.state), self.rows))


@dataclass
class WindowSchedule:
    """Row panes + windows over one relation (include/flockgpu.h `flockgpu_windows`)."""
    pane_row_offsets: np.ndarray  # int64, n_panes + 1
    win_pane_lo: np.ndarray       # int32, n_windows
    win_pane_hi: np.ndarray       # int32, n_windows

    def __post_init__(self):
        self.pane_row_offsets = np.ascontiguousarray(self.pane_row_offsets, np.int64)
        self.win_pane_lo = np.ascontiguousarray(self.win_pane_lo, np.int32)
        self.win_pane_hi = np.ascontiguousarray(self.win_pane_hi, np.int32)

    @property
    def n_windows(self) -> int:
        return len(self.win_pane_lo)

    def ffi(self) -> _ffi.Windows:
        return _cached_ffi(self, ("pane_row_offsets", "win_pane_lo", "win_pane_hi"),
                           lambda: _ffi.Windows(self.pane_row_offsets.ctypes.data_as(C.POINTER(C.c_int64)), len(self.pane_row_offsets) - 1,
                                                self.win_pane_lo.ctypes.data_as(C.POINTER(C.c_int32)),
                                                self.win_pane_hi.ctypes.data_as(C.POINTER(C.c_int32)), len(self.win_pane_lo)))

    def window_rows(self, w: int):
        return int(self.pane_row_offsets[self.win_pane_lo[w]]), int(self.pane_row_offsets[self.win_pane_hi[w]])

    @staticmethod
    def single(rows: int) -> "WindowSchedule":
        """One window over the whole relation = one `collect` call of the reference."""
        return WindowSchedule(np.array([0, rows]), np.array([0]), np.array([1]))


# ------------------------------------------------------------------ results (host copies on demand)
@dataclass
class Q2Out:
    ctx: "GpuContext"
    raw: _ffi.Q2Result
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def to_host(self):
        return (self.ctx.d2h(self.raw.auction, self.rows, np.int32), self.ctx.d2h(self.raw.price, self.rows, np.int32),
                self.offsets())


@dataclass
class Q3Out:
    ctx: "GpuContext"
    raw: _ffi.Q3Result
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def to_host(self):
        n = self.rows
        u = lambda c, nb: (self.ctx.d2h(c.offsets, n + 1, np.int32), self.ctx.d2h(c.data, int(nb), np.uint8))
        return {
            "name": u(self.raw.name, self.raw.name_bytes), "city": u(self.raw.city, self.raw.city_bytes),
            "state": u(self.raw.state, self.raw.state_bytes), "a_id": self.ctx.d2h(self.raw.a_id, n, np.int32),
            "auction_row": self.ctx.d2h(self.raw.auction_row, n, np.int32),
            "person_row": self.ctx.d2h(self.raw.person_row, n, np.int32), "offsets": self.offsets(),
        }


@dataclass
class Q5Out:
    ctx: "GpuContext"
    raw: _ffi.Q5Result
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def win_max(self):
        return np.ctypeslib.as_array(self.raw.win_max, (max(self.n_windows, 1),))[: self.n_windows].copy()

    def win_groups(self):
        return np.ctypeslib.as_array(self.raw.win_groups, (max(self.n_windows, 1),))[: self.n_windows].copy()

    def to_host(self):
        return (self.ctx.d2h(self.raw.auction, self.rows, np.int32), self.ctx.d2h(self.raw.num, self.rows, np.uint64),
                self.offsets())


@dataclass
class Q7Out:
    ctx: "GpuContext"
    raw: _ffi.Q7Result
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def win_max(self):
        return np.ctypeslib.as_array(self.raw.win_max, (max(self.n_windows, 1),))[: self.n_windows].copy()

    def to_host(self):
        n = self.rows
        return {"auction": self.ctx.d2h(self.raw.auction, n, np.int32), "price": self.ctx.d2h(self.raw.price, n, np.int32),
                "bidder": self.ctx.d2h(self.raw.bidder, n, np.int32), "b_date_time": self.ctx.d2h(self.raw.b_date_time, n, np.int64),
                "offsets": self.offsets()}


@dataclass
class Q11Out:
    ctx: "GpuContext"
    raw: _ffi.Q11Result
    n_epochs: int

    @property
    def rows(self):
        return int(self.raw.rows)

    @property
    def sessions_total(self):
        return int(self.raw.sessions_total)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.epoch_out_offsets, (self.n_epochs + 1,)).copy()

    def to_host(self):
        n = self.rows
        return {"bidder": self.ctx.d2h(self.raw.bidder, n, np.int32), "bid_count": self.ctx.d2h(self.raw.bid_count, n, np.uint64),
                "start_time": self.ctx.d2h(self.raw.start_time, n, np.int64), "end_time": self.ctx.d2h(self.raw.end_time, n, np.int64),
                "offsets": self.offsets()}


@dataclass
class Q9Out:
    ctx: "GpuContext"
    raw: _ffi.Q9Result
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def to_host(self):
        n = self.rows
        return {"auction": self.ctx.d2h(self.raw.auction, n, np.int32), "price": self.ctx.d2h(self.raw.price, n, np.int32),
                "bidder": self.ctx.d2h(self.raw.bidder, n, np.int32), "b_date_time": self.ctx.d2h(self.raw.b_date_time, n, np.int64),
                "offsets": self.offsets()}


@dataclass
class Q4Out:
    ctx: "GpuContext"
    raw: _ffi.Q4Result
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def to_host(self):
        n = self.rows
        return {"category": self.ctx.d2h(self.raw.category, n, np.int32), "avg": self.ctx.d2h(self.raw.avg_final, n, np.float64),
                "offsets": self.offsets()}


@dataclass
class Q13Out:
    ctx: "GpuContext"
    raw: _ffi.Q13Result
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def to_host(self):
        n, d = self.rows, self.ctx.d2h
        return {"auction": d(self.raw.auction, n, np.int32), "bidder": d(self.raw.bidder, n, np.int32),
                "price": d(self.raw.price, n, np.int32), "b_date_time": d(self.raw.b_date_time, n, np.int64),
                "value": d(self.raw.value, n, np.int32), "bid_row": d(self.raw.bid_row, n, np.int32),
                "side_row": d(self.raw.side_row, n, np.int32), "offsets": self.offsets()}


@dataclass
class Q8Out:
    ctx: "GpuContext"
    raw: _ffi.Q8Result
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def to_host(self):
        n = self.rows
        return {
            "p_id": self.ctx.d2h(self.raw.p_id, n, np.int32),
            "name": (self.ctx.d2h(self.raw.name.offsets, n + 1, np.int32),
                     self.ctx.d2h(self.raw.name.data, int(self.raw.name_bytes), np.uint8)),
            "person_row": self.ctx.d2h(self.raw.person_row, n, np.int32), "offsets": self.offsets(),
        }


# ------------------------------------------------------------------ communicators (include/flockgpu_comm.h)
class Comm:
    """One rank's handle of a flockgpu communicator."""

    def __init__(self, h, lib):
        self.h, self._lib = h, lib

    @property
    def rank(self) -> int:
        return self._lib.flockgpu_comm_rank(self.h)

    @property
    def size(self) -> int:
        return self._lib.flockgpu_comm_size(self.h)

    @property
    def transport(self) -> str:
        return self._lib.flockgpu_comm_transport(self.h).decode()

    def close(self):
        if self.h:
            self._lib.flockgpu_comm_destroy(self.h)
            self.h = None

    def inject_failure(self, where: int):
        """Test hook (include/flockgpu_comm.h): 1 = the next exchange fails before the agreement, 2 = after it."""
        if self._lib.flockgpu_comm_inject_failure(self.h, where) != _ffi.OK:
            raise FlockGpuError(_ffi.ERR_INVALID, "flockgpu_comm_inject_failure: bad argument")

    def set_timeout(self, seconds: float):
        """Deadline of the waits behind RCCL work (flockgpu_comm_set_timeout): a peer gone after the agreement costs at most this."""
        if self._lib.flockgpu_comm_set_timeout(self.h, float(seconds)) != _ffi.OK:
            raise FlockGpuError(_ffi.ERR_INVALID, "flockgpu_comm_set_timeout: bad argument")

    def set_max_piece_bytes(self, n: int):
        """Largest single transfer per peer (flockgpu_comm_set_max_piece_bytes); every rank must set the same value."""
        if self._lib.flockgpu_comm_set_max_piece_bytes(self.h, int(n)) != _ffi.OK:
            raise FlockGpuError(_ffi.ERR_INVALID, "flockgpu_comm_set_max_piece_bytes: bad argument")

    def phases(self, on: bool = True, reset: bool = True):
        """Per-phase stream timeline of this rank's exchange calls on / off."""
        self._lib.flockgpu_comm_phase_enable(self.h, 1 if on else 0)
        if reset:
            self._lib.flockgpu_comm_phase_reset(self.h)

    def phase_times(self) -> dict:
        """{phase: {"calls": n, "total_ms": t}} since the last reset, in the order the phases first ran."""
        n = C.c_int(0)
        buf = (_ffi.KernelStat * 32)()
        self._lib.flockgpu_comm_phase_read(self.h, buf, 32, C.byref(n))
        return {buf[i].name.decode(): {"calls": int(buf[i].launches), "total_ms": float(buf[i].total_ms)} for i in range(min(n.value, 32))}

    @staticmethod
    def local(n_ranks: int):
        """n_ranks handles for ranks that are threads of this process (flockgpu_comm_init_local)."""
        lib = _ffi.load()
        hs = (C.c_void_p * n_ranks)()
        rc = lib.flockgpu_comm_init_local(n_ranks, hs)
        if rc != _ffi.OK:
            raise FlockGpuError(rc, "flockgpu_comm_init_local failed")
        return [Comm(C.c_void_p(h), lib) for h in hs]

    @staticmethod
    def from_torch_distributed(ctx: "GpuContext", group=None, transport: str = "rccl"):
        """One process per rank: rank 0 draws the id, torch.distributed (the host's own channel) ships its 128 bytes, every rank calls
        flockgpu_comm_init_rank collectively -- one GPU per process, RCCL -- or, `transport="ipc"`, flockgpu_comm_init_ipc: the same
        protocol between processes that may share a device (hipIpc mappings + a shared-memory segment; RCCL refuses that)."""
        import torch.distributed as dist
        lib = _ffi.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = C.create_string_buffer(128)
        if rank == 0:
            rc = lib.flockgpu_comm_unique_id(buf)
            if rc != _ffi.OK:
                raise FlockGpuError(rc, "flockgpu_comm_unique_id failed")
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        h = C.c_void_p()
        init = {"rccl": lib.flockgpu_comm_init_rank, "ipc": lib.flockgpu_comm_init_ipc}[transport]
        ctx._check(init(ctx._h, box[0], world, rank, C.byref(h)))
        return Comm(h, lib)

    @staticmethod
    def ipc(ctx: "GpuContext", comm_id: bytes, n_ranks: int, rank: int):
        """flockgpu_comm_init_ipc with an id the host shipped itself (rank 0: `Comm.unique_id()`)."""
        lib = _ffi.load()
        h = C.c_void_p()
        ctx._check(lib.flockgpu_comm_init_ipc(ctx._h, comm_id, n_ranks, rank, C.byref(h)))
        return Comm(h, lib)

    @staticmethod
    def unique_id() -> bytes:
        lib = _ffi.load()
        buf = C.create_string_buffer(128)
        rc = lib.flockgpu_comm_unique_id(buf)
        if rc != _ffi.OK:
            raise FlockGpuError(rc, "flockgpu_comm_unique_id failed")
        return bytes(buf.raw)


# ------------------------------------------------------------------ the context
class GuardedBuffer:
    """A device column in guarded memory (GpuContext.guarded): pointer, element count, numpy dtype."""

    def __init__(self, ptr: int, count: int, dtype, owner=None):
        self._ptr, self._count, self.dtype, self._owner = ptr, count, dtype, owner

    def cpu(self):
        """The buffer's contents as a host torch tensor (what the tests ask of a device tensor)."""
        import torch
        return torch.from_numpy(self._owner.d2h(self._ptr, self._count, self.dtype))

    def data_ptr(self) -> int:
        return self._ptr

    def numel(self) -> int:
        return self._count

    def __len__(self) -> int:
        return self._count

    def __getitem__(self, key):   # contiguous slices only: a view of the same memory
        if not isinstance(key, slice) or key.step not in (None, 1):
            raise TypeError("GuardedBuffer: contiguous slices only")
        lo, hi, _ = key.indices(self._count)
        return GuardedBuffer(self._ptr + lo * np.dtype(self.dtype).itemsize, max(hi - lo, 0), self.dtype, self._owner)


class GpuContext:
    """Owns a `flockgpu_ctx`.  `stream` defaults to torch's current stream on `device`."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, own_stream: bool = False):
        self._lib = _ffi.load()
        self.device = device
        if stream is None and not own_stream:
            torch = _torch()
            if not torch.cuda.is_available():
                raise FlockGpuError(_ffi.ERR_HIP, "no HIP device visible: flock_amd has no CPU fallback")
            torch.cuda.set_device(device)
            stream = torch.cuda.current_stream(device).cuda_stream
        h = C.c_void_p()
        rc = self._lib.flockgpu_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        self._h = h
        if rc != _ffi.OK:
            msg = self._lib.flockgpu_last_error(h).decode() if h else "ctx_create failed"
            if h:
                self._lib.flockgpu_ctx_destroy(h)
            self._h = None
            raise FlockGpuError(rc, msg)

    # -- plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.flockgpu_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != _ffi.OK:
            raise FlockGpuError(rc, self._lib.flockgpu_last_error(self._h).decode())

    def synchronize(self):
        self._check(self._lib.flockgpu_ctx_synchronize(self._h))

    def d2h(self, dev_ptr, count: int, dtype) -> np.ndarray:
        out = np.empty(count, dtype)
        if count:
            self._check(self._lib.flockgpu_memcpy(self._h, out.ctypes.data_as(C.c_void_p), dev_ptr, out.nbytes, _ffi.D2H))
        return out

    def guarded(self, host: np.ndarray) -> "GuardedBuffer":
        """`host` copied into device memory that ENDS where mapped address space ends (flockgpu_malloc_guarded): a column placed
        there turns a kernel's read past its end into a fault.  The object stands in for a device tensor wherever the engine only
        asks for `data_ptr()` (the column structs); it lives until the ctx closes."""
        host = np.ascontiguousarray(host)
        p = C.c_void_p()
        self._check(self._lib.flockgpu_malloc_guarded(self._h, max(host.nbytes, 1), C.byref(p)))
        # (16-byte aligned start; up to 12 bytes of mapped slack behind a column whose size is not a multiple of 16)
        if host.nbytes:
            self._check(self._lib.flockgpu_memcpy(self._h, p, host.ctypes.data_as(C.c_void_p), host.nbytes, _ffi.H2D))
        return GuardedBuffer(p.value, host.size, host.dtype, self)

    def profile(self, on: bool):
        self._check(self._lib.flockgpu_profile_enable(self._h, 1 if on else 0))

    def profile_only(self, kernel_name=None):
        """Bracket only launches of `kernel_name` (None: all kernels)."""
        self._check(self._lib.flockgpu_profile_only(self._h, kernel_name.encode() if kernel_name else None))

    def profile_reset(self):
        self._check(self._lib.flockgpu_profile_reset(self._h))

    def profile_read(self) -> dict:
        n = C.c_int(0)
        buf = (_ffi.KernelStat * 64)()
        self._check(self._lib.flockgpu_profile_read(self._h, buf, 64, C.byref(n)))
        return {buf[i].name.decode(): {"launches": int(buf[i].launches), "total_ms": float(buf[i].total_ms)}
                for i in range(min(n.value, 64))}

    def profile_samples(self, kernel_name: str) -> list:
        """Per-launch durations (ms, launch order) of one kernel since the last reset."""
        n = C.c_int(0)
        buf = (C.c_float * 4096)()
        self._check(self._lib.flockgpu_profile_samples(self._h, kernel_name.encode(), buf, 4096, C.byref(n)))
        return [float(buf[i]) for i in range(min(n.value, 4096))]

    # -- operators
    def q1_project(self, bids: Bids, factor: float = 0.908):
        """q1 ProjectionExec (planner.rs:90): returns the Float64 `price` column (device tensor);
        auction / bidder / b_date_time pass through untouched (zero-copy like the reference's Arc clones)."""
        torch = _torch()
        out = torch.empty(bids.rows, dtype=torch.float64, device=f"cuda:{self.device}")
        b = bids.ffi()
        self._check(self._lib.flockgpu_q1_project(self._h, C.byref(b), factor, out.data_ptr()))
        return out

    def q2_filter(self, bids: Bids, windows: WindowSchedule, modulus: int = 123) -> Q2Out:
        b, w, r = bids.ffi(), windows.ffi(), _ffi.Q2Result()
        self._check(self._lib.flockgpu_q2_filter(self._h, C.byref(b), C.byref(w), modulus, C.byref(r)))
        return Q2Out(self, r, windows.n_windows)

    def q3_join(self, auctions: Auctions, auction_windows: WindowSchedule, persons: Persons,
                person_windows: WindowSchedule, category: int = 10, states: Sequence[str] = ("or", "id", "ca")) -> Q3Out:
        a, aw, p, pw, r = auctions.ffi(), auction_windows.ffi(), persons.ffi(), person_windows.ffi(), _ffi.Q3Result()
        lits = (C.c_char_p * len(states))(*[s.encode() for s in states])
        self._check(self._lib.flockgpu_q3_join(self._h, C.byref(a), C.byref(aw), C.byref(p), C.byref(pw), category, lits,
                                               len(states), C.byref(r)))
        return Q3Out(self, r, auction_windows.n_windows)

    def q5_hot_items(self, bids: Bids, windows: WindowSchedule) -> Q5Out:
        b, w, r = bids.ffi(), windows.ffi(), _ffi.Q5Result()
        self._check(self._lib.flockgpu_q5_hot_items(self._h, C.byref(b), C.byref(w), C.byref(r)))
        return Q5Out(self, r, windows.n_windows)

    def _device_view(self, ptr, n, dtype):
        """A torch tensor over `n` elements of library-owned device memory at `ptr` (no copy; __cuda_array_interface__)."""
        torch = _torch()
        if not n or not ptr:
            return torch.empty(0, dtype=dtype, device=f"cuda:{self.device}")
        typestr = {torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]

        class _Span:
            __cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_Span(), device=f"cuda:{self.device}")

    def json_lines_decode(self, text, fields, borrow=False):
        """Newline-delimited JSON (uint8 device tensor) -> columns: `fields` = [(name, "int32" | "int64" | "utf8")].
        Returns {name: int32 / int64 device tensor | DeviceUtf8} and the row count (`event_bytes_to_batch`,
        flock/src/transmute.rs:255-266, on the device).  borrow: the tensors are views of the library's own result columns
        (what the C ABI hands out: valid until the next call on this context) instead of copies of them."""
        torch = _torch()
        kinds = {"int32": _ffi.JSON_INT32, "int64": _ffi.JSON_INT64, "utf8": _ffi.JSON_UTF8}
        spec = (_ffi.JsonField * len(fields))(*[_ffi.JsonField(n.encode(), kinds[t]) for n, t in fields])
        cols = (_ffi.JsonColumn * len(fields))()
        rows = C.c_int64(0)
        self._check(self._lib.flockgpu_json_lines_decode(self._h, text.data_ptr(), text.numel(), spec, len(fields), cols, C.byref(rows)))
        n, dev, out = rows.value, f"cuda:{self.device}", {}
        if borrow:
            for (name, t), c in zip(fields, cols):
                if t == "utf8":
                    out[name] = DeviceUtf8(self._device_view(c.utf8.offsets, n + 1, torch.int32), self._device_view(c.utf8.data, int(c.utf8_bytes), torch.uint8))
                else:
                    out[name] = self._device_view(c.values, n, torch.int32 if t == "int32" else torch.int64)
            return out, n
        for (name, t), c in zip(fields, cols):
            if t == "utf8":
                off = torch.empty(n + 1, dtype=torch.int32, device=dev)
                data = torch.empty(max(int(c.utf8_bytes), 16), dtype=torch.uint8, device=dev)
                if n:
                    self._check(self._lib.flockgpu_memcpy(self._h, off.data_ptr(), c.utf8.offsets, (n + 1) * 4, _ffi.D2D))
                else:
                    off.zero_()
                if c.utf8_bytes:
                    self._check(self._lib.flockgpu_memcpy(self._h, data.data_ptr(), c.utf8.data, int(c.utf8_bytes), _ffi.D2D))
                out[name] = DeviceUtf8(off, data)
            else:
                dt, w = (torch.int32, 4) if t == "int32" else (torch.int64, 8)
                col = torch.empty(n, dtype=dt, device=dev)
                if n:
                    self._check(self._lib.flockgpu_memcpy(self._h, col.data_ptr(), c.values, n * w, _ffi.D2D))
                out[name] = col
        return out, n

    def q5_partial_counts(self, bids: Bids, windows: WindowSchedule):
        """q5.dag's Partial stage: COUNT(*) GROUP BY auction per PANE of the schedule.  Returns (auction int32 tensor, count
        int32 tensor holding uint32 counts, pane_out_offsets np.int64[n_panes + 1])."""
        torch = _torch()
        b, w, r = bids.ffi(), windows.ffi(), _ffi.Q5PartialResult()
        self._check(self._lib.flockgpu_q5_partial_counts(self._h, C.byref(b), C.byref(w), C.byref(r)))
        n, n_panes = int(r.rows), len(windows.pane_row_offsets) - 1
        dev = f"cuda:{self.device}"
        key = torch.empty(n, dtype=torch.int32, device=dev)
        cnt = torch.empty(n, dtype=torch.int32, device=dev)
        if n:
            self._check(self._lib.flockgpu_memcpy(self._h, key.data_ptr(), r.auction, n * 4, _ffi.D2D))
            self._check(self._lib.flockgpu_memcpy(self._h, cnt.data_ptr(), r.count, n * 4, _ffi.D2D))
        return key, cnt, np.ctypeslib.as_array(r.pane_out_offsets, (n_panes + 1,)).copy()

    def q5_hot_items_weighted(self, auction, count, windows: WindowSchedule) -> Q5Out:
        """q5.dag's FinalPartitioned stage + MAX + join over (auction, count) rows (count: int32 tensor of uint32 bits)."""
        w, r = windows.ffi(), _ffi.Q5Result()
        self._check(self._lib.flockgpu_q5_hot_items_weighted(self._h, auction.data_ptr(), count.data_ptr(), auction.numel(), C.byref(w),
                                                             C.byref(r)))
        return Q5Out(self, r, windows.n_windows)

    def q7_highest_bid(self, bids: Bids, windows: WindowSchedule) -> Q7Out:
        """q7 (q7.sql): the bids that reach the window's MAX(price); ties kept, input order."""
        b, w, r = bids.ffi(), windows.ffi(), _ffi.Q7Result()
        self._check(self._lib.flockgpu_q7_highest_bid(self._h, C.byref(b), C.byref(w), C.byref(r)))
        return Q7Out(self, r, windows.n_windows)

    def q11_user_sessions(self, bids: Bids, epoch_row_offsets, timeout_s: int, base_time_ms: int) -> Q11Out:
        """q11 (q11.sql under Window::Session, flock-function/src/aws/window/session.rs): the bidder sessions closed in every
        epoch of the run; epoch t = rows [epoch_row_offsets[t], epoch_row_offsets[t+1])."""
        off = np.ascontiguousarray(epoch_row_offsets, dtype=np.int64)
        b, r = bids.ffi(), _ffi.Q11Result()
        self._check(self._lib.flockgpu_q11_user_sessions(self._h, C.byref(b), off.ctypes.data_as(C.POINTER(C.c_int64)), len(off) - 1,
                                                         int(timeout_s), int(base_time_ms), C.byref(r)))
        return Q11Out(self, r, len(off) - 1)

    def group_rows_by_key(self, keys, n=None):
        """Stable grouping of rows by an int32 device column: (sorted keys, row numbers) as host arrays' device views."""
        n = int(keys.numel() if n is None else n)
        k, v = C.c_void_p(), C.c_void_p()
        self._check(self._lib.flockgpu_group_rows_by_key(self._h, keys.data_ptr(), n, C.byref(k), C.byref(v)))
        return self.d2h(k.value, n, np.int32), self.d2h(v.value, n, np.int32)

    def q9_winning_bids(self, auctions: Auctions, auction_windows: WindowSchedule, bids: Bids,
                        bid_windows: WindowSchedule) -> Q9Out:
        """q9 (q9.sql): the bids whose price is their auction's MAX(price) over the bids inside [a_date_time, expires]."""
        a, aw, b, bw, r = auctions.ffi_times(), auction_windows.ffi(), bids.ffi(), bid_windows.ffi(), _ffi.Q9Result()
        self._check(self._lib.flockgpu_q9_winning_bids(self._h, C.byref(a), C.byref(aw), C.byref(b), C.byref(bw), C.byref(r)))
        return Q9Out(self, r, auction_windows.n_windows)

    def q4_avg_final_by_category(self, auctions: Auctions, auction_windows: WindowSchedule, bids: Bids,
                                 bid_windows: WindowSchedule) -> Q4Out:
        """q4 (q4.sql): AVG over the auctions' final prices, per category."""
        a, aw, b, bw, r = auctions.ffi_times(), auction_windows.ffi(), bids.ffi(), bid_windows.ffi(), _ffi.Q4Result()
        self._check(self._lib.flockgpu_q4_avg_final_by_category(self._h, C.byref(a), C.byref(aw), C.byref(b), C.byref(bw),
                                                                C.byref(r)))
        return Q4Out(self, r, auction_windows.n_windows)

    def q13_side_join(self, bids: Bids, windows: WindowSchedule, side_key, side_value) -> Q13Out:
        """q13 (q13.sql): bid JOIN side_input ON auction = key; side_key / side_value are int32 device tensors."""
        b, w, r = bids.ffi(), windows.ffi(), _ffi.Q13Result()
        self._check(self._lib.flockgpu_q13_side_join(self._h, C.byref(b), C.byref(w), side_key.data_ptr(), side_value.data_ptr(),
                                                     side_key.numel(), C.byref(r)))
        return Q13Out(self, r, windows.n_windows)

    def q8_join(self, persons: Persons, person_windows: WindowSchedule, auctions: Auctions,
                auction_windows: WindowSchedule) -> Q8Out:
        p, pw, a, aw, r = persons.ffi(), person_windows.ffi(), auctions.ffi(), auction_windows.ffi(), _ffi.Q8Result()
        self._check(self._lib.flockgpu_q8_join(self._h, C.byref(p), C.byref(pw), C.byref(a), C.byref(aw), C.byref(r)))
        return Q8Out(self, r, person_windows.n_windows)

    # -- asynchronous twins (flockgpu.h "asynchronous calls"): the call runs on the ctx's worker thread -- the reference's
    # `tokio::spawn(collect(plan))`, context.rs:172-191 --, `.wait()` joins it.  One call in flight per context; calls on
    # different contexts overlap on the GPU.
    class _Pending:
        def __init__(self, gpu, keep, make):
            self._gpu, self._keep, self._make = gpu, keep, make

        def wait(self):
            self._gpu._check(self._gpu._lib.flockgpu_ctx_wait(self._gpu._h))
            out = self._make()
            self._keep = None
            return out

    def q5_hot_items_async(self, bids: Bids, windows: WindowSchedule) -> "GpuContext._Pending":
        b, w, r = bids.ffi(), windows.ffi(), _ffi.Q5Result()
        self._check(self._lib.flockgpu_q5_hot_items_async(self._h, C.byref(b), C.byref(w), C.byref(r)))
        return GpuContext._Pending(self, (bids, windows, b, w, r), lambda: Q5Out(self, r, windows.n_windows))

    def q3_join_async(self, auctions: Auctions, auction_windows: WindowSchedule, persons: Persons, person_windows: WindowSchedule,
                      category: int = 10, states: Sequence[str] = ("or", "id", "ca")) -> "GpuContext._Pending":
        a, aw, p, pw, r = auctions.ffi(), auction_windows.ffi(), persons.ffi(), person_windows.ffi(), _ffi.Q3Result()
        lits = (C.c_char_p * len(states))(*[s.encode() for s in states])
        self._check(self._lib.flockgpu_q3_join_async(self._h, C.byref(a), C.byref(aw), C.byref(p), C.byref(pw), category, lits, len(states), C.byref(r)))
        return GpuContext._Pending(self, (auctions, auction_windows, persons, person_windows, a, aw, p, pw, lits, r),
                                   lambda: Q3Out(self, r, auction_windows.n_windows))

    def q8_join_async(self, persons: Persons, person_windows: WindowSchedule, auctions: Auctions,
                      auction_windows: WindowSchedule) -> "GpuContext._Pending":
        p, pw, a, aw, r = persons.ffi(), person_windows.ffi(), auctions.ffi(), auction_windows.ffi(), _ffi.Q8Result()
        self._check(self._lib.flockgpu_q8_join_async(self._h, C.byref(p), C.byref(pw), C.byref(a), C.byref(aw), C.byref(r)))
        return GpuContext._Pending(self, (persons, person_windows, auctions, auction_windows, p, pw, a, aw, r),
                                   lambda: Q8Out(self, r, person_windows.n_windows))

    # -- in-library exchange (include/flockgpu_comm.h): every window striped over the ranks of `comm`
    def q5_hot_items_exchange(self, comm: "Comm", bids: Bids, windows: WindowSchedule) -> Q5Out:
        b, w, r = bids.ffi(), windows.ffi(), _ffi.Q5Result()
        self._check(self._lib.flockgpu_q5_hot_items_exchange(self._h, comm.h, C.byref(b), C.byref(w), C.byref(r)))
        return Q5Out(self, r, windows.n_windows)

    def q3_join_exchange(self, comm: "Comm", auctions: Auctions, auction_windows: WindowSchedule, persons: Persons,
                         person_windows: WindowSchedule, category: int = 10, states: Sequence[str] = ("or", "id", "ca")) -> Q3Out:
        a, aw, p, pw, r = auctions.ffi(), auction_windows.ffi(), persons.ffi(), person_windows.ffi(), _ffi.Q3Result()
        lits = (C.c_char_p * len(states))(*[s.encode() for s in states])
        self._check(self._lib.flockgpu_q3_join_exchange(self._h, comm.h, C.byref(a), C.byref(aw), C.byref(p), C.byref(pw), category,
                                                        lits, len(states), C.byref(r)))
        return Q3Out(self, r, auction_windows.n_windows)

    def q8_join_exchange(self, comm: "Comm", persons: Persons, person_windows: WindowSchedule, auctions: Auctions,
                         auction_windows: WindowSchedule) -> Q8Out:
        p, pw, a, aw, r = persons.ffi(), person_windows.ffi(), auctions.ffi(), auction_windows.ffi(), _ffi.Q8Result()
        self._check(self._lib.flockgpu_q8_join_exchange(self._h, comm.h, C.byref(p), C.byref(pw), C.byref(a), C.byref(aw), C.byref(r)))
        return Q8Out(self, r, person_windows.n_windows)

    def comm_barrier(self, comm: "Comm"):
        self._check(self._lib.flockgpu_comm_barrier(self._h, comm.h))

    # -- exchange building blocks (include/flockgpu.h "key-partitioned exchange")
    def partition_by_key(self, keys, windows: WindowSchedule, n_parts: int):
        """Row numbers grouped by (partition, window) + the [n_parts, n_windows] row counts.
        Returns (rows: int32 device tensor, counts: np.ndarray[int64])."""
        torch = _torch()
        w, r = windows.ffi(), _ffi.PartitionResult()
        self._check(self._lib.flockgpu_partition_by_key(self._h, keys.data_ptr(), keys.numel(), C.byref(w), n_parts, C.byref(r)))
        n_groups = n_parts * windows.n_windows
        off = np.ctypeslib.as_array(r.part_win_offsets, (n_groups + 1,)).copy()
        rows = torch.empty(int(r.rows), dtype=torch.int32, device=f"cuda:{self.device}")
        if r.rows:
            self._check(self._lib.flockgpu_memcpy(self._h, rows.data_ptr(), r.row, int(r.rows) * 4, _ffi.D2D))
        return rows, np.diff(off).reshape(n_parts, windows.n_windows)

    def partition_by_key_raw(self, keys, windows: WindowSchedule, n_parts: int) -> int:
        """The same call without the copy of the row numbers out of the ctx arena (timing: tools/gpu_partition_ab.py); returns the row count."""
        w, r = windows.ffi(), _ffi.PartitionResult()
        self._check(self._lib.flockgpu_partition_by_key(self._h, keys.data_ptr(), keys.numel(), C.byref(w), n_parts, C.byref(r)))
        return int(r.rows)

    def take(self, src, rows):
        """out[i] = src[rows[i]] for an int32 / int64 device tensor."""
        torch = _torch()
        out = torch.empty(rows.numel(), dtype=src.dtype, device=src.device)
        fn = {torch.int32: self._lib.flockgpu_take_i32, torch.int64: self._lib.flockgpu_take_i64}[src.dtype]
        self._check(fn(self._h, src.data_ptr(), rows.data_ptr(), rows.numel(), out.data_ptr()))
        return out

    def take_utf8(self, src: DeviceUtf8, rows, slot: int = 0) -> DeviceUtf8:
        torch = _torch()
        s, o, nb = src.ffi(), _ffi.Utf8(), C.c_int64(0)
        n = rows.numel()
        self._check(self._lib.flockgpu_take_utf8(self._h, C.byref(s), rows.data_ptr(), n, slot, C.byref(o), C.byref(nb)))
        dev = f"cuda:{self.device}"
        off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        data = torch.empty(max(nb.value, 16), dtype=torch.uint8, device=dev)
        self._check(self._lib.flockgpu_memcpy(self._h, off.data_ptr(), o.offsets, (n + 1) * 4, _ffi.D2D))
        if nb.value:
            self._check(self._lib.flockgpu_memcpy(self._h, data.data_ptr(), o.data, nb.value, _ffi.D2D))
        return DeviceUtf8(off, data)

    def offsets_from_lengths(self, lengths) -> "object":
        """Arrow Utf8 offsets (n + 1, int32) from value lengths (n, int32)."""
        torch = _torch()
        n = lengths.numel()
        off = torch.zeros(n + 1, dtype=torch.int32, device=lengths.device)
        if n:
            off[1:] = lengths
            self._check(self._lib.flockgpu_inclusive_scan_i32(self._h, off.data_ptr() + 4, n))
        return off
