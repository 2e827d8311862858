"""`ExecutionContext` -- the GPU twin of flock/src/runtime/context.rs, over the plan-level C ABI.

Same surface and argument meaning as the reference:

    ctx = ExecutionContext(plans=[plan_json, ...], name="SX72HzqFz1Qij4bP-00-00")
    ctx.feed_data_sources(sources)      # sources[relation][partition][batch]   (context.rs:257-325)
    batches = ctx.execute()             # -> [plan][batch]                      (context.rs:172-191)
    parts = ctx.execute_partitioned()   # -> [plan][partition][batch]           (context.rs:197-216)
    ctx.clean_data_sources()            #                                       (context.rs:227-254)

`collect(ctx, streams)` restates `actor::collect` (flock-function/src/aws/actor.rs:54-79); `explain(plan)` shows the
operator tree the engine built and what executes each node.  Plans are the
serde_json text of the reference's physical plans; data are pyarrow RecordBatches handed over through the
Arrow C Data Interface (what arrow-rs would export as FFI_ArrowArray).  A plan the engine does not recognise
raises `FlockGpuError` with status FLOCKGPU_ERR_UNSUPPORTED (the reference host would keep DataFusion for it).
"""
from __future__ import annotations

import ctypes as C
import json
from typing import List, Optional, Sequence, Union

from . import _ffi
from ._ffi import FlockGpuError
from .engine import GpuContext

_ARROW_SCHEMA_BYTES, _ARROW_ARRAY_BYTES = 72, 80


def _pa():
    import pyarrow as pa
    return pa


# An exported ArrowArray / ArrowSchema is released by calling the `release` callback stored inside it (Arrow C Data Interface:
# offset 64 of the 80-byte ArrowArray, offset 56 of the 72-byte ArrowSchema).  Re-importing the struct into pyarrow releases it,
# too, but builds a Python object per column on the way -- per fed batch, 52 batches per window at the reference's granule.
_RELEASE = C.CFUNCTYPE(None, C.c_void_p)


def _release(addr: int, offset: int):
    fn = C.c_void_p.from_address(addr + offset).value
    if fn:
        _RELEASE(fn)(addr)


class _Plan:
    def __init__(self, gpu: GpuContext, text: str, generic_only: bool = False):
        self.gpu, self.text = gpu, text
        self._lib = _ffi.load()
        raw = text.encode()
        h = C.c_void_p()
        gpu._check(self._lib.flockgpu_plan_create_ex(gpu._h, raw, len(raw), _ffi.PLAN_GENERIC_ONLY if generic_only else 0, C.byref(h)))
        self.h = h
        self.query = self._lib.flockgpu_plan_query(h)
        self.inputs = [self._lib.flockgpu_plan_input_name(h, i).decode() for i in range(self._lib.flockgpu_plan_num_inputs(h))]
        self.description = self._lib.flockgpu_plan_description(h).decode()
        self.is_shuffling = bool(self._lib.flockgpu_plan_is_shuffling(h))
        self.partitions = self._lib.flockgpu_plan_output_partitions(h)
        self._fed = []   # the library borrows fed buffers until execute / reset returns (flockgpu_plan.h)
        self._match_cache = {}

    def close(self):
        if self.h:
            self._lib.flockgpu_plan_destroy(self.h)
            self.h = None

    def matches(self, i: int, schema) -> bool:
        try:   # a function instance sees the same few schemas on every invocation: ask the library once per (leaf, schema)
            hit = self._match_cache.get((i, schema))
        except TypeError:
            hit = None
        if hit is not None:
            return hit
        ok = self._matches_uncached(i, schema)
        try:
            self._match_cache[(i, schema)] = ok
        except TypeError:
            pass
        return ok

    def _matches_uncached(self, i: int, schema) -> bool:
        buf = C.create_string_buffer(_ARROW_SCHEMA_BYTES)
        schema._export_to_c(C.addressof(buf))
        ok = self._lib.flockgpu_plan_input_matches(self.h, i, C.cast(buf, C.c_void_p)) == 1
        _pa().Schema._import_from_c(C.addressof(buf))  # takes ownership back and releases
        return ok

    def feed(self, i: int, batches: Sequence, pane: Optional[int] = None):
        """feed_data_sources for leaf i; with `pane` the batches are pane `pane` of the plan's device-side window ring
        (flockgpu_plan_feed_pane: only the new pane crosses PCIe, the ring keeps the others)."""
        pa = _pa()
        batches = [b for b in batches if b is not None]
        if not batches:
            if pane is not None:   # an empty pane still advances the ring
                self.gpu._check(self._lib.flockgpu_plan_feed_pane(self.h, i, pane, None, None, 0))
            return
        sbuf = C.create_string_buffer(_ARROW_SCHEMA_BYTES)
        batches[0].schema._export_to_c(C.addressof(sbuf))
        abufs = [C.create_string_buffer(_ARROW_ARRAY_BYTES) for _ in batches]
        for b, ab in zip(batches, abufs):
            b._export_to_c(C.addressof(ab))
        ptrs = (C.c_void_p * len(batches))(*[C.addressof(ab) for ab in abufs])
        self._fed.append(batches)
        try:
            if pane is None:
                rc = self._lib.flockgpu_plan_feed(self.h, i, C.cast(sbuf, C.c_void_p), ptrs, len(batches))
            else:
                rc = self._lib.flockgpu_plan_feed_pane(self.h, i, pane, C.cast(sbuf, C.c_void_p), ptrs, len(batches))
        finally:
            # the library only borrowed the batches (self._fed keeps them alive): hand the exported structs back
            for ab in abufs:
                _release(C.addressof(ab), 64)
            _release(C.addressof(sbuf), 56)
        self.gpu._check(rc)

    def prefetch(self, i: int, batches: Sequence, pane: int) -> bool:
        """The NEXT pane's batches of leaf i start moving to the device now, beside whatever the plan is executing
        (flockgpu_plan_prefetch_pane); `feed_prefetched(i, pane)` appends them when the pane's turn has come.  False -- nothing was
        started -- when the pane holds what a prefetch does not take (NULLs): feed it the ordinary way then."""
        batches = [b for b in batches if b is not None]
        if not batches:
            return False
        sbuf = C.create_string_buffer(_ARROW_SCHEMA_BYTES)
        batches[0].schema._export_to_c(C.addressof(sbuf))
        abufs = [C.create_string_buffer(_ARROW_ARRAY_BYTES) for _ in batches]
        for b, ab in zip(batches, abufs):
            b._export_to_c(C.addressof(ab))
        ptrs = (C.c_void_p * len(batches))(*[C.addressof(ab) for ab in abufs])
        try:
            rc = self._lib.flockgpu_plan_prefetch_pane(self.h, i, pane, C.cast(sbuf, C.c_void_p), ptrs, len(batches))
        finally:
            for ab in abufs:
                _release(C.addressof(ab), 64)
            _release(C.addressof(sbuf), 56)
        if rc == _ffi.ERR_UNSUPPORTED:
            return False
        self.gpu._check(rc)
        self._prefetched = batches     # borrowed until the reset after the pane's feed
        return True

    def feed_prefetched(self, i: int, pane: int):
        self.gpu._check(self._lib.flockgpu_plan_feed_pane(self.h, i, pane, None, None, 0))
        self._fed.append(getattr(self, "_prefetched", None))
        self._prefetched = None

    def feed_shared(self, i: int, donor: "_Plan", donor_input: int) -> bool:
        """Leaf i reads the relation `donor`'s leaf was fed, in place (flockgpu_plan_feed_shared).  False -- and nothing changed --
        when the donor does not hold what this plan reads."""
        rc = self._lib.flockgpu_plan_feed_shared(self.h, i, donor.h, donor_input)
        if rc == _ffi.ERR_UNSUPPORTED:
            return False
        self.gpu._check(rc)
        return True

    def execute_retain(self) -> int:
        """Runs the plan and leaves the result on the device for `feed_from` of the plans that consume it; returns its row count."""
        rows = C.c_int64(0)
        self.gpu._check(self._lib.flockgpu_plan_execute_retain(self.h, C.byref(rows)))
        return rows.value

    def feed_from(self, i: int, producer: "_Plan") -> bool:
        """Leaf i reads `producer`'s retained result in place.  False -- nothing changed -- when that result lacks a column the leaf reads."""
        rc = self._lib.flockgpu_plan_feed_from(self.h, i, producer.h)
        if rc == _ffi.ERR_UNSUPPORTED:
            return False
        self.gpu._check(rc)
        return True

    def execute(self):
        pa = _pa()
        sbuf = C.create_string_buffer(_ARROW_SCHEMA_BYTES)
        abuf = C.create_string_buffer(_ARROW_ARRAY_BYTES)
        self.gpu._check(self._lib.flockgpu_plan_execute(self.h, C.cast(sbuf, C.c_void_p), C.cast(abuf, C.c_void_p)))
        return pa.RecordBatch._import_from_c(C.addressof(abuf), C.addressof(sbuf))

    def execute_partitioned(self):
        """[partition] -> RecordBatch: the plan's hash partitions when it is a shuffling stage, else one batch."""
        pa = _pa()
        cap = max(self.partitions, 1)
        sbuf = C.create_string_buffer(_ARROW_SCHEMA_BYTES)
        abufs = C.create_string_buffer(_ARROW_ARRAY_BYTES * cap)
        n = C.c_int(0)
        self.gpu._check(self._lib.flockgpu_plan_execute_partitioned(self.h, C.cast(sbuf, C.c_void_p), C.cast(abufs, C.c_void_p), cap, C.byref(n)))
        schema = pa.Schema._import_from_c(C.addressof(sbuf))
        return [pa.RecordBatch._import_from_c(C.addressof(abufs) + p * _ARROW_ARRAY_BYTES, schema) for p in range(n.value)]

    def reset(self):
        self.gpu._check(self._lib.flockgpu_plan_reset(self.h))
        self._fed = []

    # -- asynchronous execute (flockgpu_plan_execute_async / flockgpu_plan_wait)
    def execute_async(self, partitioned: bool = False):
        self.gpu._check(self._lib.flockgpu_plan_execute_async(self.h, 1 if partitioned else 0))
        self._async_partitioned = partitioned

    def wait(self):
        """The batches of the execute started by `execute_async`: one RecordBatch, or [partition] -> RecordBatch."""
        pa = _pa()
        cap = max(self.partitions, 1)
        sbuf = C.create_string_buffer(_ARROW_SCHEMA_BYTES)
        abufs = C.create_string_buffer(_ARROW_ARRAY_BYTES * cap)
        n = C.c_int(0)
        self.gpu._check(self._lib.flockgpu_plan_wait(self.h, C.cast(sbuf, C.c_void_p), C.cast(abufs, C.c_void_p), cap, C.byref(n)))
        schema = pa.Schema._import_from_c(C.addressof(sbuf))
        out = [pa.RecordBatch._import_from_c(C.addressof(abufs) + p * _ARROW_ARRAY_BYTES, schema) for p in range(n.value)]
        return out if self._async_partitioned else out[0]

    # -- device-side pane ring (flockgpu_plan_ring_*)
    def ring_open(self, panes_per_window: int):
        self.gpu._check(self._lib.flockgpu_plan_ring_open(self.h, panes_per_window))

    def ring_close(self):
        self.gpu._check(self._lib.flockgpu_plan_ring_close(self.h))
        self._fed = []

    def ring_state(self):
        first, n, ppw = C.c_int64(0), C.c_int(0), C.c_int(0)
        self.gpu._check(self._lib.flockgpu_plan_ring_state(self.h, C.byref(first), C.byref(n), C.byref(ppw)))
        return first.value, n.value, ppw.value


class ExecutionContext:
    """Cloud execution context of one function: plans + name (context.rs:103-131)."""

    def __init__(self, plans: Sequence[Union[str, dict]], name: str = "flockgpu-00", gpu: Optional[GpuContext] = None,
                 device: int = 0, generic_only: bool = False, gpus: Optional[Sequence[GpuContext]] = None):
        """`gpus`: one GpuContext (stream) per plan -- what `execute` needs to run the plans side by side like the reference's tokio
        tasks; with a single `gpu` the plans share its stream and run one after the other.  `generic_only`: no fused pipeline."""
        self.name = name
        self._gpu = gpu or (gpus[0] if gpus else GpuContext(device))
        self._owns_gpu = gpu is None and not gpus
        ctxs = list(gpus) if gpus else [self._gpu] * len(plans)
        if len(ctxs) != len(plans):
            raise ValueError("ExecutionContext: one GpuContext per plan")
        self.plans: List[_Plan] = [_Plan(g, p if isinstance(p, str) else json.dumps(p), generic_only) for g, p in zip(ctxs, plans)]
        self._ring = 0

    # -- hopping windows: the device-side pane ring (flockgpu_plan.h; reference: window/hopping.rs:52-74 re-sends every window whole)
    def open_window_ring(self, panes_per_window: int):
        """From now on `feed_data_sources(sources, pane=p)` feeds ONE pane (hop seconds of events); execute() runs the window of the
        last `panes_per_window` panes, clean_data_sources() retires the oldest."""
        for plan in self.plans:
            plan.ring_open(panes_per_window)
        self._ring = panes_per_window

    def close_window_ring(self):
        for plan in self.plans:
            plan.ring_close()
            plan._prefetched = None
        self._ring = 0
        self._pre = None

    # -- context.rs:257-325
    def feed_data_sources(self, sources, pane: Optional[int] = None):
        """`sources[relation][partition][batch]`.  Every plan leaf takes the first remaining source whose first
        non-empty batch matches the leaf's columns by name (compare_schema, context.rs:402-416); a leaf without
        a match stays an empty relation (context.rs:305-314).  `pane`: see open_window_ring."""
        if (pane is None) != (not self._ring):
            raise ValueError("feed_data_sources: `pane` goes with an open window ring")
        pre = getattr(self, "_pre", None)
        if sources is None:   # the pane prefetch_data_sources announced
            if pre is None or pre["pane"] != pane:
                raise ValueError("feed_data_sources(None, pane): no prefetch for this pane")
            self._pre = None   # (whatever happens below, the announcement is used up)
            touched = set()
            for plan, i, batches in pre["later"]:
                plan.feed(i, batches, pane)
                touched.add(id(plan))
            for plan, i in pre["moving"]:
                plan.feed_prefetched(i, pane)
                touched.add(id(plan))
            for plan in self.plans:
                if id(plan) not in touched and plan.inputs:
                    plan.feed(0, [], pane)
            return
        if pre is not None:
            raise ValueError("feed_data_sources: pane %d was prefetched -- feed it with sources=None first" % pre["pane"])
        sources = [list(s) for s in sources]
        for plan in self.plans:
            fed_a_pane = False
            for i in range(len(plan.inputs)):
                found = None
                for si, partitions in enumerate(sources):
                    first = next((b for part in partitions for b in part), None)
                    if first is None:
                        continue
                    if plan.matches(i, first.schema):
                        found = si
                        break
                if found is not None:
                    partitions = sources.pop(found)
                    plan.feed(i, [b for part in partitions for b in part], pane)
                    fed_a_pane = True
            if pane is not None and not fed_a_pane and plan.inputs:   # nothing arrived in this pane: the ring still advances
                plan.feed(0, [], pane)

    def prefetch_data_sources(self, sources, pane: int):
        """The sources of pane `pane` -- the ring's NEXT pane -- matched to the leaves as feed_data_sources matches them; what a prefetch
        takes (columns without NULLs, one leaf per plan) starts crossing PCIe now, beside the current window's execute, the
        rest is kept and fed the ordinary way by `feed_data_sources(None, pane)`."""
        if not self._ring:
            raise ValueError("prefetch_data_sources: open a window ring first")
        pre = {"pane": pane, "moving": [], "later": []}
        sources = [list(s) for s in sources]
        for plan in self.plans:
            started = False
            for i in range(len(plan.inputs)):
                found = None
                for si, partitions in enumerate(sources):
                    first = next((b for part in partitions for b in part), None)
                    if first is not None and plan.matches(i, first.schema):
                        found = si
                        break
                if found is None:
                    continue
                batches = [b for part in sources.pop(found) for b in part]
                if not started and plan.prefetch(i, batches, pane):
                    started = True
                    pre["moving"].append((plan, i))
                else:
                    pre["later"].append((plan, i, batches))
        self._pre = pre

    def share_data_sources(self, donor: "ExecutionContext") -> bool:
        """Instead of feed_data_sources: every leaf reads, in place, the relation of the same name that `donor` (another function
        hosted on the same GPU context, already fed and not yet cleaned) holds on the device.  All leaves or none: False when a
        leaf finds no such relation or the donor does not keep a column it reads -- feed this context its own copy then."""
        pairs = []
        for plan in self.plans:
            for i, name in enumerate(plan.inputs):
                hit = next(((dp, di) for dp in donor.plans for di, dn in enumerate(dp.inputs) if dn == name and dn), None)
                if hit is None:
                    return False
                pairs.append((plan, i, hit[0], hit[1]))
        done = []
        for plan, i, dp, di in pairs:
            if not plan.feed_shared(i, dp, di):
                for q in done:
                    q.reset()
                return False
            done.append(plan)
        return True

    def execute_retain(self):
        """execute, with every plan's result left on the device (`_Plan.execute_retain`): [rows per plan]."""
        return [plan.execute_retain() for plan in self.plans]

    def feed_from(self, producers: Sequence["ExecutionContext"]):
        """feed_data_sources from stages hosted on the same GPU context: every leaf takes the first remaining producer plan whose
        retained result holds the columns it reads (matched by name and type, as compare_schema matches batches)."""
        left = [p for ctx in producers for p in ctx.plans]
        for plan in self.plans:
            for i in range(len(plan.inputs)):
                hit = next((p for p in left if plan.feed_from(i, p)), None)
                if hit is not None:
                    left.remove(hit)

    def _concurrent(self) -> bool:   # every plan on its own GpuContext: their executes can be in flight together
        return len(self.plans) > 1 and len({id(p.gpu) for p in self.plans}) == len(self.plans)

    # -- context.rs:172-191: one task per plan, then join them all
    def _join_all(self, started, unwrap):
        """Joins EVERY started plan, then raises the first error: a plan whose execute is never waited for keeps `async_pending` and refuses
        every later feed / reset / execute -- the reference's `futures::join_all` also awaits every task before `?`."""
        out, first = [], None
        for plan in started:
            try:
                out.append(unwrap(plan.wait()))
            except Exception as e:   # noqa: BLE001 -- collected, re-raised below
                first = first or e
                out.append(None)
        if first is not None:
            raise first
        return out

    def _start_all(self, partitioned: bool):
        started = []
        try:
            for plan in self.plans:
                plan.execute_async(partitioned)
                started.append(plan)
        except Exception:
            for plan in started:   # what did start is joined before the error leaves
                try:
                    plan.wait()
                except Exception:   # noqa: BLE001
                    pass
            raise
        return started

    def execute(self):
        if self._concurrent():
            return self._join_all(self._start_all(False), lambda b: [b])
        return [[plan.execute()] for plan in self.plans]

    # -- context.rs:197-216: [plan][partition][batch]; a shuffling stage returns its P hash partitions
    def execute_partitioned(self):
        if self._concurrent():
            return self._join_all(self._start_all(True), lambda bs: [[b] for b in bs])
        return [[[b] for b in plan.execute_partitioned()] for plan in self.plans]

    # -- context.rs:227-254
    def clean_data_sources(self):
        for plan in self.plans:
            plan.reset()

    # -- context.rs:328-337
    def is_shuffling(self) -> bool:
        """Every plan ends in `CoalesceBatchesExec <- RepartitionExec Hash` (a RoundRobin repartition shuffles nothing)."""
        return bool(self.plans) and all(p.is_shuffling for p in self.plans)

    def schema(self, index: int):
        """Output schema of plan `index` (context.rs:222-224): taken from an execution over the current inputs."""
        return self.plans[index].execute().schema

    def close(self):
        for p in self.plans:
            p.close()
        if self._owns_gpu:
            self._gpu.close()


def explain(plan: Union[str, dict]) -> str:
    """Host-only: the operator tree of `plan` with derived schemas and the executor of every node (no GPU needed)."""
    raw = (plan if isinstance(plan, str) else json.dumps(plan)).encode()
    buf = C.create_string_buffer(1 << 16)
    rc = _ffi.load().flockgpu_plan_explain(raw, len(raw), buf, len(buf))
    text = buf.value.decode()
    if rc != 0:
        raise FlockGpuError(rc, text)
    return text


def partition_scheme() -> str:
    """How this library's shuffling stages place rows (flockgpu_plan_partition_scheme): a scheduler compares it across the
    producers of a stage before it starts them -- mixed placements split one key over two consumers."""
    return _ffi.load().flockgpu_plan_partition_scheme().decode()


def collect(ctx: ExecutionContext, streams, pane: Optional[int] = None):
    """`actor::collect` (flock-function/src/aws/actor.rs:54-79): feed -> execute -> clean.  With an open window ring `streams`
    is ONE pane and the result is the window that pane completes (hopping.rs:52-74 without the re-send)."""
    ctx.feed_data_sources(streams, pane)
    if ctx.is_shuffling():
        out = ctx.execute_partitioned()
        assert len(out) == 1
        out = out[0]
    else:
        out = ctx.execute()
    ctx.clean_data_sources()
    return out
