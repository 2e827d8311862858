"""Yahoo Streaming Benchmark source + query on the device.

Mirrors `YSBSource::new(seconds, threads, events_per_second, window)` / `generate_data`
(flock/src/datasource/ysb/ysb.rs:233-305) and the benchmark's fixed window, Tumbling(10 s)
(benchmarks/src/ysb/main.rs:91).  Every event is an ad event, so epoch e holds rows [e * eps, (e + 1) * eps).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from .engine import DeviceUtf8, GpuContext, WindowSchedule
from .nexmark import Window, window_epochs

EVENT_TYPES = ("view", "click", "purchase")   # generator.rs:71-74


@dataclass
class YSBStream:
    source: "YSBSource"
    ad_id: DeviceUtf8
    event_type: DeviceUtf8
    c_ad_id: DeviceUtf8
    campaign_id: DeviceUtf8
    rows: int
    campaign_rows: int

    def window_schedule(self, window: Window = None) -> WindowSchedule:
        s = self.source
        window = window or s.window
        epochs = window_epochs(window, s.seconds)
        n_panes = s.seconds // window.size
        off = np.arange(n_panes + 1, dtype=np.int64) * window.size * s.eps
        return WindowSchedule(off, np.array([a // window.size for a, _ in epochs], np.int32),
                              np.array([b // window.size for _, b in epochs], np.int32))


class YSBSource:
    def __init__(self, seconds: int, events_per_second: int, window: Window = None, campaigns: int = 100, ads: int = 10,
                 seed: int = 0, first_event: int = 0):
        self.seconds, self.eps = seconds, events_per_second
        self.window = window or Window.tumbling(10)
        self.campaigns, self.ads, self.seed, self.first_event = campaigns, ads, seed, first_event

    def generate_data(self, ctx: GpuContext) -> YSBStream:
        import torch
        dev = f"cuda:{ctx.device}"
        lib = _ffi.load()
        n, nc = self.seconds * self.eps, self.campaigns * self.ads
        i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)
        u8 = lambda k: torch.empty(max(k, 16), dtype=torch.uint8, device=dev)
        c_ad, camp = DeviceUtf8(i32(nc + 1), u8(nc * 36)), DeviceUtf8(i32(nc + 1), u8(nc * 36))
        ctx._check(lib.flockgpu_ysb_gen_campaigns(ctx._h, self.seed, self.campaigns, self.ads, c_ad.offsets.data_ptr(),
                                                  c_ad.data.data_ptr(), camp.offsets.data_ptr(), camp.data.data_ptr()))
        ad, et = DeviceUtf8(i32(n + 1), u8(n * 36)), DeviceUtf8(i32(n + 1), u8(n * 8))
        ctx._check(lib.flockgpu_ysb_gen_events(ctx._h, self.seed, self.first_event, n, nc, ad.offsets.data_ptr(), ad.data.data_ptr(),
                                               et.offsets.data_ptr(), et.data.data_ptr()))
        ctx.synchronize()
        return YSBStream(self, ad, et, c_ad, camp, n, nc)


@dataclass
class YsbOut:
    ctx: GpuContext
    raw: _ffi.YsbResult
    n_windows: int

    @property
    def rows(self):
        return int(self.raw.rows)

    def offsets(self):
        return np.ctypeslib.as_array(self.raw.win_out_offsets, (self.n_windows + 1,)).copy()

    def to_host(self):
        n = self.rows
        return {"campaign_id": (self.ctx.d2h(self.raw.campaign_id.offsets, n + 1, np.int32),
                                self.ctx.d2h(self.raw.campaign_id.data, int(self.raw.campaign_bytes), np.uint8)),
                "count": self.ctx.d2h(self.raw.count, n, np.uint64), "offsets": self.offsets()}


def campaign_counts(ctx: GpuContext, ad_id: DeviceUtf8, event_type: DeviceUtf8, rows: int, windows: WindowSchedule,
                    c_ad_id: DeviceUtf8, campaign_id: DeviceUtf8, campaign_rows: int, event_type_lit: str = "view") -> YsbOut:
    """ysb.sql through `flockgpu_ysb_campaign_counts`."""
    lib = _ffi.load()
    ev = _ffi.YsbEventCols(ad_id.ffi(), event_type.ffi(), rows)
    ca = _ffi.YsbCampaignCols(c_ad_id.ffi(), campaign_id.ffi(), campaign_rows)
    w, r = windows.ffi(), _ffi.YsbResult()
    ctx._check(lib.flockgpu_ysb_campaign_counts(ctx._h, C.byref(ev), C.byref(w), C.byref(ca), event_type_lit.encode(), C.byref(r)))
    return YsbOut(ctx, r, windows.n_windows)


def run_ysb(ctx: GpuContext, stream: YSBStream, window: Window = None) -> YsbOut:
    return campaign_counts(ctx, stream.ad_id, stream.event_type, stream.rows, stream.window_schedule(window), stream.c_ad_id,
                           stream.campaign_id, stream.campaign_rows)
