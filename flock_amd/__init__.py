"""flock_amd -- MI355X (gfx950) execution kernel for Flock's NEXMark hot path.

Hand-written HIP kernels behind the C ABI of include/flockgpu.h; this package is the thin host
side that mirrors the reference's source / window-launcher / `collect` surface for that path.
No CPU fallback exists: without libflockgpu.so (built by `python -m flock_amd.build`) every
operator call raises.
"""
from ._ffi import FlockGpuError, LIB_PATH, load  # noqa: F401
from .engine import (Auctions, Bids, Comm, DeviceUtf8, GpuContext, Persons, WindowSchedule)  # noqa: F401
from .nexmark import (NEXMarkSource, NEXMarkStream, Window, query_window, run_query, run_query_async, synthetic_side_input,  # noqa: F401
                      window_epochs)
from .session import SessionWindows, launch_session_query  # noqa: F401

__all__ = ["FlockGpuError", "GpuContext", "Bids", "Auctions", "Persons", "DeviceUtf8", "WindowSchedule", "Comm",
           "NEXMarkSource", "NEXMarkStream", "Window", "query_window", "run_query", "run_query_async", "window_epochs", "SessionWindows", "launch_session_query",
           "load", "LIB_PATH"]
