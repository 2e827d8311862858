"""Over-reads do not show with ordinary allocations: a kernel that reads a few KB past the end of a column lands in whatever the
allocator put next to it (q3's build pass and q8's persons pass both did, for the id in front of a ragged last tile's chunks; the
second was found by accident when the pane ring's buffers became exactly sized).  Here the direct-call GPU tests -- the ones that
hand the C ABI hand-made columns: ragged windows, tiny and empty inputs, duplicates, hostile strings -- run once more in a process of
their own with EVERY such input column in memory that ends where mapped address space ends (flockgpu_malloc_guarded via
tests/devmem.py): any access past a column's end is a GPU memory fault, i.e. a failed run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (test file, -k expression): the tests whose inputs go through the files' `_dev` helpers, and the generator-based ones (their streams are
# copied into guarded memory: tests/devmem.py guard_stream); the two full-size property tests keep their 1e9-row streams where they are
RUNS = [
    ("tests/test_gpu_parity.py", "not full_size and not baseline_sizes"),
    ("tests/test_gpu_ysb.py", ""),
    ("tests/test_gpu_q11.py", ""),
    ("tests/test_gpu_json.py", ""),
]


@pytest.mark.gpu
@pytest.mark.parametrize("path,select", RUNS, ids=[r[0].split("/")[-1] for r in RUNS])
def test_direct_call_tests_with_inputs_in_guarded_memory(path, select):
    if os.environ.get("FLOCK_TEST_GUARDED"):
        pytest.skip("already the guarded run")
    cmd = [sys.executable, "-m", "pytest", path, "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + (["-k", select] if select else [])
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, FLOCK_TEST_GUARDED="1"), capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert " passed" in p.stdout and "no tests ran" not in p.stdout, tail


@pytest.mark.gpu
def test_guarded_buffers_hold_their_data_and_are_aligned():
    import numpy as np
    from flock_amd import GpuContext
    c = GpuContext(0)
    for n in (1, 3, 4, 1000, 1 << 19, (1 << 19) + 1):
        host = np.arange(n, dtype=np.int32)
        g = c.guarded(host)
        assert g.data_ptr() % 16 == 0 and g.numel() == n
        assert np.array_equal(c.d2h(g.data_ptr(), n, np.int32), host)
    c.close()


SELF_CHECK = """
import ctypes as C, numpy as np, torch
from flock_amd import GpuContext, _ffi
c = GpuContext(0)
g = c.guarded(np.arange(1000, dtype=np.int32))
rows = torch.tensor([0, 999, 999 + %d], dtype=torch.int32, device="cuda")
out = torch.empty(3, dtype=torch.int32, device="cuda")
rc = _ffi.load().flockgpu_take_i32(c._h, C.c_void_p(g.data_ptr()), C.c_void_p(rows.data_ptr()), 3, C.c_void_p(out.data_ptr()))
torch.cuda.synchronize(); c.synchronize()
print("read", out.cpu().tolist(), rc)
"""


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("FLOCK_TEST_GUARD_SELFCHECK"), reason="provokes a GPU memory fault on purpose: run by hand (FLOCK_TEST_GUARD_SELFCHECK=1)")
def test_an_over_read_of_a_guarded_buffer_faults():
    """The aid itself, opt-in because it kills a process with a GPU memory fault: a take() whose last row number lies 5000 elements
    past a guarded 1000-element column dies; the same take inside the column returns.  (Run on an MI355X box, round 4: the in-bounds
    variant prints `read [0, 999, 999] 0`, the over-read aborts the child with a memory access fault.)"""
    ok = subprocess.run([sys.executable, "-c", SELF_CHECK % 0], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0 and "read [0, 999, 999] 0" in ok.stdout, ok.stdout + ok.stderr
    bad = subprocess.run([sys.executable, "-c", SELF_CHECK % 5000], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0, bad.stdout + bad.stderr
