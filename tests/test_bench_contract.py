"""Static checks of bench.py's contract with the driver and with the kernels it names (no GPU needed)."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _kernel_names():
    names = set()
    for f in glob.glob(os.path.join(ROOT, "flock_amd", "csrc", "*.hip")):
        names |= set(re.findall(r"__global__[^;{]*?void\s+(\w+)\s*\(", open(f).read()))
    return names


def test_defaults_are_one_gpu_and_a_few_steps(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and 1 <= a.steps <= 20 and 0 <= a.warmup <= 10 and a.query == 5 and a.mode == "auto"      # auto: window-sharded at N = 1, key-partitioned exchange at N > 1
    assert bench.DEFAULT_SECONDS[5] * a.eps * 46 // 50 >= 1_000_000_000          # the headline config: 1e9 bids


def test_every_kernel_the_bench_names_exists():
    import bench
    kernels = _kernel_names()
    for q, (name, bytes_per_row, relation) in bench.DOMINANT.items():
        assert name in kernels, (q, name)
        assert bytes_per_row > 0 and relation in ("bid", "auction")
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name in re.findall(r'"(\w+_kernel)"', src):
        assert name in kernels, name


def test_launch_scopes_name_real_kernels():
    """Every LaunchScope label is a kernel of the same file set (profiles and `roofline.kernels_ms` are keyed by them)."""
    kernels = _kernel_names() | {"q5_max_kernel", "q5_select_kernel", "q3_probe_count_kernel", "q3_probe_emit_kernel"}  # template aliases
    for f in glob.glob(os.path.join(ROOT, "flock_amd", "csrc", "*.hip")):
        for label in re.findall(r'LaunchScope\s+ls\(ctx,\s*"(\w+)"\)', open(f).read()):
            assert label in kernels, (os.path.basename(f), label)
