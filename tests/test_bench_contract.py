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
    assert a.gpus == 1 and 1 <= a.steps <= 200 and 0 <= a.warmup <= 100 and a.query == 5 and a.mode == "auto"      # auto: window-sharded headline at every N; at N > 1 the key-partitioned exchange rides along as `exchange`
    assert bench.DEFAULT_SECONDS[5] * a.eps * 46 // 50 >= 1_000_000_000          # the headline config: 1e9 bids


def test_every_kernel_the_bench_names_exists():
    import bench
    # (LaunchScope labels of q3_probe_general_kernel<false> and json_parse_kernel<n, retry = true>)
    kernels = _kernel_names() | {"q3_probe_count_kernel", "json_parse_retry_kernel"}
    for q, (name, bytes_per_row, relation) in bench.DOMINANT.items():
        for one in name.split("|"):     # "a|b": the kernels a step chooses between by batch size
            assert one in kernels, (q, one)
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


def _fat_out():
    """A measurement record shaped like a full default run, with the prose that made round 2's line 22 KB."""
    roof = {"bound": "hbm", "kernel": "q5_count_kernel", "achieved": 5311.3, "peak": 8000.0, "unit": "GB/s", "frac": 0.6639, "traffic": 4353122634,
            "traffic_source": "x" * 120, "avg_launch_ms": 0.7531, "algorithmic_bytes_per_launch": 4000160000, "launches": 20,
            "kernels_ms": {f"kernel_number_{i}_kernel": 0.01 * i for i in range(30)}}
    cpu = {"value": 105387299.4, "unit": "rows/s", "cores": 64, "kind": "port", "sample": "s" * 400, "seconds": 20.1,
           "pass_seconds": {"mean": 2.9, "min": 2.8, "max": 3.0}, "acero": {"value": 1.4e8, "unit": "rows/s", "cores": 256, "kind": "port", "sample": "a" * 300}}
    entry = {"value": 6.88e11, "unit": "rows/s", "ms_per_step": 0.146, "input_rows": 100280000, "windows": 109, "result_rows": 815280,
             "roofline": dict(roof), "cpu_baseline": dict(cpu), "note": "n" * 300}
    also = {f"entry_{i}_next": dict(entry) for i in range(24)}
    also["broken"] = {"error": "RuntimeError('boom')"}
    also["exchange_1rank"] = {"q5": dict(entry, over_window_sharded_step=1.6), "q3": dict(entry), "q8": {"error": "x"}}
    return {"metric": "NEXMark rows/sec per node (q3 join, q5 agg)", "value": 9.8e11, "unit": "rows/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 1.017, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "NEXMark q5 hopping(10,5) over 1087 s x 1000000 events/s per GPU", "query": "q5", "input_rows_per_gpu": 1000040000,
                       "windows_per_gpu": 216, "parallelism": "window-sharded x1 (no data-path collective)", "result_rows": 229},
            "roofline": roof, "cpu_baseline": cpu, "q3": dict(entry), "also": also, "also_file": "gpurun_out/bench_also.json"}


def test_last_line_is_short_and_carries_roofline_and_cpu_baseline():
    import json
    import bench
    line = bench.final_line(_fat_out())
    assert "\n" not in line and len(line) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["config"]["workload"].startswith("NEXMark q5") and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["q3"]["roofline"]["frac"] <= 1 and d["also"]["entry_0_next"][0] > 0 and d["also"]["broken"] == "error"
    # an absurdly large record still ends in a parseable line below the limit (side entries are dropped first)
    fat = _fat_out()
    fat["also"].update({f"more_{i}": dict(fat["q3"]) for i in range(400)})
    line = bench.final_line(fat)
    assert len(line) < 4096 and json.loads(line)["roofline"]["frac"] > 0


def test_n_ranks_line_reports_the_configured_total_and_carries_exchange_and_weak():
    """N > 1: the headline is BASELINE.json configs[3] -- 1e9 bids IN TOTAL over the N GPUs, window-sharded, "strong"; the
    key-partitioned exchange of north_star is its own object with "strong" and the phase timeline -- or an `exchange_error` string; the
    N x 1e9-bid job of earlier rounds rides along as `weak`."""
    import json
    import bench
    out = _fat_out()
    out["n_gpus"] = 8
    out["scaling"] = "strong"
    out["config"] = {"workload": "NEXMark q5 hopping(10,5) over 1087 s x 1000000 events/s: 1000040000 input rows in total over 8 GPUs (BASELINE.json configs[3])",
                     "query": "q5", "input_rows_total": 1000040000, "windows_total": 216, "windows_this_rank": 27,
                     "parallelism": "window-sharded x8: contiguous runs of windows per GPU, no data-path collective"}
    out["weak"] = {"value": 8.1e12, "ms_per_step": 0.98, "scaling": "weak", "workload": "NEXMark q5 hopping(10,5) over 1087 s x 1000000 events/s per GPU", "roofline_frac": 0.66}
    out["exchange"] = {"value": 3.1e12, "unit": "rows/s", "scaling": "strong", "ms_per_step": 0.32, "input_rows_all_gpus": 998200000, "ranks": 8, "transport": "rccl",
                       "phases_ms": {"partial": 0.11, "partition+take": 0.08, "counts": 0.01, "all_to_all+regroup": 0.05, "final": 0.07}, "roofline": {"frac": 0.55},
                       "kernels_ms_rank0": {f"k{i}": 0.1 for i in range(40)}, "workload": "w" * 200}
    d = json.loads(bench.final_line(out))
    assert d["scaling"] == "strong" and d["n_gpus"] == 8 and d["roofline"]["frac"] <= 1 and "in total over 8 GPUs" in d["config"]["workload"]
    assert d["weak"]["scaling"] == "weak" and d["weak"]["value"] == 8.1e12
    assert d["exchange"]["scaling"] == "strong" and d["exchange"]["value"] == 3.1e12 and d["exchange"]["phases_ms"]["final"] == 0.07
    assert "kernels_ms_rank0" not in d["exchange"] and len(bench.final_line(out)) < 4096
    # round 6: the join configs ride on the N > 1 line window-sharded and "strong", each with its own roofline and cpu_baseline; the collective
    # that ran is named at the top level
    for k, cfg in (("q3", 2), ("q8", 4)):
        out[k] = {"value": 2.2e12, "unit": "rows/s", "n_gpus": 8, "ms_per_step": 0.04, "scaling": "strong", "roofline": dict(out["roofline"], frac=0.51),
                  "cpu_baseline": dict(out["cpu_baseline"]), "config": {"workload": f"NEXMark q{k[1]} ... 80000000 input rows in total over 8 GPUs (BASELINE.json configs[{cfg}])"}}
    out["collective"] = {"library": "RCCL (ncclSend / ncclRecv groups inside libflockgpu)", "ranks": 8, "transport": "rccl"}
    d = json.loads(bench.final_line(out))
    assert d["cpu_baseline"]["value"] > 0 and d["collective"]["ranks"] == 8 and d["collective"]["transport"] == "rccl"
    for k, cfg in (("q3", "configs[2]"), ("q8", "configs[4]")):
        assert d[k]["scaling"] == "strong" and d[k]["n_gpus"] == 8 and d[k]["value"] == 2.2e12 and cfg in d[k]["workload"]
        assert d[k]["roofline"]["frac"] == 0.51 and d[k]["cpu_baseline"]["value"] > 0
    assert len(bench.final_line(out)) < 4096
    out.pop("exchange")
    out["exchange_error"] = "RuntimeError('ncclCommInitRank: unhandled system error')" + "x" * 1000
    d = json.loads(bench.final_line(out))
    assert d["exchange_error"].startswith("RuntimeError") and len(d["exchange_error"]) <= 300 and d["value"] == 9.8e11


def test_windows_are_dealt_to_the_ranks_in_contiguous_runs():
    """`bench.window_shard`: the 216 Hopping(10, 5) windows of 1087 s over 1 / 2 / 4 / 8 / 300 ranks -- every window exactly once, each rank's
    slice of seconds holding exactly its windows (q5), and likewise q8's tumbling and q3's element-wise windows."""
    import bench
    from flock_amd import query_window
    from flock_amd.nexmark import window_epochs
    for q, seconds in ((5, 1087), (8, 1000), (3, 100), (5, 12)):
        wins = window_epochs(query_window(q), seconds)
        for world in (1, 2, 4, 8, 300):
            got = []
            for rank in range(world):
                first, secs, n = bench.window_shard(q, seconds, rank, world)
                local = window_epochs(query_window(q), secs) if secs else []
                assert len(local) == n
                got += [(a + first, b + first) for a, b in local]
            assert got == wins, (q, seconds, world)


def test_exchange_mode_bills_the_kernels_that_read_the_raw_rows():
    """Round 2 charged the raw column to kernels that run on filtered / pre-aggregated rows (fractions above 1)."""
    import bench
    kernels = _kernel_names()
    assert bench.DOMINANT_EXCHANGE[5][0] == "q5_partial_tile_kernel" and bench.DOMINANT_EXCHANGE[3][0] == "q3_category_flag_kernel"
    for q, (name, bpr, rel) in bench.DOMINANT_EXCHANGE.items():
        assert name in kernels and name != bench.DOMINANT[q][0] and bpr == 4.0
    stats = {"q5_partial_tile_kernel": {"launches": 10, "total_ms": 9.0}, "q5_count_kernel": {"launches": 10, "total_ms": 3.8}}
    r = bench.roofline(5, stats, {"bid": 1_000_040_000, "auction": 0}, bench.DOMINANT_EXCHANGE)
    assert r["kernel"] == "q5_partial_tile_kernel" and 0.5 < r["frac"] < 0.6


def test_gpus_n_without_a_launcher_spawns_n_ranks(monkeypatch):
    import bench

    class FakeProc:
        started = []

        def __init__(self, cmd, env=None, stdout=None):
            self.cmd, self.env, self.stdout = cmd, env, stdout
            FakeProc.started.append(self)

        def poll(self):
            return 0

        def wait(self):
            return 0

        def kill(self):
            pass

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse()
    # fewer devices than ranks: refuse, start nothing
    assert bench.spawn_ranks(args, sys.argv[1:], n_devices=1, popen=FakeProc) != 0 and not FakeProc.started
    assert bench.spawn_ranks(args, sys.argv[1:], n_devices=2, popen=FakeProc) == 0
    assert [p.env["RANK"] for p in FakeProc.started] == ["0", "1"] and all(p.env["WORLD_SIZE"] == "2" for p in FakeProc.started)
    assert all(p.env["MASTER_ADDR"] == "127.0.0.1" and p.env["LOCAL_RANK"] == p.env["RANK"] for p in FakeProc.started)
    assert len({p.env["MASTER_PORT"] for p in FakeProc.started}) == 1
    assert FakeProc.started[0].stdout is None and FakeProc.started[1].stdout is sys.stderr      # only rank 0 owns stdout
    assert FakeProc.started[0].cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    # main() takes that branch (and exits with the spawn's status) before touching torch / the GPU
    called = {}
    monkeypatch.setattr(bench, "spawn_ranks", lambda a, argv: called.setdefault("rc", 7))
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7
    assert called == {"rc": 7}


def test_a_dead_rank_fails_the_job(monkeypatch):
    import bench

    class Proc:
        def __init__(self, cmd, env=None, stdout=None):
            self.rank, self.killed = int(env["RANK"]), False

        def poll(self):
            return 3 if self.rank == 1 else (-9 if self.killed else None)

        def wait(self):
            return 0

        def kill(self):
            self.killed = True

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setattr(bench.time, "sleep", lambda s: None)
    assert bench.spawn_ranks(bench.parse(), ["--gpus", "2"], n_devices=2, popen=Proc) == 3


@__import__("pytest").mark.gpu
def test_bench_prints_one_short_line_on_the_gpu(tmp_path):
    import json
    import subprocess
    env = dict(os.environ, FLOCK_BENCH_ALSO=str(tmp_path / "also.json"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--seconds", "60", "--steps", "3", "--warmup", "1", "--no-also"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last) < 4096
    d = json.loads(last)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["kernel"] == "q5_count_kernel"
    assert d["cpu_baseline"] is None or d["cpu_baseline"]["value"] > 0
    assert json.load(open(tmp_path / "also.json"))["value"] == d["value"]
    # --gpus beyond the visible devices: refused, non-zero, no JSON on stdout
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-also", "--no-cpu"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in env.items() if k != "WORLD_SIZE"})
    assert p.returncode != 0 and "{" not in p.stdout


@__import__("pytest").mark.gpu
def test_two_ranks_on_one_gpu_run_the_configured_workload_and_the_exchange(tmp_path):
    """`--gpus 2` with both ranks on the one visible device (FLOCK_BENCH_SHARED_GPU, gloo for the barrier): RCCL refuses two ranks on one
    device, so the exchange cannot start -- the line must still be the window-sharded headline of two ranks (the configured total, "strong"), with `exchange_error`
    (or, should a transport accept it, an `exchange` object)."""
    import json
    import subprocess
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}, FLOCK_BENCH_ALSO=str(tmp_path / "also.json"),
               FLOCK_BENCH_SHARED_GPU="1", FLOCK_BENCH_EXCHANGE_TIMEOUT="90", FLOCK_BENCH_SPAWN_TIMEOUT="500", FLOCK_BENCH_STRONG_JOINS="20,40")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--seconds", "60", "--steps", "2", "--warmup", "1", "--no-also", "--cpu-threads", "4"],
                       capture_output=True, text=True, timeout=700, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    last = [l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1]
    assert len(last) < 4096
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["parallelism"].startswith("window-sharded x2")
    # the configured workload: ONE stream of 60 s in total, its windows dealt to the two ranks; the two-slices job rides along as `weak`
    assert "in total over 2 GPUs" in d["config"]["workload"] and d["config"]["windows_total"] == 11 and d["config"]["input_rows_total"] == 60 * 1_000_000 // 50 * 46
    assert d["weak"]["scaling"] == "weak" and d["weak"]["value"] > 0
    # a SCALE line is judged like the N = 1 line: the CPU baseline (rank 0's host cores, bounded sample) and the roofline ride on it, and the
    # join configs the metric names (q3: configs[2], q8: configs[4]) are there window-sharded and "strong", each with its own pair
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["roofline"]["frac"] > 0
    for k, cfg in (("q3", "configs[2]"), ("q8", "configs[4]")):
        assert d[k]["scaling"] == "strong" and d[k]["value"] > 0 and d[k]["n_gpus"] == 2 and cfg in d[k]["workload"], d[k]
        assert d[k]["cpu_baseline"]["value"] > 0 and d[k]["roofline"]["frac"] > 0, d[k]
    assert d["collective"]["ranks"] == 2 and d["collective"]["transport"] == "ipc"
    # two processes on one device: RCCL refuses that, the ipc transport (flockgpu_comm_init_ipc) carries the same exchange end to end
    assert "exchange_error" not in d, d.get("exchange_error")
    assert d["exchange"]["scaling"] == "strong" and d["exchange"]["value"] > 0 and d["exchange"]["transport"] == "ipc" and d["exchange"]["ranks"] == 2
    assert d["exchange_ok"] is True
