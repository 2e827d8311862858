"""A non-Python consumer of the C ABI: tests/c_abi/consumer.c is C99, includes only include/*.h, and drives
create -> feed two Arrow batches -> execute -> walk the Arrow structs -> release -> reset, then the pane ring with asynchronous executes,
the partition-scheme check and the guarded allocation -> destroy.
CPU: the headers compile as strict C99 and the consumer links against libflockgpu.so.  GPU: it runs, and its rows equal a plain loop."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "consumer.c")
EXE = os.path.join(ROOT, "tests", "c_abi", "consumer")


def _build():
    from flock_amd import build
    build.build()
    lib_dir = os.path.join(ROOT, "flock_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L" + lib_dir, "-lflockgpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return EXE


def test_headers_are_c99_and_the_consumer_links():
    _build()
    # every header on its own, as C, with nothing included before it
    for hdr in ("flockgpu.h", "flockgpu_plan.h", "flockgpu_comm.h"):
        p = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", hdr)],
                           capture_output=True, text=True)
        assert p.returncode == 0, (hdr, p.stderr)


@pytest.mark.gpu
def test_c_consumer_runs_one_collect_twice():
    exe = _build()
    plan = os.path.join(ROOT, "tests", "golden", "plans", "q2.json")
    p = subprocess.run([exe, plan, "123"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    blocks = p.stdout.split("--\n")
    tail = blocks[-1].splitlines()
    assert tail[:3] == ["ring pane 0 rows 16", "ring pane 1 rows 32", "ring pane 2 rows 32"]      # pane ring + asynchronous execute, from C
    assert tail[3] == "ring pane 3 rows 80 (prefetched)"                                          # the next pane uploaded ahead of its turn
    assert tail[-1].startswith("partition scheme flockgpu/")
    for inv in range(2):
        want = [(984 + 41 * i * (inv + 1), 7 * i + inv) for i in range(60) if (984 + 41 * i * (inv + 1)) % 123 == 0]
        got = [tuple(map(int, line.split())) for line in blocks[inv].splitlines()]
        assert got == want and len(want) == 20
