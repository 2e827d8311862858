"""Builds libflockgpu.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this is also the CPU-side "does it build" check
(`__graft_entry__.build()`).  The library has no torch / pybind dependency: it is a plain
C-ABI shared object (include/flockgpu.h).
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libflockgpu.so")
STAMP = os.path.join(HERE, "csrc", ".build_stamp")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",            # q1's f64 multiply must stay a plain IEEE multiply
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.h*")) + [os.path.join(HERE, "..", "include", "flockgpu.h"),
                                                             os.path.join(HERE, "..", "include", "flockgpu_plan.h")]):
        if os.path.exists(f):
            h.update(f.encode())
            h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libflockgpu.so cannot be built")


def build(force: bool = False, verbose: bool = False) -> str:
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == digest:
        return LIB
    cmd = [hipcc_path()] + FLAGS + _sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
