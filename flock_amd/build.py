"""Builds libflockgpu.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this is also the CPU-side "does it build" check
(`__graft_entry__.build()`).  The library has no torch / pybind dependency: it is a plain
C-ABI shared object (include/flockgpu.h).  Every `.hip` file is compiled to its own object
(in parallel, re-done only when the file or any header changed), then linked against RCCL
(`flockgpu_comm_*`, the in-library exchange).
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# FLOCKGPU_BUILD_EXPERIMENTAL=1: a SECOND library, libflockgpu_experimental.so, compiled with -DFLOCKGPU_EXPERIMENTAL (plus whatever
# FLOCKGPU_BUILD_DEFINES lists, e.g. "-DFLOCKGPU_AB_PLAIN_TILE_LOADS"): the only build in which the A/B knobs of common.hpp's exp_env()
# read the environment.  The shipped libflockgpu.so is always built without it.
EXPERIMENTAL = os.environ.get("FLOCKGPU_BUILD_EXPERIMENTAL", "") not in ("", "0")
_TAG = ("_" + os.environ["FLOCKGPU_BUILD_TAG"]) if EXPERIMENTAL and os.environ.get("FLOCKGPU_BUILD_TAG") else ""   # several A/B variants side by side
OBJ = os.path.join(HERE, "csrc", "build_experimental" + _TAG if EXPERIMENTAL else "build")
LIB = os.path.join(HERE, "libflockgpu_experimental" + _TAG + ".so" if EXPERIMENTAL else "libflockgpu.so")
STAMP = os.path.join(HERE, "csrc", ".build_stamp_experimental" + _TAG if EXPERIMENTAL else ".build_stamp")
CFLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",            # q1's f64 multiply must stay a plain IEEE multiply
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
]
if EXPERIMENTAL:
    CFLAGS += ["-DFLOCKGPU_EXPERIMENTAL"] + os.environ.get("FLOCKGPU_BUILD_DEFINES", "").split()
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"]
LIBS = ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    inc = os.path.join(HERE, "..", "include")
    return sorted(glob.glob(os.path.join(CSRC, "*.h*")) + glob.glob(os.path.join(inc, "*.h")))


def _sha(paths, extra=""):
    h = hashlib.sha256()
    for f in paths:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(extra.encode())
    return h.hexdigest()


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libflockgpu.so cannot be built")


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = hipcc_path()
    os.makedirs(OBJ, exist_ok=True)
    hdr = _sha(_headers(), " ".join(CFLAGS))
    jobs, objs = [], []
    for src in _sources():
        base = os.path.splitext(os.path.basename(src))[0]
        obj, stamp = os.path.join(OBJ, base + ".o"), os.path.join(OBJ, base + ".stamp")
        want = _sha([src], hdr)
        objs.append(obj)
        if force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != want:
            jobs.append((src, obj, stamp, want))

    def compile_one(job):
        src, obj, stamp, want = job
        cmd = [hipcc] + CFLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{p.stderr}")
        if p.stderr.strip():
            print(p.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(want)

    if jobs:
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 4))) as pool:
            list(pool.map(compile_one, jobs))
    link = _sha([], hdr + "".join(open(os.path.join(OBJ, os.path.splitext(os.path.basename(s))[0] + ".stamp")).read()
                                   for s in _sources()) + " ".join(LDFLAGS + LIBS))
    if jobs or force or not os.path.exists(LIB) or not os.path.exists(STAMP) or open(STAMP).read() != link:
        cmd = [hipcc] + LDFLAGS + objs + LIBS + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(STAMP, "w") as f:
            f.write(link)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
