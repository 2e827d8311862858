#!/bin/bash
# gpurun helper: q5 over key distributions the NEXMark generator never produces ("Outside the generator's envelope").
python - <<'PY'
import sys, time; sys.path.insert(0, ".")
import torch, numpy as np
from flock_amd import GpuContext, Bids, WindowSchedule
ctx = GpuContext(0)
n = 100_000_000
pane = 5_000_000
offs = np.arange(0, n + 1, pane)
sched = WindowSchedule(offs, np.arange(0, len(offs) - 2, dtype=np.int32), np.arange(2, len(offs), dtype=np.int32))
g = torch.Generator(device="cuda"); g.manual_seed(1)
def run(name, keys):
    b = Bids(auction=keys, rows=n)
    for _ in range(2): r = ctx.q5_hot_items(b, sched)
    ctx.profile_reset(); ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = ctx.q5_hot_items(b, sched)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    st = ctx.profile_read(); ctx.profile(False)
    print(name, "ms/step %.3f" % (dt * 1e3), "rows/s %.3g" % (n / dt), {k: round(v["total_ms"] / 3, 3) for k, v in st.items()}, flush=True)
run("uniform 1e4", torch.randint(1000, 11_000, (n,), dtype=torch.int32, device="cuda", generator=g))
run("uniform 1e6", torch.randint(1000, 1_001_000, (n,), dtype=torch.int32, device="cuda", generator=g))
run("uniform 1e8", torch.randint(1000, 100_001_000, (n,), dtype=torch.int32, device="cuda", generator=g))
run("uniform 2e9", torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32))
z = torch.empty(n, device="cuda").exponential_(1.0, generator=g)
run("exp skew", (1000 + (z * 20000).to(torch.int32)))
run("sorted", torch.arange(n, dtype=torch.int32, device="cuda") // 13 + 1000)
PY
