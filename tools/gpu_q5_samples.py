"""Per-launch durations of q5_count_kernel over consecutive steps (HIP events on the launch stream), and the steps' wall times: is the spread a pattern?"""
import sys, time
sys.path.insert(0, ".")
import torch
from flock_amd import GpuContext, NEXMarkSource, query_window, run_query
ctx = GpuContext(0)
src = NEXMarkSource(1087, 1_000_000, query_window(5), seed=20260925)
g = src.generate_data(ctx, relations=("bid",), bid_columns=("auction",))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    run_query(ctx, 5, g)
ctx.profile_reset(); ctx.profile_only("q5_count_kernel"); ctx.profile(True)
torch.cuda.synchronize()
marks = [time.perf_counter()]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    run_query(ctx, 5, g)
    marks.append(time.perf_counter())
torch.cuda.synchronize()
ctx.profile_read()
s = ctx.profile_samples("q5_count_kernel")
print("count ms:", [round(x, 3) for x in s])
print("step wall ms:", [round((b - a) * 1e3, 3) for a, b in zip(marks[:-1], marks[1:])])
