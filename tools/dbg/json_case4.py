import sys, json; sys.path.insert(0, ".")
import numpy as np, torch
from flock_amd import GpuContext
s = eval(sys.argv[1]); N = int(sys.argv[2])
ctx = GpuContext(0)
def dev(b):
    t = torch.zeros(len(b) + 16, dtype=torch.uint8, device="cuda"); t[:len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda(); return t[:len(b)]
lines = [json.dumps({"k": i, "s": s}).encode() for i in range(N)]
text = b"\n".join(lines) + b"\n"
got, n = ctx.json_lines_decode(dev(text), [("k", "int32"), ("s", "utf8")])
torch.cuda.synchronize()
off = got["s"].offsets.cpu().numpy(); d = got["s"].data.cpu().numpy()[:off[-1]].tobytes()
print(repr(s), N, "ok", n, off[:4].tolist(), d[:12])
