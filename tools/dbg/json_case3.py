import sys, json; sys.path.insert(0, ".")
import numpy as np, torch
from flock_amd import GpuContext
idx = int(sys.argv[1]); asc = bool(int(sys.argv[2])); N = int(sys.argv[3])
ctx = GpuContext(0)
def dev(b):
    t = torch.zeros(len(b) + 16, dtype=torch.uint8, device="cuda"); t[:len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda(); return t[:len(b)]
specials = ['', 'plain', 'quote " inside', 'back\\slash', 'tab\there', 'nl\nnl', 'uni é ü ß', 'cjk 漢字', 'emoji \U0001F600 end',
            '/slash/', '\b\f\r', 'x' * 300, 'ctl \x01\x1f']
sp = specials if idx < 0 else [specials[idx]]
lines = [json.dumps({"k": i, "s": sp[i % len(sp)]}, ensure_ascii=asc).encode() for i in range(N)]
text = b"\n".join(lines) + b"\n"
got, n = ctx.json_lines_decode(dev(text), [("k", "int32"), ("s", "utf8")])
torch.cuda.synchronize()
print(idx, asc, N, "ok", n)
