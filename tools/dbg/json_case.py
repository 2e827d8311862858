import sys, json; sys.path.insert(0, ".")
import numpy as np, torch
from flock_amd import GpuContext
case = sys.argv[1]
ctx = GpuContext(0)
fields = [("k", "int32"), ("s", "utf8")]
def dev(b):
    t = torch.zeros(len(b) + 16, dtype=torch.uint8, device="cuda"); t[:len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda(); return t[:len(b)]
lines = []
for i in range(2000):
    if case == "A": lines.append(b' { "s" : "v%d" , "k":%d }' % (i, i))
    elif case == "B": lines.append(b'{"z":[1,{"a":"}]"},[null,true,1.5e3]],"k":%d,"y":{"n":{"m":"q"}},"s":"v%d","w":-0.25}' % (i, i))
    elif case == "C": lines.append(json.dumps({"k": i, "s": ["a\"b", "t\tt", "é", "\U0001F600", "n\nn"][i % 5]}).encode())
    elif case == "D": lines.append(json.dumps({"k": i, "s": "x" * 300}).encode())
    elif case == "E": lines.append(json.dumps({"k": i, "s": ["", "plain"][i % 2]}, ensure_ascii=False).encode() + b"\r")
text = b"\n".join(lines) + b"\n"
got, n = ctx.json_lines_decode(dev(text), fields)
torch.cuda.synchronize()
print(case, "ok", n, got["k"][:3].tolist(), got["s"].offsets[:4].tolist())
