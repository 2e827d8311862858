import sys, json; sys.path.insert(0, ".")
import numpy as np, torch
from flock_amd import GpuContext
case = sys.argv[1]
ctx = GpuContext(0)
def dev(b):
    t = torch.zeros(len(b) + 16, dtype=torch.uint8, device="cuda"); t[:len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda(); return t[:len(b)]
rng = np.random.default_rng(5)
fields = [("k", "int32"), ("t", "int64"), ("s", "utf8"), ("u", "utf8")]
specials = ['', 'plain', 'quote " inside', 'back\\slash', 'tab\there', 'nl\nnl', 'uni é ü ß', 'cjk 漢字', 'emoji \U0001F600 end',
            '/slash/', '\b\f\r', 'x' * 300, 'ctl \x01\x1f']
lines = []
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
for i in range(N):
    o = {"k": int(rng.integers(-2**31, 2**31)), "t": int(rng.integers(-2**62, 2**62)), "s": specials[i % len(specials)], "u": "row%d" % i}
    extra = {"z": [1, {"a": "}]\\\""}, [None, True, 1.5e3]], "y": {"n": {"m": "\n"}}, "w": -0.25, "v": None, "k2": "k"}
    items = list(o.items()) + ([(k, extra[k]) for k in list(extra)[: i % 6]] if "x" in case else [])
    order = rng.permutation(len(items)) if "p" in case else np.arange(len(items))
    body = (", " if i % 3 else ",").join("%s%s:%s%s" % (json.dumps(items[j][0]), " " * (i % 2), "\t" * (i % 4 == 1),
                                                             json.dumps(items[j][1], ensure_ascii=bool(i % 2))) for j in order)
    lines.append((" " * (i % 3) + "{" + " " * (i % 2) + body + "}" + ("\r" if i % 5 == 0 else "")).encode())
text = b"\n".join(lines) + b"\n"
if "1" in case: fields = fields[:2]
if "2" in case: fields = fields[2:]
if "3" in case: fields = [fields[0], fields[3]]
if "4" in case: fields = [fields[0], fields[2]]
got, n = ctx.json_lines_decode(dev(text), fields)
torch.cuda.synchronize()
print(case, "ok", n)
