"""The plan parser against damaged input: every fixture under tests/golden/plans truncated, with bytes overwritten, with random sub-trees replaced
by values of the wrong kind and with members swapped -- `flockgpu_plan_explain` must answer every one of them with a status (an operator tree, or
FLOCKGPU_ERR_PLAN / FLOCKGPU_ERR_UNSUPPORTED and a message), never with a crash: the JSON arrives from another process (the reference ships plans
between cloud functions as serde_json strings, flock/src/runtime/plan.rs)."""
import ctypes as C
import json
import os
import random

import pytest

PLANS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plans")
JUNK = [None, 0, -1, 2**40, "x", [], {}, [[]], {"physical_expr": "column"}, {"execution_plan": "filter_exec"}, True, 1.5, {"physical_expr": "case_expr"},
        {"execution_plan": "window_agg_exec", "window_expr": [{}]}]


def _damage(txt, rnd):
    mode = rnd.random()
    if mode < 0.3:
        return txt[: rnd.randrange(len(txt))]
    if mode < 0.55:
        b = bytearray(txt.encode())
        for _ in range(rnd.randint(1, 6)):
            b[rnd.randrange(len(b))] = rnd.choice(b'{}[]",:0123456789-ntfxe \n')
        return b.decode("latin1")
    j = json.loads(txt)
    slots = []

    def walk(o):
        if isinstance(o, dict):
            for k, v in o.items():
                slots.append((o, k))
                walk(v)
        elif isinstance(o, list):
            for i, v in enumerate(o):
                slots.append((o, i))
                walk(v)
    walk(j)
    if mode < 0.85:
        for _ in range(rnd.randint(1, 3)):
            o, k = rnd.choice(slots)
            o[k] = rnd.choice(JUNK)
    else:
        for o, k in rnd.sample(slots, min(len(slots), 8)):
            if isinstance(o, dict) and len(o) >= 2:
                a, b = rnd.sample(list(o), 2)
                o[a], o[b] = o[b], o[a]
    return json.dumps(j)


@pytest.mark.parametrize("seed", range(4))
def test_damaged_plans_are_answered_with_a_status(seed):
    from flock_amd import _ffi
    lib = _ffi.load()
    lib.flockgpu_plan_explain.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.flockgpu_plan_explain.restype = C.c_int
    rnd = random.Random(seed)
    names = sorted(os.listdir(PLANS))
    buf = C.create_string_buffer(1 << 16)
    parsed = refused = 0
    for _ in range(600):
        raw = _damage(open(os.path.join(PLANS, rnd.choice(names))).read(), rnd).encode("latin1", "ignore")
        rc = lib.flockgpu_plan_explain(raw, len(raw), buf, len(buf))
        assert rc in (_ffi.OK, _ffi.ERR_PLAN, _ffi.ERR_UNSUPPORTED, _ffi.ERR_INVALID), rc
        parsed += rc == _ffi.OK
        refused += rc != _ffi.OK
    assert parsed > 20 and refused > 300, (parsed, refused)   # (swapped members and harmless overwrites still parse; most damage does not)
