"""JSON lines -> columns on the GPU (`event_bytes_to_batch`) vs the oracle's decoder (Python's
json module): the generator's three relations, free-form objects (any member order, white space, unknown members of any
shape, every escape incl. surrogate pairs), the error cases, and sizes that leave the LDS staging."""
import json

import numpy as np
import pytest

import oracle
from test_oracle_json import SCHEMAS, _events

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _dev_bytes(b):
    from devmem import dev, guarded
    if guarded():          # (the text ends where mapped memory ends: tests/test_gpu_guard.py)
        return dev(np.frombuffer(bytes(b), np.uint8))
    import torch
    t = torch.zeros(len(b) + 16, dtype=torch.uint8, device="cuda")   # 16-byte aligned allocation; only len(b) bytes are text
    if len(b):
        t[: len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    return t[: len(b)]


def _check(cols, want, fields, n):
    for name, t in fields:
        if t == "utf8":
            off = cols[name].offsets.cpu().numpy()
            data = cols[name].data.cpu().numpy()
            assert np.array_equal(off, want[name].offsets), name
            assert np.array_equal(data[: off[-1]], want[name].data[: off[-1]]), name
        else:
            assert np.array_equal(cols[name].cpu().numpy(), want[name]), name
            assert len(want[name]) == n


@pytest.mark.parametrize("relation,n_events", [("bid", 4000), ("auction", 4000), ("person", 4000), ("bid", 1_000_000), ("person", 300_000),
                                               ("auction", 300_000)])
def test_generated_relations(ctx, relation, n_events):
    from flock_amd.nexmark import event_bytes_to_columns
    cols = _events(relation, n_events)
    text = oracle.nexmark_json_lines(relation, cols)
    got, n = event_bytes_to_columns(ctx, _dev_bytes(text), relation)
    assert n == len(cols[SCHEMAS[relation][0][0]]) and n > 0
    _check(got, cols, SCHEMAS[relation], n)
    # no newline after the last line: same rows
    got2, n2 = event_bytes_to_columns(ctx, _dev_bytes(text[:-1]), relation)
    assert n2 == n
    _check(got2, cols, SCHEMAS[relation], n)
    # the library's own result columns, viewed in place (valid until the next call on the context): the same values
    got3, n3 = ctx.json_lines_decode(_dev_bytes(text), [(name, t) for name, t in SCHEMAS[relation]], borrow=True)
    assert n3 == n
    _check(got3, cols, SCHEMAS[relation], n)


def test_free_form_objects_and_escapes(ctx):
    rng = np.random.default_rng(5)
    fields = [("k", "int32"), ("t", "int64"), ("s", "utf8"), ("u", "utf8")]
    specials = ['', 'plain', 'quote " inside', 'back\\slash', 'tab\there', 'nl\nnl', 'uni é ü ß', 'cjk 漢字', 'emoji \U0001F600 end',
                '/slash/', '\b\f\r', 'x' * 300, 'ctl \x01\x1f']
    lines = []
    for i in range(5000):
        o = {"k": int(rng.integers(-2**31, 2**31)), "t": int(rng.integers(-2**62, 2**62)), "s": specials[i % len(specials)],
             "u": "row%d" % i}
        extra = {"z": [1, {"a": "}]\\\""}, [None, True, 1.5e3]], "y": {"n": {"m": "\n"}}, "w": -0.25, "v": None, "k2": "k"}
        items = list(o.items()) + [(k, extra[k]) for k in list(extra)[: i % 6]]
        order = rng.permutation(len(items))
        body = (", " if i % 3 else ",").join("%s%s:%s%s" % (json.dumps(items[j][0]), " " * (i % 2), "\t" * (i % 4 == 1),
                                                                 json.dumps(items[j][1], ensure_ascii=bool(i % 2))) for j in order)
        lines.append((" " * (i % 3) + "{" + " " * (i % 2) + body + "}" + ("\r" if i % 5 == 0 else "")).encode())
    text = b"\n".join(lines) + b"\n"
    want = oracle.json_lines_decode(text, fields)
    got, n = ctx.json_lines_decode(_dev_bytes(text), fields)
    assert n == 5000
    _check(got, want, fields, n)
    # the same rows without any escape in field "u": that field takes the plain byte-range gather, "s" the unescaping copy
    assert got["u"].offsets[-1].item() == sum(len("row%d" % i) for i in range(5000))
    # last-one-wins for a repeated key is what a map-building decoder does as well
    dup = b'{"k":1,"t":2,"s":"a","u":"b","k":7}\n'
    assert oracle.json_lines_decode(dup, fields)["k"].tolist() == [7]
    assert ctx.json_lines_decode(_dev_bytes(dup), fields)[0]["k"].cpu().tolist() == [7]


def test_lines_longer_than_the_staging_buffer(ctx):
    fields = [("id", "int32"), ("blob", "utf8")]
    rng = np.random.default_rng(9)
    lines = [json.dumps({"blob": "".join(chr(97 + int(c)) for c in rng.integers(0, 26, int(rng.integers(0, 900)))), "id": i},
                        separators=(",", ":")).encode() for i in range(3000)]
    text = b"\n".join(lines)
    want = oracle.json_lines_decode(text, fields)
    got, n = ctx.json_lines_decode(_dev_bytes(text), fields)
    assert n == 3000
    _check(got, want, fields, n)


@pytest.mark.parametrize("bad,code", [
    (b'{"a":1,"b":"x"}\n{"a":2}\n', "INVALID"),                        # missing field
    (b'{"a":1,"b":"x"}\n{"a":2,"b":"y"\n', "INVALID"),                 # unterminated object
    (b'{"a":1.5,"b":"x"}\n', "UNSUPPORTED"),                           # fraction
    (b'{"a":1e3,"b":"x"}\n', "UNSUPPORTED"),                           # exponent
    (b'{"a":1,"b":"x"}\n\n{"a":2,"b":"y"}\n', "UNSUPPORTED"),          # blank line
    (b'{"a":3000000000,"b":"x"}\n', "INVALID"),                        # does not fit Int32
    (b'{"a":1,"b":"bad \\q escape"}\n', "INVALID"),
    (b'{"a":1,"b":"lone \\ud800 surrogate"}\n', "INVALID"),
    (b'{"a":1,"b":"x",}\n', "INVALID"),
    (b'{"a":1,"b":"x"} trailing\n', "INVALID"),
    (b'{"a":"1","b":"x"}\n', "INVALID"),                               # string where an integer is expected
    (b'{"\\u0061":1,"b":"x"}\n', "UNSUPPORTED"),                       # escape in a key
    (b'{"a":007,"b":"x"}\n', "INVALID"),                               # leading zeros (serde_json: "invalid number")
    (b'{"b":"x", "a": -01}\n', "INVALID"),
])
def test_errors_name_the_line(ctx, bad, code):
    from flock_amd import FlockGpuError, _ffi
    with pytest.raises(FlockGpuError) as e:
        ctx.json_lines_decode(_dev_bytes(bad), [("a", "int32"), ("b", "utf8")])
    assert e.value.code == getattr(_ffi, "ERR_" + code), str(e.value)
    assert "line" in str(e.value)


def test_empty_text(ctx):
    got, n = ctx.json_lines_decode(_dev_bytes(b""), [("a", "int32"), ("b", "utf8")])
    assert n == 0 and got["a"].numel() == 0 and got["b"].offsets.cpu().tolist() == [0]


def test_compact_lines_every_digit_count_and_string_length(ctx):
    """The shape serde_json writes is walked eight bytes at a time (`parse_line_words`): integers of 1 .. 19 digits of
    either sign, INT64's ends, strings of 0 .. 40 bytes (every position of the closing quote inside an
    8-byte read), multi-byte UTF-8, and a text that ends in a digit / in the closing brace without a newline."""
    fields = [("a", "int64"), ("s", "utf8"), ("b", "int32"), ("c", "int64")]
    rng = np.random.default_rng(11)
    lines = []
    for i in range(6000):
        nd = i % 19 + 1
        a = int(rng.integers(10 ** (nd - 1), min(10 ** nd, 2 ** 63))) if nd > 1 else int(rng.integers(0, 10))
        a = min(a, 2 ** 63 - 1) * (-1 if i % 3 == 0 else 1)
        if i % 500 == 1: a = 2 ** 63 - 1
        if i % 500 == 2: a = -2 ** 63
        s = ("é" if i % 7 == 0 else "") + "abcdefghijklmnopqrstuvwxyz0123456789{}:,'"[: i % 41]
        b = int(rng.integers(-2 ** 31, 2 ** 31)) if i % 11 else [2 ** 31 - 1, -2 ** 31, 0][i % 3]
        c = int(rng.integers(0, 10 ** (i % 18 + 1)))
        lines.append(b'{"a":%d,"s":"%s","b":%d,"c":%d}' % (a, s.encode(), b, c))
    want = None
    for tail in (b"\n", b""):
        text = b"\n".join(lines) + tail
        want = want or oracle.json_lines_decode(text + b"\n" if not tail else text, fields)
        got, n = ctx.json_lines_decode(_dev_bytes(text), fields)
        assert n == len(lines)
        _check(got, want, fields, n)
    # a text that ends in a digit of a number that the line does not close: an error, not a read past the text
    from flock_amd import FlockGpuError
    with pytest.raises(FlockGpuError):
        ctx.json_lines_decode(_dev_bytes(b'{"a":1,"s":"x","b":2,"c":12345678'), fields)
    # 20 digits: out of range whichever path meets it
    with pytest.raises(FlockGpuError):
        ctx.json_lines_decode(_dev_bytes(b'{"a":12345678901234567890,"s":"x","b":2,"c":3}\n'), fields)


def test_flat_objects_in_any_order_with_white_space(ctx):
    """What `parse_line_flex` takes before the byte-wise general parser does: flat objects holding exactly the schema's members, in any
    order, with white space wherever JSON allows it (Python's default `json.dumps` separators among them).  Mixed into the same text:
    compact serde_json lines (the first walker's), and lines only the general parser accepts (an unknown member, an escape) -- every
    line must come out the same whoever parsed it."""
    fields = [("id", "int64"), ("name", "utf8"), ("n", "int32"), ("city", "utf8")]
    rng = np.random.default_rng(21)
    lines = []
    for i in range(8000):
        o = {"id": int(rng.integers(-2 ** 62, 2 ** 62)) if i % 5 else int(rng.integers(0, 10 ** (i % 19))), "name": "name-%d" % i + "x" * (i % 23),
             "n": int(rng.integers(-2 ** 31, 2 ** 31)), "city": ["", "Paris", "São Paulo", "a" * 40][i % 4]}
        items = list(o.items())
        order = rng.permutation(4) if i % 7 else np.arange(4)
        kind = i % 6
        if kind == 0:      # compact, schema order or not
            line = "{" + ",".join('%s:%s' % (json.dumps(items[j][0]), json.dumps(items[j][1], ensure_ascii=False)) for j in order) + "}"
        elif kind in (1, 2, 3):   # Python's default separators; more white space; tabs and a trailing \r
            sep, col = [(", ", ": "), (" ,  ", "  :\t"), (",\t", ":  ")][kind - 1]
            line = " " * (i % 3) + "{" + " " * (i % 2) + sep.join('%s%s%s' % (json.dumps(items[j][0]), col, json.dumps(items[j][1], ensure_ascii=False))
                                                                  for j in order) + " " * (i % 4 == 1) + "}" + ("\r" if i % 9 == 0 else "")
        elif kind == 4:    # an unknown member: the general parser's
            line = json.dumps(dict([items[j] for j in order] + [("extra", [1, {"a": "}"}])]), ensure_ascii=False)
        else:              # an escape in a value
            o2 = dict(o, name=o["name"] + ' "q" \\ \n')
            line = json.dumps({k: o2[k] for k in [items[j][0] for j in order]})
        lines.append(line.encode())
    text = b"\n".join(lines) + b"\n"
    want = oracle.json_lines_decode(text, fields)
    got, n = ctx.json_lines_decode(_dev_bytes(text), fields)
    assert n == len(lines)
    _check(got, want, fields, n)
    # errors a flat line can hold are still the general parser's errors
    from flock_amd import FlockGpuError, _ffi
    for bad, code in ((b'{"id": 1, "name": "a", "n": 2}\n', "INVALID"),                       # a member is missing
                      (b'{"id": 1, "name": "a", "n": 2, "city": "x", "id": 3}\n', None),      # a repeated key: last one wins (a map-building decoder)
                      (b'{"id": 1.5, "name": "a", "n": 2, "city": "x"}\n', "UNSUPPORTED"),
                      (b'{"id": 01, "name": "a", "n": 2, "city": "x"}\n', "INVALID"),
                      (b'{"id": 1, "name": "a", "n": 2, "city": "x"} x\n', "INVALID")):
        if code is None:
            assert ctx.json_lines_decode(_dev_bytes(bad), fields)[0]["id"].cpu().tolist() == [3]
            continue
        with pytest.raises(FlockGpuError) as e:
            ctx.json_lines_decode(_dev_bytes(bad), fields)
        assert e.value.code == getattr(_ffi, "ERR_" + code), (bad, str(e.value))


@pytest.mark.parametrize("seed", range(12))
def test_random_schemas_and_writers(ctx, seed):
    """Randomised: 1 .. 9 fields of random types and names (1 .. 31 bytes: every key-word count), integers of every digit count, strings
    with and without escapes, and per text one of four writers -- serde_json's compact shape, Python's separators, members shuffled per
    line, and a mix with unknown members -- so that all three parsers (compact walk, flat-object walker, general) and the routing
    between their two kernels (first call, steady state, a change of writer on the same context) meet the same oracle."""
    rng = np.random.default_rng(1000 + seed)
    n_fields = int(rng.integers(1, 10))
    names = []
    while len(names) < n_fields:
        ln = int(rng.integers(1, 32))
        nm = "".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz_0123456789"), ln))
        if nm not in names:
            names.append(nm)
    types = [["int32", "int64", "utf8"][int(rng.integers(0, 3))] for _ in names]
    fields = list(zip(names, types))

    def value(t, i):
        if t == "int32":
            return int(rng.integers(-2 ** 31, 2 ** 31)) if i % 3 else int(rng.integers(0, 10 ** int(rng.integers(1, 10))))
        if t == "int64":
            d = int(rng.integers(1, 19))
            return int(rng.integers(0, 10 ** d)) * (-1 if i % 4 == 0 else 1)
        k = int(rng.integers(0, 6))
        return ["", "x" * int(rng.integers(0, 40)), "é漢" * int(rng.integers(0, 5)), 'q"uote', "back\\slash\n", "plain %d" % i][k]

    def lines(writer, n):
        out = []
        for i in range(n):
            items = [(nm, value(t, i)) for nm, t in fields]
            if writer in (2, 3):
                items = [items[j] for j in rng.permutation(len(items))]
            if writer == 3 and i % 5 == 0:
                items.insert(int(rng.integers(0, len(items) + 1)), ("unknown member", [1, {"k": "}"}, None]))
            if writer == 0:
                out.append(json.dumps(dict(items), separators=(",", ":"), ensure_ascii=bool(i % 2)).encode())
            elif writer == 1:
                out.append(json.dumps(dict(items), ensure_ascii=bool(i % 2)).encode())
            else:
                sep = [",", ", ", " ,\t"][i % 3]
                out.append(("{" + sep.join("%s%s%s" % (json.dumps(k), [":", ": ", " : "][i % 3], json.dumps(v, ensure_ascii=bool(i % 2))) for k, v in items) + "}").encode())
        return b"\n".join(out) + (b"\n" if n % 2 else b"")
    for writer in rng.permutation(4).tolist() + [0, 1, 1, 0]:     # (the same context throughout: the route hint carries over)
        text = lines(writer, int(rng.integers(1, 3000)))
        want = oracle.json_lines_decode(text, fields)
        got, n = ctx.json_lines_decode(_dev_bytes(text), fields)
        assert n == len(want[names[0]].offsets) - 1 if types[0] == "utf8" else n == len(want[names[0]])
        _check(got, want, fields, n)
