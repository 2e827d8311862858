"""The payload codec (flock_amd/payload.py): the hand-written record-batch header and its parser
against Arrow C++ (pyarrow), `Encoding`'s framings, the serde_json shape of `Payload`; and, on the GPU, device batches ->
payload -> device batches with the body compared byte for byte with the one Arrow's own writer produces."""
import json
import struct

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")

from flock_amd import payload as P  # noqa: E402


def _host_body(arrays):
    """Buffers of pyarrow arrays in IPC order, padded to 8 bytes: what the device packer must produce."""
    body, nodes, bufs = bytearray(), [], []
    for a in arrays:
        nodes.append((len(a), a.null_count))
        for i, b in enumerate(a.buffers()):
            if i == 0:                                   # validity: absent for non-nullable data
                bufs.append((len(body), 0))
                continue
            raw = b.to_pybytes()
            if pa.types.is_string(a.type):
                offsets = np.frombuffer(a.buffers()[1], np.int32)[: len(a) + 1]
                raw = raw[: 4 * (len(a) + 1)] if i == 1 else raw[: int(offsets[-1])]
            else:
                raw = raw[: a.type.bit_width // 8 * len(a)]
            bufs.append((len(body), len(raw)))
            body += raw + b"\0" * ((-len(raw)) % 8)
    return bytes(body), nodes, bufs


def _batch(n, seed=0):
    rng = np.random.default_rng(seed)
    names = ["", "a", "bc", "sarah white", "x" * 40]
    return pa.record_batch([
        pa.array(rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)),
        pa.array([names[i] for i in rng.integers(0, len(names), n)], type=pa.utf8()),
        pa.array(rng.integers(0, 2**62, n).astype(np.uint64)),
        pa.array(rng.integers(1_436_918_400_000, 1_436_918_500_000, n), type=pa.timestamp("ms")),
        pa.array(rng.random(n)),
    ], schema=pa.schema([pa.field("auction", pa.int32(), False), pa.field("name", pa.utf8(), False), pa.field("num", pa.uint64(), False),
                         pa.field("b_date_time", pa.timestamp("ms"), False), pa.field("price", pa.float64(), False)]))


@pytest.mark.parametrize("n", [0, 1, 5, 1000])
def test_header_is_read_by_arrow_and_arrow_headers_are_parsed(n):
    batch = _batch(n, n)
    body, nodes, bufs = _host_body(batch.columns)
    header = P.record_batch_header(n, nodes, bufs, len(body))
    msg = pa.ipc.read_message(pa.py_buffer(P.encapsulate(header, body)))
    assert pa.ipc.read_record_batch(msg, batch.schema).equals(batch)
    # Arrow's own message for the same batch: same body, and its header parses to the same description
    theirs = pa.ipc.read_message(batch.serialize())
    assert theirs.body.to_pybytes() == body
    rows, t_nodes, t_bufs, t_len = P.parse_record_batch_header(theirs.metadata.to_pybytes())
    assert (rows, t_nodes, t_len) == (n, nodes, len(body))
    assert [(o, l) for o, l in t_bufs] == bufs
    assert P.parse_record_batch_header(header) == (n, nodes, bufs, len(body))


def test_schema_bytes_round_trip():
    schema = _batch(3).schema
    raw = P.schema_to_bytes(schema)
    assert P.schema_from_bytes(raw).equals(schema)
    assert raw == schema.serialize().to_pybytes()[8:8 + len(raw)]


@pytest.mark.parametrize("name", ["Zstd", "Snappy", "Lz4", "None"])
def test_encodings_round_trip_with_the_references_framing(name):
    enc = P.Encoding(name)
    rng = np.random.default_rng(1)
    for data in (b"", b"a", bytes(rng.integers(0, 4, 100_000).astype(np.uint8)), bytes(rng.integers(0, 256, 70_000).astype(np.uint8))):
        packed = enc.compress(data)
        assert enc.decompress(packed) == data
        if name == "Lz4":        # lz4::block::compress(.., prepend_size = true)
            assert struct.unpack("<I", packed[:4])[0] == len(data)
        if name == "Zstd" and data:
            assert packed[:4] == b"\x28\xb5\x2f\xfd" and P._zstd_content_size(packed) == len(data)
    assert P.Encoding() == P.Encoding("Zstd")
    with pytest.raises(NotImplementedError):
        P.Encoding("Zlib").compress(b"x")


def test_payload_json_has_the_serde_shape():
    p = P.Payload(data=[P.DataFrame(b"\x01\x02", b"\xff")], schema=b"\x07", uuid=P.Uuid("q5-1-2", 3, 4), encoding=P.Encoding("Zstd"),
                  datasource={"Payload": True}, query_number=5, metadata={"k": "v"})
    o = json.loads(p.to_json())
    assert o["data"] == [{"header": [1, 2], "body": [255]}] and o["schema"] == [7] and o["data2"] == [] and o["schema2"] == []
    assert o["uuid"] == {"qid": "q5-1-2", "seq_num": 3, "seq_len": 4} and o["encoding"] == "Zstd"
    assert o["datasource"] == {"Payload": True} and o["query_number"] == 5 and o["shuffle_id"] is None and o["metadata"] == {"k": "v"}
    q = P.Payload.from_json(p.to_json())
    assert q.data == p.data and q.schema == p.schema and q.uuid == p.uuid and q.encoding == p.encoding and not q.is_empty_data()
    assert P.Payload().is_empty_data()


def test_reference_size_goldens():
    """The two Arrow Flight sizes the reference asserts (payload.rs:309: 1856 for 37 UK cities; :402: 3453248 for 21275
    citibike trips) follow from the batch shapes alone (tests/golden/payload_sizes.json, tools/make_payload_golden.py): body
    with one all-ones validity bitmap per field, as its arrow-rs writer emits, + the 80 + 16 * (fields + buffers)-byte header."""
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "payload_sizes.json")))
    for name, case in g.items():
        cols = [(c["type"], c["value_bytes"]) for c in case["columns"]]
        header, body = P.flight_data_sizes(case["rows"], cols, validity=True)
        assert header + body == case["reference_flight_data_size"], name
        h2, b2 = P.flight_data_sizes(case["rows"], cols, validity=False)
        assert h2 == header and b2 < body


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _device_batch(batch):
    import torch
    from flock_amd import DeviceUtf8
    cols = []
    kinds = {pa.int32(): "int32", pa.int64(): "int64", pa.uint64(): "uint64", pa.float64(): "float64", pa.timestamp("ms"): "timestamp_ms",
             pa.utf8(): "utf8"}
    fields = [(f.name, kinds[f.type]) for f in batch.schema]
    for col, (_, k) in zip(batch.columns, fields):
        if k == "utf8":
            off = np.frombuffer(col.buffers()[1], np.int32)[: len(col) + 1].copy()
            data = np.frombuffer(col.buffers()[2], np.uint8)[: off[-1]].copy() if col.buffers()[2] is not None else np.zeros(0, np.uint8)
            data = np.concatenate([data, np.zeros(16, np.uint8)])
            cols.append(DeviceUtf8(torch.from_numpy(off).cuda(), torch.from_numpy(data).cuda()))
        else:
            np_t = {"int32": np.int32, "int64": np.int64, "uint64": np.uint64, "float64": np.float64, "timestamp_ms": np.int64}[k]
            host = np.frombuffer(col.buffers()[1], np_t)[: len(col)].copy()
            cols.append(torch.from_numpy(host.view(np.int64) if k == "uint64" else host).cuda())
    return P.DeviceBatch(fields, cols, len(batch))


@pytest.mark.gpu
@pytest.mark.parametrize("n,encoding", [(0, "Zstd"), (1, "None"), (777, "Zstd"), (200_000, "Lz4"), (50_000, "Snappy")])
def test_device_batches_through_a_payload(ctx, n, encoding):
    b1, b2 = _batch(n, 1), _batch(max(n // 3, 1), 2)
    # without validity bitmaps the body is Arrow C++'s body byte for byte, and Arrow reads the header
    cpp = P.to_payload(ctx, [_device_batch(b1)], [], P.Uuid("q-1-1", 0, 1), True, P.Encoding(encoding), validity=False)
    header, body = cpp.encoding.decompress(cpp.data[0].header), cpp.encoding.decompress(cpp.data[0].body)
    assert body == pa.ipc.read_message(b1.serialize()).body.to_pybytes()
    assert pa.ipc.read_record_batch(pa.ipc.read_message(pa.py_buffer(P.encapsulate(header, body))), P.schema_from_bytes(cpp.schema)).equals(b1)
    # the default is the reference writer's variant (an all-ones validity bitmap per field): read by Arrow just the same, and
    # of the size flight_data_sizes predicts (the formula behind the reference's size goldens)
    pay = P.to_payload(ctx, [_device_batch(b1), _device_batch(b1)], [_device_batch(b2)], P.Uuid("q-1-1", 0, 1), True, P.Encoding(encoding))
    assert len(pay.data) == 2 and len(pay.data2) == 1 and pay.datasource == {"Payload": True}
    h3, b3 = P.batch_to_flight_data(ctx, _device_batch(b1))
    assert pay.encoding.decompress(pay.data[0].body) == b3 and pay.encoding.decompress(pay.data[0].header) == h3
    assert len(b3) == len(body) + len(b1.schema) * ((((n + 7) // 8) + 7) & ~7)
    kinds = {pa.int32(): "int32", pa.int64(): "int64", pa.uint64(): "uint64", pa.float64(): "float64", pa.timestamp("ms"): "timestamp_ms"}
    shape = [("utf8", len(c.buffers()[2]) if c.buffers()[2] is not None else 0) if pa.types.is_string(c.type) else (kinds[c.type], None) for c in b1.columns]
    if all(k != "utf8" for k, _ in shape) or n == 0:
        assert (len(h3), len(b3)) == P.flight_data_sizes(n, [(k, v if k != "utf8" else 0) for k, v in shape])
    assert pa.ipc.read_record_batch(pa.ipc.read_message(pa.py_buffer(P.encapsulate(h3, b3))), b1.schema).equals(b1)
    back = P.flight_data_to_batch(ctx, h3, b3, _device_batch(b1).fields)
    assert back.rows == n and (n == 0 or back.columns[0].cpu().numpy().tobytes() == np.frombuffer(b1.columns[0].buffers()[1], np.uint8)[: 4 * n].tobytes())
    # over the wire and back onto the device
    got1, got2 = P.Payload.from_json(pay.to_json()).to_record_batch(ctx)
    for got, want in ((got1[0], b1), (got1[1], b1), (got2[0], b2)):
        assert got.rows == len(want) and [f[0] for f in got.fields] == want.schema.names
        for col, w in zip(got.columns, want.columns):
            if pa.types.is_string(w.type):
                off = col.offsets.cpu().numpy()
                data = col.data.cpu().numpy()[: off[-1]].tobytes()
                assert [data[off[i]:off[i + 1]].decode() for i in range(len(w))] == w.to_pylist()
            else:
                host = col.cpu().numpy()
                assert host.tobytes() == np.frombuffer(w.buffers()[1], np.uint8)[: host.nbytes].tobytes()


@pytest.mark.gpu
def test_query_output_as_payload(ctx):
    """q5's result columns (auction Int32, num UInt64) as the payload the next function would receive."""
    import torch
    from flock_amd import NEXMarkSource, Window, run_query
    w = Window.hopping(10, 5)
    g = NEXMarkSource(30, 20_000, w, seed=4).generate_data(ctx, relations=("bid",), bid_columns=("auction",))
    r = run_query(ctx, 5, g)
    a, n, off = r.to_host()
    dev = f"cuda:{ctx.device}"
    batch = P.DeviceBatch([("auction", "int32"), ("num", "uint64")],
                          [torch.from_numpy(a).to(dev), torch.from_numpy(n.view(np.int64)).to(dev)], len(a))
    pay = P.to_payload(ctx, [batch], [], P.Uuid("q5-0-0", 0, 1), False)
    (got,), none = P.Payload.from_json(pay.to_json()).to_record_batch(ctx)
    assert none == [] and got.rows == len(a)
    assert np.array_equal(got.columns[0].cpu().numpy(), a) and np.array_equal(got.columns[1].cpu().numpy().view(np.uint64), n)


def test_json_lines_size_golden():
    """payload.rs:365-372: the citibike batch through arrow's json::LineDelimitedWriter is 9436023 bytes -- one compact
    serde_json object per row and a newline, which is the format the JSON ingest (json.hip) decodes.  The number is recomputed
    from the reference's CSV by tools/make_payload_golden.py (where the reference tree is present) and frozen next to the literal."""
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "payload_sizes.json")))["citibike"]
    assert g["json_lines_bytes"] == g["reference_json_lines_bytes"] == 9436023
    head = open(os.path.join(os.path.dirname(__file__), "golden", "citibike_head.jsonl")).read().splitlines()
    assert len(head) == 256 and json.loads(head[0])["start station id"] == 3275 and json.loads(head[0])["tripduration"] == "756"


@pytest.mark.gpu
def test_json_ingest_decodes_the_citibike_lines(ctx):
    """The first 256 citibike rows as those JSON lines through flockgpu_json_lines_decode: the Int32 and Utf8 members against
    Python's json (Float64 members are skipped: the decoder's types are Int32 / Int64 / Utf8)."""
    import os
    import torch
    text = open(os.path.join(os.path.dirname(__file__), "golden", "citibike_head.jsonl"), "rb").read()
    want = [json.loads(line) for line in text.splitlines()]
    fields = [("tripduration", "utf8"), ("start station id", "int32"), ("start station name", "utf8"), ("bikeid", "int32"),
              ("usertype", "utf8"), ("birth year", "int32"), ("gender", "int32")]
    dev = torch.frombuffer(bytearray(text + b"\0" * 16), dtype=torch.uint8).cuda()[: len(text)]
    cols, n = ctx.json_lines_decode(dev, fields)
    assert n == len(want)
    for name, kind in fields:
        if kind == "utf8":
            off = cols[name].offsets.cpu().numpy()
            raw = cols[name].data.cpu().numpy()[: off[-1]].tobytes()
            assert [raw[off[i]:off[i + 1]].decode() for i in range(n)] == [w[name] for w in want], name
        else:
            assert cols[name].cpu().numpy().tolist() == [w[name] for w in want], name
