"""The wire format between Flock's cloud functions, for batches that live in HBM.

Reference: `Payload` (flock/src/runtime/payload.rs:118-192) = per record batch a `DataFrame {header, body}` holding the two
halves of its Arrow Flight data (`flight_data_from_arrow_batch`, flock/src/transmute.rs:155-170,190-205), each compressed
with `Encoding` (flock/src/encoding.rs:57-100, default Zstd level 3), the IPC schema message (`schema_to_bytes`,
transmute.rs:124-129), the `Uuid`, and bookkeeping fields; the struct travels as serde_json.

Here the BODY of a batch -- its buffers end to end, each padded to 8 bytes -- is assembled on the device
(`flockgpu_ipc_pack_body`) and crosses PCIe once; the header is the small flatbuffer `Message{RecordBatch{length, nodes,
buffers}}` written by `record_batch_header` below; compression stays on the CPU, as in the reference.  The other direction
parses the header, moves the body to the device once, and the columns are views into it.

Interoperability is checked against Arrow C++ (pyarrow) in tests/test_payload.py: its reader accepts header + body written
here, and the body equals the one its writer produces for the same batch, byte for byte."""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# Arrow types of the query outputs (benchmarks/src/nexmark/query/q*_plan.fmt): name -> (bytes per value | None for Utf8)
ARROW_WIDTH = {"int32": 4, "int64": 8, "uint64": 8, "float64": 8, "timestamp_ms": 8, "utf8": None}


# ---------------------------------------------------------------- Encoding (flock/src/encoding.rs)
def _zstd_content_size(frame: bytes) -> int:
    """Frame_Content_Size of a zstd frame (RFC 8878 section 3.1.1.1): zstd::block::compress always records it."""
    if frame[:4] != b"\x28\xb5\x2f\xfd":
        raise ValueError("not a zstd frame")
    fhd = frame[4]
    fcs_flag, single, dict_flag = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[dict_flag]
    size = (1 if single else 0, 2, 4, 8)[fcs_flag]
    if size == 0:
        raise ValueError("zstd frame without a content size")
    v = int.from_bytes(frame[pos:pos + size], "little")
    return v + 256 if size == 2 else v


class Encoding:
    """`Encoding::{Snappy, Lz4, Zstd, None}` with the reference's framing (encoding.rs:57-100): raw snappy, an LZ4 block
    behind its little-endian uncompressed size (`prepend_size = true`), a zstd frame, or the bytes themselves."""
    NAMES = ("Snappy", "Lz4", "Zlib", "Zstd", "None")

    def __init__(self, name: str = "Zstd"):   # Default for Encoding = Zstd (encoding.rs:50-54)
        if name not in self.NAMES:
            raise ValueError(name)
        self.name = name

    def __eq__(self, other):
        return isinstance(other, Encoding) and other.name == self.name

    def __repr__(self):
        return f"Encoding::{self.name}"

    def compress(self, data: bytes) -> bytes:
        import pyarrow as pa
        if self.name == "None":
            return bytes(data)
        if self.name == "Zstd":
            return pa.compress(data, codec="zstd", asbytes=True)
        if self.name == "Snappy":
            return pa.compress(data, codec="snappy", asbytes=True)
        if self.name == "Lz4":
            return struct.pack("<I", len(data)) + pa.compress(data, codec="lz4_raw", asbytes=True)
        raise NotImplementedError(self.name)   # Zlib: `unimplemented!()` in the reference as well

    def decompress(self, data: bytes) -> bytes:
        import pyarrow as pa
        if self.name == "None":
            return bytes(data)
        if self.name == "Zstd":
            return pa.decompress(data, decompressed_size=_zstd_content_size(data), codec="zstd", asbytes=True)
        if self.name == "Snappy":
            n, shift, pos = 0, 0, 0      # raw snappy starts with the uncompressed length as a varint
            while True:
                b = data[pos]
                n |= (b & 0x7F) << shift
                pos += 1
                if not b & 0x80:
                    break
                shift += 7
            return pa.decompress(data, decompressed_size=n, codec="snappy", asbytes=True)
        if self.name == "Lz4":
            (n,) = struct.unpack("<I", data[:4])
            return pa.decompress(data[4:], decompressed_size=n, codec="lz4_raw", asbytes=True)
        raise NotImplementedError(self.name)


# ---------------------------------------------------------------- the record-batch header (a flatbuffer, written by hand)
def record_batch_header(n_rows: int, nodes: Sequence[Tuple[int, int]], buffers: Sequence[Tuple[int, int]], body_length: int) -> bytes:
    """org.apache.arrow.flatbuf.Message { version: V5, header: RecordBatch { length, nodes: [FieldNode(length, null_count)],
    buffers: [Buffer(offset, length)] }, bodyLength } -- what `FlightData.data_header` holds for a record batch.

    Fixed layout (every int64 on an 8-byte boundary), 80 + 16 * (fields + buffers) bytes -- the size arrow-rs's writer produces,
    which is what the reference's Flight-size goldens count (payload.rs:309, :402):
        0 root offset | 4 Message vtable (12 B) | 16 Message table | 36 RecordBatch vtable (10 B) | 48 RecordBatch table
        | 68 nodes count, 72 nodes | 4 B pad, buffers count, buffers"""
    nodes_pos = 68                                   # element count; elements start 8-aligned at 72
    nodes_end = 72 + 16 * len(nodes)
    bufs_pos = nodes_end + 4                         # (4 bytes of padding before the count keep the elements 8-aligned)
    total = bufs_pos + 4 + 16 * len(buffers)
    out = bytearray(total)
    struct.pack_into("<I", out, 0, 16)                                   # root -> Message table
    struct.pack_into("<6H", out, 4, 12, 20, 16, 18, 4, 8)                # vtable: size, table size, version, header_type, header, bodyLength
    struct.pack_into("<iIqhB", out, 16, 12, 48 - 20, body_length, 4, 3)   # soffset, header -> RecordBatch, bodyLength, V5 (= 4), RecordBatch (= 3)
    struct.pack_into("<5H", out, 36, 10, 20, 8, 4, 16)                   # vtable: size, table size, length, nodes, buffers
    struct.pack_into("<iIqI", out, 48, 12, nodes_pos - 52, n_rows, bufs_pos - 64)
    struct.pack_into("<I", out, nodes_pos, len(nodes))
    for i, (length, nulls) in enumerate(nodes):
        struct.pack_into("<qq", out, 72 + 16 * i, length, nulls)
    struct.pack_into("<I", out, bufs_pos, len(buffers))
    for i, (off, length) in enumerate(buffers):
        struct.pack_into("<qq", out, bufs_pos + 4 + 16 * i, off, length)
    return bytes(out)


def parse_record_batch_header(header: bytes):
    """(n_rows, nodes, buffers, body_length) of any valid Message{RecordBatch} flatbuffer (arrow-rs, Arrow C++ or the one above)."""
    def table_field(table: int, idx: int) -> Optional[int]:
        vt = table - struct.unpack_from("<i", header, table)[0]
        vt_len = struct.unpack_from("<H", header, vt)[0]
        slot = 4 + 2 * idx
        if slot >= vt_len:
            return None
        off = struct.unpack_from("<H", header, vt + slot)[0]
        return table + off if off else None

    def follow(pos: int) -> int:
        return pos + struct.unpack_from("<I", header, pos)[0]

    msg = follow(0)
    htype = table_field(msg, 1)
    if htype is None or header[htype] != 3:
        raise ValueError("not a RecordBatch message")
    body_pos = table_field(msg, 3)
    body_length = struct.unpack_from("<q", header, body_pos)[0] if body_pos else 0
    rb = follow(table_field(msg, 2))
    if table_field(rb, 3) is not None:
        raise NotImplementedError("compressed IPC bodies (BodyCompression) are not used by the reference")
    length_pos = table_field(rb, 0)
    n_rows = struct.unpack_from("<q", header, length_pos)[0] if length_pos else 0

    def struct_vector(idx: int):
        p = table_field(rb, idx)
        if p is None:
            return []
        v = follow(p)
        n = struct.unpack_from("<I", header, v)[0]
        return [struct.unpack_from("<qq", header, v + 4 + 16 * i) for i in range(n)]
    return n_rows, struct_vector(1), struct_vector(2), body_length


def encapsulate(header: bytes, body: bytes) -> bytes:
    """The stream framing Arrow C++ reads (continuation marker, padded metadata length, metadata, body)."""
    pad = (-len(header)) % 8
    return b"\xff\xff\xff\xff" + struct.pack("<i", len(header) + pad) + header + b"\0" * pad + body


# ---------------------------------------------------------------- batches in HBM
@dataclass
class DeviceBatch:
    """A record batch whose buffers are device tensors: `columns[i]` is a 1-D tensor (fixed width) or a DeviceUtf8."""
    fields: List[Tuple[str, str]]          # (name, one of ARROW_WIDTH)
    columns: list
    rows: int

    def arrow_schema(self):
        import pyarrow as pa
        t = {"int32": pa.int32(), "int64": pa.int64(), "uint64": pa.uint64(), "float64": pa.float64(),
             "timestamp_ms": pa.timestamp("ms"), "utf8": pa.utf8()}
        return pa.schema([pa.field(n, t[k], nullable=False) for n, k in self.fields])


def schema_to_bytes(schema) -> bytes:
    """`schema_to_bytes` (transmute.rs:124-129): the IPC Schema message flatbuffer without the stream framing."""
    raw = schema.serialize().to_pybytes()
    (n,) = struct.unpack_from("<i", raw, 4)
    return raw[8:8 + n]


def schema_from_bytes(data: bytes):
    import pyarrow as pa
    pad = (-len(data)) % 8
    return pa.ipc.read_schema(pa.py_buffer(b"\xff\xff\xff\xff" + struct.pack("<i", len(data) + pad) + data + b"\0" * pad))


def flight_data_sizes(rows: int, columns: Sequence[Tuple[str, Optional[int]]], validity: bool = True) -> Tuple[int, int]:
    """(header bytes, body bytes) of a batch's Arrow Flight data from its shape alone: columns = (type, total value bytes of
    a Utf8 column).  validity = True is the reference's writer (arrow-rs of its day writes an all-ones bitmap of
    ceil(rows / 8) bytes for EVERY field, `write_array_data`); header + body are its two size goldens (payload.rs:309, :402;
    tests/test_payload.py)."""
    pad = lambda n: (n + 7) & ~7
    width = {"int8": 1, "int32": 4, "int64": 8, "uint64": 8, "float64": 8, "timestamp_ms": 8}
    body, n_buf = 0, 0
    for kind, value_bytes in columns:
        body += pad((rows + 7) // 8) if validity else 0
        if kind == "utf8":
            body += pad(4 * (rows + 1)) + pad(value_bytes)
            n_buf += 3
        else:
            body += pad(width[kind] * rows)
            n_buf += 2
    return len(record_batch_header(rows, [(rows, 0)] * len(columns), [(0, 0)] * n_buf, body)), body


def _layout(batch: DeviceBatch, validity_bits=None):
    """Buffers of the batch in IPC order: per field the validity bitmap (an all-ones device buffer when `validity_bits` is
    given -- what the reference's writer emits --, else absent, as Arrow C++ writes non-nullable data), then offsets / values."""
    from .engine import DeviceUtf8
    bufs, nodes = [], []
    for (name, kind), col in zip(batch.fields, batch.columns):
        nodes.append((batch.rows, 0))
        bufs.append((validity_bits, (batch.rows + 7) // 8) if validity_bits is not None else (None, 0))
        if kind == "utf8":
            assert isinstance(col, DeviceUtf8)
            n_bytes = int(col.offsets[batch.rows].item()) if batch.rows else 0
            bufs.append((col.offsets, 4 * (batch.rows + 1)))
            bufs.append((col.data, n_bytes))
        else:
            bufs.append((col, ARROW_WIDTH[kind] * batch.rows))
    return nodes, bufs


def batch_to_flight_data(ctx, batch: DeviceBatch, keep_view: bool = False, validity: bool = True):
    """(data_header, data_body) of `flight_data_from_arrow_batch` for a device batch: body packed on the device, one D2H
    into pinned memory.  keep_view: return the body as a numpy view of the context's pinned buffer (valid until the next
    call) instead of a bytes copy -- what a caller that compresses it right away wants.  validity (default): an all-ones
    bitmap per field, the bytes the reference's arrow-rs writer emits; False leaves it out like Arrow C++ (readers accept both)."""
    import ctypes as C
    import torch
    from . import _ffi
    ones = None
    if validity:   # all-ones bitmap, shared by every field (the last byte's spare bits are 1 as well, as `with_bitset` leaves them)
        ones = torch.full(((batch.rows + 7) // 8 + 16,), 255, dtype=torch.uint8, device=f"cuda:{ctx.device}")
    nodes, bufs = _layout(batch, ones)
    arr = (_ffi.IpcBuffer * len(bufs))(*[_ffi.IpcBuffer(t.data_ptr() if t is not None and n else None, n) for t, n in bufs])
    total = C.c_int64(0)
    ctx._check(ctx._lib.flockgpu_ipc_pack_body(ctx._h, arr, len(bufs), None, 0, C.byref(total)))
    # device staging buffer and pinned host buffer are kept (grow-only) on the context: pageable D2H runs at ~3 GB/s
    cache = ctx.__dict__.setdefault("_ipc_buffers", {})
    if cache.get("cap", -1) < total.value:
        cap = max(int(total.value * 5 // 4), 1 << 16)
        cache.update(cap=cap, dev=torch.empty(cap, dtype=torch.uint8, device=f"cuda:{ctx.device}"),
                     host=torch.empty(cap, dtype=torch.uint8).pin_memory())
    body_dev, host = cache["dev"], cache["host"]
    ctx._check(ctx._lib.flockgpu_ipc_pack_body(ctx._h, arr, len(bufs), body_dev.data_ptr(), total.value, C.byref(total)))
    host[: total.value].copy_(body_dev[: total.value])       # (the ctx stream is a blocking stream: ordered with torch's)
    body = host.numpy()[: total.value] if keep_view else host.numpy()[: total.value].tobytes()
    offs, pos = [], 0
    for _, n in bufs:
        offs.append((pos, n))
        pos += (n + 7) & ~7
    return record_batch_header(batch.rows, nodes, offs, total.value), body


def flight_data_to_batch(ctx, header: bytes, body: bytes, fields: Sequence[Tuple[str, str]]) -> DeviceBatch:
    """`flight_data_to_arrow_batch` onto the device: ONE H2D copy of the body; columns are views into it (a buffer that
    is not 16-byte aligned inside the body -- IPC aligns to 8 -- is re-based by a device copy, the kernels load 16 bytes)."""
    import torch
    from .engine import DeviceUtf8
    n_rows, nodes, buffers, body_length = parse_record_batch_header(header)
    if len(nodes) != len(fields):
        raise ValueError("schema and record batch disagree on the number of fields")
    dev = torch.zeros(len(body) + 16, dtype=torch.uint8, device=f"cuda:{ctx.device}")
    if body:
        dev[: len(body)] = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(dev.device)

    def view(off, length, dtype, count):
        raw = dev[off: off + length]
        if raw.data_ptr() % 16:
            raw = raw.clone()
        return raw.view(dtype)[:count]
    cols, b = [], 0
    for (name, kind), (length, nulls) in zip(fields, nodes):
        if nulls:
            raise NotImplementedError("nullable input columns (the NEXMark / YSB schemas have none)")
        b += 1                                                        # validity
        if kind == "utf8":
            off = view(*buffers[b], torch.int32, n_rows + 1)
            data = dev[buffers[b + 1][0]: buffers[b + 1][0] + max(buffers[b + 1][1], 16)]
            cols.append(DeviceUtf8(off, data.clone() if data.data_ptr() % 16 else data))
            b += 2
        else:
            dt = {"int32": torch.int32, "int64": torch.int64, "uint64": torch.int64, "float64": torch.float64,
                  "timestamp_ms": torch.int64}[kind]
            cols.append(view(*buffers[b], dt, n_rows))
            b += 1
    return DeviceBatch(list(fields), cols, int(n_rows))


# ---------------------------------------------------------------- Payload (flock/src/runtime/payload.rs)
@dataclass
class Uuid:
    qid: str = ""
    seq_num: int = 0
    seq_len: int = 0


@dataclass
class DataFrame:
    header: bytes = b""
    body: bytes = b""


@dataclass
class Payload:
    data: List[DataFrame] = field(default_factory=list)
    schema: bytes = b""
    data2: List[DataFrame] = field(default_factory=list)
    schema2: bytes = b""
    uuid: Uuid = field(default_factory=Uuid)
    encoding: Encoding = field(default_factory=Encoding)
    datasource: object = None            # serde value of `DataSource`; `to_payload` sets {"Payload": sync}
    query_number: Optional[int] = None
    shuffle_id: Optional[int] = None
    metadata: Optional[Dict[str, str]] = None

    # serde_json of the struct: Vec<u8> (with or without serde_bytes) is an array of numbers, unit enum variants are strings
    def to_json(self) -> str:
        frame = lambda f: {"header": list(f.header), "body": list(f.body)}
        return json.dumps({
            "data": [frame(f) for f in self.data], "schema": list(self.schema),
            "data2": [frame(f) for f in self.data2], "schema2": list(self.schema2),
            "uuid": {"qid": self.uuid.qid, "seq_num": self.uuid.seq_num, "seq_len": self.uuid.seq_len},
            "encoding": self.encoding.name, "datasource": self.datasource, "query_number": self.query_number,
            "shuffle_id": self.shuffle_id, "metadata": self.metadata}, separators=(",", ":"))

    @staticmethod
    def from_json(text) -> "Payload":
        o = json.loads(text)
        frame = lambda f: DataFrame(bytes(f["header"]), bytes(f["body"]))
        return Payload([frame(f) for f in o["data"]], bytes(o["schema"]), [frame(f) for f in o.get("data2", [])],
                       bytes(o.get("schema2", [])), Uuid(**o["uuid"]), Encoding(o["encoding"]), o.get("datasource"),
                       o.get("query_number"), o.get("shuffle_id"), o.get("metadata"))

    def is_empty_data(self) -> bool:
        return not self.data and not self.data2

    def to_record_batch(self, ctx) -> Tuple[List[DeviceBatch], List[DeviceBatch]]:
        """`Payload::to_record_batch` (payload.rs:134-170): unmarshal (decompress) + Flight data -> batches, on the device."""
        def side(frames, schema_bytes):
            if not frames:
                return []
            schema = schema_from_bytes(schema_bytes)
            fields = [(f.name, _kind_of(f.type)) for f in schema]
            return [flight_data_to_batch(ctx, self.encoding.decompress(f.header), self.encoding.decompress(f.body), fields)
                    for f in frames]
        return side(self.data, self.schema), side(self.data2, self.schema2)


def _kind_of(t) -> str:
    import pyarrow as pa
    for k, a in (("int32", pa.int32()), ("int64", pa.int64()), ("uint64", pa.uint64()), ("float64", pa.float64()),
                 ("timestamp_ms", pa.timestamp("ms")), ("utf8", pa.utf8())):
        if t == a:
            return k
    raise NotImplementedError(f"Arrow type {t}")


def to_payload(ctx, batch1: Sequence[DeviceBatch], batch2: Sequence[DeviceBatch], uuid: Uuid, sync: bool,
               encoding: Optional[Encoding] = None, validity: bool = True) -> Payload:
    """`to_payload` (transmute.rs:176-216) for batches in HBM."""
    encoding = encoding or Encoding()

    def frames(batches):
        out = []
        for b in batches:
            header, body = batch_to_flight_data(ctx, b, keep_view=encoding.name != "None", validity=validity)
            out.append(DataFrame(encoding.compress(header), encoding.compress(body)))
        return out
    p = Payload(uuid=uuid, encoding=encoding, datasource={"Payload": bool(sync)})
    if batch1:
        p.data, p.schema = frames(batch1), schema_to_bytes(batch1[0].arrow_schema())
    if batch2:
        p.data2, p.schema2 = frames(batch2), schema_to_bytes(batch2[0].arrow_schema())
    return p
