#!/usr/bin/env python3
"""Writes tests/golden/payload_sizes.json from the two CSV files the reference's payload tests read
(flock/src/runtime/payload.rs:288-309 and :312-402): only the SHAPE of each batch -- row count and, per column, its Arrow
type and the total byte length of its values -- plus the Arrow Flight size the reference asserts for it.  Run in the
development container (needs /root/reference)."""
import csv
import json
import os

REF = "/root/reference/flock/src/tests/data"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shape(path, types):
    rows = list(csv.reader(open(os.path.join(REF, path))))[1:]
    cols = []
    for c, t in enumerate(types):
        cols.append({"type": t, "value_bytes": sum(len(r[c].encode()) for r in rows) if t == "utf8" else None})
    return {"rows": len(rows), "columns": cols}


out = {
    "uk_cities": dict(shape("uk_cities_with_headers.csv", ["utf8", "float64", "float64"]), reference_flight_data_size=1856,
                      reference_assert="flock/src/runtime/payload.rs:309"),
    "citibike": dict(shape("JC-202011-citibike-tripdata.csv",
                           ["utf8", "utf8", "utf8", "int32", "utf8", "float64", "float64", "int32", "utf8", "float64", "float64", "int32",
                            "utf8", "int32", "int8"]), reference_flight_data_size=3453248, reference_assert="flock/src/runtime/payload.rs:402"),
}
# payload.rs:365-372: the same citibike batch through arrow's json::LineDelimitedWriter = one compact serde_json object per row
CITI_NAMES = next(csv.reader(open(os.path.join(REF, "JC-202011-citibike-tripdata.csv"))))
CITI_TYPES = ["utf8", "utf8", "utf8", "int32", "utf8", "float64", "float64", "int32", "utf8", "float64", "float64", "int32", "utf8", "int32", "int8"]


def json_lines(rows):
    lines = []
    for r in rows:
        o = {n: (v if t == "utf8" else (float(v) if t == "float64" else int(v))) for n, t, v in zip(CITI_NAMES, CITI_TYPES, r)}
        lines.append(json.dumps(o, separators=(",", ":"), ensure_ascii=False))
    return lines


citi_rows = list(csv.reader(open(os.path.join(REF, "JC-202011-citibike-tripdata.csv"))))[1:]
lines = json_lines(citi_rows)
out["citibike"]["json_lines_bytes"] = sum(len(l.encode()) + 1 for l in lines)
out["citibike"]["reference_json_lines_bytes"] = 9436023
out["citibike"]["reference_json_assert"] = "flock/src/runtime/payload.rs:372"
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "payload_sizes.json"), "w"), indent=1)
# small data fixtures for the GPU tests (the GPU box has no reference tree): the 37 UK cities the reference's arena / payload
# tests read, and the first 256 citibike rows as the JSON lines arrow's writer would emit
uk = list(csv.reader(open(os.path.join(REF, "uk_cities_with_headers.csv"))))[1:]
json.dump({"city": [r[0] for r in uk], "lat": [float(r[1]) for r in uk], "lng": [float(r[2]) for r in uk]},
          open(os.path.join(ROOT, "tests", "golden", "uk_cities.json"), "w"), indent=0)
with open(os.path.join(ROOT, "tests", "golden", "citibike_head.jsonl"), "w") as f:
    f.write("\n".join(lines[:256]) + "\n")
print(json.dumps(out)[:300])
