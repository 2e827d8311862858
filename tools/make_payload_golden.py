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
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "payload_sizes.json"), "w"), indent=1)
print(json.dumps(out)[:300])
