#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of tools/gpu_profile.sh from gpurun_out/prof/ into profiles/<tag>/ and rebuilds
profiles/traffic.json (corrected HBM bytes per launch of every query's dominant kernel, MI355X_MICROARCH.md HBM
section: FETCH_SIZE counts half of a 16 B / lane coalesced stream on gfx950 -> x2; WRITE_SIZE 1:1)."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
if tag.startswith("-"):
    sys.exit("usage: collect_profiles.py <tag, e.g. r05>")
src, dst = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in sorted(os.listdir(src)):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
DOMINANT = {5: "q5_count_kernel", 2: "q2_flag_kernel", 3: "q3_probe_flag_kernel", 8: "q8_sellers_bitmap_inline_kernel", 7: "q7_max_kernel", 9: "aq_final_kernel",
            13: "q13_flag_kernel"}
traffic = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs, "
                       "tools/gpu_profile.sh): bytes = 2 * FETCH_SIZE_KB * 1024 (gfx950 reports half of a 16 B/lane coalesced stream, "
                       "MI355X_MICROARCH.md HBM section) + WRITE_SIZE_KB * 1024.  Source CSVs: profiles/%s/q*_pmc_*.csv" % tag}
# (file prefix, kernel): the query rows, the "next" side entries and the general-path rows
ROWS = [(f"q{q}", kern) for q, kern in DOMINANT.items()] + [("q8", "q8_sellers_bitmap_kernel")] + [("q11", "sort_emit_kernel"), ("ysb", "ysb_count_kernel"), ("json", "json_parse_kernel"),
                                                            ("q3_general", "q3_probe_flag_kernel"), ("q8_general", "q8_key_bitmap_wide_kernel"),
                                                            ("q3_hash", "q3_window_join_lds_kernel"), ("q8_hash", "q8_sellers_part_kernel"),
                                                            ("arch", "pred_flag_kernel"), ("arch", "dense_group_kernel"), ("arch", "join_probe_unique_flag_kernel"), ("arch", "join_hash_probe_flag_kernel"), ("arch", "utf8_emit_long_kernel"),
                                                            ("expr", "valprog_kernel<false"), ("expr", "valprog_kernel<true"),
                                                            ("arch", "sort_emit_kernel"), ("arch", "utf8_emit_kernel"), ("arch", "gather_i32_kernel"),
                                                            ("q5_uniform", "q5_part_tile_kernel"), ("q4", "aq_final_kernel"),
                                                            ("q3_1e8", "q3_probe_flag_small_kernel")]
for q, kern in ROWS:
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(dst, f"{q}_pmc_{c}.csv")
        if not os.path.exists(p):
            continue
        tot, launches = 0.0, 0   # every instantiation of a template kernel, weighted by its launches (what bench.py's average is)
        for r in csv.DictReader(open(p)):
            match = kern + "(" in r["kernel"] or kern + "<" in r["kernel"] or ("<" in kern and kern in r["kernel"])
            if kern == "pred_flag_kernel":   # (the relation's one ragged tile is a launch of its own, `<false, ..>`: the full-tile instance is the pass)
                match = "pred_flag_kernel<true" in r["kernel"]
            if match:
                tot += float(r[f"avg_{c}_KB"]) * int(r["launches"])
                launches += int(r["launches"])
        if launches:
            vals[c] = tot / launches
    # bench.py's LaunchScope labels (the key it looks traffic up under): the inline / wide / small variants run under the plain names
    label = {"q8_key_bitmap_wide_kernel": "q8_sellers_bitmap_kernel", "q8_sellers_bitmap_inline_kernel": "q8_sellers_bitmap_kernel",
             "q3_probe_general_kernel": "q3_probe_count_kernel"}.get(kern, kern)
    name = label if not q.endswith(("_general", "_uniform", "_hash")) and q not in ("q4", "q3_1e8") else f"{label}@{q}"
    if q == "q3_1e8":   # (round 5: the small instance runs under its own LaunchScope label)
        name = "q3_probe_flag_small_kernel@1e8_events"
    arch_op = {"pred_flag_kernel": "filter", "dense_group_kernel": "groupby", "join_probe_unique_flag_kernel": "join", "join_hash_probe_flag_kernel": "join_sparse", "sort_emit_kernel": "sort"}.get(kern) if q == "arch" else None
    expr_row = {"valprog_kernel<false": "expr_project", "valprog_kernel<true": "expr_filter"}.get(kern) if q == "expr" else None
    if expr_row:   # the expression evaluator's two bench rows: one profiled run, the projection and the filter instance of one kernel
        name = f"valprog_kernel@{expr_row}"
    if q == "arch":   # the reference's operator harness: one profiled run holds all four plans
        name = f"{kern}@arch_{arch_op}" if arch_op else f"{kern}@arch"
    if len(vals) == 2 and name not in traffic:
        traffic[name] = int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024)
        detail = {"fetch_KB_raw": vals["FETCH_SIZE"], "write_KB": vals["WRITE_SIZE"]}
        try:   # the algorithmic bytes of the profiled run: bench.py attaches the entry only to a run of the same size
            under = json.load(open(os.path.join(dst, f"{q}_bench_under_rocprof.json")))
            detail["alg_bytes"] = (under.get("roofline") or {}).get("algorithmic_bytes_per_launch")
            if arch_op:
                detail["alg_bytes"] = ((under.get(arch_op) or {}).get("generic", {}).get("roofline") or {}).get("algorithmic_bytes_per_launch")
            if expr_row:
                detail["alg_bytes"] = ((under.get(expr_row) or {}).get("roofline") or {}).get("algorithmic_bytes_per_launch")
        except Exception:
            pass
        traffic[name + "_detail"] = detail
# rows whose PMC passes were not part of this collection keep the entry of the collection that had them (a partial re-profile -- QUERIES / SIDES /
# GENERALS of tools/gpu_profile.sh -- must not drop the others' traffic)
_path = os.path.join(ROOT, "profiles", "traffic.json")
if os.path.exists(_path):
    for k, v in json.load(open(_path)).items():
        traffic.setdefault(k, v)
json.dump(traffic, open(_path, "w"), indent=1)
print(json.dumps(traffic, indent=1))
