#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output directories (kernel-trace stats and --pmc passes) into per-kernel averages.

usage: tools/summarize_pmc.py <dir> [<dir> ...] [--json out.json]
Finds every *counter_collection.csv and *kernel_trace.csv below the directories.  Counter values are averaged per dispatch
and per kernel name (template arguments kept).  FETCH_SIZE/WRITE_SIZE are reported in bytes with the gfx950
corrections of /opt/skills/guides/MI355X_MICROARCH.md (section HBM): the counters are in KiB, and FETCH_SIZE
reads exactly half of a wide coalesced stream, so 'fetch_bytes_x2' is the figure to compare with a byte count.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void nl::", "").replace("nl::", "")
    return name.split("(")[0][:60]


def main():
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    roots = [a for a in sys.argv[1:] if a != "--json" and a != out_json]
    counters = defaultdict(lambda: defaultdict(list))
    for path in [p for r in roots for p in glob.glob(os.path.join(r, "**", "*counter_collection.csv"), recursive=True)]:
        with open(path, newline="") as f:
            rd = csv.DictReader(f)
            per_dispatch = defaultdict(float)
            for row in rd:
                k = row.get("Kernel_Name") or row.get("kernel_name") or "?"
                c = row.get("Counter_Name") or row.get("counter_name")
                v = float(row.get("Counter_Value") or row.get("counter_value") or 0)
                d = row.get("Dispatch_Id") or row.get("dispatch_id") or "0"
                per_dispatch[(k, c, d)] += v  # one row per (dispatch, counter, dimension instance)
            for (k, c, d), v in per_dispatch.items():
                counters[short(k)][c].append(v)
    durations = defaultdict(list)
    for path in [p for r in roots for p in glob.glob(os.path.join(r, "**", "*kernel_trace.csv"), recursive=True)]:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                try:
                    k = row.get("Kernel_Name") or "?"
                    durations[short(k)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
                except (KeyError, ValueError):
                    pass
    summary = {}
    for k in sorted(set(counters) | set(durations)):
        e = {}
        if durations.get(k):
            d = durations[k]
            e["launches"] = len(d)
            e["avg_us"] = sum(d) / len(d)
            e["min_us"] = min(d)
        for c, vals in counters.get(k, {}).items():
            e[c] = sum(vals) / len(vals)
            e[c + "_n"] = len(vals)
        if "FETCH_SIZE" in e:
            e["fetch_bytes_x2"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
        summary[k] = e
    for k, e in summary.items():
        print(k)
        for kk, v in e.items():
            print(f"    {kk:28s} {v:,.3f}" if isinstance(v, float) else f"    {kk:28s} {v}")
    if out_json:
        json.dump(summary, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
