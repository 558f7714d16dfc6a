#!/usr/bin/env python3
"""Summarise rocprofv3 outputs: per-kernel mean counter values from *_counter_collection.csv files and per-kernel
duration statistics from *_kernel_trace.csv files.

  python tools/pmc_summary.py [--match SUBSTR ...] [--json OUT] DIR_OR_CSV ...

Kernel names are shortened to their template head.  With --json the per-kernel means are also written as
{kernel: {counter: mean_per_dispatch, "dispatches": n, "avg_us": t}} - profiles/*_pmc.json files are made this way and
bench.py reads `roofline.traffic` from them (FETCH_SIZE / WRITE_SIZE are in KB; see MI355X_MICROARCH.md, HBM section).
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--match", action="append", default=[])
    ap.add_argument("--json")
    ap.add_argument("--skip-first", type=int, default=0, help="ignore the first N dispatches of every kernel (warm-up)")
    ap.add_argument("--stamp-csrc", action="store_true",
                    help="store bench.csrc_fingerprint() as \"_csrc_sha256_16\": bench.py reports traffic from a profile only when "
                         "it was taken from the kernel sources it runs")
    a = ap.parse_args()
    files = []
    for p in a.paths:
        if os.path.isdir(p):
            files += sorted(glob.glob(os.path.join(p, "**", "*.csv"), recursive=True))
        else:
            files.append(p)
    vals = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> [per-dispatch values]
    durs = collections.defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = rd.fieldnames or []
            if "Counter_Name" in cols:
                per = collections.defaultdict(float)
                for r in rd:
                    per[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
                for (k, d, c), v in per.items():
                    vals[k][c].append((int(d), v))
            elif "Start_Timestamp" in cols and "Kernel_Name" in cols:
                for r in rd:
                    durs[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    out = {}
    names = sorted(set(vals) | set(durs))
    for k in names:
        if a.match and not any(m in k for m in a.match):
            continue
        row = {}
        d = [v for _, v in sorted(durs.get(k, []))][a.skip_first:]
        if d:
            row["dispatches"] = len(d)
            row["avg_us"] = sum(d) / len(d)
            row["min_us"] = min(d)
        for c, lst in sorted(vals.get(k, {}).items()):
            v = [x for _, x in sorted(lst)][a.skip_first:]
            if v:
                row[c] = sum(v) / len(v)
                row.setdefault("dispatches", len(v))
        if row:
            out[k] = row
    for k, row in out.items():
        print(k)
        for c, v in row.items():
            print(f"    {c:32s} {v:16.1f}")
    if a.json:
        if a.stamp_csrc:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            out["_csrc_sha256_16"] = bench.csrc_fingerprint()
        with open(a.json, "w") as fh:
            json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
