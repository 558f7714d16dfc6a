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
    grids = collections.defaultdict(set)   # kernel -> grid sizes it was launched with: one instantiation serving two launch shapes (the
    #                                        BPTT step: both encoder directions / the decoder's one) also gets a row per shape, "name @grid=N"
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = rd.fieldnames or []
            if "Counter_Name" in cols:
                per = collections.defaultdict(float)
                grid_of = {}
                for r in rd:
                    per[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
                    grid_of[(short(r["Kernel_Name"]), r["Dispatch_Id"])] = r.get("Grid_Size", "")
                for (k, d, c), v in per.items():
                    vals[k][c].append((int(d), v))
                    grids[k].add(grid_of[(k, d)])
                    vals[k + " @grid=" + grid_of[(k, d)]][c].append((int(d), v))
            elif "Start_Timestamp" in cols and "Kernel_Name" in cols:
                for r in rd:
                    k, us = short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    gs = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
                    durs[k].append((int(r["Dispatch_Id"]), us))
                    durs[k + " @grid=" + gs].append((int(r["Dispatch_Id"]), us))
                    grids[k].add(gs)
    out = {}
    names = sorted(set(vals) | set(durs))
    names = [k for k in names if " @grid=" not in k or len(grids[k.split(" @grid=")[0]]) > 1]   # per-shape rows only where shapes differ
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
