"""profiles/<tag>_<name>_summary.md from gpurun_out/prof_<tag>_<name> (tools/profile_mode.sh): the kernel table of one bench mode
(bf16 compute mode, the LSTM extension, ...) and, for its recurrent kernels, what the PMC counters say about the roof that binds
them: HBM bytes per launch ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, gfx950 read-side correction of MI355X_MICROARCH.md) against the
launch time, beside MFMA busy and the share of wave cycles spent waiting.   python tools/profile_mode.py r04 bf16"""
import csv
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_{name}")


def line(fn):
    return json.loads([x for x in open(os.path.join(src, fn)) if x.startswith("{")][-1])


prof, plain = line("bench_line.json"), line("bench_line_unprofiled.json")
args = open(os.path.join(src, "args.txt")).read().strip()
rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"# Round {tag[1:].lstrip('0')}, `{name}` mode (`bench.py {args.split(' --steps')[0]}`), 1x MI355X", "",
       f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py {args}` (`tools/profile_mode.sh`; all "
       "launches of the process, warm-up included).",
       f"Bench line of the profiled run: {prof['value']} seq/s, {prof['ms_per_step']} ms/step; unprofiled, same box: "
       f"{plain['value']} seq/s, {plain['ms_per_step']} ms/step.", "", plain["config"]["workload"], "",
       "| kernel | launches | avg us | total ms | % of kernel time |", "|---|---|---|---|---|"]
for r in rows[:22]:
    out.append(f"| `{r['Name'].replace('void ', '')[:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
               f"{float(r['TotalDurationNs']) / 1e6:.2f} | {100 * float(r['TotalDurationNs']) / tot:.1f} |")
pmc = json.load(open(os.path.join(src, "pmc.json")))
out += ["", "## Which roof binds the recurrent kernels (PMC, separate passes: tools/pmc_run.sh; means per dispatch)", "",
        "HBM GB/s = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B / launch time; 8000 GB/s peak.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
        "GRBM_GUI_ACTIVE / 8).  A kernel far below BOTH roofs with a large waiting share is bound by latency / instruction issue, not by bytes.", "",
        "| kernel | avg us | HBM MB / launch | HBM GB/s | of HBM peak | MFMA busy | wave-cycles waiting | VALU insts / wave |", "|---|---|---|---|---|---|---|---|"]
for k, r in sorted(((k, v) for k, v in pmc.items() if not k.startswith("_")), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("dispatches", 0)):
    if not any(t in k for t in ("gru_s", "lstm_s", "gemm_kernel<TileCfg<256", "gemm_kernel<TileCfg<128, 128", "dgi_mfma")) or "FETCH_SIZE" not in r:
        continue
    mb = (2 * r["FETCH_SIZE"] + r.get("WRITE_SIZE", 0)) * 1024 / 1e6
    gbs = mb / 1e3 / (r["avg_us"] * 1e-6)
    busy = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1024 * r.get("GRBM_GUI_ACTIVE", 1) / 8, 1)
    out.append(f"| `{k[:100]}` | {r['avg_us']:.1f} | {mb:.1f} | {gbs:.0f} | {gbs / 8000:.2f} | {100 * busy:.0f} % | "
               f"{100 * r.get('SQ_WAIT_ANY', 0) / max(r.get('SQ_WAVE_CYCLES', 1), 1):.0f} % | {r.get('SQ_INSTS_VALU', 0) / max(r.get('SQ_WAVES', 1), 1):.0f} |")
open(os.path.join(root, "profiles", f"{tag}_{name}_summary.md"), "w").write("\n".join(out) + "\n")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(root, "profiles", f"{tag}_{name}_kernel_stats.csv"))
json.dump(plain, open(os.path.join(root, "profiles", f"{tag}_bench_line_{name}.json"), "w"))
print("\n".join(out))
