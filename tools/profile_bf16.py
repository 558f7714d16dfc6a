"""profiles/<tag>_bf16_summary.md from gpurun_out/prof_<tag>_bf16 (tools/profile_bf16.sh)."""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_bf16")


def line(fn):
    return json.loads([x for x in open(os.path.join(src, fn)) if x.startswith("{")][-1])


prof, plain = line("bench_line.json"), line("bench_line_unprofiled.json")
steps = prof["steps"] + prof["warmup"] + 0
rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"# Round {tag[1:].lstrip('0')}, bf16 compute mode (`bench.py --dtype bf16`), 1x MI355X: kernels of the profiled run", "",
       "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --dtype bf16 --steps 20 --warmup 5 "
       "--no-extra-legs --no-cpu-baseline --no-class` (`tools/profile_bf16.sh`; all launches of the process, warm-up included).",
       f"Bench line of the profiled run: {prof['value']} seq/s, {prof['ms_per_step']} ms/step; unprofiled, same box: "
       f"{plain['value']} seq/s, {plain['ms_per_step']} ms/step.  Saved gates are bf16 `[T,B,H,4]` in this mode (DESIGN 5.6).", "",
       "| kernel | launches | avg us | total ms | % of kernel time |", "|---|---|---|---|---|"]
for r in rows[:24]:
    out.append(f"| `{r['Name'].replace('void ', '')[:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
               f"{float(r['TotalDurationNs']) / 1e6:.2f} | {100 * float(r['TotalDurationNs']) / tot:.1f} |")
open(os.path.join(root, "profiles", f"{tag}_bf16_summary.md"), "w").write("\n".join(out) + "\n")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(root, "profiles", f"{tag}_bf16_kernel_stats.csv"))
json.dump(plain, open(os.path.join(root, "profiles", f"{tag}_bench_line_bf16.json"), "w"))
print("\n".join(out[:14]))
