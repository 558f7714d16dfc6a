#!/usr/bin/env python3
"""profiles/<tag>_bench_n1_summary.md from the outputs of tools/profile_round.sh (gpurun_out/prof_<tag>/): the bench line of
the profiled run, the per-kernel table of the training leg (calls / step, average, ms / step) from the rocprofv3 kernel trace,
the CLaSS leg's kernels, and the PMC readings of the recurrent kernels (traffic as bench.py reports it)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
out = os.path.join(root, "profiles")


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    i = n.find("(")
    return n if i < 0 else n[:i]


line = json.load(open(os.path.join(src, "bench_line.json")))
rows = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_trace.csv"))))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].replace("void ", "").startswith("adam_segs")]
per_step = 1   # one Adam launch per step since round 6
if not adam:   # older libraries: three adam_step launches per step (duplicate embedding segment twice + the rest)
    adam, per_step = [i for i, r in enumerate(rows) if r["Kernel_Name"].replace("void ", "").startswith("adam_step")], 3
steps_total = len(adam) // per_step
lo = adam[per_step * 5 - 1] + 1           # skip the 5 warm-up steps
hi = adam[-1] + 1
nsteps = steps_total - 5
def grid(r):
    return int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)


# one instantiation can serve two launch shapes (the eight-wave BPTT step: both encoder directions in one launch / the decoder's one
# direction): such kernels get a row per shape, labelled with the launch's workgroup count
shapes = collections.defaultdict(set)
for r in rows[lo:hi]:
    shapes[short(r["Kernel_Name"])].add(grid(r))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    k = short(r["Kernel_Name"])
    if len(shapes[k]) > 1:
        k += " [%d workgroups]" % (grid(r) // max(int(r.get("Workgroup_Size_X", 1) or 1), 1))
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
md = [f"# Round {int(tag[1:3])} profile ({tag}), 1x MI355X", "",
      "Regenerate: `bash tools/profile_round.sh %s` on the GPU box, then `python tools/profile_report.py %s` (this file, "
      "`%s_kernel_stats.csv`, `%s_pmc.json`)." % (tag, tag, tag, tag), "",
      "Command profiled: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 "
      "--no-extra-legs` (headline training leg + CPU baseline leg + CLaSS leg; the default command adds the bf16-mode and "
      "config-C legs and a sustained region, whose launches would mix into the per-step table).", "",
      "Bench line of the profiled run (the profiler costs a few %):", "", "```", json.dumps(line), "```", "",
      f"## Training leg: kernels of the {nsteps} timed steps (kernels on the side stream overlap in time)", "",
      "| kernel | launches / step | avg us | ms / step | % of kernel time |", "|---|---|---|---|---|"]
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    md.append(f"| `{k[:150]}` | {n / nsteps:.1f} | {us / n:.1f} | {us / nsteps / 1e3:.3f} | {100 * us / tot:.1f} |")
md += ["", f"Sum of kernel time {tot / nsteps / 1e3:.2f} ms per step; wall {line['ms_per_step']} ms per step.",
       f"Launches per step: {sum(v[0] for v in agg.values()) / nsteps:.0f}, of which torch / runtime glue (at::native::*, __amd_rocclr_*): "
       f"{sum(v[0] for k, v in agg.items() if k.startswith('at::') or k.startswith('__amd')) / nsteps:.0f} "
       f"({sum(v[1] for k, v in agg.items() if k.startswith('at::') or k.startswith('__amd')) / nsteps:.0f} us).", ""]
cl = collections.defaultdict(lambda: [0, 0.0])
for r in rows[hi:]:
    k = short(r["Kernel_Name"])
    cl[k][0] += 1
    cl[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
md += ["## CLaSS leg (BASELINE.json configs[3]: 1 M proposals per variant; warm-up round included)", "",
       "| kernel | launches | avg ms | total ms |", "|---|---|---|---|"]
for k, (n, us) in sorted(cl.items(), key=lambda kv: -kv[1][1])[:10]:
    md.append(f"| `{k[:120]}` | {n} | {us / n / 1e3:.3f} | {us / 1e3:.1f} |")
pmc = json.load(open(os.path.join(src, tag + "_pmc.json")))
md += ["", "## PMC readings of the recurrent kernels (separate passes: tools/pmc_run.sh; means per dispatch over the training leg)", "",
       "FETCH_SIZE / WRITE_SIZE in KB; `traffic` = (2 x FETCH + WRITE) x 1024 B as bench.py reports it (gfx950 read-side correction, "
       "MI355X_MICROARCH.md HBM section).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).", "",
       "| kernel | avg us | FETCH KB | WRITE KB | traffic MB | MFMA busy | VALU insts / wave | LDS bank-conflict share | wave-cycles waiting |",
       "|---|---|---|---|---|---|---|---|---|"]
pmc = {k: v for k, v in pmc.items() if not k.startswith("_")}
for k, r in sorted(pmc.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("dispatches", 0)):
    if not any(t in k for t in ("gru_", "gemm_kernel<TileCfg<256", "gemm_kernel<TileCfg<192", "lstm_")) or "FETCH_SIZE" not in r:
        continue
    busy = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1024 * r.get("GRBM_GUI_ACTIVE", 1) / 8, 1)
    md.append(f"| `{k[:110]}` | {r.get('avg_us', 0):.1f} | {r['FETCH_SIZE']:.0f} | {r.get('WRITE_SIZE', 0):.0f} | "
              f"{(2 * r['FETCH_SIZE'] + r.get('WRITE_SIZE', 0)) * 1024 / 1e6:.1f} | {100 * busy:.0f} % | "
              f"{r.get('SQ_INSTS_VALU', 0) / max(r.get('SQ_WAVES', 1), 1):.0f} | "
              f"{100 * r.get('SQ_LDS_BANK_CONFLICT', 0) / max(r.get('SQ_LDS_IDX_ACTIVE', 1), 1):.0f} % | "
              f"{100 * r.get('SQ_WAIT_ANY', 0) / max(r.get('SQ_WAVE_CYCLES', 1), 1):.0f} % |")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, f"{tag}_bench_n1_summary.md"), "w").write("\n".join(md) + "\n")
# launches per step of the profiled (eager) step go into the PMC json next to its source fingerprint: bench.py quotes them from there
pj = os.path.join(src, tag + "_pmc.json")
d = json.load(open(pj))
d["_launches_per_step"] = round(sum(v[0] for v in agg.values()) / nsteps, 1)
d["_torch_glue_launches_per_step"] = round(sum(v[0] for k, v in agg.items() if k.startswith('at::') or k.startswith('__amd')) / nsteps, 1)
json.dump(d, open(pj, "w"), indent=1, sort_keys=True)
for a, b in ((f"{tag}_kernel_stats.csv", f"{tag}_bench_n1_kernel_stats.csv"), (f"{tag}_pmc.json", f"{tag}_pmc.json"),
             ("bench_line.json", f"{tag}_bench_line.json"), ("bench_line_bf16.json", f"{tag}_bench_line_bf16.json")):
    p = os.path.join(src, a)
    if os.path.exists(p):
        open(os.path.join(out, b), "w").write(open(p).read())
print("\n".join(md[:60]))
