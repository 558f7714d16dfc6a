#!/bin/bash
# rocprofv3 kernel trace of the training leg only (no CPU baseline, no CLaSS leg): per-kernel totals of the timed steps.
#   bash tools/quick_trace.sh [extra bench flags]      (on the GPU box; output under gpurun_out/qt/)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/qt
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-class --no-extra-legs "$@" > $OUT/line.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/qt/trace/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
nm = lambda r: r['Kernel_Name'].replace('void ', '')
adam = [i for i, r in enumerate(rows) if nm(r).startswith('adam_segs')]     # one per step (round 6); older libraries: three adam_step launches
per = 1
if not adam:
    adam, per = [i for i, r in enumerate(rows) if nm(r).startswith('adam_step')], 3
lo, hi = adam[per * 4 - 1] + 1, adam[-1] + 1
n = len(adam) // per - 4
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    k = r['Kernel_Name'].replace('void ', '')
    k = k[:k.find('(')] if '(' in k else k
    agg[k[:110]][0] += 1
    agg[k[:110]][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
span = (int(rows[hi - 1]['End_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3
print(f"steps {n}  kernel time {tot / n:.1f} us/step  span {span / n:.1f} us/step  launches/step {sum(v[0] for v in agg.values()) / n:.0f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1] / n:8.1f} us/step  x{v[0] / n:5.1f}  avg {v[1] / v[0]:7.1f}  {k}")
PY
python tools/step_timeline.py $OUT/trace > $OUT/timeline.txt 2>&1 || tail -3 $OUT/timeline.txt
