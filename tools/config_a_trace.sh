#!/bin/bash
# Where the reference-default configuration at batch 32 spends its step: rocprofv3 kernel trace of the
# config-A / batch-32 leg, eager and replayed from the captured hipGraph: kernel time vs span per step, launch cadence.
#   bash tools/config_a_trace.sh        (on the GPU box; output gpurun_out/ca/{eager,graph}.txt)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/ca
rm -rf $OUT; mkdir -p $OUT
for mode in eager graph; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$mode -o t -- python - $mode > $OUT/$mode.line 2> $OUT/$mode.err <<'PY'
import sys, types, json
sys.argv = ['bench.py'] + sys.argv[1:]
import torch
import bench
a = types.SimpleNamespace(cell='gru', steps=60, warmup=10)
r = bench.train_leg(a, torch.device('cuda:0'), 0, 1, 'f32', 80, 1, 32, 25, 60, 10, graph=(sys.argv[1] == 'graph'), z_dim=100)
print(json.dumps({k: r[k] for k in ('value', 'ms_per_step', 'host_enqueue_ms_per_step')}))
PY
  tail -1 $OUT/$mode.line > $OUT/$mode.txt
  python tools/step_timeline.py $OUT/$mode >> $OUT/$mode.txt 2>&1
  python - $OUT/$mode >> $OUT/$mode.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: r['Kernel_Name'].replace('void ', '')
adam = [i for i, r in enumerate(rows) if nm(r).startswith('adam_segs')] or [i for i, r in enumerate(rows) if nm(r).startswith('adam_step')][2::3]
lo, hi = adam[-21] + 1, adam[-1] + 1
ks = rows[lo:hi]
kt = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks) / 20e3
span = (int(ks[-1]['End_Timestamp']) - int(ks[0]['Start_Timestamp'])) / 20e3
print(f"# last 20 steps: {len(ks) / 20:.0f} launches/step, kernel time {kt:.1f} us/step, span {span:.1f} us/step, mean start-to-start {span / (len(ks) / 20):.2f} us")
PY
done
head -1 $OUT/eager.txt $OUT/graph.txt; tail -1 $OUT/eager.txt $OUT/graph.txt
