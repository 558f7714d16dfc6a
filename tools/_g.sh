set -x
mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_bf16.py tests/test_gpu_parity.py tests/test_gpu_tiles.py -q -x -m gpu > gpurun_out/r3k/pytest.log 2>&1; tail -15 gpurun_out/r3k/pytest.log
cd /tmp && export TMPDIR=/tmp
for st in 1 0; do
  rm -rf /tmp/rp$st
  CPG_BF16_STORE=$st rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp$st -o t -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-class > /dev/null 2>&1
  f=$(find /tmp/rp$st -name "*kernel_stats.csv" | head -1)
  echo "== bf16_store=$st"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:5]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["TotalDurationNs"])/1e6:9.2f} ms')
PY
done
cd $GRAFT_REPO_ROOT
for st in 1 0 1 0; do
    CPG_BF16_STORE=$st timeout 600 python bench.py --dtype bf16 --steps 40 --warmup 10 --no-extra-legs --no-cpu-baseline --no-class 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bf16_store=$st', d['value'], d['ms_per_step'])"
done
for i in 1 2 3; do
timeout 600 python bench.py --steps 40 --warmup 10 --no-extra-legs --no-cpu-baseline --no-class 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('f32', d['value'], d['ms_per_step'])"
done
