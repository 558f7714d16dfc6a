set -x
mkdir -p gpurun_out/r3b
python tools/kb.py > gpurun_out/r3b/kb.log 2>&1
for o in "tn_tile=192x128" "tn_tile=192x128 tn_split=16" "tn_tile=256x128 tn_split=8" "tn_tile=256x128 tn_split=10"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python tools/kb.py --only wgrad $args >> gpurun_out/r3b/kb.log 2>&1
  python tools/kb.py --only wgrad --dtype bf16 $args >> gpurun_out/r3b/kb.log 2>&1
done
cat gpurun_out/r3b/kb.log | grep "^\["
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r3b/pytest_all.log 2>&1; tail -25 gpurun_out/r3b/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3b/bench_n1.json 2> gpurun_out/r3b/bench_n1.err; tail -3 gpurun_out/r3b/bench_n1.err; head -c 600 gpurun_out/r3b/bench_n1.json
