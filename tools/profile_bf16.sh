#!/bin/bash
# rocprofv3 kernel table of the bf16 compute mode's training leg (bench.py --dtype bf16), on the GPU box:
#   bash tools/profile_bf16.sh r03   -> gpurun_out/prof_<tag>_bf16/{kernel_stats.csv, bench_line.json}; then locally
#   python tools/profile_bf16.py r03 -> profiles/<tag>_bf16_summary.md, profiles/<tag>_bf16_kernel_stats.csv
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_${TAG}_bf16
mkdir -p $OUT
export TMPDIR=/tmp CPG_BENCH_NO_TORCH_PROFILER=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-class > $OUT/bench_line.json 2> $OUT/bench.err
find $OUT/trace -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-class > $OUT/bench_line_unprofiled.json 2>/dev/null
echo done
