#!/bin/bash
# Kernel table + PMC readings of ONE bench mode's training leg on the GPU box (the named configurations next to the headline):
#   bash tools/profile_mode.sh r04 bf16 --dtype bf16          bash tools/profile_mode.sh r04 lstm --cell lstm
#   bash tools/profile_mode.sh r04 lstm_bf16 --cell lstm --dtype bf16
# -> gpurun_out/prof_<tag>_<name>/{kernel_stats.csv, bench_line.json, bench_line_unprofiled.json, pmc.json}; then locally
#   python tools/profile_mode.py r04 bf16   -> profiles/<tag>_<name>_summary.md (+ _kernel_stats.csv)
set -e
TAG=$1; NAME=$2; shift 2
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_${TAG}_${NAME}
mkdir -p $OUT
export TMPDIR=/tmp CPG_BENCH_NO_TORCH_PROFILER=1
ARGS="$* --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-class"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/bench_line.json 2> $OUT/bench.err
find $OUT/trace -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python bench.py $ARGS > $OUT/bench_line_unprofiled.json 2>/dev/null
bash tools/pmc_run.sh $OUT/pmc -- python bench.py $* --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline --no-class > $OUT/pmc.log 2>&1
python tools/pmc_summary.py --skip-first 2 --json $OUT/pmc.json $OUT/pmc > $OUT/pmc_summary.txt
echo "$ARGS" > $OUT/args.txt
echo done
