#!/bin/bash
# One command that regenerates the round's profile evidence on the GPU box (outputs under gpurun_out/prof_<tag>/; copy the
# summaries into profiles/).   bash tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the judged line)  -> kernel stats csv + bench line
#   2. PMC passes (tools/pmc_run.sh: SQ / LDS / FETCH_SIZE / WRITE_SIZE, separate runs) over the training leg only
#      -> per-kernel counter means as JSON (profiles/<tag>_pmc.json is what bench.py reads `roofline.traffic` from)
set -e
TAG=${1:-r02}
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
bash tools/pmc_run.sh $OUT/pmc -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-class > $OUT/pmc.log 2>&1
python tools/pmc_summary.py --skip-first 2 --json $OUT/${TAG}_pmc.json $OUT/pmc > $OUT/pmc_summary.txt
find $OUT/trace -name "*_kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
echo done
