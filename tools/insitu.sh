#!/bin/bash
# In-situ A/B on the GPU box: the training step (bench.py headline leg only) with each diagnostic library under build_variants/
# and with option sets given as arguments ("tn_tile=192x128 tn_split=8" ...): ms per step and the per-launch averages of the
# recurrent kernel families AS THEY RUN INSIDE THE STEP - a synthetic per-kernel micro-benchmark (tools/kb.py) has pointed the
# other way more than once.   bash tools/insitu.sh ["opt=val opt=val" ...]
cd "$(dirname "$0")/.."
one() {
  python bench.py --steps ${INSITU_STEPS:-10} --warmup 3 --no-extra-legs --no-class --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); f=[d['roofline']]+d['extra']['families']
print('%-28s %7.3f ms/step  %s' % (sys.argv[1], d['ms_per_step'], {x['family']: x.get('us', x.get('avg_launch_us')) for x in f}))" "$1"
}
one current
for so in build_variants/libcpg_*.so; do
  [ -e "$so" ] || continue
  CPG_LIB_PATH=$PWD/$so one "$(basename $so .so)"
done
for opts in "$@"; do
  env $(for kv in $opts; do k=${kv%%=*}; echo "CPG_$(echo $k | tr a-z A-Z)=${kv#*=}"; done) bash -c "$(declare -f one); one '$opts'"
done
one current
