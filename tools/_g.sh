one() {
  (cd $1 && python bench.py --steps 40 --warmup 10 --no-extra-legs --no-cpu-baseline --no-class 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$2', d['value'], d['ms_per_step'])")
}
for i in 1 2 3; do
  one $GRAFT_REPO_ROOT new
  one $GRAFT_REPO_ROOT/build_variants/prev prev
done
