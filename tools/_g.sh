one() {
  (python bench.py --steps 40 --warmup 10 --no-extra-legs --no-cpu-baseline --no-class 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])")
}
for i in 1 2 3; do
  X_PREDRAW=1 X_MMDSIDE=1 one both
  X_PREDRAW=0 X_MMDSIDE=1 one mmd_only
  X_PREDRAW=1 X_MMDSIDE=0 one predraw_only
  X_PREDRAW=0 X_MMDSIDE=0 one neither
done
