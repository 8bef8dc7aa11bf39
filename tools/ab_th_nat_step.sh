# same-box A/B/A/B of the CaiT-S24 step with the 8-byte (nat0) and 16-byte (nat1) P' / dS fragment loads
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
L=$O/ab_th_nat_step.log
: > $L
for v in nat0 nat1 nat0 nat1; do
  SAVIT_EXP_LIB=$v python3 bench.py --model cait_s_24 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/step_$v.json 2>/dev/null
  python3 -c "
import json; b=json.load(open('$O/step_$v.json')); print('$v', b['value'], 'img/s', b['ms_per_step'], 'ms')" >> $L
done
cat $L
