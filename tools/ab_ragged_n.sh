# same-box A/B of the 320 x 256 / 256 x 256 ping-pong tiles on N = 1152, K = 384 (qkv of the d = 384 models): experiment build `rag`
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
L=$O/ab_ragged_n.log
: > $L
export SAVIT_EXP_LIB=rag
for rep in 1 2; do
for v in 0 1 2; do
  for m in "vit_s_patch16 256" "cait_s_24 256"; do
    set -- $m
    SAVIT_PP_RAGGED_N=$v python3 bench.py --model $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/rag.json 2>/dev/null
    python3 -c "
import json; b=json.load(open('$O/rag.json')); print('ragged_n=$v', '$1', b['value'], 'img/s', b['ms_per_step'], 'ms')" >> $L
  done
done
done
for v in 0 1 2; do
  echo "=== labels, ragged_n=$v" >> $L
  SAVIT_PP_RAGGED_N=$v python3 tools/profile_step.py vit_s_patch16 256 2>&1 | grep -E "^(qkv|proj|fc1|sum) " >> $L
done
cat $L
