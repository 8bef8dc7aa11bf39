set -e
mkdir -p gpurun_out/r6g
L=gpurun_out/r6g/ab_sc1.log
for lib in sc0 sc1 sc0 sc1; do
  echo "=== lib=$lib" >> $L
  SAVIT_EXP_LIB=$lib python tools/profile_step.py vit_b_patch16 128 2>&1 | grep -E "^(qkv|proj|fc1|fc2|fc2.dgrad|fc1.dgrad|proj.dgrad|qkv.dgrad|ln1|ln2|attn|attn.bwd|ln1.bwd|ln2.bwd|sum) " >> $L
  SAVIT_EXP_LIB=$lib python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['kernel_breakdown_ms']['sum_step_net_of_brackets'])" >> $L
done
