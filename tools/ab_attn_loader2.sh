set -e
mkdir -p gpurun_out/r6e
L=gpurun_out/r6e/ab_attn_loader2.log
for lib in al alp alk alpk al alp alk alpk; do
  echo "=== lib=$lib" >> $L
  SAVIT_EXP_LIB=$lib python tools/attn_bench.py 2>&1 | grep "^B=" >> $L
done
SAVIT_EXP_LIB=alpk python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention_bwd" -x 2>&1 | tail -2 >> $L
