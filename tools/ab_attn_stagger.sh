set -e
mkdir -p gpurun_out/r6e
L=gpurun_out/r6e/ab_attn_stagger.log
for lib in afl st6 st12 st20 afl st6 st12 st20; do
  echo "=== lib=$lib" >> $L
  SAVIT_EXP_LIB=$lib python tools/attn_bench.py 2>&1 | grep "^B=" >> $L
done
