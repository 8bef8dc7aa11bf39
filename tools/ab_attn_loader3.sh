set -e
mkdir -p gpurun_out/r6e
L=gpurun_out/r6e/ab_attn_loader3.log
SAVIT_EXP_LIB=afl python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" -x 2>&1 | tail -2 >> $L
for v in "0 0" "1 1" "0 0" "1 1"; do
  set -- $v
  echo "=== fwd loader=$1 bwd loader=$2 (KV_EARLY build)" >> $L
  SAVIT_EXP_LIB=afl SAVIT_ATTN_FWD_LOADER=$1 SAVIT_ATTN_BWD_LOADER=$2 python tools/attn_bench.py 2>&1 | grep "^B=" >> $L
  SAVIT_EXP_LIB=afl SAVIT_ATTN_FWD_LOADER=$1 SAVIT_ATTN_BWD_LOADER=$2 python tools/attn_bench.py 256 197 6 64 2>&1 | grep "^B=" >> $L
  SAVIT_EXP_LIB=afl SAVIT_ATTN_FWD_LOADER=$1 SAVIT_ATTN_BWD_LOADER=$2 python tools/attn_bench.py 256 197 3 64 2>&1 | grep "^B=" >> $L
done
for v in "0 0" "1 1"; do
  set -- $v
  echo "=== step: fwd loader=$1 bwd loader=$2" >> $L
  SAVIT_EXP_LIB=afl SAVIT_ATTN_FWD_LOADER=$1 SAVIT_ATTN_BWD_LOADER=$2 python tools/profile_step.py vit_b_patch16 128 2>&1 | grep -E "^(attn|attn.bwd|sum) " >> $L
done
