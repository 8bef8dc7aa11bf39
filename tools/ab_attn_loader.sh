# A/B of the loader-wave attention backward (round 6): experiment build `al` reads SAVIT_ATTN_BWD_LOADER
set -e
mkdir -p gpurun_out/r6e
L=gpurun_out/r6e/ab_attn_loader.log
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" -x 2>&1 | tail -3 >> $L
export SAVIT_EXP_LIB=al
for v in 0 1 0 1; do
  echo "=== loader=$v" >> $L
  SAVIT_ATTN_BWD_LOADER=$v python tools/attn_bench.py >> $L 2>&1
  SAVIT_ATTN_BWD_LOADER=$v python tools/attn_bench.py 256 197 6 64 >> $L 2>&1
done
for v in 0 1; do
  echo "=== step, loader=$v" >> $L
  SAVIT_ATTN_BWD_LOADER=$v python tools/profile_step.py vit_b_patch16 128 2>&1 | grep -E "^(attn|attn.bwd|proj.dgrad|qkv.dgrad|sum) " >> $L
done
