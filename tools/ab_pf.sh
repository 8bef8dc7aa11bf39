set -e
mkdir -p gpurun_out/r6b
export SAVIT_EXP_LIB=pf
for cfg in "0 2 0" "1 2 0" "1 2 2" "1 0 0" "1 4 0" "0 2 0" "1 2 0"; do
  set -- $cfg
  echo "=== prefetch=$1 lead=$2 policy=$3" >> gpurun_out/r6b/ab_pf.log
  SAVIT_PP_PREFETCH=$1 SAVIT_PP_PREFETCH_LEAD=$2 SAVIT_PP_PREFETCH_POLICY=$3 python tools/profile_step.py vit_b_patch16 128 2>&1 | grep -E "^(proj|fc2|fc2.dgrad|fc1|qkv|fc1.dgrad|proj.dgrad|qkv.dgrad|ln2|ln1|sum) " >> gpurun_out/r6b/ab_pf.log
done
unset SAVIT_EXP_LIB
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" -x 2>&1 | tail -3 >> gpurun_out/r6b/ab_pf.log
