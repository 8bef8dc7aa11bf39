set -e
mkdir -p gpurun_out/r6i
L=gpurun_out/r6i/ab_rowgroup.log
export SAVIT_EXP_LIB=rg
for g in 0 1 2 3 4 6 8 12 0; do
  echo "=== row group $g (0 = the library's choice: 8 at K = 768, 4 for GELU', 1 at K = 3072)" >> $L
  SAVIT_PP320_ROW_GROUP=$g python tools/profile_step.py vit_b_patch16 128 2>&1 | grep -E "^(qkv|proj|fc1|fc2|fc2.dgrad|fc1.dgrad|proj.dgrad|qkv.dgrad|sum) " >> $L
done
