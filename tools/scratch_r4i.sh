mkdir -p gpurun_out/r4i
for v in "" w4s3 w4s4 ""; do
  SAVIT_EXP_LIB=$v python tools/bench_wgrad_group.py 768 3072 25216 2 > gpurun_out/r4i/wg_$v.log 2>&1
  echo "variant '$v':"; grep -E "round|max rel" gpurun_out/r4i/wg_$v.log | sed 's/per-weight launches.*| grouped/grouped/'
done
