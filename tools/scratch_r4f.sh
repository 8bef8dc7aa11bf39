mkdir -p gpurun_out/r4f
for n in 0 1 2 3 8 4 5 6 7 0; do
  SAVIT_EXP_LIB=abl$n python tools/profile_step.py > gpurun_out/r4f/prof_abl$n.log 2>&1
  echo "abl$n: $(grep -E '^fc1 |^fc2.dgrad|^qkv  |^fc1.dgrad' gpurun_out/r4f/prof_abl$n.log | awk '{printf "%s %s us | ", $1, $5}')"
done
