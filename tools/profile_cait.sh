# kernel stats + JSON line of BASELINE config 4 (CaiT-S24, 256 images) on HEAD's library -> gpurun_out/prof_cait/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cait
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c -o c -- python3 $GRAFT_REPO_ROOT/bench.py --model cait_s_24 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/bench.json 2> $OUT/err.log
cp $(find $OUT/c -name "*kernel_stats.csv" | head -1) $OUT/r06_cait_s24_kernel_stats.csv
tail -1 $OUT/bench.json > $OUT/r06_cait_s24_bench.json
rm -rf $OUT/c
cd $GRAFT_REPO_ROOT
python3 bench.py --model cait_s_24 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/unprofiled.json 2>> $OUT/err.log
python3 -c "
import json
for f in ('r06_cait_s24_bench.json','unprofiled.json'):
    b=json.load(open('$OUT/'+f)); print(f, b['value'], b['ms_per_step'])
"
