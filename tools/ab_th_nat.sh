# A/B of the 16-byte natural-order P' / dS fragment loads of th_pv and the scores backward (round 6): experiment builds nat0 / nat1
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
L=$O/ab_th_nat.log
: > $L
python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "talking" 2>&1 | tail -3 >> $L
for v in nat0 nat1 nat0 nat1; do
  echo "=== $v" >> $L
  SAVIT_EXP_LIB=$v python tools/th_bench.py 2>&1 | grep "^B=" >> $L
done
for v in nat0 nat1; do
  echo "=== per kernel, $v" >> $L
  (cd /tmp && SAVIT_EXP_LIB=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o th -- python3 $GRAFT_REPO_ROOT/tools/th_bench.py > /dev/null 2>&1)
  python - $O/prof_$v >> $L <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "th_" in r["Name"]:
        print(f"  {float(r['AverageNs']) / 1e3:8.1f} us x {r['Calls']:>4}  {r['Name'][:90]}")
PY
done
