#!/bin/bash
# FETCH_SIZE of the grouped weight-gradient kernel for experiment builds (ring depth): run on the GPU box.  usage: wgrad_fetch_probe.sh NAME...
set -u
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/wgf_$v
  rm -rf $OUT; mkdir -p $OUT
  SAVIT_EXP_LIB=$v rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o f -- python3 $GRAFT_REPO_ROOT/tools/bench_wgrad_group.py > $OUT/run.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemm_wgrad_group_kernel" in r["Kernel_Name"]:
        d["group"].append(float(r["Counter_Value"]))
v = d["group"]
print("$v", "launches", len(v), "FETCH (doubled) per launch MB", round(2 * sum(v) / len(v) * 1024 / 1e6, 1))
PY
  grep "round 2" $OUT/run.log | cut -c1-150
  rm -rf $OUT
done
