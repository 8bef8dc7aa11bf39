#!/bin/bash
# PMC view of the talking-heads row kernels (run on the GPU box): VALU instructions / busy cycles / wait cycles per kernel
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/thpmc
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/th_rows_bench.py > $OUT/$tag.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True)
if not f:
    print("no csv for $tag")
else:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        for key in ("th_softmax_bwd_kernel", "th_softmax_fwd_kernel"):
            if key in n:
                d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
done
rm -rf $OUT
