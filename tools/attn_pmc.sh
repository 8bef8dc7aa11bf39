# PMC view of the attention kernels at DeiT-B's layer (round 6): wave cycles, waits, VALU / LDS / MFMA activity.  Run on the GPU box from the repo root.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc1 -o a -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -o b -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py > $OUT/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $OUT/attn_pmc.log 2>&1
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6l")
for d in ("pmc1", "pmc2"):
    f = glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        print(d, "no counter file", glob.glob(os.path.join(out, d, "**", "*.csv"), recursive=True)); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "attn_" not in k: continue
        import re
        k = re.search(r"(attn_\w+<\d+>)", k).group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(d, k)
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {x / n[(k, c)]:16.0f} per launch")
PY
rm -rf $OUT/pmc1 $OUT/pmc2
cat $OUT/attn_pmc.log
