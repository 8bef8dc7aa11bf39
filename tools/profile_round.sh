#!/bin/bash
# Round profiles of the headline workload (run on the GPU box from the repo root): rocprofv3 kernel stats (the one-stream schedule,
# the ViT engines' default since round 2, and the two-stream one) and the PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of the one-stream schedule, summarised into profiles/.
# Usage: bash tools/profile_round.sh r02      (writes gpurun_out/prof_<tag>/..., then copy the summaries into profiles/)
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs"
SAVIT_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- python3 $B > $OUT/serial_bench.json 2> $OUT/serial.err
echo serial done
SAVIT_OVERLAP_WGRAD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/overlap -o o -- python3 $B > $OUT/overlap_bench.json 2> $OUT/overlap.err
echo overlap done
P="$GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
SAVIT_OVERLAP_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $P > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo fetch done
SAVIT_OVERLAP_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $P > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo write done
SAVIT_OVERLAP_WGRAD=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o m -- python3 $P > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
echo mfma done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
python3 tools/mfma_util_summary.py $OUT/pmc_mfma $OUT/${TAG}_mfma_util.json > $OUT/mfma_summary.txt 2>&1
cp $(find $OUT/serial -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_serial_kernel_stats.csv
cp $(find $OUT/overlap -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_overlap_kernel_stats.csv
tail -1 $OUT/serial_bench.json > $OUT/${TAG}_serial_bench.json
tail -1 $OUT/overlap_bench.json > $OUT/${TAG}_overlap_bench.json
# the other BASELINE configs (2: DeiT-S 256 img, 4: CaiT-S24 256 img, 5: ViT-L/16-384 256 img): kernel stats of the same bench harness
cd /tmp
for cfgname in "deit_s --model vit_s_patch16 --batch 256 --steps 10 --warmup 3" "cait_s24 --model cait_s_24 --batch 256 --steps 5 --warmup 2" "vit_l384 --model vit_l_patch16 --img-size 384 --batch 256 --steps 4 --warmup 2"; do
  set -- $cfgname
  nm=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$nm -o c -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-other-configs > $OUT/${nm}_bench.json 2> $OUT/$nm.err
  echo $nm done
  cp $(find $OUT/$nm -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${nm}_kernel_stats.csv
  tail -1 $OUT/${nm}_bench.json > $OUT/${TAG}_${nm}_bench.json
  rm -rf $OUT/$nm
done
cd $GRAFT_REPO_ROOT
# the raw traces are large: keep only the summaries
rm -rf $OUT/serial $OUT/overlap $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
cat $OUT/pmc_summary.txt; cat $OUT/mfma_summary.txt; head -12 $OUT/${TAG}_serial_kernel_stats.csv
