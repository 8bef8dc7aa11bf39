#!/bin/bash
# Round profiles of the headline workload (run on the GPU box from the repo root): rocprofv3 kernel stats (the one-stream schedule,
# the ViT engines' default since round 2, and the two-stream one) and the PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of the one-stream schedule, summarised into profiles/.
# Usage: bash tools/profile_round.sh r02      (writes gpurun_out/prof_<tag>/..., then copy the summaries into profiles/)
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
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
# BASELINE config 4 (CaiT-S24, 256 images): kernel stats of the same bench harness
C="$GRAFT_REPO_ROOT/bench.py --model cait_s_24 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cait -o c -- python3 $C > $OUT/cait_bench.json 2> $OUT/cait.err
echo cait done
cd $GRAFT_REPO_ROOT
cp $(find $OUT/cait -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cait_s24_kernel_stats.csv
tail -1 $OUT/cait_bench.json > $OUT/${TAG}_cait_s24_bench.json
rm -rf $OUT/cait
# the raw traces are large: keep only the summaries
rm -rf $OUT/serial $OUT/overlap $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
cat $OUT/pmc_summary.txt; cat $OUT/mfma_summary.txt; head -12 $OUT/${TAG}_serial_kernel_stats.csv
