#!/bin/bash
# Same-box attribution of the round-5 engine changes on the headline workload (DeiT-B/16, 128 images): each switch off in turn, the
# default build before, between and after.  usage (GPU box, repo root): bash tools/ab_round5.sh > gpurun_out/ab_round5.log
run() {
  env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %8.1f img/s  %7.3f ms/step  wgrad launch %.4f ms  launches %s' % ('$*' or 'default', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j.get('launches_per_train_step')))"
}
run SAVIT_DUMMY=0
run SAVIT_ROWS_TILE=0
run SAVIT_CLS_FWD=0
run SAVIT_CLS_ONLY_LAST=0
run SAVIT_DUMMY=0
run SAVIT_WPE_GROUPED=0
run SAVIT_WGRAD_SMALL_GROUPS=1
run SAVIT_WGRAD_FIRST_TOUCH=0
run SAVIT_DEFER_LN_FINALIZE=0
run SAVIT_DUMMY=0
