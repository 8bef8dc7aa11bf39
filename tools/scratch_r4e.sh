mkdir -p gpurun_out/r4e
python -m pytest tests/test_timing_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "wgrad or timing or grouped or gemm or attention" > gpurun_out/r4e/tests.log 2>&1; tail -5 gpurun_out/r4e/tests.log
python tools/bench_wgrad_group.py 384 1536 50432 6 > gpurun_out/r4e/wg_s_t384.log 2>&1
SAVIT_GROUP_TILE=256 python tools/bench_wgrad_group.py 384 1536 50432 6 > gpurun_out/r4e/wg_s_t256.log 2>&1
grep round gpurun_out/r4e/wg_*.log
python bench.py --model vit_s_patch16 --batch 256 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4e/s_base.json 2>>gpurun_out/r4e/b.err
python tools/cu_thief_probe.py > gpurun_out/r4e/cu_thief.log 2>&1; cat gpurun_out/r4e/cu_thief.log | grep -v "^{"
python - <<'P'
import json
p=json.loads(open('gpurun_out/r4e/s_base.json').read().strip().splitlines()[-1])
print('deit-s', p['value'], p['ms_per_step'], p['roofline']['kernel'], p['roofline']['avg_launch_ms'], p['roofline']['launches_per_step'], p.get('roofline_valid'))
P
