mkdir -p gpurun_out/r4g
python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_kernels_gpu.py tests/test_train_cli.py -m gpu -x -q -k "not attention and not talking and not th_" > gpurun_out/r4g/tests.log 2>&1; tail -5 gpurun_out/r4g/tests.log
python tools/profile_step.py > gpurun_out/r4g/prof.log 2>&1; grep -E "adamw|cast|sumsq|zero|^sum" gpurun_out/r4g/prof.log
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4g/b.json 2>gpurun_out/r4g/b.err
python - <<'P'
import json
p=json.loads(open('gpurun_out/r4g/b.json').read().strip().splitlines()[-1])
print(p['value'], p['ms_per_step'], p['roofline']['kernel'], p['roofline']['avg_launch_ms'], p.get('roofline_valid'), p['kernel_breakdown_ms'])
P
