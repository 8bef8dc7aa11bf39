mkdir -p gpurun_out/r4j
python -m pytest tests/ -m gpu -x -q > gpurun_out/r4j/tests.log 2>&1; tail -6 gpurun_out/r4j/tests.log
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4j/b.json 2>gpurun_out/r4j/b.err
python - <<'P'
import json
p=json.loads(open('gpurun_out/r4j/b.json').read().strip().splitlines()[-1])
print(p['value'], p['ms_per_step'], p['roofline']['kernel'], p['roofline']['avg_launch_ms'], p.get('roofline_valid'), p['kernel_breakdown_ms'])
P
