mkdir -p gpurun_out/r4k
python -m pytest tests/ -m gpu -x -q --deselect tests/test_timing_gpu.py::test_timer_measures_a_gate_kernel -k "timing or ddp or fullsize or fp32 or hf or input or mixer or tnt or train or golden" > gpurun_out/r4k/tests.log 2>&1; tail -4 gpurun_out/r4k/tests.log
python -m pytest tests/test_timing_gpu.py -m gpu -x -q > gpurun_out/r4k/tests2.log 2>&1; tail -2 gpurun_out/r4k/tests2.log
for i in 1 2; do python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4k/b$i.json 2>gpurun_out/r4k/b.err; done
python - <<'P'
import json
for i in (1,2):
    p=json.loads(open(f'gpurun_out/r4k/b{i}.json').read().strip().splitlines()[-1])
    print(p['value'], p['ms_per_step'], p['roofline']['avg_launch_ms'], p.get('roofline_valid'), p['kernel_breakdown_ms'])
P
