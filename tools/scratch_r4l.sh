mkdir -p gpurun_out/r4l
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "wgrad_grouped or grouped_and" > gpurun_out/r4l/tests.log 2>&1; tail -4 gpurun_out/r4l/tests.log
for t in 384 640; do SAVIT_GROUP_TILE=$t python tools/bench_wgrad_group.py 384 1536 50432 12 > gpurun_out/r4l/wg_$t.log 2>&1; grep round gpurun_out/r4l/wg_$t.log | sed 's/per-weight launches.*| grouped/grouped/'; done
for t in 384 640 384 640; do SAVIT_WGRAD_TILE=$t python bench.py --model vit_s_patch16 --batch 256 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4l/s_$t.json 2>>gpurun_out/r4l/b.err
python - $t <<'P'
import json,sys
p=json.loads(open(f'gpurun_out/r4l/s_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('deit-s tile', sys.argv[1], p['value'], p['ms_per_step'], p['roofline']['kernel'], p['roofline']['avg_launch_ms'], p['roofline']['launches_per_step'], p.get('roofline_valid'))
P
done
for t in 384 640; do SAVIT_WGRAD_TILE=$t python bench.py --model cait_s_24 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r4l/c_$t.json 2>>gpurun_out/r4l/b.err
python - $t <<'P'
import json,sys
p=json.loads(open(f'gpurun_out/r4l/c_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('cait tile', sys.argv[1], p['value'], p['ms_per_step'], p['roofline']['kernel'], p['roofline']['avg_launch_ms'], p['roofline']['launches_per_step'], p.get('roofline_valid'))
P
done
tail -3 gpurun_out/r4l/b.err
