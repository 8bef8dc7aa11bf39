mkdir -p gpurun_out/r4o
SAVIT_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4o/b2.json 2> gpurun_out/r4o/b2.err; echo rc=$?; tail -c 600 gpurun_out/r4o/b2.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r4o/b2.json').read().strip().splitlines() if x.startswith('{')]
p=json.loads(l[-1])
print(p['value'], p['n_gpus'], p['ms_per_step'], p['roofline']['kernel'], p['roofline']['avg_launch_ms'], p.get('roofline_valid'), p.get('allreduce_exposed_ms'))
print({k:v for k,v in p['config']['distributed'].items() if k!='ranks'})
P
