mkdir -p gpurun_out/r4n
bash tools/profile_round.sh r04 > gpurun_out/prof_r04.log 2>&1; tail -32 gpurun_out/prof_r04.log | head -30
python tools/cu_thief_probe.py > gpurun_out/r4n/cu_thief.log 2>&1; grep -v "^{" gpurun_out/r4n/cu_thief.log | grep -v amdgpu
SAVIT_RESERVED_CUS=16 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r4n/b_res16.json 2> gpurun_out/r4n/b_res16.err; tail -c 300 gpurun_out/r4n/b_res16.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4n/bench_driver_cmd.json 2> gpurun_out/r4n/bench_driver_cmd.err
python - <<'P'
import json
for f in ('gpurun_out/r4n/b_res16.json','gpurun_out/r4n/bench_driver_cmd.json'):
    p=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, p['value'], p['ms_per_step'], p['roofline']['kernel'], p['roofline']['avg_launch_ms'], p['roofline']['frac'], p.get('roofline_valid'), p.get('roofline_checks'))
    for k,v in p.get('other_configs',{}).items(): print('   ',k[:40], v['value'], v['ms_per_step'], v.get('roofline'), v.get('step_roofline',{}).get('frac'))
    if 'cpu_baseline' in p: print('   cpu', p['cpu_baseline']['value'], p['cpu_baseline']['cores'])
P
