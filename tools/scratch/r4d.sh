set -x
mkdir -p gpurun_out/r4d
python -m pytest tests/test_timing_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "wgrad or timing or grouped or gemm" > gpurun_out/r4d/tests.log 2>&1; tail -5 gpurun_out/r4d/tests.log
python tools/dump_gemm_shapes.py > gpurun_out/r4d/gemm_shapes.json 2> gpurun_out/r4d/shapes.err; tail -3 gpurun_out/r4d/shapes.err
for r in 1 2; do
python tools/bench_wgrad_group.py 768 3072 25216 2 > gpurun_out/r4d/wg_b_mf16_$r.log 2>&1
SAVIT_EXP_LIB=mf32 python tools/bench_wgrad_group.py 768 3072 25216 2 > gpurun_out/r4d/wg_b_mf32_$r.log 2>&1
done
python tools/bench_wgrad_group.py 384 1536 50432 6 > gpurun_out/r4d/wg_s_t384.log 2>&1
SAVIT_GROUP_TILE=256 python tools/bench_wgrad_group.py 384 1536 50432 6 > gpurun_out/r4d/wg_s_t256.log 2>&1
grep round gpurun_out/r4d/wg_*.log
for r in 1 2; do
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4d/b_mf16_$r.json 2>gpurun_out/r4d/b.err
SAVIT_EXP_LIB=mf32 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4d/b_mf32_$r.json 2>>gpurun_out/r4d/b.err
done
python bench.py --model vit_s_patch16 --batch 256 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4d/s_base.json 2>>gpurun_out/r4d/b.err
SAVIT_EXP_LIB=ppk SAVIT_PP_MIN_K=384 python bench.py --model vit_s_patch16 --batch 256 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4d/s_ppk.json 2>>gpurun_out/r4d/b.err
python bench.py --model cait_s_24 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r4d/c_base.json 2>>gpurun_out/r4d/b.err
python bench.py --model vit_l_patch16 --img-size 384 --batch 256 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r4d/l_base.json 2>>gpurun_out/r4d/b.err
for f in gpurun_out/r4d/*.json; do python - "$f" <<'P'
import json,sys
try:
    p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], p['value'], p['ms_per_step'], p['roofline']['kernel'], p['roofline']['avg_launch_ms'], p.get('roofline_valid'))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
tail -5 gpurun_out/r4d/b.err
