mkdir -p gpurun_out/r4r
python tools/attn_bench.py 256 577 16 64 2>&1 | grep -v amdgpu
python tools/attn_bench.py 64 577 16 48 2>&1 | grep -v amdgpu
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "attention or vit_l or ViT_L or 384" > gpurun_out/r4r/tests.log 2>&1; tail -4 gpurun_out/r4r/tests.log
