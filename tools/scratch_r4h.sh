mkdir -p gpurun_out/r4h
python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "adamw or train_steps or grouped or golden or known or parity" > gpurun_out/r4h/tests.log 2>&1; tail -4 gpurun_out/r4h/tests.log
python tools/profile_step.py vit_l_patch16 256 384 > gpurun_out/r4h/prof_l.log 2>&1; grep -v amdgpu gpurun_out/r4h/prof_l.log
python tools/profile_step.py > gpurun_out/r4h/prof_b.log 2>&1; grep -E "adamw|cast|sumsq|^sum" gpurun_out/r4h/prof_b.log
