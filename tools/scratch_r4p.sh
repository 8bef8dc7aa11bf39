mkdir -p gpurun_out/r4p
python -m pytest tests/ -m gpu -x -q > gpurun_out/r4p/tests.log 2>&1; tail -5 gpurun_out/r4p/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4p/smoke.log 2>&1; tail -2 gpurun_out/r4p/smoke.log
