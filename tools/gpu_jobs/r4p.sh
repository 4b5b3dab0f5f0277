mkdir -p gpurun_out/r4p
python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r4p/gpu_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4p/gpu_tests.log; tail -n 25 gpurun_out/r4p/gpu_tests.log
