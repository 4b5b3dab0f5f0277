mkdir -p gpurun_out/r4o
python -m pytest tests -x -q -m gpu > gpurun_out/r4o/gpu_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4o/gpu_tests.log; tail -n 4 gpurun_out/r4o/gpu_tests.log
