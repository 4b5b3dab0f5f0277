mkdir -p gpurun_out/r4q_misc
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3" > gpurun_out/r4q_misc/conv_tests.log 2>&1; tail -n 2 gpurun_out/r4q_misc/conv_tests.log
BENCH_ARGS=--allow-probes ROUNDS=4 bash tools/ab_env.sh "" "MTBC_LIB=$PWD/tools/experiments/bin/libmtbc_prev.so" > gpurun_out/r4q_misc/ab_prologue.log 2>&1; cat gpurun_out/r4q_misc/ab_prologue.log
bash tools/experiments/tp_r4_hard_noda16.sh 1 2 3 4 5 6
