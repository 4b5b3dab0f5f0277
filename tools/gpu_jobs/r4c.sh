mkdir -p gpurun_out/r4c
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad" > gpurun_out/r4c/wgrad_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4c/wgrad_tests.log; tail -n 4 gpurun_out/r4c/wgrad_tests.log
python tools/wgrad_probe.py 1 32 > gpurun_out/r4c/wgrad_probe_fix.txt 2>&1; tail -n 2 gpurun_out/r4c/wgrad_probe_fix.txt
PROBE_REDUCE=launch python tools/wgrad_probe.py 1 32 0 > gpurun_out/r4c/wgrad_probe_launch.txt 2>&1; tail -n 2 gpurun_out/r4c/wgrad_probe_launch.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline > gpurun_out/r4c/bench1.json 2> gpurun_out/r4c/bench1.err; tail -c 300 gpurun_out/r4c/bench1.json
python tests/studies/try_emul3.py 256 2 bf16 40 1e-3 > gpurun_out/r4c/emul3_bf16_k40.log 2>&1; tail -n 1 gpurun_out/r4c/emul3_bf16_k40.log
python tests/studies/try_emul3.py 512 1 f16 40 1e-3 > gpurun_out/r4c/emul3_f16_512_k40.log 2>&1; tail -n 1 gpurun_out/r4c/emul3_f16_512_k40.log
python -m pytest tests/test_model_gpu.py -q -m gpu -k "16bit_mfma" > gpurun_out/r4c/emul_tests.log 2>&1; tail -n 5 gpurun_out/r4c/emul_tests.log
