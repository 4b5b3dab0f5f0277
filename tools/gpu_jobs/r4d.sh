mkdir -p gpurun_out/r4d
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad" > gpurun_out/r4d/wgrad_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4d/wgrad_tests.log; tail -n 3 gpurun_out/r4d/wgrad_tests.log
python tools/wgrad_probe.py 1 32 0 > gpurun_out/r4d/wgrad_probe_fix.txt 2>&1; tail -n 1 gpurun_out/r4d/wgrad_probe_fix.txt
PROBE_REDUCE=launch python tools/wgrad_probe.py 1 32 0 > gpurun_out/r4d/wgrad_probe_launch.txt 2>&1; tail -n 1 gpurun_out/r4d/wgrad_probe_launch.txt
python -m pytest tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r4d/model_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4d/model_tests.log; tail -n 6 gpurun_out/r4d/model_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline > gpurun_out/r4d/bench1.json 2> gpurun_out/r4d/bench1.err; tail -c 200 gpurun_out/r4d/bench1.json; tail -n 3 gpurun_out/r4d/bench1.err
PEROP_MIN=0.02 python tools/per_op.py MTUNetPlusPlus 32 256 bf16 > gpurun_out/r4d/per_op_bf16.log 2>&1; head -n 8 gpurun_out/r4d/per_op_bf16.log
for c in "0.08,0.14" "0.15,0.15" "0.25,0.15"; do
  python tools/train_parity.py --steps 3000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 1000 --eval-batches 16 --dtypes bf16 --seed 1 --hard --hard-contrast $c > gpurun_out/r4d/hard_cal_$c.log 2>&1; tail -n 4 gpurun_out/r4d/hard_cal_$c.log
done
