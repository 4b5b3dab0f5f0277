mkdir -p gpurun_out/r4k
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad and not c8" > gpurun_out/r4k/f32_wgrad_tests.log 2>&1; tail -n 2 gpurun_out/r4k/f32_wgrad_tests.log
for sh in "24 24 256" "72 24 256" "96 24 256" "120 24 256" "144 24 256" "48 48 128"; do python tools/experiments/f32_wgrad_one.py $sh; done > gpurun_out/r4k/f32_wgrad_swap.log 2>&1; grep TF gpurun_out/r4k/f32_wgrad_swap.log
python bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > gpurun_out/r4k/bench_f32.json 2> gpurun_out/r4k/bench_f32.err; python -c "
import json; d=json.loads(open('gpurun_out/r4k/bench_f32.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['wgrad']['frac_of_mfma_peak'], d['roofline']['all_3x3_conv'])"
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "fused_step_matches_oracle or golden" > gpurun_out/r4k/model_f32_tests.log 2>&1; tail -n 2 gpurun_out/r4k/model_f32_tests.log
bash tools/experiments/tp_r4_hard.sh 6 7
