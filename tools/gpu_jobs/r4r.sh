mkdir -p gpurun_out/r4r
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4r/bench.json 2> gpurun_out/r4r/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r4r/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_launch_roofs']['frac'], d['roofline']['wgrad']['frac_of_hbm_peak'], d['fp32_parity_mode']['value'], d['fp32_parity_mode']['ms_per_step'], d['fp32_parity_mode']['roofline']['frac'], d['fp32_parity_mode']['roofline']['wgrad']['frac_of_mfma_peak'], d['fp32_parity_mode']['roofline']['all_3x3_conv']['frac_of_mfma_peak'])"
bash tools/experiments/tp_r4_hard_noda16.sh 7 8 9 10
python -m pytest tests -q -m gpu > gpurun_out/r4r/gpu_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4r/gpu_tests.log; tail -n 3 gpurun_out/r4r/gpu_tests.log
