mkdir -p gpurun_out/r4s
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4s/bench.json 2> gpurun_out/r4s/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r4s/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_launch_roofs']['frac'], d['roofline']['wgrad']['frac_of_hbm_peak'], d['fp32_parity_mode']['value'], d['fp32_parity_mode']['roofline']['wgrad']['frac_of_mfma_peak'], d['fp32_parity_mode']['roofline']['all_3x3_conv']['frac_of_mfma_peak'])"
ROUNDS=3 bash tools/ab_env.sh "" "MTBC_DA16=1" > gpurun_out/r4s/ab_da16.log 2>&1; cat gpurun_out/r4s/ab_da16.log
python -m pytest tests -q -m gpu > gpurun_out/r4s/gpu_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4s/gpu_tests.log; tail -n 3 gpurun_out/r4s/gpu_tests.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -n 1
