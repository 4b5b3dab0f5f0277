mkdir -p gpurun_out/r4n
bash tools/experiments/tp_r4_hard_f16.sh 6 7 8 9 10
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r4n/smoke.log 2>&1; tail -n 1 gpurun_out/r4n/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4n/bench_driver_like.json 2> gpurun_out/r4n/bench_driver_like.err; python -c "
import json; d=json.loads(open('gpurun_out/r4n/bench_driver_like.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_launch_roofs']['frac'], d['roofline']['wgrad']['frac_of_hbm_peak'], d['fp32_parity_mode']['value'], d['fp32_parity_mode']['roofline']['wgrad']['frac_of_mfma_peak'], d['fp32_parity_mode']['roofline']['all_3x3_conv']['frac_of_mfma_peak'], d['cpu_baseline']['value'])"
