mkdir -p gpurun_out/r4h
for sh in "24 24 256" "72 24 256" "144 24 256" "48 48 128" "192 48 128" "96 96 64"; do python tools/experiments/f32_wgrad_one.py $sh; done > gpurun_out/r4h/f32_wgrad_fixed.log 2>&1; grep TF gpurun_out/r4h/f32_wgrad_fixed.log
export MTBC_LIB=$PWD/multi_task_breast_cancer_amd/libmtbc_hip_probes.so
for sh in "24 24 256" "72 24 256" "144 24 256" "48 48 128" "192 48 128"; do MTBC_WGRAD_PACK24=1 python tools/experiments/f32_wgrad_one.py $sh; done > gpurun_out/r4h/f32_wgrad_fixed_pack24.log 2>&1; grep TF gpurun_out/r4h/f32_wgrad_fixed_pack24.log
MTBC_WG_TS=1 python tools/experiments/f32_wgrad_one.py 24 24 256 2>&1 | grep wg_ts | tail -n 1 > gpurun_out/r4h/f32_wg_ts_fixed.log
MTBC_WG_TS=1 MTBC_WGRAD_PACK24=1 python tools/experiments/f32_wgrad_one.py 24 24 256 2>&1 | grep wg_ts | tail -n 1 >> gpurun_out/r4h/f32_wg_ts_fixed.log
MTBC_WG_TS=1 python tools/experiments/f32_wgrad_one.py 192 48 128 2>&1 | grep wg_ts | tail -n 1 >> gpurun_out/r4h/f32_wg_ts_fixed.log
cut -c1-330 gpurun_out/r4h/f32_wg_ts_fixed.log
BENCH_ARGS=--allow-probes ROUNDS=3 bash tools/ab_env.sh "MTBC_C8_NW3=4" "MTBC_C8_NW3=8" > gpurun_out/r4h/ab_nw3.log 2>&1; cat gpurun_out/r4h/ab_nw3.log
unset MTBC_LIB
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3" > gpurun_out/r4h/conv_tests.log 2>&1; tail -n 2 gpurun_out/r4h/conv_tests.log
python bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > gpurun_out/r4h/bench_f32.json 2> gpurun_out/r4h/bench_f32.err; python -c "
import json; d=json.loads(open('gpurun_out/r4h/bench_f32.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['wgrad']['frac_of_mfma_peak'], d['roofline']['all_3x3_conv'])"
bash tools/experiments/tp_r4_hard.sh 1
