#!/bin/bash
# round 4, closing run at HEAD: bf16 kernel stats + PMC traffic re-collected (tools/profile_round.sh's first three passes), the driver's bench command traced and
# untraced, the whole GPU test suite, smoke
cd /root/repo
set -e
OUT=gpurun_out/r4u
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-roofline --no-parity-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16_stats -- python3 bench.py --dtype bf16 --steps 5 --warmup 2 $COMMON > $OUT/bf16_stats.log 2>&1
cp "$(ls $OUT/bf16_stats/*/*kernel_stats.csv | head -1)" $OUT/r04_bf16_kernel_stats.csv
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/bf16_fetch -- python3 bench.py --dtype bf16 --steps 2 --warmup 1 $COMMON > $OUT/bf16_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/bf16_write -- python3 bench.py --dtype bf16 --steps 2 --warmup 1 $COMMON > $OUT/bf16_write.log 2>&1
echo "write done"
python3 tools/pmc_traffic.py $OUT/bf16_fetch $OUT/bf16_write $OUT/r04_hbm_traffic_bf16.json
cp $OUT/r04_hbm_traffic_bf16.json profiles/r04_hbm_traffic_bf16.json        # bench.py reads roofline.traffic from here
rm -rf $OUT/bf16_fetch $OUT/bf16_write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver_stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_driver_command.json 2> $OUT/driver_stats.err
cp "$(ls $OUT/driver_stats/*/*kernel_stats.csv | head -1)" $OUT/r04_driver_command_kernel_stats.csv
rm -rf $OUT/driver_stats $OUT/bf16_stats
echo "driver command traced"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench.json 2> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/r04_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['per_launch_roofs']['frac'], r['traffic_over_algorithmic'], r['wgrad']['frac_of_hbm_peak'], d['fp32_parity_mode']['value'], d['fp32_parity_mode']['roofline']['wgrad']['frac_of_mfma_peak'], d['fp32_parity_mode']['roofline']['all_3x3_conv']['frac_of_mfma_peak'])"
set +e
python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "rc $?" >> $OUT/gpu_tests.log; tail -n 3 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -n 1
