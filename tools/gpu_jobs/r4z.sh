#!/bin/bash
# round 4: fp16 static loss scale 2^12 -> 2^16: the fp16 whole-model tests, the configs[4]-shaped bench line (NaN guard), smoke
cd /root/repo
mkdir -p gpurun_out/r4z
timeout -k 10 330 python -m pytest tests/test_model_gpu.py tests/test_next_rows_gpu.py -q -m gpu -k "f16 or 512 or graph_replayed or validation" > gpurun_out/r4z/f16_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4z/f16_tests.log; tail -n 3 gpurun_out/r4z/f16_tests.log
python bench.py --size 512 --dtype f16 --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode > gpurun_out/r4z/bench_config4.json 2> gpurun_out/r4z/bench_config4.err; python -c "
import json; d=json.loads(open('gpurun_out/r4z/bench_config4.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
