#!/bin/bash
# round 4: the step as a hipGraph replay (FusedTrainStep(graph=True) / MTBC_GRAPH=1, mtbc_adam_args.dynamic): its test, the host-issue / step-time comparison,
# an interleaved bench A/B, then the whole GPU suite and smoke at HEAD
cd /root/repo
OUT=gpurun_out/r4w
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -q -m gpu -k "graph_replayed or deterministic" > $OUT/graph_test.log 2>&1; echo "rc $?" >> $OUT/graph_test.log; tail -n 4 $OUT/graph_test.log
grep -q "rc 0" $OUT/graph_test.log || exit 1
timeout -k 10 300 python tools/experiments/graph_replay.py bf16 32 256 > $OUT/graph_replay.txt 2>&1; tail -n 8 $OUT/graph_replay.txt
ROUNDS=2 bash tools/ab_env.sh "" "MTBC_GRAPH=1" > $OUT/ab_graph.log 2>&1; cat $OUT/ab_graph.log
python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "rc $?" >> $OUT/gpu_tests.log; tail -n 3 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -n 1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['fp32_parity_mode']['value'])"
