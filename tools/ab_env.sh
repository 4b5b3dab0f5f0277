#!/bin/bash
# A/B of plan switches / probe switches on ONE box: interleaved short bench runs, ms per step.
#   tools/ab_env.sh "NAME1=VAL1 NAME2=VAL2" "NAME3=VAL3" ...     (each argument = the environment of one arm; "" = default)
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for arm in "$@"; do
    ms=$(env $arm python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r [${arm:-default}] $ms ms"
  done
done
