#!/bin/bash
# A/B of two builds of the library on ONE box: per-op tables of the same step (tools/per_op.py) with MTBC_LIB pointing at each.
#   tools/ab_ops.sh OUTDIR [arch batch size dtype]      (B = multi_task_breast_cancer_amd/libmtbc_hip_prev.so)
OUT=${1:-gpurun_out/ab}; shift
ARGS=${@:-MTUNetPlusPlus 32 256 bf16}
mkdir -p $OUT
for round in 1 2; do
  PEROP_MIN=0.02 python tools/per_op.py $ARGS > $OUT/new_$round.log 2>&1
  MTBC_LIB=$PWD/multi_task_breast_cancer_amd/libmtbc_hip_prev.so PEROP_MIN=0.02 python tools/per_op.py $ARGS > $OUT/prev_$round.log 2>&1
done
for f in new_1 prev_1 new_2 prev_2; do echo "== $f"; head -14 $OUT/$f.log | tail -13; done
