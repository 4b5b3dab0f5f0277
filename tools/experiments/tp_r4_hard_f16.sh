#!/bin/bash
# Round 4: the fp16 mode (configs[4]'s arithmetic: fp16 MFMA operands, static loss scale 4096 -- the default when this ran, 65536 since --, 16-bit gathered activation gradients) on the task that can fail, same
# protocol and seeds as tools/experiments/tp_r4_hard.sh -- paired by seed with that sweep's fp32 runs (same batch stream, same initial weights).
# usage: tools/experiments/tp_r4_hard_f16.sh SEED [SEED ...]     (about 2 minutes per seed)
# NOTE: run while the 16-bit gathered activation gradients were the default; MTBC_DA16=1 reproduces that plan.
export MTBC_DA16=1
mkdir -p gpurun_out/r4q_f16
C="--steps ${STEPS:-12000} --batch 16 --size 256 --lr 3e-4 --cosine --eval-every ${EVERY:-3000} --eval-batches 32 --dtypes f16 --hard --loss-scale 4096"
for seed in "$@"; do
  python tools/train_parity.py $C --seed $seed --out gpurun_out/r4q_f16/tp_s${seed}.json > gpurun_out/r4q_f16/tp_s${seed}.log 2>&1
  echo "seed $seed: $(grep "step *${STEPS:-12000} " gpurun_out/r4q_f16/tp_s${seed}.log | tr '\n' ' ')" | tee -a gpurun_out/r4q_f16/summary_$1.txt
done
