#!/bin/bash
# Round 4: the SHIPPED fp16 plan (fp16 MFMA operands, static loss scale 4096 -- the default when this ran, 65536 since --, fp32 gathered activation gradients) on the task that can fail, same protocol and seeds
# as tp_r4_hard.sh; paired by seed with that sweep's fp32 runs.   usage: tools/experiments/tp_r4_hard_f16_shipped.sh SEED [SEED ...]     (about 2 minutes per seed)
mkdir -p gpurun_out/r4q_f16_shipped
C="--steps ${STEPS:-12000} --batch 16 --size 256 --lr 3e-4 --cosine --eval-every ${EVERY:-3000} --eval-batches 32 --dtypes f16 --hard --loss-scale 4096"
for seed in "$@"; do
  python tools/train_parity.py $C --seed $seed --out gpurun_out/r4q_f16_shipped/tp_s${seed}.json > gpurun_out/r4q_f16_shipped/tp_s${seed}.log 2>&1
  echo "seed $seed: $(grep "step *${STEPS:-12000} " gpurun_out/r4q_f16_shipped/tp_s${seed}.log | tr '\n' ' ')" | tee -a gpurun_out/r4q_f16_shipped/summary_$1.txt
done
