#!/bin/bash
# Round 4: the bf16 mode with fp32 (not 16-bit) gathered activation gradients (MTBC_NO_DA16=1 when this ran; the shipped default since -- the variable is refused now) on the task that can fail, same protocol and seeds as tp_r4_hard.sh --
# does the 16-bit gradient storage delay the climb the default bf16 plan shows mid-run?   usage: tools/experiments/tp_r4_hard_noda16.sh SEED [SEED ...]   (2 minutes per seed)
mkdir -p gpurun_out/r4q_noda16
C="--steps ${STEPS:-12000} --batch 16 --size 256 --lr 3e-4 --cosine --eval-every ${EVERY:-3000} --eval-batches 32 --dtypes bf16 --hard"
for seed in "$@"; do
  python tools/train_parity.py $C --seed $seed --out gpurun_out/r4q_noda16/tp_s${seed}.json > gpurun_out/r4q_noda16/tp_s${seed}.log 2>&1
  echo "seed $seed: $(grep "step *${STEPS:-12000} " gpurun_out/r4q_noda16/tp_s${seed}.log | tr '\n' ' ')" | tee -a gpurun_out/r4q_noda16/summary_$1.txt
done
