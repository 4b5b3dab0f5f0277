#!/bin/bash
# Round 3: the MTBC_DA16 arm (gathered activation gradients stored as bf16) under the 6000-step protocol of tp_r3.sh, bf16 mode only; paired by seed with
# the default-plan bf16 runs of gpurun_out/r3q.   usage: tools/experiments/tp_r3_da16.sh SEED [SEED ...]
mkdir -p gpurun_out/r3q_da16
C="--steps 6000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 3000 --eval-batches 32 --dtypes bf16"
for seed in "$@"; do
  MTBC_DA16=1 python tools/train_parity.py $C --seed $seed --out gpurun_out/r3q_da16/tp_s${seed}.json > gpurun_out/r3q_da16/tp_s${seed}.log 2>&1
  echo "seed $seed DA16: $(grep 'step  6000' gpurun_out/r3q_da16/tp_s${seed}.log | tr '\n' ' ')" | tee -a gpurun_out/r3q_da16/summary_$1.txt
done
