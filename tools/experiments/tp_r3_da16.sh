#!/bin/bash
# HISTORICAL (rounds 2 - 3): MTBC_NO_DA16 / the defaults of that time; since round 4 the 16-bit gathered gradients are the opt-in MTBC_DA16=1 and MTBC_NO_DA16 is refused.
# Round 3: the 16-bit gathered activation gradients under the 6000-step protocol of tp_r3.sh, bf16 mode only, paired by seed.  When this ran, the 16-bit
# gradients were the opt-in arm (MTBC_DA16=1, paired with the then-default fp32-gradient runs of gpurun_out/r3q).  The default was flipped on its result:
# the switch is now MTBC_NO_DA16 (MTBC_DA16 is refused, switches.REMOVED), so the pairing today is MTBC_NO_DA16=1 (fp32 gradients) vs the default plan.
# usage: tools/experiments/tp_r3_da16.sh SEED [SEED ...]      -> gpurun_out/r3q_da16/ holds the NON-default arm
mkdir -p gpurun_out/r3q_da16
C="--steps 6000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 3000 --eval-batches 32 --dtypes bf16"
for seed in "$@"; do
  MTBC_NO_DA16=1 python tools/train_parity.py $C --seed $seed --out gpurun_out/r3q_da16/tp_s${seed}.json > gpurun_out/r3q_da16/tp_s${seed}.log 2>&1
  echo "seed $seed NO_DA16: $(grep 'step  6000' gpurun_out/r3q_da16/tp_s${seed}.log | tr '\n' ' ')" | tee -a gpurun_out/r3q_da16/summary_$1.txt
done
