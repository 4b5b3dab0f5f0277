#!/bin/bash
# Round 4: paired fp32-mode vs bf16-mode quality sweep on the task that can fail (synthetic.synthetic_batch(hard=True)): STEPS (default 12000) steps, batch 16, cosine schedule,
# hard Dice / accuracy on 32 held-out batches.   usage: tools/experiments/tp_r4_hard.sh SEED [SEED ...]      (about 8 minutes per seed at 12000 steps: 2 seeds per gpurun call; the 6000-step runs of seeds 1 - 4 had not converged)
# NOTE: these runs were made while the 16-bit gathered activation gradients were the bf16 plan's default; they are the opt-in MTBC_DA16=1 since (set here so that the
# command still reproduces the recorded runs; fp32 mode ignores it).  The shipped bf16 plan is what tp_r4_hard_noda16.sh ran.
export MTBC_DA16=1
mkdir -p gpurun_out/r4q
C="--steps ${STEPS:-12000} --batch 16 --size 256 --lr 3e-4 --cosine --eval-every ${EVERY:-3000} --eval-batches 32 --dtypes f32,bf16 --hard"
for seed in "$@"; do
  python tools/train_parity.py $C --seed $seed --out gpurun_out/r4q/tp_s${seed}.json > gpurun_out/r4q/tp_s${seed}.log 2>&1
  echo "seed $seed: $(grep "step *${STEPS:-12000} " gpurun_out/r4q/tp_s${seed}.log | tr '\n' ' ')" | tee -a gpurun_out/r4q/summary_$1.txt
done
