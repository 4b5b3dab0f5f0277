#!/bin/bash
# Round 3: paired fp32-mode vs bf16-mode quality sweep (north_star: Dice / accuracy within +-0.2 pt), 6000 steps per run so that
# the runs that 3000 steps left between 0.81 and 0.97 get to finish.   usage: tools/experiments/tp_r3.sh SEED [SEED ...]
mkdir -p gpurun_out/r3q
C="--steps 6000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 3000 --eval-batches 32 --dtypes f32,bf16"
for seed in "$@"; do
  python tools/train_parity.py $C --seed $seed --out gpurun_out/r3q/tp_s${seed}.json > gpurun_out/r3q/tp_s${seed}.log 2>&1
  echo "seed $seed: $(grep 'step  6000' gpurun_out/r3q/tp_s${seed}.log | tr '\n' ' ')" | tee -a gpurun_out/r3q/summary_$1.txt
done
