#!/bin/bash
# Round 4: the fp16 plan on ONE seed of the task that can fail under the plan switches (each a perturbation of where values are rounded / which kernels run):
# does the slow mode of seed 3 survive them?   usage: tools/experiments/tp_r4_hard_f16_arms.sh SEED "ENV1" "ENV2" ...   (9000 steps, ~85 s per arm)
seed=$1; shift
O=gpurun_out/r4q_f16_arms
mkdir -p $O
C="--steps ${STEPS:-9000} --batch 16 --size 256 --lr 3e-4 --cosine --eval-every 3000 --eval-batches 32 --dtypes f16 --hard"
for arm in "$@"; do
  tag=$(echo "${arm:-default}" | tr ' =' '__')
  env $arm python tools/train_parity.py $C --seed $seed --out $O/tp_s${seed}_${tag}.json > $O/tp_s${seed}_${tag}.log 2>&1
  echo "seed $seed [$arm]: $(grep -h 'val dice' $O/tp_s${seed}_${tag}.log | sed 's/.*step *\([0-9]*\) .*val dice \([0-9.]*\) acc \([0-9.]*\).*/\1:\2\/\3/' | tr '\n' ' ')" | tee -a $O/summary.txt
done
