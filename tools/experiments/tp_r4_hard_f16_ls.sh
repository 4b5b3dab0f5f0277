#!/bin/bash
# Round 4: the fp16 plan (fp32 gathered activation gradients) with another static loss scale (LS, default 65536) on the task that can fail; same protocol and
# seeds as tp_r4_hard_f16_shipped.sh (which ran at 4096).   usage: LS=65536 tools/experiments/tp_r4_hard_f16_ls.sh SEED [SEED ...]     (about 2 minutes per seed)
LS=${LS:-65536}
O=${OUT:-gpurun_out/r4q_f16_ls${LS}}      # OUT=...: e.g. a longer schedule (STEPS=18000) at the shipped scale, kept apart from the 12000-step runs
mkdir -p $O
C="--steps ${STEPS:-12000} --batch 16 --size 256 --lr 3e-4 --cosine --eval-every ${EVERY:-3000} --eval-batches 32 --dtypes f16 --hard --loss-scale $LS"
for seed in "$@"; do
  python tools/train_parity.py $C --seed $seed --out $O/tp_s${seed}.json > $O/tp_s${seed}.log 2>&1
  echo "seed $seed: $(grep "step *${STEPS:-12000} " $O/tp_s${seed}.log | tr '\n' ' ')" | tee -a $O/summary_$1.txt
done
