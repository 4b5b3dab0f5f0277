# third sweep: the default plan after the change (conv outputs stored as fp16, fp32 activation gradients) on the 15 seeds of sweeps 1 + 2
mkdir -p gpurun_out/r03b
C="--steps 3000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 3000 --eval-batches 32 --dtypes bf16"
for seed in 1 2 3 4 5 6 21 22 23 24 25 26 1993 7 11; do
  python tools/train_parity.py $C --seed $seed --out gpurun_out/r03b/tp_default_s${seed}.json > gpurun_out/r03b/tp_default_s${seed}.log 2>&1
  echo "seed $seed default(fp16 z) $(grep 'step  3000' gpurun_out/r03b/tp_default_s${seed}.log)" | tee -a gpurun_out/r03b/summary.txt
done
