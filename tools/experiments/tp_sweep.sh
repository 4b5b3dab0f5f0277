# held-out Dice after 3000 steps, bf16, 6 seeds x {default, MTBC_NO_DA16, MTBC_NO_Z16}: is the 16-bit storage of conv outputs / gathered gradients visible in quality?
# HISTORICAL (rounds 2 - 3): MTBC_NO_DA16 / the defaults of that time; since round 4 the 16-bit gathered gradients are the opt-in MTBC_DA16=1 and MTBC_NO_DA16 is refused.
mkdir -p gpurun_out/r02z
C="--steps 3000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 3000 --eval-batches 32 --dtypes bf16"
for seed in 1 2 3 4 5 6; do
  for arm in default MTBC_NO_DA16 MTBC_NO_Z16; do
    if [ $arm = default ]; then E=""; else E="$arm=1"; fi
    env $E python tools/train_parity.py $C --seed $seed --out gpurun_out/r02z/tp_${arm}_s${seed}.json > gpurun_out/r02z/tp_${arm}_s${seed}.log 2>&1
    echo "seed $seed $arm $(grep 'step  3000' gpurun_out/r02z/tp_${arm}_s${seed}.log)" | tee -a gpurun_out/r02z/summary.txt
  done
done
