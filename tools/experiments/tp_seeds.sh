mkdir -p gpurun_out/r02y
C="--steps 3000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 1000 --eval-batches 32 --dtypes bf16"
MTBC_NO_Z16=1 python tools/train_parity.py $C --out gpurun_out/r02y/tp_noz16_s1993.json > gpurun_out/r02y/tp_noz16_s1993.log 2>&1; tail -2 gpurun_out/r02y/tp_noz16_s1993.log | head -1
python tools/train_parity.py $C --seed 7 --out gpurun_out/r02y/tp_z16_s7.json > gpurun_out/r02y/tp_z16_s7.log 2>&1; tail -2 gpurun_out/r02y/tp_z16_s7.log | head -1
MTBC_NO_Z16=1 python tools/train_parity.py $C --seed 7 --out gpurun_out/r02y/tp_noz16_s7.json > gpurun_out/r02y/tp_noz16_s7.log 2>&1; tail -2 gpurun_out/r02y/tp_noz16_s7.log | head -1
python tools/train_parity.py $C --seed 11 --out gpurun_out/r02y/tp_z16_s11.json > gpurun_out/r02y/tp_z16_s11.log 2>&1; tail -2 gpurun_out/r02y/tp_z16_s11.log | head -1
MTBC_NO_Z16=1 python tools/train_parity.py $C --seed 11 --out gpurun_out/r02y/tp_noz16_s11.json > gpurun_out/r02y/tp_noz16_s11.log 2>&1; tail -2 gpurun_out/r02y/tp_noz16_s11.log | head -1
