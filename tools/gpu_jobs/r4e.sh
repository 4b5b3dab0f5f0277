mkdir -p gpurun_out/r4e
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline > gpurun_out/r4e/bench_default.json 2> gpurun_out/r4e/bench_default.err; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4e/bench_default.json
BENCH_ARGS=--allow-probes ROUNDS=3 bash tools/ab_env.sh "" "MTBC_LIB=$PWD/tools/experiments/bin/libmtbc_xcdteams.so" > gpurun_out/r4e/ab_xcd_teams.log 2>&1; cat gpurun_out/r4e/ab_xcd_teams.log
PEROP_MIN=0.02 python tools/per_op.py MTUNetPlusPlus 32 256 bf16 > gpurun_out/r4e/per_op_bf16.log 2>&1; head -n 12 gpurun_out/r4e/per_op_bf16.log
MTBC_LIB=$PWD/tools/experiments/bin/libmtbc_xcdteams.so PEROP_MIN=0.02 python tools/per_op.py MTUNetPlusPlus 32 256 bf16 > gpurun_out/r4e/per_op_bf16_xcdteams.log 2>&1; grep "OP_IN_BWD  " gpurun_out/r4e/per_op_bf16_xcdteams.log | head -n 2
python -m pytest tests -x -q -m gpu -k "instnorm or coop" > gpurun_out/r4e/in_tests.log 2>&1; tail -n 2 gpurun_out/r4e/in_tests.log
for c in "0.35,0.2"; do
  python tools/train_parity.py --steps 6000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 2000 --eval-batches 16 --dtypes bf16 --seed 1 --hard --hard-contrast $c > gpurun_out/r4e/hard_cal_$c.log 2>&1; tail -n 4 gpurun_out/r4e/hard_cal_$c.log
done
