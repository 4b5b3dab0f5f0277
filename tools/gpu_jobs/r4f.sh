mkdir -p gpurun_out/r4f
ROUNDS=3 bash tools/ab_env.sh "" "MTBC_X_PL3_SEPARATE=1" > gpurun_out/r4f/ab_pl3.log 2>&1; cat gpurun_out/r4f/ab_pl3.log
python -m pytest tests/test_coop_safety_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/r4f/coop_dp_tests.log 2>&1; tail -n 2 gpurun_out/r4f/coop_dp_tests.log
export MTBC_LIB=$PWD/multi_task_breast_cancer_amd/libmtbc_hip_probes.so
MTBC_WG_TS=1 python tools/experiments/f32_wgrad_one.py 24 24 256 > gpurun_out/r4f/f32_wg_ts.log 2>&1
MTBC_WG_TS=1 python tools/experiments/f32_wgrad_one.py 192 48 128 >> gpurun_out/r4f/f32_wg_ts.log 2>&1
MTBC_WG_TS=1 python tools/experiments/f32_wgrad_one.py 96 96 64 >> gpurun_out/r4f/f32_wg_ts.log 2>&1
grep "wg_ts" gpurun_out/r4f/f32_wg_ts.log | tail -n 3
unset MTBC_LIB
R=$PWD
cd /tmp && export TMPDIR=/tmp
for sh in "24 24 256" "192 48 128"; do
  tag=$(echo $sh | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/r4f/pmc1_$tag -- python3 $R/tools/experiments/f32_wgrad_one.py $sh > $R/gpurun_out/r4f/pmc1_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/r4f/pmc2_$tag -- python3 $R/tools/experiments/f32_wgrad_one.py $sh > $R/gpurun_out/r4f/pmc2_$tag.log 2>&1
done
cd $R
for d in gpurun_out/r4f/pmc*_*/; do python3 tools/pmc_summary.py $d conv3x3_wgrad_mfma > ${d%/}.txt 2>&1; done; tail -n 12 gpurun_out/r4f/pmc1_24_24_256.txt
python tools/train_parity.py --steps 6000 --batch 16 --size 256 --lr 3e-4 --cosine --eval-every 2000 --eval-batches 32 --dtypes f32,bf16 --seed 1 --hard > gpurun_out/r4f/hard_b16_s1.log 2>&1; tail -n 8 gpurun_out/r4f/hard_b16_s1.log
