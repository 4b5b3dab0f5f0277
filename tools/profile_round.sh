#!/bin/bash
# Regenerates the rocprofv3 evidence for profiles/ on the GPU box (run from the repo root through gpurun).
# usage: tools/profile_round.sh TAG      -> gpurun_out/prof_TAG/...   (copy what is to be judged into profiles/)
#   per dtype (bf16, f32), each in its OWN rocprofv3 run (PMC passes never combined with --stats or other trace domains):
#     1. --kernel-trace --stats                                   -> TAG_{dt}_kernel_stats.csv
#     2. --kernel-trace --pmc FETCH_SIZE ; 3. --pmc WRITE_SIZE    -> TAG_hbm_traffic_{dt}.json   (tools/pmc_traffic.py)
#     4. --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE  -> TAG_mfma_busy_{dt}.txt
#   plus the bench lines of BASELINE.json configs[2] (MTnnUNet B=64) and configs[4]'s per-GPU shape (512x512 fp16 B=16),
#   the per-op HIP-event logs (tools/per_op.py, bf16 and f32), and the kernel stats of the EXACT driver command
#   (python bench.py --gpus 1 --steps 20 --warmup 5: bf16 + fp32 parity mode + rooflines in one trace) beside its JSON line.
set -e
TAG=${1:-r04}
OUT=gpurun_out/prof_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-roofline --no-parity-mode"
for DT in bf16 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${DT}_stats -- python3 bench.py --dtype $DT --steps 5 --warmup 2 $COMMON > $OUT/${DT}_stats.log 2>&1
  f=$(ls $OUT/${DT}_stats/*/*kernel_stats.csv | head -1); cp "$f" $OUT/${TAG}_${DT}_kernel_stats.csv
  echo "stats $DT done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${DT}_fetch -- python3 bench.py --dtype $DT --steps 2 --warmup 1 $COMMON > $OUT/${DT}_fetch.log 2>&1
  echo "fetch $DT done"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${DT}_write -- python3 bench.py --dtype $DT --steps 2 --warmup 1 $COMMON > $OUT/${DT}_write.log 2>&1
  echo "write $DT done"
  python3 tools/pmc_traffic.py $OUT/${DT}_fetch $OUT/${DT}_write $OUT/${TAG}_hbm_traffic_${DT}.json
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${DT}_mfma -- python3 bench.py --dtype $DT --steps 2 --warmup 1 $COMMON > $OUT/${DT}_mfma.log 2>&1
  python3 tools/pmc_mfma.py $OUT/${DT}_mfma > $OUT/${TAG}_mfma_busy_${DT}.txt
  echo "mfma $DT done"
done
# BASELINE.json configs[2] and configs[4] (per-GPU shape): one JSON line each + kernel stats
python3 bench.py --arch MTnnUNet --batch 64 --steps 20 --warmup 5 > $OUT/${TAG}_bench_config2_mtnnunet_b64.json 2> $OUT/config2.err
echo "config2 bench done"
python3 bench.py --size 512 --dtype f16 --batch 16 --steps 20 --warmup 5 > $OUT/${TAG}_bench_config4_512_f16_b16.json 2> $OUT/config4.err
echo "config4 bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -- python3 bench.py --arch MTnnUNet --batch 64 --steps 5 --warmup 2 $COMMON > $OUT/c2_stats.log 2>&1
f=$(ls $OUT/c2_stats/*/*kernel_stats.csv | head -1); cp "$f" $OUT/${TAG}_config2_mtnnunet_b64_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -- python3 bench.py --size 512 --dtype f16 --batch 16 --steps 5 --warmup 2 $COMMON > $OUT/c4_stats.log 2>&1
f=$(ls $OUT/c4_stats/*/*kernel_stats.csv | head -1); cp "$f" $OUT/${TAG}_config4_512_f16_b16_kernel_stats.csv
python3 tools/wgrad_probe.py 1 32 1 > $OUT/${TAG}_wgrad_probe.txt 2>&1 || true
PEROP_MIN=0.02 python3 tools/per_op.py MTUNetPlusPlus 32 256 bf16 > $OUT/${TAG}_per_op_bf16.log 2>&1
PEROP_MIN=0.05 python3 tools/per_op.py MTUNetPlusPlus 32 256 f32 > $OUT/${TAG}_per_op_f32.log 2>&1
echo "per-op done"
# the driver's own command, traced: the JSON line and the kernel stats of the SAME run
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver_stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_command.json 2> $OUT/driver_stats.err
f=$(ls $OUT/driver_stats/*/*kernel_stats.csv | head -1); cp "$f" $OUT/${TAG}_driver_command_kernel_stats.csv
echo "all done"
