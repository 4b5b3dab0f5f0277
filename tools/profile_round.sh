#!/bin/bash
# Regenerates the rocprofv3 evidence for profiles/ on the GPU box (run from the repo root through gpurun).
# usage: tools/profile_round.sh TAG      -> gpurun_out/prof_TAG_{bf16,f32}_{stats,fetch,write}
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
for DT in bf16 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${DT}_stats -- python3 bench.py --dtype $DT --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode > gpurun_out/prof_${TAG}_${DT}_stats.log 2>&1
  echo "stats $DT done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_${DT}_fetch -- python3 bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode > gpurun_out/prof_${TAG}_${DT}_fetch.log 2>&1
  echo "fetch $DT done"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_${DT}_write -- python3 bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode > gpurun_out/prof_${TAG}_${DT}_write.log 2>&1
  echo "write $DT done"
  python3 tools/pmc_traffic.py gpurun_out/prof_${TAG}_${DT}_fetch gpurun_out/prof_${TAG}_${DT}_write gpurun_out/${TAG}_hbm_traffic_${DT}.json
  f=$(ls gpurun_out/prof_${TAG}_${DT}_stats/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/${TAG}_${DT}_kernel_stats.csv
done
