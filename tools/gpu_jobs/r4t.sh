#!/bin/bash
# round 4: split counts that keep a split's channel blocks on one XCD (3x3 weight gradient, ConvT weight gradient): per-launch probe + A/B + the op tests
cd /root/repo
mkdir -p gpurun_out/r4t
export MTBC_LIB=$PWD/multi_task_breast_cancer_amd/libmtbc_hip_probes.so
for v in 0 16 8; do
  echo "== MTBC_WG_NS8=$v" >> gpurun_out/r4t/wgrad_probe.txt
  MTBC_WG_NS8=$v PROBE_REDUCE=launch timeout -k 10 200 python tools/wgrad_probe.py 1 32 1 >> gpurun_out/r4t/wgrad_probe.txt 2>&1 || exit 1
done
BENCH_ARGS=--allow-probes ROUNDS=2 bash tools/ab_env.sh "MTBC_WG_NS8=0 MTBC_CT_WG_XCD=0" "MTBC_CT_WG_XCD=0" "MTBC_WG_NS8=0" "" "MTBC_WG_NS8=8" > gpurun_out/r4t/ab.log 2>&1
cat gpurun_out/r4t/ab.log
unset MTBC_LIB
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "wgrad or convT" > gpurun_out/r4t/ops_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4t/ops_tests.log; tail -n 3 gpurun_out/r4t/ops_tests.log
