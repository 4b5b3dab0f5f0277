#!/bin/bash
# round 4: ConvT k=2 weight-gradient task count re-swept now that a split's tasks share one XCD's L2 (probes build, interleaved)
cd /root/repo
mkdir -p gpurun_out/r4y
export MTBC_LIB=$PWD/multi_task_breast_cancer_amd/libmtbc_hip_probes.so
BENCH_ARGS=--allow-probes ROUNDS=2 bash tools/ab_env.sh "" "MTBC_CT_WG_TASKS=1024" "MTBC_CT_WG_TASKS=2048" "MTBC_CT_WG_TASKS=3072" "MTBC_CT_WG_TASKS=4096" > gpurun_out/r4y/ab.log 2>&1
cat gpurun_out/r4y/ab.log
