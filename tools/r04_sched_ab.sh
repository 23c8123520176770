#!/bin/bash
# the compiler's scheduling strategy for the single-frame pair kernels (is a wave's frame a chain of dependent instructions?):
# tools/ab.py --build --common "-DPZ_DEV_SUBSET=721" ilp="-mllvm -amdgpu-sched-strategy=max-ilp" iter="-mllvm -amdgpu-sched-strategy=iterative-ilp" memcl="-mllvm -amdgpu-sched-strategy=max-memory-clause"
set -e
O=gpurun_out/r04_sched
mkdir -p $O
python tools/ab.py --slices 2048 base ilp iter memcl > $O/ab_sched_strategy_hh_cold.log 2>&1
python tools/ab.py --ai --slices 2048 base+t ilp+t iter+t memcl+t > $O/ab_sched_strategy_cfg3_cold.log 2>&1
python tools/ab.py --slices 2048 base+p ilp+p iter+p memcl+p > $O/ab_sched_strategy_packed_cold.log 2>&1
tail -n 6 $O/*.log
