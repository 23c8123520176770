#!/bin/bash
# a second long run of the randomized configuration sweep on the final build, other configurations (PZ_SWEEP_SEED)
O=gpurun_out/r04_sweep2
mkdir -p $O
PZ_SWEEP_SEED=4 PZ_SWEEP_TRIALS=60000 timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -k randomized_config_sweep -s > $O/r04_config_sweep_seed4_60000_final_build.log 2>&1; echo "sweep rc=$?"; tail -4 $O/r04_config_sweep_seed4_60000_final_build.log
