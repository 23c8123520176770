#!/bin/bash
O=gpurun_out/r04_sweep
mkdir -p $O
PZ_SWEEP_TRIALS=40000 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -k randomized_config_sweep -s > $O/r04_config_sweep_40000_final_build.log 2>&1; echo "sweep rc=$?"; tail -4 $O/r04_config_sweep_40000_final_build.log
