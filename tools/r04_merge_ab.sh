#!/bin/bash
# a human player's move in the same basic block as the ball's world step (two independent chains for the scheduler):
# tools/ab.py --build --common "-DPZ_DEV_SUBSET=721" merge=-DPZ_MERGE_MOVE=1 mergeilp="-DPZ_MERGE_MOVE=1 -mllvm -amdgpu-sched-strategy=max-ilp" mergeiter="-DPZ_MERGE_MOVE=1 -mllvm -amdgpu-sched-strategy=iterative-ilp"
set -e
O=gpurun_out/r04_merge
mkdir -p $O
python tools/ab.py --slices 2048 base merge mergeilp mergeiter > $O/ab_merge_move_hh_cold.log 2>&1
python tools/ab.py base merge mergeilp mergeiter > $O/ab_merge_move_hh_hot.log 2>&1
python tools/ab.py --ai --slices 2048 base+t merge+t mergeilp+t mergeiter+t > $O/ab_merge_move_cfg3_cold.log 2>&1
python tools/ab.py --slices 2048 base+p merge+p mergeilp+p mergeiter+p > $O/ab_merge_move_packed_cold.log 2>&1
tail -n 6 $O/*.log
