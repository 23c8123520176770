#!/bin/bash
# Round 4, config 3, third pass (on top of the shipped early stores): the computer's wave moves the ball and issues its
# two gathers before the players' round start and the action decode (eg); its own player stored in front of the
# barrier too (es3); the state written in front of the observation rows (sf).
set -e
O=gpurun_out/r04_cfg3
mkdir -p $O
python tools/ab.py --ai --slices 2048 base+t eg+t es3+t eg3+t sf+t egsf+t > $O/ab_early_gather_cold_tape.log 2>&1
python tools/ab.py --ai base+t eg+t es3+t sf+t > $O/ab_early_gather_hot_tape.log 2>&1
python tools/ab.py --ai --slices 2048 base+tp eg+tp > $O/ab_early_gather_packed.log 2>&1
tail -n 9 $O/ab_early_gather_cold_tape.log $O/ab_early_gather_hot_tape.log $O/ab_early_gather_packed.log
