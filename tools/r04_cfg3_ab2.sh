#!/bin/bash
# Round 4, config 3, second pass: what the human player's wave can do while it waits at the exchange barrier
# (its own player's columns / + the ball's position, trail and rotation stored in front of the barrier) and the
# look-up after a collision taken over by the human player's wave.
set -e
O=gpurun_out/r04_cfg3
mkdir -p $O
python tools/ab.py --ai --slices 2048 base+t early+t early2+t early2h+t early1h+t hum+t > $O/ab_early_stores_and_after_hit_by_human.log 2>&1
python tools/ab.py --ai --slices 2048 base+tp early2h+tp hum+tp > $O/ab_after_hit_by_human_packed.log 2>&1
python tools/ab.py --ai base+t early+t early2+t early2h+t > $O/ab_early_stores_hot_tape.log 2>&1
tail -n 9 $O/ab_early_stores_and_after_hit_by_human.log $O/ab_after_hit_by_human_packed.log $O/ab_early_stores_hot_tape.log
