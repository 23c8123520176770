#!/bin/bash
# Round 4, config 3: look-ahead rows (pz_physics.hpp) -- the computer's wave works out the NEXT frame's look-ups and
# draws behind its stores and leaves them in a row per game; the next launch's decision needs no gather and no Philox.
# Variants: tools/ab.py --build --common "-DPZ_DEV_SUBSET=721" look1=-DPZ_LOOKAHEAD=1 look2=-DPZ_LOOKAHEAD=2
# (look2: the human player's wave also defers its round-start boldness draw behind its stores, as in a human-vs-human launch)
set -e
O=gpurun_out/r04_look
mkdir -p $O
python tools/ab.py --ai --slices 2048 base+t look1+t look2+t > $O/ab_lookahead_cold_tape.log 2>&1
python tools/ab.py --ai base+t look1+t look2+t > $O/ab_lookahead_hot_tape.log 2>&1
python tools/ab.py --ai --slices 2048 base+tp look1+tp look2+tp > $O/ab_lookahead_packed.log 2>&1
tail -n 8 $O/*.log
