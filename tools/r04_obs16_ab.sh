#!/bin/bash
# int16 observation rows in the k-frame launches: flushed with 8-byte lanes (one conflict-free LDS read per lane and piece,
# nine 512-byte stores per tensor) against the shipped 16-byte lanes (two reads at a 32-byte lane stride: two-way bank
# conflicts, five 1 KB stores).  Variants: tools/ab.py --build --common "-DPZ_DEV_SUBSET=15596" l8=-DPZ_OBS16_LANES8=1
set -e
O=gpurun_out/r04_obs16
mkdir -p $O
python tools/ab.py --rollout 32 base+h l8+h > $O/ab_rollout_hh_int16.log 2>&1
python tools/ab.py --rollout 32 --ai base+th l8+th > $O/ab_rollout_p2_computer_int16.log 2>&1
python tools/ab.py --rollout 32 --tape base+h l8+h > $O/ab_tape_hh_int16.log 2>&1
tail -n 6 $O/*.log
