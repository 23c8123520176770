#!/bin/bash
# Round 4: the k-frame kernels after the scalar-pressure work (PLAIN instantiations, column offsets recomputed behind the
# frame loop, the reward table parked in VGPRs) against round 3's library (ab_r03.so) and against their own generic forms.
set -e
O=gpurun_out/r04_kframe
mkdir -p $O
python tools/ab.py --rollout 32 r03 base generic > $O/ab_rollout_hh.log 2>&1
python tools/ab.py --rollout 32 --ai r03+t base+t generic+t > $O/ab_rollout_p2_computer.log 2>&1
python tools/ab.py --rollout 32 --tape r03 base generic > $O/ab_tape_hh.log 2>&1
python tools/ab.py --rollout 32 --tape --ai r03+t base+t generic+t > $O/ab_tape_p2_computer.log 2>&1
tail -n 6 $O/*.log
