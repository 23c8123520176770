#!/bin/bash
set -e
O=gpurun_out/r04_kframe
mkdir -p $O
python tools/ab.py --rollout 32 generic base r03 > $O/ab_rollout_hh_order2.log 2>&1
python tools/ab.py --rollout 32 base r03 generic > $O/ab_rollout_hh_order3.log 2>&1
python tools/ab.py --rollout 128 --frames 6400 r03 base generic > $O/ab_rollout_hh_k128.log 2>&1
tail -n 5 $O/ab_rollout_hh_order2.log $O/ab_rollout_hh_order3.log $O/ab_rollout_hh_k128.log
