#!/bin/bash
# look-ahead rows, what costs what (timing-only variants, results wrong by construction; PZ_LOOK_TIMING bits: 1 every lane
# served, 2 no Philox in the tail, 4 no gathers in the tail, 8 no row stores, 16 no row loads)
set -e
O=gpurun_out/r04_look
mkdir -p $O
python tools/ab.py --ai --slices 2048 --no-check base+t look1+t t31+t t15+t t13+t t11+t t7+t t1+t > $O/ablate_lookahead_parts.log 2>&1
tail -n 12 $O/ablate_lookahead_parts.log
