#!/bin/bash
# a PLAIN instantiation of the single-frame pair kernel (no fused wrapper / statistics / float rows: those configuration
# words compile-time constants): 3 111 -> 2 291 instructions.  tools/ab.py --build --common "-DPZ_DEV_SUBSET=705" plain=-DPZ_PAIR_PLAIN=1
set -e
O=gpurun_out/r04_plain
mkdir -p $O
python tools/ab.py --slices 2048 base plain > $O/ab_pair_plain_hh_cold.log 2>&1
python tools/ab.py base plain > $O/ab_pair_plain_hh_hot.log 2>&1
python tools/ab.py --ai --slices 2048 base+t plain+t > $O/ab_pair_plain_cfg3_cold.log 2>&1
python tools/ab.py --n 4096 base plain > $O/ab_pair_plain_4096.log 2>&1
python tools/ab.py --n 262144 base plain > $O/ab_pair_plain_262144.log 2>&1
tail -n 4 $O/*.log
