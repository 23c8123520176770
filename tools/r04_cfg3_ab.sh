#!/bin/bash
# Round 4, config 3 (player 2 = computer, flight tables, 65 536 games, cold action tape): the review's two A/B questions
# (32 games per wave; the human wave's own-player stores in front of the exchange barrier) and the cumulative
# compile-time ablation of the launch.  Variants: tools/ab.py --build --common "-DPZ_DEV_SUBSET=705" ....
set -e
O=gpurun_out/r04_cfg3
mkdir -p $O
python tools/ab.py --ai --slices 2048 base+t g32ai+t early+t g32early+t > $O/ab_g32_and_early_stores.log 2>&1
python tools/ab.py --slices 2048 base g32hh s512 s768 > $O/ab_hh_g32_and_store_ablation.log 2>&1
python tools/ab.py --ai --slices 2048 base+t s512+t s768+t s806+t s807+t s815+t s831+t > $O/ablate_stores_first.log 2>&1
python tools/ab.py --ai --slices 2048 base+t a6+t a38+t a39+t a47+t a63+t s831+t > $O/ablate_decision_first.log 2>&1
tail -n 12 $O/*.log
