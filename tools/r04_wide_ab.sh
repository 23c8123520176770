#!/bin/bash
# human vs human pair kernel: the always-written state columns as 16-byte pieces through LDS (3 stores per wave instead of
# 9 / 11 dword stores; same bytes, same columns).  tools/ab.py --build --common "-DPZ_DEV_SUBSET=705" wide=-DPZ_WIDE_STATE_STORES=1
set -e
O=gpurun_out/r04_wide
mkdir -p $O
python tools/ab.py --slices 2048 base wide > $O/ab_wide_state_stores_cold_tape.log 2>&1
python tools/ab.py base wide > $O/ab_wide_state_stores_hot_tape.log 2>&1
python tools/ab.py --wrappers --slices 2048 base wide > $O/ab_wide_state_stores_cfg5.log 2>&1
python tools/ab.py --n 262144 base wide > $O/ab_wide_state_stores_262144.log 2>&1
python tools/ab.py --n 4096 base wide > $O/ab_wide_state_stores_4096.log 2>&1
tail -n 5 $O/*.log
