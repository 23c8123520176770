#!/bin/bash
# Round 4: long parity runs of the final build against the CPU oracle on every lane (tests/soak.py).
O=gpurun_out/r04_soak
mkdir -p $O
timeout -k 10 500 python tests/soak.py --frames 60000 --every 10000 2>&1 | tee $O/r04_soak_65536x60000_final_build.log | tail -3
timeout -k 10 200 python tests/soak.py --frames 20000 --every 5000 --packed 2>&1 | tee $O/r04_soak_packed_65536x20000_final_build.log | tail -3
timeout -k 10 200 python tests/soak.py --frames 19200 --every 4800 --rollout 32 2>&1 | tee $O/r04_soak_kframe_rollout32_65536x19200_final_build.log | tail -3
timeout -k 10 200 python tests/soak.py --frames 19200 --every 4800 --rollout 160 --tape 2>&1 | tee $O/r04_soak_kframe_tape160_65536x19200_final_build.log | tail -3
timeout -k 10 200 python tests/soak.py --frames 19200 --every 4800 --rollout 64 --tape --packed 2>&1 | tee $O/r04_soak_kframe_tape64_packed_65536x19200_final_build.log | tail -3
