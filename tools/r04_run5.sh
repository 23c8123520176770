#!/bin/bash
O=gpurun_out/r04_run5
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputest.log
tail -25 $O/gputest.log
