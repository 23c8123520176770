#!/bin/bash
# Headline (65 536 games) and config 2 (4 096 games) graph-replayed launches under ROCclr runtime knobs:
# does the inter-kernel part of a launch (dispatch, fences, kernarg fetch) move with any of them?
# Each setting is its own process (the knobs are read at HIP initialisation).  Output: gpurun_out/r04_knobs/summary.log
set -u
out=gpurun_out/r04_knobs; mkdir -p $out
run() { # name, env assignments...
  name=$1; shift
  for n in 65536 4096; do
    env "$@" timeout -k 10 300 python bench.py --no-configs --no-cpu --num-envs $n > $out/${name}_$n.json 2> $out/${name}_$n.err
    python - "$name" "$n" $out/${name}_$n.json <<'PY' >> gpurun_out/r04_knobs/summary.log
import json, sys
name, n, path = sys.argv[1:4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(f"{name:44s} n={n:>6s}  {d['ms_per_step']*1e3:7.3f} us/launch  kernel {d['roofline'].get('launch_us', float('nan')):7.3f}  value {d['value']/1e9:6.3f} G")
except Exception as e:
    print(f"{name:44s} n={n:>6s}  FAILED {e!r}")
PY
  done
  tail -2 $out/summary.log
}
: > $out/summary.log
run base            PZ_NOP=1
run base_again      PZ_NOP=1
run dev_kernarg_0   HIP_FORCE_DEV_KERNARG=0
run dev_kernarg_1   HIP_FORCE_DEV_KERNARG=1
run graph_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run graph_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run opt_flush_0     AMD_OPT_FLUSH=0
run opt_flush_1     AMD_OPT_FLUSH=1
run kernarg_copy_0  DEBUG_HIP_KERNARG_COPY_OPT=0
run kernarg_copy_1  DEBUG_HIP_KERNARG_COPY_OPT=1
run fgs_kernarg_0   ROC_USE_FGS_KERNARG=0
run fgs_kernarg_1   ROC_USE_FGS_KERNARG=1
run sys_signal_0    ROC_SYSTEM_SCOPE_SIGNAL=0
run hwq_1           GPU_MAX_HW_QUEUES=1
run graph_batch_1   DEBUG_HIP_GRAPH_BATCH_SIZE=1
run graph_batch_4096 DEBUG_HIP_GRAPH_BATCH_SIZE=4096
run base_last       PZ_NOP=1
cat $out/summary.log
