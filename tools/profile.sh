#!/bin/bash
# tools/profile.sh TAG [SECTION ...] -- rocprofv3 evidence for profiles/ (run on the GPU box through gpurun).
#
#   kt        --kernel-trace --stats of the default bench line (all single-GPU BASELINE configs in one run)
#   pmc_hh    FETCH_SIZE / WRITE_SIZE / SQ passes on the headline (random/random, 65 536 games)
#   pmc_cfg3  the same passes on config 3 (player 2 = computer, flight look-up tables)
#   pmc_cfg3q the same passes on config 3 with the power-hit table alone (the landing point predicted in the kernel)
#   pmc_cfg3c the same passes on config 3 with the flight predictors computed in the kernel (scout-wave launch)
#   pmc_big   FETCH_SIZE / WRITE_SIZE at 524 288 games (config 4's total size: past the Infinity Cache)
#   kt_roll   --kernel-trace --stats of pz_rollout_random / pz_step_many (k = 32)
#   pmc_roll  the four passes on the k-frame kernels (pz_rollout_random human and player 2 = computer, pz_step_many; k = 32)
#   kt_hh     --kernel-trace --stats of the headline workload ALONE (no other config shares its kernel row)
#   kt_cfg3   --kernel-trace --stats of config 3 on both tables ALONE (in `kt` its kernel row also holds the power-hit-table runs)
#   calib     FETCH_SIZE / WRITE_SIZE on known-byte kernels of the step kernels' access widths (tools/calib_traffic.hip)
#   pmc_more  FETCH_SIZE / WRITE_SIZE of the remaining bench entries: config 2 (4 096 games), config 5 (fused wrappers),
#             config 3 on the packed format, int32 state + int16 observations
#   pmc_pk    the four passes on the headline workload with the packed state format (65 536 games)
#   pmc_pkbig FETCH_SIZE / WRITE_SIZE / kernel stats at 524 288 games with the packed state format
#   pmc_ph    FETCH_SIZE / WRITE_SIZE / kernel stats: packed state + int16 observations, 65 536 and 524 288 games
#
# Every PMC pass is its own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; gpurun refuses --pmc together with the API trace domains).
# Raw output lands in gpurun_out/prof_$TAG/; tools/pmc_summary.py condenses it into profiles/.
set -u
TAG=${1:?tag}
shift
SECTIONS=${*:-kt kt_hh kt_cfg3 pmc_hh pmc_cfg3 pmc_cfg3q pmc_cfg3c pmc_big pmc_more kt_roll pmc_roll pmc_pk pmc_pkbig pmc_ph calib}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PY=python3
# the build the counters are collected on, recorded BEFORE the first run (pmc_summary.py refuses another one afterwards)
$PY tools/pmc_summary.py --record-build "$OUT/build_at_collection.json" || { echo "could not record the build"; exit 1; }
SQ_A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ_B="SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"

run() {  # name, rocprof args..., -- , program args
    local name=$1
    shift
    echo "== $name" | tee -a "$OUT/log.txt"
    rocprofv3 -d "$OUT/$name" -o run -f csv "$@" > "$OUT/$name.log" 2>&1 || { echo "rocprofv3 failed: $name (see $OUT/$name.log)" | tee -a "$OUT/log.txt"; tail -5 "$OUT/$name.log"; return 1; }
}

pmc_fw() {  # prefix, bench args...: the two traffic passes only
    local p=$1
    shift
    run "${p}_fetch" --kernel-trace --pmc FETCH_SIZE -- $PY bench.py "$@" || return 1
    run "${p}_write" --kernel-trace --pmc WRITE_SIZE -- $PY bench.py "$@" || return 1
}

pmc_set() {  # prefix, bench args...
    local p=$1
    shift
    run "${p}_fetch" --kernel-trace --pmc FETCH_SIZE -- $PY bench.py "$@" || return 1
    run "${p}_write" --kernel-trace --pmc WRITE_SIZE -- $PY bench.py "$@" || return 1
    run "${p}_sqa" --kernel-trace --pmc $SQ_A -- $PY bench.py "$@" || return 1
    run "${p}_sqb" --kernel-trace --pmc $SQ_B -- $PY bench.py "$@" || return 1
}

for s in $SECTIONS; do
    case $s in
    kt) run kt --kernel-trace --stats -- $PY bench.py --no-cpu || exit 1 ;;
    pmc_hh) pmc_set hh --no-cpu --no-configs --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1 ;;
    pmc_cfg3) pmc_set cfg3 --no-cpu --no-configs --p2-computer --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1 ;;
    pmc_cfg3q) pmc_set cfg3q --no-cpu --no-configs --p2-computer --flight-tables power_hit --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1 ;;
    pmc_cfg3c) pmc_set cfg3c --no-cpu --no-configs --p2-computer --no-flight-tables --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1 ;;
    pmc_big)
        run big_fetch --kernel-trace --pmc FETCH_SIZE -- $PY bench.py --no-cpu --no-configs --num-envs 524288 --steps 30 --warmup 10 --burn-in 256 --launch cabi || exit 1
        run big_write --kernel-trace --pmc WRITE_SIZE -- $PY bench.py --no-cpu --no-configs --num-envs 524288 --steps 30 --warmup 10 --burn-in 256 --launch cabi || exit 1
        run big_kt --kernel-trace --stats -- $PY bench.py --no-cpu --no-configs --num-envs 524288 --steps 300 --warmup 50 --burn-in 256 --launch cabi || exit 1
        ;;
    pmc_pk) pmc_set pk --no-cpu --no-configs --state-format packed --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1 ;;
    pmc_pkbig)
        run pkbig_fetch --kernel-trace --pmc FETCH_SIZE -- $PY bench.py --no-cpu --no-configs --state-format packed --num-envs 524288 --steps 30 --warmup 10 --burn-in 256 --launch cabi || exit 1
        run pkbig_write --kernel-trace --pmc WRITE_SIZE -- $PY bench.py --no-cpu --no-configs --state-format packed --num-envs 524288 --steps 30 --warmup 10 --burn-in 256 --launch cabi || exit 1
        run pkbig_kt --kernel-trace --stats -- $PY bench.py --no-cpu --no-configs --state-format packed --num-envs 524288 --steps 300 --warmup 50 --burn-in 256 --launch cabi || exit 1
        ;;
    pmc_ph)
        run ph_fetch --kernel-trace --pmc FETCH_SIZE -- $PY bench.py --no-cpu --no-configs --state-format packed --int16-obs --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1
        run ph_write --kernel-trace --pmc WRITE_SIZE -- $PY bench.py --no-cpu --no-configs --state-format packed --int16-obs --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1
        run phbig_fetch --kernel-trace --pmc FETCH_SIZE -- $PY bench.py --no-cpu --no-configs --state-format packed --int16-obs --num-envs 524288 --steps 30 --warmup 10 --burn-in 256 --launch cabi || exit 1
        run phbig_write --kernel-trace --pmc WRITE_SIZE -- $PY bench.py --no-cpu --no-configs --state-format packed --int16-obs --num-envs 524288 --steps 30 --warmup 10 --burn-in 256 --launch cabi || exit 1
        run phbig_kt --kernel-trace --stats -- $PY bench.py --no-cpu --no-configs --state-format packed --int16-obs --num-envs 524288 --steps 300 --warmup 50 --burn-in 256 --launch cabi || exit 1
        ;;
    pmc_more)
        pmc_fw cfg2 --no-cpu --no-configs --num-envs 4096 --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1
        pmc_fw cfg5 --no-cpu --no-configs --wrappers --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1
        pmc_fw pkcfg3 --no-cpu --no-configs --p2-computer --state-format packed --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1
        pmc_fw i16 --no-cpu --no-configs --int16-obs --steps 40 --warmup 10 --burn-in 512 --launch cabi || exit 1
        ;;
    kt_roll) run roll --kernel-trace --stats -- $PY bench.py --no-cpu --no-configs --rollouts || exit 1 ;;
    pmc_roll) pmc_set roll --no-cpu --no-configs --rollouts || exit 1 ;;
    kt_hh) run hh_kt --kernel-trace --stats -- $PY bench.py --no-cpu --no-configs || exit 1 ;;
    kt_cfg3) run cfg3_kt --kernel-trace --stats -- $PY bench.py --no-cpu --no-configs --p2-computer || exit 1 ;;
    calib)
        [ -x tools/bin/calib_traffic ] || { mkdir -p tools/bin && hipcc -O3 --offload-arch=gfx950 tools/calib_traffic.hip -o tools/bin/calib_traffic; } || exit 1
        run calib_fetch --kernel-trace --pmc FETCH_SIZE -- tools/bin/calib_traffic || exit 1
        run calib_write --kernel-trace --pmc WRITE_SIZE -- tools/bin/calib_traffic || exit 1
        ;;
    *) echo "unknown section $s"; exit 2 ;;
    esac
done
# condense on the box (gpurun brings back at most 64 MiB of gpurun_out/), then drop the raw traces
$PY tools/pmc_summary.py "$TAG" > "$OUT/summary.log" 2>&1 || { echo "pmc_summary failed"; tail -5 "$OUT/summary.log"; exit 1; }
find "$OUT" -mindepth 1 -maxdepth 1 -type d ! -name out -exec rm -rf {} +
echo "done: $OUT" | tee -a "$OUT/log.txt"
