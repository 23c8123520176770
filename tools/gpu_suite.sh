#!/bin/bash
# tools/gpu_suite.sh CASE [TAG] -- the GPU-box runs of a round, one parametrised script (run through gpurun; output under
# gpurun_out/TAG/, copy what is to be judged into profiles/).  Every step runs under its own `timeout -k 10`, and a step
# that fails ends the script: no GPU step is started behind a failed one.
#
#   check   GPU test suite, __graft_entry__.smoke(), the default bench line (+ its --configs-out file)
#   tables  config 3 in the three flight-table modes (both / power-hit table only / none), interleaved in one process on one
#           library, cold and hot action tape; then DESIGN 4.2's early stores re-measured on the round's kernels
#   bench   the default bench line and the --extra line only
#   driver  the bench line exactly as the driver runs it (--steps 20 --warmup 5)
#   variants  the bench line's other workloads and launch modes, headline only (--no-configs): config 3, config 5, packed,
#           int16 rows, hot tape, direct C-ABI launches, env.step(), two ranks on one device over gloo -- each must print
#           a line with its in-run oracle parity true
#   edge    config 3's early stores: A/B of the LDS hand-shake against round 4's unordered form and against no
#           early stores, then the deterministic proof of the edge: the computer's wave held back ~16 000 cycles in front of
#           its first load must stay bit-exact with the hand-shake and breaks without it.  The variants are diagnostic
#           builds (csrc/pz_diagnostic.hpp), compiled on the box by the case itself (subset 705: seconds each)
#   edge_soak  the one-computer configurations (config 3, p1 computer) for 60 000 frames against the oracle on a build whose
#           computer's wave is ALWAYS late (bits 8-15 = 1, full library): every launch takes the hand-shake's late-store path
#   chains  one batch as two / four sub-batch chains in ONE hipGraph (fork / join at its ends) and as separate graphs on
#           separate streams, human vs human and config 3, 65 536 and 131 072 games (tools/chains.py)
#   soak    long parity runs against the CPU oracle on every lane (tests/soak.py), single-frame and k-frame, both formats
#   sweep   the randomized C-ABI configuration sweep, PZ_SWEEP_TRIALS (default 40 000) configurations
#   profile tools/profile.sh TAG (all sections): the rocprofv3 evidence of the round
set -u
CASE=${1:?case}
TAG=${2:-r06_$CASE}
O=gpurun_out/$TAG
mkdir -p "$O"
step() {  # seconds, log, command...: run one step, stop the script when it fails
    local secs=$1 log=$2
    shift 2
    echo "== $*" | tee -a "$O/steps.log"
    timeout -k 10 "$secs" "$@" > "$O/$log" 2>&1
    local rc=$?
    echo "rc=$rc ($log)" | tee -a "$O/steps.log"
    [ $rc -eq 0 ] || { tail -n 25 "$O/$log"; exit $rc; }
}
bench_summary() {
    python3 - "$@" <<'PY'
import json, sys
for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(path, len(json.dumps(d)), "bytes;", json.dumps({k: d[k] for k in ("value", "ms_per_step")}), d["config"].get("build_id"))
    print("  ", {k: v for k, v in r.items() if isinstance(v, (int, float, bool)) or v is None})
PY
}
case $CASE in
check)
    step 1000 gputest.log python3 -m pytest tests -m gpu -x -q
    tail -n 4 "$O/gputest.log"
    step 300 smoke.log python3 -c "import __graft_entry__ as g; g.smoke()"
    tail -n 1 "$O/smoke.log"
    timeout -k 10 600 python3 bench.py --configs-out "$O/bench_configs.json" > "$O/bench_default.json" 2> "$O/bench_default.err" || { echo "bench failed"; tail "$O/bench_default.err"; exit 1; }
    bench_summary "$O/bench_default.json"
    ;;
bench)
    timeout -k 10 600 python3 bench.py --configs-out "$O/bench_configs.json" > "$O/bench_default.json" 2> "$O/bench_default.err" || { echo "bench failed"; tail "$O/bench_default.err"; exit 1; }
    timeout -k 10 900 python3 bench.py --extra --configs-out "$O/bench_extra_configs.json" > "$O/bench_default_extra.json" 2> "$O/bench_extra.err" || { echo "bench --extra failed"; tail "$O/bench_extra.err"; exit 1; }
    bench_summary "$O/bench_default.json" "$O/bench_default_extra.json"
    ;;
driver)
    timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2> "$O/bench_driver.err" || { echo "bench failed"; tail "$O/bench_driver.err"; exit 1; }
    bench_summary "$O/bench_driver.json"
    ;;
variants)
    for v in "--p2-computer" "--wrappers" "--state-format packed" "--int16-obs" "--action-tape hot --steps 20 --warmup 5" \
             "--launch cabi --steps 500" "--launch api --steps 500" "--p2-computer --no-flight-tables" "--p1-computer"; do
        name=$(echo "$v" | tr -c 'a-z0-9\n' '_')
        step 300 "bench$name.json" python3 bench.py --no-configs --min-time 0.1 --cpu-seconds 2 $v
    done
    step 400 quickstart.log python3 examples/quickstart.py
    PZ_BENCH_ONE_DEVICE=1 timeout -k 10 400 python3 bench.py --gpus 2 --no-configs --min-time 0.1 --dist-backend gloo --p2-computer > "$O/bench_two_ranks_cfg3.json" 2> "$O/bench_two_ranks_cfg3.err" || { echo "two ranks failed"; tail "$O/bench_two_ranks_cfg3.err"; exit 1; }
    python3 - "$O"/bench*.json <<'PY'
import json, sys
bad = 0
for path in sys.argv[1:]:
    d = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
    r, c = d["roofline"], d["config"]
    ok = r["parity_first_last_rank_bit_exact"] is True and len(json.dumps(d)) < 8000
    bad += not ok
    print(f"{path.rsplit('/', 1)[1]:60s} {d['value'] / 1e9:6.2f} G  {r['launch_us']:6.2f} us  frac {r['frac']:.3f}  traffic {r['traffic']}  stale {r['traffic_stale']}  "
          f"tape {c['action_tape']}  parity {r['parity_first_last_rank_bit_exact']} ({r['parity_ranks_checked']} ranks)  {len(json.dumps(d))} B")
sys.exit(1 if bad else 0)
PY
    ;;
tables)
    # (pz_diagnostic.hpp bits: 32 = no early stores, 16 = early stores without the hand-shake; subset 2753 = the single-frame
    # kernels -- pair and single-wave / scout --, human vs human and player 2 = computer)
    step 600 build_variants.log python3 tools/ab.py --build --subset 2753 early0=32 unordered=16
    step 400 ab_flight_table_modes_cold_tape.log python3 tools/ab.py --ai --slices 2048 --samples base+t base+q base
    step 400 ab_flight_table_modes_hot_tape.log python3 tools/ab.py --ai --samples base+t base+q base
    step 400 ab_early_stores_cold_tape.log python3 tools/ab.py --ai --slices 2048 --samples base+t early0+t unordered+t
    step 400 ab_early_stores_hot_tape.log python3 tools/ab.py --ai --samples base+t early0+t unordered+t
    tail -n 6 "$O"/ab_*.log
    ;;
edge)
    step 600 build_variants.log python3 tools/ab.py --build --subset 705 unordered=16 early0=32 delayedge=0x200 delayunordered=0x210
    step 300 ab_early_store_edge_cold_tape.log python3 tools/ab.py --ai --slices 2048 base+t unordered+t early0+t
    step 300 ab_early_store_edge_hot_tape.log python3 tools/ab.py --ai base+t unordered+t early0+t
    step 300 early_store_edge_partner_held_back.log python3 tools/ab.py --ai --slices 2048 base+t delayedge+t delayunordered+t
    tail -n 8 "$O"/*.log
    ;;
edge_soak)
    step 900 build_variants.log python3 tools/ab.py --build delayfull=0x100
    step 600 soak_one_computer_partner_always_late_65536x60000.log python3 tests/soak.py --frames 60000 --every 10000 --only "computer, flight tables" --lib tools/bin/ab_delayfull.so
    tail -n 4 "$O"/soak_one_computer*.log
    ;;
chains)
    step 300 chains_65536.log python3 tools/chains.py --separate
    step 300 chains_65536_cfg3.log python3 tools/chains.py --ai --separate
    step 300 chains_131072.log python3 tools/chains.py --n 131072 --separate
    step 300 chains_65536_policy_fused.log python3 tools/chains.py --random --separate
    tail -n 6 "$O"/chains*.log
    ;;
soak)
    step 500 soak_65536x60000.log python3 tests/soak.py --frames 60000 --every 10000
    step 200 soak_packed_65536x20000.log python3 tests/soak.py --frames 20000 --every 5000 --packed
    step 200 soak_kframe_rollout32_65536x19200.log python3 tests/soak.py --frames 19200 --every 4800 --rollout 32
    step 200 soak_kframe_tape160_65536x19200.log python3 tests/soak.py --frames 19200 --every 4800 --rollout 160 --tape
    step 200 soak_kframe_tape64_packed_65536x19200.log python3 tests/soak.py --frames 19200 --every 4800 --rollout 64 --tape --packed
    tail -n 3 "$O"/soak*.log
    ;;
sweep)
    export PZ_SWEEP_TRIALS=${PZ_SWEEP_TRIALS:-40000}
    step 1100 "config_sweep_${PZ_SWEEP_TRIALS}.log" python3 -m pytest tests/test_gpu_parity.py -x -q -k randomized_config_sweep -s
    tail -n 4 "$O"/config_sweep*.log
    ;;
profile)
    bash tools/profile.sh "$TAG"
    ;;
*)
    echo "unknown case $CASE"
    exit 2
    ;;
esac
