#!/bin/bash
# GPU suite, smoke() and the default bench line of the tree as it stands (second session of round 4)
O=gpurun_out/r04_final2
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/gputest.log
tail -6 $O/gputest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 $O/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench rc=$rc"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_final2/bench_default.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','build_id')}), d['roofline']['frac'], d['cpu_baseline']['value'])
for k,v in d['roofline'].get('by_config',{}).items(): print('  ',k, {kk:vv for kk,vv in v.items() if kk in ('launch_us','us_per_frame','frac','parity_bit_exact')})
PY
