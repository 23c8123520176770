#!/bin/bash
# final build of round 4 (pz_probe_launch added; the step kernels instruction-for-instruction those of be5d7d8da7016c63):
# GPU suite, smoke(), the two bench lines
O=gpurun_out/r04_final3
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/gputest.log
tail -4 $O/gputest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 $O/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py --extra > $O/bench_default_extra.json 2> $O/bench_extra.err; echo "bench extra rc=$?"
python - <<'PY'
import json
for f in ("bench_default", "bench_default_extra"):
    d=json.loads(open(f'gpurun_out/r04_final3/{f}.json').read().strip().splitlines()[-1])
    print(f, json.dumps({k:d[k] for k in ('value','ms_per_step','build_id')}), d['roofline']['frac'])
    print('  launch_floor', {k:v for k,v in d['roofline']['launch_floor'].items() if k != 'note'})
    for k,v in d['roofline'].get('by_config',{}).items(): print('  ',k, {kk:vv for kk,vv in v.items() if kk in ('launch_us','us_per_frame','frac','parity_bit_exact')})
PY
