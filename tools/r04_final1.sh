#!/bin/bash
O=gpurun_out/r04_final
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputest.log
tail -6 $O/gputest.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 900 python bench.py --extra > $O/bench_default_extra.json 2> $O/bench_extra.err; echo "bench extra rc=$?"
python - <<'PY'
import json
for f in ("bench_default", "bench_default_extra"):
    d=json.loads(open(f'gpurun_out/r04_final/{f}.json').read().strip().splitlines()[-1])
    print(f, json.dumps({k:d[k] for k in ('value','ms_per_step','build_id')}))
    for k,v in d['roofline'].get('by_config',{}).items(): print('  ',k, v)
    for k,v in d.get('extra',{}).items():
        if isinstance(v,dict) and ('launch_us' in v or 'us_per_frame' in v): print('  extra',k,{kk:vv for kk,vv in v.items() if kk in ('launch_us','wall_us_per_step','us_per_frame','host_us_per_call')})
PY
