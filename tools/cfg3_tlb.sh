# TLB counters of the config-3 step launch (diagnostic): bash tools/cfg3_tlb.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cfg3_tlb; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum -d $O -o t --output-format csv -- python3 $R/bench.py --p2-computer --steps 200 --warmup 20 --no-cpu --no-configs > $O/bench.log 2>&1
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r["Grid_Size"]) >= 65536 and ("step_pair_kernel" in r["Kernel_Name"] or "step_kernel" in r["Kernel_Name"]):
            rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, len(next(iter(d.values()))))
PY
find $O -name "*.csv" -size +1M -delete
