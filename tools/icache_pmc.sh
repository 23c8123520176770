# Instruction-fetch counters of the single-frame step launches (diagnostic): bash tools/icache_pmc.sh
# Two passes per workload (headline, config 3): the I-cache's requests / hits / misses, and the fetch count with its
# accumulated in-flight level (level / fetches = average fetch latency in cycles).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; rm -rf $O; mkdir -p $O
ARGS="--steps 40 --warmup 10 --burn-in 512 --no-cpu --no-configs --launch cabi"
for wl in hh cfg3; do
    extra=""; [ $wl = cfg3 ] && extra="--p2-computer"
    rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/${wl}_a -o t --output-format csv -- python3 $R/bench.py $ARGS $extra > $O/${wl}_a.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/${wl}_b -o t --output-format csv -- python3 $R/bench.py $ARGS $extra > $O/${wl}_b.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*_[ab]")):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if int(r["Grid_Size"]) >= 65536 and "step_pair_kernel" in r["Kernel_Name"]:
                rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in rows.items():
        print(d.rsplit("/", 1)[1], k, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, len(next(iter(c.values()))), flush=True)
PY
find $O -name "*.csv" -size +1M -delete
