# rocprofv3 counter passes over tools/placement.py cases (diagnostic): bash tools/placement_pmc.sh "<placement args>" "<counters>" ...
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pl_pmc
rm -rf $O && mkdir -p $O
ARGS="$1"; shift
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/tools/placement.py $ARGS --replays 2 > $O/p$i.log 2>&1
  grep "us/frame" $O/p$i.log | awk '{print $1,$2,$3,$4,$5,$6}' | tr '\n' ';'; echo
  python3 $R/tools/pmc_cases.py $O/p$i 192
done
find $O -name "*.csv" -size +1M -delete
