# per-channel L2 counters (rocprofv3 JSON keeps the 16 x 8 TCC instances apart) of tools/bin/wplace's sets (diagnostic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wpl_json; rm -rf $O; mkdir -p $O
export WPLACE_REPS=2
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_BUSY TCC_EA0_WRREQ_LEVEL -d $O -o w --output-format json -- $R/tools/bin/wplace 65536 32 6 > $O/w.log 2>&1
cat $O/w.log | tail -n 20
python3 $R/tools/wplace_channels.py $O/w_results.json
rm -f $O/w_results.json
