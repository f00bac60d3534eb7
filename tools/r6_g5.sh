cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/g5_ctc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ctc -- python $GRAFT_REPO_ROOT/tools/ctc_probe.py > $O/probe.log 2>&1; echo rc=$?
ls -R $O | head -20
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -8 "$f" | cut -c1-220; fi
