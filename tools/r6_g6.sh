cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/g6_ctc
mkdir -p $O
timeout 600 python -m pytest tests/test_ctc.py -m gpu -x -q > $O/pytest_ctc.log 2>&1; echo pytest rc=$?; tail -3 $O/pytest_ctc.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ctc -- python $GRAFT_REPO_ROOT/tools/ctc_probe.py > $O/probe.log 2>&1; echo rc=$?
grep utterances $O/probe.log
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -5 "$f" | cut -c1-200; fi
