cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/g8
mkdir -p $O
timeout 600 python tools/step_copies.py > $O/step_copies.txt 2>&1; echo rc=$?
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-profile --no-legs --no-same --cpu-rows 0 > $O/bench_trace.log 2>&1; echo rc=$?
tail -2 $O/bench_trace.log | cut -c1-400
ls -la $O
