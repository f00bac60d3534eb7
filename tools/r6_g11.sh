cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/g11
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.log 2>&1; echo pytest rc=$?; tail -15 $O/pytest_gpu.log
