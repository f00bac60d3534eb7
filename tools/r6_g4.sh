cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python tools/ctc_probe.py 2>&1 | grep -v "^W2026" | tail -2
timeout 600 python -m pytest tests/test_ctc.py -m gpu -q -x 2>&1 | tail -2
