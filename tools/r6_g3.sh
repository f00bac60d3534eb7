cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ctc.py tests/test_attention.py -m gpu -q -x > gpurun_out/g3_pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/g3_pytest.log
timeout 900 python bench.py > gpurun_out/g3_bench_full.log 2>&1; echo bench rc=$?; tail -1 gpurun_out/g3_bench_full.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step']); print('ctc', json.dumps(d.get('ctc'))[:1500]); print('parity_grade', d.get('parity_grade')); print('dtw', json.dumps(d.get('dtw'))[:600])"
