cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/g14
mkdir -p $O
timeout 1500 python -m pytest tests/test_dropout_parity.py tests/test_gemm.py tests/test_fullsize.py tests/test_model.py tests/test_ctc.py -m gpu -x -q > $O/pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/pytest.log
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export SS_AMD_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/abtmp/lib_old.so; else unset SS_AMD_LIBRARY; fi
    python bench.py --steps 30 --warmup 5 --no-profile --no-legs --no-same --cpu-rows 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3))"
  done
done
unset SS_AMD_LIBRARY
python - <<'PY' 2>&1 | tail -2
import json, torch, bench
print(json.dumps(bench.ctc_leg(torch.device('cuda:0'))['ctc_loss'])[:300])
PY
