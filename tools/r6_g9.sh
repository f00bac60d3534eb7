cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/g9
mkdir -p $O
timeout 900 python -m pytest tests/test_loss_mel.py tests/test_pipeline.py tests/test_abi.py tests/test_torch_ops.py -m gpu -x -q > $O/pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/pytest.log
timeout 600 python - > $O/mel_leg.txt 2>&1 <<'PY'
import json, torch, bench
dev = torch.device('cuda:0')
print(json.dumps(bench.mel_leg(dev)))
print(json.dumps(bench.mel_leg(dev, n_utt=256, seconds=8.0)))
PY
cat $O/mel_leg.txt | tail -3 | cut -c1-700
