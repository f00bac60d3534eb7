cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_attention.py -m gpu -q -x -k "x3_planes" > gpurun_out/g2_pytest_attn.log 2>&1; echo attn rc=$?; tail -3 gpurun_out/g2_pytest_attn.log
timeout 1200 python -m pytest tests/test_model.py tests/test_fullsize.py tests/test_dropout_parity.py -m gpu -q -x -k "bf16x3 or attention_mask" > gpurun_out/g2_pytest_x3.log 2>&1; echo x3 rc=$?; tail -5 gpurun_out/g2_pytest_x3.log
timeout 600 python bench.py --dtype fp32x3 --no-legs --cpu-rows 0 --steps 8 > gpurun_out/g2_bench_x3.log 2>&1; echo bench rc=$?; tail -1 gpurun_out/g2_bench_x3.log | cut -c1-300
SS_AMD_X3_ATTENTION=0 timeout 600 python bench.py --dtype fp32x3 --no-legs --cpu-rows 0 --steps 8 > gpurun_out/g2_bench_x3_noattn.log 2>&1; echo bench rc=$?; tail -1 gpurun_out/g2_bench_x3_noattn.log | cut -c1-300
timeout 600 python tools/x3_gemm_probe.py > gpurun_out/g2_x3_probe.txt 2>&1; echo probe rc=$?; cat gpurun_out/g2_x3_probe.txt
