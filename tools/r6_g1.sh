cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm.py -m gpu -q -x -k "planes" > gpurun_out/g1_pytest_gemm.log 2>&1; echo gemm rc=$?; tail -3 gpurun_out/g1_pytest_gemm.log
timeout 600 python tools/x3_gemm_probe.py > gpurun_out/g1_x3_probe.txt 2>&1; echo probe rc=$?; cat gpurun_out/g1_x3_probe.txt
timeout 900 python -m pytest tests/test_model.py tests/test_fullsize.py -m gpu -q -x -k "bf16x3" > gpurun_out/g1_pytest_x3.log 2>&1; echo x3 rc=$?; tail -5 gpurun_out/g1_pytest_x3.log
timeout 600 python bench.py --dtype fp32x3 --no-legs --cpu-rows 0 --steps 8 > gpurun_out/g1_bench_x3_planes.log 2>&1; echo bench rc=$?; tail -1 gpurun_out/g1_bench_x3_planes.log | cut -c1-600
SS_AMD_X3_PLANES=0 timeout 600 python bench.py --dtype fp32x3 --no-legs --cpu-rows 0 --steps 8 > gpurun_out/g1_bench_x3_regs.log 2>&1; echo bench rc=$?; tail -1 gpurun_out/g1_bench_x3_regs.log | cut -c1-600
