cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/x3; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm.py tests/test_attention.py -m gpu -q -x -k "bf16x3" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_fullsize.py -m gpu -q -x -s -k "bf16x3 or fp32_vs" 2>&1 | grep -v "^$" | cut -c1-600 | tail -12
SS_AMD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --dtype fp32x3 --steps 3 --warmup 1 --cpu-rows 0 --no-legs --no-profile --no-same > $O/bench.log 2>&1
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) 4 > $O/kernel_stats.txt; rm -rf $O/kt
head -12 $O/kernel_stats.txt | cut -c1-90,100-170
timeout 300 python bench.py --dtype fp32x3 --steps 6 --warmup 2 --cpu-rows 0 --no-legs --no-profile --no-same 2>&1 | tail -1 | cut -c1-250
