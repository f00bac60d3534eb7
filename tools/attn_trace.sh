#!/bin/bash
# per-kernel durations of the attention kernels at the benchmark shape (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/attn_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $R/tools/bin/attn_bench ${ATTN_BENCH_ARGS:-110 8 200 96 100 0.2 10} > $O/kt.log 2>&1
python $R/tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) 10 > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
cut -c1-70,100-200 $O/kernel_stats.txt | head -14
