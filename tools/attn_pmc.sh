#!/bin/bash
# SQ counters of the attention kernels at the benchmark shape (two passes of <= 8 SQ counters, --pmc with --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/attn_pmc; rm -rf $O; mkdir -p $O
B=${ATTN_BENCH_ARGS:-"110 8 200 96 100 0.2 3"}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/p1 -o p1 -- $R/tools/bin/attn_bench $B > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $O/p2 -o p2 -- $R/tools/bin/attn_bench $B > $O/p2.log 2>&1
for p in p1 p2; do python $R/tools/pmc_kernel_table.py $(find $O/$p -name "*.db" | head -1) attn; done > $O/table.txt 2>&1
rm -rf $O/p1 $O/p2
cat $O/table.txt
