#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
timeout 120 ./tools/bin/gemm_bench 20 dw > $O/dw_full.txt 2>&1; grep -v "128-wide" $O/dw_full.txt
for a in 8; do echo "== ABL $a"; SS_GEMM_DW_ABL=$a timeout 120 ./tools/bin/gemm_bench 20 dw 2>&1 | grep "KT hr4"; done > $O/dw_abl.txt 2>&1
cat $O/dw_abl.txt
