#!/bin/bash
# round 4: ablation table of the K-contiguous dW kernel (a -DG8_DW_ABLATION build of gemm8_dw.hip) + idle analysis of the rotated loop
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
( echo "# tools/bin/gemm_bench 20 dw under SS_GEMM_DW_ABL (compile-time ablation masks, results wrong by construction):"
  echo "# 0 = the kernel; 1 no global loads in the steady K steps; 2 no transposes / LDS writes; 3 = 1 + 2; 4 no MFMAs; 7 = 1 + 2 + 4 (fragment reads + barriers only); 8 no C update; 11 = 1 + 2 + 8"
  for a in 0 1 2 3 4 7 8 11; do echo "== SS_GEMM_DW_ABL=$a"; SS_GEMM_DW_ABL=$a timeout 120 ./tools/bin/gemm_bench 20 dw 2>&1 | grep "KT hr4"; done ) > $O/dw_ablation.txt 2>&1
cat $O/dw_ablation.txt
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 10 --warmup 4 --cpu-rows 0 --no-legs --no-profile --no-same > $O/kt_bench.log 2>&1
python tools/gpu_idle.py $(find $O/kt -name "*.db" | head -1) > $O/gpu_idle_rotated.txt; cat $O/gpu_idle_rotated.txt
rm -rf $O/kt
