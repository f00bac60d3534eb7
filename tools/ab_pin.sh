#!/bin/bash
# A/B of two BUILDS of libsilent_speech_hip.so (and of the burst / spread schedule of the 8-wave GEMM) inside the real training step and in
# tools/gemm_bench, on ONE box -- boxes differ by several percent, so only numbers of one call compare.
#   OTHER=/path/to/other/libsilent_speech_hip.so bash tools/ab_pin.sh        (e.g. the library of the previous commit, built in a worktree)
summ() { tail -1 $1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$2', {k:round(d[k],3) for k in ('value','ms_per_step')}, 'serial', round(d['roofline']['serial_kernel_ms_per_step'],3), [(k['kernel'][:14], round(k['ms_per_step'],3)) for k in d['roofline']['kernels'][:2]])
"; }
OTHER=${OTHER:-$PWD/tools/bin/oldlib/libsilent_speech_hip.so}
for r in 1 2; do
  if [ -f "$OTHER" ]; then SS_AMD_LIBRARY=$OTHER timeout 200 python bench.py --cpu-rows 0 --no-legs --steps 20 > gpurun_out/ab_other.log 2>/dev/null; summ gpurun_out/ab_other.log other; fi
  SS_GEMM8_PIN=0 timeout 200 python bench.py --cpu-rows 0 --no-legs --steps 20 > gpurun_out/ab_pin0.log 2>/dev/null; summ gpurun_out/ab_pin0.log burst
  timeout 200 python bench.py --cpu-rows 0 --no-legs --steps 20 > gpurun_out/ab_pin3.log 2>/dev/null; summ gpurun_out/ab_pin3.log spread
done
if [ -f "$OTHER" ]; then LD_LIBRARY_PATH=$(dirname $OTHER) ./tools/bin/gemm_bench 20 kc 2>&1 | grep -E "auto" | sed 's/^/other /'; fi
./tools/bin/gemm_bench 20 kc 2>&1 | grep -E "auto" | sed 's/^/this  /'
