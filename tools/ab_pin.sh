# A/B of two builds / schedules of the 8-wave GEMM inside the real training step, on ONE box (boxes differ by several percent)
summ() { tail -1 $1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$2', {k:round(d[k],3) for k in ('value','ms_per_step')}, 'serial', round(d['roofline']['serial_kernel_ms_per_step'],3), [(k['kernel'][:14], round(k['ms_per_step'],3)) for k in d['roofline']['kernels'][:2]])
"; }
for r in 1 2; do
SS_AMD_LIBRARY=$PWD/tools/bin/oldlib/libsilent_speech_hip.so timeout 200 python bench.py --cpu-rows 0 --no-legs --steps 20 > gpurun_out/ab_old.log 2>/dev/null; summ gpurun_out/ab_old.log old
SS_GEMM8_PIN=0 timeout 200 python bench.py --cpu-rows 0 --no-legs --steps 20 > gpurun_out/ab_pin0.log 2>/dev/null; summ gpurun_out/ab_pin0.log pin0
timeout 200 python bench.py --cpu-rows 0 --no-legs --steps 20 > gpurun_out/ab_pin3.log 2>/dev/null; summ gpurun_out/ab_pin3.log pin3
done
LD_LIBRARY_PATH=$PWD/tools/bin/oldlib ./tools/bin/gemm_bench 20 kc 2>&1 | grep -E "auto" | sed 's/^/old /'
./tools/bin/gemm_bench 20 kc 2>&1 | grep -E "auto|ni9" | sed 's/^/new /'
