#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python bench.py --cpu-rows 0 --no-legs > $O/bench.log 2>$O/bench.err; echo bench rc=$?; tail -3 $O/bench.err
tail -1 $O/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, 'unprof', d['config']['unprofiled'], 'host', d['config']['host_enqueue_ms_per_step'], 'serial', d['roofline']['serial_kernel_ms_per_step'])
for k in d['roofline']['kernels']: print('%-40s %8.1f %s frac %.3f  %.3f ms/step x%.0f' % (k['kernel'], k['achieved'], k['unit'], k['frac'], k['ms_per_step'], k['launches_per_step']))
"
timeout 300 python tools/host_profile.py 12 > $O/host_profile.txt 2>&1; head -60 $O/host_profile.txt | cut -c1-150
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 12 --warmup 4 --cpu-rows 0 --no-legs --no-profile --no-same > $O/kt_bench.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/gpu_idle.py $DB 0.6 > $O/gpu_idle_rotated.txt; cat $O/gpu_idle_rotated.txt
python tools/rocprof_summary.py $DB 16 > $O/kernel_stats.txt; head -30 $O/kernel_stats.txt | cut -c1-60,100-170
rm -rf $O/kt
