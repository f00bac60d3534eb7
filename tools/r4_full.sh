#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r4f; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python bench.py --cpu-rows 0 --no-legs > $O/bench.log 2>$O/bench.err; echo bench rc=$?; tail -2 $O/bench.err
tail -1 $O/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
u=d['config']['unprofiled']
print({k:d[k] for k in ('value','ms_per_step')}, 'unprof rot %.3f same %.3f' % (u['ms_per_step_rotated'], u['ms_per_step_same_batch']), 'host', d['config']['host_enqueue_ms_per_step'], 'serial', d['roofline']['serial_kernel_ms_per_step'])
"
timeout 300 python tools/host_profile.py 12 > $O/host_profile.txt 2>&1; head -40 $O/host_profile.txt | cut -c1-150
