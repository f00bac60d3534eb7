#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r4g; rm -rf $O; mkdir -p $O
for cfg in "SS_BN_RED_THREADS=256" "SS_BN_RED_THREADS=0" "SS_BN_RED_THREADS=0 SS_BN_CHUNKS=1024" "SS_BN_RED_THREADS=0 SS_BN_CHUNKS=512" "SS_BN_RED_THREADS=96 SS_BN_CHUNKS=1536"; do
env $cfg timeout 300 python bench.py --cpu-rows 0 --no-legs --no-same --steps 10 > $O/bench.log 2>$O/bench.err; echo "bench $cfg rc=$?"
tail -1 $O/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, 'serial', d['roofline']['serial_kernel_ms_per_step'])
for k in d['roofline']['kernels']:
    if 'bn_' in k['kernel'] or 'colsum' in k['kernel']: print('%-40s %8.1f %s frac %.3f  %.3f ms/step x%.0f' % (k['kernel'], k['achieved'], k['unit'], k['frac'], k['ms_per_step'], k['launches_per_step']))
"
done
