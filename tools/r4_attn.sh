#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_attention.py tests/test_dropout_parity.py tests/test_fullsize.py tests/test_model.py tests/test_ctc.py tests/test_inference.py -m gpu -q -x 2>&1 | tail -3
./tools/bin/attn_bench > $O/attention_bench.txt 2>&1; cat $O/attention_bench.txt
timeout 300 python bench.py --cpu-rows 0 --no-legs > $O/bench.log 2>$O/bench.err; echo bench rc=$?
tail -1 $O/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
u=d['config']['unprofiled']
print({k:d[k] for k in ('value','ms_per_step')}, 'unprof rot %.3f same %.3f' % (u['ms_per_step_rotated'], u['ms_per_step_same_batch']), 'host', d['config']['host_enqueue_ms_per_step'], 'serial', d['roofline']['serial_kernel_ms_per_step'])
for k in d['roofline']['kernels'][:16]: print('%-40s %8.1f %s frac %.3f  %.3f ms/step x%.0f' % (k['kernel'], k['achieved'], k['unit'], k['frac'], k['ms_per_step'], k['launches_per_step']))
"
