#!/bin/bash
# Round 5, attention: parity of the transposed-score kernels on the MI355X, A/B against the 16 x 16 resident kernels (SS_ATTN_T=0), SQ counters.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${1:-r5a}; mkdir -p $O
timeout 900 python -m pytest tests/test_attention.py tests/test_dropout_parity.py tests/test_abi.py -m gpu -q > $O/pytest_attn.log 2>&1; echo pytest rc=$?; tail -4 $O/pytest_attn.log
tools/bin/attn_bench > $O/attn_t.txt 2>&1; SS_ATTN_T=0 tools/bin/attn_bench > $O/attn_old.txt 2>&1; cat $O/attn_t.txt $O/attn_old.txt
summ() { tail -1 $1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, 'enqueue', d['config']['host_enqueue_ms_per_step'], 'serial', d['roofline']['serial_kernel_ms_per_step'], 'rot', d['config']['unprofiled'])
for k in d['roofline']['kernels']: print('%-40s %8.1f %s frac %.3f  %.3f ms/step x%.0f' % (k['kernel'], k['achieved'], k['unit'], k['frac'], k['ms_per_step'], k['launches_per_step']))
"; }
timeout 400 python bench.py --cpu-rows 0 --no-legs --steps 10 > $O/bench_t.log 2>&1; echo bench rc=$?; summ $O/bench_t.log
SS_ATTN_T=0 timeout 400 python bench.py --cpu-rows 0 --no-legs --steps 10 > $O/bench_old.log 2>&1; echo bench old rc=$?; summ $O/bench_old.log
bash tools/attn_pmc.sh > $O/attn_pmc.txt 2>&1; cat $O/attn_pmc.txt | tail -30
