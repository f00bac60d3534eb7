timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_d.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/pytest_gpu_d.log
timeout 400 python bench.py --cpu-rows 0 --no-legs > gpurun_out/bench_d.log 2>&1; echo bench rc=$?; tail -1 gpurun_out/bench_d.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['host_enqueue_ms_per_step'], 'serial', d['roofline']['serial_kernel_ms_per_step'])
for k in d['roofline']['kernels']: print('%-40s %8.1f %s frac %.3f  %.3f ms/step x%.0f' % (k['kernel'], k['achieved'], k['unit'], k['frac'], k['ms_per_step'], k['launches_per_step']))
"
