export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $O/pytest_gpu.log
python bench.py > $O/bench.json.log 2>$O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 6 --warmup 2 --cpu-rows 0 > $O/kt_bench.log 2>&1
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) 8 > $O/kernel_stats.txt
SS_AMD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python bench.py --steps 6 --warmup 2 --cpu-rows 0 --no-profile > $O/ks_bench.log 2>&1
python tools/rocprof_summary.py $(find $O/ks -name "*.db" | head -1) 8 > $O/serial_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o pf -- python bench.py --steps 2 --warmup 1 --cpu-rows 0 --no-profile > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o pw -- python bench.py --steps 2 --warmup 1 --cpu-rows 0 --no-profile > $O/pw.log 2>&1
python tools/pmc_summary.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -rf $O/kt $O/ks $O/pf $O/pw
cat $O/pytest_gpu.log; tail -1 $O/bench.json.log | cut -c1-400; head -6 $O/kernel_stats.txt | cut -c1-60,100-170; head -6 $O/serial_kernel_stats.txt | cut -c1-60,100-170; head -5 $O/pmc_traffic.txt
