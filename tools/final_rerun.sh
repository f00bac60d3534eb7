cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/final2; rm -rf $O; mkdir -p $O

python bench.py > $O/bench.json.log 2>$O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 8 --warmup 4 --cpu-rows 0 --no-legs --no-profile --no-same > $O/kt_bench.log 2>&1
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) 12 > $O/kernel_stats.txt
python tools/gpu_idle.py $(find $O/kt -name "*.db" | head -1) 0.5 > $O/gpu_idle_rotated.txt
rm -rf $O/kt
timeout 300 python tools/pipeline_profile.py > $O/pipeline_profile.txt 2>&1
tail -1 $O/bench.json.log | cut -c1-300; cat $O/gpu_idle_rotated.txt | head -5
