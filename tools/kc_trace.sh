cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/kctrace; rm -rf $O; mkdir -p $O
SS_AMD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 6 --warmup 2 --cpu-rows 0 --no-legs --no-profile > $O/bench.log 2>&1
python tools/kc_launches.py $(find $O/kt -name "*.db" | head -1) 8 > $O/launches.txt
rm -rf $O/kt; head -120 $O/launches.txt
