# per-kernel-name totals of the 8-wave GEMM in the training step for two builds (rocprofv3 kernel trace, serial schedule)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for tag in old new; do
  O=gpurun_out/abtrace_$tag; rm -rf $O; mkdir -p $O
  if [ $tag = old ]; then export SS_AMD_LIBRARY=$R/tools/bin/oldlib/libsilent_speech_hip.so; else unset SS_AMD_LIBRARY; fi
  SS_AMD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 6 --warmup 2 --cpu-rows 0 --no-legs --no-profile > $O/bench.log 2>&1
  python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) 8 | grep -E "gemm8_kc|total kernel" | cut -c1-70,100-150
  rm -rf $O/kt
done
