#!/bin/bash
# per-kernel durations of the DTW probe cases (rocprofv3 kernel trace; tools/dtw_probe.py case indices as arguments)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/dtwtr; rocprofv3 --kernel-trace --stats -d /tmp/dtwtr -o t -- python $R/tools/dtw_probe.py $c > /tmp/dtwtr.log 2>&1
  echo "== case $c: $(grep ' us ' /tmp/dtwtr.log | tail -1)"
  python $R/tools/rocprof_summary.py $(find /tmp/dtwtr -name "*.db" | head -1) 1 | grep -i "dtw\|^kernel" | cut -c1-40,100-170
done
