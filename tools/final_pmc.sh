#!/bin/bash
# The HBM traffic counters alone (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, kernel trace only) -> profiles/r05_pmc_traffic.*,
# then the bench line that reads them (gpurun_out/final2/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/final3; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o pf -- python bench.py --steps 4 --warmup 1 --cpu-rows 0 --no-legs --no-profile --no-same > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o pw -- python bench.py --steps 4 --warmup 1 --cpu-rows 0 --no-legs --no-profile --no-same > $O/pw.log 2>&1
python tools/pmc_summary.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -rf $O/pf $O/pw
cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json
head -6 $O/pmc_traffic.txt | cut -c1-200
bash tools/final_rerun.sh
