#!/bin/bash
# Same-box A/B of two library builds inside the training step, per kernel instantiation (rocprofv3 kernel stats):
#   tools/ab_prof.sh <dir of the other libsilent_speech_hip.so> [name filter]     (on the GPU box; the other build e.g. in tools/bin/ablib)
export TMPDIR=/tmp; mkdir -p gpurun_out
OTHER=${1:-tools/bin/ablib}; FILT=${2:-gemm8_kc}
for L in ${LIBS:-$OTHER silent_speech_amd/lib $OTHER silent_speech_amd/lib}; do
  D=gpurun_out/abp_$$; rm -rf $D
  SS_AMD_LIBRARY=$PWD/$L/libsilent_speech_hip.so rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py --cpu-rows 0 --no-legs --no-profile --no-same --steps 8 --warmup 4 > $D.log 2>&1
  echo "== $L"; f=$(find $D -name "*.db" | head -1)
  python3 - "$f" "$FILT" <<'PY'
import sqlite3,sys
cur=sqlite3.connect(sys.argv[1]).cursor()
tot=0
for n,c,s_,a in cur.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"):
    if sys.argv[2] in n:
        tot+=s_
        print('  %-50s calls %5d avg %9.1f us total %9.3f ms' % (n[n.index('<'):][:50] if '<' in n else n[:50], c, a/1e3, s_/1e6))
print('  total %.3f ms' % (tot/1e6))
PY
  rm -rf $D $D.log
done
