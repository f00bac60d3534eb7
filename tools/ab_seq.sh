#!/bin/bash
# Same-box A/B of two library builds, launch by launch: the last LAST dispatches of the kernels matching FILT in the training step, side by side.
#   tools/ab_seq.sh <dir of the other library> [name filter] [LAST]
export TMPDIR=/tmp; mkdir -p gpurun_out
OTHER=${1:-tools/bin/ablib}; FILT=${2:-gemm8_kc}; LAST=${3:-132}
i=0
for L in $OTHER silent_speech_amd/lib; do
  D=gpurun_out/abs_$i; rm -rf $D
  SS_AMD_LIBRARY=$PWD/$L/libsilent_speech_hip.so rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py --cpu-rows 0 --no-legs --no-profile --no-same --steps 8 --warmup 4 > $D.log 2>&1
  i=$((i+1))
done
python3 - "$FILT" "$LAST" <<'PY'
import sqlite3,sys,glob
def seq(d):
    cur=sqlite3.connect(glob.glob(d+'/**/*.db',recursive=True)[0]).cursor()
    cols=[r[1] for r in cur.execute("pragma table_info(kernels)")]
    g='grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
    q="select name, start, end-start%s from kernels order by start" % ((', '+g) if g else '')
    rows=[r for r in cur.execute(q) if sys.argv[1] in r[0]]
    return rows[-int(sys.argv[2]):]
a=seq('gpurun_out/abs_0'); b=seq('gpurun_out/abs_1')
ta=tb=0
for x,y in zip(a,b):
    na=x[0][x[0].index('<'):][:34]; ta+=x[2]; tb+=y[2]
    print('%-36s grid %6s  other %8.1f  this %8.1f  %+6.1f %%' % (na, x[3] if len(x)>3 else '?', x[2]/1e3, y[2]/1e3, 100.0*(y[2]-x[2])/x[2]))
print('sum other %.3f ms this %.3f ms' % (ta/1e6, tb/1e6))
PY
rm -rf gpurun_out/abs_0 gpurun_out/abs_1 gpurun_out/abs_0.log gpurun_out/abs_1.log
