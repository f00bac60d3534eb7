#!/bin/bash
# Round-end measurement on the GPU box: GPU tests, the default bench line, rocprofv3 kernel traces (overlapped and serial), the HBM
# traffic counters (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, kernel trace only) and the SQ counters of the attention kernels.
# Everything lands in gpurun_out/final/; copy what is to be judged into profiles/ (tracked).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $O/pytest_gpu.log
SS_ATTN_T=0 ./tools/bin/attn_bench > $O/attention_bench_16x16.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 8 --warmup 4 --cpu-rows 0 --no-legs --no-profile --no-same > $O/kt_bench.log 2>&1
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) 12 > $O/kernel_stats.txt
python tools/gpu_idle.py $(find $O/kt -name "*.db" | head -1) 0.5 > $O/gpu_idle_rotated.txt
SS_AMD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python bench.py --steps 8 --warmup 4 --cpu-rows 0 --no-legs --no-profile --no-same > $O/ks_bench.log 2>&1
python tools/rocprof_summary.py $(find $O/ks -name "*.db" | head -1) 12 > $O/serial_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o pf -- python bench.py --steps 4 --warmup 1 --cpu-rows 0 --no-legs --no-profile --no-same > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o pw -- python bench.py --steps 4 --warmup 1 --cpu-rows 0 --no-legs --no-profile --no-same > $O/pw.log 2>&1
python tools/pmc_summary.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -rf $O/kt $O/ks $O/pf $O/pw
cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json            # the bench line's `traffic` fields come from THIS run's counters
python bench.py > $O/bench.json.log 2>$O/bench.err
for ss in 1 0; do SS_AMD_SIDE_STREAM=$ss python bench.py --no-profile --no-legs --cpu-rows 0 --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side_stream=$ss', d['ms_per_step'], d['config']['unprofiled'])"; done > $O/side_stream_ab.txt 2>&1
bash tools/attn_trace.sh > /dev/null 2>&1; cp gpurun_out/attn_trace/kernel_stats.txt $O/attention_kernel_stats.txt
bash tools/attn_pmc.sh > /dev/null 2>&1; cp gpurun_out/attn_pmc/table.txt $O/attention_sq_counters.txt
./tools/bin/attn_bench > $O/attention_bench.txt 2>&1
./tools/bin/mfma_peak 4000 > $O/mfma_peak.txt 2>&1
./tools/bin/gemm_bench 20 > $O/gemm_bench.txt 2>&1
./tools/bin/gemm_bench 20 epi >> $O/gemm_bench.txt 2>&1
timeout 300 python tools/host_profile.py 12 > $O/host_profile.txt 2>&1
# the f32-storage / bf16 x 3 mode: serial kernel trace of its step, and its GEMM against the exact-f32 / bf16 kernels and its own copy floor
SS_AMD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kx -o kx -- python bench.py --dtype fp32x3 --steps 3 --warmup 1 --cpu-rows 0 --no-legs --no-profile --no-same > $O/kx_bench.log 2>&1
python tools/rocprof_summary.py $(find $O/kx -name "*.db" | head -1) 4 > $O/x3_kernel_stats.txt; rm -rf $O/kx
PYTHONPATH=. timeout 300 python tools/x3_gemm_probe.py > $O/x3_gemm_probe.txt 2>&1
(./tools/bin/dtw_bench 64 1000 10; ./tools/bin/dtw_bench 256 1000 10; SS_DTW_DEBUG=8 ./tools/bin/dtw_bench 64 1000 10; SS_DTW_DEBUG=8 ./tools/bin/dtw_bench 256 1000 10; ./tools/bin/ta_probe) > $O/dtw_bench.txt 2>&1
# the 8-wave GEMM tile by tile (measurement build with stamps: tools/g8_stamps.sh, built here beforehand), against the previous build where present
if [ -x tools/bin/g8_stamps ] && [ -f tools/bin/stamplib/libsilent_speech_hip.so ]; then
  (for s in "22000 2304 768" "22000 768 768" "22000 768 2304" "22000 3072 768"; do tools/bin/g8_stamps $s; done
   if [ -x tools/bin/g8_stamps_old ]; then for s in "22000 2304 768" "22000 768 768" "22000 768 2304" "22000 3072 768"; do echo "== $s, 12 rotating output / A buffers: this build | the build before the ring / direct epilogue"; tools/bin/g8_stamps $s 12 | head -1; tools/bin/g8_stamps_old $s 12 | head -1; done; fi) > $O/g8_stamps.txt 2>&1
fi
if [ -f tools/bin/ablib/libsilent_speech_hip.so ]; then bash tools/ab_seq.sh tools/bin/ablib gemm8_kc 66 2>&1 | grep -v "^W2026" > $O/g8_ab_in_step.txt; fi
timeout 300 python tools/pipeline_profile.py > $O/pipeline_profile.txt 2>&1
cat $O/pytest_gpu.log; cat $O/side_stream_ab.txt; cat $O/mfma_peak.txt; tail -1 $O/bench.json.log | cut -c1-400; head -8 $O/kernel_stats.txt | cut -c1-60,100-170; head -6 $O/serial_kernel_stats.txt | cut -c1-60,100-170; head -8 $O/pmc_traffic.txt; cat $O/gpu_idle_rotated.txt; cat $O/attention_bench.txt; cat $O/dtw_bench.txt; PYTHONPATH=. python tools/mel_probe.py > $O/mel_probe.txt 2>&1; PYTHONPATH=. python tools/ctc_probe.py 2>&1 | grep utterances > $O/ctc_probe.txt; cat $O/mel_probe.txt $O/ctc_probe.txt
