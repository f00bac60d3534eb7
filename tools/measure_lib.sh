#!/bin/bash
# Measurement / A-B build of the library next to the product build (tools/bin/stamplib/libsilent_speech_hip.so): what the release library
# deliberately does NOT contain --
#   -DG8_STAMPS        per-tile wall-clock stamps in gemm8_kc_kernel                       (tools/bin/g8_stamps)
#   -DATTN_T_MEASURE   phase stamps of the transposed-score attention forward (SS_ATTN_T_STAMPS=1, tools/bin/attn_bench) and
#                      SS_ATTN_T_SKIP=1|2 (time one backward kernel of the pair)
#   -DSS_ATTN_RES16    the LDS-resident 16 x 16 attention family of rounds 1-4 (SS_ATTN_T=0: A/B against the transposed-score kernels)
# Use:  LD_LIBRARY_PATH=tools/bin/stamplib tools/bin/attn_bench   or   SS_AMD_LIBRARY=tools/bin/stamplib/libsilent_speech_hip.so python ...
set -e
cd "$(dirname "$0")/.."
make -C silent_speech_amd/csrc hip > /dev/null
mkdir -p build/stamp tools/bin/stamplib
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isilent_speech_amd/csrc -Wno-unused-result -Wno-inline-asm -DG8_STAMPS -DATTN_T_MEASURE -DSS_ATTN_RES16 ${G8_EXTRA}"
for f in gemm8 attention attention_t; do /opt/rocm/bin/hipcc $FLAGS -c silent_speech_amd/csrc/$f.hip -o build/stamp/$f.o & done; wait
OBJS=$(ls build/hip/*.o | grep -v -E '/(gemm8|attention|attention_t)\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/stamplib/libsilent_speech_hip.so $OBJS build/stamp/gemm8.o build/stamp/attention.o build/stamp/attention_t.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/g8_stamps.cpp -o tools/bin/g8_stamps -ldl -Ltools/bin/stamplib -lsilent_speech_hip -Wl,-rpath,'$ORIGIN/stamplib'
