#!/bin/bash
# Measurement build of the library with per-tile wall-clock stamps in gemm8_kc_kernel (-DG8_STAMPS), next to the product build:
#   tools/g8_stamps.sh            builds tools/bin/stamplib/libsilent_speech_hip.so and tools/bin/g8_stamps
#   LD_LIBRARY_PATH=tools/bin/stamplib tools/bin/g8_stamps [M N K]     (on the GPU box)
set -e
cd "$(dirname "$0")/.."
make -C silent_speech_amd/csrc hip > /dev/null
mkdir -p build/stamp tools/bin/stamplib
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isilent_speech_amd/csrc -Wno-unused-result -Wno-inline-asm -DG8_STAMPS ${G8_EXTRA}"
/opt/rocm/bin/hipcc $FLAGS -c silent_speech_amd/csrc/gemm8.hip -o build/stamp/gemm8.o
OBJS=$(ls build/hip/*.o | grep -v '/gemm8\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/stamplib/libsilent_speech_hip.so $OBJS build/stamp/gemm8.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/g8_stamps.cpp -o tools/bin/g8_stamps -ldl -Ltools/bin/stamplib -lsilent_speech_hip -Wl,-rpath,'$ORIGIN/stamplib'
