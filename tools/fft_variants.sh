#!/bin/bash
# Builds variants of the library that differ in fft.hip's occupancy targets (measurement only) into build/var/: libbsrnn_fftS_I.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/var
F="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize"
for v in "$@"; do
  s=${v%_*}; i=${v#*_}
  ( /opt/rocm/bin/hipcc $F -DFFT_OCC_STFT=$s -DFFT_OCC_ISTFT=$i -c speechseparation_amd/csrc/fft.hip -o build/var/fft_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined -o build/var/libbsrnn_fft$v.so \
      $(ls build/obj/*.o | grep -v /fft.o) build/var/fft_$v.o ) &
done
wait
