#!/bin/bash
# Builds variants of the library that differ in lstm.hip's OVL_DBG knob (measurement only) into build/var/: libbsrnn_dbgN.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/var
F="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize"
for v in "$@"; do
  ( /opt/rocm/bin/hipcc $F -DOVL_DBG=$v -c speechseparation_amd/csrc/lstm.hip -o build/var/lstm_dbg$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined -o build/var/libbsrnn_dbg$v.so \
      $(ls build/obj/*.o | grep -v /lstm.o) build/var/lstm_dbg$v.o ) &
done
wait
ls -la build/var/*.so
