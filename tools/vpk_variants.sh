#!/bin/bash
# Variant builds of the library for the co-residency hazard bisection (tools/vpk_scalarize.py): fft.hip compiled WITH the SLP vectorizer to a
# device listing, chosen classes of v_pk_*_f32 rewritten to scalar pairs, the listing assembled and wrapped back into an object that replaces
# build/obj/fft.o in the link.   bash tools/vpk_variants.sh NAME CLASSES [extra vpk_scalarize.py arguments]   ->  build/libbsrnn_vpk_NAME.so
set -e
NAME=$1; CLASSES=$2; shift 2
LLVM=/opt/rocm/lib/llvm/bin
F=speechseparation_amd/csrc/fft.hip
W=build/vpk; mkdir -p $W
[ -f $W/src_slp.s ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o $W/src_slp.s $F 2>/dev/null
if [ "$CLASSES" = "-" ]; then cp $W/src_slp.s $W/fft_$NAME.s; else python3 tools/vpk_scalarize.py $W/src_slp.s $W/fft_$NAME.s $CLASSES "$@"; fi
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/fft_$NAME.s -o $W/fft_$NAME.dev.o
$LLVM/ld.lld -shared $W/fft_$NAME.dev.o -o $W/fft_$NAME.hsaco
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/fft_$NAME.hsaco -output=$W/fft_$NAME.hipfb
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/fft_$NAME.hipfb -c $F -o $W/fft_$NAME.o 2>/dev/null
OBJS=$(ls build/obj/*.o | grep -v "/fft.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined -o build/libbsrnn_vpk_$NAME.so $OBJS $W/fft_$NAME.o
echo "built build/libbsrnn_vpk_$NAME.so: $(grep -c 'v_pk_add_f32' $W/fft_$NAME.s) add $(grep -c 'v_pk_mul_f32' $W/fft_$NAME.s) mul $(grep -c 'v_pk_fma_f32' $W/fft_$NAME.s) fma $(grep -c 'v_pk_mov_b32' $W/fft_$NAME.s) mov packed left"
