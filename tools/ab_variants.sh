#!/bin/bash
# A/B of compile-time variants of one kernel file on ONE GPU box (box-to-box spread is +-3 %, more than most tuning steps):
# build extra copies of libfmmbem_hip.so here, ship them with the snapshot, select with FMMBEM_LIB in one gpurun call.
#
#   tools/ab_variants.sh kernels_m2l.hip FMMBEM_M2L_XCD_CHUNK 4 16 64      # -> build/variants/lib_FMMBEM_M2L_XCD_CHUNK_<v>.so
#   gpurun -- 'for v in 4 16 64; do FMMBEM_LIB=$PWD/build/variants/lib_FMMBEM_M2L_XCD_CHUNK_$v.so python bench.py --no-cpu-baseline --no-accuracy | tail -1; done'
#
# build/ is git-ignored; remove it afterwards (it travels with every gpurun snapshot).
set -e
src=$1; macro=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)
cs=$here/fmm-bem-relaxed_amd/csrc
make -C "$cs" > /dev/null
mkdir -p "$here/build/variants"
for v in "$@"; do
  obj=$here/build/variants/${src%.*}_${macro}_$v.o
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I"$cs" -I"$here/include" -D$macro=$v -c -o "$obj" "$cs/$src"
  objs=""
  for o in host_plan mesh_io kernels_near kernels_far kernels_m2l plan; do
    if [ "$o" = "${src%.*}" ]; then objs="$objs $obj"; else objs="$objs $cs/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$here/build/variants/lib_${macro}_$v.so" $objs
  echo "build/variants/lib_${macro}_$v.so"
done
