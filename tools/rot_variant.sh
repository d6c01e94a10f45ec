#!/bin/bash
# Experiment build of the rotation M2L kernel: tools/rot_variant.sh NAME [extra hipcc flags, e.g. -DFMMBEM_ROT_ONLY=10]
# -> fmm-bem-relaxed_amd/variants/libfmmbem_hip_NAME.so (same ABI; select with FMMBEM_LIB=...).  Not part of the product build.
set -e
cd "$(dirname "$0")/../fmm-bem-relaxed_amd/csrc"
name=$1; shift
mkdir -p ../variants /tmp/rotvar
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000 \
  "$@" -c -o /tmp/rotvar/kr_$name.o kernels_m2l_rot.hip
# the same check as the product build: a variant with a DPP hazard or an early touch of a load in flight computes wrong
# numbers, and its timing means nothing
# (ROT_NOCHECK=1: for a timing-only experiment whose NUMBERS are wrong anyway -- never for a build the checker refuses because a
# load in flight is touched: the value touched may be an index or an address, and the unchecked p = 12 build of round 4 faulted on the GPU)
# The checker ALWAYS runs.  ROT_NOCHECK=1 tolerates only what a wait state would cure (DPP hazards: --nop-mask != 0x0) -- a report
# that holds an early touch of a load in flight stops the build whatever the variable says.
if ! python3 ../../tools/check_rot_isa.py /tmp/rotvar/kr_$name.o > /tmp/rotvar/kr_$name.isa 2>&1; then
  cat /tmp/rotvar/kr_$name.isa
  mask=$(python3 ../../tools/check_rot_isa.py --nop-mask /tmp/rotvar/kr_$name.isa)
  if [ -z "$ROT_NOCHECK" ] || [ "$mask" = "0x0" ] || grep -q "early touches of loads in flight [1-9]" /tmp/rotvar/kr_$name.isa; then
    echo "rot_variant: refused (ROT_NOCHECK only waives DPP hazards; mask $mask)"; exit 1
  fi
  echo "rot_variant: DPP hazards waived for a timing-only build (mask $mask): the NUMBERS of this library are wrong"
else
  cat /tmp/rotvar/kr_$name.isa
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/libfmmbem_hip_$name.so host_plan.o mesh_io.o kernels_near.o kernels_far.o kernels_m2l.o /tmp/rotvar/kr_$name.o kernels_m2m_rot.o kernels_l2l_rot.o kernels_shift.o krylov.o plan.o
echo built ../variants/libfmmbem_hip_$name.so
