# Where does an M2L wavefront wait?  Instruction-cache and issue counters of m2l_rot_kernel<P> on the bench workload.
# usage (GPU box): bash tools/pmc_m2l_ifetch.sh TAG "10 12 8"
set -e
TAG=${1:-r04}
ORDERS=${2:-10}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/$TAG/avail.txt 2>&1 || true
pick() { for c in "$@"; do if grep -qw "$c" $R/gpurun_out/$TAG/avail.txt; then printf "%s " $c; fi; done; }
A=$(pick SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL)
B=$(pick SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU)
C=$(pick SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_FLAT)
echo "A: $A"; echo "B: $B"; echo "C: $C"
for P in $ORDERS; do
  for S in A B C; do
    eval CS=\$$S
    [ -z "$CS" ] && continue
    rocprofv3 --pmc $CS --output-format csv -d $R/gpurun_out/$TAG/p${P}_$S -o s -- python3 $R/tools/rot_time.py --child --p $P --steps 3 > $R/gpurun_out/$TAG/p${P}_$S.log 2>&1
    (cd $R && python tools/pmc_kernel.py gpurun_out/$TAG/p${P}_$S m2l_rot_kernel) >> $R/gpurun_out/$TAG/summary.txt
  done
done
cat $R/gpurun_out/$TAG/summary.txt
