# The reference's own examples/StokesBEM.cpp (compiled unmodified against include/fmmbem/compat, oracle/_ref/StokesBEM_ref) under its
# flags, on the GPU.  usage (GPU box): bash tools/reference_stokes_driver_flags.sh OUTDIR
OUT=${1:-gpurun_out/drv_stokes}
mkdir -p $OUT
SCRATCH=$(mktemp -d)      # the reference generators dump test.vert / test.face into the working directory
B=$PWD/oracle/_ref/StokesBEM_ref
run() { name=$1; shift; echo "== $name: $*"; (cd $SCRATCH && timeout -k 10 600 $B "$@") > $OUT/$name.txt 2>&1; echo "rc $?"; grep -i "Solver:\|Final residual\|rhs error\|Area error\|Fx\|drag\|setup\|solve :" $OUT/$name.txt | tail -8; }
run default      -recursions 4 -p 10
run fixed_p      -recursions 4 -p 10 -fixed_p
run pmin6        -recursions 4 -p 10 -pmin 6
run fgmres       -recursions 4 -p 10 -fgmres
run fgmres_diag  -recursions 4 -p 10 -fgmres -diag
run local        -recursions 4 -p 10 -local
run rbc          -p 8 -rbc 4
run cells2       -recursions 3 -p 8 -cells 2
run kfine        -recursions 4 -p 8 -k 4 -kfine 13
run nosparse     -recursions 3 -p 8 -disable_sparse
run tol1e7       -recursions 4 -p 10 -solver_tol 1e-7
