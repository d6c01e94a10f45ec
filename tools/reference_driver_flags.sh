# The reference's own examples/LaplaceBEM.cpp (compiled unmodified against include/fmmbem/compat, oracle/_ref/LaplaceBEM_ref) run with
# the flags its help text lists, on the GPU.  usage (GPU box): bash tools/reference_driver_flags.sh OUTDIR
OUT=${1:-gpurun_out/drv}
mkdir -p $OUT
SCRATCH=$(mktemp -d)      # the reference generators dump test.vert / test.face into the working directory
B=$PWD/oracle/_ref/LaplaceBEM_ref
run() { name=$1; shift; echo "== $name: $*"; (cd $SCRATCH && timeout -k 10 300 $B "$@") > $OUT/$name.txt 2>&1; echo "rc $?"; grep -i "iteration\|error\|Solver:\|Precond\|time" $OUT/$name.txt | tail -8; }
run default      -recursions 6 -p 12 -theta 0.5
run fixed_p      -recursions 6 -p 12 -fixed_p
run second_kind  -recursions 6 -p 12 -second_kind
run diagonal     -recursions 6 -p 12 -diagonal
run fgmres       -recursions 6 -p 12 -fgmres
run fgmres_bd    -recursions 6 -p 12 -fgmres -diagonal
run local        -recursions 6 -p 12 -local
run tol1e8       -recursions 6 -p 12 -solver_tol 1e-8
run k4           -recursions 5 -p 8 -k 4
run ncrit32      -recursions 6 -p 10 -ncrit 32 -theta 0.4
