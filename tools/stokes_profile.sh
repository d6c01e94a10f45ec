# The three rocprofv3 passes of tools/round_profile.sh for `bench.py --workload stokes_rbc` (config 4).
# usage: bash tools/stokes_profile.sh r02h [stokes_rbc_traction]      (second argument: the workload, default stokes_rbc)
set -e
TAG=${1:-r02}
WL=${2:-stokes_rbc}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stokes_trace -o t -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-accuracy > $R/gpurun_out/${TAG}_stokes_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_stokes_fetch -o f -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-accuracy --steps 5 --warmup 1 > $R/gpurun_out/${TAG}_stokes_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_stokes_write -o w -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-accuracy --steps 5 --warmup 1 > $R/gpurun_out/${TAG}_stokes_write.log 2>&1
cd $R
python tools/prof_summary.py kernel-stats gpurun_out/${TAG}_stokes_trace gpurun_out/${TAG}_stokes_kernel_stats.md | head -12
python tools/prof_summary.py pmc gpurun_out/${TAG}_stokes_fetch gpurun_out/${TAG}_stokes_write gpurun_out/${TAG}_stokes_pmc.json 524288 1 | tail -5
python bench.py --workload $WL > gpurun_out/${TAG}_bench_stokes.json 2> gpurun_out/${TAG}_bench_stokes.err
tail -c 400 gpurun_out/${TAG}_bench_stokes.json
