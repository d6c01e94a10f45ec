set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01n_stokes_trace -o t -- python3 $R/bench.py --workload stokes_rbc --no-cpu-baseline --no-accuracy > $R/gpurun_out/r01n_stokes_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01n_stokes_fetch -o f -- python3 $R/bench.py --workload stokes_rbc --no-cpu-baseline --no-accuracy --steps 5 --warmup 1 > $R/gpurun_out/r01n_stokes_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01n_stokes_write -o w -- python3 $R/bench.py --workload stokes_rbc --no-cpu-baseline --no-accuracy --steps 5 --warmup 1 > $R/gpurun_out/r01n_stokes_write.log 2>&1
cd $R
python tools/prof_summary.py kernel-stats gpurun_out/r01n_stokes_trace gpurun_out/r01n_stokes_kernel_stats.md | head -12
python tools/prof_summary.py pmc gpurun_out/r01n_stokes_fetch gpurun_out/r01n_stokes_write gpurun_out/r01n_stokes_pmc.json 524288 1 | tail -5
tail -1 gpurun_out/r01n_stokes_trace.log | cut -c1-200
