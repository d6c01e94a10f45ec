set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02k_p2_trace -o t -- python3 $R/bench.py --p 2 --no-cpu-baseline --no-accuracy > $R/gpurun_out/r02k_p2_trace.log 2>&1
cd $R
python tools/prof_summary.py kernel-stats gpurun_out/r02k_p2_trace gpurun_out/r02k_p2_kernel_stats.md > /dev/null
head -24 gpurun_out/r02k_p2_kernel_stats.md
tail -c 700 gpurun_out/r02k_p2_trace.log
