# Round profile: rocprofv3 kernel trace + separate PMC passes for the default bench workload, summaries under gpurun_out/<tag>_*
# (copy what is to be kept into profiles/).  usage (on the GPU box, via gpurun):  bash tools/round_profile.sh r02h
# The program stands directly behind `--` (python3 itself: no env / shell hop under the profiler), PMC passes are their own runs.
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -o t -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 5 --warmup 1 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 5 --warmup 1 > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/${TAG}_sq -o s -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 4 --warmup 1 > $R/gpurun_out/${TAG}_sq.log 2>&1
cd $R
python tools/prof_summary.py kernel-stats gpurun_out/${TAG}_trace gpurun_out/${TAG}_kernel_stats.md > /dev/null
python tools/prof_summary.py pmc gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_pmc.json 1048576 1 > /dev/null
python tools/pmc_kernel.py gpurun_out/${TAG}_sq m2l > gpurun_out/${TAG}_pmc_sq_m2l.txt
python tools/pmc_kernel.py gpurun_out/${TAG}_sq near_spmv >> gpurun_out/${TAG}_pmc_sq_m2l.txt
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
head -16 gpurun_out/${TAG}_kernel_stats.md
cat gpurun_out/${TAG}_pmc_sq_m2l.txt
