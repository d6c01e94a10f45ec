set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1
tail -2 gpurun_out/t.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
tail -2 gpurun_out/smoke.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01m_trace -o t -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01m_bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01m_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 5 --warmup 1 > $R/gpurun_out/r01m_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01m_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 5 --warmup 1 > $R/gpurun_out/r01m_write.log 2>&1
cd $R
python tools/prof_summary.py kernel-stats gpurun_out/r01m_trace gpurun_out/r01m_kernel_stats.md > /dev/null
python tools/prof_summary.py pmc gpurun_out/r01m_fetch gpurun_out/r01m_write gpurun_out/r01m_pmc.json 1048576 1 > /dev/null
cp gpurun_out/r01m_pmc.json profiles/pmc_near_spmv.json
python bench.py > gpurun_out/r01m_bench.log 2>&1
tail -1 gpurun_out/r01m_bench.log
head -14 gpurun_out/r01m_kernel_stats.md
