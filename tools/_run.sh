set -e
python -m pytest tests/test_gpu_parity.py tests/test_stokes.py tests/test_evaluators.py -x -q -k "matrix_free or shards or evaluator or traction" > gpurun_out/r03h_tests.log 2>&1 || { tail -40 gpurun_out/r03h_tests.log; exit 1; }
tail -2 gpurun_out/r03h_tests.log
for W in laplace stokes_rbc stokes_rbc_traction; do
python bench.py --matrix-free --workload $W --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r03h_bench_matfree_$W.json 2> gpurun_out/r03h_bench_matfree_$W.err || { tail -20 gpurun_out/r03h_bench_matfree_$W.err; exit 1; }
python -c "import json,sys; d=json.loads(open('gpurun_out/r03h_bench_matfree_$W.json').read().strip().splitlines()[-1]); print('$W', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['launch_ms'], d['plan_build_s'], d.get('rel_l2_vs_direct_sample'))"
done
