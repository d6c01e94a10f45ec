set -e
for W in stokes_rbc stokes_rbc_traction; do
python bench.py --workload $W > gpurun_out/r03l_bench_$W.json 2> gpurun_out/r03l_bench_$W.err || { tail -20 gpurun_out/r03l_bench_$W.err; exit 1; }
python -c "import json,sys; d=json.loads(open('gpurun_out/r03l_bench_$W.json').read().strip().splitlines()[-1]); print('$W', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('cpu_baseline'))"
done
python tools/solve_config5.py > gpurun_out/r03l_config5.json 2> gpurun_out/r03l_config5.err || { tail -20 gpurun_out/r03l_config5.err; exit 1; }
tail -c 1500 gpurun_out/r03l_config5.json
