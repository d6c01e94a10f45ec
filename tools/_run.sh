set -e
python -m pytest tests/test_gpu_parity.py tests/test_stokes.py tests/test_edge_cases.py tests/test_random_meshes.py -x -q -m gpu > gpurun_out/r03n_tests.log 2>&1 || { tail -40 gpurun_out/r03n_tests.log; exit 1; }
tail -2 gpurun_out/r03n_tests.log
python tools/lowp_overlap.py --stage=l2p "FMMBEM_GRAPH=0" -- 2 6 10 12 2>&1 | tee gpurun_out/r03n_l2p.txt
python bench.py --workload stokes_rbc --no-cpu-baseline --no-accuracy > gpurun_out/r03n_stokes.json 2>/dev/null
python -c "import json; d=json.loads(open('gpurun_out/r03n_stokes.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms'])"
