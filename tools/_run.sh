set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r03e_gputests.log 2>&1 || { tail -40 gpurun_out/r03e_gputests.log; exit 1; }
tail -2 gpurun_out/r03e_gputests.log
python tools/lowp_overlap.py "FMMBEM_FUSE_LEVELS=0,FMMBEM_P2M_STREAM=0" "FMMBEM_FUSE_LEVELS=1,FMMBEM_P2M_STREAM=0" "FMMBEM_FUSE_LEVELS=1,FMMBEM_P2M_STREAM=1" 2>&1 | tee gpurun_out/r03e_sweep.txt
python bench.py --no-cpu-baseline > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err
python -c "import json; d=json.loads(open('gpurun_out/r03e_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'])"
python bench.py --no-cpu-baseline --p 2 > gpurun_out/r03e_bench_p2.json 2> gpurun_out/r03e_bench_p2.err
python -c "import json; d=json.loads(open('gpurun_out/r03e_bench_p2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'])"
