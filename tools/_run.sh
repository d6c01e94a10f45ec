set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r03i_gputests.log 2>&1 || { tail -40 gpurun_out/r03i_gputests.log; exit 1; }
tail -2 gpurun_out/r03i_gputests.log
python tools/lowp_overlap.py "FMMBEM_GRAPH=0" "FMMBEM_GRAPH=1" -- 1 2 3 10 2>&1 | tee gpurun_out/r03i_graph.txt
echo "--- shard_time, graphs off"; FMMBEM_GRAPH=0 python tools/shard_time.py 2>&1 | tail -4 | tee gpurun_out/r03i_shard_time.txt
echo "--- shard_time, graphs on"; FMMBEM_GRAPH=1 python tools/shard_time.py 2>&1 | tail -4 | tee -a gpurun_out/r03i_shard_time.txt
