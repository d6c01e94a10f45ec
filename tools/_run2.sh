for th in 0 1; do
export FMMBEM_MULTI_ISSUE_THREADS=$th
echo "issuing threads: $th"
bash tools/_run.sh
done
unset FMMBEM_MULTI_ISSUE_THREADS
timeout -k 10 400 python -m pytest tests/test_gpu_multi_device.py -x -q -m gpu 2>&1 | tail -2
