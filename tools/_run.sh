set -e
python -m pytest tests/test_stokes.py tests/test_cpp_adapter.py -x -q -m gpu > gpurun_out/r03j_tests.log 2>&1 || { tail -40 gpurun_out/r03j_tests.log; exit 1; }
tail -2 gpurun_out/r03j_tests.log
