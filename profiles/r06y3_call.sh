set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_walk_gpu.py tests/test_parity_configs_gpu.py tests/test_trace_gpu.py tests/test_build_gpu.py tests/test_reference_suite_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06y_tests2.txt; cat gpurun_out/r06y_tests2.txt
