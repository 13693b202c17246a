# round 6: trace/walk tests on the cleaned-up tree (classic writer with grouped stores, rules A-C), the tightened gradient test,
# gather tests with the straight-line kernel, and the in-process gather A/B
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_walk_gpu.py tests/test_parity_configs_gpu.py tests/test_trace_gpu.py tests/test_gather_gpu.py tests/test_train_gpu.py tests/test_reference_suite_gpu.py -m gpu -q -s 2>&1 | grep -v "^W2026" | grep "ReLU decisions\|passed\|failed\|Error\|assert" | cut -c1-700 > gpurun_out/r06g_tests.txt; cat gpurun_out/r06g_tests.txt
timeout 600 python profiles/r06g_ops_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06g_ops_ab.txt; cat gpurun_out/r06g_ops_ab.txt
