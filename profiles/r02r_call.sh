cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_build_gpu.py tests/test_trace_gpu.py tests/test_parity_configs_gpu.py tests/test_reference_suite_gpu.py -q -m gpu -x 2>&1 | grep -v "^W2026" | grep -E "passed|failed|rror|assert" | tail -6
python profiles/r02k_small_batch.py 2>&1 | grep -v "^W2026" | tee gpurun_out/r02r_leaf_width.txt
