cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_configs_gpu.py tests/test_walk_gpu.py tests/test_trace_gpu.py -q -m gpu -x 2>&1 | grep -v "^W2026" | grep -E "passed|failed|rror" | tail -5
python profiles/r02l_fatlog.py 2>&1 | grep -v "^W2026" | tee gpurun_out/r02l_fatlog.txt
