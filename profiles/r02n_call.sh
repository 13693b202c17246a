cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=4 2>&1 | grep -v "^W2026" | grep -E "passed|failed|rror|^[0-9.]+s " | tail -8 | tee gpurun_out/r02n_gpu_tests.txt
