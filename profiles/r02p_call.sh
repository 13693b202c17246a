cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_configs_gpu.py tests/test_walk_gpu.py tests/test_trace_gpu.py -q -m gpu -x 2>&1 | grep -v "^W2026" | grep -E "passed|failed|rror" | tail -5
python profiles/r02p_specfill.py 2>&1 | grep -v "^W2026" | tee gpurun_out/r02p_specfill.txt
cd /tmp; rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py 45000 2 frame > /tmp/log.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -8
