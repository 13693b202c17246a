cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do echo "=== default run $i"; timeout 600 python -m pytest "tests/test_parity_configs_gpu.py::test_adversarial_meshes_bit_exact" tests/test_sharding_gpu.py tests/test_render_gpu.py tests/test_sort_vectors.py -q -m gpu 2>&1 | grep -v "^W2026" | tail -6; done
timeout 600 python -m pytest tests/test_parity_configs_gpu.py tests/test_trace_gpu.py tests/test_walk_gpu.py -q -m gpu 2>&1 | grep -v "^W2026" | tail -6
python profiles/c5_once.py 150000 3 1048576 2>&1 | grep -v "^W2026" | tail -2
cd /tmp; rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py 150000 3 1048576 > /tmp/log.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/pp -name "*.db" | head -1) | head -9 | cut -c1-50,112-160
python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -7
