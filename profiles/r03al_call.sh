# round 3, call AL: training buffers quad-major [F/4][n][4]: 16-byte stores and tile loads
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TETRANERF_HIP_POISON=1 timeout 900 python -m pytest tests/test_train_gpu.py tests/test_plugin_gpu.py tests/test_render_gpu.py -x -q -m gpu > gpurun_out/r03al_tests.txt 2>&1; tail -5 gpurun_out/r03al_tests.txt
cd /tmp
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > $GRAFT_REPO_ROOT/gpurun_out/r03al_train_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r03al_train_kernel_stats.txt
cd $GRAFT_REPO_ROOT; grep -v "^W2026\|^E2026" gpurun_out/r03al_train_once.txt | tail -1; head -16 gpurun_out/r03al_train_kernel_stats.txt | cut -c1-80,112-160
