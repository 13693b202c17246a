# quick check of a k_mlp_backward variant: training parity tests + per-kernel time of the training loop
set -x
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > $GRAFT_REPO_ROOT/gpurun_out/r03p_train_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r03p_train_kernel_stats.txt
cd $GRAFT_REPO_ROOT; grep -v "^W2026\|^E2026" gpurun_out/r03p_train_once.txt | tail -2; head -8 gpurun_out/r03p_train_kernel_stats.txt | cut -c1-80,112-160
