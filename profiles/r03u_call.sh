# round 3, call U: k_mlp_forward with ping-pong weight stages: MLP / render / training parity tests, training breakdown, render bench
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_train_gpu.py tests/test_plugin_gpu.py tests/test_sharding_gpu.py -x -q -m gpu 2>&1 | tail -3
cd /tmp
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > $GRAFT_REPO_ROOT/gpurun_out/r03u_train_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r03u_train_kernel_stats.txt
cd $GRAFT_REPO_ROOT; grep -v "^W2026\|^E2026" gpurun_out/r03u_train_once.txt | tail -1; head -9 gpurun_out/r03u_train_kernel_stats.txt | cut -c1-80,112-160
timeout 600 python profiles/r02j_render_bench.py 2>&1 | grep "^frame\|^C4" > gpurun_out/r03u_render_bench.txt; cat gpurun_out/r03u_render_bench.txt
